// Model.h -- drop-in for PT_sv5_/Model.h:10-43 (scene layout consumed by the SampleRenderer ctor).
#pragma once
#include <cstdint>
#include <vector>
#include "Material.h"

struct TriangleMesh {
    std::vector<float3> vertex;
    std::vector<float3> normal;      // kept for layout; the path shades with the face normal
    std::vector<float2> texcoord;
    std::vector<uint3> index;
    Material material;
    int diffuseTextureID{1};         // the reference's default; loaders overwrite it, set -1 for procedural meshes
};

struct Texture {
    ~Texture() { delete[] pixel; }
    uint32_t* pixel{nullptr};
    int2 resolution = make_int2(-1, -1);
};

struct Model {
    ~Model()
    {
        for (auto* m : meshes) delete m;
        for (auto* t : textures) delete t;
    }
    std::vector<TriangleMesh*> meshes;
    std::vector<Texture*> textures;
};

// addBox of Model.cpp:219-291: 12 triangles, 36 unshared vertices, untextured.
inline void addBox(Model* model, Material& mat, const float3& pos, const float3& extend)
{
    const float sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1}, sz[8] = {1, 1, 1, 1, -1, -1, -1, -1};
    float3 P[8];                     // A B C D (front, z+) E F G H (back, z-)
    for (int k = 0; k < 8; k++) P[k] = make_float3(sx[k] * extend.x + pos.x, sy[k] * extend.y + pos.y, sz[k] * extend.z + pos.z);
    enum { A, B, C, D, E, F, G, H };
    const int tri[12][3] = {{A, B, C}, {A, C, D}, {E, H, G}, {E, G, F}, {E, A, D}, {E, D, H},
                            {B, F, G}, {B, G, C}, {D, C, G}, {D, G, H}, {E, A, B}, {E, B, F}};
    const float3 nrm[6] = {make_float3(0, 0, 1), make_float3(0, 0, -1), make_float3(-1, 0, 0),
                           make_float3(1, 0, 0), make_float3(0, 1, 0), make_float3(0, -1, 0)};
    TriangleMesh* mesh = new TriangleMesh;
    for (int t = 0; t < 12; t++) {
        for (int v = 0; v < 3; v++) {
            mesh->vertex.push_back(P[tri[t][v]]);
            mesh->normal.push_back(nrm[t / 2]);
            mesh->texcoord.push_back(make_float2(0.f, 0.f));
        }
        mesh->index.push_back(make_uint3(3 * t, 3 * t + 1, 3 * t + 2));
    }
    mesh->material = mat;
    mesh->diffuseTextureID = -1;     // the reference leaves the default 1 here and reads textureObjects[1]
    model->meshes.push_back(mesh);
}
