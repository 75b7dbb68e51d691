// Model.h -- drop-in for PT_sv5_/Model.h:10-43 (scene layout consumed by the SampleRenderer ctor).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
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

inline Model* modelFromHandle(fovpt_model* h);

// glTF 2.0 (.gltf / .glb) -> Model with sutil::Scene's node rules (sutil/Scene.cpp:109-442), one mesh per triangle primitive in
// world space: the library's host-side loader (fovpt_model_load_gltf), no tinygltf needed.  Throws std::runtime_error.
inline Model* loadGLTF(const std::string& gltfFile)
{
    fovpt_model* h = nullptr;
    if (fovpt_model_load_gltf(gltfFile.c_str(), &h) != FOVPT_OK) throw std::runtime_error(fovpt_last_error(nullptr));
    return modelFromHandle(h);
}

// loadOBJ of Model.cpp:138-217 (declared Model.h:42): the scene as the reference's loader builds it -- one mesh per
// (shape, material), per-shape vertex and texture maps, Kd -> color, Ke -> emission, textures as stbi_load gives them,
// mirrored along y.  The work is done by the library's host-side loader (fovpt_model_load_obj, csrc/model_loader.cpp);
// throws std::runtime_error like the reference (:160-162).  Link against libfovpt.so.
inline Model* loadOBJ(const std::string& objFile)
{
    fovpt_model* h = nullptr;
    if (fovpt_model_load_obj(objFile.c_str(), &h) != FOVPT_OK) throw std::runtime_error(fovpt_last_error(nullptr));
    return modelFromHandle(h);
}

// copies what the library's loader built into the reference's Model / TriangleMesh / Texture (Model.h:10-43) and frees the handle
inline Model* modelFromHandle(fovpt_model* h)
{
    int nm = 0, nt = 0;
    fovpt_model_counts(h, &nm, &nt);
    Model* model = new Model;
    for (int k = 0; k < nm; k++) {
        fovpt_model_mesh d;
        fovpt_model_get_mesh(h, k, &d);
        TriangleMesh* mesh = new TriangleMesh;
        mesh->vertex.resize(d.num_vertices); mesh->normal.resize(d.num_normals); mesh->texcoord.resize(d.num_texcoords); mesh->index.resize(d.num_triangles);
        if (d.num_vertices) std::memcpy(mesh->vertex.data(), d.vertex, sizeof(float3) * d.num_vertices);
        if (d.num_normals) std::memcpy(mesh->normal.data(), d.normal, sizeof(float3) * d.num_normals);
        for (uint32_t i = 0; i < d.num_texcoords; i++) mesh->texcoord[i] = make_float2(d.texcoord[2 * i], d.texcoord[2 * i + 1]);
        if (d.num_triangles) std::memcpy(mesh->index.data(), d.index, sizeof(uint3) * d.num_triangles);
        static_assert(sizeof(float3) == sizeof(fovpt_float3) && sizeof(uint3) == sizeof(fovpt_uint3), "vector layouts");
        std::memcpy(&mesh->material, &d.material, sizeof(Material));
        mesh->diffuseTextureID = d.diffuse_texture_id;
        model->meshes.push_back(mesh);
    }
    for (int k = 0; k < nt; k++) {
        const uint32_t* px = nullptr;
        int w = 0, hh = 0;
        fovpt_model_get_texture(h, k, &px, &w, &hh);
        Texture* t = new Texture;
        t->resolution = make_int2(w, hh);
        t->pixel = new uint32_t[(size_t)w * hh];
        std::memcpy(t->pixel, px, sizeof(uint32_t) * (size_t)w * hh);
        model->textures.push_back(t);
    }
    fovpt_model_destroy(h);
    return model;
}
