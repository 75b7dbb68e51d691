// loadOBJ of include/Model.h the way PT_sv5_/main.cpp:130-145 calls it; dumps the Model as raw bytes on stdout for
// tests/test_ref_pin_cpu.py to compare with the reference's own loadOBJ (ref_loaders.npz):
//   int32 num_meshes, num_textures; per mesh: int32 nv, nn, nt, ni, textureID; 104 B material; vertex, normal, texcoord,
//   index arrays; per texture: int32 w, h; pixels.  A second argument that cannot be loaded must throw std::runtime_error.
#include <cstdio>
#include "Model.h"

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    if (argc > 2) {
        try { Model* bad = loadOBJ(argv[2]); delete bad; return 3; }
        catch (const std::runtime_error& e) { fprintf(stderr, "%s\n", e.what()); }
    }
    Model* model = loadOBJ(argv[1]);
    const int counts[2] = {(int)model->meshes.size(), (int)model->textures.size()};
    fwrite(counts, 4, 2, stdout);
    for (const TriangleMesh* m : model->meshes) {
        const int n[5] = {(int)m->vertex.size(), (int)m->normal.size(), (int)m->texcoord.size(), (int)m->index.size(), m->diffuseTextureID};
        fwrite(n, 4, 5, stdout);
        fwrite(&m->material, sizeof(Material), 1, stdout);
        fwrite(m->vertex.data(), sizeof(float3), m->vertex.size(), stdout);
        fwrite(m->normal.data(), sizeof(float3), m->normal.size(), stdout);
        fwrite(m->texcoord.data(), sizeof(float2), m->texcoord.size(), stdout);
        fwrite(m->index.data(), sizeof(uint3), m->index.size(), stdout);
    }
    for (const Texture* t : model->textures) {
        fwrite(&t->resolution, 4, 2, stdout);
        fwrite(t->pixel, 4, (size_t)t->resolution.x * t->resolution.y, stdout);
    }
    delete model;
    return 0;
}
