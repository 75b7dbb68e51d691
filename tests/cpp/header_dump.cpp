// Dumps what the drop-in host headers (include/Material.h, include/Model.h) produce, as raw bytes on stdout, for
// tests/test_ref_pin_cpu.py to compare with the reference's own Material / TriangleMesh / addBox (ref_vectors.npz).
//   [104 B default Material][int32 default diffuseTextureID][36*3 float vertex][36*3 float normal][36*2 float texcoord]
//   [12*3 uint32 index][104 B box material][int32 box diffuseTextureID]
#include <cstdio>
#include <cstring>
#include "Model.h"

int main(int argc, char** argv)
{
    float c[3] = {0, 0, 0}, h[3] = {1, 1, 1};
    if (argc == 7) for (int k = 0; k < 3; k++) { sscanf(argv[1 + k], "%f", &c[k]); sscanf(argv[4 + k], "%f", &h[k]); }
    Material m;
    fwrite(&m, sizeof(m), 1, stdout);
    TriangleMesh tm;
    fwrite(&tm.diffuseTextureID, 4, 1, stdout);
    Model model;
    Material mat;
    addBox(&model, mat, make_float3(c[0], c[1], c[2]), make_float3(h[0], h[1], h[2]));
    const TriangleMesh* b = model.meshes[0];
    if (b->vertex.size() != 36 || b->normal.size() != 36 || b->texcoord.size() != 36 || b->index.size() != 12) return 2;
    fwrite(b->vertex.data(), sizeof(float3), 36, stdout);
    fwrite(b->normal.data(), sizeof(float3), 36, stdout);
    fwrite(b->texcoord.data(), sizeof(float2), 36, stdout);
    fwrite(b->index.data(), sizeof(uint3), 12, stdout);
    fwrite(&b->material, sizeof(Material), 1, stdout);
    fwrite(&b->diffuseTextureID, 4, 1, stdout);
    return 0;
}
