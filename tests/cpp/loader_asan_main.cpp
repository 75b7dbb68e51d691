// Test driver (tests/test_loaders_cpu.py::test_oversized_image_headers_under_asan): the library's host-side loader,
// csrc/model_loader.cpp, compiled with g++ -fsanitize=address together with this file.  Feeds every file named on the
// command line through fovpt_image_load_float4 (any other extension), fovpt_image_load_rgba8 (.jpg), fovpt_model_load_obj (.obj)
// or fovpt_model_load_gltf (.gltf / .glb) and prints the return codes; AddressSanitizer aborts the process on any out-of-bounds access.
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/fovpt.h"

static std::string g_err;
void fovpt_internal_set_error(const char* text) { g_err = text ? text : ""; }     // (lives in fovpt_api.hip in the library)

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; i++) {
        const std::string f = argv[i];
        g_err.clear();
        int rc;
        if (f.size() > 4 && f.substr(f.size() - 4) == ".obj") {
            fovpt_model* m = nullptr;
            rc = fovpt_model_load_obj(f.c_str(), &m);
            int nm = 0, nt = 0;
            if (rc == 0) { fovpt_model_counts(m, &nm, &nt); fovpt_model_destroy(m); }
            printf("%s obj rc=%d meshes=%d textures=%d\n", f.c_str(), rc, nm, nt);
        } else if ((f.size() > 5 && f.substr(f.size() - 5) == ".gltf") || (f.size() > 4 && f.substr(f.size() - 4) == ".glb")) {
            // every array the model hands out is read to its stated end: a short one trips the sanitizer here, as it would in
            // fovpt_set_scene
            fovpt_model* m = nullptr;
            rc = fovpt_model_load_gltf(f.c_str(), &m);
            int nm = 0, nt = 0;
            double sum = 0.0;
            if (rc == 0) {
                fovpt_model_counts(m, &nm, &nt);
                for (int k = 0; k < nm; k++) {
                    fovpt_model_mesh M;
                    if (fovpt_model_get_mesh(m, k, &M)) continue;
                    for (uint32_t v = 0; v < M.num_vertices; v++) {
                        sum += M.vertex[v].x;
                        if (M.texcoord) sum += M.texcoord[2 * v] + M.texcoord[2 * v + 1];      // one per VERTEX, as fovpt_set_scene reads them
                    }
                    for (uint32_t t = 0; t < M.num_triangles; t++) sum += (double)M.index[t].x + M.index[t].y + M.index[t].z;
                }
                for (int k = 0; k < nt; k++) {
                    const uint32_t* px = nullptr; int w = 0, h = 0;
                    if (fovpt_model_get_texture(m, k, &px, &w, &h) == 0 && w > 0 && h > 0) sum += px[0] + px[(size_t)w * h - 1];
                }
                fovpt_model_destroy(m);
            }
            printf("%s gltf rc=%d meshes=%d textures=%d sum=%g\n", f.c_str(), rc, nm, nt, sum);
        } else if ((f.size() > 4 && f.substr(f.size() - 4) == ".jpg")) {
            int w = 0, h = 0;
            uint32_t* px = nullptr;
            rc = fovpt_image_load_rgba8(f.c_str(), &w, &h, &px);
            unsigned long long sum = 0;
            if (rc == 0) { for (size_t k = 0; k < (size_t)w * h; k++) sum += px[k]; fovpt_image_free_rgba8(px); }
            printf("%s jpeg rc=%d %dx%d sum=%llu\n", f.c_str(), rc, w, h, sum);
        } else {
            int w = 0, h = 0;
            fovpt_float4* px = nullptr;
            rc = fovpt_image_load_float4(f.c_str(), &w, &h, &px);
            printf("%s image rc=%d %dx%d\n", f.c_str(), rc, w, h);
            if (rc == 0) fovpt_image_free(px);
        }
    }
    return 0;
}
