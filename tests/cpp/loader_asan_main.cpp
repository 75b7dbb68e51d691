// Test driver (tests/test_loaders_cpu.py::test_oversized_image_headers_under_asan): the library's host-side loader,
// csrc/model_loader.cpp, compiled with g++ -fsanitize=address together with this file.  Feeds every file named on the
// command line through fovpt_image_load_float4 (any extension) or fovpt_model_load_obj (.obj) and prints the return
// codes; AddressSanitizer aborts the process on any out-of-bounds access.
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
