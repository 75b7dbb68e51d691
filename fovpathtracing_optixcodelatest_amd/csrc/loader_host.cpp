// loader_host.cpp -- the two symbols model_loader.cpp needs from the rest of the library, for libfovpt_loader.so: the scene and
// image ingestion of the C ABI (fovpt_model_*, fovpt_image_*; include/fovpt.h) as a HOST-ONLY shared object -- no HIP runtime, no
// GPU -- for tools that only read files (asset pipelines, the python package's loaders on a machine without ROCm).  libfovpt.so
// carries the same loader code; in it fovpt_api.hip provides these two.
#include <string>

#include "../../include/fovpt.h"

static std::string g_loader_error;
void fovpt_internal_set_error(const char* text) { g_loader_error = text ? text : ""; }

extern "C" const char* fovpt_last_error(const fovpt_ctx*) { return g_loader_error.c_str(); }     // (there is no context in this library)
