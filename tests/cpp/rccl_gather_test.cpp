// The multi-GPU path of a C++ host (include/fovpt.h, fovpt_comm_* / fovpt_gather_frame), on ONE GPU with a communicator of
// one rank: renders a frame with the drop-in SampleRenderer, gathers it through plan -> pack -> RCCL send / recv -> unpack
// on the library's stream, and writes both the renderer's own frame and the gathered one for the python test to compare.
// With N ranks the same program runs as N processes (multi-GPU):
//     rccl_gather_test <out> <rank> <world> <id file> [device]
// rank 0 creates the communicator's unique id and publishes it through <id file> (written under another name, then renamed); the
// other ranks wait for the file; every rank renders its tiles on its own GPU (device = rank unless given) and takes part in the
// gather; rank 0 then renders the unsharded frame too and writes it beside the gathered one.  Unmeasured on a multi-GPU node so far
// (no such node has been available to the builder): the one-rank form is what tests/test_gpu_extra.py runs.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "SimplePathtracer.h"

// (the four HIP runtime calls this program needs, declared by hand: hip_runtime_api.h brings its own float3 / make_float3,
// which are the shim's job here)
extern "C" {
int hipMalloc(void** ptr, size_t size);
int hipFree(void* ptr);
int hipMemset(void* dst, int value, size_t size);
int hipMemcpy(void* dst, const void* src, size_t size, int kind);
}
enum { hipSuccess = 0, hipMemcpyDeviceToHost = 2 };

int main(int argc, char** argv)
{
    const char* out = argc > 1 ? argv[1] : "rccl_out.bin";
    const int rank = argc > 3 ? atoi(argv[2]) : 0, world = argc > 3 ? atoi(argv[3]) : 1;
    const char* id_file = argc > 4 ? argv[4] : nullptr;
    const int device = argc > 5 ? atoi(argv[5]) : rank;
    if (rank < 0 || world < 1 || rank >= world || (world > 1 && !id_file)) { printf("usage: %s <out> [<rank> <world> <id file> [device]]\n", argv[0]); return 2; }
    try {
        Model* model = new Model;
        Material grey; grey.color = make_float3(0.7f, 0.7f, 0.7f); grey.emission = make_float3(0.0f);
        Material red; red.color = make_float3(0.8f, 0.1f, 0.1f); red.emission = make_float3(0.0f);
        addBox(model, grey, make_float3(0, -1.0f, 0), make_float3(6, 0.5f, 6));
        addBox(model, red, make_float3(0, 0.5f, 0), make_float3(1, 1, 1));
        const int2 fbSize = make_int2(160, 96);
        std::vector<float4> sky((size_t)fbSize.x * fbSize.y, make_float4(2.5f, 2.5f, 2.5f, 1.0f));
        ProbeData probe;
        probe.width = fbSize.x; probe.height = fbSize.y; probe.data = sky.data();
        probe.BuildCDF();
        sutil::Camera camera(make_float3(4, 3, 6), make_float3(0, 0.5f, 0), make_float3(0, 1, 0), 45.0f, fbSize.x / float(fbSize.y));

        SampleRenderer sample(model, world > 1 ? device : 0);
        sample.resize(fbSize);
        sample.setCamera(camera);
        sample.setProbe(probe);
        fovpt_config cfg = sample.config();
        cfg.r_inner = 12; cfg.r_outer = 36; cfg.spp_periphery = 1; cfg.spp_middle = 2; cfg.spp_fovea = 8;
        cfg.rank = rank; cfg.world = world;
        sample.setConfig(cfg);
        sample.launchParams.frame.c.x = fbSize.x / 2;
        sample.launchParams.frame.c.y = fbSize.y / 2;

        char id[FOVPT_COMM_ID_BYTES];
        if (rank == 0) {
            if (fovpt_comm_get_unique_id(id)) throw std::runtime_error(fovpt_last_error(nullptr));
            if (id_file) {                                      // publish: complete file under a temporary name, then rename
                const std::string tmp = std::string(id_file) + ".tmp";
                FILE* f = fopen(tmp.c_str(), "wb");
                if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) throw std::runtime_error("cannot write the id file");
                fclose(f);
                if (rename(tmp.c_str(), id_file) != 0) throw std::runtime_error("cannot publish the id file");
            }
        } else {
            bool got = false;
            for (int tries = 0; tries < 1200 && !got; tries++) {   // up to two minutes for rank 0 to get there
                FILE* f = fopen(id_file, "rb");
                if (f) { got = fread(id, 1, sizeof(id), f) == sizeof(id); fclose(f); }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            if (!got) throw std::runtime_error("no unique id from rank 0");
        }
        sample.commInit(id, rank, world);

        uint32_t* gathered = nullptr;
        const size_t n = (size_t)fbSize.x * fbSize.y;
        if (hipMalloc((void**)&gathered, n * 4) != hipSuccess || hipMemset(gathered, 0, n * 4) != hipSuccess) throw std::runtime_error("hipMalloc");
        for (int frame = 0; frame < 3; frame++) {              // frames back to back: the gather of one runs beside the next
            sample.launchParams.frame.subframe_index = 0;
            sample.renderAsync();
            sample.gatherFrame(0, gathered);
        }
        std::vector<uint32_t> pixels(n), full(n);
        sample.downloadPixels(pixels.data());                   // synchronises
        if (hipMemcpy(full.data(), gathered, n * 4, hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("hipMemcpy");
        if (rank != 0) { printf("ok rank %d of %d took part\n", rank, world); (void)hipFree(gathered); delete model; return 0; }
        if (world > 1) {                                        // what the gathered frame must equal: the frame of one GPU
            cfg.rank = 0; cfg.world = 1;
            sample.setConfig(cfg);
            sample.launchParams.frame.subframe_index = 0;
            sample.render();
            sample.downloadPixels(pixels.data());
        }
        FILE* f = fopen(out, "wb");
        fwrite(pixels.data(), 4, n, f);
        fwrite(full.data(), 4, n, f);
        fclose(f);
        size_t same = 0;
        for (size_t i = 0; i < n; i++) same += pixels[i] == full[i];
        printf("ok gathered %zu of %zu pixels identical\n", same, n);
        (void)hipFree(gathered);
        delete model;
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
