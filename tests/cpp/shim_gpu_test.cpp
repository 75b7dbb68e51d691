// Uses the drop-in C++ API the way PT_sv5_/main.cpp:297-306,423 does, on a box scene built with
// addBox (or, with a second argument, on the OBJ file it names, through loadOBJ), and writes the rgba8 frame to a file
// for the python test to compare with the oracle.
#include <cstdio>
#include <vector>
#include "SimplePathtracer.h"

struct HostTarget {                      // stands in for sutil::CUDAOutputBuffer<uint32_t>
    uint32_t* dev;
    uint32_t* map() { return dev; }
    void unmap() {}
};

int main(int argc, char** argv)
{
    const char* out = argc > 1 ? argv[1] : "shim_out.bin";
    try {
        Model* model;
        if (argc > 2) {
            model = loadOBJ(argv[2]);                 // main.cpp:130-145: the scene from an OBJ file
        } else {
            model = new Model;
            Material grey; grey.color = make_float3(0.7f, 0.7f, 0.7f); grey.emission = make_float3(0.0f);
            Material red; red.color = make_float3(0.8f, 0.1f, 0.1f); red.emission = make_float3(0.0f);
            addBox(model, grey, make_float3(0, -1.0f, 0), make_float3(6, 0.5f, 6));
            addBox(model, red, make_float3(0, 0.5f, 0), make_float3(1, 1, 1));
        }
        const int2 fbSize = make_int2(160, 96);
        std::vector<float4> sky((size_t)fbSize.x * fbSize.y, make_float4(2.5f, 2.5f, 2.5f, 1.0f));   // loadColor, main.cpp:175-187
        ProbeData probe;
        if (argc > 3) {                               // loadProbe, main.cpp:160-171, without stb_image
            float4* data = nullptr;
            int resX = 0, resY = 0;
            if (fovpt_image_load_float4(argv[3], &resX, &resY, (fovpt_float4**)&data)) throw std::runtime_error(fovpt_last_error(nullptr));
            probe.width = resX; probe.height = resY; probe.data = data;
        } else {
            probe.width = fbSize.x; probe.height = fbSize.y; probe.data = sky.data();
        }
        probe.BuildCDF();
        sutil::Camera camera(make_float3(4, 3, 6), make_float3(0, 0.5f, 0), make_float3(0, 1, 0), 45.0f, fbSize.x / float(fbSize.y));

        SampleRenderer sample(model);
        sample.resize(fbSize);
        sample.setCamera(camera);
        sample.setProbe(probe);
        fovpt_config cfg = sample.config();
        cfg.r_inner = 12; cfg.r_outer = 36; cfg.spp_periphery = 1; cfg.spp_middle = 2; cfg.spp_fovea = 8;
        sample.setConfig(cfg);
        sample.launchParams.frame.c.x = fbSize.x / 2;
        sample.launchParams.frame.c.y = fbSize.y / 2;
        sample.launchParams.frame.subframe_index = 0;
        sample.render();
        std::vector<uint32_t> pixels((size_t)fbSize.x * fbSize.y);
        sample.downloadPixels(pixels.data());
        FILE* f = fopen(out, "wb");
        fwrite(pixels.data(), 4, pixels.size(), f);
        fclose(f);
        printf("ok subframe_index=%u\n", sample.launchParams.frame.subframe_index);
        bool threw = false;
        try { ProbeData bad; sample.setProbe(bad); } catch (const std::runtime_error&) { threw = true; }
        if (!threw) { printf("setProbe(invalid) did not throw\n"); return 2; }
        delete model;
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
