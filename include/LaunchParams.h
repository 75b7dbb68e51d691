// LaunchParams.h -- drop-in for PT_sv5_/LaunchParams.h:49-91.  248 bytes, same member names, so
// caller code such as `sample.launchParams.frame.c.x = ...` (main.cpp:366-367) compiles unchanged.
#pragma once
#include <cstddef>
#include "Material.h"
#include "Probe.h"

enum { RAY_TYPE_RADIANCE = 0, RAY_TYPE_OCCLUSION = 1, RAY_TYPE_COUNT = 2 };
typedef unsigned long long OptixTraversableHandle;   // here: the scene handle of fovpt_set_scene

struct LaunchParams {
    struct {
        float4* accum_buffer = nullptr;
        uchar4* frame_buffer = nullptr;
        float4* color_buffer = nullptr;
        float4* normal_buffer = nullptr;
        float4* albedo_buffer = nullptr;
        int2 size = make_int2(0, 0);
        unsigned int subframe_index = 0;
        uint3 factor = make_uint3(1, 1, 1);
        int fillSize = 1;
        uint2 c = make_uint2(0, 0);
        float r_inner = 0.0f, r_outer = 0.0f;
        uint2 offset = make_uint2(0, 0);
        unsigned int redraw = 0;
    } frame;
    struct {
        float3 eye, U, V, W;
    } camera;
    unsigned int samples_per_launch = 0;
    OptixTraversableHandle traversable = 0;
    Probe probe = {};
    int2 viewportSize = make_int2(0, 0);
    float white = 0.0f;
};
static_assert(sizeof(LaunchParams) == sizeof(fovpt_launch_params), "LaunchParams must stay 248 bytes");
static_assert(offsetof(LaunchParams, camera) == 104 && offsetof(LaunchParams, traversable) == 160 &&
              offsetof(LaunchParams, probe) == 168 && offsetof(LaunchParams, viewportSize) == 232, "LaunchParams offsets");
