// Probe.h -- drop-in for PT_sv5_/Probe.h (host ProbeData) and the device Probe of Probe.cuh:6-21.
#pragma once
#include <stdexcept>
#include <vector>
#include "fovpt.h"
#include "fovpt_vec.h"

typedef float4 Color;                 // maths.h:34
typedef fovpt_probe Probe;            // device view: 64 bytes, pointers into HBM

struct ProbeData {
    int width = 0;
    int height = 0;
    Color* data = nullptr;            // caller-owned, width*height texels
    float3 offset = make_float3(0.0f);
    bool valid = false;
    float* pdfValuesX = nullptr;
    float* cdfValuesX = nullptr;
    float* pdfValuesY = nullptr;
    float* cdfValuesY = nullptr;

    // Row-conditional and marginal CDF over texel luminance, sequential fp32 sums (Probe.h:29-77).
    // Runs on the host in the reference too; implemented once, in libfovpt.
    void BuildCDF()
    {
        const size_t n = (size_t)width * (size_t)height;
        pdfValuesX = new float[n]; cdfValuesX = new float[n];
        pdfValuesY = new float[height]; cdfValuesY = new float[height];
        if (fovpt_probe_build_cdf(width, height, (const fovpt_float4*)data, pdfValuesX, cdfValuesX, pdfValuesY, cdfValuesY) != FOVPT_OK)
            throw std::runtime_error("ProbeData::BuildCDF: bad probe");
        valid = true;
    }
};
