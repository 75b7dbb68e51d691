// fovpt_vec.h -- the handful of CUDA vector types the reference's host API is written in
// (vector_types.h / vector_functions.h), for translation units that are not compiled by nvcc/hipcc.
// Same member names, sizes and alignments, so Model / LaunchParams keep their layout.
#pragma once
#include <cstdint>

#if !defined(__VECTOR_TYPES_H__) && !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && !defined(FOVPT_NO_VECTOR_TYPES)
struct float3 { float x, y, z; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(8) uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float3 make_float3(float s) { return make_float3(s, s, s); }
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline int2 make_int2(int x, int y) { int2 r; r.x = x; r.y = y; return r; }
static inline uint2 make_uint2(unsigned int x, unsigned int y) { uint2 r; r.x = x; r.y = y; return r; }
static inline uint3 make_uint3(unsigned int x, unsigned int y, unsigned int z) { uint3 r = {x, y, z}; return r; }
#endif
