// tools/micro/gather_rate2.hip -- the node fetch again, separating its knobs: LOADS = 16-byte loads per lane and step (2 = the
// 128-byte node, 1 = a 64-byte node), FMAS = dependent vector instructions between the load and the next address (0, or 34 =
// a node step), blocks of 256 threads per CU (2, 4, 8 = 2, 4, 8 waves per SIMD); tables of 16 KB (L1), 2 MB (L2), 64 MB (Infinity Cache);
// dependent chain through a permutation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <utility>
template <int LOADS, int FMAS>
__global__ __launch_bounds__(256, 8) void k(const float4* __restrict__ tab, uint32_t mask, int iters, float4* out, float c0, float c1)
{
    const uint32_t j = threadIdx.x & 3u;
    uint32_t node = ((blockIdx.x * 64u + (threadIdx.x >> 2)) * 2654435761u) & mask;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        const float4* p = (const float4*)((const char*)tab + (((size_t)node << (LOADS == 2 ? 7 : 6)) | (j << (LOADS == 2 ? 5 : 4))));
        float4 a = p[0];
        float x = a.x;
        if (LOADS == 2) { const float4 b = p[1]; x += b.y; }
#pragma unroll
        for (int f = 0; f < FMAS; f++) x = __builtin_fmaf(x, c0, c1);      // c0 = 1, c1 = 0: x unchanged, but the chain is real
        acc += x;
        node = (__float_as_uint(a.z) + (uint32_t)i + (uint32_t)(x * 0.0f)) & mask;
        node = __builtin_amdgcn_mov_dpp((int)node, 0, 0xf, 0xf, true);
    }
    out[blockIdx.x * 256 + threadIdx.x] = make_float4(acc, 0, 0, 0);
}
int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate / 1e6;
    float4* out; (void)hipMalloc(&out, sizeof(float4) * 256 * cus * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (size_t bytes : {(size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20})
    for (int loads = 2; loads >= 1; loads--) {
        const size_t recb = loads == 2 ? 128 : 64, nrec = bytes / recb;
        std::vector<uint32_t> h(bytes / 4, 0x3f800000u);
        // the successor table is a PERMUTATION: walkers that start on different records never meet (with a random function
        // they coalesce within a few hundred steps and every quad of a wave ends up reading the same record)
        std::vector<uint32_t> perm(nrec);
        for (size_t r = 0; r < nrec; r++) perm[r] = (uint32_t)r;
        uint32_t s = 777;
        for (size_t r = nrec - 1; r > 0; r--) { s = s * 1664525u + 1013904223u; const size_t o = (size_t)(((uint64_t)(s >> 4) * (r + 1)) >> 28); std::swap(perm[r], perm[o]); }
        for (size_t r = 0; r < nrec; r++)
            for (size_t jj = 0; jj < 4; jj++) h[(r * recb + jj * (recb / 4)) / 4 + 2] = perm[r];
        float4* tab; (void)hipMalloc(&tab, bytes); (void)hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice);
        for (int fmas : {0, 34}) for (int bpc : {2, 4, 8}) {
            const int grid = cus * bpc, iters = 2000;
            auto launch = [&](int it) {
                if (loads == 2 && fmas == 0) hipLaunchKernelGGL((k<2, 0>), dim3(grid), dim3(256), 0, 0, tab, (uint32_t)(nrec - 1), it, out, 1.0f, 0.0f);
                if (loads == 2 && fmas == 34) hipLaunchKernelGGL((k<2, 34>), dim3(grid), dim3(256), 0, 0, tab, (uint32_t)(nrec - 1), it, out, 1.0f, 0.0f);
                if (loads == 1 && fmas == 0) hipLaunchKernelGGL((k<1, 0>), dim3(grid), dim3(256), 0, 0, tab, (uint32_t)(nrec - 1), it, out, 1.0f, 0.0f);
                if (loads == 1 && fmas == 34) hipLaunchKernelGGL((k<1, 34>), dim3(grid), dim3(256), 0, 0, tab, (uint32_t)(nrec - 1), it, out, 1.0f, 0.0f); };
            launch(50); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("table %6zu KB, loads %d x 16 B, %2d dependent fmas, %d waves/SIMD: %4.0f ns per step and wave = %5.0f cycles at %.1f GHz; %5.1f wave-steps per us and CU\n",
                   bytes >> 10, loads, fmas, bpc, ms * 1e6 / iters, ms * 1e6 / iters * ghz, ghz, (double)iters * bpc * 4 / (ms * 1e3));
        }
        (void)hipFree(tab);
    }
    return 0;
}
