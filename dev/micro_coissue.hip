// Development micro-benchmark: do bf16 MFMAs and ordinary VALU work overlap on a CDNA4 SIMD?  Each wave runs ITER rounds of
// NM independent-chain MFMAs (16x16x32 bf16, 16 cycles each) and NV dependent-free VALU fmas (4 cycles each), either interleaved
// in one instruction stream (mode 0) or as an MFMA phase followed by a VALU phase (mode 1); WPS waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o dev/micro_coissue dev/micro_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NM, int NV, int MODE>
__global__ __launch_bounds__(256) void k_co(float* __restrict__ out, int iters, float seed) {
    f32x4 acc[8];
    float v[8];
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); v[e] = seed * e; acc[e] = f32x4{0, 0, 0, 0}; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < (NM > 0 ? NM : 1); ++g) {
                if (NM > 0) acc[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[g & 7], 0, 0, 0);
#pragma unroll
                for (int w = 0; w < (NM > 0 ? NV / NM : NV); ++w) v[w & 7] = fmaf(v[w & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < NM; ++g) acc[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[g & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < NV; ++w) v[w & 7] = fmaf(v[w & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e] + acc[e][0] + acc[e][1] + acc[e][2] + acc[e][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int NM, int NV, int MODE>
void run(float* out, int wps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * wps;  // one 256-thread workgroup = one wave per SIMD of a CU
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_co<NM, NV, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_co<NM, NV, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("NM=%3d NV=%3d mode=%d waves/SIMD=%d : %8.1f us  (%.0f ns per round per SIMD; MFMA alone would be %d cycles, VALU alone %d)\n", NM, NV,
           MODE, wps, ms * 1e3, ms * 1e6 / iters, NM * 16, NV * 4);
}
int main() {
    float* out; hipMalloc(&out, 4096);
    for (int wps = 1; wps <= 2; ++wps) {
        run<24, 0, 1>(out, wps); run<0, 96, 1>(out, wps);
        run<24, 96, 0>(out, wps); run<24, 96, 1>(out, wps);
        run<24, 48, 0>(out, wps); run<24, 48, 1>(out, wps);
        run<24, 192, 0>(out, wps); run<24, 192, 1>(out, wps);
    }
    return 0;
}
