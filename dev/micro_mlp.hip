// Development micro-benchmark: streaming rate of the K1 access pattern (64 rows x 128 B per wave and step) when only 8 waves per
// CU are resident (two 256-thread workgroups pinned by their LDS footprint) with DEPTH steps of loads in flight per wave.
// hipcc --offload-arch=gfx950 -O3 -o dev/micro_mlp dev/micro_mlp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH, int WORK>
__global__ __launch_bounds__(256) void k_mlp(const float* __restrict__ X, long n, long ld, float* __restrict__ out) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (row0 + 64 > n) return;
    const float* p = X + (row0 + i) * ld + 8 * q;
    f32x4 buf[DEPTH][8];
    f32x4 acc = f32x4{0, 0, 0, 0};
    auto issue = [&](int c, f32x4(&b)[8]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            b[2 * t] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (long)16 * t * ld + 32 * c));
            b[2 * t + 1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (long)16 * t * ld + 32 * c + 4));
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) issue(s, buf[s]);
    for (int c0 = 0; c0 < 16; c0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            f32x4 v = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 8; ++t) v += buf[s][t];
            if (c0 + s + DEPTH < 16) issue(c0 + s + DEPTH, buf[s]);
            __builtin_amdgcn_sched_barrier(0);
            for (int w = 0; w < WORK; ++w) v = v * 1.0001f + 0.5f;   // stands for the chunk's compute time
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
    if (pad[0] == 1.2345f) out[0] = 1;
}
template <int DEPTH, int WORK>
void run(const float* X, long n, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (int)(n / 256);
    const size_t lds = 70 * 1024;  // two workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp<DEPTH, WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_mlp<DEPTH, WORK>), dim3(blocks), dim3(256), lds, 0, X, n, 512L, out);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_mlp<DEPTH, WORK>), dim3(blocks), dim3(256), lds, 0, X, n, 512L, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("n=%ld  depth=%d  work=%4d  %8.1f us  %6.2f TB/s\n", n, DEPTH, WORK, ms * 1e3, n * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
    const long n = 1000000;
    float *X, *out; hipMalloc(&X, n * 2048); hipMalloc(&out, 4096); hipMemset(X, 0, n * 2048); hipMemset(out, 0, 4096);
    run<1, 0>(X, n, out); run<2, 0>(X, n, out); run<4, 0>(X, n, out);
    run<1, 100>(X, n, out); run<2, 100>(X, n, out); run<4, 100>(X, n, out);
    run<1, 250>(X, n, out); run<2, 250>(X, n, out); run<4, 250>(X, n, out);
    return 0;
}
