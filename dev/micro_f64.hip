// Development micro-benchmark: issue cost and dependent latency of fp64 VALU instructions on one SIMD (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o dev/micro_f64 dev/micro_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void k_fma(double* out, long long* cyc, int iters, double a, double b) {
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = fma(x[c], a, b);
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void k_mul32(float* out, long long* cyc, int iters, float a, float b) {
    float x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3f + c;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = fmaf(x[c], a, b);
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void k_int(int* out, long long* cyc, int iters, int a) {
    int x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x + c;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = (x[c] ^ a) + (x[c] >> 3);
    }
    long long t1 = clock64();
    int s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rcp(double* out, long long* cyc, int iters) {
    double x = threadIdx.x + 1.5;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) x = __builtin_amdgcn_rcp(x) + 1.0;
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_div(double* out, long long* cyc, int iters, double a) {
    double x = threadIdx.x + 1.5;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) x = a / x + 1.0;
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 8 * 16);
    const int iters = 4096;
    long long h;
    auto rep = [&](const char* n, int threads, int ch, auto launch) {
        launch(threads); launch(threads); hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-34s threads %4d: %.2f cycles per instruction (per wave)\n", n, threads, (double)h / iters / ch);
    };
    for (int th : {64, 256, 512, 1024}) {
        rep("v_fma_f64 dependent chain", th, 1, [&](int t) { k_fma<1><<<1, t>>>(out, cyc, iters, 1.0000001, 1e-9); });
        rep("v_fma_f64 2 chains", th, 2, [&](int t) { k_fma<2><<<1, t>>>(out, cyc, iters, 1.0000001, 1e-9); });
        rep("v_fma_f64 4 chains", th, 4, [&](int t) { k_fma<4><<<1, t>>>(out, cyc, iters, 1.0000001, 1e-9); });
        rep("v_fma_f64 8 chains", th, 8, [&](int t) { k_fma<8><<<1, t>>>(out, cyc, iters, 1.0000001, 1e-9); });
        rep("v_fma_f32 dependent chain", th, 1, [&](int t) { k_mul32<1><<<1, t>>>((float*)out, cyc, iters, 1.0000001f, 1e-9f); });
        rep("v_fma_f32 8 chains", th, 8, [&](int t) { k_mul32<8><<<1, t>>>((float*)out, cyc, iters, 1.0000001f, 1e-9f); });
        rep("int xor/shift/add (3 ops) dep", th, 3, [&](int t) { k_int<1><<<1, t>>>((int*)out, cyc, iters, 12345); });
        rep("int xor/shift/add (3 ops) x8", th, 24, [&](int t) { k_int<8><<<1, t>>>((int*)out, cyc, iters, 12345); });
        rep("v_rcp_f64 + add dependent (2 ops)", th, 2, [&](int t) { k_rcp<<<1, t>>>(out, cyc, iters); });
        rep("fp64 IEEE divide + add (chain)", th, 1, [&](int t) { k_div<<<1, t>>>(out, cyc, iters, 3.0); });
    }
    return 0;
}
