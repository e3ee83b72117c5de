// fp64 MFMA (v_mfma_f64_16x16x4_f64) on gfx950: cycles per instruction in a dependent accumulation chain and with 2 / 4 independent
// accumulators, one wave and two waves on a SIMD.   build: hipcc --offload-arch=gfx950 -O3 -o dev/micro_f64mfma dev/micro_f64mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, long long* cyc, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
void run(int threads) {
    double* out; long long* cyc; long long h;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
    const int iters = 1000;
    hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d accumulators, %d waves in the workgroup (%d per SIMD): %.1f clock64 ticks per MFMA of wave 0\n", NACC, threads / 64, (threads / 64 + 3) / 4,
           double(h) / (iters * 8.0 * NACC));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(64); run<2>(64); run<4>(64);
    run<1>(256); run<1>(512); run<2>(512); run<4>(512);
    return 0;
}
