// Development micro-benchmark: HBM read rate of a row-major n x 512 fp32 matrix when every wave owns ROWS rows and visits
// each row in pieces of SEG bytes (one piece per row per step), as the GEMM kernels do.
// hipcc --offload-arch=gfx950 -O3 -o dev/micro_stream dev/micro_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// lane l: row = l / LPR, piece = l % LPR (LPR = SEG / 16 lanes per row); a wave-instruction covers 64 / LPR rows
template <int SEG, int ROWS>
__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ X, long n, long ld, float* __restrict__ out) {
    constexpr int LPR = SEG / 16, RPI = 64 / LPR, NI = ROWS / RPI;  // rows per instruction, instructions per step
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 + ROWS > n) return;
    const float* p = X + (row0 + lane / LPR) * ld + 4 * (lane % LPR);
    f32x4 acc = f32x4{0, 0, 0, 0};
    for (int k = 0; k < 512 * 4 / SEG; ++k) {
        f32x4 v[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (long)j * RPI * ld + k * (SEG / 4)));
#pragma unroll
        for (int j = 0; j < NI; ++j) acc += v[j];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}
template <int SEG, int ROWS>
void run(const float* X, long n, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (int)(n / (4 * ROWS));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<SEG, ROWS>), dim3(blocks), dim3(256), 0, 0, X, n, 512L, out);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_stream<SEG, ROWS>), dim3(blocks), dim3(256), 0, 0, X, n, 512L, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("n=%ld  SEG=%4d B  ROWS/wave=%3d  %8.1f us  %6.2f TB/s\n", n, SEG, ROWS, ms * 1e3, n * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
    for (long n : {1000000L, 100000L}) {
        float *X, *out; hipMalloc(&X, n * 2048); hipMalloc(&out, 4096);
        hipMemset(X, 0, n * 2048);
        run<64, 64>(X, n, out); run<128, 64>(X, n, out); run<256, 64>(X, n, out); run<512, 64>(X, n, out); run<1024, 64>(X, n, out);
        run<128, 32>(X, n, out); run<256, 32>(X, n, out); run<512, 32>(X, n, out); run<256, 16>(X, n, out); run<1024, 16>(X, n, out);
        hipFree(X); hipFree(out);
    }
    return 0;
}
