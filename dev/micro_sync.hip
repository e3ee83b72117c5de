// calibration: cost of one barriered step in a single resident workgroup (development only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_bar(int iters, double* out) {
    __shared__ double sm[1024];
    sm[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < iters; ++i) __syncthreads();
    out[threadIdx.x] = sm[threadIdx.x];
}
__global__ void k_bar_lds(int iters, double* out) {
    __shared__ double sm[2048];
    sm[threadIdx.x] = threadIdx.x; sm[threadIdx.x + 1024] = 1.0;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        double v = sm[(threadIdx.x + i) & 1023];
        double w = sm[1024 + ((threadIdx.x * 7 + i) & 1023)];
        __syncthreads();
        sm[threadIdx.x] = v * 0.5 + w;
        __syncthreads();
    }
    out[threadIdx.x] = sm[threadIdx.x];
}
__global__ void k_bar_div(int iters, double* out) {
    __shared__ double sm[1024];
    sm[threadIdx.x] = threadIdx.x + 1.5;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        double a = sm[(threadIdx.x + i) & 1023];
        double t = 1.0 / (fabs(a) + sqrt(a * a + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0);
        __syncthreads();
        sm[threadIdx.x] = c + t;
        __syncthreads();
    }
    out[threadIdx.x] = sm[threadIdx.x];
}
__global__ void k_clock(long long* out) {
    long long t0 = wall_clock64(), c0 = clock64();
    while (wall_clock64() - t0 < 100000) {}  // 1 ms at 100 MHz
    long long t1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; }
}
int main() {
    double* d; hipMalloc(&d, 8192); long long* dl; hipMalloc(&dl, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {64, 256, 640, 1024}) {
        for (int which = 0; which < 3; ++which) {
            int iters = 2000;
            auto run = [&] { if (which == 0) k_bar<<<1, threads>>>(iters, d); else if (which == 1) k_bar_lds<<<1, threads>>>(iters, d); else k_bar_div<<<1, threads>>>(iters, d); };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); run(); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("threads=%4d %s: %.1f ns per iteration\n", threads, which == 0 ? "barrier only     " : which == 1 ? "2 lds rd + wr, 2 bar" : "div+2sqrt+div, 2 bar", ms * 1e6 / iters);
        }
    }
    k_clock<<<1, 64>>>(dl); hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, dl, 16, hipMemcpyDeviceToHost);
    printf("clock: %lld shader cycles in %lld x10ns  => %.0f MHz (single wave busy-wait)\n", h[1], h[0], (double)h[1] / (h[0] * 10e-9) / 1e6);
    return 0;
}
