// dev: how long after a kernel's last instruction does the host know? hipStreamSynchronize against hipStreamQuery polling against a
// flag the kernel stores into pinned host memory (the kernel runs ~1 ms, so a blocking wait has gone to sleep by then)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_spin(long long ticks, volatile int* flag, int value, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0) {
        *sink = value;
        __threadfence_system();
        __hip_atomic_store((int*)flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
    int* sink; CK(hipMalloc((void**)&sink, 64));
    *flag = 0;
    const long long ticks = 100000;   // 100 MHz wall clock: 1 ms
    std::vector<double> a, b, c;
    for (int rep = 0; rep < 60; ++rep) {
        const int v = rep * 3 + 1;
        double t0 = now_us();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, flag, v, sink);
        while (*(volatile int*)flag != v) { }
        a.push_back(now_us() - t0);
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, flag, v + 1, sink);
        CK(hipStreamSynchronize(s));
        b.push_back(now_us() - t0);
        t0 = now_us();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, flag, v + 2, sink);
        while (hipStreamQuery(s) == hipErrorNotReady) { }
        c.push_back(now_us() - t0);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("launch -> host knows, 1 ms kernel: flag in pinned memory %.1f us, hipStreamSynchronize %.1f us, hipStreamQuery polling %.1f us\n", med(a), med(b), med(c));
    return 0;
}
