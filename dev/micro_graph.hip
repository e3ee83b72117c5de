// dev: the floor of a chain of tiny DEPENDENT kernels -- plain stream launches vs the same chain replayed as a hipGraph
// build: hipcc --offload-arch=gfx950 -O3 -w -o dev/micro_graph dev/micro_graph.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
int main() {
    double* p; hipMalloc(&p, 8 * 4096); hipMemset(p, 0, 8 * 4096);
    hipStream_t s; hipStreamCreate(&s);
    const int N = 50, reps = 20;
    auto chain = [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(16), dim3(256), 0, s, p, 4096); };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    chain(); hipStreamSynchronize(s);
    hipEventRecord(e0, s); for (int r = 0; r < reps; ++r) chain(); hipEventRecord(e1, s); hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
    printf("stream launches : %.2f us per kernel\n", ms * 1e3 / (N * reps));
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
    printf("graph replay    : %.2f us per kernel\n", ms * 1e3 / (N * reps));
    // host time to enqueue
    auto t0 = std::chrono::steady_clock::now(); for (int r = 0; r < reps; ++r) chain(); auto t1 = std::chrono::steady_clock::now(); hipStreamSynchronize(s);
    printf("host enqueue    : %.2f us per kernel (stream)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * reps));
    t0 = std::chrono::steady_clock::now(); for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s); t1 = std::chrono::steady_clock::now(); hipStreamSynchronize(s);
    printf("host enqueue    : %.2f us per kernel (graph)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * reps));
    return 0;
}
