// Development micro-benchmark (not part of the product): times the one-workgroup fp64 kernels in isolation.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPETAL_DEBUG_COUNTERS -o micro_small dev/micro_small.hip
#include "../petal-decomposition_amd/csrc/hip_ops.hip"
#include <random>
#include <cstring>
using namespace petal;
int main(int argc, char** argv) {
    int L = argc > 1 ? atoi(argv[1]) : 80;
    char err[256];
    Dev* d = dev_create(0, nullptr, err, sizeof(err));
    if (!d) { printf("no dev: %s\n", err); return 1; }
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    int M = 512;
    std::vector<double> B(M * L), S(L * L);
    for (auto& v : B) v = nd(rng);
    // graded columns (sigma ratio 1e3)
    for (int i = 0; i < M; ++i) for (int j = 0; j < L; ++j) B[i * L + j] *= pow(10.0, -3.0 * j / L);
    for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int k = 0; k < M; ++k) s += B[k * L + i] * B[k * L + j]; S[i * L + j] = s; }
    if (const char* mk = getenv("MATRIX")) {   // special matrices for the eigen-solver: diag, ident, blocks (two decoupled halves), arrow
        for (auto& v : S) v = 0;
        for (int i = 0; i < L; ++i) {
            if (!strcmp(mk, "diag")) S[i * L + i] = 1.0 + i;
            else if (!strcmp(mk, "ident")) S[i * L + i] = 1.0;
            else if (!strcmp(mk, "blocks")) { S[i * L + i] = 2.0 + 0.37 * i; int j = i + 1; if (j < L && j != L / 2) S[i * L + j] = S[j * L + i] = 0.5; }
            else if (!strcmp(mk, "arrow")) { S[i * L + i] = 1.0 + i; S[i] = S[i * L] = i ? 0.3 : 1.0; }
        }
    }
    double *dS = (double*)dev_alloc(d, 8 * L * L), *dA = (double*)dev_alloc(d, 8 * L * L), *dV = (double*)dev_alloc(d, 8 * L * L), *dw = (double*)dev_alloc(d, 8 * L), *dT = (double*)dev_alloc(d, 8 * L * L);
    dev_h2d(d, dS, S.data(), 8 * L * L);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn) {
        fn(); dev_sync(d);
        hipEventRecord(e0, (hipStream_t)dev_stream(d));
        for (int i = 0; i < 10; ++i) fn();
        hipEventRecord(e1, (hipStream_t)dev_stream(d));
        dev_sync(d);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s L=%d  %.2f us\n", name, L, ms * 100.0);
    };
    timeit("eigh(copy+jacobi)", [&] { dev_d2d(d, dA, dS, 8 * L * L); op_eigh(d, dA, L, L, dV, L, dw, 1e-8); });
    timeit("chol_inv", [&] { op_chol_inv(d, dS, L, L, dT, L, 1e-13); });
    timeit("dgemm 512xLxL", [&] { op_dgemm(d, false, false, 512, L, L, 1.0, dS, L, dT, L, 0.0, dA, L); });
    timeit("symdecorr", [&] { if (L <= 1024) op_symdecorr(d, L, dS, dA, 0); });
    std::vector<double> w(L), V(L * L), T(L * L);
    dev_d2d(d, dA, dS, 8 * L * L); op_eigh(d, dA, L, L, dV, L, dw, 1e-8);
    dev_d2h(d, w.data(), dw, 8 * L); dev_d2h(d, V.data(), dV, 8 * L * L); dev_sync(d);
    double maxres = 0;
    for (int j = 0; j < L; ++j) for (int i = 0; i < L; ++i) { double s = 0; for (int k = 0; k < L; ++k) s += S[i * L + k] * V[k * L + j]; maxres = fmax(maxres, fabs(s - w[j] * V[i * L + j])); }
    if (L <= 8) { for (int j = 0; j < L; ++j) { printf("w[%d] = %.6f  v =", j, w[j]); for (int i = 0; i < L; ++i) printf(" %9.2e", V[i * L + j]); printf("\n"); } }
    double maxorth = 0;
    for (int a = 0; a < L; ++a) for (int b = a; b < L; ++b) { double s = 0; for (int k = 0; k < L; ++k) s += V[k * L + a] * V[k * L + b]; maxorth = fmax(maxorth, fabs(s - (a == b))); }
    printf("eig residual max |S v - w v| = %.3e, max |V^T V - I| = %.3e (w0 = %.3e, wlast = %.3e)\n", maxres, maxorth, w[0], w[L - 1]);
    op_chol_inv(d, dS, L, L, dT, L, 1e-13); dev_d2h(d, T.data(), dT, 8 * L * L); dev_sync(d);
    double maxo = 0;  // T^T S T = I
    for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int a = 0; a < L; ++a) for (int b = 0; b < L; ++b) s += T[a * L + i] * S[a * L + b] * T[b * L + j]; maxo = fmax(maxo, fabs(s - (i == j))); }
    printf("chol_inv: max |T^T S T - I| = %.3e\n", maxo);
#ifdef PETAL_DEBUG_COUNTERS
    int h[4]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof(h)); printf("debug counters: sweeps=%d\n", h[0]);
    long long cyc[16]; hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
    printf("thread-0 cycles: params=%lld bar=%lld colphase=%lld bar=%lld rowphase=%lld bar=%lld\n", cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], cyc[5]);
    printf("chol: factor+scale cycles=%lld  inverse cycles=%lld (over 12 calls)\n", cyc[6], cyc[7]);
    printf("chol phases: update=%lld diag=%lld panel=%lld | invdiag=%lld a=%lld b=%lld\n", cyc[8], cyc[9], cyc[10], cyc[11], cyc[12], cyc[13]);
    printf("chol diag split: factor=%lld inverse=%lld barrier=%lld\n", cyc[14], cyc[15], cyc[9]);
#endif
    return 0;
}
