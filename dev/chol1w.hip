// dev: the re-basing Cholesky (RT form) in the registers of ONE wave -- no barrier, no LDS array except one 16 x 16 transpose per
// block row.  build: hipcc --offload-arch=gfx950 -O3 -o dev/chol1w dev/chol1w.hip ; run: dev/chol1w
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <cstdint>
typedef double cf64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_d(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm_d(double x, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
template <int NEWTON>
__device__ __forceinline__ double rsqrt_pos(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < NEWTON; ++it) { const double e = fma(-d * y, y, 1.0); y = fma(0.5 * y, e, y); }
    return y;
}
#include "chol1w_kernel.h"
#include "chol4w_kernel.h"

// ---- host side ---------------------------------------------------------------------------------------------------------------
static double urand(unsigned long long& s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return ((s >> 11) + 0.5) / 9007199254740992.0; }
static double nrand(unsigned long long& s) { const double u = urand(s), v = urand(s); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_copy(const double* a, double* b, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }

template <int NB, int NEWTON, bool IL = true, int KIND = 0>
static void run(int L, double decay, int dead_col) {
    constexpr int M = 16 * NB;
    const int rows = 512;
    unsigned long long s = 12345 + L;
    // Y: rows x L with graded, strongly correlated columns (what the iterate looks like in an early power iteration)
    std::vector<double> Y((size_t)rows * L), G((size_t)M * M, 0.0);
    std::vector<double> base(rows);
    for (int r = 0; r < rows; ++r) base[r] = nrand(s);
    for (int c = 0; c < L; ++c) {
        const double sc = pow(decay, c);
        for (int r = 0; r < rows; ++r) Y[(size_t)r * L + c] = (c == dead_col) ? 0.0 : sc * (nrand(s) + 3.0 * base[r]);
    }
    for (int i = 0; i < L; ++i)
        for (int j = i; j < L; ++j) {
            long double a = 0;
            for (int r = 0; r < rows; ++r) a += (long double)Y[(size_t)r * L + i] * Y[(size_t)r * L + j];
            G[(size_t)i * M + j] = (double)a;
            if (j > i) G[(size_t)j * M + i] = 1e300;   // the lower triangle must not be read
        }
    for (int i = 0; i < L; ++i) G[(size_t)i * M + i] = G[(size_t)i * M + i];
    double *dG, *dG0, *dT; int* dnd; long long* dcyc;
    hipMalloc(&dG, sizeof(double) * M * M); hipMalloc(&dG0, sizeof(double) * M * M); hipMalloc(&dT, sizeof(double) * M * M);
    hipMalloc(&dnd, 4); hipMalloc(&dcyc, 8 * 64);
    hipMemcpy(dG, G.data(), sizeof(double) * M * M, hipMemcpyHostToDevice);
    hipMemcpy(dG0, G.data(), sizeof(double) * M * M, hipMemcpyHostToDevice);
    hipMemset(dT, 0xff, sizeof(double) * M * M); hipMemset(dnd, 0, 4);
    if (KIND == 0) hipLaunchKernelGGL((k_chol_rt<NB, NEWTON, IL>), dim3(1), dim3(64), 0, 0, dG, L, (long)M, dT, (long)M, 1e-14, dnd, L, dcyc);
    else hipLaunchKernelGGL((k_chol_rt4<NB>), dim3(1), dim3(256), 0, 0, dG, L, (int64_t)M, dT, (int64_t)M, 1e-14, dnd, L);
    if (hipDeviceSynchronize() != hipSuccess) printf("KERNEL FAILED\n");
    std::vector<double> T((size_t)M * M); int nd = -1; long long cyc[64];
    hipMemcpy(T.data(), dT, sizeof(double) * M * M, hipMemcpyDeviceToHost);
    hipMemcpy(&nd, dnd, 4, hipMemcpyDeviceToHost); hipMemcpy(cyc, dcyc, 8 * 64, hipMemcpyDeviceToHost);
    // reconstruct R: above-diagonal blocks as given, diagonal blocks = inverse of T_JJ (host, upper triangular)
    std::vector<double> R((size_t)M * M, 0.0);
    double worst_low = 0;
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < M; ++c) {
            const double v = T[(size_t)r * M + c];
            if ((r >> 4) > (c >> 4) || ((r >> 4) == (c >> 4) && r > c) || r >= L || c >= L) worst_low = fmax(worst_low, std::isnan(v) ? 1e300 : fabs(v));
            else if ((r >> 4) < (c >> 4)) R[(size_t)r * M + c] = v;
        }
    for (int J = 0; J < NB; ++J) {   // R_JJ = T_JJ^-1 by back substitution per column
        const int jb = 16 * J;
        for (int c = 0; c < 16 && jb + c < L; ++c) {
            // solve T_JJ x = e_c  (upper triangular)
            double x[16] = {0};
            for (int r = 15; r >= 0; --r) {
                if (jb + r >= L) continue;
                const double trr = T[(size_t)(jb + r) * M + jb + r];
                if (trr == 0.0) { x[r] = 0.0; continue; }   // dead column
                double a = (r == c) ? 1.0 : 0.0;
                for (int q = r + 1; q < 16 && jb + q < L; ++q) a -= T[(size_t)(jb + r) * M + jb + q] * x[q];
                x[r] = a / trr;
            }
            for (int r = 0; r < 16 && jb + r < L; ++r) R[(size_t)(jb + r) * M + jb + c] = x[r];
        }
    }
    // backward error of R^T R = G, relative to sqrt(G_ii G_jj), dead columns excluded
    double berr = 0;
    for (int i = 0; i < L; ++i)
        for (int j = i; j < L; ++j) {
            if (i == dead_col || j == dead_col) continue;
            long double a = 0;
            for (int k = 0; k <= i; ++k) a += (long double)R[(size_t)k * M + i] * R[(size_t)k * M + j];
            const double den = sqrt(G[(size_t)i * M + i] * G[(size_t)j * M + j]);
            berr = fmax(berr, fabs((double)a - G[(size_t)i * M + j]) / den);
        }
    // timing: back-to-back launches, G rewritten by another kernel in between (as in the fit: the Gram kernel writes it)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    float ms_pair = 0, ms_copy = 0, ms_empty = 0;
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(k_copy, dim3((M * M + 255) / 256), dim3(256), 0, 0, dG0, dG, M * M);
            if (KIND == 0) hipLaunchKernelGGL((k_chol_rt<NB, NEWTON, IL>), dim3(1), dim3(64), 0, 0, dG, L, (long)M, dT, (long)M, 1e-14, dnd, L, (long long*)nullptr);
            else hipLaunchKernelGGL((k_chol_rt4<NB>), dim3(1), dim3(256), 0, 0, dG, L, (int64_t)M, dT, (int64_t)M, 1e-14, dnd, L);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_pair, e0, e1);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(k_copy, dim3((M * M + 255) / 256), dim3(256), 0, 0, dG0, dG, M * M);
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, (int*)nullptr);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_copy, e0, e1);
        hipEventRecord(e0);
        for (int i = 0; i < 2 * reps; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, (int*)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_empty, e0, e1);
    }
    printf("kind=%d NB=%d il=%d L=%d newton=%d decay=%.3f dead=%d: backward err %.2e, junk outside the factor %.1e, ndead %d | in-kernel %lld cycles "
           "(load %lld, rows:", KIND, NB, (int)IL, L, NEWTON, decay, dead_col, berr, worst_low, nd, cyc[63], cyc[0]);
    for (int J = 0; J < NB; ++J) printf(" %lld+%lld", cyc[1 + 2 * J], cyc[2 + 2 * J]);
    printf(") | copy+chol %.2f us, copy+empty %.2f us, empty %.2f us -> chol over an empty kernel %.2f us\n", ms_pair * 1e3 / reps,
           ms_copy * 1e3 / reps, ms_empty * 1e3 / (2 * reps), (ms_pair - ms_copy) * 1e3 / reps);
    hipFree(dG); hipFree(dG0); hipFree(dT); hipFree(dnd); hipFree(dcyc);
}
int main() {
    run<5, 1, false>(74, 0.85, -1);
    run<5, 1, false, 1>(74, 0.85, -1);
    run<5, 1, false, 1>(74, 0.70, -1);
    run<5, 1, false, 1>(74, 0.85, 20);
    run<5, 1, false, 1>(80, 0.85, 79);
    run<5, 1, false, 1>(65, 0.85, 0);
    run<3, 1, false, 1>(48, 0.85, -1);
    run<2, 1, false, 1>(20, 0.85, -1);
    run<1, 1, false, 1>(10, 0.85, -1);
    run<4, 1, false, 1>(64, 0.85, 33);
    run<6, 1, false, 1>(90, 0.9, -1);
    run<7, 1, false, 1>(100, 0.9, -1);
    run<8, 1, false, 1>(128, 0.9, 5);
    run<9, 1, false>(138, 0.92, -1);
    run<9, 1, false, 1>(138, 0.92, -1);
    run<9, 1, false, 1>(144, 0.92, 77);
    run<9, 1, false, 1>(129, 0.95, 128);
    return 0;
}
