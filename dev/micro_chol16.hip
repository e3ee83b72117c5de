// dev: what paces a 16 x 16 fp64 Cholesky in the registers of ONE wave (the diagonal blocks of k_chol_inv2)?
// build: hipcc --offload-arch=gfx950 -O3 -o dev/micro_chol16 dev/micro_chol16.hip ; run: dev/micro_chol16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ double readlane_d(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm_d(double x, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
template <int NEWTON>
__device__ __forceinline__ double fast_rsqrt_pos(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < NEWTON; ++it) { const double e = fma(-d * y, y, 1.0); y = fma(0.5 * y, e, y); }
    return y;
}
// V: 0 = bpermute layout, branch on ok, rinv/dead to LDS inside the loop (what the library does)
//    1 = same, no branch (select), no LDS writes inside the loop
//    2 = as 1 with the library rsqrt
//    3 = as 1 with ONE Newton step
//    4 = as 1, pivot through bpermute instead of readlane (no SGPR)
//    5 = one column per lane, readlane multipliers (round-2 form)
template <int V>
__global__ __launch_bounds__(64) void k(const double* __restrict__ G, double* __restrict__ out, long long* cyc, int reps, double rel_tol) {
    __shared__ double S[16][17];
    __shared__ double rinv[16];
    __shared__ int dead[16];
    const int tid = threadIdx.x, c = tid & 15, g = tid >> 4;
    for (int e = tid; e < 256; e += 64) S[e >> 4][e & 15] = G[e];
    __syncthreads();
    long long t0 = clock64();
    double keep = 0;
    for (int rep = 0; rep < reps; ++rep) {
        if (V == 5) {
            double a[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = (tid < 16 && i <= tid) ? S[i][tid] : 0.0;
            const double gv = tid < 16 ? S[tid][tid] : 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double dii = readlane_d(a[i], i), gi = readlane_d(gv, i);
                const bool ok = (gi > 0.0) && (dii > rel_tol * gi);
                const double inv = ok ? rsqrt(dii) : 0.0;
                a[i] *= inv;
#pragma unroll
                for (int kk = i + 1; kk < 16; ++kk) a[kk] -= readlane_d(a[i], kk) * a[i];
            }
            keep += a[15] + a[3];
        } else {
            double a[4], gdv[16];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = S[g + 4 * m][c];
#pragma unroll
            for (int i = 0; i < 16; ++i) gdv[i] = S[i][i];
            double myinv = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gi = i & 3, mi = i >> 2;
                const double dii = (V == 4) ? bperm_d(a[mi], gi * 16 + i) : readlane_d(a[mi], gi * 16 + i);
                const double su_c = bperm_d(a[mi], gi * 16 + c);
                double su_k[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) su_k[m] = (4 * m + 3 > i) ? bperm_d(a[mi], gi * 16 + g + 4 * m) : 0.0;
                const bool ok = (gdv[i] > 0.0) && (dii > rel_tol * gdv[i]);
                double inv;
                if (V == 0) inv = ok ? fast_rsqrt_pos<2>(dii) : 0.0;
                else if (V == 2) { const double r = rsqrt(dii); inv = ok ? r : 0.0; }
                else if (V == 3) { const double r = fast_rsqrt_pos<1>(dii); inv = ok ? r : 0.0; }
                else { const double r = fast_rsqrt_pos<2>(dii); inv = ok ? r : 0.0; }
                const double r_own = a[mi] * inv, rc = su_c * inv;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int kq = g + 4 * m;
                    const double rk = su_k[m] * inv;
                    if (m == mi) a[m] = (g == gi) ? r_own : (kq > i ? a[m] - rk * rc : a[m]);
                    else if (4 * m + 3 > i) a[m] = (kq > i) ? a[m] - rk * rc : a[m];
                }
                if (V == 0) { if (tid == 0) { rinv[i] = inv; dead[i] = inv > 0.0 ? 0 : 1; } }
                else myinv = (tid == i) ? inv : myinv;
            }
            if (V != 0 && tid < 16) { rinv[tid] = myinv; dead[tid] = myinv > 0.0 ? 0 : 1; }
            keep += a[0] + a[3];
        }
    }
    long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
    out[tid] = keep + rinv[tid & 15] + dead[tid & 15];
}
int main() {
    std::vector<double> h(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + abs(i - j)) + 0.01 * ((i * 7 + j * 3) % 5 + (j * 7 + i * 3) % 5);
    double *G, *out; long long* cyc;
    hipMalloc(&G, 256 * 8); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    hipMemcpy(G, h.data(), 256 * 8, hipMemcpyHostToDevice);
    const int reps = 2000;
#define RUN(V) do { for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, G, out, cyc, reps, 1e-15); hipDeviceSynchronize(); \
    long long hc; double ho[64]; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 64 * 8, hipMemcpyDeviceToHost); \
    printf("V%d: %.0f cycles per 16x16 factorisation = %.0f per pivot  (check %.6f)\n", V, double(hc) / reps, double(hc) / reps / 16, ho[0] / reps); } while (0)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5);
    return 0;
}
