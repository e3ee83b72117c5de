// cpu_ops.cpp -- HOST-MEMORY SIMULATION of petal-decomposition_amd/csrc/ops.h.  TEST INFRASTRUCTURE.
//
// Built ONLY by tests/ (and linked with the product's algo.cpp/api.cpp into tests/_build/
// libpetal_hostsim.so) so that the host algorithms and the sample-sharded collective path can run
// without a GPU (CPU test suite, world_size-2 gloo test).  It is never part of the product library
// libpetal_hip.so: that links hip_ops.hip and fails loudly without a gfx950 device.
//
// Every op is the plain-loop definition of the contract in ops.h, in fp64.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

#include "../petal-decomposition_amd/csrc/ops.h"

namespace petal {

struct Dev { int tag = 0; int gemm_mode = 1; double opt[OPT_COUNT] = {}; };

// the same option table as the device library (ops.h PetalOpt), defaults from the same environment variables, read once per ctx
Dev* dev_create(int, void*, char*, size_t) {
    Dev* d = new Dev();
    auto on = [](const char* name) { return std::getenv(name) != nullptr; };
    auto num = [](const char* name, double dflt) { const char* e = std::getenv(name); return e ? std::atof(e) : dflt; };
    d->opt[OPT_TWO_PLANE] = on("PETAL_NO_P2") ? 0 : 1;
    d->opt[OPT_TWO_PLANE_OMEGA] = on("PETAL_NO_P2_OMEGA") ? 0 : 1;
    d->opt[OPT_TWO_PLANE_ITERATE] = on("PETAL_NO_P2_ITERATE") ? 0 : 1;
    d->opt[OPT_STEERING] = on("PETAL_NO_POW3_FAST") ? 0 : 1;
    d->opt[OPT_FUSED_PASS] = on("PETAL_NO_POW3") ? 0 : 1;
    d->opt[OPT_FUSED_PASS_MIN_ROWS] = num("PETAL_POW3_MIN_ROWS", 8192);
    d->opt[OPT_VERDICT_THRESHOLD] = num("PETAL_P2_VERDICT_THR", 4e-6);
    d->opt[OPT_MEANS_FOLD_ROWS] = on("PETAL_NO_MEANS_FOLD") ? -1 : num("PETAL_MEANS_FOLD_ROWS", 200000);
    d->opt[OPT_GRAM_SPLIT] = on("PETAL_NO_GRAM3") ? 0 : 1;
    d->opt[OPT_GRAM_SPLIT_HOOK] = on("PETAL_GRAM_SPLIT") ? 1 : 0;
    d->opt[OPT_D2H_KERNEL] = 1;
    d->opt[OPT_ROW_PAD] = on("PETAL_NO_ROW_PAD") ? 0 : 1;
    d->opt[OPT_EIGH_JACOBI] = on("PETAL_EIGH_JACOBI") ? 1 : 0;
    d->opt[OPT_POISON] = 0;
    return d;
}
void dev_set_option(Dev* d, int opt, double value) {
    if (opt < 0 || opt >= OPT_COUNT) throw std::invalid_argument("unknown ctx option");
    d->opt[opt] = value;
}
double dev_option(const Dev* d, int opt) {
    if (opt < 0 || opt >= OPT_COUNT) throw std::invalid_argument("unknown ctx option");
    return d->opt[opt];
}
void dev_destroy(Dev* d) { delete d; }
void* dev_stream(Dev*) { return nullptr; }
void* dev_alloc(Dev*, size_t bytes) {
    void* p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::runtime_error("host sim: out of memory");
    return p;
}
void dev_free(Dev*, void* p) { std::free(p); }
void dev_memset(Dev*, void* p, int v, size_t bytes) { std::memset(p, v, bytes); }
void dev_h2d(Dev*, void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
void dev_h2d_async(Dev*, void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
void dev_d2h(Dev*, void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
size_t dev_view_limit(Dev*) { return size_t(8) << 20; }   // the device library's ring slot: the same limit, so the CPU suite sees a caller that ignores it
const void* dev_d2h_view(Dev* d, const void* src, size_t bytes) {
    if (bytes > dev_view_limit(d)) throw std::logic_error("dev_d2h_view: larger than a ring slot");
    return src;
}
const void* dev_h2d_view(Dev*, const void* src, size_t) { return src; }
void dev_d2h_multi(Dev*, int nseg, void* const* dst, const void* const* src, const size_t* bytes) { for (int i = 0; i < nseg; ++i) if (bytes[i]) std::memcpy(dst[i], src[i], bytes[i]); }
void dev_d2d(Dev*, void* dst, const void* src, size_t bytes) { std::memmove(dst, src, bytes); }
void dev_copy2d(Dev*, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int) {
    for (size_t i = 0; i < height; ++i)
        std::memcpy(static_cast<char*>(dst) + i * dpitch, static_cast<const char*>(src) + i * spitch, width);
}
void dev_sync(Dev*) {}
void dev_set_profiling(Dev*, int) {}
void dev_abort(Dev*) {}
void dev_make_current(Dev*) {}
int dev_push_current(Dev*) { return -1; }
void dev_pop_current(Dev*, int) {}
void dev_set_gemm_mode(Dev* d, int mode) { d->gemm_mode = mode; }   // 0: the split-product mode's two-plane roundings are simulated
int dev_gemm_mode(const Dev* d) { return d->gemm_mode; }
void dev_reset_timing(Dev*) {}
void dev_set_tag(Dev* d, int tag) { d->tag = tag; }
void dev_fork(Dev*, bool) {}
void dev_fork_end(Dev*) {}
void dev_join(Dev*) {}
void dev_fork_abort(Dev*) {}
KernelTiming dev_timing(Dev*) { return KernelTiming{}; }
void* dev_span_begin(Dev*, int) { return nullptr; }
void dev_span_end(Dev*, void*) {}

namespace {
inline double ld(const void* p, int dt, int64_t i) {
    return dt == F64 ? static_cast<const double*>(p)[i] : double(static_cast<const float*>(p)[i]);
}
inline void st(void* p, int dt, int64_t i, double v) {
    if (dt == F64) static_cast<double*>(p)[i] = v;
    else static_cast<float*>(p)[i] = float(v);
}
// value as the device kernel sees it after centring in the storage dtype
inline double centred(const void* X, int dt, int64_t idx, const void* mu, int64_t j) {
    if (!mu) return ld(X, dt, idx);
    if (dt == F64) return static_cast<const double*>(X)[idx] - static_cast<const double*>(mu)[j];
    return double(static_cast<const float*>(X)[idx] - static_cast<const float*>(mu)[j]);
}
}  // namespace

void op_pack_strided(Dev*, int dt, const void* src, int64_t n, int64_t d, int64_t rs, int64_t cs, void* dst,
                     int64_t ld_dst, int64_t d_pad) {
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < d_pad; ++j) st(dst, dt, i * ld_dst + j, j < d ? ld(src, dt, i * rs + j * cs) : 0.0);
}
void op_unpack_strided(Dev*, int dt, const void* src, int64_t n, int64_t d, int64_t ld_src, void* dst, int64_t rs,
                       int64_t cs, const double* scale) {
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < d; ++j) st(dst, dt, i * rs + j * cs, ld(src, dt, i * ld_src + j) * (scale ? scale[j] : 1.0));
}
void op_colsum(Dev*, int dt, const void* X, int64_t n, int64_t d, int64_t ldx, double* out, bool with_sq) {
    for (int64_t j = 0; j < (with_sq ? 2 * d : d); ++j) out[j] = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < d; ++j) {
            const double v = ld(X, dt, i * ldx + j);
            out[j] += v;
            if (with_sq) out[d + j] += v * v;
        }
}
// the sum of the two leading bf16 pieces of (float)v, round-to-nearest-even each (hip_ops.hip: k_trsm_pack<NB, true>, k_xp3<NPL = 2>)
static double two_plane(double v) {
    auto bf = [](float f) {
        uint32_t u; std::memcpy(&u, &f, 4);
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        float r; std::memcpy(&r, &u, 4); return r;
    };
    const float f = float(v), h = bf(f), m = bf(f - h);
    return double(h) + double(m);
}
void op_gemm_xp(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P0,
                int64_t N, int64_t ldp0, const void* bias, void* Z, int64_t ldz, double* sumsq, int p_planes, bool steering) {
    const bool x2 = steering && p_planes == 2 && dt == F32 && d->gemm_mode == 0 && !sumsq && N > 80 && d->opt[OPT_STEERING] != 0;
    // (split-product mode, fp32 data: a caller that accepts a two-plane P gets one, as on the device)
    std::vector<double> p2;
    const double* P = P0;
    int64_t ldp = ldp0;
    if (p_planes == 2 && dt == F32 && d->gemm_mode == 0 && !sumsq) {
        p2.resize(size_t(K) * N);
        for (int64_t k = 0; k < K; ++k)
            for (int64_t j = 0; j < N; ++j) p2[k * N + j] = two_plane(P0[k * ldp0 + j]);
        P = p2.data(); ldp = N;
    }
    std::vector<double> row(K), acc(N);
    double ss = 0;
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t k = 0; k < K; ++k) { row[k] = centred(X, dt, i * ldx + k, mu, k); ss += row[k] * row[k]; }
        if (x2) for (int64_t k = 0; k < K; ++k) row[k] = two_plane(double(float(row[k])));   // (the device's steering pass: Xc on two planes)
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int64_t k = 0; k < K; ++k) {
            const double a = row[k];
            if (a == 0.0) continue;
            const double* p = P + k * ldp;
            for (int64_t j = 0; j < N; ++j) acc[j] += a * (dt == F32 ? double(float(p[j])) : p[j]);
        }
        for (int64_t j = 0; j < N; ++j) st(Z, dt, i * ldz + j, acc[j] + (bias ? ld(bias, dt, j) : 0.0));
    }
    if (sumsq) *sumsq += ss;
}
void op_gemm_atb(Dev* d, int dt, const void* A, int64_t lda, int64_t M, const void* muA, const void* B, int64_t ldb,
                 int64_t N, const void* muB, int64_t n, double* C, int64_t ldc, bool precise, bool steering) {
    const bool p4 = steering && !precise && dt == F32 && d->gemm_mode == 0 && N > 80 && d->opt[OPT_STEERING] != 0;
    for (int64_t m = 0; m < M; ++m)
        for (int64_t j = 0; j < N; ++j) C[m * ldc + j] = 0;
    std::vector<double> a(M), b(N);
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t m = 0; m < M; ++m) a[m] = centred(A, dt, i * lda + m, muA, m);
        for (int64_t j = 0; j < N; ++j) b[j] = centred(B, dt, i * ldb + j, muB, j);
        if (p4) {   // (the device's steering pass: both operands on two planes)
            for (int64_t m = 0; m < M; ++m) a[m] = two_plane(double(float(a[m])));
            for (int64_t j = 0; j < N; ++j) b[j] = two_plane(double(float(b[j])));
        }
        for (int64_t m = 0; m < M; ++m) {
            if (a[m] == 0.0) continue;
            double* c = C + m * ldc;
            for (int64_t j = 0; j < N; ++j) c[j] += a[m] * b[j];
        }
    }
}
bool op_gram_split(Dev* d, const void* X, int64_t n, int64_t dd, int64_t dp, int64_t ldx, const void* mu, double* C, int64_t ldc,
                   double* mu64_fold, double n_total) {
    const bool off = d->opt[OPT_GRAM_SPLIT] == 0;
    if (off || d->gemm_mode == 1 || n < 64 || dd < 4) return false;   // (the simulation takes every shape the split-product modes would)
    if (mu64_fold) {
        // the device path's arithmetic (hip_ops.hip, k_gram5 SUMS): a provisional centre from a strided row sample, the Gram matrix and
        // the column sums about it, then the move to the true centre
        const bool no_fold = d->opt[OPT_MEANS_FOLD_ROWS] < 0;
        if (no_fold || !mu) return false;
        float* muT = static_cast<float*>(const_cast<void*>(mu));
        const float* x = static_cast<const float*>(X);
        const int64_t ns = std::min<int64_t>(n, 4096), stride = n / ns;
        for (int64_t j = 0; j < dp; ++j) {
            double sacc = 0;
            if (j < dd) for (int64_t i = 0; i < ns; ++i) sacc += x[i * stride * ldx + j];
            muT[j] = j < dd ? float(sacc / double(ns)) : 0.f;
        }
        op_gemm_atb(d, F32, X, ldx, dp, mu, X, ldx, dp, mu, n, C, ldc, false);
        std::vector<double> delta(size_t(dp), 0.0);
        for (int64_t j = 0; j < dd; ++j) {
            double sacc = 0;
            for (int64_t i = 0; i < n; ++i) sacc += double(x[i * ldx + j]) - double(muT[j]);
            delta[j] = sacc / n_total;
        }
        for (int64_t f = 0; f < dd; ++f)
            for (int64_t g = 0; g < dd; ++g) C[f * ldc + g] -= n_total * delta[f] * delta[g];
        for (int64_t j = 0; j < dp; ++j) {
            const double m = j < dd ? double(muT[j]) + delta[j] : 0.0;
            mu64_fold[j] = m;
            muT[j] = float(m);
        }
        return true;
    }
    op_gemm_atb(d, F32, X, ldx, dp, mu, X, ldx, dp, mu, n, C, ldc, false);
    return true;
}
void op_flip_key(Dev*, const double* t, double* key, int64_t L, const int* flag) {
    if (flag) key[L] = flag[0] != 0 ? 3.0 : (flag[1] != 0 ? 1.0 : (flag[2] != 0 ? 2.0 : 0.0));
    for (int64_t j = 0; j < L; ++j) {
        uint64_t bits = 0;
        const double a = t[j] < 0 ? 0.0 : t[j];
        std::memcpy(&bits, &a, 8);
        const uint64_t row = t[L + j] < double(1ll << 28) ? uint64_t(t[L + j]) : (uint64_t(1) << 28) - 1;
        const uint64_t payload = (((uint64_t(1) << 28) - 1 - row) << 1) | (t[2 * L + j] < 0 ? 1u : 0u);
        bits = (bits & ~((uint64_t(1) << 29) - 1)) | (t[j] < 0 ? 0 : payload);
        std::memcpy(&key[j], &bits, 8);
    }
}
void op_col_absmax(Dev*, int dt, const void* U, int64_t n, int64_t L, int64_t ldu, int64_t row_offset, double* absmax,
                   double* idx, double* sign) {
    for (int64_t j = 0; j < L; ++j) {
        double best = -1, bi = std::numeric_limits<double>::infinity(), bs = 1;
        for (int64_t i = 0; i < n; ++i) {
            const double v = ld(U, dt, i * ldu + j), a = std::fabs(v);
            if (i == 0 || a > best) { best = a; bi = double(row_offset + i); bs = std::signbit(v) ? -1.0 : 1.0; }
        }
        absmax[j] = best; idx[j] = bi; sign[j] = bs;
    }
}
void op_scale_cols(Dev*, int dt, void* A, int64_t n, int64_t L, int64_t lda, const double* s) {
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < L; ++j) st(A, dt, i * lda + j, ld(A, dt, i * lda + j) * s[j]);
}
void op_logcosh_rows(Dev*, int dt, const void* X, int64_t r, int64_t c, int64_t ldx, void* G, int64_t ldg, double* gp) {
    for (int64_t i = 0; i < r; ++i) {
        double s = 0;
        for (int64_t j = 0; j < c; ++j) {
            const double g = std::tanh(ld(X, dt, i * ldx + j));
            st(G, dt, i * ldg + j, g);
            s += 1.0 - g * g;
        }
        gp[i] = s;
    }
}

// ---- small f64 ops ---------------------------------------------------------------------------
void op_dgemm(Dev*, bool ta, bool tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
              const double* B, int64_t ldb, double beta, double* C, int64_t ldc, const double* colscale) {
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) {
            double s = 0;
            for (int64_t k = 0; k < K; ++k) s += (ta ? A[k * lda + i] : A[i * lda + k]) * (tb ? B[j * ldb + k] : B[k * ldb + j]);
            C[i * ldc + j] = alpha * s * (colscale ? colscale[j] : 1.0) + (beta != 0.0 ? beta * C[i * ldc + j] : 0.0);
        }
}
void op_sigma_inv(Dev*, const double* lam, double* sig, double* inv, int64_t count, double thr) {
    for (int64_t i = 0; i < count; ++i) sig[i] = std::sqrt(std::max(lam[i], 0.0));
    const double s0 = count ? sig[0] : 0.0;
    for (int64_t i = 0; i < count; ++i) inv[i] = (sig[i] > thr * s0 && sig[i] > 0) ? 1.0 / sig[i] : 0.0;
}
void op_scale_pad_cols(Dev*, const double* V, int64_t ldv, const double* inv, int64_t rows, int64_t r, int64_t rp, double* P) {
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < rp; ++j) P[i * rp + j] = j < r ? V[i * ldv + j] * inv[j] : 0.0;
}
void op_transpose_out(Dev*, int dt, const double* V, int64_t ldv, int64_t d, int64_t k, void* comp) {
    for (int64_t j = 0; j < k; ++j)
        for (int64_t i = 0; i < d; ++i) {
            if (dt == F32) static_cast<float*>(comp)[j * d + i] = float(V[i * ldv + j]);
            else static_cast<double*>(comp)[j * d + i] = V[i * ldv + j];
        }
}
void op_components_out(Dev*, int dt, const double* Bt, int64_t ldb, const double* Uh, int64_t ldu, const double* lam, double thr,
                       int64_t d, int64_t L, int64_t k, void* comp) {
    const double s0 = std::sqrt(std::max(lam[0], 0.0));
    for (int64_t j = 0; j < k; ++j) {
        const double sj = std::sqrt(std::max(lam[j], 0.0)), inv = (sj > thr * s0 && sj > 0) ? 1.0 / sj : 0.0;
        for (int64_t i = 0; i < d; ++i) {
            double a = 0;
            for (int64_t l = 0; l < L; ++l) a += Bt[i * ldb + l] * Uh[l * ldu + j];
            if (dt == F32) static_cast<float*>(comp)[j * d + i] = float(a * inv);
            else static_cast<double*>(comp)[j * d + i] = a * inv;
        }
    }
}
void op_chol_inv(Dev*, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead, int64_t Lz,
                 int64_t ndead_cols) {
    if (Lz < L) Lz = L;
    std::vector<double> R(size_t(L) * L, 0.0);
    std::vector<char> dead(L, 0);
    for (int64_t j = 0; j < L; ++j) {  // row-by-row upper Cholesky, G = R^T R
        double s = G[j * ldg + j];
        for (int64_t k = 0; k < j; ++k) s -= R[k * L + j] * R[k * L + j];
        const double gjj = G[j * ldg + j];
        if (!(gjj > 0) || !(s > rel_tol * gjj)) { dead[j] = 1; continue; }  // dependent column: dropped
        const double rjj = std::sqrt(s);
        R[j * L + j] = rjj;
        for (int64_t c = j + 1; c < L; ++c) {
            double v = G[j * ldg + c];
            for (int64_t k = 0; k < j; ++k) v -= R[k * L + j] * R[k * L + c];
            R[j * L + c] = v / rjj;
        }
    }
    if (ndead) { int c = 0; for (int64_t j = 0; j < (ndead_cols > 0 ? std::min(L, ndead_cols) : L); ++j) c += dead[j]; if (c > *ndead) *ndead = c; }
    // T = R^{-1} by back substitution per column; dead columns -> 0 (and are skipped as rows)
    for (int64_t i = 0; i < Lz; ++i)
        for (int64_t j = 0; j < Lz; ++j) T[i * ldt + j] = 0;
    for (int64_t j = 0; j < L; ++j) {
        if (dead[j]) continue;
        T[j * ldt + j] = 1.0 / R[j * L + j];
        for (int64_t i = j - 1; i >= 0; --i) {
            if (dead[i]) continue;
            double s = 0;
            for (int64_t k = i + 1; k <= j; ++k) s += R[i * L + k] * T[k * ldt + j];
            T[i * ldt + j] = -s / R[i * L + i];
        }
    }
}
void op_ritz_residual(Dev*, const double* CV, int64_t ldc, const double* Vr, int64_t ldv, int64_t rows, int64_t nc, const double* theta,
                      const int* flag, double* out3, double* w_out, const int* flag2) {
    double worst = 0, bad = 0;
    if (w_out) for (int64_t j = 0; j < nc; ++j) w_out[j] = theta[j];
    for (int64_t j = 0; j < nc; ++j) {
        double s2 = 0;
        for (int64_t i = 0; i < rows; ++i) { const double r = CV[i * ldc + j] - theta[j] * Vr[i * ldv + j]; s2 += r * r; }
        if (!std::isfinite(s2)) bad = 1;
        else worst = std::max(worst, s2);
    }
    if (flag && *flag != 0) bad = 1;
    if (flag2 && *flag2 != 0) bad = 1;
    out3[0] = worst; out3[1] = nc > 0 ? theta[0] : 0.0; out3[2] = bad;
}
void op_whiten_k(Dev*, const double* U, int64_t ldu, const double* lam, int64_t rows, int64_t nc, int64_t ncp, double scale,
                 double* KT, double* KTs) {
    for (int64_t j = 0; j < ncp; ++j) {
        double f = 0;
        if (j < nc) {  // sign normalisation: first component of largest magnitude positive (see hip_ops.hip)
            double best = -1.0, sgn = 1.0;
            for (int64_t i = 0; i < rows; ++i)
                if (std::fabs(U[i * ldu + j]) > best) { best = std::fabs(U[i * ldu + j]); sgn = U[i * ldu + j] < 0 ? -1.0 : 1.0; }
            const double sg = std::sqrt(std::max(lam[j], 0.0));
            f = sg > 0 ? sgn / sg : 0.0;
        }
        for (int64_t i = 0; i < rows; ++i) {
            const double v = j < nc ? U[i * ldu + j] * f : 0.0;
            KT[i * ncp + j] = v;
            KTs[i * ncp + j] = v * scale;
        }
    }
}
void op_eigh(Dev* dv, double* A, int64_t L, int64_t lda, double* V, int64_t ldv, double* w, double tol_rel, bool clustered, int64_t Lz,
             int64_t /*ncheck*/, int* verdict, bool verdict_fresh, double /*gap_tol_override*/) {
    if (verdict && verdict_fresh) *verdict = 0;
    // TEST HOOK (this simulation only): PETAL_OPT_EIGH_JACOBI = 2 makes every optimistic solve report "eigenvalues too close", as the
    // device's two-stage solver does on clustered spectra -- the host logic behind that verdict (RandomizedPca repeats its small stage
    // alone with the Jacobi solver, `petal_stats.eigh_redo`; sharded fits agree on it) then runs in the CPU suite.  The solve below
    // is the same Jacobi iteration either way, so the repeated stage must reproduce the first one's numbers.
    if (verdict && !clustered && L > 0 && dv->opt[OPT_EIGH_JACOBI] == 2.0) *verdict |= 1;
    for (int64_t r = 0; r < Lz; ++r)
        for (int64_t c = 0; c < Lz; ++c)
            if (r >= L || c >= L) V[r * ldv + c] = 0.0;
    for (int64_t i = 0; i < L; ++i)
        for (int64_t j = 0; j < L; ++j) V[i * ldv + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        // the device's stopping rule (hip_ops.hip, wg_jacobi_violation): a_pq^2 <= tol^2 |a_pp a_qq| or below the rounding floor
        double diag = 0, viol = 0;
        for (int64_t i = 0; i < L; ++i) diag += A[i * lda + i] * A[i * lda + i];
        if (!(diag > 0) || !(diag < 1e300)) break;
        for (int64_t i = 0; i < L; ++i)
            for (int64_t j = i + 1; j < L; ++j) {
                const double v = A[i * lda + j];
                if (v != 0) viol = std::max(viol, v * v / std::max(tol_rel * tol_rel * std::fabs(A[i * lda + i] * A[j * lda + j]), 1e-32 * diag));
            }
        if (!(viol > 1.0)) break;
        for (int64_t p = 0; p < L - 1; ++p)
            for (int64_t q = p + 1; q < L; ++q) {
                const double apq = A[p * lda + q];
                if (apq == 0) continue;
                const double theta = (A[q * lda + q] - A[p * lda + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
                for (int64_t k = 0; k < L; ++k) {  // columns p,q
                    const double akp = A[k * lda + p], akq = A[k * lda + q];
                    A[k * lda + p] = cs * akp - sn * akq;
                    A[k * lda + q] = sn * akp + cs * akq;
                }
                for (int64_t k = 0; k < L; ++k) {  // rows p,q
                    const double apk = A[p * lda + k], aqk = A[q * lda + k];
                    A[p * lda + k] = cs * apk - sn * aqk;
                    A[q * lda + k] = sn * apk + cs * aqk;
                }
                for (int64_t k = 0; k < L; ++k) {
                    const double vkp = V[k * ldv + p], vkq = V[k * ldv + q];
                    V[k * ldv + p] = cs * vkp - sn * vkq;
                    V[k * ldv + q] = sn * vkp + cs * vkq;
                }
            }
    }
    std::vector<int64_t> order(L);
    for (int64_t i = 0; i < L; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return A[a * lda + a] > A[b * lda + b]; });
    std::vector<double> Vs(size_t(L) * L);
    for (int64_t j = 0; j < L; ++j) {
        w[j] = A[order[j] * lda + order[j]];
        for (int64_t i = 0; i < L; ++i) Vs[i * L + j] = V[i * ldv + order[j]];
    }
    for (int64_t i = 0; i < L; ++i)
        for (int64_t j = 0; j < L; ++j) V[i * ldv + j] = Vs[i * L + j];
}
void op_jacobi_svd_rows(Dev*, double* A, int64_t L, int64_t lda, double* U, int64_t ldu, double* s_inv, int*) {
    std::vector<double> G(size_t(L) * L, 0.0);
    for (int64_t i = 0; i < L; ++i) G[i * L + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int64_t p = 0; p < L - 1; ++p)
            for (int64_t q = p + 1; q < L; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int64_t j = 0; j < L; ++j) { al += A[p * lda + j] * A[p * lda + j]; be += A[q * lda + j] * A[q * lda + j]; ga += A[p * lda + j] * A[q * lda + j]; }
                if (!(std::fabs(ga) > 1e-15 * std::sqrt(al * be))) continue;
                rotated = true;
                const double zeta = (be - al) / (2 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                const double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
                for (int64_t j = 0; j < L; ++j) {
                    const double ap = A[p * lda + j], aq = A[q * lda + j];
                    A[p * lda + j] = cs * ap - sn * aq; A[q * lda + j] = sn * ap + cs * aq;
                    const double gp = G[p * L + j], gq = G[q * L + j];
                    G[p * L + j] = cs * gp - sn * gq; G[q * L + j] = sn * gp + cs * gq;
                }
            }
        if (!rotated) break;
    }
    std::vector<double> nrm(L);
    std::vector<int64_t> idx(L);
    for (int64_t i = 0; i < L; ++i) { double a = 0; for (int64_t j = 0; j < L; ++j) a += A[i * lda + j] * A[i * lda + j]; nrm[i] = std::sqrt(a); idx[i] = i; }
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return nrm[a] < nrm[b]; });
    for (int64_t j = 0; j < L; ++j) {
        s_inv[j] = nrm[idx[j]] > 0 ? 1.0 / nrm[idx[j]] : 0.0;
        for (int64_t i = 0; i < L; ++i) U[i * ldu + j] = G[idx[j] * L + i];
    }
}
void op_refill_zero_cols(Dev*, double* Y, int64_t rows, int64_t cols, int64_t ldy, const double* Src, int64_t lds) {
    for (int64_t j = 0; j < cols; ++j) {
        bool zero = true;
        for (int64_t i = 0; i < rows && zero; ++i) zero = Y[i * ldy + j] == 0.0;
        if (zero)
            for (int64_t i = 0; i < rows; ++i) Y[i * ldy + j] = Src[i * lds + j];
    }
}
void op_dscal(Dev*, double* x, int64_t count, double alpha) {
    for (int64_t i = 0; i < count; ++i) x[i] *= alpha;
}
void op_daxpy(Dev*, int64_t count, double alpha, const double* x, double* y) {
    for (int64_t i = 0; i < count; ++i) y[i] += alpha * x[i];
}
void op_dvec(Dev*, int mode, const double* x, double* y, int64_t count, double thr) {
    const double x0 = count ? x[0] : 0;
    for (int64_t i = 0; i < count; ++i) {
        if (mode == 0) y[i] = std::sqrt(std::max(x[i], 0.0));
        else if (mode == 2) y[i] = x[i] * x[i];
        else y[i] = (x[i] > thr * x0 && x[i] > 0) ? 1.0 / x[i] : 0.0;
    }
}
void op_dscale_cols(Dev*, double* A, int64_t M, int64_t N, int64_t lda, const double* s) {
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) A[i * lda + j] *= s[j];
}
void op_cvt_from_f64(Dev*, int dt, void* dst, const double* src, int64_t count) {
    for (int64_t i = 0; i < count; ++i) st(dst, dt, i, src[i]);
}
void op_pad_to_f64(Dev*, int dt, double* dst, int64_t rows_p, int64_t cols_p, const void* src, int64_t rows, int64_t cols, int64_t lds,
                   double* zero_ptr, int64_t zero_count) {
    for (int64_t i = 0; i < zero_count; ++i) zero_ptr[i] = 0.0;
    for (int64_t r = 0; r < rows_p; ++r)
        for (int64_t c = 0; c < cols_p; ++c) dst[r * cols_p + c] = (r < rows && c < cols) ? ld(src, dt, r * lds + c) : 0.0;
}
void op_cvt_to_f64(Dev*, int dt, double* dst, const void* src, int64_t count) {
    for (int64_t i = 0; i < count; ++i) dst[i] = ld(src, dt, i);
}

void op_colmean(Dev* d, int dt, const void* X, int64_t n, int64_t dd, int64_t ldx, double n_total, double* mu64, void* muT, bool with_sq) {
    op_colsum(d, dt, X, n, dd, ldx, mu64, with_sq);
    for (int64_t j = 0; j < dd; ++j) mu64[j] /= n_total;
    op_cvt_from_f64(d, dt, muT, mu64, dd);
}
void op_gemm_xp_prod(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* A, int64_t M,
                     int64_t lda, const double* T, int64_t N, int64_t ldt, double* P_out, int64_t ldpo, void* Z, int64_t ldz) {
    std::vector<double> tmp;
    double* P = P_out;
    int64_t ldp = ldpo;
    if (!P) { tmp.resize(size_t(K) * N); P = tmp.data(); ldp = N; }
    op_dgemm(d, false, false, K, N, M, 1.0, A, lda, T, ldt, 0.0, P, ldp);
    op_gemm_xp(d, dt, X, n, K, ldx, mu, P, N, ldp, nullptr, Z, ldz, nullptr);
}
void op_gemm_xp_absmax(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N,
                       int64_t ldp, void* Z, int64_t ldz, int64_t row_offset, double* absmax, double* idx, double* sign, bool) {
    op_gemm_xp(d, dt, X, n, K, ldx, mu, P, N, ldp, nullptr, Z, ldz, nullptr);
    op_col_absmax(d, dt, Z, n, N, ldz, row_offset, absmax, idx, sign);
}
void op_gemm_xp_prod_absmax(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* A, int64_t M,
                            int64_t lda, const double* T, int64_t N, int64_t ldt, double* P_out, int64_t ldpo, void* Z, int64_t ldz,
                            int64_t row_offset, double* absmax, double* idx, double* sign, bool, bool a_rt) {
    if (a_rt) throw std::logic_error("host sim: op_chol_rt never returns the RT form");
    op_gemm_xp_prod(d, dt, X, n, K, ldx, mu, A, M, lda, T, N, ldt, P_out, ldpo, Z, ldz);
    op_col_absmax(d, dt, Z, n, N, ldz, row_offset, absmax, idx, sign);
}
// (the simulation always forms the explicit inverse: the RT form is a device detail)
bool op_chol_rt(Dev* d, int, int64_t, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead, int64_t Lz,
                int64_t ndead_cols) {
    op_chol_inv(d, G, L, ldg, T, ldt, rel_tol, ndead, Lz, ndead_cols);
    return false;
}
void op_trsm_right(Dev*, const double*, int64_t, int64_t, const double*, int64_t, int64_t, double*, int64_t) {
    throw std::logic_error("host sim: op_trsm_right without an RT-form factor");
}
void op_rebase_xp(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* G, int64_t L,
                  int64_t ldg, double rel_tol, int* ndead, const double* A, int64_t M, int64_t lda, double* T, int64_t ldt,
                  double* P_out, int64_t ldpo, void* Z, int64_t ldz, int p_planes, bool steering) {
    op_chol_inv(d, G, L, ldg, T, ldt, rel_tol, ndead, M);
    if (dt == F32 && d->gemm_mode == 0 && p_planes == 2) {   // the two-plane iterate: P_out itself is rounded, every later use sees it
        op_dgemm(d, false, false, K, M, M, 1.0, A, lda, T, ldt, 0.0, P_out, ldpo);
        for (int64_t k = 0; k < K; ++k)
            for (int64_t j = 0; j < M; ++j) P_out[k * ldpo + j] = two_plane(P_out[k * ldpo + j]);
        op_gemm_xp(d, dt, X, n, K, ldx, mu, P_out, M, ldpo, nullptr, Z, ldz, nullptr, 2, steering);
        return;
    }
    op_gemm_xp_prod(d, dt, X, n, K, ldx, mu, A, M, lda, T, M, ldt, P_out, ldpo, Z, ldz);
}

// the fused power-iteration pass: simulated for fp32 data in the split-product mode at ANY width (the device kernel exists for
// K = 512, N <= 80), so that the host sequencing of the fused pipeline is covered by the CPU suite
bool op_power_pass_applies(Dev* d, int dt, const void*, int64_t n, int64_t K, int64_t, const void*, int64_t N) {
    const bool off = d->opt[OPT_FUSED_PASS] == 0;
    return !off && dt == F32 && d->gemm_mode == 0 && n >= 64 && K % 16 == 0 && N % 16 == 0 && N <= 80;
}
bool op_power_pass(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N, int64_t ldp,
                   void* Z, int64_t ldz, double* Y, int64_t ldy, bool steering) {
    if (!op_power_pass_applies(d, dt, X, n, K, ldx, mu, N)) return false;
    const bool no_fast = d->opt[OPT_STEERING] == 0;
    if (steering && !Z && !no_fast) {
        // the device's steering pass (k_pow3f): the centred X, the iterate and z each rounded to two bf16 planes
        std::vector<double> xc(size_t(n) * K), z(size_t(n) * N, 0.0), p2(size_t(K) * N);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t f = 0; f < K; ++f) xc[i * K + f] = two_plane(double(float(centred(X, dt, i * ldx + f, mu, f))));
        for (int64_t f = 0; f < K; ++f)
            for (int64_t j = 0; j < N; ++j) p2[f * N + j] = two_plane(P[f * ldp + j]);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t f = 0; f < K; ++f) {
                const double a = xc[i * K + f];
                if (a == 0.0) continue;
                for (int64_t j = 0; j < N; ++j) z[i * N + j] += a * p2[f * N + j];
            }
        for (auto& v : z) v = two_plane(double(float(v)));
        for (int64_t f = 0; f < K; ++f)
            for (int64_t j = 0; j < N; ++j) Y[f * ldy + j] = 0.0;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t f = 0; f < K; ++f) {
                const double a = xc[i * K + f];
                if (a == 0.0) continue;
                for (int64_t j = 0; j < N; ++j) Y[f * ldy + j] += a * z[i * N + j];
            }
        return true;
    }
    std::vector<float> ztmp;
    if (!Z) { ztmp.resize(size_t(n) * N); Z = ztmp.data(); ldz = N; }
    op_gemm_xp(d, dt, X, n, K, ldx, mu, P, N, ldp, nullptr, Z, ldz, nullptr, 2);
    op_gemm_atb(d, dt, X, ldx, K, mu, Z, ldz, N, nullptr, n, Y, ldy, false);
    return true;
}
bool op_power_pass_means(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t dcols, int64_t ldx, double n_total, const double* P,
                         int64_t N, int64_t ldp, int64_t L, double* Y, int64_t ldy, double* mu64, void* muT, double* ssq_scratch, double* tv) {
    const bool off = d->opt[OPT_MEANS_FOLD_ROWS] < 0;
    if (off || L >= N || !op_power_pass_applies(d, dt, X, n, K, ldx, muT, N)) return false;
    // as on the device: a provisional centre from a strided row sample, the exact sums about it from the pass, then the move
    const int64_t ns = std::min<int64_t>(n, 4096), stride = n / ns;
    op_colmean(d, dt, X, ns, K, ldx * stride, double(ns), mu64, muT, false);
    std::vector<double> sums(K, 0.0);
    double ssq = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t f = 0; f < K; ++f) { const double v = centred(X, dt, i * ldx + f, muT, f); sums[f] += v; ssq += v * v; }
    if (!op_power_pass(d, dt, X, n, K, ldx, muT, P, N, ldp, nullptr, 0, Y, ldy, /*steering=*/true)) return false;   // (as on the device)
    std::vector<double> t(N, 0.0);
    double q = 0;
    for (int64_t f = 0; f < K; ++f) {
        const double del = f < dcols ? sums[f] / n_total : 0.0;
        sums[f] = del;
        q += del * del;
        for (int64_t j = 0; j < L; ++j) t[j] += del * two_plane(P[f * ldp + j]);
    }
    for (int64_t f = 0; f < K; ++f) {
        for (int64_t j = 0; j < N; ++j) Y[f * ldy + j] = j == N - 1 ? 0.0 : Y[f * ldy + j] - n_total * sums[f] * t[j];
        mu64[f] += sums[f];
        st(muT, dt, f, mu64[f]);
    }
    *ssq_scratch = ssq;
    *tv = std::max(ssq - n_total * q, 0.0);
    return true;
}
bool op_rebase_power_pass(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* G, int64_t L,
                          int64_t ldg, double rel_tol, int* ndead, const double* A, int64_t M, int64_t lda, double* T, int64_t ldt,
                          double* P_out, int64_t ldpo, void* Z, int64_t ldz, double* Y, int64_t ldy, bool steering) {
    if (!op_power_pass_applies(d, dt, X, n, K, ldx, mu, M) || L == 0 || P_out == nullptr) return false;
    op_chol_inv(d, G, L, ldg, T, ldt, rel_tol, ndead, M);
    op_dgemm(d, false, false, K, M, M, 1.0, A, lda, T, ldt, 0.0, P_out, ldpo);
    for (int64_t k = 0; k < K; ++k)
        for (int64_t j = 0; j < M; ++j) P_out[k * ldpo + j] = two_plane(P_out[k * ldpo + j]);
    return op_power_pass(d, dt, X, n, K, ldx, mu, P_out, M, ldpo, Z, ldz, Y, ldy, steering);
}
void op_tail_verdict(Dev*, const double* lam, int64_t L, int64_t k, const double* mu_sq, int64_t dp, int64_t d, double n_total,
                     const double* tv, double eps2, double thr, int* flag2) {
    if (L <= 0 || k <= 0) return;
    double total = tv ? *tv : 0.0, head = 0;
    if (mu_sq) { total = 0; for (int64_t j = 0; j < d; ++j) total += std::max(0.0, mu_sq[dp + j] - n_total * mu_sq[j] * mu_sq[j]); }
    for (int64_t j = 0; j < L; ++j) head += std::max(lam[j], 0.0);
    const double T = std::sqrt(std::max(lam[L - 1], 0.0) * std::max(total - head, 0.0) / double(d));
    for (int64_t j = 0; j < k && j < L; ++j) {
        const double lj = lam[j];
        if (!(lj > 0.0)) continue;
        double gap = j + 1 < L ? lj - std::max(lam[j + 1], 0.0) : lj;
        if (j > 0) gap = std::min(gap, lam[j - 1] - lj);
        if (eps2 * T / lj / std::max(gap / lj, 1e-3) > thr) flag2[1] = 1;
    }
}

// ---- FastICA ---------------------------------------------------------------------------------
void op_ica_prepare(Dev*, int, const void*, int64_t, int64_t, int64_t) {}
void op_symdecorr(Dev* d, int64_t nc, const double* Win, double* Wout, int mode, int* zero2) {
    if (zero2) { zero2[0] = 0; zero2[1] = 0; }
    std::vector<double> S(size_t(nc) * nc), Z(size_t(nc) * nc), w(nc), M(size_t(nc) * nc);
    op_dgemm(d, false, true, nc, nc, nc, 1.0, Win, nc, Win, nc, 0.0, S.data(), nc);  // W W^T (ica.rs:369)
    op_eigh(d, S.data(), nc, nc, Z.data(), nc, w.data());                            // columns of Z = eigenvectors
    // textbook: (Z D Z^T)_ij = sum_k Z[i][k] s_k Z[j][k]   (eigenvector k = column k of Z)
    // literal (ica.rs:370-380): v = heev buffer read row-major = Z_asc^T (LAPACK order: ascending),
    // v's COLUMN c is scaled by s_c, then v . v_saved^T  =>  (Z_asc^T D Z_asc)_ij = sum_c Z_asc[c][i] s_c Z_asc[c][j].
    // That form depends on LAPACK's eigenvector signs; for nc == 2 LAPACK's 2x2 solver (dlaev2 + ascending
    // sort) returns a SYMMETRIC Z for any PSD input, which makes the literal form equal the textbook one.
    const bool literal = mode == 1 && nc > 2;
    std::vector<double> sg(nc, 1.0);  // sign normalisation of the eigenvectors (first largest component positive), see hip_ops.hip
    if (literal)
        for (int64_t c = 0; c < nc; ++c) {
            double best = -1.0;
            for (int64_t k = 0; k < nc; ++k) {
                const double v = Z[k * nc + c];
                if (std::fabs(v) > best) { best = std::fabs(v); sg[c] = v < 0.0 ? -1.0 : 1.0; }
            }
        }
    for (int64_t i = 0; i < nc; ++i)
        for (int64_t j = 0; j < nc; ++j) {
            double acc = 0;
            for (int64_t k = 0; k < nc; ++k) {
                if (literal) {
                    const double sk = 1.0 / std::sqrt(w[nc - 1 - k]);  // ascending eigenvalue k
                    acc += Z[k * nc + (nc - 1 - i)] * sk * Z[k * nc + (nc - 1 - j)];
                } else {
                    acc += Z[i * nc + k] * (1.0 / std::sqrt(w[k])) * Z[j * nc + k];
                }
            }
            M[i * nc + j] = literal ? acc * sg[nc - 1 - i] * sg[nc - 1 - j] : acc;
        }
    op_dgemm(d, false, false, nc, nc, nc, 1.0, M.data(), nc, Win, nc, 0.0, Wout, nc);
}

void op_ica_step(Dev*, int dt, const void* X1T, int64_t n, int64_t nc, int64_t ldx, const double* W, double* GX_gp,
                 const int* state) {
    if (state && state[0]) return;
    double* GX = GX_gp;
    double* gp = GX_gp + nc * nc;
    for (int64_t i = 0; i < nc * nc + nc; ++i) GX_gp[i] = 0;
    std::vector<double> x(nc);
    for (int64_t s = 0; s < n; ++s) {
        for (int64_t j = 0; j < nc; ++j) x[j] = ld(X1T, dt, s * ldx + j);
        for (int64_t i = 0; i < nc; ++i) {
            double wx = 0;
            for (int64_t j = 0; j < nc; ++j) wx += (dt == F32 ? double(float(W[i * nc + j])) : W[i * nc + j]) * x[j];
            const double g = std::tanh(wx);
            gp[i] += 1.0 - g * g;
            for (int64_t j = 0; j < nc; ++j) GX[i * nc + j] += g * x[j];
        }
    }
}

volatile int* dev_host_progress(Dev*) { static int p[4]; return p; }
void op_ica_tail(Dev* d, int64_t nc, double n_total, double* W, const double* GX_gp, int mode, double tol, int* state,
                 int iter, int* progress) {
    if (state[0]) return;
    const double* GX = GX_gp;
    const double* gp = GX_gp + nc * nc;
    std::vector<double> D(size_t(nc) * nc), W1(size_t(nc) * nc);
    const double pinv = 1.0 / n_total;
    for (int64_t i = 0; i < nc; ++i)
        for (int64_t j = 0; j < nc; ++j) D[i * nc + j] = GX[i * nc + j] * pinv - gp[i] * pinv * W[i * nc + j];  // ica.rs:334-342
    op_symdecorr(d, nc, D.data(), W1.data(), mode);  // ica.rs:343
    double lim = 0;
    for (int64_t i = 0; i < nc; ++i) {
        double dot = 0;
        for (int64_t j = 0; j < nc; ++j) dot += W1[i * nc + j] * (mode == 1 ? W[j * nc + i] : W[i * nc + j]);  // ica.rs:345-349
        const double v = std::fabs(std::fabs(dot) - 1.0);
        if (v > lim) lim = v;
    }
    for (int64_t i = 0; i < nc * nc; ++i) W[i] = W1[i];
    if (lim < tol) { state[0] = 1; state[1] = iter + 1; }  // ica.rs:355-357
    if (progress) { progress[1] = iter + 1; if (state[0]) progress[0] = iter + 1; }
}

}  // namespace petal
