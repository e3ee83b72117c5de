// ops.h -- the device-operation layer the host algorithms (algo.cpp) are written against.
//
// The product library links exactly one implementation: hip_ops.hip (hand-written gfx950 kernels).
// tests/ additionally build a host-memory simulation of the same functions from oracle/cpu_ops.cpp
// (TEST INFRASTRUCTURE, never linked into the product) so that the host algorithms and the
// sample-sharded collective path can be exercised without a GPU (world_size-2 gloo test).
//
// All matrices are row-major.  "f64 small" matrices live in device memory as doubles: everything
// that is O(l^2), O(d l) or O(nc^2) is kept in fp64 so that only the O(n) streams are fp32.
#pragma once
#include <cstddef>
#include <cstdint>

namespace petal {

enum DType : int { F32 = 0, F64 = 1 };
inline size_t dtype_size(int dt) { return dt == F64 ? 8 : 4; }

struct Dev;  // opaque: stream, allocator cache, event pool (hip) / nothing much (cpu sim)

// hot-kernel tags: launches issued while a tag is set are bracketed with events when profiling is on
// (TAG_STREAM: the other kernels of a fit whose time falls with the rows a rank holds -- the means pass, U = Z (T Uh) -- so that
// fit time minus every tagged kernel is the REPLICATED small-matrix chain, the part of a sharded fit that does not shrink with N)
enum Tag : int { TAG_NONE = 0, TAG_XP = 1, TAG_ATB = 2, TAG_ICA = 3, TAG_COMM = 4, TAG_POW = 5, TAG_STREAM = 6, TAG_COUNT = 7 };
struct KernelTiming {
    double ms[TAG_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    int64_t launches[TAG_COUNT] = {0, 0, 0, 0, 0, 0, 0};
};

// ---- lifetime / memory ---------------------------------------------------------------------
Dev*  dev_create(int device, void* stream, char* err, size_t errlen);  // nullptr on failure
void  dev_destroy(Dev*);
void* dev_stream(Dev*);
void* dev_alloc(Dev*, size_t bytes);      // cached; never returns nullptr (throws std::runtime_error)
void  dev_free(Dev*, void*);
void  dev_memset(Dev*, void* p, int v, size_t bytes);
void  dev_h2d(Dev*, void* dst, const void* src, size_t bytes);
void  dev_h2d_async(Dev*, void* dst, const void* src, size_t bytes);  // src is consumed before the call returns; no host wait
void  dev_d2h(Dev*, void* dst, const void* src, size_t bytes);  // dst is valid after the next dev_sync
void  dev_d2h_multi(Dev*, int nseg, void* const* dst, const void* const* src, const size_t* bytes);   // several dev_d2h in one launch (nseg <= 8)
const void* dev_h2d_view(Dev*, const void* src, size_t bytes);   // host bytes staged in the pinned ring; returns its device-visible address (nullptr: too large), valid until the next dev_sync
const void* dev_d2h_view(Dev*, const void* src, size_t bytes);   // the result where the copy lands (pinned ring): readable after the next dev_sync, until the next copy is queued
size_t dev_view_limit(Dev*);                                     // the largest dev_d2h_view / dev_h2d_view (one ring slot); dev_d2h_view THROWS above it
void  dev_d2d(Dev*, void* dst, const void* src, size_t bytes);
// pitched copies (bytes); kind: 0 h2d, 1 d2h, 2 d2d
void  dev_copy2d(Dev*, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int kind);
void  dev_sync(Dev*);
void  dev_set_profiling(Dev*, int level);   // 0 off, 1 one sampled launch per tag and fit, 2 every tagged launch
// Options of a device context: every switch that selects WHICH arithmetic or kernel form a fit runs.  The defaults come from the
// environment ONCE, at dev_create (the variable named with each); afterwards only dev_set_option (petal_ctx_set_option) changes
// them -- no kernel launcher reads the environment.  The values equal include/petal_hip.h's PETAL_OPT_*.
enum PetalOpt {
    OPT_TWO_PLANE = 0,         // (1) optimistic first run of an fp32 RandomizedPca fit: sketch matrix and re-based iterates on two bf16 planes,
                               //     behind the spectrum verdict; 0 = three planes throughout                       env PETAL_NO_P2=1 -> 0
    OPT_TWO_PLANE_OMEGA = 1,   // (1) ... the sketch matrix alone                                                     env PETAL_NO_P2_OMEGA=1 -> 0
    OPT_TWO_PLANE_ITERATE = 2, // (1) ... the re-based iterates alone                                                 env PETAL_NO_P2_ITERATE=1 -> 0
    OPT_STEERING = 3,          // (1) the passes before the last one round Xc and z to two planes as well             env PETAL_NO_POW3_FAST=1 -> 0
    OPT_FUSED_PASS = 4,        // (1) the fused power-iteration pass where the kernel exists                          env PETAL_NO_POW3=1 -> 0
    OPT_FUSED_PASS_MIN_ROWS = 5, // (8192) fewer rows: K1 + K2                                                        env PETAL_POW3_MIN_ROWS
    OPT_VERDICT_THRESHOLD = 6, // (4e-6) the two-plane verdict redoes a fit whose estimated component error exceeds it  env PETAL_P2_VERDICT_THR
    OPT_MEANS_FOLD_ROWS = 7,   // (200000) single-rank fp32 fits gather the means inside the first pass from this many rows on; < 0: never
                               //                                                      env PETAL_MEANS_FOLD_ROWS, PETAL_NO_MEANS_FOLD=1 -> -1
    OPT_GRAM_SPLIT = 8,        // (1) FastICA whitening: the optimistic split-product covariance (k_gram5); 0 = fp64 products   env PETAL_NO_GRAM3=1 -> 0
    OPT_GRAM_SPLIT_HOOK = 9,   // (0) test hook: petal_gemm_atb sends Gram products to the split-product Gram kernel  env PETAL_GRAM_SPLIT=1
    OPT_D2H_KERNEL = 10,       // (1) small results leave through a copy kernel into the pinned ring; 0 = hipMemcpyAsync  env PETAL_D2H_MEMCPY=1 -> 0
    OPT_ROW_PAD = 11,          // (1) copied inputs land with 128 B of row padding when the natural pitch is a multiple of 1 KiB  env PETAL_NO_ROW_PAD=1 -> 0
    OPT_EIGH_JACOBI = 12,      // (0) symmetric eigenproblems go straight to the Jacobi solvers                        env PETAL_EIGH_JACOBI=1
    OPT_POISON = 13,           // (0) every workspace block is filled with NaN patterns when handed out (tests)         env PETAL_POISON=1
    OPT_COUNT = 14
};
void   dev_set_option(Dev*, int opt, double value);
double dev_option(const Dev*, int opt);
void  dev_set_gemm_mode(Dev*, int mode);    // 0 = split-product (bf16x3) GEMM kernels for fp32 data, 1 = fp32-MFMA kernels
int   dev_gemm_mode(const Dev*);
void  dev_make_current(Dev*);               // hipSetDevice(the ctx's device) on the calling thread
int   dev_push_current(Dev*);               // the same, returning the thread's previous device (-1: none) ...
void  dev_pop_current(Dev*, int prev);      // ... which this restores: every ABI entry point brackets its work with the pair
void  dev_abort(Dev*);                     // error path: wait for the stream, drop queued device-to-host hand-overs
void  dev_reset_timing(Dev*);
void  dev_set_tag(Dev*, int tag);
// side stream for work that does not depend on the main chain: launches between dev_fork and dev_fork_end go to it (behind
// everything queued so far; at once with after_main = false), dev_join makes the main stream wait for it; dev_fork_abort is the
// error path
void  dev_fork(Dev*, bool after_main = true);
void  dev_fork_end(Dev*);
void  dev_join(Dev*);
void  dev_fork_abort(Dev*);
KernelTiming dev_timing(Dev*);            // resolves pending events (call after dev_sync)
// Brackets stream work that is not one of this file's kernels (the collective) with the same event machinery as the tagged
// kernels: begin returns a token (nullptr when this launch is not sampled), end records the closing event.
void* dev_span_begin(Dev*, int tag);
void  dev_span_end(Dev*, void* token);

// ---- O(n) streaming ops --------------------------------------------------------------------
// dst[i*ld_dst + j] = src[i*rs + j*cs] for j < d, 0 for d <= j < d_pad   (device -> device gather)
void op_pack_strided(Dev*, int dtype, const void* src, int64_t n, int64_t d, int64_t rs, int64_t cs,
                     void* dst, int64_t ld_dst, int64_t d_pad);
// dst[i*rs + j*cs] = src[i*ld_src + j] (device -> device scatter), j < d
void op_unpack_strided(Dev*, int dtype, const void* src, int64_t n, int64_t d, int64_t ld_src,
                       void* dst, int64_t rs, int64_t cs, const double* scale = nullptr);   // scale (device, d doubles): dst = src * scale[j]
// out[j] = sum_i X[i][j]  (fp64, deterministic order)
// with_sq: out has 2 d entries, [column sums | column sums of squares]
void op_colsum(Dev*, int dtype, const void* X, int64_t n, int64_t d, int64_t ldx, double* out, bool with_sq = false);
// single-rank column means in one go: mu64[j] = (sum_i X_ij) / n_total, mu64[d + j] = sum_i X_ij^2 (with_sq), muT = the means in dtype
void op_colmean(Dev*, int dtype, const void* X, int64_t n, int64_t d, int64_t ldx, double n_total, double* mu64, void* muT, bool with_sq);
// Z[n x N] = (X[n x K] - mu) . P[K x N] * 1 + bias        (mu, bias nullable; mu/bias in dtype)
// P is an f64 small matrix (ldp).  sumsq (nullable, fp64 scalar): += sum_ij (X_ij - mu_j)^2.
// colscale (nullable, f64[N]): Z[:, j] *= colscale[j].
// p_planes = 2: the caller accepts P rounded to the sum of its two leading bf16 pieces (a random sketch matrix: any matrix
// serves) -- the split-product kernel then forms five piece products instead of six; other paths ignore it.
void op_gemm_xp(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu,
                const double* P, int64_t N, int64_t ldp, const void* bias,
                void* Z, int64_t ldz, double* sumsq, int p_planes = 3, bool steering = false);
// steering (op_gemm_xp, op_rebase_xp with p_planes = 2; op_gemm_atb): the product only STEERS a power iteration -- the split-product
// kernels may round the LARGE operands (Xc; Xc and Z) to two bf16 planes as well: four piece products per tile (the forms for more than
// 80 columns; see op_power_pass).
// op_gemm_xp with P = A . T formed on the fly (A: K x M, lda; T: M x N, ldt; both fp64 small matrices): the re-basing
// product Y = Yp T of the power iteration goes straight into the GEMM kernel's operand planes instead of through a GEMM
// launch of its own.  P_out (nullable, K x N fp64, ldpo) also receives the product.
void op_gemm_xp_prod(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu,
                     const double* A, int64_t M, int64_t lda, const double* T, int64_t N, int64_t ldt,
                     double* P_out, int64_t ldpo, void* Z, int64_t ldz);
// op_gemm_xp_prod followed by op_col_absmax over the N columns of the product Z (svd_flip's scan, pca.rs:826-839), the scan taken
// from the product kernel's accumulators where that kernel runs (no second pass over Z).
void op_gemm_xp_prod_absmax(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu,
                            const double* A, int64_t M, int64_t lda, const double* T, int64_t N, int64_t ldt,
                            double* P_out, int64_t ldpo, void* Z, int64_t ldz,
                            int64_t row_offset, double* absmax, double* idx, double* sign, bool store_product = true, bool a_rt = false);
// (a_rt: A is not a matrix to multiply with but an upper-triangular R in RT form, as op_chol_rt left it -- P = R^-1 T by blocked back
//  substitution; only after op_chol_rt returned true)
// (store_product = false: Z may be left unwritten where the scan comes out of the product kernel's accumulators)
// op_gemm_xp with the same scan (absmax / idx / sign: N doubles each, as op_col_absmax delivers them over the leading N columns)
void op_gemm_xp_absmax(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N,
                       int64_t ldp, void* Z, int64_t ldz, int64_t row_offset, double* absmax, double* idx, double* sign,
                       bool store_product = true);
// One re-basing step of the power iteration: G (L x L, ldg) = R^T R, P_out (K x M fp64, ldpo) = A R^-1 (A: K x M, lda; columns
// L .. M of the result are zero), Z = (X - mu) . P_out.  Same results contract as op_chol_inv(G -> T, Lz = M) followed by
// op_gemm_xp_prod(A, T); T (M x M, ldt) is SCRATCH here -- it may hold R^-1 or a factored form of it, callers must not read it.
// p_planes = 2: P_out may be ROUNDED to the sum of its two leading bf16 pieces (in P_out itself and in the product's operand) where
// that saves a piece product; 3: P_out is the re-based iterate to fp64 / fp32 accuracy.
void op_rebase_xp(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* G, int64_t L,
                  int64_t ldg, double rel_tol, int* ndead, const double* A, int64_t M, int64_t lda, double* T, int64_t ldt,
                  double* P_out, int64_t ldpo, void* Z, int64_t ldz, int p_planes = 2, bool steering = false);
// One FUSED power-iteration pass, the two products of pca.rs:711 + 714 in ONE pass over X:
//     Y (K x N fp64, ldy) = (X - mu)^T ((X - mu) P),     Z (nullable, n x N, ldz) = (X - mu) P
// with P rounded to the sum of its two leading bf16 pieces (a caller that needs P to fp32 accuracy uses op_gemm_xp + op_gemm_atb).
// Returns false, NOTHING done, where no fused kernel exists for the shape / mode (op_power_pass_applies says so up front).
bool op_power_pass_applies(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, int64_t N);
// steering: the pass only STEERS a power iteration (its Y feeds the next re-basing, nothing else; Z == NULL): the kernel may then also
// round Xc and z to two bf16 planes -- four piece products per tile in both products, two barriers per stage (k_pow3f).  The caller
// owns the consequences: rpca_fit runs such passes on its optimistic path only, under the spectral verdict.
bool op_power_pass(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N, int64_t ldp,
                   void* Z, int64_t ldz, double* Y, int64_t ldy, bool steering = false);
// The FIRST fused pass of a fit with the means pass folded in (single rank, fp32, centring; L < N: a padding column is free):
// mu64 / muT come back as the column means of X (d real columns of K), *tv as sum (X - mean)^2, and Y = Xc^T (Xc P) about that
// mean -- X is read once, not twice: the kernel centres about the means of a row SAMPLE, gathers the exact column sums and the sum
// of squares about that provisional centre in the same pass, and a one-workgroup kernel moves Y, the means and the variance to the
// true centre (a rank-one correction of relative size (delta / sigma)^2).  ssq_scratch: one device double.  False: nothing done.
bool op_power_pass_means(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t d, int64_t ldx, double n_total, const double* P,
                         int64_t N, int64_t ldp, int64_t L, double* Y, int64_t ldy, double* mu64, void* muT, double* ssq_scratch, double* tv);
// the same behind one re-basing step (arguments as op_rebase_xp, p_planes = 2): P_out = A R^-1 rounded, then the fused pass with it
bool op_rebase_power_pass(Dev*, int dtype, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* G, int64_t L,
                          int64_t ldg, double rel_tol, int* ndead, const double* A, int64_t M, int64_t lda, double* T, int64_t ldt,
                          double* P_out, int64_t ldpo, void* Z, int64_t ldz, double* Y, int64_t ldy, bool steering = false);
// Was a 16-bit (two-plane) rounding of the sketch matrix and of the re-based iterates harmless for the spectrum this fit found?
// A rounding E of the basis P (|E_ij| <= eps2 |P_ij|) reaches the next iterate as C E, C = Xc^T Xc: directions the block already
// spans are harmless, the part from BEYOND the block -- sum_{i > L} sigma_i^2 v_i (v_i^T E), of size eps2 T with
// T^2 = sum_{i > L} sigma_i^4 / d <= lam_{L-1} (tv - sum_j lam_j) / d -- tilts wanted direction j by eps2 T / lam_j, and the
// component that comes out by that over the relative gap to its neighbours.  lam (L, descending): the eigenvalues of B B^T (squared
// singular values); mu_sq ([means | sums of squares], 2 dp; nullable): the total variance is sum_j (sq_j - n_total mu_j^2), else *tv.
// Sets flag2[1] = 1 when that estimate exceeds thr for one of the k wanted components (flag2[0] is not touched).
void op_tail_verdict(Dev*, const double* lam, int64_t L, int64_t k, const double* mu_sq, int64_t dp, int64_t d, double n_total,
                     const double* tv, double eps2, double thr, int* flag2);
// C[M x N] (f64, ldc) = (A - muA)^T . (B - muB),  A: n x M (lda), B: n x N (ldb), reduction over n rows.
// precise: every product and the whole accumulation in fp64 (needed where the result's small eigenvalues
// matter: exact Pca, FastICA whitening); otherwise fp32 MFMA chunks combined in fp64.
void op_gemm_atb(Dev*, int dtype, const void* A, int64_t lda, int64_t M, const void* muA,
                 const void* B, int64_t ldb, int64_t N, const void* muB, int64_t n,
                 double* C, int64_t ldc, bool precise = false, bool steering = false);
// C (dp x dp fp64, ldc; only the leading d x d block is non-zero) = (X - mu)^T (X - mu) for fp32 X by EXACT bf16-piece products with
// fp32 accumulation over row chunks (fp64 across them): good to ~1e-6 of C's largest entries -- enough for a whitening whose wanted
// eigenvalues lie within two decades (the caller checks), not for an exact Pca.  False, nothing done: shape / mode not covered.
// mu64_fold != NULL: the column means are formed by the same call (mu: provisional centre in, true means out; mu64_fold: true means, fp64)
bool op_gram_split(Dev*, const void* X, int64_t n, int64_t d, int64_t dp, int64_t ldx, const void* mu, double* C, int64_t ldc,
                   double* mu64_fold = nullptr, double n_total = 0.0);
// per column j < L of U (n x L): absmax[j] = max_i |U_ij|, idx[j] = row_offset + first such i,
// sign[j] = U_ij >= 0 ? +1 : -1 (sign of -0.0 / 0.0 follows f64::signum: +1 for +0, -1 for -0).
// n == 0: absmax = -1, idx = +inf, sign = +1.
void op_col_absmax(Dev*, int dtype, const void* U, int64_t n, int64_t L, int64_t ldu, int64_t row_offset,
                   double* absmax, double* idx, double* sign);
// svd_flip across ranks, fp32 data: key[j] = the fp64 image of absmax[j] with (2^28 - 1 - row[j], sign[j] < 0) packed into
// the 29 low mantissa bits an fp32 magnitude leaves zero (absmax < 0, an empty shard, gives key 0); `triple` is
// op_col_absmax's [absmax | row | sign] (3 L).  A MAX all-reduce of the keys elects the first element of maximal magnitude.
// flag != nullptr (three ints): key[L] = flag[0] != 0 ? 3 : flag[1] != 0 ? 1 : flag[2] != 0 ? 2 : 0 -- a replicated decision word riding the same MAX
// all-reduce, so that every rank branches on the agreed value; the stronger redo (2) wins
void op_flip_key(Dev*, const double* triple, double* key, int64_t L, const int* flag = nullptr);
// A[i][j] *= s[j] (dtype matrix, f64 scale vector), i < n, j < L
void op_scale_cols(Dev*, int dtype, void* A, int64_t n, int64_t L, int64_t lda, const double* s);
// G = tanh(X) elementwise (n x c), gp[j]... logcosh KAT helper: rows are components:
// G[i][j] = tanh(X[i][j]); gp[i] = sum_j (1 - G_ij^2)   (f64 out, un-normalised)
void op_logcosh_rows(Dev*, int dtype, const void* X, int64_t r, int64_t c, int64_t ldx, void* G, int64_t ldg, double* gp);

// ---- FastICA fused step (ica.rs:332-333) -----------------------------------------------------
// X1T: n x nc (ld) whitened samples (sample-major).  W: nc x nc f64.
// GX[i][j] = sum_s tanh(w_i . x_s) x_s[j];  gp[i] = sum_s (1 - tanh(w_i . x_s)^2)     (f64 out)
// state[0] != 0 (converged) -> no-op.
// Call once in front of a fixed-point loop over X1T (and again whenever X1T's CONTENTS change): lets the implementation prepare what
// is constant over the loop's iterations (the device library: the bf16 planes of X1).  op_ica_step works without it.
void op_ica_prepare(Dev*, int dtype, const void* X1T, int64_t n, int64_t nc, int64_t ld);
void op_ica_step(Dev*, int dtype, const void* X1T, int64_t n, int64_t nc, int64_t ld,
                 const double* W, double* GX_gp /* nc*nc + nc contiguous */, const int* state);
// ica.rs:334-358 on one workgroup: D = GX/n_total - gp/n_total (.) W; W1 = symdecorr(D); lim; update.
// state = {done, n_iter}; iter is the 0-based index of this iteration.  W is replaced by W1 unless done.
// progress (nullable): dev_host_progress()'s array; the kernel publishes {n_iter if converged else 0, iter + 1} there when it
// finishes, so the host can follow the loop without a synchronisation.
void op_ica_tail(Dev*, int64_t nc, double n_total, double* W, const double* GX_gp, int mode, double tol,
                 int* state, int iter, int* progress = nullptr);
// four ints of pinned host memory that kernels can store to (system-scope stores over PCIe); the host reads them directly
volatile int* dev_host_progress(Dev*);
// Wout = symmetric_decorrelation(Win) (ica.rs:363-381)
void op_symdecorr(Dev*, int64_t nc, const double* Win, double* Wout, int mode, int* zero2 = nullptr);  // zero2 (nullable): two device ints to clear

// ---- f64 small-matrix ops ----------------------------------------------------------------------
// colscale (nullable, N values): column j of alpha op(A) op(B) is multiplied by colscale[j] (beta must be 0)
void op_dgemm(Dev*, bool ta, bool tb, int64_t M, int64_t N, int64_t K, double alpha,
              const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
              const double* colscale = nullptr);
// sig[i] = sqrt(max(lam[i], 0)); inv[i] = sig[i] > thr * sig[0] ? 1 / sig[i] : 0   (the two op_dvec steps of an SVD from
// eigenvalues, in one launch)
void op_sigma_inv(Dev*, const double* lam, double* sig, double* inv, int64_t count, double thr);
// P (rows x rp row-major) = [ V[:, :r] diag(inv) | 0 ]  (V: ldv)
void op_scale_pad_cols(Dev*, const double* V, int64_t ldv, const double* inv, int64_t rows, int64_t r, int64_t rp, double* P);
// comp (k x d row-major, dtype) = V[:d, :k]^T
void op_transpose_out(Dev*, int dtype, const double* V, int64_t ldv, int64_t d, int64_t k, void* comp);
// comp[j][i] (k x d row-major, the input dtype) = (Bt u_j)[i] / sigma_j: rows of V^T from Bt (d x L, ldb), the eigenvectors Uh
// (columns, ldu) and eigenvalues lam of B B^T; sigma_j = sqrt(max(lam_j, 0)), 1 / sigma_j = 0 at or below thr * sigma_0.
void op_components_out(Dev*, int dtype, const double* Bt, int64_t ldb, const double* Uh, int64_t ldu, const double* lam, double thr,
                       int64_t d, int64_t L, int64_t k, void* comp);
// G (L x L, SPD up to rounding) = R^T R;  T = R^{-1} (upper triangular, L x L, ldt).
// A pivot with r_jj^2 <= rel_tol * G_jj (or G_jj <= 0) marks column j as dependent: T[:, j] = 0.
// Only the upper triangle of G is read.  ndead (nullable, device int): *ndead = max(*ndead, number of dependent columns).
// Lz > L: T is additionally zero-filled out to Lz x Lz (the padded extent of the caller's buffers).
// ndead_cols > 0: only dependent columns j < ndead_cols count towards *ndead (orders the one-workgroup kernel takes; larger
// orders count them all).
// The factorisation of op_chol_inv WITHOUT the explicit inverse where the device has the kernels for it (fp32 data in the split-product
// mode, L <= 140, Lz a multiple of 16 <= 144, 64 <= n < 2^31 rows in the product that follows): returns true and leaves T in "RT form"
// -- diagonal 16 x 16 blocks T_JJ = R_JJ^-1, the blocks above them R itself, zeros below -- which only op_trsm_right and
// op_gemm_xp_prod_absmax(a_rt = true) understand; otherwise it IS op_chol_inv and returns false.  (Round 6: the last factorisation of
// a RandomizedPca fit on the register-resident kernel of the power iterations, 15 instead of 40 us at l = 74.)
bool op_chol_rt(Dev*, int dtype, int64_t n, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead,
                int64_t Lz, int64_t ndead_cols);
// out (rows x M fp64, ldo) = A (rows x M, lda) . R^-1 for R in RT form (M x M, ldt); rows a multiple of 16
void op_trsm_right(Dev*, const double* A, int64_t rows, int64_t lda, const double* RT, int64_t M, int64_t ldt, double* out, int64_t ldo);
void op_chol_inv(Dev*, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead = nullptr,
                 int64_t Lz = 0, int64_t ndead_cols = 0);
// symmetric PSD A (L x L) -> eigenvalues w (descending) and eigenvectors in the COLUMNS of V.  A may be destroyed.
// tol_rel: off-diagonal elements are annihilated down to |a_pq| <= tol_rel sqrt(a_pp a_qq) (graded matrices keep the
// relative accuracy of their small eigenvalues) or to the 1e-16 ||diag|| rounding floor; 1e-15 for fp64 data, 1e-8 is
// ample when the matrix was formed from fp32 data.
// clustered = true: the caller expects eigenvalues closer than the two-stage solver's gap tolerance (the Ritz values of a
// subspace iteration: a block of noise-level eigenvalues) -- go to the Jacobi solver directly instead of paying for both
// Lz > L: V is an Lz x Lz matrix (leading dimension ldv) whose rows / columns L .. Lz - 1 must come out zero (padding for the
// GEMM kernels); the solver's first kernel writes those zeros itself, which saves the caller a memset launch.
// ncheck > 0: only the leading ncheck eigenpairs are delivered to full accuracy -- the two-stage solver's closeness verdict
// looks at those alone (a cluster further down, e.g. the noise-level Ritz values of a subspace iteration, gets vectors that
// are accurate to eps ||A|| / gap only and need not be mutually orthogonal; it no longer sends the whole problem to Jacobi).
// verdict (nullable, device int): instead of running the Jacobi fallback behind a flagged two-stage solve, OR a non-zero value into
// *verdict and deliver the two-stage result as it is (orders the register-resident kernels take; other orders ignore it).
void op_eigh(Dev*, double* A, int64_t L, int64_t lda, double* V, int64_t ldv, double* w, double tol_rel = 1e-15,
             bool clustered = false, int64_t Lz = 0, int64_t ncheck = 0, int* verdict = nullptr, bool verdict_fresh = false,
             double gap_tol_override = 0.0);
// (gap_tol_override > 0: the closeness threshold of the two-stage solver, relative to ||A||, instead of the one tol_rel implies --
// a caller whose matrix came from fp32 data can accept eigenvectors to eps / 1e-8 while still asking the fallback for 1e-15)
// (verdict_fresh: *verdict is cleared first instead of accumulated into; orders that ignore the verdict leave it 0)
// Residual verdict of a Rayleigh-Ritz step without a host round trip: with the Ritz vectors Vr (rows x >= nc, ldv), their
// images CV = C Vr (ldc) and the Ritz values theta,  out[0] = max_{j < nc} ||CV[:, j] - theta_j Vr[:, j]||_2^2,  out[1] = theta[0],
// out[2] = 1 if one of those nc norms is not finite or *flag / *flag2 (nullable device ints: op_eigh's verdict; the pivots the
// orthonormalisations of the iteration dropped) is non-zero.  One launch, no clear needed; the caller reads `out` with its other results.
void op_ritz_residual(Dev*, const double* CV, int64_t ldc, const double* Vr, int64_t ldv, int64_t rows, int64_t nc, const double* theta,
                      const int* flag, double* out3, double* w_out = nullptr, const int* flag2 = nullptr);   // w_out (nullable): receives theta[0 .. nc)
// FastICA whitening matrix from the eigenpairs of the covariance (ica.rs:190-208) in one launch:
//   KT[i][j]  = s_j U[i][j] / sigma_j,  sigma_j = sqrt(max(lam_j, 0)) (0 where sigma_j == 0),  j < nc;  0 for nc <= j < ncp
//   s_j = +-1 normalises the eigenvector's sign: its first component of largest magnitude becomes positive (an eigen-solver's
//   own sign is arbitrary and may flip under a last-bit perturbation; the whitening rows must not)
//   KTs[i][j] = KT[i][j] * scale          (scale = sqrt(n): X1 = K X sqrt(n))
// U: rows x >= nc (ldu), KT / KTs: rows x ncp.
void op_whiten_k(Dev*, const double* U, int64_t ldu, const double* lam, int64_t rows, int64_t nc, int64_t ncp, double scale,
                 double* KT, double* KTs);
// Singular value decomposition of a small square matrix A (L x L fp64, lda) by ONE-SIDED (Hestenes) Jacobi on its rows --
// the route that keeps the relative accuracy of small singular values (eps kappa of the row-scaled matrix, not eps kappa^2
// like an eigen-decomposition of A A^T): rows p, q are rotated until all are mutually orthogonal, G A = W, so A = G^T W and
// the rows of G are the LEFT singular vectors, the row norms of W the singular values.
//   U (L x L, ldu): column j = the left singular vector of the j-th singular value in ASCENDING order,
//   s_inv (L):      1 / (that singular value)  (descending; 0 where the singular value is 0).
// (This is the order its one caller wants: A = R^-1 of a thin QR X = Q R, whose left singular vectors are the right
// singular vectors of X and whose inverse singular values are X's, largest first.)  A is destroyed.
// nonconv (nullable, device int): set to 1 when 40 sweeps did not orthogonalise the rows (never cleared here).
void op_jacobi_svd_rows(Dev*, double* A, int64_t L, int64_t lda, double* U, int64_t ldu, double* s_inv, int* nonconv = nullptr);
// Every column j < cols of Y (rows x cols, ldy) that is exactly zero is replaced by column j of Src (lds): the robust re-basing
// of RandomizedPca refills the directions its dependence test dropped with fresh (random) ones instead of shrinking the block.
void op_refill_zero_cols(Dev*, double* Y, int64_t rows, int64_t cols, int64_t ldy, const double* Src, int64_t lds);
// x[i] *= alpha
void op_dscal(Dev*, double* x, int64_t count, double alpha);
// y[i] += alpha * x[i]
void op_daxpy(Dev*, int64_t count, double alpha, const double* x, double* y);
// elementwise helpers on f64 vectors:  mode 0: y = sqrt(max(x,0)); mode 1: y = x > thr*x[0] ? 1/x : 0; mode 2: y = x^2
void op_dvec(Dev*, int mode, const double* x, double* y, int64_t count, double thr);
// A[i][j] *= s[j]  (f64 matrix M x N)
void op_dscale_cols(Dev*, double* A, int64_t M, int64_t N, int64_t lda, const double* s);
void op_cvt_from_f64(Dev*, int dtype, void* dst, const double* src, int64_t count);
void op_cvt_to_f64(Dev*, int dtype, double* dst, const void* src, int64_t count);
// dst (rows_p x cols_p fp64, zero padded) <- the leading rows x cols block of the row-major device matrix src (ld lds)
// (zero_ptr, zero_count: a small fp64 range the same launch clears -- the caller's accumulators -- instead of a memset of its own)
void op_pad_to_f64(Dev*, int dtype, double* dst, int64_t rows_p, int64_t cols_p, const void* src, int64_t rows, int64_t cols, int64_t lds,
                   double* zero_ptr = nullptr, int64_t zero_count = 0);

}  // namespace petal
