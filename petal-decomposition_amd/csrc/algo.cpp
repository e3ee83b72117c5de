// algo.cpp -- host side of the hot path: the reference's fit()/transform() algorithms restated as
// a sequence of device operations (ops.h).  Nothing here touches matrix elements of the O(n)
// operands on the host; the host only sequences kernels, moves O(d l) results and talks to the
// collective hook.
//
// Reference call stacks: SURVEY.md section 3; algorithm: SURVEY.md section 10.
#include <algorithm>
#include <chrono>
#include <cstdint>

#include "ctx.h"

namespace petal {

namespace {

double get_elem(const void* p, int dt, int64_t i) {
    return dt == F64 ? static_cast<const double*>(p)[i] : static_cast<double>(static_cast<const float*>(p)[i]);
}
void put_elem(void* p, int dt, int64_t i, double v) {
    if (dt == F64) static_cast<double*>(p)[i] = v;
    else static_cast<float*>(p)[i] = static_cast<float>(v);
}

void check_matrix(const petal_matrix& m, const char* what) {
    if (m.dtype != PETAL_F32 && m.dtype != PETAL_F64) invalid_input(std::string(what) + ": unsupported dtype");
    if (m.rows < 0 || m.cols < 0) invalid_input(std::string(what) + ": negative shape");
    if (m.space != PETAL_HOST && m.space != PETAL_DEVICE) invalid_input(std::string(what) + ": bad memory space");
    if (m.rows > 0 && m.cols > 0 && m.data == nullptr) invalid_input(std::string(what) + ": null data");
    // linalg.rs:52-53, 76-79: LAPACK dims are i32 in the reference; this path is 64-bit throughout but
    // keeps the n*d element count addressable.
    if (m.rows > (int64_t(1) << 40) || m.cols > (int64_t(1) << 24)) invalid_input(std::string(what) + ": too many rows/columns");
}

struct RankInfo { double n_total = 0; int64_t row_offset = 0; };

// Everything a sample-sharded fit needs from the other ranks before its first product, in ONE all-reduce (round 2 issued
// three: row counts, column sums, rank 0's random draw):
//     buf = [ row counts (world) | column sums (w_sums: 0, dp or 2 dp with the sums of squares) | draw (ndraw) ]
//  - counts: every rank writes its own row count into its slot, zeros elsewhere: the SUM is the table of counts;
//  - draw: the replicated random input (Omega, pca.rs:701-705; w_init, ica.rs:210-214).  In the crate ONE model-owned
//    generator draws it; with one process per GPU every rank has its own generator, so rank 0's draw is the one every rank
//    uses: the other ranks contribute zeros and the SUM hands them rank 0's values (x + 0 + ... + 0 = x).
// The one host synchronisation of a sharded fit's start happens here (the counts decide shapes and error returns).
struct ShardPrologue {
    RankInfo ri;
    DBuf buf;
    double* sums = nullptr;  // device, w_sums doubles (all ranks' column sums [| sums of squares])
    double* draw = nullptr;  // device, ndraw doubles (rank 0's draw widened to fp64)
};
ShardPrologue shard_prologue(petal_ctx& c, const DevMat& X, int64_t w_sums, bool with_sq, int dt_draw, const void* host_draw, int64_t ndraw) {
    ShardPrologue p;
    const int64_t total = int64_t(c.world) + w_sums + ndraw;
    p.buf = DBuf(c.dev, sizeof(double) * size_t(total));
    double* base = p.buf.f64();
    p.sums = base + c.world;
    p.draw = p.sums + w_sums;
    std::vector<double> h(c.world, 0.0);
    h[c.rank] = double(X.n);
    dev_h2d_async(c.dev, base, h.data(), sizeof(double) * c.world);
    if (w_sums) op_colsum(c.dev, X.dtype, X.p, X.n, X.dp, X.ld, p.sums, with_sq);
    if (ndraw) {
        if (c.rank == 0) {
            DBuf raw(c.dev, dtype_size(dt_draw) * size_t(ndraw));
            dev_h2d_async(c.dev, raw.p, host_draw, raw.bytes);
            op_cvt_to_f64(c.dev, dt_draw, p.draw, raw.p, ndraw);
        } else {
            dev_memset(c.dev, p.draw, 0, sizeof(double) * size_t(ndraw));
        }
    }
    allreduce_f64(c, base, total, PETAL_SUM);
    dev_d2h(c.dev, h.data(), base, sizeof(double) * c.world);
    dev_sync(c.dev);
    for (int i = 0; i < c.world; ++i) { p.ri.n_total += h[i]; if (i < c.rank) p.ri.row_offset += int64_t(h[i]); }
    return p;
}

// column means (pca.rs:520-528 / ica.rs:174): mu64 (device f64[dp]) and muT (device dtype[dp]); zeros if !centering
// with_sq (centering only): mu64 has 2 dp entries, the second half holds the column sums of squares over all ranks,
// from the same pass over X (total variance = sum_j (sq_j - n mu_j^2), formed in fp64 by the caller).
// reduced_sums (sharded fits): the all-reduced [column sums | sums of squares] of shard_prologue(); single-rank fits make
// their own pass over X here.
void column_means_into(petal_ctx& c, const DevMat& X, double n_total, bool centering, double* mu64, DBuf& muT, bool with_sq,
                       const double* reduced_sums = nullptr) {
    const int64_t w = (with_sq && centering) ? 2 * X.dp : X.dp;
    muT = DBuf(c.dev, dtype_size(X.dtype) * X.dp);
    if (!centering) {
        dev_memset(c.dev, mu64, 0, sizeof(double) * w);
        dev_memset(c.dev, muT.p, 0, muT.bytes);
        return;
    }
    if (!reduced_sums) {  // one pass + one finishing launch (sum of the block partials, 1 / n, the dtype copy)
        op_colmean(c.dev, X.dtype, X.p, X.n, X.dp, X.ld, n_total, mu64, muT.p, w > X.dp);
        return;
    }
    dev_d2d(c.dev, mu64, reduced_sums, sizeof(double) * w);
    op_dscal(c.dev, mu64, X.dp, 1.0 / n_total);
    op_cvt_from_f64(c.dev, X.dtype, muT.p, mu64, X.dp);
}
void column_means(petal_ctx& c, const DevMat& X, double n_total, bool centering, DBuf& mu64, DBuf& muT, bool with_sq = false,
                  const double* reduced_sums = nullptr) {
    mu64 = DBuf(c.dev, sizeof(double) * ((with_sq && centering) ? 2 * X.dp : X.dp));
    column_means_into(c, X, n_total, centering, mu64.f64(), muT, with_sq, reduced_sums);
}

// svd_flip's decision (pca.rs:826-839) for the columns of a row-sharded U: sign of the first
// element of maximal magnitude over ALL ranks' rows.  Returns +1/-1 per column.
std::vector<double> signs_from_triple(const std::vector<double>& h, int64_t L) {
    std::vector<double> sg(L, 1.0);
    if (int64_t(h.size()) == L) {  // sharded fp32 path: the winning packed keys, sign in bit 0
        for (int64_t j = 0; j < L; ++j) {
            uint64_t bits = 0;
            std::memcpy(&bits, &h[j], 8);
            sg[j] = (bits & 1) ? -1.0 : 1.0;
        }
        return sg;
    }
    for (int64_t j = 0; j < L; ++j) sg[j] = h[2 * L + j] < 0 ? -1.0 : 1.0;
    return sg;
}
// `deferred` given (single rank, or sharded fp32): the decision data is only QUEUED for the host (no sync here); the
// caller decodes it with signs_from_triple() after its own dev_sync, so a fit ends with one synchronisation.
// `slot` (device, 4 L doubles): the decision data is only WRITTEN there -- [absmax | row | sign] in the first 3 L, the
// all-reduced keys of the sharded fp32 form in the last L -- for a caller that ships it to the host with its other results
// (flip_slot_keys() says which part to decode with signs_from_triple()).  Not for sharded fp64 (three dependent rounds).
bool flip_slot_keys(const petal_ctx& c, int dtype) { return sharded(c) && dtype == F32; }
// scanned = true: the product kernel already left the (max, row, sign) triple in the slot (op_gemm_xp_prod_absmax)
// `verdict` (sharded fp32 only; device int): slot[4 L] = (*verdict != 0) rides the same MAX all-reduce, so every rank reads the
// AGREED redo decision (a rank that branched on its own copy of a replicated flag could leave the others waiting in a collective).
void flip_signs_to_slot(petal_ctx& c, int dtype, const void* U, int64_t n, int64_t L, int64_t ldu, int64_t row_offset, double* slot,
                        bool scanned = false, const int* verdict = nullptr) {
    if (L == 0 && !(verdict && flip_slot_keys(c, dtype))) return;
    if (!scanned && L > 0) op_col_absmax(c.dev, dtype, U, n, L, ldu, row_offset, slot, slot + L, slot + 2 * L);
    if (flip_slot_keys(c, dtype)) {
        op_flip_key(c.dev, slot, slot + 3 * L, L, verdict);
        allreduce_f64(c, slot + 3 * L, L + (verdict ? 1 : 0), PETAL_MAX);
    }
}
// A yes/no decision that changes which collectives follow (redo of a fit, the accurate route) must be the SAME on every rank.
// The flags behind these decisions come from replicated kernels on all-reduced inputs and are expected to be bit-identical, but
// "expected" is not a protocol: a one-rank disagreement would desynchronise the all-reduce sequence and hang the job.  Sharded
// fits therefore branch on the MAX over the ranks (one 8-byte all-reduce + round trip; only on paths that already synchronise).
double agree_max(petal_ctx& c, double local) {
    if (!sharded(c)) return local;
    DBuf w(c.dev, sizeof(double));
    double v = local;
    dev_h2d(c.dev, w.p, &v, sizeof(double));
    allreduce_f64(c, w.f64(), 1, PETAL_MAX);
    dev_d2h(c.dev, &v, w.p, sizeof(double));
    dev_sync(c.dev);
    return v;
}
bool agree_any(petal_ctx& c, bool local) { return agree_max(c, local ? 1.0 : 0.0) != 0.0; }
std::vector<double> flip_signs(petal_ctx& c, int dtype, const void* U, int64_t n, int64_t L, int64_t ldu,
                               int64_t row_offset, std::vector<double>* deferred = nullptr) {
    std::vector<double> h(3 * L), sg(L, 1.0);
    if (L == 0) { if (deferred) deferred->assign(3 * L, 0.0); return sg; }
    DBuf r(c.dev, sizeof(double) * 3 * L);
    op_col_absmax(c.dev, dtype, U, n, L, ldu, row_offset, r.f64(), r.f64() + L, r.f64() + 2 * L);
    if (!sharded(c)) {
        if (deferred) {
            deferred->assign(3 * L, 0.0);
            dev_d2h(c.dev, deferred->data(), r.p, r.bytes);
            return sg;
        }
        dev_d2h(c.dev, h.data(), r.p, r.bytes);
        dev_sync(c.dev);
        return signs_from_triple(h, L);
    }
    DBuf g(c.dev, sizeof(double) * L);
    std::vector<double> gm(L), cand(L), win(L);
    if (dtype == F32) {
        // ONE all-reduce: |u| of an fp32 value leaves the low 29 mantissa bits of its fp64 image free, so
        // key = |u| with (2^28 - 1 - row, sign) packed into those bits orders by |u| first, then by LOWEST row:
        // MAX over the ranks picks exactly the element svd_flip would (pca.rs:826-839).  Rows < 2^28.  The key is
        // packed on the device (op_flip_key), so nothing here waits for the host.
        op_flip_key(c.dev, r.f64(), g.f64(), L);
        allreduce_f64(c, g.f64(), L, PETAL_MAX);
        if (deferred) {
            deferred->assign(L, 0.0);
            dev_d2h(c.dev, deferred->data(), g.p, g.bytes);
            return sg;
        }
        dev_d2h(c.dev, win.data(), g.p, g.bytes);
        dev_sync(c.dev);
        return signs_from_triple(win, L);
    }
    dev_d2d(c.dev, g.p, r.p, g.bytes);
    allreduce_f64(c, g.f64(), L, PETAL_MAX);
    dev_d2h(c.dev, h.data(), r.p, r.bytes);
    dev_d2h(c.dev, gm.data(), g.p, g.bytes);
    dev_sync(c.dev);
    const double inf = std::numeric_limits<double>::infinity();
    for (int64_t j = 0; j < L; ++j) cand[j] = (h[j] == gm[j]) ? h[L + j] : inf;
    dev_h2d(c.dev, g.p, cand.data(), g.bytes);
    allreduce_f64(c, g.f64(), L, PETAL_MIN);
    dev_d2h(c.dev, win.data(), g.p, g.bytes);
    dev_sync(c.dev);
    for (int64_t j = 0; j < L; ++j) cand[j] = (cand[j] == win[j] && std::isfinite(win[j])) ? h[2 * L + j] : 0.0;
    dev_h2d(c.dev, g.p, cand.data(), g.bytes);
    allreduce_f64(c, g.f64(), L, PETAL_SUM);
    dev_d2h(c.dev, win.data(), g.p, g.bytes);
    dev_sync(c.dev);
    for (int64_t j = 0; j < L; ++j) sg[j] = win[j] < 0 ? -1.0 : 1.0;
    return sg;
}

// Y (M x LP, f64) <- orthonormal basis of range(Y) by Cholesky-QR in fp64 (stands where the reference re-bases
// the small d x l iterate with pivoted LU, pca.rs:712-713).  One round leaves ||Y^T Y - I|| ~ 1e-16 cond(Y)^2,
// far below the fp32 rounding the basis undergoes when it is packed for the MFMA kernel.
void orthonormalize_small(petal_ctx& c, DBuf& Y, int64_t M, int64_t L, int64_t LP, double tol) {
    DBuf G(c.dev, sizeof(double) * LP * LP), T(c.dev, sizeof(double) * LP * LP), Y2(c.dev, Y.bytes);
    op_dgemm(c.dev, true, false, LP, LP, M, 1.0, Y.f64(), LP, Y.f64(), LP, 0.0, G.f64(), LP);
    op_chol_inv(c.dev, G.f64(), L, LP, T.f64(), LP, tol, nullptr, LP);
    op_dgemm(c.dev, false, false, M, LP, LP, 1.0, Y.f64(), LP, T.f64(), LP, 0.0, Y2.f64(), LP);
    std::swap(Y, Y2);
}

// Top-nc eigenpairs of the symmetric PSD matrix C (dp x dp fp64, device) by block subspace iteration with
// Rayleigh-Ritz, all in fp64 on the small-matrix kernels.  The one-workgroup Jacobi solver needs ~150 ms at d = 256
// (its matrices do not fit LDS); when only nc << d pairs are wanted (FastICA with n_components, Pca with k < d) a
// block of p = nc + 16 vectors converges in a few products.  Convergence: ||C v - w v|| <= 1e-12 w_0 for every wanted pair.
//
// Two ways to run it:
//  * resid3 == nullptr (synchronous): products and Rayleigh-Ritz steps until the residuals pass, a host round trip per
//    Rayleigh-Ritz step; returns false when the shape does not qualify or 40 products did not converge (the caller then
//    runs the full solver).
//  * resid3 != nullptr (optimistic): exactly two products and ONE Rayleigh-Ritz step are enqueued -- what the synchronous
//    loop needs on every spectrum with a gap behind the wanted pairs -- V / w are written and the residual verdict goes to
//    resid3 (device, 3 doubles: op_ritz_residual) for the caller to read with its own results: no host synchronisation
//    here.  topk_verdict_ok() decodes it; a caller that finds it failed redoes its fit with the synchronous form.
// The Ritz problem (order p) goes to the two-stage eigen-solver with the closeness verdict restricted to the nc wanted
// pairs: the unconverged tail of the block is a cluster of noise-level Ritz values whose vectors only have to span it.
bool topk_applies(int64_t d, int64_t dp, int64_t nc) {
    const int64_t p = std::min<int64_t>(round_up(nc + 16, 16), dp);
    return !(nc <= 0 || p >= d || d <= 88 || p > 512);  // (beyond that the Rayleigh-Ritz solves cost more than they save)
}
// tol: the accepted residual ||C v - theta v|| / theta_0.  The eigenvector of a SMALL wanted eigenvalue inherits residual / gap:
// at sigma_k = 1e-3 sigma_1 with 15 % gaps, 1e-12 lets that vector be off by 4e-6 (measured 1.4e-7 on fp64 data, where the parity
// bar is 1e-9) -- fp64 data therefore iterate down to 3e-14, a few times the rounding floor of the product C Q; fp32 data,
// whose vectors are pinned to ~1e-6 at best, keep 1e-12.
bool topk_verdict_ok(const double* r3, double tol = 1e-12) {
    return r3[2] == 0.0 && std::isfinite(r3[0]) && r3[1] > 0 && std::sqrt(std::max(r3[0], 0.0)) <= tol * r3[1];
}
// gap_tol: the two-stage solver's closeness threshold for the Ritz problem (relative to ||H||; pairs closer than it go to the
// Jacobi fallback, which always works to 1e-15) -- 1e-8 when C came from fp32 data (eigenvectors to eps / 1e-8 ~ 1e-8 are
// beyond what the data pins), 1e-5 for fp64 data (2e-11).  With the fp64 threshold on fp32 data every spectrum graded over
// more than five decades counted as "clustered" at its small end and each Rayleigh-Ritz step ran the full Jacobi solve (170 +
// 40 us per step at p = 48, two steps per tall exact-Pca fit).
bool topk_eigh(petal_ctx& c, const double* C, int64_t d, int64_t dp, int64_t nc, double* V, double* w, double gap_tol,
               double* resid3 = nullptr, double verdict_tol = 1e-12) {
    if (!topk_applies(d, dp, nc)) return false;
    const int64_t p = std::min<int64_t>(round_up(nc + 16, 16), dp);
    Dev* dv = c.dev;
    DBuf Q(dv, sizeof(double) * dp * p), Y(dv, sizeof(double) * dp * p), G(dv, sizeof(double) * p * p), T(dv, sizeof(double) * p * p);
    DBuf H(dv, sizeof(double) * p * p), S(dv, sizeof(double) * p * p), th(dv, sizeof(double) * p), R(dv, sizeof(double) * dp * p);
    DBuf r3(dv, sizeof(double) * 3), vf(dv, 64);
    // vf: [0] the Ritz problem's closeness verdict, [1] the pivots the orthonormalisations below DROPPED.  A dropped column is a lost
    // direction: from the random start every column of C Q carries eigen-direction j at lambda_j / lambda_1 of its length, the Gram matrix
    // of the block squares that, and beyond lambda_1 / lambda_j ~ 1e7 the Cholesky pivot of direction j falls under the 1e-14 rule -- its
    // Ritz value came back an exact ZERO with a zero residual, i.e. 'converged' (exact Pca of uncentred data 40 sigma off centre, 50 x 256,
    // both data types: the last four of 22 singular values 0 instead of 1.2 ... 0.9; dev/fuzz_round6.py, round 6).  Any dropped pivot
    // now fails the verdict and the caller's full eigen-solve takes over.
    int* const ndrop = vf.as<int>() + 1;
    dev_memset(dv, vf.p, 0, 64);
    if (!c.topk_seed || c.topk_seed_d != d || c.topk_seed_dp != dp || c.topk_seed_p != p) {   // (once per shape and ctx)
        std::vector<double> h(size_t(dp) * p, 0.0);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        for (int64_t i = 0; i < d; ++i)
            for (int64_t j = 0; j < p; ++j) {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                h[size_t(i) * p + j] = double(int64_t(st >> 11)) / double(1ll << 52) - 1.0;
            }
        if (c.topk_seed) dev_free(dv, c.topk_seed);
        c.topk_seed = static_cast<double*>(dev_alloc(dv, sizeof(double) * dp * p));
        c.topk_seed_d = d; c.topk_seed_dp = dp; c.topk_seed_p = p;
        dev_h2d(dv, c.topk_seed, h.data(), sizeof(double) * dp * p);
    }
    const double* seed = c.topk_seed;
    // dst = orthonormal basis of range(src): two Cholesky-QR rounds (one leaves ||Q^T Q - I|| ~ eps cond(src)^2, up to 1e-4
    // after a product with C: good enough for the basis the NEXT product is applied to, not for a Rayleigh-Ritz step)
    auto orth = [&](DBuf& src, DBuf& dst, int rounds = 2) {
        const double* in = src.f64();
        for (int rep = 2 - rounds; rep < 2; ++rep) {
            op_dgemm(dv, true, false, p, p, dp, 1.0, in, p, in, p, 0.0, G.f64(), p);
            double* out = rep == 0 ? R.f64() : dst.f64();
            // (round 6: the register-resident factorisation + a triangular solve where the device has them -- p = 48 / 80 at the
            // configs: 3 x 21 -> 3 x 9 us of a configs[2] fit; the explicit inverse otherwise)
            if (op_chol_rt(dv, F64, dp, G.f64(), p, p, T.f64(), p, 1e-14, ndrop, p, 0)) op_trsm_right(dv, in, dp, p, T.f64(), p, p, out, p);
            else op_dgemm(dv, false, false, dp, p, p, 1.0, in, p, T.f64(), p, 0.0, out, p);
            in = out;
        }
    };
    // The nc wanted Ritz pairs written straight into the caller's V (ld = dp) and w, their images R = C (Q S) and the verdict: the residual norms, and the eigen-solver's closeness flag for the nc wanted pairs (its Jacobi
    // fallback launches, which return at once on every separated spectrum, are not issued on the optimistic run: a flagged Ritz
    // problem fails its verdict and the caller's redo solves it with the fallback in place)
    auto rayleigh_ritz = [&](double* out3) {
        op_dgemm(dv, true, false, p, p, dp, 1.0, Q.f64(), p, Y.f64(), p, 0.0, H.f64(), p);    // Rayleigh quotient
        // (only the optimistic caller, who redoes the fit on a bad verdict, skips the fallback: here a flagged Ritz problem is solved by Jacobi as before)
        op_eigh(dv, H.f64(), p, p, S.f64(), p, th.f64(), 1e-15, false, 0, nc, resid3 ? vf.as<int>() : nullptr, true, gap_tol);
        op_dgemm(dv, false, false, dp, nc, p, 1.0, Q.f64(), p, S.f64(), p, 0.0, V, dp);        // Ritz vectors
        op_dgemm(dv, false, false, dp, nc, p, 1.0, Y.f64(), p, S.f64(), p, 0.0, R.f64(), p);   // C (Q S)
        op_ritz_residual(dv, R.f64(), p, V, dp, dp, nc, th.f64(), resid3 ? vf.as<int>() : nullptr, out3, w, ndrop);
    };
    auto deliver = [&] {};
    double h3[3], prev_rel = -1.0;
    int bad_rates = 0;   // consecutive checks whose measured rate said "cannot get there" (one transient stall must not
                         // send a spectrum that converges a few checks later to the full eigen-solve)
    for (int it = 0; it < 40; ++it) {
        if (it == 0) op_dgemm(dv, false, false, dp, p, dp, 1.0, C, dp, seed, p, 0.0, Y.f64(), p);   // (the start block is read-only)
        orth(Y, Q, it % 2 == 1 ? 2 : 1);   // (the Rayleigh-Ritz step of the odd iterations needs the orthonormal basis)
        op_dgemm(dv, false, false, dp, p, dp, 1.0, C, dp, Q.f64(), p, 0.0, Y.f64(), p);  // Y = C Q
        if (it % 2 == 1) {  // Rayleigh-Ritz + residual check every second product (never converged after the first)
            if (resid3) {
                rayleigh_ritz(resid3);
                deliver();
                return true;
            }
            rayleigh_ritz(r3.f64());
            dev_d2h(dv, h3, r3.p, sizeof(h3));
            dev_sync(dv);
            if (topk_verdict_ok(h3, verdict_tol)) { deliver(); return true; }
            if (h3[2] != 0.0) return false;   // (a non-finite residual, or an orthonormalisation dropped a pivot: the full eigen-solve)
            // No gap behind the wanted pairs (a smoothly decaying spectrum): the residual falls by (lambda_{p+1} / lambda_nc)^2 per
            // check and would need hundreds of them.  Give up as soon as the measured rate says the remaining budget cannot get
            // there -- the caller's full eigen-solve costs about eight checks at d = 512 -- instead of running all twenty
            // (20000 x 512 without a gap: 6 of the fit's 10 ms were these products).
            const double rel = (std::isfinite(h3[0]) && h3[1] > 0) ? std::sqrt(std::max(h3[0], 0.0)) / h3[1] : -1.0;
            if (rel < 0) return false;
            if (prev_rel > 0) {
                const double rate = rel / prev_rel;
                const bool hopeless = !(rate < 1.0) || std::log(verdict_tol / rel) / std::log(rate) > 8.0;
                bad_rates = hopeless ? bad_rates + 1 : 0;
                // (sharded fits: C is the all-reduced covariance and every kernel here is replicated, so all ranks take the same
                // branch, as they do for the convergence verdict above)
                if (bad_rates >= 2 || (hopeless && it >= 9)) return false;
            }
            prev_rel = rel;
        }
    }
    return false;
}

// fp64 data whose wanted singular values fall below the accuracy floor of the Gram routes (an eigen-decomposition of
// Xc^T Xc keeps sigma_k to eps (sigma_1 / sigma_k)^2; the crate's gesvd, linalg.rs:70-91, to eps sigma_1 / sigma_k): thin QR of
// Xc by Cholesky-QR2 in fp64 -- T1 = chol(Xc^T Xc)^-1, Q1 = Xc T1, T2 = chol(Q1^T Q1)^-1, R^-1 = T1 T2, two more passes
// over X -- and a one-sided Jacobi SVD of R^-1, whose left singular vectors are the right singular vectors of Xc.
// C: the (all-reduced) Gram matrix Xc^T Xc.  Fills V (leading d x d block, ld = dp: column j = right singular vector j) and
// sig (d values, descending).  Returns false, nothing written, when a Cholesky pivot broke down (kappa beyond ~1e7 or
// rank-deficient data: the caller keeps the Gram-route result) or the order is beyond the one-workgroup Jacobi kernel.
bool accurate_small_svd(petal_ctx& c, const DevMat& X, const void* muT, const double* C, int64_t d, double* V, double* sig) {
    const int64_t dp = X.dp, n = X.n;
    if (X.dtype != F64 || d > 1024 || d < 1) return false;
    Dev* dv = c.dev;
    DBuf T1(dv, sizeof(double) * dp * dp), T2(dv, sizeof(double) * dp * dp), T(dv, sizeof(double) * dp * dp), G2(dv, sizeof(double) * dp * dp);
    DBuf nd(dv, 64);
    dev_memset(dv, nd.p, 0, nd.bytes);
    int hdead = 0;
    // (every rank takes the same exits: the pivot flags are replicated, and agree_any() makes that a protocol)
    auto broke_down = [&] {
        dev_d2h(dv, &hdead, nd.p, sizeof(int));
        dev_sync(dv);
        return agree_any(c, hdead != 0);
    };
    op_chol_inv(dv, C, d, dp, T1.f64(), dp, 1e-15, nd.as<int>(), dp);
    if (broke_down()) return false;   // before the second copy of X is allocated or formed
    DBuf Q1(dv, sizeof(double) * size_t(std::max<int64_t>(n, 1)) * dp);
    op_gemm_xp(dv, F64, X.p, n, dp, X.ld, muT, T1.f64(), dp, dp, nullptr, Q1.p, dp, nullptr);
    op_gemm_atb(dv, F64, Q1.p, dp, dp, nullptr, Q1.p, dp, dp, nullptr, n, G2.f64(), dp, true);
    allreduce_f64(c, G2.f64(), dp * dp, PETAL_SUM);
    op_chol_inv(dv, G2.f64(), d, dp, T2.f64(), dp, 1e-15, nd.as<int>(), dp);
    op_dgemm(dv, false, false, dp, dp, dp, 1.0, T1.f64(), dp, T2.f64(), dp, 0.0, T.f64(), dp);
    if (broke_down()) return false;
    // (into temporaries: a Jacobi run that used up its sweeps raises the same flag, and the caller's Gram-route result stays)
    DBuf Vt(dv, sizeof(double) * dp * dp), st(dv, sizeof(double) * dp);
    dev_memset(dv, Vt.p, 0, Vt.bytes);
    op_jacobi_svd_rows(dv, T.f64(), d, dp, Vt.f64(), dp, st.f64(), nd.as<int>());
    if (broke_down()) return false;
    dev_copy2d(dv, V, sizeof(double) * dp, Vt.p, sizeof(double) * dp, sizeof(double) * d, size_t(d), 2);
    dev_d2d(dv, sig, st.p, sizeof(double) * d);
    return true;
}
// (sigma_k / sigma_1 of the Gram route's own eigenvalues below this: its sigma_k is no longer good to 1e-9)
constexpr double GRAM_ROUTE_FLOOR = 3.1622776601683794e-4;  // 10^-3.5

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

void finish_stats(petal_ctx& c, const Timer& t) {
    dev_sync(c.dev);
    c.stats.fit_ms = t.ms();
    KernelTiming kt = dev_timing(c.dev);
    c.stats.xp_ms = kt.ms[TAG_XP];
    c.stats.xp_launches = kt.launches[TAG_XP];
    c.stats.atb_ms = kt.ms[TAG_ATB];
    c.stats.atb_launches = kt.launches[TAG_ATB];
    c.stats.ica_step_ms = kt.ms[TAG_ICA];
    c.stats.ica_step_launches = kt.launches[TAG_ICA];
    c.stats.pow_ms = kt.ms[TAG_POW];
    c.stats.pow_launches = kt.launches[TAG_POW];
    c.stats.stream_ms = kt.ms[TAG_STREAM];
    c.stats.stream_launches = kt.launches[TAG_STREAM];
    c.stats.allreduce_ms = kt.ms[TAG_COMM];
    c.stats.allreduce_timed = kt.launches[TAG_COMM];
}

}  // namespace

void allreduce_f64(petal_ctx& c, double* dev_buf, int64_t count, int op) {
    if (!sharded(c) || count == 0) return;
    if (!c.allreduce) device_error("world_size > 1 but no collective hook installed (petal_ctx_set_collective)");
    void* span = dev_span_begin(c.dev, TAG_COMM);
    int rc = c.allreduce(c.allreduce_user, dev_buf, count, PETAL_F64, op, dev_stream(c.dev));
    dev_span_end(c.dev, span);
    if (rc != 0) device_error("collective all-reduce failed with code " + std::to_string(rc));
    c.stats.allreduce_calls += 1;
    c.stats.allreduce_bytes += double(sizeof(double)) * double(count);
}

// ---------------------------------------------------------------------------------------------
DevMat ingest(petal_ctx& c, const petal_matrix& x) {
    check_matrix(x, "input");
    DevMat m;
    m.dtype = x.dtype;
    m.n = x.rows;
    m.d = x.cols;
    m.dp = round_up(std::max<int64_t>(x.cols, 1), 16);
    const size_t esz = dtype_size(x.dtype);
    if (m.n == 0) { m.ld = m.dp; return m; }
    const bool unit = (x.col_stride == 1 || x.cols <= 1);
    if (x.space == PETAL_DEVICE && unit && x.cols == m.dp && x.row_stride >= x.cols &&
        (reinterpret_cast<uintptr_t>(x.data) % 16) == 0 && (x.row_stride * esz) % 16 == 0) {
        m.p = x.data;  // zero-copy: already in the layout the kernels stream
        m.ld = x.row_stride;
        m.zero_copy = true;
        c.stats.x_row_pitch_bytes = int64_t(size_t(m.ld) * esz);
        c.stats.x_zero_copy = 1;
        return m;
    }
    // Row pitch of the copy.  A pitch that is a multiple of 1 KiB (2048 B at d = 512 fp32) puts column chunk c of EVERY row on the
    // same few memory channels; the matrix is being copied anyway, so the copy lands with 128 more bytes per row (measured with
    // dev/pitch_probe.py, EXPERIMENTS.md round 3: K1 -5 %, K2 -7 % at 100000 x 512).  The padding is never read.
    m.ld = m.dp;
    if (m.n >= 4096 && (size_t(m.dp) * esz) % 1024 == 0 && dev_option(c.dev, OPT_ROW_PAD) != 0) m.ld = m.dp + int64_t(128 / esz);
    m.owned = DBuf(c.dev, esz * size_t(m.n) * m.ld);
    m.p = m.owned.p;
    c.stats.x_row_pitch_bytes = int64_t(size_t(m.ld) * esz);
    c.stats.x_zero_copy = 0;
    if (x.space == PETAL_DEVICE) {
        op_pack_strided(c.dev, x.dtype, x.data, m.n, m.d, x.row_stride, x.col_stride, m.owned.p, m.ld, m.dp);
        return m;
    }
    if (m.dp != m.d) dev_memset(c.dev, m.owned.p, 0, m.owned.bytes);
    if (unit && x.row_stride >= x.cols) {
        dev_copy2d(c.dev, m.owned.p, m.ld * esz, x.data, size_t(x.row_stride) * esz, size_t(m.d) * esz, size_t(m.n), 0);
    } else {
        // Arbitrary HOST ndarray view (a transposed / Fortran-order array, a strided slice: legal in the crate, pca.rs:509-531).
        // With non-negative strides the view lives inside ONE contiguous span of its buffer: that span is uploaded as it lies (one
        // copy at PCIe rate) and the gather into the row-major layout runs on the DEVICE (op_pack_strided, as for device inputs)
        // -- round 4 gathered element by element on one host thread, 51 M memcpy calls = 0.3 s at configs[1] against a 1.2 ms fit.
        // Views with a negative stride, or whose span is more than twice their data, keep the host gather.
        const int64_t span = (m.n - 1) * x.row_stride + (m.d - 1) * x.col_stride + 1;   // elements, first to last
        if (x.row_stride >= 0 && x.col_stride >= 0 && span <= 2 * m.n * m.d + 1024) {
            DBuf raw(c.dev, esz * size_t(span));
            dev_h2d(c.dev, raw.p, x.data, raw.bytes);
            op_pack_strided(c.dev, x.dtype, raw.p, m.n, m.d, x.row_stride, x.col_stride, m.owned.p, m.ld, m.dp);
            return m;   // (raw goes back to the stream-ordered pool behind the pack kernel)
        }
        std::vector<char> tmp(esz * size_t(m.n) * m.d);
        const char* src = static_cast<const char*>(x.data);
        for (int64_t i = 0; i < m.n; ++i)
            for (int64_t j = 0; j < m.d; ++j)
                std::memcpy(&tmp[(size_t(i) * m.d + j) * esz], src + (i * x.row_stride + j * x.col_stride) * int64_t(esz), esz);
        dev_copy2d(c.dev, m.owned.p, m.ld * esz, tmp.data(), size_t(m.d) * esz, size_t(m.d) * esz, size_t(m.n), 0);
        dev_sync(c.dev);
    }
    return m;
}

void emit(petal_ctx& c, int dtype, const void* src, int64_t n, int64_t cols, int64_t ld, const petal_matrix& out, const double* scale) {
    check_matrix(out, "output");
    if (out.dtype != dtype) invalid_input("output dtype differs from input dtype");
    if (out.rows != n || out.cols != cols) invalid_input("output has the wrong shape");
    if (n == 0 || cols == 0) return;
    const size_t esz = dtype_size(dtype);
    if (out.space == PETAL_DEVICE) {   // (scale: a per-column factor applied on the way out, device array of `cols` doubles)
        op_unpack_strided(c.dev, dtype, src, n, cols, ld, out.data, out.row_stride, out.col_stride, scale);
        return;
    }
    if (scale) op_scale_cols(c.dev, dtype, const_cast<void*>(src), n, cols, ld, scale);
    if ((out.col_stride == 1 || cols == 1) && out.row_stride >= cols) {
        dev_copy2d(c.dev, out.data, size_t(out.row_stride) * esz, src, size_t(ld) * esz, size_t(cols) * esz, size_t(n), 1);
        dev_sync(c.dev);
        return;
    }
    std::vector<char> tmp(esz * size_t(n) * cols);
    dev_copy2d(c.dev, tmp.data(), size_t(cols) * esz, src, size_t(ld) * esz, size_t(cols) * esz, size_t(n), 1);
    dev_sync(c.dev);
    char* dst = static_cast<char*>(out.data);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < cols; ++j)
            std::memcpy(dst + (i * out.row_stride + j * out.col_stride) * int64_t(esz), &tmp[(size_t(i) * cols + j) * esz], esz);
}

// ---------------------------------------------------------------------------------------------
// RandomizedPca::inner_fit (pca.rs:509-550) -> randomized_svd (pca.rs:668-686) -> range finder (689-718)
void rpca_fit(petal_ctx& c, const petal_matrix& x, int64_t k, int64_t n_oversample, int64_t n_iter, bool centering,
              const void* omega, void* components, void* means, void* singular, void* total_variance,
              const petal_matrix* y_out) {
    Timer timer;
    dev_reset_timing(c.dev);
    c.stats = petal_stats{};
    check_matrix(x, "input");
    if (k < 0 || n_oversample < 0 || n_iter < 0) invalid_input("negative parameter");
    const int dt = x.dtype;
    const int64_t d = x.cols, l_req = k + n_oversample;
    // fp32 input: the total variance comes from column sums of squares gathered in the means pass (exact products, fp64
    // sums: the cancellation in sum x^2 - n mu^2 costs (mu / sigma)^2 ulps of fp64, far below fp32 resolution); fp64 input
    // keeps the centred sum fused into the first product
    const bool tv_from_sq = centering && dt == F32;
    DevMat X;
    ShardPrologue pro;
    RankInfo ri;
    if (sharded(c)) {
        // (argument errors that every rank sees alike are raised BEFORE the collective, so no rank is left waiting in it)
        if (d > 0 && l_req > 0 && omega == nullptr) invalid_input("omega is required");
        X = ingest(c, x);
        pro = shard_prologue(c, X, centering ? (tv_from_sq ? 2 * X.dp : X.dp) : 0, tv_from_sq, dt, omega, d * l_req);
        ri = pro.ri;
    } else {
        ri.n_total = double(x.rows);
    }
    const int64_t n_total = int64_t(ri.n_total);
    if (n_total < k || d < k)  // pca.rs:513-518
        invalid_input("every dimension should be at least " + std::to_string(k));
    if (centering && n_total == 0) {  // mean_axis -> None (pca.rs:521-525): Ok, model untouched
        if (y_out && (y_out->rows != 0)) invalid_input("output has the wrong shape");
        return;
    }
    const int64_t L = std::min(l_req, std::min(n_total, d));  // the reference's min(nrows, ncols) slices, pca.rs:710/713
    if (L > 0 && omega == nullptr) invalid_input("omega is required");
    if (d == 0 || L == 0) {  // nothing to decompose: k == 0 here
        if (total_variance) put_elem(total_variance, dt, 0, 0.0);
        if (y_out) emit(c, dt, nullptr, x.rows, 0, 0, *y_out);
        return;
    }
    const int64_t LP = round_up(L, 16);
    if (!sharded(c)) X = ingest(c, x);
    const int64_t n = X.n, dp = X.dp;
    const size_t esz = dtype_size(dt);
    const double tol_drop = (dt == F32 ? 1e-6 : 1e-13);
    const double tol_tall = (dt == F32 ? 1e-12 : 1e-13);   // dependence test of the tall-side Gram matrices (fp64 products of the stored iterate)

    // Everything the host reads back sits in ONE device buffer behind the all-reduced [Yp | tv] pair, so a fit ends with one
    // device-to-host copy (five separate small copies cost 4 us each):
    //   res = [ Yp (dp LP) | tv | ndead (pivot, heavy tail) | neig (eigen-solver) | lam (LP) | mu64 (2 dp) | flip (4 LP) ]
    // and the components leave in their own k x d buffer, already in the caller's layout and type (op_components_out): 128 KB at
    // configs[1] instead of the 320 KB fp64 d x l matrix V, and nothing for the host to transpose.
    const int64_t o_tv = dp * LP, o_dead = o_tv + 1, o_eig = o_dead + 1, o_lam = o_eig + 1, o_mu = o_lam + LP, o_flip = o_mu + 2 * dp, res_len = o_flip + 4 * LP + 1;  // (+ 1: the agreed redo verdict of a sharded fp32 fit)
    // (the components sit right behind the block: both leave in one copy)
    const size_t comp_bytes = esz * size_t(k) * d;
    DBuf res(c.dev, sizeof(double) * res_len + std::max<size_t>(comp_bytes, 8));
    void* const comp_dev = res.f64() + res_len;
    double* const Yp = res.f64();
    double* const tvp = res.f64() + o_tv;
    int* const ndead = reinterpret_cast<int*>(res.f64() + o_dead);   // three verdict words in a row: [0] pivot breakdowns, [1] heavy tail,
    int* const neig = ndead + 2;                                     // [2] eigenvalues too close for the two-stage eigen-solver's vectors
    static_assert(sizeof(double) == 2 * sizeof(int), "verdict words");
    double* const lam = res.f64() + o_lam;
    double* const mu64 = res.f64() + o_mu;
    double* const flip = res.f64() + o_flip;
    DBuf muT;
    // Omega (pca.rs:701-705): d x l_req row-major host draw -> first L columns, padded to dp x LP, f64.  Single rank: the raw draw
    // is staged in the pinned ring without a host wait and widened on the device; sharded: rank 0's draw came back from the
    // prologue's all-reduce, already fp64.
    DBuf P;
    // The column-means pass is queued FIRST: the device starts on it at once, and the host's copy of Omega into the pinned ring
    // runs beside it.
    // Single-rank fp32 fits whose first product is a FUSED pass gather the means inside it (op_power_pass_means: X is read once for
    // the means, the sums of squares and the first power iteration together -- round 4 spent a pass of its own on the first two,
    // the reference three: pca.rs:520-533): the means pass is then not queued here but decided in the pipeline.
    // (From 200000 rows on: at configs[1]'s 100000 x 512 the matrix sits in the Infinity Cache, the means pass costs 37 us and what
    // replaces it -- the sample's means, the all-ones column and the squares in the first pass, the move to the true centre -- costs
    // the same; at 1e6 rows the pass is 0.33 ms of HBM time and the fold takes 5 % off the fit.)
    // (fp32 data, up to FOUR iterations: the sketch Z = Xc Omega is re-based on the tall side before its product with Xc^T -- see the
    // pipeline -- so the first product pair is not a fused pass and cannot gather the means)
    const bool rebase_sketch = dt == F32 && n_iter >= 1 && n_iter <= 4;
    // (... and, at any n_iter, the second attempt of a fit whose first one lost a pivot: see the attempt loop)
    bool rebase_now = rebase_sketch;
    const double fold_rows = dev_option(c.dev, OPT_MEANS_FOLD_ROWS);   // (PETAL_OPT_MEANS_FOLD_ROWS; negative: never)
    const bool fold_means = !sharded(c) && tv_from_sq && dev_gemm_mode(c.dev) == 0 && n_iter >= 3 && !rebase_sketch && L < LP && fold_rows >= 0 && double(n) >= fold_rows &&
                            op_power_pass_applies(c.dev, dt, X.p, n, dp, X.ld, X.p, LP) && dev_option(c.dev, OPT_TWO_PLANE) != 0;
    bool means_done = false, tv_direct = false;
    auto means_pass = [&] {
        if (!tv_from_sq) dev_memset(c.dev, mu64 + dp, 0, sizeof(double) * dp);
        dev_set_tag(c.dev, TAG_STREAM);
        column_means_into(c, X, ri.n_total, centering, mu64, muT, tv_from_sq, sharded(c) && centering ? pro.sums : nullptr);
        dev_set_tag(c.dev, TAG_NONE);
        means_done = true;
    };
    if (!fold_means) means_pass();
    else muT = DBuf(c.dev, esz * size_t(dp));   // (mu64: means and, in its second half's first slot, the variance -- both written by the pass)
    if (sharded(c)) {
        P = DBuf(c.dev, sizeof(double) * dp * LP);
        op_pad_to_f64(c.dev, F64, P.f64(), dp, LP, pro.draw, d, L, l_req, tvp, 3 + LP);
    } else {
        // (queued behind the means pass, so the host's copy into the ring runs beside it.  No transfer of its own: the widening
        // kernel reads the draw from the pinned ring over the link.  A side stream for this, joined in front of the first product,
        // measured the same; one for the components' write-out further down COST 46 us a fit -- an event recorded on the main
        // stream and the wait for the join are bubbles of their own, far longer than the kernel trace shows: EXPERIMENTS.md)
        P = DBuf(c.dev, sizeof(double) * dp * LP);
        const void* view = dev_h2d_view(c.dev, omega, esz * size_t(d) * l_req);
        if (view) {
            op_pad_to_f64(c.dev, dt, P.f64(), dp, LP, view, d, L, l_req, tvp, 3 + LP);  // (also clears tv, ndead, lam for the first pipeline run)
        } else {   // a draw beyond a ring slot
            DBuf raw(c.dev, esz * size_t(d) * l_req);
            dev_h2d_async(c.dev, raw.p, omega, raw.bytes);
            op_pad_to_f64(c.dev, dt, P.f64(), dp, LP, raw.p, d, L, l_req, tvp, 3 + LP);
        }
    }
    DBuf Z(c.dev, esz * size_t(std::max<int64_t>(n, 1)) * LP), Z1(c.dev, esz * size_t(std::max<int64_t>(n, 1)) * LP);
    c.stats.pass_flops = 2.0 * double(n) * double(d) * double(l_req);
    c.stats.pass_bytes = double(esz) * (double(n) * d + double(n) * l_req + double(d) * l_req);

    DBuf Gb(c.dev, sizeof(double) * LP * LP);
    double* const G = Gb.f64();
    DBuf T(c.dev, sizeof(double) * LP * LP), Y(c.dev, sizeof(double) * dp * LP);
    DBuf Bt(c.dev, sizeof(double) * dp * LP), S(c.dev, sizeof(double) * LP * LP), Uh(c.dev, sizeof(double) * LP * LP);
    DBuf M2(c.dev, sizeof(double) * LP * LP);
    void* Uout = nullptr;  // where the pipeline left U (n x LP; its first kp columns)
    const bool slot_flip = !sharded(c) || dt == F32;  // (sharded fp64 decides the signs in three dependent all-reduce rounds)
    const int64_t kp = std::min(LP, round_up(std::max<int64_t>(k, 1), 16));
    // The whole device pipeline.  It runs OPTIMISTICALLY first (robust = false): every power iteration re-bases with
    // the single-Cholesky fast path and no host round trip; the kernels record the worst pivot breakdown in `ndead`,
    // which is read together with the results.  Only if a breakdown happened is the fit redone with robust = true.
    // Two operands of the split-product K1 may be ROUNDED to 16 significant bits (two bf16 planes: five piece products instead
    // of six) -- the sketch matrix when power iterations follow, and the re-based iterates.  That is a change of basis plus a
    // perturbation of the spanned subspace: harmless when the spectrum falls off behind the block (what the rounding injects from
    // beyond the block is damped by every later product), NOT on slowly decaying spectra, where it costs parity with a run from
    // the un-rounded Omega 10 - 80x (measured, EXPERIMENTS.md round 5).  So the first run uses them OPTIMISTICALLY (exact = false)
    // and a verdict formed from the spectrum the fit itself found (op_tail_verdict) sends a heavy-tailed fit back through the
    // pipeline with three planes everywhere (exact = true).  Nothing is remembered from fit to fit: the same input gives the same
    // bits whatever the ctx ran before (a caller who knows the data selects PETAL_GEMM_SPLIT_BF16X3_EXACT and skips the first run).
    const bool two_plane_applies = dt == F32 && dev_gemm_mode(c.dev) == 0 && n_iter > 0;
    const double p2_thr = dev_option(c.dev, OPT_VERDICT_THRESHOLD);
    // (what the small stage below needs from the passes: hoisted so that it can run again by itself)
    const void* Usrc = nullptr;  // the n x LP matrix U is formed from (U = Usrc . M2) ...
    void* Ubuf = nullptr;        // ... and the buffer it goes to
    bool t_rt = false;           // T holds R in RT form (op_chol_rt), not the explicit inverse
    bool steered = false;        // fused / steering passes ran: the heavy-tail verdict prices their rounding too
    int pipeline_runs = 0;       // (every run after the first clears the verdict words itself -- the re-based retry of an OPTIMISTIC run too)
    auto pipeline = [&](bool robust, bool exact) {
    const int planes = (n_iter > 0 && !robust && !exact) ? 2 : 3;
    if (pipeline_runs++ > 0) dev_memset(c.dev, tvp, 0, sizeof(double) * (3 + LP));  // tv, ndead, neig, lam (only lam[0 .. L) is written below); the first run's were cleared with Omega
    // The FUSED power-iteration pass Y' = Xc^T (Xc P) (one pass over X where K1 + K2 make two; it needs P on two planes, so it
    // belongs to the optimistic run): every product pair of the loop below, the last one also storing Z.
    const bool use_pow = planes == 2 && op_power_pass_applies(c.dev, dt, X.p, n, dp, X.ld, muT.p, LP);
    // STEERING products: every product of the optimistic run whose result only feeds the next re-basing may round its
    // large operands to two bf16 planes like the iterate (the fused passes do; K1 / K2 in their forms for more than 80 columns do)
    const bool steer = planes == 2 && dt == F32 && n_iter >= 3;
    bool have_yp = false;   // Yp already holds Xc^T Z for the current basis
    if (!means_done) {      // (fold_means: the first run of the pipeline)
        if (use_pow) {
            dev_set_tag(c.dev, TAG_POW);
            // mu64 = [means | tv, scratch, 0 ...]: the total variance lands where the sums of squares would have
            have_yp = op_power_pass_means(c.dev, dt, X.p, n, dp, d, X.ld, ri.n_total, P.f64(), LP, LP, L, Yp, LP, mu64, muT.p, mu64 + dp + 1, mu64 + dp);
            dev_set_tag(c.dev, TAG_NONE);
        }
        if (have_yp) { means_done = true; tv_direct = true; }
        else means_pass();
    }
    if (!have_yp && use_pow && n_iter >= 3 && !rebase_now && tv_from_sq) {
        dev_set_tag(c.dev, TAG_POW);
        have_yp = op_power_pass(c.dev, dt, X.p, n, dp, X.ld, muT.p, P.f64(), LP, LP, nullptr, LP, Yp, LP, /*steering=*/true);   // pca.rs:707 + 711
        dev_set_tag(c.dev, TAG_NONE);
        if (have_yp) allreduce_f64(c, Yp, dp * LP, PETAL_SUM);
    }
    if (!have_yp) {
    // Z = Xc . Omega (pca.rs:707); total_variance = sum Xc^2 (pca.rs:533) is fused into this product unless tv_from_sq
    dev_set_tag(c.dev, TAG_XP);
    // (with power iterations behind it the sketch matrix may be ANY matrix: the optimistic run lets the kernel round Omega to two
    // bf16 planes -- five piece products; n_iter = 0 and the robust redo keep Omega as given)
    op_gemm_xp(c.dev, dt, X.p, n, dp, X.ld, muT.p, P.f64(), LP, LP, nullptr, Z.p, LP, tv_from_sq ? nullptr : tvp, planes, steer);
    dev_set_tag(c.dev, TAG_NONE);

    // A SHORT iteration (n_iter 1 - 4) on fp32 data re-bases the sketch Z = Xc Omega on the tall side first, as the crate does
    // with its first LU (pca.rs:709).  Every column of Z is dominated by sigma_1, so the un-rebased double product Xc^T (Xc Omega)
    // carries direction j at (sigma_j / sigma_1)^2 of a column -- 1e-6 on the planted spectra -- under an fp32 accumulation that
    // is good to 6e-8 of it: the block's weakest directions come back 6 % junk, and with only one or two products behind it the
    // fit is off by 2e-4 where the crate's own f32 path holds 1e-5 (measured, EXPERIMENTS.md round 5).  With Z = Q R first, a
    // column of Xc^T Q carries direction j at sigma_j / sigma_1.  Later products wash the junk out at the rate of the spectrum's
    // gaps: at n_iter = 3 it was gone for k = 64 of 512 features (3e-6) but NOT for 87 ... 126 components (1.5e-4 ... 3.6e-4
    // where the oracle run in float32 holds 1.4e-5 ... 8e-5: dev/fuzz_round6.py, round 6), so three iterations re-base too --
    // only their first product pair: the later ones are fused / steering passes as for any n_iter >= 3.  FOUR iterations likewise: a later
    // seed of the same sweep had 60000 x 1024, k = 93, n_iter = 4 at 1.07e-4 against the float32 oracle's 1.0e-5.  From five iterations
    // on the two extra passes over Z are not spent (no miss of this kind in ~220 random cases at n_iter 5 / 7 with up to 127 components, 425 in all).
    }
    const void* Zfirst = Z.p;   // what the first product with Xc^T reads
    if (rebase_now && !robust) {
        op_gemm_atb(c.dev, dt, Z.p, LP, LP, nullptr, Z.p, LP, LP, nullptr, n, G, LP, /*precise=*/true);
        allreduce_f64(c, G, LP * LP, PETAL_SUM);
        op_chol_inv(c.dev, G, L, LP, T.f64(), LP, tol_tall, nullptr, LP);
        op_gemm_xp(c.dev, dt, Z.p, n, LP, LP, nullptr, T.f64(), LP, LP, nullptr, Z1.p, LP, nullptr);
        Zfirst = Z1.p;
    }
    double* Pcur = P.f64();  // the orthonormal basis the current Z was formed with
    for (int64_t it = 0; it < n_iter; ++it) {  // pca.rs:708-715
        if (!have_yp) {
            dev_set_tag(c.dev, TAG_ATB);
            op_gemm_atb(c.dev, dt, X.p, X.ld, dp, muT.p, it == 0 ? Zfirst : Z.p, LP, LP, nullptr, n, Yp, LP, false, steer);  // Yp = Xc^T Z (pca.rs:711)
            dev_set_tag(c.dev, TAG_NONE);
            allreduce_f64(c, Yp, dp * LP, PETAL_SUM);
        }
        have_yp = false;
        // Re-base the d x l iterate (stands for the two pivoted-LU re-basings of pca.rs:709-713).  The next product only
        // needs SOME well-conditioned basis of range(Yp): P = Yp T spans range(Yp) exactly for any invertible
        // triangular T, so the accuracy of T only decides how well-conditioned P is, never which subspace it spans (the
        // final thin QR restores orthonormality).
        if (!robust) {
            // fast path: one fp64 Cholesky of Yp^T Yp, valid while every pivot stays positive (cond(Yp) <~ 3e7: errors
            // of a few per cent in the weakest pivots merely leave cond(P) ~ 1.x); breakdowns are recorded in ndead.
            // Y = Yp R^-1 is formed inside the next product's operand-packing kernel (op_rebase_xp), by substitution, not by a
            // launch of its own; R^-1 itself is not needed during the iteration.
            op_dgemm(c.dev, true, false, LP, LP, dp, 1.0, Yp, LP, Yp, LP, 0.0, G, LP);
            if (use_pow) {   // re-base, then BOTH products of the next iteration in one pass (the last one keeps Z for U = Z (T Uh))
                const bool last = it + 1 == n_iter;
                dev_set_tag(c.dev, TAG_POW);
                have_yp = op_rebase_power_pass(c.dev, dt, X.p, n, dp, X.ld, muT.p, G, L, LP, 1e-15, ndead, Yp, LP, LP, T.f64(), LP, Y.f64(), LP,
                                               last ? Z.p : nullptr, LP, Yp, LP, /*steering=*/steer && !last);   // pca.rs:714 + 711
                // (steer: n_iter >= 3 -- the short iterations keep every pass at five / six piece products: ADVICE round 5)
                dev_set_tag(c.dev, TAG_NONE);
                if (have_yp) allreduce_f64(c, Yp, dp * LP + (last ? 1 : 0), PETAL_SUM);   // (the last one: [ Xc^T Z | sum Xc^2 ])
            }
            if (!have_yp) {
            dev_set_tag(c.dev, TAG_XP);   // (only the product kernel itself is bracketed)
            op_rebase_xp(c.dev, dt, X.p, n, dp, X.ld, muT.p, G, L, LP, 1e-15, ndead, Yp, LP, LP, T.f64(), LP, Y.f64(), LP, Z.p, LP, planes,
                         steer && it + 1 < n_iter);  // pca.rs:714 (the last iteration's Z is the iterate the results are made of: exact)
            dev_set_tag(c.dev, TAG_NONE);
            }
        } else {
            // ill-conditioned iterate: precondition with the tall side first.  Z = Xc P gives Z^T Z = P^T (Xc^T Z) =
            // P^T Yp without a pass over Z (op_chol_inv reads the upper triangle only); T = chol(Z^T Z)^-1 is applied on
            // the small side, Xc^T (Z T) = Yp T, and the now moderately conditioned Yp T gets its own Cholesky-QR.
            op_dgemm(c.dev, true, false, LP, LP, dp, 1.0, Pcur, LP, Yp, LP, 0.0, G, LP);
            op_chol_inv(c.dev, G, L, LP, T.f64(), LP, tol_drop, nullptr, LP);
            op_dgemm(c.dev, false, false, dp, LP, LP, 1.0, Yp, LP, T.f64(), LP, 0.0, Y.f64(), LP);
            // A column the dependence test dropped (its part of Z below ~1e-3 of the part already spanned: from a random start
            // every column of Z = Xc Omega is dominated by sigma_1, so a spectrum wider than 1e3 loses its tail here) is
            // refilled with the matching column of Omega instead of staying zero: orthogonalised against the kept columns
            // below, the next product shows it the largest direction still missing, and the block regains its width within an
            // iteration or two -- what the crate's pivoted LU (pca.rs:709-713) achieves by never letting the block lose
            // rank.  Truly rank-deficient data keeps dropping the refilled columns, down to the final QR (sigma = 0).
            op_refill_zero_cols(c.dev, Y.f64(), dp, L, LP, P.f64(), LP);
            orthonormalize_small(c, Y, dp, L, LP, 1e-13);
            dev_set_tag(c.dev, TAG_XP);
            op_gemm_xp(c.dev, dt, X.p, n, dp, X.ld, muT.p, Y.f64(), LP, LP, nullptr, Z.p, LP, nullptr);  // pca.rs:714
            dev_set_tag(c.dev, TAG_NONE);
        }
        Pcur = Y.f64();
    }

    // thin QR of Z (pca.rs:716) and B = Q^T Xc (pca.rs:681).
    t_rt = false;
    steered = use_pow || steer;
    if (n_iter > 0 && !robust) {
        // ONE more pass over X serves both.  Z = Xc Pcur, so Z^T Z = Pcur^T (Xc^T Z) = Pcur^T Yp with Yp = Xc^T Z, the very
        // product B needs: with H = Pcur^T Yp = R^T R and T = R^-1, Q = Z T and B = Q^T Xc = (Yp T)^T.  No pass over Z, no
        // second Gram matrix.  Why one Cholesky suffices here although cond(Z)^2 ~ 1e7 exceeds fp32: Pcur is orthonormal and
        // spans a nearly invariant subspace of Xc^T Xc, so the fp32 rounding of Pcur and Z perturbs Z^T Z MULTIPLICATIVELY,
        // (I + E)^T H with ||E|| ~ 1e-8, and ||Q^T Q - I|| ~ ||E|| cond(R) ~ 1e-5 instead of eps cond(Z)^2 (measured 1e-6 ..
        // 1e-5 up to cond(Z) = 2.5e4; the singular values match the two-pass Cholesky-QR2 form to its own accuracy).  Not so
        // for a non-orthonormal Pcur (n_iter = 0: the raw Omega) and on the robust redo: those keep Cholesky-QR2 below.
        if (!have_yp) {   // (the fused pass of the last iteration has formed this product already)
            dev_set_tag(c.dev, TAG_ATB);
            op_gemm_atb(c.dev, dt, X.p, X.ld, dp, muT.p, Z.p, LP, LP, nullptr, n, Yp, LP);
            dev_set_tag(c.dev, TAG_NONE);
            allreduce_f64(c, Yp, dp * LP + 1, PETAL_SUM);  // [ Xc^T Z | sum Xc^2 ]
        }
        op_dgemm(c.dev, true, false, LP, LP, dp, 1.0, Pcur, LP, Yp, LP, 0.0, G, LP);
        // (a pivot lost here -- H is only symmetric / positive up to the fp32 rounding of Pcur and Z -- is recorded like the
        // breakdowns of the power iterations: the fit is then redone on the robust Cholesky-QR2 path instead of silently
        // dropping a component.  On the planted spectra the pivot ratios r_jj^2 / H_jj stay above 0.9: Pcur's columns come out
        // of a Cholesky-QR ordered like the singular vectors, so Z's columns are nearly orthogonal.)
        // (only a lost pivot among the first k columns touches an output component: exactly low-rank fp32 data, whose columns beyond
        // the rank are dropped here with sigma = 0 -- the correct answer -- no longer pays for a second, robust fit)
        // (the inverse of R is not needed explicitly either: B^T = Yp R^-1 and R^-1 Uh are triangular solves -- where the device has the
        // kernels the factorisation is the register-resident one of the power iterations, 15 instead of 40 us at l = 74)
        t_rt = slot_flip && op_chol_rt(c.dev, dt, n, G, L, LP, T.f64(), LP, tol_drop, ndead, LP, std::max<int64_t>(k, 1));
        if (!slot_flip) op_chol_inv(c.dev, G, L, LP, T.f64(), LP, tol_drop, ndead, LP, std::max<int64_t>(k, 1));
        Usrc = Z.p; Ubuf = Z1.p;
    } else {
        // Cholesky-QR2: Z1 = Z T1, Q = Z1 T2 with T2 folded into the small side.  The Gram matrices of the tall side are formed
        // with fp64 products (`precise`): Z = Xc Omega of a wide spectrum has cond(Z)^2 ~ 1e7 .. 1e8, and a Gram matrix from fp32
        // products (good to 6e-8 of its norm) would have to DROP the weakest columns as noise -- n_iter = 0 on a planted 1e3
        // spectrum then lost a column of the oversampling tail and came back 9e-3 off where the crate's f32 QR (linalg.rs:127-147)
        // holds 5e-6; with exact products every pivot above the rounding of Z itself is kept.
        op_gemm_atb(c.dev, dt, Z.p, LP, LP, nullptr, Z.p, LP, LP, nullptr, n, G, LP, /*precise=*/true);
        allreduce_f64(c, G, LP * LP, PETAL_SUM);
        op_chol_inv(c.dev, G, L, LP, T.f64(), LP, tol_tall, nullptr, LP);
        op_gemm_xp(c.dev, dt, Z.p, n, LP, LP, nullptr, T.f64(), LP, LP, nullptr, Z1.p, LP, nullptr);
        op_gemm_atb(c.dev, dt, Z1.p, LP, LP, nullptr, Z1.p, LP, LP, nullptr, n, G, LP, /*precise=*/true);
        dev_set_tag(c.dev, TAG_ATB);
        op_gemm_atb(c.dev, dt, X.p, X.ld, dp, muT.p, Z1.p, LP, LP, nullptr, n, Yp, LP);  // B^T = Xc^T Q (pca.rs:681)
        dev_set_tag(c.dev, TAG_NONE);
        allreduce_f64(c, G, LP * LP, PETAL_SUM);
        allreduce_f64(c, Yp, dp * LP + 1, PETAL_SUM);
        op_chol_inv(c.dev, G, L, LP, T.f64(), LP, tol_tall, nullptr, LP);  // T2
        Usrc = Z1.p; Ubuf = Z.p;
    }
    if (t_rt) op_trsm_right(c.dev, Yp, dp, LP, T.f64(), LP, LP, Bt.f64(), LP);
    else op_dgemm(c.dev, false, false, dp, LP, LP, 1.0, Yp, LP, T.f64(), LP, 0.0, Bt.f64(), LP);

    };  // pipeline

    // The small stage: economy SVD of B, the verdicts, the components, U and svd_flip's column scan.  It runs behind every pipeline
    // and ONCE MORE BY ITSELF (jacobi = true) when the only complaint about a run is the eigen-solver's: eigenvalues among the k
    // wanted ones too close, relative to ||B B^T||, for the two-stage solver's vectors.  Until round 6 that verdict shared a word with
    // the pivot breakdowns and sent the whole fit through the robust pipeline -- twice the passes over X, and, far from
    // convergence, a DIFFERENT subspace (the robust iteration drops and refills dependent columns): uncentred data 3 sigma off
    // centre (the mean direction 1.6e4 x the block's weakest singular value), k = 100, came back 6.7e-3 / 1.9e-3 / 5.0e-4 off the oracle at
    // n_iter 3 / 4 / 5 where this stage alone, repeated with the Jacobi solver on the same B, gives 2.0e-5 / 1.2e-4 / 3.0e-5
    // (dev/fuzz_round6.py, dev/r6_case_a.py).
    auto small_stage = [&](bool robust, bool exact, bool jacobi) {
    if (jacobi) dev_memset(c.dev, ndead, 0, 2 * sizeof(double));   // (the heavy-tail verdict is formed again from the new spectrum)
    // economy SVD of B (l x d) (svddc, pca.rs:682): eigen-decomposition of B B^T in fp64
    op_dgemm(c.dev, true, false, LP, LP, dp, 1.0, Bt.f64(), LP, Bt.f64(), LP, 0.0, S.f64(), LP);
    // only the leading L x L block of S is non-zero (columns L..LP-1 of every iterate are exact zero padding)
    // Optimistic run: the two-stage solver alone, its closeness verdict restricted to the k pairs that reach the outputs
    // (the oversampling tail may cluster at the noise floor: its vectors only have to span it) and OR-ed into `ndead` -- a
    // flagged spectrum (exact multiplicities among the wanted singular values) redoes the fit on the robust path, which
    // solves with Jacobi.  Saves the two fallback launches that return at once on every separated spectrum.
    if (!robust && !jacobi) op_eigh(c.dev, S.f64(), L, LP, Uh.f64(), LP, lam, dt == F32 ? 1e-8 : 1e-15, false, LP, std::max<int64_t>(k, 1), neig);
    else op_eigh(c.dev, S.f64(), L, LP, Uh.f64(), LP, lam, dt == F32 ? 1e-8 : 1e-15, true, LP);  // (zero padding of Uh included)
    // (the two-plane verdict, from the spectrum just found)
    if (!robust && !exact && two_plane_applies)
        op_tail_verdict(c.dev, lam, L, std::max<int64_t>(k, 1), (tv_from_sq && !tv_direct) ? mu64 : nullptr, dp, d, ri.n_total,
                        tv_direct ? mu64 + dp : tvp, steered ? 4e-6 * 1.2 : 4e-6, p2_thr, ndead);   // (steering passes round Xc and z too: where the estimate
                        // matters -- slowly decaying spectra -- that adds at most 16 % to what P's rounding costs: dev/x2_model.py)
    // rows of V^T: v_j = B^T u_j / sigma_j, sigma_j = sqrt(lam_j) (the host takes the same square roots of lam)
    op_components_out(c.dev, dt, Bt.f64(), LP, Uh.f64(), LP, lam, dt == F32 ? 1e-7 : 1e-12, d, L, k, comp_dev);

    // U = Q Uh = Usrc (T Uh) (pca.rs:683) and svd_flip (pca.rs:684)
    // (only the k columns svd_flip signs and fit_transform returns: kp = k rounded up to whole 16-column tiles; the small
    // product T Uh is formed by the operand-packing kernel of the n x l product)
    dev_set_tag(c.dev, TAG_STREAM);
    if (slot_flip)   // ... and svd_flip's column scan by its epilogue
        op_gemm_xp_prod_absmax(c.dev, dt, Usrc, n, LP, LP, nullptr, T.f64(), LP, LP, Uh.f64(), kp, LP, M2.f64(), LP, Ubuf, LP,
                               ri.row_offset, flip, flip + kp, flip + 2 * kp, /*store_product=*/y_out != nullptr, /*a_rt=*/t_rt);
    else
        op_gemm_xp_prod(c.dev, dt, Usrc, n, LP, LP, nullptr, T.f64(), LP, LP, Uh.f64(), kp, LP, M2.f64(), LP, Ubuf, LP);
    dev_set_tag(c.dev, TAG_NONE);
    Uout = Ubuf;
    };  // small_stage

    // results (pca.rs:543-547): ONE small copy queued behind the pipeline, ONE synchronisation; the host reads it where it
    // lands (the pinned ring) and writes the caller's arrays in one pass, svd_flip's sign applied on the way
    const size_t hres_len = size_t(res_len - o_tv);
    const double* hres = nullptr;
    const void* hcomp = nullptr;
    std::vector<double> sg, keep;
    std::vector<char> comp_keep;
    double t_q = 0, t_s = 0;
    const bool never_p2 = dev_option(c.dev, OPT_TWO_PLANE) == 0;
    bool robust = false, exact = !two_plane_applies || never_p2;
    bool jacobi = false;   // the small stage alone is being repeated with the Jacobi eigen-solver
    for (int attempt = 0; attempt < 8; ++attempt) {
        if (!jacobi) pipeline(robust, exact);
        small_stage(robust, exact, jacobi);
        if (slot_flip) flip_signs_to_slot(c, dt, Uout, n, kp, LP, ri.row_offset, flip, true, !robust ? ndead : nullptr);
        // (one view for both while they fit a ring slot; large components -- k = 128 at d = 16384, k = 512 at d = 4096 -- leave by a
        // plain copy of their own and only the small block is viewed: ADVICE round 4)
        const bool one_view = sizeof(double) * hres_len + comp_bytes <= dev_view_limit(c.dev);
        hres = static_cast<const double*>(dev_d2h_view(c.dev, tvp, sizeof(double) * hres_len + (one_view ? comp_bytes : 0)));
        hcomp = hres + hres_len;
        if (!one_view) {
            comp_keep.resize(comp_bytes);
            dev_d2h(c.dev, comp_keep.data(), comp_dev, comp_bytes);
            hcomp = comp_keep.data();
        }
        t_q = timer.ms();
        dev_sync(c.dev);
        t_s = timer.ms();
        // (the sharded fp64 path queues more copies through the ring below -- agree_any(), flip_signs(): it takes everything out
        // first; the small block is copied on every path)
        keep.assign(hres, hres + hres_len);
        if (one_view && sharded(c) && !flip_slot_keys(c, dt)) {
            comp_keep.assign(static_cast<const char*>(hcomp), static_cast<const char*>(hcomp) + comp_bytes);
            hcomp = comp_keep.data();
        }
        if (robust) break;
        // the verdict words: [0] pivot breakdowns -> the robust path; [1] a heavy tail behind the block -> the same pipeline with
        // three-plane operands; [2] a cluster among the wanted eigenvalues -> the small stage again with the Jacobi solver (the passes
        // over X stand).  Sharded fits branch on the MAX over the ranks.
        int hdead[3] = {0, 0, 0};
        std::memcpy(hdead, &keep[o_dead - o_tv], sizeof(hdead));
        double code = hdead[0] != 0 ? 3.0 : (hdead[1] != 0 ? 1.0 : (hdead[2] != 0 ? 2.0 : 0.0));
        if (sharded(c)) code = flip_slot_keys(c, dt) ? keep[o_flip - o_tv + 4 * kp] : agree_max(c, code);  // the agreed verdict
        if (code >= 3.0 && !rebase_now && n_iter >= 1) {
            // A lost pivot is, on full-rank data, the FIRST product pair's: from a random start every column of Xc^T (Xc Omega)
            // is dominated by sigma_1, and beyond sigma_1 / sigma_l ~ 5e3 its Gram matrix is singular to fp64 (uncentred data far off
            // centre: the mean direction is sigma_1).  The robust path would answer with a different iteration (dependent columns dropped
            // and refilled): 1.9e-3 / 5e-4 off the oracle at n_iter 4 / 5 on such data, where the same pipeline with the sketch re-based
            // on the tall side first -- what short iterations do anyway, and the crate's first LU -- holds 1e-4 / 3e-5.  So that is
            // tried first; rank-deficient data loses its pivot again and takes the robust path one run later.  (fp64 data too: its first pair
            // breaks down beyond sigma_1 / sigma_l ~ 5e3 all the same -- (sigma_1 / sigma_l)^4 against 1e-15 -- while the re-based sketch's
            // tall-side Gram matrix holds to ~3e6; uncentred fp64 data 300 sigma off centre came back 1e-5 ... 1e-7 off at n_iter 1 - 3 on the
            // robust path: dev/fuzz_hostsim6.py)
            rebase_now = true; jacobi = false; c.stats.rpca_redo = 3;
            continue;
        }
        if (code >= 3.0) { robust = true; exact = true; jacobi = false; c.stats.rpca_redo = 2; continue; }
        if (code == 1.0 && !exact) { exact = true; jacobi = false; c.stats.rpca_redo = 1; continue; }
        if (code == 2.0 && !jacobi) { jacobi = true; c.stats.eigh_redo = 1; continue; }
        break;
    }
    if (slot_flip) {
        const double* hf = &keep[o_flip - o_tv];
        sg = flip_slot_keys(c, dt) ? signs_from_triple(std::vector<double>(hf + 3 * kp, hf + 4 * kp), kp)
                                   : signs_from_triple(std::vector<double>(hf, hf + 3 * kp), kp);
    }
    if (!slot_flip) sg = flip_signs(c, dt, Uout, n, kp, LP, ri.row_offset);   // (sharded fp64: three more dependent rounds)
    const double* hlam = &keep[o_lam - o_tv];
    std::vector<double> hs(size_t(std::max<int64_t>(k, 1)), 0.0);
    for (int64_t j = 0; j < k; ++j) hs[j] = std::sqrt(std::max(hlam[j], 0.0));
    const double* hmu = &keep[o_mu - o_tv];
    double htv = keep[0];
    if (tv_direct) htv = hmu[dp];   // (the means were gathered inside the first fused pass: the variance came with them)
    else if (tv_from_sq) {  // sum (x - mu)^2 = sum x^2 - n mu^2 per column, in fp64
        htv = 0;
        for (int64_t j = 0; j < d; ++j) htv += std::max(0.0, hmu[dp + j] - ri.n_total * hmu[j] * hmu[j]);
    }
    {   // non-finite input: the crate's gesdd reports info != 0 -> "did not converge" (linalg.rs:115).  Checked on what came back anyway --
        // means, sums of squares, the spectrum: replicated values, so every rank of a sharded fit raises -- and on the RAW values (the
        // max(., 0) clamps above turn a NaN into a zero)
        bool finite = std::isfinite(keep[0]) && std::isfinite(htv);
        for (int64_t j = 0; j < d; ++j) finite = finite && std::isfinite(hmu[j]) && (!tv_from_sq || tv_direct || std::isfinite(hmu[dp + j]));
        for (int64_t j = 0; j < k; ++j) finite = finite && std::isfinite(hlam[j]);
        if (!finite) linalg_error("did not converge");
    }
    for (int64_t j = 0; j < k; ++j) {   // the components leave the ring with svd_flip's sign on row j (pca.rs:684)
        if (dt == F32) {
            const float* src = static_cast<const float*>(hcomp) + j * d;
            float* r = static_cast<float*>(components) + j * d;
            if (sg[j] < 0) for (int64_t i = 0; i < d; ++i) r[i] = -src[i];
            else std::memcpy(r, src, sizeof(float) * size_t(d));
        } else {
            const double* src = static_cast<const double*>(hcomp) + j * d;
            double* r = static_cast<double*>(components) + j * d;
            if (sg[j] < 0) for (int64_t i = 0; i < d; ++i) r[i] = -src[i];
            else std::memcpy(r, src, sizeof(double) * size_t(d));
        }
        put_elem(singular, dt, j, hs[j]);
    }
    if (dt == F32) { float* m = static_cast<float*>(means); for (int64_t i = 0; i < d; ++i) m[i] = float(hmu[i]); }
    else std::memcpy(means, hmu, sizeof(double) * size_t(d));
    put_elem(total_variance, dt, 0, htv);
    if (y_out) {  // fit_transform: U[:, :k] * sigma (transform_with_u, pca.rs:758-779)
        std::vector<double> sc(LP, 0.0);
        for (int64_t j = 0; j < k; ++j) sc[j] = sg[j] * hs[j];
        DBuf dsc(c.dev, sizeof(double) * LP);
        dev_h2d(c.dev, dsc.p, sc.data(), dsc.bytes);
        emit(c, dt, Uout, n, k, LP, *y_out, dsc.f64());
    }
    (void)t_q; (void)t_s;
    c.stats.means_folded = tv_direct ? 1 : 0;
    finish_stats(c, timer);
}

// ---------------------------------------------------------------------------------------------
// Pca::inner_fit (pca.rs:195-231).  The reference asks LAPACK for the full n x n U; only its first
// min(n,d) columns are ever read (svd_flip zips U columns with V^T rows; transform_with_u takes k),
// so the thin factorisation is computed: eigen-decomposition of the d x d Gram matrix in fp64.
void pca_fit(petal_ctx& c, const petal_matrix& x, int64_t k, bool centering, void* components, void* means,
             void* singular, void* total_variance, const petal_matrix* y_out) {
    Timer timer;
    dev_reset_timing(c.dev);
    c.stats = petal_stats{};
    check_matrix(x, "input");
    if (k < 0) invalid_input("negative parameter");
    const int dt = x.dtype;
    const int64_t d = x.cols;
    DevMat X;
    ShardPrologue pro;
    RankInfo ri;
    if (sharded(c)) {
        X = ingest(c, x);
        pro = shard_prologue(c, X, centering ? X.dp : 0, false, dt, nullptr, 0);
        ri = pro.ri;
    } else {
        ri.n_total = double(x.rows);
    }
    const int64_t n_total = int64_t(ri.n_total);
    if (n_total < k || d < k) invalid_input("every dimension should be at least " + std::to_string(k));  // pca.rs:199-204
    if (centering && n_total == 0) return;  // pca.rs:207-211
    if (d == 0 || n_total == 0) {
        if (total_variance) put_elem(total_variance, dt, 0, 0.0);
        if (y_out) emit(c, dt, nullptr, x.rows, 0, 0, *y_out);
        return;
    }
    if (!sharded(c)) X = ingest(c, x);
    const int64_t n = X.n, dp = X.dp;
    const size_t esz = dtype_size(dt);
    DBuf mu64, muT;
    column_means(c, X, ri.n_total, centering, mu64, muT, false, sharded(c) && centering ? pro.sums : nullptr);

    DBuf C(c.dev, sizeof(double) * dp * dp), V(c.dev, sizeof(double) * dp * dp), lam(c.dev, sizeof(double) * dp);
    DBuf sig(c.dev, sizeof(double) * dp), inv(c.dev, sizeof(double) * dp), diag(c.dev, sizeof(double) * dp), r3(c.dev, sizeof(double) * 3);
    DBuf compd(c.dev, esz * size_t(std::max<int64_t>(k, 1)) * d), U, fslot;
    const bool slot_flip = !sharded(c) || dt == F32;  // (sharded fp64 decides the signs in three dependent all-reduce rounds)
    std::vector<double> hdiag(dp), hs(dp), hmu(dp), deferred, sg;
    int64_t r = 0, rp = 16;
    // The device pipeline: nothing is read back in the middle except what a decision needs -- the subspace iteration's residual
    // verdict every second product (an optimistic fixed-length iteration, as FastICA's whitening runs it, was measured: a tall
    // Pca whose spectrum needs four products instead of two then pays the whole pipeline twice, 1.10 -> 1.74 ms) and, for fp64
    // data, the spectrum that chooses the accurate route -- and one synchronisation at the end delivers the results.  The
    // components leave in the caller's layout and type.
    auto pipeline = [&](bool optimistic, bool route_check) -> bool {
    op_gemm_atb(c.dev, dt, X.p, X.ld, dp, muT.p, X.p, X.ld, dp, muT.p, n, C.f64(), dp, true);
    allreduce_f64(c, C.f64(), dp * dp, PETAL_SUM);
    // total_variance = sigma . sigma over ALL singular values (pca.rs:224) = trace of the Gram matrix: its diagonal is set aside
    // before the eigen-solvers overwrite C
    dev_copy2d(c.dev, diag.p, sizeof(double), C.p, (dp + 1) * sizeof(double), sizeof(double), size_t(dp), 2);
    dev_memset(c.dev, V.p, 0, V.bytes);
    dev_memset(c.dev, lam.p, 0, lam.bytes);
    // only the top-k pairs reach the outputs (components, singular values, the k columns of U that svd_flip signs)
    DBuf Ckeep;  // (the eigen-solvers may destroy their input; the accurate route of fp64 fits factors the Gram matrix again)
    if (dt == F64) { Ckeep = DBuf(c.dev, C.bytes); dev_d2d(c.dev, Ckeep.p, C.p, C.bytes); }
    const double vtol = dt == F32 ? 1e-12 : 3e-14;
    const bool partial = topk_eigh(c, C.f64(), d, dp, k, V.f64(), lam.f64(), dt == F32 ? 1e-8 : 1e-5, optimistic ? r3.f64() : nullptr, vtol);
    // (order d, not the padded dp: the zero padding would only add dp - d exact zero eigenvalues, a cluster that sends the
    // two-stage solver to its Jacobi fallback; V and lam beyond d stay at the zeros set above)
    // (... and the closeness verdict of the two-stage solver covers the k pairs that reach the outputs: the noise-floor
    // eigenvalues behind them cluster on every planted spectrum and would send each such fit to the Jacobi solver -- 98 + 24 us of
    // configs[0]'s 0.34 ms -- for vectors whose only use is a discarded sign)
    if (!partial) op_eigh(c.dev, C.f64(), d, dp, V.f64(), dp, lam.f64(), dt == F32 ? 1e-8 : 1e-15, false, 0, std::max<int64_t>(k, 1));
    // fp64 data with wanted singular values below 10^-3.5 sigma_1 (by the Gram route's own estimate): the QR + one-sided
    // Jacobi route keeps them to eps sigma_1 / sigma_k like the crate's gesvd (linalg.rs:70-91); two more passes over X
    // (the check needs the spectrum on the host: the first run skips it and the caller looks at the singular values that come back
    // with the results -- a fit that does need the route, rare, is run again with the check in place; every fp64 fit used to pay
    // a host round trip in the middle of its queue for it)
    bool accurate = false;
    if (dt == F64 && k > 0 && route_check) {
        std::vector<double> hl(k);
        dev_d2h(c.dev, hl.data(), lam.p, sizeof(double) * k);
        dev_sync(c.dev);
        // (lam[k-1] <= 0 or fewer samples than features: rank-deficient, the accurate route's Cholesky would only break down --
        // after allocating a second copy of X and two more passes)
        const bool low = hl[0] > 0 && hl[k - 1] > 0 && n_total >= d && std::sqrt(hl[k - 1] / hl[0]) < GRAM_ROUTE_FLOOR;
        if (agree_any(c, low)) accurate = accurate_small_svd(c, X, muT.p, Ckeep.f64(), d, V.f64(), sig.f64());
    }
    if (!accurate) op_sigma_inv(c.dev, lam.f64(), sig.f64(), inv.f64(), dp, dt == F32 ? 1e-6 : 1e-10);
    else op_dvec(c.dev, 1, sig.f64(), inv.f64(), dp, dt == F32 ? 1e-6 : 1e-10);

    // U[:, j] = Xc v_j / sigma_j for j < r: only the columns svd_flip looks at (all min(n, d) of them in the crate;
    // the signs of columns >= k never reach an output)
    r = partial ? k : std::min(n_total, d);
    rp = round_up(std::max<int64_t>(r, 1), 16);
    DBuf Pm(c.dev, sizeof(double) * dp * rp);
    op_scale_pad_cols(c.dev, V.f64(), dp, inv.f64(), dp, std::min(r, dp), rp, Pm.f64());
    U = DBuf(c.dev, esz * size_t(std::max<int64_t>(n, 1)) * rp);
    if (slot_flip) {
        // pca.rs:223: svd_flip's scan comes out of the product kernel's accumulators (no second pass over U; U itself is only
        // stored for fit_transform); decision data queued, decoded below
        fslot = DBuf(c.dev, sizeof(double) * (4 * rp + 1));
        op_gemm_xp_absmax(c.dev, dt, X.p, n, dp, X.ld, muT.p, Pm.f64(), rp, rp, U.p, rp, ri.row_offset, fslot.f64(), fslot.f64() + rp,
                          fslot.f64() + 2 * rp, /*store_product=*/y_out != nullptr);
        flip_signs_to_slot(c, dt, U.p, n, rp, rp, ri.row_offset, fslot.f64(), true);
        deferred.assign(size_t(4 * rp), 0.0);   // (leaves with the other results below)
    } else {
        op_gemm_xp(c.dev, dt, X.p, n, dp, X.ld, muT.p, Pm.f64(), rp, rp, nullptr, U.p, rp, nullptr);
        sg = flip_signs(c, dt, U.p, n, r, rp, ri.row_offset);
    }
    op_transpose_out(c.dev, dt, V.f64(), dp, d, k, compd.p);

    double h3[3] = {0, 1, 0};
    {   // every result in one launch
        void* dsts[6] = {hdiag.data(), hs.data(), hmu.data(), components, h3, deferred.data()};
        const void* srcs[6] = {diag.p, sig.p, mu64.p, compd.p, r3.p, fslot.p};
        const size_t lens[6] = {diag.bytes, sig.bytes, sizeof(double) * size_t(dp), k > 0 ? esz * size_t(k) * d : 0,
                                (partial && optimistic) ? sizeof(h3) : 0, slot_flip ? sizeof(double) * size_t(4 * rp) : 0};
        dev_d2h_multi(c.dev, 6, dsts, srcs, lens);
    }
    dev_sync(c.dev);
    return !(partial && optimistic) || topk_verdict_ok(h3, vtol);
    };  // pipeline
    pipeline(false, false);
    if (dt == F64 && k > 0) {
        const bool low = hs[0] > 0 && hs[k - 1] > 0 && n_total >= d && hs[k - 1] / hs[0] < GRAM_ROUTE_FLOOR;
        if (agree_any(c, low)) pipeline(false, true);
    }
    if (slot_flip)   // (sharded fp32: the all-reduced keys behind the triple; else the triple itself)
        sg = flip_slot_keys(c, dt) ? signs_from_triple(std::vector<double>(deferred.begin() + 3 * rp, deferred.begin() + 4 * rp), rp)
                                   : signs_from_triple(std::vector<double>(deferred.begin(), deferred.begin() + 3 * rp), rp);
    double tvar = 0;
    for (int64_t j = 0; j < d; ++j) tvar += hdiag[j];
    // (non-finite input: the crate's gesvd comes back with info != 0 -> "did not converge", linalg.rs:84; the trace is replicated,
    // so every rank of a sharded fit raises)
    if (!std::isfinite(tvar)) linalg_error("did not converge");
    for (int64_t j = 0; j < k; ++j) {   // svd_flip's sign on row j of the components (already in place)
        if (sg[j] < 0) {
            if (dt == F32) { float* row = static_cast<float*>(components) + j * d; for (int64_t i = 0; i < d; ++i) row[i] = -row[i]; }
            else { double* row = static_cast<double*>(components) + j * d; for (int64_t i = 0; i < d; ++i) row[i] = -row[i]; }
        }
        put_elem(singular, dt, j, hs[j]);
    }
    for (int64_t i = 0; i < d; ++i) put_elem(means, dt, i, hmu[i]);
    put_elem(total_variance, dt, 0, tvar);
    if (y_out) {
        std::vector<double> sc(rp, 0.0);
        for (int64_t j = 0; j < k; ++j) sc[j] = sg[j] * hs[j];
        DBuf dsc(c.dev, sizeof(double) * rp);
        dev_h2d(c.dev, dsc.p, sc.data(), dsc.bytes);
        emit(c, dt, U.p, n, k, rp, *y_out, dsc.f64());
    }
    finish_stats(c, timer);
}

// ---------------------------------------------------------------------------------------------
// free fn transform (pca.rs:726-750) / FastIca::transform (ica.rs:120-131)
// May the product kernel write its n x cols result straight into the caller's matrix?  A device matrix of the same type with unit
// column stride, no padding columns to drop (cols a multiple of 16) and 16-byte aligned rows: then there is no staging buffer and
// no copy-out pass (1e6 x 64 floats: 0.15 ms of a 0.9 ms transform).  The shape / type checks of emit() are made here as well.
static bool writes_in_place(const petal_matrix& out, int dtype, int64_t n, int64_t cols, const petal_matrix& in) {
    check_matrix(out, "output");
    if (out.dtype != dtype) invalid_input("output dtype differs from input dtype");
    if (out.rows != n || out.cols != cols) invalid_input("output has the wrong shape");
    const size_t esz = dtype_size(dtype);
    if (!(out.space == PETAL_DEVICE && n > 0 && cols > 0 && cols % 16 == 0 && out.col_stride == 1 && out.row_stride >= cols &&
          (size_t(out.row_stride) * esz) % 16 == 0 && (reinterpret_cast<uintptr_t>(out.data) & 15) == 0))
        return false;
    // (an output that overlaps the input keeps the staged form: the product would read rows it has already overwritten)
    if (in.space == PETAL_DEVICE && in.rows > 0 && in.cols > 0) {
        const auto span = [&](const petal_matrix& m, uintptr_t& lo, uintptr_t& hi) {
            const int64_t last = (m.rows - 1) * std::abs(m.row_stride) + (m.cols - 1) * std::abs(m.col_stride) + 1;
            const uintptr_t p0 = reinterpret_cast<uintptr_t>(m.data), ext = size_t(last) * dtype_size(m.dtype);
            lo = (m.row_stride < 0 || m.col_stride < 0) ? p0 - std::min<uintptr_t>(p0, ext) : p0;   // (a reversed view reaches below its base)
            hi = p0 + ext;
        };
        uintptr_t alo, ahi, blo, bhi;
        span(in, alo, ahi); span(out, blo, bhi);
        if (alo < bhi && blo < ahi) return false;
    }
    return true;
}
// k x d components (the caller's type, row-major) -> the fp64 operand the product kernels take: P[i][j] = comp[j][i] (transposed = true,
// dp x kp) or P[j][i] = comp[j][i] (kp x dp), zero padded
static std::vector<double> components_operand(const void* components, int dt, int64_t k, int64_t d, int64_t rows, int64_t cols, bool transposed) {
    std::vector<double> h(size_t(rows) * cols, 0.0);
    auto fill = [&](auto* comp) {
        for (int64_t j = 0; j < k; ++j)
            for (int64_t i = 0; i < d; ++i) h[transposed ? size_t(i) * cols + j : size_t(j) * cols + i] = double(comp[j * d + i]);
    };
    if (dt == F64) fill(static_cast<const double*>(components)); else fill(static_cast<const float*>(components));
    return h;
}
void transform(petal_ctx& c, const petal_matrix& x, const void* components, const void* means, int64_t k, int64_t d,
               bool centering, const petal_matrix& y_out) {
    check_matrix(x, "input");
    if (x.cols != d) invalid_input("# of columns should be " + std::to_string(d));  // pca.rs:736-741
    const int dt = x.dtype;
    if (x.rows == 0 || k == 0) { emit(c, dt, nullptr, x.rows, k, 0, y_out); return; }
    DevMat X = ingest(c, x);
    const int64_t dp = X.dp, kp = round_up(k, 16);
    const bool in_place = writes_in_place(y_out, dt, X.n, k, x);
    // (operands through the pinned ring, queued: no host wait in front of the product)
    const std::vector<double> hP = components_operand(components, dt, k, d, dp, kp, true);
    DBuf P(c.dev, sizeof(double) * dp * kp), muT(c.dev, dtype_size(dt) * dp);
    dev_h2d_async(c.dev, P.p, hP.data(), P.bytes);
    std::vector<char> hmu(dtype_size(dt) * dp, 0);
    if (centering) std::memcpy(hmu.data(), means, dtype_size(dt) * d);
    dev_h2d_async(c.dev, muT.p, hmu.data(), muT.bytes);
    DBuf Y;
    if (!in_place) Y = DBuf(c.dev, dtype_size(dt) * size_t(X.n) * kp);
    op_gemm_xp(c.dev, dt, X.p, X.n, dp, X.ld, centering ? muT.p : nullptr, P.f64(), kp, kp, nullptr, in_place ? y_out.data : Y.p,
               in_place ? y_out.row_stride : kp, nullptr);
    if (!in_place) emit(c, dt, Y.p, X.n, k, kp, y_out);
    dev_sync(c.dev);
}

// inverse_transform (pca.rs:788-811)
void inverse_transform(petal_ctx& c, const petal_matrix& y, const void* components, const void* means, int64_t k,
                       int64_t d, bool centering, const petal_matrix& x_out) {
    check_matrix(y, "input");
    if (y.cols != k) invalid_input("# of columns should be " + std::to_string(k));  // pca.rs:798-803
    const int dt = y.dtype;
    if (y.rows == 0 || d == 0) { emit(c, dt, nullptr, y.rows, d, 0, x_out); return; }
    DevMat Y = ingest(c, y);  // n x kp
    const int64_t kp = Y.dp, dp = round_up(d, 16);
    const bool in_place = writes_in_place(x_out, dt, Y.n, d, y);
    const std::vector<double> hP = components_operand(components, dt, k, d, kp, dp, false);
    DBuf P(c.dev, sizeof(double) * kp * dp), muT(c.dev, dtype_size(dt) * dp);
    dev_h2d_async(c.dev, P.p, hP.data(), P.bytes);
    std::vector<char> hmu(dtype_size(dt) * dp, 0);
    if (centering) std::memcpy(hmu.data(), means, dtype_size(dt) * d);
    dev_h2d_async(c.dev, muT.p, hmu.data(), muT.bytes);
    DBuf Xo;
    if (!in_place) Xo = DBuf(c.dev, dtype_size(dt) * size_t(Y.n) * dp);
    op_gemm_xp(c.dev, dt, Y.p, Y.n, kp, Y.ld, nullptr, P.f64(), dp, dp, centering ? muT.p : nullptr, in_place ? x_out.data : Xo.p,
               in_place ? x_out.row_stride : dp, nullptr);
    if (!in_place) emit(c, dt, Xo.p, Y.n, d, dp, x_out);
    dev_sync(c.dev);
}

// ---------------------------------------------------------------------------------------------
namespace {

// ica_par (ica.rs:319-361) on a device-resident, sample-major X1T (n x ncp).  W (device f64 nc x nc)
// holds w_init on entry and the result on exit.  Returns n_iter.
// The decorrelation of w_init (ica.rs:329) does not depend on the data: fastica_fit runs it on the side stream under the whitening
// (ica_prepare between dev_fork and dev_fork_end), ica_par right before its loop.  W: w_init in, the decorrelated iterate out;
// state: the loop's {converged at, iterations done} words, cleared by the same launch.
struct IcaStart { DBuf W0, state; };
IcaStart ica_prepare(petal_ctx& c, int64_t nc, DBuf& W, int mode) {
    // (the caller's buffer holds w_init; the iterate lives in a buffer of this function's and the two are exchanged -- no copy)
    IcaStart s{DBuf(c.dev, W.bytes), DBuf(c.dev, 2 * sizeof(int))};
    std::swap(W, s.W0);
    op_symdecorr(c.dev, nc, s.W0.f64(), W.f64(), mode, s.state.as<int>());
    return s;
}
int64_t ica_loop(petal_ctx& c, int dt, const void* X1T, int64_t n, int64_t nc, int64_t ld, double n_total, DBuf& W,
                 double tol, int64_t max_iter, int mode, IcaStart* prepared = nullptr) {
    IcaStart own;
    if (!prepared) { own = ica_prepare(c, nc, W, mode); prepared = &own; }
    DBuf& state = prepared->state;
    DBuf GX(c.dev, sizeof(double) * (nc * nc + nc));
    op_ica_prepare(c.dev, dt, X1T, n, nc, ld);   // (what is constant over the loop: the bf16 planes of X1, made once)
    int hstate[2] = {0, 0};
    auto enqueue = [&](int64_t it, int* progress) {
        dev_set_tag(c.dev, TAG_ICA);
        op_ica_step(c.dev, dt, X1T, n, nc, ld, W.f64(), GX.f64(), state.as<int>());  // ica.rs:332-333
        dev_set_tag(c.dev, TAG_NONE);
        allreduce_f64(c, GX.f64(), nc * nc + nc, PETAL_SUM);
        op_ica_tail(c.dev, nc, n_total, W.f64(), GX.f64(), mode, tol, state.as<int>(), int(it), progress);  // ica.rs:334-358
    };
    int64_t n_iter = -1;
    if (!sharded(c)) {
        // Converged iterations are no-ops on the device.  The host follows the loop WITHOUT synchronising: the tail kernel
        // stores {converged at, iterations done} to pinned host memory, the host enqueues iteration `it` once the device has
        // finished iteration it - RUN_AHEAD (so the queue never runs dry and never runs far ahead) and stops at the flag: at
        // most RUN_AHEAD no-op iterations are launched after convergence (the round-1 schedule synchronised after 4, 12, 28,
        // ... iterations: two round trips and up to three wasted iterations for a fit that converges at iteration 9).
        // The iteration count itself comes from those words too: the loop ends without a synchronisation of its own.
        constexpr int64_t RUN_AHEAD = 2;
        volatile int* hp = dev_host_progress(c.dev);
        hp[0] = 0; hp[1] = 0;
        auto t_progress = std::chrono::steady_clock::now();  // when the device last reported a finished iteration
        int last_seen = 0;
        bool synced = false;
        int64_t it = 0;
        for (; it < max_iter && !hp[0]; ++it) {
            int spins = 0;
            while (!hp[0] && it - int64_t(hp[1]) >= RUN_AHEAD) {
                if ((++spins & 0xFFFF) != 0) continue;
                const auto now = std::chrono::steady_clock::now();
                if (hp[1] != last_seen) { last_seen = hp[1]; t_progress = now; }
                // a device that stopped reporting for 30 s (not: a fit that has been running for 30 s): blocking wait
                if (std::chrono::duration<double>(now - t_progress).count() > 30.0) { dev_sync(c.dev); t_progress = now; synced = true; break; }
            }
            if (hp[0]) break;
            enqueue(it, const_cast<int*>(hp));
        }
        if (hp[0]) {
            n_iter = hp[0];                       // converged: the tail kernel published the iteration it stopped at
        } else {                                  // ran to max_iter (or the progress words failed): the device state decides
            (void)synced;
            dev_d2h(c.dev, hstate, state.p, sizeof(hstate));
            dev_sync(c.dev);
            n_iter = hstate[0] ? hstate[1] : max_iter;
        }
    } else {
        // several ranks: every rank must enqueue the SAME number of iterations (each carries an all-reduce), so the flag --
        // identical on all ranks after the replicated tail -- is read at fixed iteration counts, every fourth iteration
        int64_t it = 0;
        while (it < max_iter) {
            const int64_t stop = std::min<int64_t>(max_iter, it + 4);
            for (; it < stop; ++it) enqueue(it, nullptr);
            dev_d2h(c.dev, hstate, state.p, sizeof(hstate));
            dev_sync(c.dev);
            if (hstate[0]) break;
        }
        n_iter = hstate[0] ? hstate[1] : max_iter;
    }
    c.stats.n_iter = n_iter;
    c.stats.ica_step_flops = 4.0 * double(nc) * nc * double(n);
    c.stats.ica_step_bytes = double(dtype_size(dt)) * nc * double(n);
    return n_iter;
}

void check_finite_w(const std::vector<double>& w) {
    for (double v : w)
        if (!std::isfinite(v)) linalg_error("cannot compute eigenvalues");  // the reference panics here (ica.rs:369)
}

}  // namespace

// FastIca::inner_fit (ica.rs:167-221)
void fastica_fit(petal_ctx& c, const petal_matrix& x, int64_t n_components, double tol, int64_t max_iter, int mode,
                 const void* w_init, void* components, void* means, int64_t* n_iter, const petal_matrix* y_out) {
    Timer timer;
    dev_reset_timing(c.dev);
    c.stats = petal_stats{};
    check_matrix(x, "input");
    const int dt = x.dtype;
    const int64_t d = x.cols;
    if (n_iter) *n_iter = 0;
    DevMat X;
    ShardPrologue pro;
    RankInfo ri;
    // sharded: the caller sizes w_init before the row counts are known -- n_components, or the number of features
    const int64_t nc_draw = n_components > 0 ? n_components : d;
    if (sharded(c)) {
        if (d > 0 && w_init == nullptr) invalid_input("w_init is required");  // (before the collective: every rank alike)
        X = ingest(c, x);
        pro = shard_prologue(c, X, X.dp, false, dt, w_init, nc_draw * nc_draw);
        ri = pro.ri;
    } else {
        ri.n_total = double(x.rows);
    }
    const int64_t n_total = int64_t(ri.n_total);
    if (n_total == 0) return;  // ica.rs:174-176
    int64_t nc = n_components > 0 ? n_components : std::min(n_total, d);  // ica.rs:173
    if (nc > std::min(n_total, d)) invalid_input("n_components should be at most min(n_samples, n_features)");
    if (nc == 0 || d == 0) return;
    if (w_init == nullptr) invalid_input("w_init is required");
    if (sharded(c) && nc != nc_draw) invalid_input("a sharded fit with fewer samples than features needs n_components");
    if (!sharded(c)) X = ingest(c, x);
    const int64_t n = X.n, dp = X.dp, ncp = round_up(nc, 16);
    const size_t esz = dtype_size(dt);
    DBuf mu64, muT;
    // Single-rank fp32 fits whose covariance comes from the split-product Gram kernel gather the column means in that same pass over X
    // (op_gram_split with mu64_fold: a provisional centre from a row sample, the column sums about it beside the diagonal tiles, the
    // move to the true centre in the reduction) -- the means pass of its own (0.18 ms at 500000 x 512) is only queued when that
    // path does not apply.
    const bool fold_means = !sharded(c) && dt == F32 && dp >= 256 && dev_gemm_mode(c.dev) != 1;
    bool means_done = false;
    auto means_pass = [&] {
        column_means(c, X, ri.n_total, true, mu64, muT, false, sharded(c) ? pro.sums : nullptr);
        means_done = true;
    };
    if (!fold_means) means_pass();

    // The whole device pipeline.  It runs OPTIMISTICALLY first: the whitening's eigenpairs come from two products and one
    // Rayleigh-Ritz step of the subspace iteration with no host round trip, whose residual verdict is read together with the
    // results; only if that verdict fails is the fit redone with the synchronous, residual-controlled iteration.
    std::vector<double> hC(size_t(nc) * dp), hmu(dp);
    DBuf X1T(c.dev, esz * size_t(std::max<int64_t>(n, 1)) * ncp);
    int64_t iters = 0;
    bool gram_fast = false;
    auto pipeline = [&](bool optimistic) -> bool {
        // whitening (ica.rs:189-208): left singular vectors / values of Xc^T == eigenpairs of Xc^T Xc
        DBuf C(c.dev, sizeof(double) * dp * dp), U(c.dev, sizeof(double) * dp * dp), lam(c.dev, sizeof(double) * dp), r3(c.dev, sizeof(double) * 3);
        // The covariance of fp32 data on the OPTIMISTIC run: exact bf16-piece products with fp32 accumulation over row chunks
        // (op_gram_split: 1.35 instead of 2.7 ms at 500000 x 512, 0.14 instead of 0.34 ms at 200000 x 256, good to ~2e-6 of the largest
        // entry) -- enough for a whitening whose kept eigenvalues lie within two decades, which the verdict below checks on the spectrum
        // found; wider spectra, the redo and fp64 data take the fp64-MFMA Gram matrix.  (From 256 features on: one full tile.)
        if (!means_done) {   // (fold_means: the first run of the pipeline)
            mu64 = DBuf(c.dev, sizeof(double) * dp);
            muT = DBuf(c.dev, esz * dp);
            gram_fast = optimistic && op_gram_split(c.dev, X.p, n, d, dp, X.ld, muT.p, C.f64(), dp, mu64.f64(), ri.n_total);
            if (gram_fast) means_done = true;
            else means_pass();
        } else {
            gram_fast = optimistic && dt == F32 && dp >= 256 && op_gram_split(c.dev, X.p, n, d, dp, X.ld, muT.p, C.f64(), dp);
        }
        if (!gram_fast) op_gemm_atb(c.dev, dt, X.p, X.ld, dp, muT.p, X.p, X.ld, dp, muT.p, n, C.f64(), dp, true);
        allreduce_f64(c, C.f64(), dp * dp, PETAL_SUM);
        // w_init and its symmetric decorrelation (ica.rs:210-216, 329) need nothing from the data: they go to the side stream and run
        // under the whitening (one workgroup, 40 us at 32 components, that the main chain used to wait for); queued BEHIND the
        // Gram launch on the host side, so that the device is already busy while the host prepares them
        dev_fork(c.dev, false);   // (nothing of the main stream's queue is needed: the prologue's draw was synchronised with)
        DBuf W(c.dev, sizeof(double) * nc * nc);   // (allocated while forked: a block no main-stream kernel can still be using)
        if (sharded(c)) {  // rank 0's draw, from the prologue's all-reduce
            dev_d2d(c.dev, W.p, pro.draw, W.bytes);
        } else {
            std::vector<double> h(size_t(nc) * nc);
            for (int64_t i = 0; i < nc * nc; ++i) h[i] = get_elem(w_init, dt, i);
            dev_h2d_async(c.dev, W.p, h.data(), W.bytes);  // (staged through the pinned ring: no host wait)
        }
        IcaStart start = ica_prepare(c, nc, W, mode);
        dev_fork_end(c.dev);
        DBuf Ckeep;  // (see pca_fit)
        if (dt == F64) { Ckeep = DBuf(c.dev, C.bytes); dev_d2d(c.dev, Ckeep.p, C.p, C.bytes); }
        const double vtol = dt == F32 ? 1e-12 : 3e-14;
        const bool topk = topk_eigh(c, C.f64(), d, dp, nc, U.f64(), lam.f64(), dt == F32 ? 1e-8 : 1e-5, optimistic ? r3.f64() : nullptr, vtol);  // only the first nc pairs are used below
        if (!topk) {
            dev_memset(c.dev, U.p, 0, U.bytes);
            dev_memset(c.dev, lam.p, 0, lam.bytes);
            op_eigh(c.dev, C.f64(), d, dp, U.f64(), dp, lam.f64(), dt == F32 ? 1e-8 : 1e-15, false, 0, nc);  // (closeness verdict over the nc pairs used)
        }
        if (dt == F64) {  // the whitening divides by sigma: below the Gram route's floor take the accurate route (see pca_fit)
            std::vector<double> hl(nc);
            dev_d2h(c.dev, hl.data(), lam.p, sizeof(double) * nc);
            dev_sync(c.dev);
            const bool low = hl[0] > 0 && hl[nc - 1] > 0 && n_total >= d && std::sqrt(hl[nc - 1] / hl[0]) < GRAM_ROUTE_FLOOR;
            if (agree_any(c, low)) {  // (rank-deficient data stay on the Gram route: see pca_fit)
                DBuf sg(c.dev, sizeof(double) * dp);
                if (accurate_small_svd(c, X, muT.p, Ckeep.f64(), d, U.f64(), sg.f64())) op_dvec(c.dev, 2, sg.f64(), lam.f64(), d, 0.0);
            }
        }
        // K^T = U[:, :nc] / sigma (ica.rs:190-203), zero padded, and K^T sqrt(n) (ica.rs:204-208), one launch
        DBuf KT(c.dev, sizeof(double) * dp * ncp), KTs(c.dev, sizeof(double) * dp * ncp);
        op_whiten_k(c.dev, U.f64(), dp, lam.f64(), dp, nc, ncp, std::sqrt(ri.n_total), KT.f64(), KTs.f64());
        op_gemm_xp(c.dev, dt, X.p, n, dp, X.ld, muT.p, KTs.f64(), ncp, ncp, nullptr, X1T.p, ncp, nullptr);

        dev_join(c.dev);   // the decorrelated w_init (side stream) meets the whitened data
        iters = ica_loop(c, dt, X1T.p, n, nc, ncp, ri.n_total, W, tol, max_iter, mode, &start);  // ica.rs:216

        // components = W K (ica.rs:217); everything the host reads comes back behind ONE synchronisation
        DBuf Cm(c.dev, sizeof(double) * nc * dp);
        op_dgemm(c.dev, false, true, nc, dp, nc, 1.0, W.f64(), nc, KT.f64(), ncp, 0.0, Cm.f64(), dp);
        double h3[3] = {0, 1, 0}, hl2[2] = {1, 1};
        {
            void* dsts[5] = {hC.data(), hmu.data(), h3, &hl2[0], &hl2[1]};
            const void* srcs[5] = {Cm.p, mu64.p, r3.p, lam.p, lam.f64() + (nc - 1)};
            const size_t lens[5] = {Cm.bytes, sizeof(double) * size_t(dp), (topk && optimistic) ? sizeof(h3) : 0,
                                    gram_fast ? sizeof(double) : 0, gram_fast ? sizeof(double) : 0};
            dev_d2h_multi(c.dev, 5, dsts, srcs, lens);
        }
        dev_sync(c.dev);
        // (the split-product covariance stands only where its ~2e-6 lam_1 is small beside the smallest kept eigenvalue)
        // (within ONE decade: at the edge the covariance's error is 2e-5 of the smallest kept eigenvalue, which the whitening divides
        // by -- the level of the crate's own f32 SVD whitening, ica.rs:189-208; the round-5 bound of two decades left 2e-4 there,
        // ADVICE round 5)
        const bool spectrum_ok = !gram_fast || (hl2[0] > 0 && hl2[1] >= 1e-1 * hl2[0]);
        return !agree_any(c, ((topk && optimistic) && !topk_verdict_ok(h3, vtol)) || !spectrum_ok);   // every rank redoes, or none
    };
    bool redone = false;
    if (!pipeline(true)) { redone = true; pipeline(false); }
    if (n_iter) *n_iter = iters;
    check_finite_w(hC);
    for (int64_t i = 0; i < nc; ++i)
        for (int64_t j = 0; j < d; ++j) put_elem(components, dt, i * d + j, hC[size_t(i) * dp + j]);
    for (int64_t j = 0; j < d; ++j) put_elem(means, dt, j, hmu[j]);
    if (y_out) {  // fit_transform (ica.rs:155-156): (components . Xc^T)^T = Xc . components^T
        DBuf CT(c.dev, sizeof(double) * dp * ncp);
        std::vector<double> hCT(size_t(dp) * ncp, 0.0);
        for (int64_t i = 0; i < nc; ++i)
            for (int64_t j = 0; j < dp; ++j) hCT[size_t(j) * ncp + i] = hC[size_t(i) * dp + j];
        dev_h2d(c.dev, CT.p, hCT.data(), CT.bytes);
        op_gemm_xp(c.dev, dt, X.p, n, dp, X.ld, muT.p, CT.f64(), ncp, ncp, nullptr, X1T.p, ncp, nullptr);
        emit(c, dt, X1T.p, n, nc, ncp, *y_out);
    }
    c.stats.ica_redo = redone ? 1 : 0;
    c.stats.ica_gram_split = gram_fast ? 1 : 0;
    c.stats.means_folded = (fold_means && gram_fast) ? 1 : 0;
    finish_stats(c, timer);
}

// ica_par (ica.rs:319-361) with the crate's nc x n component-major input
void ica_par(petal_ctx& c, const petal_matrix& x1, double tol, int64_t max_iter, int mode, const void* w_init,
             void* w_out, int64_t* n_iter) {
    Timer timer;
    dev_reset_timing(c.dev);
    c.stats = petal_stats{};
    check_matrix(x1, "input");
    const int dt = x1.dtype;
    const int64_t nc = x1.rows, n = x1.cols;
    if (nc == 0) { if (n_iter) *n_iter = 0; return; }
    petal_matrix xt = x1;  // view the transpose: sample-major n x nc
    xt.rows = n; xt.cols = nc; xt.row_stride = x1.col_stride; xt.col_stride = x1.row_stride;
    DevMat X1T = ingest(c, xt);
    DBuf keep;
    const void* xp = X1T.p;
    if (n == 0) { keep = DBuf(c.dev, 64); xp = keep.p; }
    DBuf W(c.dev, sizeof(double) * nc * nc);
    std::vector<double> h(size_t(nc) * nc);
    RankInfo ri;
    if (sharded(c)) {
        ShardPrologue pro = shard_prologue(c, X1T, 0, false, dt, w_init, nc * nc);
        ri = pro.ri;
        dev_d2d(c.dev, W.p, pro.draw, W.bytes);
        dev_sync(c.dev);  // (pro's buffer returns to the allocator; stream order keeps the copy ahead of any reuse anyway)
    } else {
        ri.n_total = double(n);
        for (int64_t i = 0; i < nc * nc; ++i) h[i] = get_elem(w_init, dt, i);
        dev_h2d(c.dev, W.p, h.data(), W.bytes);
    }
    const int64_t iters = ica_loop(c, dt, xp, n, nc, X1T.ld, ri.n_total, W, tol, max_iter, mode);
    dev_d2h(c.dev, h.data(), W.p, W.bytes);
    dev_sync(c.dev);
    check_finite_w(h);
    for (int64_t i = 0; i < nc * nc; ++i) put_elem(w_out, dt, i, h[i]);
    if (n_iter) *n_iter = iters;
    finish_stats(c, timer);
}

void symmetric_decorrelation(petal_ctx& c, const void* w, int64_t nc, int dtype, int mode, void* out) {
    if (nc <= 0) return;
    std::vector<double> h(size_t(nc) * nc);
    for (int64_t i = 0; i < nc * nc; ++i) h[i] = get_elem(w, dtype, i);
    DBuf Wi(c.dev, sizeof(double) * nc * nc), Wo(c.dev, sizeof(double) * nc * nc);
    dev_h2d(c.dev, Wi.p, h.data(), Wi.bytes);
    op_symdecorr(c.dev, nc, Wi.f64(), Wo.f64(), mode);
    dev_d2h(c.dev, h.data(), Wo.p, Wo.bytes);
    dev_sync(c.dev);
    check_finite_w(h);
    for (int64_t i = 0; i < nc * nc; ++i) put_elem(out, dtype, i, h[i]);
}

void logcosh(petal_ctx& c, const petal_matrix& x, const petal_matrix& g_out, void* gprime_out) {
    check_matrix(x, "input");
    const int dt = x.dtype;
    DevMat X = ingest(c, x);
    const int64_t r = X.n, cc = X.d;
    if (r == 0) return;
    DBuf G(c.dev, dtype_size(dt) * size_t(r) * X.dp), gp(c.dev, sizeof(double) * r);
    op_logcosh_rows(c.dev, dt, X.p, r, cc, X.ld, G.p, X.dp, gp.f64());
    std::vector<double> h(r);
    dev_d2h(c.dev, h.data(), gp.p, gp.bytes);
    emit(c, dt, G.p, r, cc, X.dp, g_out);
    dev_sync(c.dev);
    for (int64_t i = 0; i < r; ++i) put_elem(gprime_out, dt, i, cc ? h[i] / double(cc) : 0.0);  // ica.rs:391-396
}

void svd_flip(petal_ctx& c, const petal_matrix& u, const petal_matrix& vt) {
    check_matrix(u, "u");
    check_matrix(vt, "vt");
    if (u.dtype != vt.dtype) invalid_input("u and vt differ in dtype");
    const int dt = u.dtype;
    const int64_t m = std::min(u.cols, vt.rows);  // zip of U columns and V^T rows (pca.rs:819)
    if (m == 0 || u.rows == 0) return;
    DevMat U = ingest(c, u);
    std::vector<double> sg = flip_signs(c, dt, U.p, U.n, m, U.ld, 0);
    // apply on the host-visible matrices through a device round trip of U and V^T
    DevMat V = ingest(c, vt);
    std::vector<double> su(U.dp, 1.0);
    for (int64_t j = 0; j < m; ++j) su[j] = sg[j];
    DBuf dsu(c.dev, sizeof(double) * U.dp);
    dev_h2d(c.dev, dsu.p, su.data(), dsu.bytes);
    DBuf Uc(c.dev, dtype_size(dt) * size_t(U.n) * U.dp);
    dev_copy2d(c.dev, Uc.p, U.dp * dtype_size(dt), U.p, U.ld * dtype_size(dt), U.dp * dtype_size(dt), size_t(U.n), 2);
    op_scale_cols(c.dev, dt, Uc.p, U.n, U.d, U.dp, dsu.f64());
    emit(c, dt, Uc.p, U.n, U.d, U.dp, u);
    // rows of V^T: scale row j by sg[j] == scale columns of V; do it through P = diag(sg) V^T on the small side
    std::vector<double> hv(size_t(vt.rows) * vt.cols);
    DBuf Vc(c.dev, dtype_size(dt) * size_t(V.n) * V.dp);
    dev_copy2d(c.dev, Vc.p, V.dp * dtype_size(dt), V.p, V.ld * dtype_size(dt), V.dp * dtype_size(dt), size_t(V.n), 2);
    // a row scale is a column scale of the transpose; V^T is tiny (m x d): finish on the host copy
    std::vector<char> raw(dtype_size(dt) * size_t(V.n) * V.dp);
    dev_d2h(c.dev, raw.data(), Vc.p, raw.size());
    dev_sync(c.dev);
    for (int64_t j = 0; j < m; ++j)
        if (sg[j] < 0)
            for (int64_t i = 0; i < V.d; ++i) put_elem(raw.data(), dt, j * V.dp + i, -get_elem(raw.data(), dt, j * V.dp + i));
    dev_h2d(c.dev, Vc.p, raw.data(), raw.size());
    emit(c, dt, Vc.p, V.n, V.d, V.dp, vt);
    dev_sync(c.dev);
}

// ---------------------------------------------------------------------------------------------
// the two X-streaming GEMMs on their own (parity tests / roofline measurement)
void gemm_xp(petal_ctx& c, const petal_matrix& x, const void* mu, const void* p, int64_t N, const void* bias,
             const petal_matrix& z_out) {
    Timer timer;
    dev_reset_timing(c.dev);
    c.stats = petal_stats{};
    check_matrix(x, "input");
    const int dt = x.dtype;
    if (N < 0) invalid_input("negative parameter");
    if (x.rows == 0 || N == 0) { emit(c, dt, nullptr, x.rows, N, 0, z_out); return; }
    DevMat X = ingest(c, x);
    const int64_t K = X.dp, NP = round_up(N, 16);
    std::vector<double> hP(size_t(K) * NP, 0.0);
    for (int64_t i = 0; i < x.cols; ++i)
        for (int64_t j = 0; j < N; ++j) hP[size_t(i) * NP + j] = get_elem(p, dt, i * N + j);
    DBuf P(c.dev, sizeof(double) * K * NP), muT(c.dev, dtype_size(dt) * K), bT(c.dev, dtype_size(dt) * NP);
    dev_h2d(c.dev, P.p, hP.data(), P.bytes);
    std::vector<char> hmu(dtype_size(dt) * K, 0), hb(dtype_size(dt) * NP, 0);
    if (mu) std::memcpy(hmu.data(), mu, dtype_size(dt) * x.cols);
    if (bias) std::memcpy(hb.data(), bias, dtype_size(dt) * N);
    dev_h2d(c.dev, muT.p, hmu.data(), muT.bytes);
    dev_h2d(c.dev, bT.p, hb.data(), bT.bytes);
    DBuf Z(c.dev, dtype_size(dt) * size_t(X.n) * NP);
    c.stats.pass_flops = 2.0 * double(X.n) * double(x.cols) * double(N);
    c.stats.pass_bytes = double(dtype_size(dt)) * (double(X.n) * x.cols + double(X.n) * N + double(x.cols) * N);
    dev_set_tag(c.dev, TAG_XP);
    op_gemm_xp(c.dev, dt, X.p, X.n, K, X.ld, mu ? muT.p : nullptr, P.f64(), NP, NP, bias ? bT.p : nullptr, Z.p, NP, nullptr);
    dev_set_tag(c.dev, TAG_NONE);
    emit(c, dt, Z.p, X.n, N, NP, z_out);
    finish_stats(c, timer);
}

// one power iteration on its own: the fused pass where it exists, K1 + K2 otherwise (parity tests / roofline measurement)
void power_pass(petal_ctx& c, const petal_matrix& x, const void* mu, const void* p, int64_t N, double* y_out, const petal_matrix* z_out,
                int* fused_out) {
    Timer timer;
    dev_reset_timing(c.dev);
    c.stats = petal_stats{};
    check_matrix(x, "input");
    const int dt = x.dtype;
    if (N < 0) invalid_input("negative parameter");
    if (fused_out) *fused_out = 0;
    if (x.cols == 0 || N == 0) { if (z_out) emit(c, dt, nullptr, x.rows, N, 0, *z_out); return; }
    if (x.rows == 0) { std::memset(y_out, 0, sizeof(double) * x.cols * N); if (z_out) emit(c, dt, nullptr, 0, N, 0, *z_out); return; }
    DevMat X = ingest(c, x);
    const int64_t K = X.dp, NP = round_up(N, 16);
    std::vector<double> hP(size_t(K) * NP, 0.0);
    for (int64_t i = 0; i < x.cols; ++i)
        for (int64_t j = 0; j < N; ++j) hP[size_t(i) * NP + j] = get_elem(p, dt, i * N + j);
    DBuf P(c.dev, sizeof(double) * K * NP), muT(c.dev, dtype_size(dt) * K), Y(c.dev, sizeof(double) * K * NP);
    dev_h2d(c.dev, P.p, hP.data(), P.bytes);
    std::vector<char> hmu(dtype_size(dt) * K, 0);
    if (mu) std::memcpy(hmu.data(), mu, dtype_size(dt) * x.cols);
    dev_h2d(c.dev, muT.p, hmu.data(), muT.bytes);
    DBuf Z(c.dev, dtype_size(dt) * size_t(X.n) * NP);
    c.stats.pass_flops = 4.0 * double(X.n) * double(x.cols) * double(N);
    c.stats.pass_bytes = double(dtype_size(dt)) * (double(X.n) * x.cols + 2.0 * double(x.cols) * N);
    const void* mup = mu ? muT.p : nullptr;
    dev_set_tag(c.dev, TAG_POW);
    const bool fused = op_power_pass(c.dev, dt, X.p, X.n, K, X.ld, mup, P.f64(), NP, NP, z_out ? Z.p : nullptr, NP, Y.f64(), NP);
    dev_set_tag(c.dev, TAG_NONE);
    if (!fused) {
        dev_set_tag(c.dev, TAG_XP);
        op_gemm_xp(c.dev, dt, X.p, X.n, K, X.ld, mup, P.f64(), NP, NP, nullptr, Z.p, NP, nullptr);
        dev_set_tag(c.dev, TAG_ATB);
        op_gemm_atb(c.dev, dt, X.p, X.ld, K, mup, Z.p, NP, NP, nullptr, X.n, Y.f64(), NP);
        dev_set_tag(c.dev, TAG_NONE);
    }
    if (fused_out) *fused_out = fused ? 1 : 0;
    std::vector<double> h(size_t(K) * NP);
    dev_d2h(c.dev, h.data(), Y.p, Y.bytes);
    dev_sync(c.dev);
    for (int64_t i = 0; i < x.cols; ++i)
        for (int64_t j = 0; j < N; ++j) y_out[i * N + j] = h[size_t(i) * NP + j];
    if (z_out) emit(c, dt, Z.p, X.n, N, NP, *z_out);
    finish_stats(c, timer);
}

void gemm_atb(petal_ctx& c, const petal_matrix& a, const void* mu_a, const petal_matrix* b, const void* mu_b, double* c_out) {
    Timer timer;
    dev_reset_timing(c.dev);
    c.stats = petal_stats{};
    check_matrix(a, "a");
    if (b) {
        check_matrix(*b, "b");
        if (b->rows != a.rows || b->dtype != a.dtype) invalid_input("a and b differ in rows or dtype");
    }
    const int dt = a.dtype;
    const int64_t M = a.cols, N = b ? b->cols : a.cols;
    if (M == 0 || N == 0) return;
    if (a.rows == 0) { std::memset(c_out, 0, sizeof(double) * M * N); return; }
    DevMat A = ingest(c, a);
    DevMat B;
    if (b) B = ingest(c, *b);
    const DevMat& Bm = b ? B : A;
    DBuf muA(c.dev, dtype_size(dt) * A.dp), muB(c.dev, dtype_size(dt) * Bm.dp);
    std::vector<char> ha(dtype_size(dt) * A.dp, 0), hb(dtype_size(dt) * Bm.dp, 0);
    if (mu_a) std::memcpy(ha.data(), mu_a, dtype_size(dt) * M);
    if (mu_b) std::memcpy(hb.data(), mu_b, dtype_size(dt) * N);
    dev_h2d(c.dev, muA.p, ha.data(), muA.bytes);
    dev_h2d(c.dev, muB.p, hb.data(), muB.bytes);
    DBuf C(c.dev, sizeof(double) * A.dp * Bm.dp);
    c.stats.pass_flops = 2.0 * double(A.n) * double(M) * double(N);
    c.stats.pass_bytes = double(dtype_size(dt)) * (double(A.n) * M + double(A.n) * N + double(M) * N);
    dev_set_tag(c.dev, TAG_ATB);
    // (test hook: PETAL_GRAM_SPLIT=1 sends a Gram matrix -- b == NULL, both sides centred alike -- to the split-product Gram kernels of
    // the FastICA whitening, so that the parity tests reach them on exact-integer data)
    const bool gram_hook = !b && dt == F32 && ((mu_a == nullptr) == (mu_b == nullptr)) && dev_option(c.dev, OPT_GRAM_SPLIT_HOOK) != 0 &&
                           (!mu_a || std::memcmp(ha.data(), hb.data(), ha.size()) == 0);
    if (!(gram_hook && op_gram_split(c.dev, A.p, A.n, M, A.dp, A.ld, mu_a ? muA.p : nullptr, C.f64(), A.dp)))
        op_gemm_atb(c.dev, dt, A.p, A.ld, A.dp, mu_a ? muA.p : nullptr, Bm.p, Bm.ld, Bm.dp, mu_b ? muB.p : nullptr, A.n, C.f64(), Bm.dp);
    dev_set_tag(c.dev, TAG_NONE);
    std::vector<double> h(size_t(A.dp) * Bm.dp);
    dev_d2h(c.dev, h.data(), C.p, C.bytes);
    dev_sync(c.dev);
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) c_out[i * N + j] = h[size_t(i) * Bm.dp + j];
    finish_stats(c, timer);
}

}  // namespace petal
