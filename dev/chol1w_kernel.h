// k_chol_rt<NB>: G = R^T R for an order L <= 16 NB in the registers of ONE wave, "RT form" output (diagonal 16 x 16 blocks: T_JJ = R_JJ^-1,
// blocks above them: R, zeros below).  Block (I, J), I <= J, of the working matrix sits in the accumulator layout of
// v_mfma_f64_16x16x4_f64 -- register m of lane (g, c) = (lane >> 4, lane & 15) is element (g + 4 m, c) of the block -- which is at once
//   * the layout the 16 x 16 diagonal factorisation wants (four rows of a column per lane, exchanges through the LDS crossbar),
//   * the B operand of an MFMA for the block itself and the A operand for its TRANSPOSE, k-slots declared as k' = (l >> 4) + 4 r,
// so the panel R_JK = T_JJ^T S_JK and the trailing update S_KM -= R_JK^T R_JM take the registers as they are: no barrier, no LDS
// image of the matrix, one 2-KB LDS transpose per block row (T_JJ^T, which the elimination of [S_JJ | I] leaves, -> T_JJ).
// The blocks not yet factored are kept NEGATED (N = -S), so the trailing update is a plain accumulation N_KM += R_JK^T R_JM.
// Only what the next diagonal block needs is on the dependent chain: R_J,J+1 and N_J+1,J+1; the rest of row J's panel and trailing
// update is issued between the pivots of block row J + 1, whose chain (reciprocal square root -> scale -> update) leaves the
// matrix pipe idle.
__device__ __forceinline__ constexpr int chol_rt_idx(int NB, int I, int J) { return I * NB - (I * (I - 1)) / 2 + (J - I); }
// compile-time loops: every block index below is a constant expression, so the blocks are registers (a #pragma unroll that the
// optimizer gives up on -- it did, on the sliced list of deferred blocks -- turns the whole array into scratch memory)
template <int I0, int I1, class F>
__device__ __forceinline__ void chol_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        chol_static_for<I0 + 1, I1>(f);
    }
}
#define CHOL_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
template <int NB, int NEWTON, bool INTERLEAVE = true>
__global__ __launch_bounds__(64) void k_chol_rt(const double* __restrict__ G, int L, long ldg, double* __restrict__ T, long ldt, double rel_tol,
                                                int* __restrict__ ndead_out, int ncount, long long* __restrict__ cyc) {
    __shared__ double sm[16 * 17];
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const long long t_start = cyc ? clock64() : 0;
    cf64x4 S[NB * (NB + 1) / 2];
    double gd[NB];
    // (clamped addresses and a select instead of predicated loads: no exec-mask branches in the load phase)
    chol_static_for<0, NB>([&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        gd[I] = G[(long)min(16 * I + c, L - 1) * (ldg + 1)];
        gd[I] = (16 * I + c < L) ? gd[I] : 0.0;
        chol_static_for<I, NB>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int row = 16 * I + g + 4 * m, col = 16 * J + c;
                const int rr = (I == J && row > col) ? col : row, cc = (I == J && row > col) ? row : col;   // the upper triangle only
                const double v = G[(long)min(rr, L - 1) * ldg + min(cc, L - 1)];
                S[chol_rt_idx(NB, I, J)][m] = (row < L && col < L) ? -v : 0.0;
            }
        });
    });
    chol_static_for<1, NB>([&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        chol_static_for<0, I>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
#pragma unroll
            for (int m = 0; m < 4; ++m) T[(long)(16 * I + g + 4 * m) * ldt + 16 * J + c] = 0.0;
        });
    });
    if (cyc && lane == 0) cyc[0] = clock64() - t_start;
    int cdead = 0;
    const int nlim = min(L, ncount);
    cf64x4 negA_prev = cf64x4{0.0, 0.0, 0.0, 0.0};
    cf64x4 pacc[4];
    // What block row Jp leaves for later -- four MFMAs per block, in dependency order:
    //   b <  q              : its panel blocks K = Jp + 2 + b   (R_JpK = (-T)^T N_JpK, the A operand kept in negA_prev)
    //   b <  2 q            : the rest of block row Jp + 1, M = Jp + 2 + (b - q)
    //   b <  2 q + q(q+1)/2 : the trailing blocks (K, M), Jp + 2 <= K <= M              (q = NB - Jp - 2)
    // The MFMAs are issued ONE slot at a time between the pieces of the next block row's pivot chain: a wave issues in order, so a
    // run of dependent MFMAs (the four k-steps of a block) would hold the chain's VALU instructions behind 64 cycles each.  Blocks
    // are walked in groups of GS, k-step by k-step, so that the MFMAs of one slot never depend on each other.
    auto deferred_mfma = [&](auto Jpc, auto GSc, auto ec) {
        constexpr int Jp = decltype(Jpc)::value, GS = decltype(GSc)::value, e = decltype(ec)::value, q = NB - Jp - 2;
        constexpr int nblk = 2 * q + q * (q + 1) / 2;
        constexpr int b = GS * (e / (4 * GS)) + e % GS, r = (e % (4 * GS)) / GS, mem = e % GS;
        if constexpr (b < nblk) {
            if constexpr (b < q) {
                constexpr int K = Jp + 2 + b;
                if constexpr (r == 0) pacc[mem] = cf64x4{0.0, 0.0, 0.0, 0.0};
                pacc[mem] = CHOL_MFMA(negA_prev[r], S[chol_rt_idx(NB, Jp, K)][r], pacc[mem]);
                if constexpr (r == 3) {
                    S[chol_rt_idx(NB, Jp, K)] = pacc[mem];
#pragma unroll
                    for (int m = 0; m < 4; ++m) T[(long)(16 * Jp + g + 4 * m) * ldt + 16 * K + c] = pacc[mem][m];
                }
            } else if constexpr (b < 2 * q) {
                constexpr int Mb = Jp + 2 + (b - q);
                S[chol_rt_idx(NB, Jp + 1, Mb)] = CHOL_MFMA(S[chol_rt_idx(NB, Jp, Jp + 1)][r], S[chol_rt_idx(NB, Jp, Mb)][r], S[chol_rt_idx(NB, Jp + 1, Mb)]);
            } else {
                constexpr int t = b - 2 * q;
                constexpr int K = [] { int k = Jp + 2, u = t; while (u >= NB - k) { u -= NB - k; ++k; } return k; }();
                constexpr int Mb = [] { int k = Jp + 2, u = t; while (u >= NB - k) { u -= NB - k; ++k; } return k + u; }();
                S[chol_rt_idx(NB, K, Mb)] = CHOL_MFMA(S[chol_rt_idx(NB, Jp, K)][r], S[chol_rt_idx(NB, Jp, Mb)][r], S[chol_rt_idx(NB, K, Mb)]);
            }
        }
    };
    chol_static_for<0, NB>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        const long long t_row = cyc ? clock64() : 0;
        const int jb = 16 * J;
        cf64x4 D = -S[chol_rt_idx(NB, J, J)], Id;
#pragma unroll
        for (int m = 0; m < 4; ++m) Id[m] = (g + 4 * m == c) ? 1.0 : 0.0;
        // pivot i is accepted when it exceeds rel_tol x the original diagonal entry (which must be positive): one threshold per column
        const double thr = (gd[J] > 0.0) ? rel_tol * gd[J] : __builtin_inf();
        constexpr int qp = NB - (J - 1) - 2, Jprev = J > 0 ? J - 1 : 0;
        constexpr int nblk = (INTERLEAVE && J > 0 && qp > 0) ? 2 * qp + qp * (qp + 1) / 2 : 0;
        constexpr int GS = nblk * 4 <= 64 ? 1 : (nblk * 4 <= 128 ? 2 : (nblk * 4 <= 192 ? 3 : 4));   // MFMAs per slot (64 slots per block row)
        constexpr int ngrp = (nblk + GS - 1) / GS, nmf = ngrp * 4 * GS;    // (list length with the last group padded)
        constexpr int per_slot = (nmf + 63) / 64;
        chol_static_for<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value, mi = i >> 2, gi = i & 3, src = 16 * gi;
            auto slot = [&](auto sc) {
                if constexpr (nblk > 0) {
                    constexpr int sidx = 4 * i + decltype(sc)::value;
                    constexpr int e0 = sidx * per_slot < nmf ? sidx * per_slot : nmf, e1 = (sidx + 1) * per_slot < nmf ? (sidx + 1) * per_slot : nmf;
                    __builtin_amdgcn_sched_barrier(0);
                    chol_static_for<e0, e1>([&](auto ec) { deferred_mfma(std::integral_constant<int, Jprev>{}, std::integral_constant<int, GS>{}, ec); });
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            const double dii = readlane_d(D[mi], src + i);
            const double thi = readlane_d(thr, i);
            const bool ok = dii > thi;
            // the UNSCALED pivot row travels while the reciprocal square root is formed
            const double su_c = bperm_d(D[mi], src + c), su_ci = bperm_d(Id[mi], src + c);
            slot(std::integral_constant<int, 0>{});
            double y = __builtin_amdgcn_rsq(dii);
            const double e1n = fma(-dii * y, y, 1.0);
            slot(std::integral_constant<int, 1>{});
            y = fma(0.5 * y, e1n, y);
            if constexpr (NEWTON > 1) { const double e2n = fma(-dii * y, y, 1.0); y = fma(0.5 * y, e2n, y); }
            const double inv = ok ? y : 0.0;
            const double ninv2 = -(inv * inv);
            slot(std::integral_constant<int, 2>{});
            // S[k][i], the multiplier of row k, sits in column i of the (symmetric) block: lane i of the 16-lane row that holds row k
            // -- a DPP row broadcast instead of an exchange through the LDS crossbar.  w = -S[i][c] / d_i: the update is S[k][c] += S[k][i] w
            const double w = su_c * ninv2, wi = su_ci * ninv2;
            {   // the register that holds row i: lanes g == gi scale it, g > gi update, g < gi are finished
                double bk = __builtin_amdgcn_update_dpp(0.0, D[mi], 0x150 + i, 0xf, 0xf, false);
                bk = (g > gi) ? bk : 0.0;
                const double v = (g == gi) ? inv : 1.0;
                Id[mi] = fma(bk, wi, Id[mi] * v);
                D[mi] = fma(bk, w, D[mi] * v);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m <= mi) continue;
                const double bk = __builtin_amdgcn_update_dpp(0.0, D[m], 0x150 + i, 0xf, 0xf, false);
                Id[m] = fma(bk, wi, Id[m]);
                D[m] = fma(bk, w, D[m]);
            }
            slot(std::integral_constant<int, 3>{});
            if (INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        });
        // T_JJ^T (lower triangular, in Id) -> T_JJ through LDS
#pragma unroll
        for (int m = 0; m < 4; ++m) sm[(g + 4 * m) * 17 + c] = Id[m];
        cf64x4 A;
#pragma unroll
        for (int r = 0; r < 4; ++r) A[r] = sm[c * 17 + g + 4 * r];
        {   // a dropped column has T_jj = 0
            bool dead = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) dead = dead || (g + 4 * r == c && jb + c < nlim && !(A[r] > 0.0));
            cdead += __builtin_popcountll(__ballot(dead));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(long)(jb + g + 4 * r) * ldt + jb + c] = A[r];
        const cf64x4 negA = -A;
        const long long t_diag = cyc ? clock64() : 0;
        // on the chain: R_J,J+1 = T_JJ^T S_J,J+1 = (-T_JJ)^T N_J,J+1 and N_J+1,J+1 += R_J,J+1^T R_J,J+1
        if constexpr (J + 1 < NB) {
            cf64x4 acc = cf64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = CHOL_MFMA(negA[r], S[chol_rt_idx(NB, J, J + 1)][r], acc);
            S[chol_rt_idx(NB, J, J + 1)] = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) S[chol_rt_idx(NB, J + 1, J + 1)] = CHOL_MFMA(acc[r], acc[r], S[chol_rt_idx(NB, J + 1, J + 1)]);
#pragma unroll
            for (int m = 0; m < 4; ++m) T[(long)(jb + g + 4 * m) * ldt + 16 * (J + 1) + c] = acc[m];
        }
        negA_prev = negA;
        if constexpr (!INTERLEAVE && NB - J - 2 > 0) {
            constexpr int q = NB - J - 2;
            chol_static_for<0, 4 * (2 * q + q * (q + 1) / 2)>([&](auto ec) { deferred_mfma(Jc, std::integral_constant<int, 1>{}, ec); });
        }
        if (cyc && lane == 0) { const long long t = clock64(); cyc[1 + 2 * J] = t_diag - t_row; cyc[2 + 2 * J] = t - t_diag; }
    });
    if (lane == 0 && ndead_out && cdead > *ndead_out) *ndead_out = cdead;
    if (cyc && lane == 0) cyc[63] = clock64() - t_start;
}
