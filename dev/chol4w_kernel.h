// k_chol_rt4<NB>: the RT-form Cholesky on FOUR waves, one per SIMD.  A wave cannot overlap its own MFMAs with its own VALU work (measured
// on k_chol_rt: interleaving them between the pivots gains nothing), so the three other matrix pipes of the CU take the panel and the
// trailing update while wave 0 runs nothing but the pivot chain:
//   wave 0   factors the diagonal block D_J (k_chol_rt's 16-pivot chain: DPP row-broadcast multipliers, [D | I] -> T_JJ^T), publishes
//            -T_JJ, forms R_J,J+1 and N_J+1,J+1 += R_J,J+1^T R_J,J+1 from the two blocks the owner of column J + 1 handed over, goes on;
//   wave h   (1..3) owns whole block COLUMNS M >= 2 of the working matrix in registers (accumulator layout, negated: N = -S): after
//            -T_JJ is out it forms its panel blocks R_JM, publishes them, and after every panel is out updates its blocks
//            N_KM += R_JK^T R_JM, the two that wave 0 needs next (column J + 2) first.
// Two workgroup barriers per block row; what crosses them goes through LDS in register order (lane-contiguous 8-byte words: no bank
// conflicts, no layout change: a block's registers are the B operand for the block and the A operand for its transpose).
template <int NB>
__device__ __forceinline__ constexpr int chol4_owner(int M) {   // columns dealt out from the last (longest) one, boustrophedon over the 3 helpers
    const int idx = NB - 1 - M, round = idx / 3, pos = idx % 3;
    return (round & 1) ? 3 - pos : 1 + pos;
}
// (not __syncthreads(): that also waits for the global stores of the factor -- vmcnt -- a microsecond per barrier; only LDS crosses here)
__device__ __forceinline__ void chol4_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void chol4_put(double* slot, const cf64x4& v, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) slot[r * 64 + lane] = v[r];
}
__device__ __forceinline__ cf64x4 chol4_get(const double* slot, int lane) {
    cf64x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = slot[r * 64 + lane];
    return v;
}
// block (I, J) of G, negated, upper triangle only, zero beyond L (clamped addresses + select: no exec-mask branches)
__device__ __forceinline__ cf64x4 chol4_load(const double* __restrict__ G, int L, int64_t ldg, int I, int J, int g, int c) {
    cf64x4 v;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = 16 * I + g + 4 * m, col = 16 * J + c;
        const int rr = (I == J && row > col) ? col : row, cc = (I == J && row > col) ? row : col;
        const double x = G[(int64_t)min(rr, L - 1) * ldg + min(cc, L - 1)];
        v[m] = (row < L && col < L) ? -x : 0.0;
    }
    return v;
}
__device__ __forceinline__ void chol4_store(double* __restrict__ T, int64_t ldt, int I, int J, int g, int c, const cf64x4& v) {
#pragma unroll
    for (int m = 0; m < 4; ++m) T[(int64_t)(16 * I + g + 4 * m) * ldt + 16 * J + c] = v[m];
}
#define CHOL4_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// helper wave H: all of its work, unrolled over the block rows (its blocks are registers)
template <int NB, int H>
__device__ __forceinline__ void chol4_helper(const double* __restrict__ G, int L, int64_t ldg, double* __restrict__ T, int64_t ldt,
                                             double* s_tjj, double* s_hand, double* s_panel, int lane) {
    const int c = lane & 15, g = lane >> 4;
    // S[M][K]: block (K, M) of an owned column M (K <= M); columns that are not mine stay unused (and cost nothing)
    cf64x4 S[NB][NB];
    chol_static_for<2, NB>([&](auto Mc) {
        constexpr int M = decltype(Mc)::value;
        if constexpr (chol4_owner<NB>(M) == H)
            chol_static_for<0, M + 1>([&](auto Kc) { constexpr int K = decltype(Kc)::value; S[M][K] = chol4_load(G, L, ldg, K, M, g, c); });
    });
    // zeros below the block diagonal of the output: rows I = H, H + 3, ...
    chol_static_for<1, NB>([&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        if constexpr (I % 3 == H % 3)
            chol_static_for<0, I>([&](auto Jc) { chol4_store(T, ldt, I, decltype(Jc)::value, g, c, cf64x4{0.0, 0.0, 0.0, 0.0}); });
    });
    chol_static_for<0, NB>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        // hand column J + 1's two leading blocks to wave 0 (complete through block row J - 1; column 1 is wave 0's own)
        if constexpr (J >= 1 && J + 1 < NB) {
            if constexpr (chol4_owner<NB>(J + 1) == H) {
                chol4_put(s_hand, S[J + 1][J], lane);
                chol4_put(s_hand + 256, S[J + 1][J + 1], lane);
            }
        }
        chol4_sync();   // barrier 1: -T_JJ is out
        if constexpr (J + 2 < NB) {
            const cf64x4 negA = chol4_get(s_tjj, lane);
            chol_static_for<J + 2, NB>([&](auto Mc) {
                constexpr int M = decltype(Mc)::value;
                if constexpr (chol4_owner<NB>(M) == H) {
                    cf64x4 acc = cf64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = CHOL4_MFMA(negA[r], S[M][J][r], acc);
                    S[M][J] = acc;   // R_JM
                    chol4_put(s_panel + ((J & 1) * NB + M) * 256, acc, lane);
                    chol4_store(T, ldt, J, M, g, c, acc);
                }
            });
        }
        chol4_sync();   // barrier 2: every panel block of row J is out
        if constexpr (J + 2 < NB) {
            // N_KM += R_JK^T R_JM for my columns M >= J + 2, K = J + 1 .. M; R_JK read once per K
            chol_static_for<J + 1, NB>([&](auto Kc) {
                constexpr int K = decltype(Kc)::value;
                constexpr bool any = [] { for (int M = (K > J + 2 ? K : J + 2); M < NB; ++M) if (chol4_owner<NB>(M) == H) return true; return false; }();
                if constexpr (any) {
                    const cf64x4 rk = chol4_get(s_panel + ((J & 1) * NB + K) * 256, lane);
                    chol_static_for<(K > J + 2 ? K : J + 2), NB>([&](auto Mc) {
                        constexpr int M = decltype(Mc)::value;
                        if constexpr (chol4_owner<NB>(M) == H) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) S[M][K] = CHOL4_MFMA(rk[r], S[M][J][r], S[M][K]);
                        }
                    });
                }
            });
        }
    });
}

template <int NB>
__global__ __launch_bounds__(256) void k_chol_rt4(const double* __restrict__ G, int L, int64_t ldg, double* __restrict__ T, int64_t ldt,
                                                  double rel_tol, int* __restrict__ ndead_out, int ncount) {
    __shared__ double s_tr[16 * 17];            // wave 0's transpose scratch
    __shared__ double s_tjj[256];               // -T_JJ in register order
    __shared__ double s_hand[512];              // the two blocks handed to wave 0
    __shared__ double s_panel[2 * NB * 256];    // R_JM of block row J, double buffered
    __shared__ double s_thr[16 * NB];           // the acceptance threshold of every pivot: rel_tol x the original diagonal
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave == 1) { chol4_helper<NB, 1>(G, L, ldg, T, ldt, s_tjj, s_hand, s_panel, lane); return; }
    if (wave == 2) { chol4_helper<NB, 2>(G, L, ldg, T, ldt, s_tjj, s_hand, s_panel, lane); return; }
    if (wave == 3) { chol4_helper<NB, 3>(G, L, ldg, T, ldt, s_tjj, s_hand, s_panel, lane); return; }
    const int c = lane & 15, g = lane >> 4;
    cf64x4 D = -chol4_load(G, L, ldg, 0, 0, g, c);
    cf64x4 Off = cf64x4{0.0, 0.0, 0.0, 0.0}, Next = cf64x4{0.0, 0.0, 0.0, 0.0};
    if (NB > 1) { Off = chol4_load(G, L, ldg, 0, 1, g, c); Next = chol4_load(G, L, ldg, 1, 1, g, c); }
    int cdead = 0;
    const int nlim = min(L, ncount);
    for (int e = lane; e < 16 * NB; e += 64) {
        const double gdj = G[(int64_t)min(e, L - 1) * (ldg + 1)];
        s_thr[e] = (e < L && gdj > 0.0) ? rel_tol * gdj : __builtin_inf();
    }
    for (int J = 0; J < NB; ++J) {
        const int jb = 16 * J;
        const double thr = s_thr[jb + c];       // (wave 0's own writes: LDS operations of a wave stay in order)
        cf64x4 Id;
#pragma unroll
        for (int m = 0; m < 4; ++m) Id[m] = (g + 4 * m == c) ? 1.0 : 0.0;
        chol_static_for<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value, mi = i >> 2, gi = i & 3, src = 16 * gi;
            const double dii = readlane_d(D[mi], src + i);
            const double thi = readlane_d(thr, i);
            const bool ok = dii > thi;
            const double su_c = bperm_d(D[mi], src + c), su_ci = bperm_d(Id[mi], src + c);
            double y = __builtin_amdgcn_rsq(dii);
            const double en = fma(-dii * y, y, 1.0);
            y = fma(0.5 * y, en, y);
            const double inv = ok ? y : 0.0;
            const double ninv2 = -(inv * inv);
            const double w = su_c * ninv2, wi = su_ci * ninv2;
            {
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
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int m = 0; m < 4; ++m) s_tr[(g + 4 * m) * 17 + c] = Id[m];
        cf64x4 A;
#pragma unroll
        for (int r = 0; r < 4; ++r) A[r] = s_tr[c * 17 + g + 4 * r];
        const cf64x4 negA = -A;
        chol4_put(s_tjj, negA, lane);
        {
            bool dead = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) dead = dead || (g + 4 * r == c && jb + c < nlim && !(A[r] > 0.0));
            cdead += __builtin_popcountll(__ballot(dead));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(int64_t)(jb + g + 4 * r) * ldt + jb + c] = A[r];
        chol4_sync();   // barrier 1
        cf64x4 R = cf64x4{0.0, 0.0, 0.0, 0.0};
        if (J + 1 < NB) {
            if (J > 0) { Off = chol4_get(s_hand, lane); Next = chol4_get(s_hand + 256, lane); }
#pragma unroll
            for (int r = 0; r < 4; ++r) R = CHOL4_MFMA(negA[r], Off[r], R);
            chol4_put(s_panel + ((J & 1) * NB + J + 1) * 256, R, lane);
#pragma unroll
            for (int m = 0; m < 4; ++m) T[(int64_t)(jb + g + 4 * m) * ldt + jb + 16 + c] = R[m];
        }
        chol4_sync();   // barrier 2
        if (J + 1 < NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Next = CHOL4_MFMA(R[r], R[r], Next);
            D = -Next;
        }
    }
    if (lane == 0 && ndead_out && cdead > *ndead_out) *ndead_out = cdead;
}
#undef CHOL4_MFMA
