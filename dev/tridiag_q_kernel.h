// dev (round 6, NOT in the library): Householder tridiagonalisation with the matrix in REGISTERS -- built, parity-green
// (tests/test_gpu_eigh.py 58 / 58) and no faster than k_tridiag_r: 78.2 us at l = 74 (k_tridiag_r: 79), 629 us at l = 138 (k_tridiag_w: 254;
// 251 VGPRs + 448 B of scratch at two waves per SIMD).  A step is a chain of ~25 dependent stages of 100 cycles (LDS round trip ->
// squares -> 8-lane sum -> rsq / rcp with their Newton steps -> scale -> 5-deep FMA chains -> 8-lane sums -> LDS -> barrier -> ...), not
// LDS bandwidth: taking 75 % of the LDS instructions out of it changes nothing.  EXPERIMENTS.md round 6.
// ---- k_tridiag_q (round 6): the same reduction with the MATRIX IN REGISTERS.  A step of k_tridiag_r / k_tridiag_w moves the whole
// trailing block through LDS three times (read for p = tau A22 v, read + write for the rank-2 update): 84 LDS instructions per
// thread and step at l = 74, 670 per workgroup -- at the 128 B/clk of the LDS that IS the step (2600 cycles).  Here thread
// (g, sub) = (tid >> 3, tid & 7) owns rows g + 64 r and the absolute column pairs {2 sub, 2 sub + 1} + 16 s of A in registers for the
// whole kernel (f64x2 a[R][S]: 40 VGPRs at l <= 80, 108 at l <= 144); only vectors cross the LDS: row k (published by its owner group
// at the end of step k - 1, behind that step's second barrier) and p.  Every group computes sigma, beta, tau and p^T v redundantly and
// bit-identically as before.  Dead rows / columns (index <= k) need no guards in the update: their v and w are exact zeros.  The step
// is a template on the first live column slot S0 (and row slot S0 / 4): the block shrinks by whole slots.  Reflector k, d_k, e_k,
// tau_k and v_k . v_{k-1} go straight to global memory from one group (the barriers here wait for LDS only, not for those stores).
template <int S, int R, int S0>
__device__ __forceinline__ void triq_step(f64x2 (&a)[R][S], f64x2 (&vprev)[S], int L, int k, double* __restrict__ rowbuf,
                                          double* __restrict__ sp, double* __restrict__ dd, double* __restrict__ ee,
                                          double* __restrict__ HV, double* __restrict__ tau, double* __restrict__ gg) {
    constexpr int R0 = S0 >> 2, LB = (16 * S + 16 > 64 * R) ? 16 * S + 16 : 64 * R;   // (row k is read by column AND by row index)
    const int tid = threadIdx.x, g = tid >> 3, sub = tid & 7;
    const double* rb = rowbuf + (k & 1) * LB;
    // ---- reflector k from row k ----
    const double alpha = rb[k + 1], dk = rb[k];
    f64x2 x[S];
    double sq0 = 0, sq1 = 0;
#pragma unroll
    for (int s = S0; s < S; ++s) {
        const int c = 2 * sub + 16 * s;
        x[s] = *reinterpret_cast<const f64x2*>(rb + c);
        if (s == S0) {                                            // (only the leading live slot holds columns that have left the block)
            x[s].x = c > k + 1 ? x[s].x : 0.0;
            x[s].y = c + 1 > k + 1 ? x[s].y : 0.0;
        }
        sq0 = fma(x[s].x, x[s].x, sq0);
        sq1 = fma(x[s].y, x[s].y, sq1);
    }
    double xi[R];
#pragma unroll
    for (int r = R0; r < R; ++r) xi[r] = rb[g + 64 * r];
    const double sigma = oct_sum_f64(sq0 + sq1);
    double beta = alpha, tk = 0.0, scale = 0.0;
    if (sigma > 0.0) {                                            // (uniform: every thread holds the same bits)
        const double n2 = fma(alpha, alpha, sigma);
        double rs = __builtin_amdgcn_rsq(n2);
        double rc = __builtin_amdgcn_rcp(fma(n2, rs, fabs(alpha)));
        rs = rs * fma(-0.5 * n2 * rs, rs, 1.5);
        rs = rs * fma(-0.5 * n2 * rs, rs, 1.5);
        const double nrm = n2 * rs;
        beta = -copysign(nrm, alpha);
        const double den = fabs(alpha) + nrm;
        rc = fma(fma(-den, rc, 1.0), rc, rc);
        rc = fma(fma(-den, rc, 1.0), rc, rc);
        scale = copysign(rc, alpha);
        tk = den * rs;
    }
    f64x2 v[S];
#pragma unroll
    for (int s = S0; s < S; ++s) {
        const int c = 2 * sub + 16 * s;
        v[s].x = (s == S0 && c == k + 1) ? 1.0 : x[s].x * scale;
        v[s].y = (s == S0 && c + 1 == k + 1) ? 1.0 : x[s].y * scale;
    }
    double vi[R];
#pragma unroll
    for (int r = R0; r < R; ++r) {
        const int i = g + 64 * r;
        vi[r] = i > k + 1 ? xi[r] * scale : (i == k + 1 ? 1.0 : 0.0);
    }
    if (g == 1) {                                                 // the step's outputs (global, fire and forget)
        double* hv = HV + (int64_t)k * L;
#pragma unroll
        for (int s = S0; s < S; ++s) {
            const int c = 2 * sub + 16 * s;
            if (c < L) hv[c] = v[s].x;
            if (c + 1 < L) hv[c + 1] = v[s].y;
        }
        if (sub == 0) { dd[k] = dk; ee[k] = beta; tau[k] = tk; }
    }
    {   // v_k . v_{k-1} (lets the back-transformation apply two reflectors per reduction round)
        double g0 = 0, g1 = 0;
#pragma unroll
        for (int s = S0; s < S; ++s) { g0 = fma(v[s].x, vprev[s].x, g0); g1 = fma(v[s].y, vprev[s].y, g1); vprev[s] = v[s]; }
        const double gs = oct_sum_f64(g0 + g1);
        if (tid == 0) gg[k] = k >= 1 ? gs : 0.0;
    }
    if (tk != 0.0) {                                              // (uniform)
        // ---- p = tau A22 v ----
#pragma unroll
        for (int r = R0; r < R; ++r) {
            double a0 = 0, a1 = 0;
#pragma unroll
            for (int s = S0; s < S; ++s) { a0 = fma(a[r][s].x, v[s].x, a0); a1 = fma(a[r][s].y, v[s].y, a1); }
            const double ps = oct_sum_f64(a0 + a1);
            const int i = g + 64 * r;
            if (sub == 0) sp[i] = i > k ? tk * ps : 0.0;
        }
    }
    lds_barrier();
    if (tk != 0.0) {
        f64x2 w[S];
        double pv0 = 0, pv1 = 0;
#pragma unroll
        for (int s = S0; s < S; ++s) {
            w[s] = *reinterpret_cast<const f64x2*>(sp + 2 * sub + 16 * s);
            pv0 = fma(w[s].x, v[s].x, pv0);
            pv1 = fma(w[s].y, v[s].y, pv1);
        }
        double wi[R];
#pragma unroll
        for (int r = R0; r < R; ++r) wi[r] = sp[g + 64 * r];
        const double K = -0.5 * tk * oct_sum_f64(pv0 + pv1);
#pragma unroll
        for (int s = S0; s < S; ++s) { w[s].x = fma(K, v[s].x, w[s].x); w[s].y = fma(K, v[s].y, w[s].y); }
#pragma unroll
        for (int r = R0; r < R; ++r) {                            // A22 -= v w^T + w v^T
            const double wr = fma(K, vi[r], wi[r]);
#pragma unroll
            for (int s = S0; s < S; ++s) {
                a[r][s].x = fma(-vi[r], w[s].x, fma(-wr, v[s].x, a[r][s].x));
                a[r][s].y = fma(-vi[r], w[s].y, fma(-wr, v[s].y, a[r][s].y));
            }
        }
    }
    // row k + 1 (final once this update is in) for the next step: its owner group publishes it into the other buffer
    if (g == ((k + 1) & 63)) {
        double* nb = rowbuf + ((k + 1) & 1) * LB;
#pragma unroll
        for (int s = S0; s < S; ++s) *reinterpret_cast<f64x2*>(nb + 2 * sub + 16 * s) = a[R0][s];   // ((k + 1) >> 6 == R0 on this rung)
    }
    lds_barrier();
}
template <int S, int R>  // L <= 16 S, L <= 64 R
__global__ __launch_bounds__(512) void k_tridiag_q(const double* __restrict__ A, int L, int64_t lda, double* __restrict__ dd,
                                                   double* __restrict__ ee, double* __restrict__ HV,
                                                   double* __restrict__ tau, double* __restrict__ gg,
                                                   int* __restrict__ flag, double* __restrict__ V, int64_t ldv, int Lz, int reset_flag) {
    constexpr int LB = (16 * S + 16 > 64 * R) ? 16 * S + 16 : 64 * R;
    __shared__ __attribute__((aligned(16))) double rowbuf[2 * LB];
    __shared__ __attribute__((aligned(16))) double sp[64 * R + 16];
    const int tid = threadIdx.x, g = tid >> 3, sub = tid & 7;
    f64x2 a[R][S], vprev[S];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = g + 64 * r, c = 2 * sub + 16 * s;
            const double* row = A + (int64_t)min(i, L - 1) * lda;
            const double v0 = row[min(c, L - 1)], v1 = row[min(c + 1, L - 1)];
            a[r][s].x = (i < L && c < L) ? v0 : 0.0;
            a[r][s].y = (i < L && c + 1 < L) ? v1 : 0.0;
        }
#pragma unroll
    for (int s = 0; s < S; ++s) vprev[s] = f64x2{0.0, 0.0};
    for (int e = tid; e < Lz * Lz; e += 512) {                   // the caller's zero padding of V (rows / columns L .. Lz - 1)
        const int r = e / Lz, c = e - r * Lz;
        if (r >= L || c >= L) V[(int64_t)r * ldv + c] = 0.0;
    }
    for (int e = tid; e < (L - 2) * L; e += 512) HV[e] = 0.0;    // (a step writes the live part of its reflector row only)
    for (int e = tid; e < L; e += 512) { gg[e] = 0.0; if (e + 2 >= L) { tau[e] = 0.0; ee[e] = 0.0; } }
    for (int e = tid; e < 64 * R + 16; e += 512) sp[e] = 0.0;
    for (int e = tid; e < 2 * LB; e += 512) rowbuf[e] = 0.0;
    if (tid == 0 && reset_flag) *flag = 0;   // (a caller-owned verdict word accumulates: it is not reset here)
    __syncthreads();                          // (the zeroing of HV / gg above is ordered in front of the steps' stores to them)
    if (g == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) *reinterpret_cast<f64x2*>(rowbuf + 2 * sub + 16 * s) = a[0][s];
    }
    __syncthreads();
    chol_static_for<0, S>([&](auto S0c) {
        constexpr int S0 = decltype(S0c)::value;
        // the steps whose first live column k + 1 lies in slot S0: k = 16 S0 - 1 .. 16 S0 + 14
        const int k0 = max(16 * S0 - 1, 0), k1 = min(16 * S0 + 15, L - 2);
        for (int k = k0; k < k1; ++k) triq_step<S, R, S0>(a, vprev, L, k, rowbuf, sp, dd, ee, HV, tau, gg);
    });
    // the last 2 x 2 block: row L - 2 was published by the last step (or is row 0 / 1 of a matrix without steps)
    if (tid == 0) {
        const double* rb = rowbuf + ((L - 2) & 1) * LB;
        dd[L - 2] = rb[L - 2];
        ee[L - 2] = rb[L - 1];
    }
    if (g == ((L - 1) & 63) && sub == (((L - 1) & 15) >> 1)) {
        double dl = 0.0;
        chol_static_for<0, R>([&](auto rc) {
            chol_static_for<0, S>([&](auto sc) {
                constexpr int r = decltype(rc)::value, s = decltype(sc)::value;
                if (r == ((L - 1) >> 6) && s == ((L - 1) >> 4)) dl = ((L - 1) & 1) ? a[r][s].y : a[r][s].x;
            });
        });
        dd[L - 1] = dl;
    }
}

