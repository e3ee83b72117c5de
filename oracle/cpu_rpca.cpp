// cpu_rpca.cpp -- TEST INFRASTRUCTURE (never linked into the product, never on the product path).
//
// A dependency-free, threaded (OpenMP) C++ restatement of the reference's RandomizedPca fit, the second CPU baseline
// SURVEY.md section 7 step 1(c) / BASELINE.md section 4 ask for next to the numpy + LAPACK oracle: no BLAS, no LAPACK -- its
// own GEMM loops, partial-pivot LU, Householder QR and Jacobi SVD -- in the data's own precision T like the generic crate
// (A = f32 for the BASELINE configs).  Each function cites the reference lines it follows (relative to the crate root).
// Pinned by tests/test_cpu_rpca.py against the numpy oracle (which the reference's own known-answer tests pin).
//
//   RandomizedPca::inner_fit        src/pca.rs:509-550
//   randomized_svd                  src/pca.rs:668-686
//   randomized_range_finder         src/pca.rs:689-718   (lair's lu::Factorized::into_pl at :709, :712; linalg::qr at :716)
//   linalg::svddc (gesdd 'S')       src/linalg.rs:101-122 -> here: one-sided Jacobi on the l x d matrix
//   svd_flip                        src/pca.rs:815-850
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

template <class T>
struct Mat {  // row-major
    int64_t r = 0, c = 0;
    std::vector<T> v;
    Mat() = default;
    Mat(int64_t r_, int64_t c_) : r(r_), c(c_), v(size_t(r_) * c_, T(0)) {}
    T* row(int64_t i) { return v.data() + size_t(i) * c; }
    const T* row(int64_t i) const { return v.data() + size_t(i) * c; }
};

// C (m x n) = A (m x k) . B (k x n): `input.dot(&pl)` (pca.rs:707, 714), rows of A in parallel
template <class T>
Mat<T> gemm_nn(const Mat<T>& A, const Mat<T>& B) {
    Mat<T> C(A.r, B.c);
    const int64_t k = A.c, n = B.c;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < A.r; ++i) {
        T* ci = C.row(i);
        const T* ai = A.row(i);
        for (int64_t p = 0; p < k; ++p) {
            const T a = ai[p];
            const T* bp = B.row(p);
            for (int64_t j = 0; j < n; ++j) ci[j] += a * bp[j];
        }
    }
    return C;
}

// C (k x n) = A^T (k x m) . B (m x n): `input.t().dot(&pl)` (pca.rs:711), `q.t().dot(input)` (pca.rs:681): row blocks of
// A / B in parallel into per-thread partial sums, added in thread order
template <class T>
Mat<T> gemm_tn(const Mat<T>& A, const Mat<T>& B) {
    const int64_t k = A.c, n = B.c, m = A.r;
    Mat<T> C(k, n);
    const int nt = omp_get_max_threads();
    std::vector<Mat<T>> part(nt);
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        Mat<T> P(k, n);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < m; ++i) {
            const T* ai = A.row(i);
            const T* bi = B.row(i);
            for (int64_t p = 0; p < k; ++p) {
                const T a = ai[p];
                T* cp = P.row(p);
                for (int64_t j = 0; j < n; ++j) cp[j] += a * bi[j];
            }
        }
        part[t] = std::move(P);
    }
    for (int t = 0; t < nt; ++t)
        if (!part[t].v.empty())
            for (size_t e = 0; e < C.v.size(); ++e) C.v[e] += part[t].v[e];
    return C;
}

// lair::decomposition::lu::Factorized::from(a).into_pl() restated (call sites pca.rs:709, 712; lair 0.8 is not vendored:
// LU with partial row pivoting, P.L with L unit lower trapezoidal, m x min(m, n)); the caller's column slice
// `q.slice(s![.., ..min(nrows, ncols)])` (pca.rs:710, 713) is included
template <class T>
Mat<T> lu_pl(Mat<T> A) {
    const int64_t m = A.r, n = A.c, kk = std::min(m, n);
    std::vector<int64_t> perm(m);
    for (int64_t i = 0; i < m; ++i) perm[i] = i;  // perm[i] = original row now at position i
    for (int64_t j = 0; j < kk; ++j) {
        int64_t piv = j;
        T best = std::fabs(A.row(j)[j]);
        for (int64_t i = j + 1; i < m; ++i) {
            const T a = std::fabs(A.row(i)[j]);
            if (a > best) { best = a; piv = i; }
        }
        if (piv != j) {
            std::swap_ranges(A.row(j), A.row(j) + n, A.row(piv));
            std::swap(perm[j], perm[piv]);
        }
        const T d = A.row(j)[j];
        if (d == T(0)) continue;
        const T* rj = A.row(j);
#pragma omp parallel for schedule(static) if (m - j > 2048)
        for (int64_t i = j + 1; i < m; ++i) {
            T* ri = A.row(i);
            const T f = ri[j] / d;
            ri[j] = f;
            for (int64_t c = j + 1; c < n; ++c) ri[c] -= f * rj[c];
        }
    }
    Mat<T> PL(m, kk);  // row perm[i] of P.L is row i of L
    for (int64_t i = 0; i < m; ++i) {
        T* o = PL.row(perm[i]);
        for (int64_t c = 0; c < kk; ++c) o[c] = c < i ? A.row(i)[c] : (c == i ? T(1) : T(0));
    }
    return PL;
}

// linalg::qr (linalg.rs:127-147: gelqf + unglq of the transposed view == Householder QR): thin Q, m x min(m, n)
template <class T>
Mat<T> qr_thin(Mat<T> A) {
    const int64_t m = A.r, n = A.c, kk = std::min(m, n);
    std::vector<std::vector<T>> vs(kk);
    std::vector<T> taus(kk, T(0));
    for (int64_t j = 0; j < kk; ++j) {
        double nrm2 = 0;
#pragma omp parallel for reduction(+ : nrm2) schedule(static)
        for (int64_t i = j; i < m; ++i) nrm2 += double(A.row(i)[j]) * double(A.row(i)[j]);
        const T alpha = A.row(j)[j];
        const T nrm = T(std::sqrt(nrm2));
        std::vector<T> v(m - j, T(0));
        if (nrm == T(0)) { vs[j] = std::move(v); continue; }
        const T beta = alpha >= T(0) ? -nrm : nrm;
        taus[j] = (beta - alpha) / beta;
        const T scale = T(1) / (alpha - beta);
        v[0] = T(1);
        for (int64_t i = j + 1; i < m; ++i) v[i - j] = A.row(i)[j] * scale;
        // A[j:, j+1:] -= tau v (v^T A[j:, j+1:])
        const int64_t nc = n - j - 1;
        if (nc > 0) {
            std::vector<double> w(nc, 0.0);
#pragma omp parallel
            {
                std::vector<double> wl(nc, 0.0);
#pragma omp for schedule(static) nowait
                for (int64_t i = j; i < m; ++i) {
                    const T vi = v[i - j];
                    const T* ri = A.row(i) + j + 1;
                    for (int64_t c = 0; c < nc; ++c) wl[c] += double(vi) * double(ri[c]);
                }
#pragma omp critical
                for (int64_t c = 0; c < nc; ++c) w[c] += wl[c];
            }
            const T tau = taus[j];
#pragma omp parallel for schedule(static)
            for (int64_t i = j; i < m; ++i) {
                const T f = tau * v[i - j];
                T* ri = A.row(i) + j + 1;
                for (int64_t c = 0; c < nc; ++c) ri[c] -= f * T(w[c]);
            }
        }
        vs[j] = std::move(v);
    }
    Mat<T> Q(m, kk);
    for (int64_t i = 0; i < kk; ++i) Q.row(i)[i] = T(1);
    for (int64_t j = kk - 1; j >= 0; --j) {  // Q = H_0 H_1 ... H_{kk-1} [I; 0]
        const std::vector<T>& v = vs[j];
        if (taus[j] == T(0)) continue;
        const int64_t nc = kk - j;
        std::vector<double> w(nc, 0.0);
#pragma omp parallel
        {
            std::vector<double> wl(nc, 0.0);
#pragma omp for schedule(static) nowait
            for (int64_t i = j; i < m; ++i) {
                const T vi = v[i - j];
                const T* ri = Q.row(i) + j;
                for (int64_t c = 0; c < nc; ++c) wl[c] += double(vi) * double(ri[c]);
            }
#pragma omp critical
            for (int64_t c = 0; c < nc; ++c) w[c] += wl[c];
        }
        const T tau = taus[j];
#pragma omp parallel for schedule(static)
        for (int64_t i = j; i < m; ++i) {
            const T f = tau * v[i - j];
            T* ri = Q.row(i) + j;
            for (int64_t c = 0; c < nc; ++c) ri[c] -= f * T(w[c]);
        }
    }
    return Q;
}

// economy SVD of B (l x d, l <= d) by one-sided (Hestenes) Jacobi on the rows: B = U diag(s) Vt with U l x l, Vt l x d
// (stands for linalg::svddc -> gesdd('S'), linalg.rs:101-122); singular values descending
template <class T>
void svd_rows(const Mat<T>& B, Mat<double>& U, std::vector<double>& s, Mat<double>& Vt) {
    const int64_t l = B.r, d = B.c;
    Mat<double> W(l, d);
    for (size_t e = 0; e < W.v.size(); ++e) W.v[e] = double(B.v[e]);
    Mat<double> R(l, l);  // accumulated row rotations: W = R B
    for (int64_t i = 0; i < l; ++i) R.row(i)[i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double worst = 0;
        for (int64_t p = 0; p < l - 1; ++p)
            for (int64_t q = p + 1; q < l; ++q) {
                double a = 0, b = 0, g = 0;
                const double* wp = W.row(p);
                const double* wq = W.row(q);
                for (int64_t c = 0; c < d; ++c) { a += wp[c] * wp[c]; b += wq[c] * wq[c]; g += wp[c] * wq[c]; }
                if (g == 0.0 || std::fabs(g) <= 1e-15 * std::sqrt(a * b)) continue;
                worst = std::max(worst, std::fabs(g) / std::sqrt(a * b));
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                double* xp = W.row(p);
                double* xq = W.row(q);
                for (int64_t c = 0; c < d; ++c) { const double u = xp[c], w = xq[c]; xp[c] = cs * u - sn * w; xq[c] = sn * u + cs * w; }
                double* rp = R.row(p);
                double* rq = R.row(q);
                for (int64_t c = 0; c < l; ++c) { const double u = rp[c], w = rq[c]; rp[c] = cs * u - sn * w; rq[c] = sn * u + cs * w; }
            }
        if (worst <= 1e-15) break;
    }
    std::vector<double> nrm(l);
    std::vector<int64_t> order(l);
    for (int64_t i = 0; i < l; ++i) {
        double a = 0;
        for (int64_t c = 0; c < d; ++c) a += W.row(i)[c] * W.row(i)[c];
        nrm[i] = std::sqrt(a);
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return nrm[x] > nrm[y]; });
    U = Mat<double>(l, l); Vt = Mat<double>(l, d); s.assign(l, 0.0);
    for (int64_t j = 0; j < l; ++j) {
        const int64_t i = order[j];
        s[j] = nrm[i];
        for (int64_t c = 0; c < d; ++c) Vt.row(j)[c] = nrm[i] > 0 ? W.row(i)[c] / nrm[i] : 0.0;
        for (int64_t c = 0; c < l; ++c) U.row(c)[j] = R.row(i)[c];  // B = R^T W  =>  column j of U is row i of R
    }
}

template <class T>
int rpca_fit(const T* x, int64_t n, int64_t d, int64_t k, int64_t n_oversample, int64_t n_iter, int centering, const double* omega,
             double* components, double* singular, double* means, double* total_variance) {
    if (n < k || d < k) return 1;  // pca.rs:513-518
    const int64_t lreq = k + n_oversample;
    // means, centred copy (pca.rs:520-531)
    std::vector<double> mu(d, 0.0);
    Mat<T> X(n, d);
    if (centering) {
#pragma omp parallel
        {
            std::vector<double> ml(d, 0.0);
#pragma omp for schedule(static) nowait
            for (int64_t i = 0; i < n; ++i)
                for (int64_t j = 0; j < d; ++j) ml[j] += double(x[i * d + j]);
#pragma omp critical
            for (int64_t j = 0; j < d; ++j) mu[j] += ml[j];
        }
        for (int64_t j = 0; j < d; ++j) mu[j] /= double(n);
    }
    std::vector<T> muT(d);
    for (int64_t j = 0; j < d; ++j) muT[j] = T(mu[j]);
    double tv = 0;
#pragma omp parallel for reduction(+ : tv) schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < d; ++j) {
            const T v = x[i * d + j] - muT[j];
            X.row(i)[j] = v;
            tv += double(v) * double(v);  // pca.rs:533
        }
    // randomized_range_finder (pca.rs:689-718)
    Mat<T> P(d, lreq);
    for (int64_t e = 0; e < d * lreq; ++e) P.v[e] = T(omega[e]);  // pca.rs:701-705 (f64 draw cast to A::Real)
    Mat<T> Q = gemm_nn(X, P);                                      // pca.rs:707
    for (int64_t it = 0; it < n_iter; ++it) {                      // pca.rs:708-715
        Mat<T> pl = lu_pl(std::move(Q));                           // :709-710
        Mat<T> Y = gemm_tn(X, pl);                                 // :711
        pl = lu_pl(std::move(Y));                                  // :712-713
        Q = gemm_nn(X, pl);                                        // :714
    }
    Q = qr_thin(std::move(Q));                                     // :716
    Mat<T> B = gemm_tn(Q, X);                                      // pca.rs:681  (l x d)
    Mat<double> Uh, Vt;
    std::vector<double> s;
    svd_rows(B, Uh, s, Vt);                                        // pca.rs:682
    // U = Q Uh (pca.rs:683) only for svd_flip (pca.rs:684, 815-850): sign of the first max-|.| entry of each column
    const int64_t l = B.r;
    Mat<T> UhT(l, l);
    for (int64_t i = 0; i < l; ++i)
        for (int64_t j = 0; j < l; ++j) UhT.row(i)[j] = T(Uh.row(i)[j]);
    Mat<T> U = gemm_nn(Q, UhT);
    for (int64_t j = 0; j < std::min(l, k); ++j) {
        double best = -1, val = 0;
        for (int64_t i = 0; i < n; ++i) {
            const double a = std::fabs(double(U.row(i)[j]));
            if (a > best) { best = a; val = double(U.row(i)[j]); }
        }
        const double sg = val < 0 ? -1.0 : 1.0;
        for (int64_t c = 0; c < d; ++c) components[j * d + c] = sg * Vt.row(j)[c];
        singular[j] = s[j];
    }
    for (int64_t j = 0; j < d; ++j) means[j] = mu[j];
    *total_variance = tv;
    return 0;
}

}  // namespace

extern "C" {
int oracle_rpca_fit_f32(const float* x, int64_t n, int64_t d, int64_t k, int64_t n_oversample, int64_t n_iter, int centering,
                        const double* omega, int threads, double* components, double* singular, double* means, double* total_variance) {
    if (threads > 0) omp_set_num_threads(threads);
    return rpca_fit<float>(x, n, d, k, n_oversample, n_iter, centering, omega, components, singular, means, total_variance);
}
int oracle_rpca_fit_f64(const double* x, int64_t n, int64_t d, int64_t k, int64_t n_oversample, int64_t n_iter, int centering,
                        const double* omega, int threads, double* components, double* singular, double* means, double* total_variance) {
    if (threads > 0) omp_set_num_threads(threads);
    return rpca_fit<double>(x, n, d, k, n_oversample, n_iter, centering, omega, components, singular, means, total_variance);
}
int oracle_rpca_max_threads(void) { return omp_get_max_threads(); }
}
