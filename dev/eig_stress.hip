// Development stress test (not part of the product): op_eigh on adversarial symmetric matrices against a host Jacobi solver.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dev/eig_stress dev/eig_stress.hip petal-decomposition_amd/csrc/algo.cpp petal-decomposition_amd/csrc/rccl.cpp -ldl
// Reports, per matrix family, the worst eigenvalue error, eigen-residual and loss of orthogonality over sizes and seeds.
#include "../petal-decomposition_amd/csrc/hip_ops.hip"
#include <algorithm>
#include <cstring>
#include <random>
#include <string>
using namespace petal;

static void host_jacobi(std::vector<double> a, int n, std::vector<double>& w) {  // eigenvalues only, cyclic Jacobi
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) (i == j ? dg : off) += a[i * n + j] * a[i * n + j];
        if (off <= 1e-32 * dg || off == 0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0) continue;
                const double th = (a[q * n + q] - a[p * n + p]) / (2 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1)), c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; ++k) { const double x = a[k * n + p], y = a[k * n + q]; a[k * n + p] = c * x - s * y; a[k * n + q] = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = a[p * n + k], y = a[q * n + k]; a[p * n + k] = c * x - s * y; a[q * n + k] = s * x + c * y; }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = a[i * n + i];
    std::sort(w.begin(), w.end(), [](double x, double y) { return x > y; });
}

int main(int argc, char** argv) {
    char err[256];
    Dev* d = dev_create(0, nullptr, err, sizeof(err));
    if (!d) { printf("no dev: %s\n", err); return 1; }
    const int sizes[] = {3, 4, 7, 16, 17, 31, 64, 74, 80, 81, 100, 128, 129, 138, 139, 141, 142, 150};
    const char* fams[] = {"gram_g0", "gram_g4", "gram_g8", "gram_g14", "indefinite", "tridiag", "penta", "diag_plus_rank1", "near_split_1e-6",
                          "near_split_1e-10", "near_split_1e-13", "near_split_1e-15", "wilkinson", "cluster_1e-7", "cluster_1e-10", "lowrank_noise",
                          "scaled_1e120", "scaled_1e-120", "scaled_1e160", "equal_spaced_diag_rot", "two_blocks_rot", "ones_plus_diag"};
    const int maxL = 150;
    double *dA = (double*)dev_alloc(d, 8 * maxL * maxL), *dV = (double*)dev_alloc(d, 8 * maxL * maxL), *dw = (double*)dev_alloc(d, 8 * maxL);
    int bad = 0;
    for (const char* fam : fams) {
        for (double tol : {1e-8, 1e-15}) {
            double worst_w = 0, worst_r = 0, worst_o = 0;
            int worst_L = 0;
            for (int L : sizes) {
                for (int seed = 0; seed < 3; ++seed) {
                    std::mt19937_64 rng(1000 * L + seed);
                    std::normal_distribution<double> nd;
                    std::vector<double> S(L * L, 0.0);
                    auto rand_orth = [&](std::vector<double>& Q) {  // Gram-Schmidt of a random matrix
                        Q.assign(L * L, 0);
                        for (auto& v : Q) v = nd(rng);
                        for (int j = 0; j < L; ++j) {
                            for (int rep = 0; rep < 2; ++rep)
                                for (int k = 0; k < j; ++k) { double s = 0; for (int i = 0; i < L; ++i) s += Q[i * L + j] * Q[i * L + k]; for (int i = 0; i < L; ++i) Q[i * L + j] -= s * Q[i * L + k]; }
                            double nn = 0; for (int i = 0; i < L; ++i) nn += Q[i * L + j] * Q[i * L + j];
                            nn = 1 / sqrt(nn); for (int i = 0; i < L; ++i) Q[i * L + j] *= nn;
                        }
                    };
                    auto from_spectrum = [&](const std::vector<double>& lam) {
                        std::vector<double> Q; rand_orth(Q);
                        for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int k = 0; k < L; ++k) s += Q[i * L + k] * lam[k] * Q[j * L + k]; S[i * L + j] = s; }
                        for (int i = 0; i < L; ++i) for (int j = 0; j < i; ++j) S[i * L + j] = S[j * L + i];
                    };
                    const std::string f = fam;
                    if (f.rfind("gram_g", 0) == 0) {
                        const double g = atof(fam + 6);
                        const int M = 2 * L + 8;
                        std::vector<double> B(M * L);
                        for (int i = 0; i < M; ++i) for (int j = 0; j < L; ++j) B[i * L + j] = nd(rng) * pow(10.0, -0.5 * g * j / L);
                        for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int k = 0; k < M; ++k) s += B[k * L + i] * B[k * L + j]; S[i * L + j] = s; }
                    } else if (f == "indefinite") {
                        for (int i = 0; i < L; ++i) for (int j = i; j < L; ++j) S[i * L + j] = S[j * L + i] = nd(rng);
                    } else if (f == "tridiag" || f == "penta") {
                        const int bw = f == "tridiag" ? 1 : 2;
                        for (int i = 0; i < L; ++i) for (int j = i; j < std::min(L, i + bw + 1); ++j) S[i * L + j] = S[j * L + i] = nd(rng);
                    } else if (f == "diag_plus_rank1") {
                        std::vector<double> u(L); for (auto& v : u) v = nd(rng);
                        for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) S[i * L + j] = 0.3 * u[i] * u[j] + (i == j ? 1.0 + i : 0.0);
                    } else if (f.rfind("near_split_", 0) == 0) {
                        const double eps = atof(fam + 11);
                        for (int i = 0; i < L; ++i) { S[i * L + i] = 1.0 + 0.5 * (i % 7) + 0.01 * i; if (i + 1 < L) S[i * L + i + 1] = S[(i + 1) * L + i] = (i % 5 == 2) ? eps : 0.4; }
                    } else if (f == "wilkinson") {
                        for (int i = 0; i < L; ++i) { S[i * L + i] = fabs(i - (L - 1) / 2.0); if (i + 1 < L) S[i * L + i + 1] = S[(i + 1) * L + i] = 1.0; }
                    } else if (f.rfind("cluster_", 0) == 0) {
                        const double eps = atof(fam + 8);
                        std::vector<double> lam(L); for (int i = 0; i < L; ++i) lam[i] = (i % 3 == 0 ? 2.0 : 1.0) + eps * i;
                        from_spectrum(lam);
                    } else if (f == "lowrank_noise") {
                        std::vector<double> lam(L); for (int i = 0; i < L; ++i) lam[i] = i < L / 4 + 1 ? 10.0 / (1 + i) : 1e-9 * (1 + 0.3 * nd(rng));
                        from_spectrum(lam);
                    } else if (f.rfind("scaled_", 0) == 0) {
                        const double sc = atof(fam + 7);
                        std::vector<double> lam(L); for (int i = 0; i < L; ++i) lam[i] = sc * pow(0.9, i);
                        from_spectrum(lam);
                    } else if (f == "equal_spaced_diag_rot") {   // integers, rotated by a few exact Givens rotations of angle 45 deg in planes (0,1),(2,3)..
                        for (int i = 0; i < L; ++i) S[i * L + i] = 1.0 + i;
                        for (int i = 0; i + 1 < L; i += 2) { const double a = S[i * L + i], b = S[(i + 1) * L + i + 1]; S[i * L + i] = S[(i + 1) * L + i + 1] = 0.5 * (a + b); S[i * L + i + 1] = S[(i + 1) * L + i] = 0.5 * (b - a); }
                    } else if (f == "two_blocks_rot") {
                        for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) if ((i < L / 2) == (j < L / 2)) S[i * L + j] = (i == j ? 3.0 + 0.1 * i : 1.0 / (1 + abs(i - j)));
                    } else if (f == "ones_plus_diag") {
                        for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) S[i * L + j] = 1.0 + (i == j ? 0.5 * i : 0.0);
                    }
                    double nrmA = 0; for (double v : S) nrmA = fmax(nrmA, fabs(v));
                    dev_h2d(d, dA, S.data(), 8 * L * L);
                    (void)hipMemsetAsync(dV, 0, 8 * L * L, (hipStream_t)dev_stream(d));
                    op_eigh(d, dA, L, L, dV, L, dw, tol);
                    std::vector<double> w(L), V(L * L), wr;
                    dev_d2h(d, w.data(), dw, 8 * L); dev_d2h(d, V.data(), dV, 8 * L * L); dev_sync(d);
                    host_jacobi(S, L, wr);
                    const double scale = fmax(fabs(wr[0]), fabs(wr[L - 1]));
                    double ew = 0, er = 0, eo = 0;
                    for (int j = 0; j < L; ++j) ew = fmax(ew, fabs(w[j] - wr[j]) / scale);
                    if (f == "scaled_1e160") { bool fin = true; for (double v : w) fin = fin && std::isfinite(v); if (!fin) { ew = 1; } }
                    for (int j = 0; j < L; ++j) for (int i = 0; i < L; ++i) { double s = 0; for (int k = 0; k < L; ++k) s += S[i * L + k] * V[k * L + j]; er = fmax(er, fabs(s - w[j] * V[i * L + j]) / scale); }
                    for (int a = 0; a < L; ++a) for (int b = a; b < L; ++b) { double s = 0; for (int k = 0; k < L; ++k) s += V[k * L + a] * V[k * L + b]; eo = fmax(eo, fabs(s - (a == b))); }
                    if (!(ew == ew) || !(er == er) || !(eo == eo)) { ew = er = eo = 1; }
                    if (eo > worst_o || er > worst_r) worst_L = L;
                    worst_w = fmax(worst_w, ew); worst_r = fmax(worst_r, er); worst_o = fmax(worst_o, eo);
                    (void)nrmA;
                }
            }
            // budgets: eigenvalues and residuals eps-level; orthogonality eps / gap_tol (2e-8 for tol 1e-8, 2e-11 for 1e-15)
            const double bo = tol >= 1e-9 ? 2e-7 : 2e-10, br = tol >= 1e-9 ? 1e-9 : 1e-12;
            const bool ok = worst_w < 1e-12 && worst_r < br && worst_o < bo;
            if (!ok) ++bad;
            printf("%-22s tol %.0e  eigenvalues %.1e  residual %.1e  orthogonality %.1e (worst at L = %d)  %s\n", fam, tol, worst_w, worst_r, worst_o, worst_L, ok ? "ok" : "FAIL");
        }
    }
    printf("%d families out of budget\n", bad);
    return bad != 0;
}
