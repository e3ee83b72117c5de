// Development micro-benchmark of K1 / K2 at the north-star shape with experiment macros
// (-DPETAL_EXP_AFIXED / -DPETAL_EXP_BFIXED / -DPETAL_EXP_NOSTORE): which stream bounds the kernel?
#include "../petal-decomposition_amd/csrc/hip_ops.hip"
#include <random>
using namespace petal;
int main(int argc, char** argv) {
    int64_t n = argc > 1 ? atoll(argv[1]) : 1000000; int d = 512, l = 74, LP = 80;
    char err[256]; Dev* dv = dev_create(0, nullptr, err, sizeof(err)); if (!dv) { printf("%s\n", err); return 1; }
    float* X = (float*)dev_alloc(dv, n * d * 4); float* Z = (float*)dev_alloc(dv, n * LP * 4);
    std::vector<float> h(1 << 20); std::mt19937 g(1); std::normal_distribution<float> nd; for (auto& v : h) v = nd(g);
    for (int64_t off = 0; off < n * d; off += h.size()) dev_h2d(dv, X + off, h.data(), std::min<int64_t>(h.size(), n * d - off) * 4);
    std::vector<double> P(d * LP, 0.0); for (int i = 0; i < d; ++i) for (int j = 0; j < l; ++j) P[i * LP + j] = nd(g);
    double* dP = (double*)dev_alloc(dv, d * LP * 8); dev_h2d(dv, dP, P.data(), d * LP * 8);
    float* mu = (float*)dev_alloc(dv, d * 4); dev_memset(dv, mu, 0, d * 4);
    double* C = (double*)dev_alloc(dv, (size_t)d * LP * 8);
    dev_set_profiling(dv, true);
    for (int rep = 0; rep < 2; ++rep) {
        dev_reset_timing(dv);
        for (int i = 0; i < 5; ++i) { dev_set_tag(dv, TAG_XP); op_gemm_xp(dv, F32, X, n, d, d, mu, dP, LP, LP, nullptr, Z, LP, nullptr); dev_set_tag(dv, TAG_ATB); op_gemm_atb(dv, F32, X, d, d, mu, Z, LP, LP, nullptr, n, C, LP); }
        dev_sync(dv); KernelTiming kt = dev_timing(dv);
        if (rep) printf("n=%lld  K1 %.4f ms  %.1f TF   K2 %.4f ms  %.1f TF\n", (long long)n, kt.ms[TAG_XP] / kt.launches[TAG_XP], 2.0 * n * d * l / (kt.ms[TAG_XP] / kt.launches[TAG_XP] * 1e-3) / 1e12,
                        kt.ms[TAG_ATB] / kt.launches[TAG_ATB], 2.0 * n * d * l / (kt.ms[TAG_ATB] / kt.launches[TAG_ATB] * 1e-3) / 1e12);
    }
    return 0;
}
