// petal_decomposition.hpp -- header-only C++17 facade over the C ABI (petal_hip.h) with the reference crate's
// type and method names (petabi/petal-decomposition v0.9.0, src/lib.rs:17-18):
//
//   Pca / PcaBuilder                      src/pca.rs:41-283
//   RandomizedPca / RandomizedPcaBuilder  src/pca.rs:317-663
//   FastIca / FastIcaBuilder              src/ica.rs:41-308
//
// fit / transform / fit_transform / inverse_transform take and return `Array2<A>` (row-major, A = float | double),
// errors are thrown as `DecompositionError` with the crate's messages (src/lib.rs:22-28).  Models own their RNG and
// advance it on every fit like the crate (src/pca.rs:532, src/ica.rs:211); the default RNG is the crate's
// `Mcg128Xsl64` (rand_pcg) with a Ziggurat StandardNormal -- a restatement of un-vendored third-party code whose
// exact stream is NOT pinned by any reference test ("stream parity unpinned").
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "petal_hip.h"

namespace petal_decomposition {

struct DecompositionError : std::runtime_error {
    enum Kind { InvalidInput, LinalgError } kind;
    DecompositionError(Kind k, const std::string& m)
        : std::runtime_error((k == InvalidInput ? "invalid matrix: " : "linear algerba operation failed: ") + m), kind(k) {}
};

template <class A> struct DTypeOf;
template <> struct DTypeOf<float> { static constexpr int value = PETAL_F32; };
template <> struct DTypeOf<double> { static constexpr int value = PETAL_F64; };

// minimal owned row-major matrix (ndarray::Array2)
template <class A>
struct Array2 {
    std::vector<A> data;
    int64_t rows = 0, cols = 0;
    Array2() = default;
    Array2(int64_t r, int64_t c, A v = A(0)) : data(size_t(r) * c, v), rows(r), cols(c) {}
    Array2(std::initializer_list<std::initializer_list<A>> init) {
        rows = int64_t(init.size());
        cols = rows ? int64_t(init.begin()->size()) : 0;
        for (auto& r : init) data.insert(data.end(), r.begin(), r.end());
    }
    A& operator()(int64_t i, int64_t j) { return data[size_t(i) * cols + j]; }
    const A& operator()(int64_t i, int64_t j) const { return data[size_t(i) * cols + j]; }
    int64_t nrows() const { return rows; }
    int64_t ncols() const { return cols; }
    petal_matrix view() const {
        return petal_matrix{const_cast<A*>(data.data()), rows, cols, cols, 1, DTypeOf<A>::value, PETAL_HOST};
    }
};

// ---- rand_pcg::Mcg128Xsl64 + rand_distr::StandardNormal (Ziggurat), restated ---------------------------------
class Pcg {  // rand_pcg::Mcg128Xsl64: state *= MULT; output = rotr64(hi ^ lo, state >> 122)
  public:
    // Mcg128Xsl64::new(state) / Pcg64Mcg::new (src/pca.rs:991): the state is forced odd
    explicit Pcg(unsigned __int128 state) : state_(state | 1) {}
    // SeedableRng::from_seed(seed.to_be_bytes()) as the crate's with_seed does (src/pca.rs:356-359, src/ica.rs:75-78):
    // rand_pcg reads the 16 seed bytes as a little-endian u128, i.e. the big-endian bytes of `seed` byte-swapped
    static Pcg from_seed_be_bytes(unsigned __int128 seed) {
        unsigned __int128 sw = 0;
        for (int i = 0; i < 16; ++i) sw |= ((seed >> (8 * i)) & 0xff) << (8 * (15 - i));
        return Pcg(sw);
    }
    uint64_t next_u64() {
        state_ *= (((unsigned __int128)0x2360ED051FC65DA4ull) << 64) | 0x4385DF649FCCF645ull;
        const unsigned rot = unsigned(state_ >> 122);
        const uint64_t x = uint64_t(state_ >> 64) ^ uint64_t(state_);
        return (x >> rot) | (x << ((64 - rot) & 63));
    }
    double next_f64() { return double(next_u64() >> 11) * (1.0 / 9007199254740992.0); }            // rand Standard: [0, 1)
    double next_f64_open01() { return double(next_u64() >> 12) * (1.0 / 4503599627370496.0) + (1.0 / 9007199254740992.0); }  // Open01
    // rand_distr::StandardNormal: the 256-layer Ziggurat (tables recomputed from the published recurrence)
    double standard_normal() {
        static const Tables t;
        for (;;) {
            const uint64_t bits = next_u64();
            const int i = int(bits & 0xff);
            // (bits >> 12) as the mantissa of a float in [2, 4), minus 3: u in [-1, 1)
            const double u = double(int64_t(bits >> 12)) * (2.0 / 4503599627370496.0) - 1.0;
            const double x = u * t.x[i];
            if (std::fabs(x) < t.x[i + 1]) return x;
            if (i == 0) {  // tail beyond R
                double xx = 1.0, yy = 0.0;
                while (-2.0 * yy < xx * xx) {
                    xx = std::log(next_f64_open01()) / Tables::R;
                    yy = std::log(next_f64_open01());
                }
                return u < 0 ? xx - Tables::R : Tables::R - xx;
            }
            if (t.f[i + 1] + (t.f[i] - t.f[i + 1]) * next_f64() < std::exp(-0.5 * x * x)) return x;
        }
    }

  private:
    struct Tables {
        static constexpr double R = 3.654152885361009;
        double x[257], f[257];
        Tables() {
            const double v = 0.00492867323399;  // area of each layer
            x[0] = v / std::exp(-0.5 * R * R);
            x[1] = R;
            for (int i = 2; i < 256; ++i) x[i] = std::sqrt(-2.0 * std::log(v / x[i - 1] + std::exp(-0.5 * x[i - 1] * x[i - 1])));
            x[256] = 0.0;
            for (int i = 0; i < 257; ++i) f[i] = std::exp(-0.5 * x[i] * x[i]);
        }
    };
    unsigned __int128 state_;
};

inline unsigned __int128 random_seed() {
    std::random_device rd;
    unsigned __int128 s = 0;
    for (int i = 0; i < 4; ++i) s = (s << 32) | rd();
    return s;
}

// one context per process / GPU, shared by the models (src: models are plain data, the ctx is the device handle)
class Context {
  public:
    explicit Context(int device = 0, void* stream = nullptr) {
        if (petal_ctx_create(device, stream, &ctx_) != PETAL_OK || !ctx_)
            throw DecompositionError(DecompositionError::LinalgError, "no usable gfx950 device (there is no CPU fallback)");
    }
    ~Context() { petal_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    petal_ctx* get() const { return ctx_; }
    void check(int rc) const {
        if (rc == PETAL_OK) return;
        const std::string msg = petal_last_error(ctx_);
        throw DecompositionError(rc == PETAL_INVALID_INPUT ? DecompositionError::InvalidInput : DecompositionError::LinalgError, msg);
    }
    static Context& global() { static Context c; return c; }

  private:
    petal_ctx* ctx_ = nullptr;
};

namespace detail {
template <class A>
struct PcaState {  // src/pca.rs:41-51 / 317-329
    Array2<A> components;
    int64_t n_samples = 0;
    std::vector<A> means;
    A total_variance = A(0);
    std::vector<A> singular;
    bool centering = true;
    int64_t k = 0;
    Context* ctx = nullptr;
    Context& context() const { return ctx ? *ctx : Context::global(); }

    Array2<A> transform(const Array2<A>& input) const {  // src/pca.rs:726-750
        const int64_t d = int64_t(means.size());
        if (input.ncols() != d) throw DecompositionError(DecompositionError::InvalidInput, "# of columns should be " + std::to_string(d));
        Array2<A> y(input.nrows(), k);
        petal_matrix mx = input.view(), my = y.view();
        context().check(petal_transform(context().get(), &mx, components.data.data(), means.data(), k, d, centering, &my));
        return y;
    }
    Array2<A> inverse_transform(const Array2<A>& input) const {  // src/pca.rs:788-811
        const int64_t d = int64_t(means.size());
        if (input.ncols() != k) throw DecompositionError(DecompositionError::InvalidInput, "# of columns should be " + std::to_string(k));
        Array2<A> x(input.nrows(), d);
        petal_matrix my = input.view(), mx = x.view();
        context().check(petal_inverse_transform(context().get(), &my, components.data.data(), means.data(), k, d, centering, &mx));
        return x;
    }
    std::vector<A> explained_variance_ratio() const {  // src/pca.rs:101-105
        std::vector<A> r(singular.size());
        for (size_t i = 0; i < r.size(); ++i) r[i] = singular[i] * singular[i] / total_variance;
        return r;
    }
};
}  // namespace detail

template <class A>
class Pca {  // src/pca.rs:41-232
  public:
    explicit Pca(int64_t n_components, bool centering = true, Context* ctx = nullptr) {
        st_.k = n_components; st_.centering = centering; st_.ctx = ctx;
        st_.components = Array2<A>(n_components, 0);
    }
    const Array2<A>& components() const { return st_.components; }
    const std::vector<A>& mean() const { return st_.means; }
    int64_t n_components() const { return st_.k; }
    const std::vector<A>& singular_values() const { return st_.singular; }
    std::vector<A> explained_variance_ratio() const { return st_.explained_variance_ratio(); }
    void fit(const Array2<A>& input) { inner_fit(input, nullptr); }
    Array2<A> fit_transform(const Array2<A>& input) {
        if (st_.centering && input.nrows() == 0) { inner_fit(input, nullptr); return Array2<A>(0, st_.k ? input.ncols() : 0); }
        Array2<A> y(input.nrows(), st_.k);
        inner_fit(input, &y);
        return y;
    }
    Array2<A> transform(const Array2<A>& input) const { return st_.transform(input); }
    Array2<A> inverse_transform(const Array2<A>& input) const { return st_.inverse_transform(input); }

  private:
    void inner_fit(const Array2<A>& input, Array2<A>* y) {
        const int64_t d = input.ncols(), k = st_.k;
        Array2<A> comp(k, d);
        std::vector<A> means(d), sing(k);
        A tv = A(0);
        petal_matrix mx = input.view(), my{};
        if (y) my = y->view();
        st_.context().check(petal_pca_fit(st_.context().get(), &mx, k, st_.centering, comp.data.data(), means.data(), sing.data(),
                                          &tv, y ? &my : nullptr));
        if (st_.centering && input.nrows() == 0) return;
        st_.components = std::move(comp); st_.means = std::move(means); st_.singular = std::move(sing);
        st_.total_variance = tv; st_.n_samples = input.nrows();
    }
    detail::PcaState<A> st_;
};

class PcaBuilder {  // src/pca.rs:246-283
  public:
    explicit PcaBuilder(int64_t n_components) : k_(n_components) {}
    static PcaBuilder new_(int64_t n_components) { return PcaBuilder(n_components); }
    PcaBuilder& centering(bool c) { centering_ = c; return *this; }
    PcaBuilder& context(Context* c) { ctx_ = c; return *this; }
    template <class A> Pca<A> build() const { return Pca<A>(k_, centering_, ctx_); }

  private:
    int64_t k_; bool centering_ = true; Context* ctx_ = nullptr;
};

template <class A, class R = Pcg>
class RandomizedPca {  // src/pca.rs:317-551
  public:
    RandomizedPca(int64_t n_components, R rng, bool centering = true, Context* ctx = nullptr) : rng_(rng) {
        st_.k = n_components; st_.centering = centering; st_.ctx = ctx;
        st_.components = Array2<A>(n_components, 0);
    }
    static RandomizedPca with_seed(int64_t n_components, unsigned __int128 seed) { return RandomizedPca(n_components, R::from_seed_be_bytes(seed)); }
    static RandomizedPca with_rng(int64_t n_components, R rng) { return RandomizedPca(n_components, rng); }
    const Array2<A>& components() const { return st_.components; }
    const std::vector<A>& mean() const { return st_.means; }
    int64_t n_components() const { return st_.k; }
    const std::vector<A>& singular_values() const { return st_.singular; }
    std::vector<A> explained_variance_ratio() const { return st_.explained_variance_ratio(); }
    void fit(const Array2<A>& input) { inner_fit(input, nullptr); }
    Array2<A> fit_transform(const Array2<A>& input) {
        Array2<A> y(input.nrows(), st_.k);
        inner_fit(input, &y);
        return y;
    }
    Array2<A> transform(const Array2<A>& input) const { return st_.transform(input); }
    Array2<A> inverse_transform(const Array2<A>& input) const { return st_.inverse_transform(input); }
    static constexpr int64_t N_OVERSAMPLE = 10, N_ITER = 7;  // src/pca.rs:679-680

  private:
    void inner_fit(const Array2<A>& input, Array2<A>* y) {
        const int64_t d = input.ncols(), k = st_.k, l = k + N_OVERSAMPLE;
        // the crate draws Omega only after the shape check and the mean (src/pca.rs:513-532, 701-705); an input it
        // rejects or returns early on must not advance the model's RNG
        std::vector<A> omega;
        const bool will_draw = !(input.nrows() < k || d < k) && !(st_.centering && input.nrows() == 0);
        if (will_draw) {
            omega.resize(size_t(d) * l);
            for (auto& v : omega) v = A(rng_.standard_normal());  // row-major d x l fill, f64 draw cast to A
        }
        Array2<A> comp(k, d);
        std::vector<A> means(d), sing(k);
        A tv = A(0);
        petal_matrix mx = input.view(), my{};
        if (y) my = y->view();
        st_.context().check(petal_rpca_fit(st_.context().get(), &mx, k, N_OVERSAMPLE, N_ITER, st_.centering,
                                           omega.empty() ? nullptr : omega.data(), comp.data.data(), means.data(), sing.data(),
                                           &tv, y ? &my : nullptr));
        if (st_.centering && input.nrows() == 0) return;
        st_.components = std::move(comp); st_.means = std::move(means); st_.singular = std::move(sing);
        st_.total_variance = tv; st_.n_samples = input.nrows();
    }
    R rng_;
    detail::PcaState<A> st_;
};

template <class R = Pcg>
class RandomizedPcaBuilder {  // src/pca.rs:564-663
  public:
    explicit RandomizedPcaBuilder(int64_t n_components) : k_(n_components), rng_(R::from_seed_be_bytes(random_seed())) {}
    static RandomizedPcaBuilder new_(int64_t n_components) { return RandomizedPcaBuilder(n_components); }
    static RandomizedPcaBuilder with_rng(R rng, int64_t n_components) { RandomizedPcaBuilder b(n_components); b.rng_ = rng; return b; }
    RandomizedPcaBuilder& seed(unsigned __int128 s) { rng_ = R::from_seed_be_bytes(s); return *this; }
    RandomizedPcaBuilder& centering(bool c) { centering_ = c; return *this; }
    RandomizedPcaBuilder& context(Context* c) { ctx_ = c; return *this; }
    template <class A> RandomizedPca<A, R> build() const { return RandomizedPca<A, R>(k_, rng_, centering_, ctx_); }

  private:
    int64_t k_; R rng_; bool centering_ = true; Context* ctx_ = nullptr;
};

template <class A, class R = Pcg>
class FastIca {  // src/ica.rs:41-221
  public:
    explicit FastIca(R rng, Context* ctx = nullptr) : rng_(rng), ctx_(ctx) {}
    static FastIca with_seed(unsigned __int128 seed) { return FastIca(R::from_seed_be_bytes(seed)); }
    static FastIca with_rng(R rng) { return FastIca(rng); }
    void fit(const Array2<A>& input) { inner_fit(input, nullptr); }
    Array2<A> fit_transform(const Array2<A>& input) {
        const int64_t nc = std::min(input.nrows(), input.ncols());
        if (input.nrows() == 0) return Array2<A>(0, input.ncols());  // src/ica.rs:174-176
        Array2<A> y(input.nrows(), nc);
        inner_fit(input, &y);
        return y;
    }
    Array2<A> transform(const Array2<A>& input) const {  // src/ica.rs:120-131
        const int64_t d = int64_t(means_.size());
        if (input.ncols() != d) throw DecompositionError(DecompositionError::InvalidInput, "too many columns");
        Array2<A> y(input.nrows(), components_.nrows());
        petal_matrix mx = input.view(), my = y.view();
        context().check(petal_transform(context().get(), &mx, components_.data.data(), means_.data(), components_.nrows(), d, 1, &my));
        return y;
    }
    const Array2<A>& components() const { return components_; }
    int64_t n_iter() const { return n_iter_; }
    int mode = PETAL_ICA_TEXTBOOK;  // PETAL_ICA_REFERENCE_LITERAL follows src/ica.rs:345-349, 369-380 as written

  private:
    Context& context() const { return ctx_ ? *ctx_ : Context::global(); }
    void inner_fit(const Array2<A>& input, Array2<A>* y) {
        if (input.nrows() == 0) return;
        const int64_t d = input.ncols(), nc = std::min(input.nrows(), d);  // src/ica.rs:173
        std::vector<A> w_init(size_t(nc) * nc);
        for (auto& v : w_init) v = A(rng_.standard_normal());  // src/ica.rs:210-214
        Array2<A> comp(nc, d);
        std::vector<A> means(d);
        int64_t it = 0;
        petal_matrix mx = input.view(), my{};
        if (y) my = y->view();
        context().check(petal_fastica_fit(context().get(), &mx, 0, 1e-4, 200, mode, w_init.data(), comp.data.data(), means.data(),
                                          &it, y ? &my : nullptr));
        components_ = std::move(comp); means_ = std::move(means); n_iter_ = it;
    }
    R rng_;
    Context* ctx_;
    Array2<A> components_;
    std::vector<A> means_;
    int64_t n_iter_ = 0;
};

template <class R = Pcg>
class FastIcaBuilder {  // src/ica.rs:244-308
  public:
    FastIcaBuilder() : rng_(R::from_seed_be_bytes(random_seed())) {}
    static FastIcaBuilder new_() { return FastIcaBuilder(); }
    static FastIcaBuilder with_rng(R rng) { FastIcaBuilder b; b.rng_ = rng; return b; }
    FastIcaBuilder& seed(unsigned __int128 s) { rng_ = R::from_seed_be_bytes(s); return *this; }
    FastIcaBuilder& context(Context* c) { ctx_ = c; return *this; }
    template <class A> FastIca<A, R> build() const { return FastIca<A, R>(rng_, ctx_); }

  private:
    R rng_; Context* ctx_ = nullptr;
};

}  // namespace petal_decomposition
