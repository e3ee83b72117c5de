// ctx.h -- petal_ctx and small host-side helpers shared by algo.cpp / api.cpp.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/petal_hip.h"
#include "ops.h"

struct petal_ctx {
    petal::Dev* dev = nullptr;
    std::string err;
    petal_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    int rank = 0, world = 1;
    int profiling = 0;
    bool force_collective = false;   // PETAL_OPT_FORCE_COLLECTIVE (default: env PETAL_FORCE_COLLECTIVE at petal_ctx_create)
    petal_stats stats{};
    void* rccl = nullptr;  // the built-in RCCL communicator (rccl.cpp), when petal_ctx_init_rccl installed it
    // the fixed pseudo-random start block of the subspace iteration (topk_eigh), kept on the device per shape: generating it on the
    // host and uploading it cost 20 us of idle device per exact Pca / FastICA fit
    double* topk_seed = nullptr;
    int64_t topk_seed_d = 0, topk_seed_dp = 0, topk_seed_p = 0;
};

namespace petal {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] inline void invalid_input(const std::string& m) { throw Error(PETAL_INVALID_INPUT, m); }
[[noreturn]] inline void linalg_error(const std::string& m) { throw Error(PETAL_LINALG_ERROR, m); }
[[noreturn]] inline void device_error(const std::string& m) { throw Error(PETAL_DEVICE_ERROR, m); }

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// true when the fit must take the sharded code path: several ranks, or PETAL_OPT_FORCE_COLLECTIVE with a collective
// installed (runs the complete multi-rank path -- packing kernels, all-reduces -- on a one-rank group: test / timing aid)
inline bool sharded(const petal_ctx& c) {
    return c.world > 1 || (c.allreduce != nullptr && c.force_collective);
}

// rccl.cpp: the built-in collective
void rccl_unique_id(void* out128);
void rccl_init(petal_ctx& c, const void* unique_id128, int rank, int world);
void rccl_release(petal_ctx& c);
void rccl_info(const petal_ctx& c, int* count, int* device, int* rank);

// RAII device buffer from the ctx's caching allocator
struct DBuf {
    Dev* dev = nullptr;
    void* p = nullptr;
    size_t bytes = 0;
    DBuf() = default;
    DBuf(Dev* d, size_t b) : dev(d), p(b ? dev_alloc(d, b) : nullptr), bytes(b) {}
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : dev(o.dev), p(o.p), bytes(o.bytes) { o.p = nullptr; }
    DBuf& operator=(DBuf&& o) noexcept {
        if (this != &o) { release(); dev = o.dev; p = o.p; bytes = o.bytes; o.p = nullptr; }
        return *this;
    }
    ~DBuf() { release(); }
    void release() { if (p) dev_free(dev, p); p = nullptr; }
    double* f64() const { return static_cast<double*>(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// a row-major matrix in device memory, K-dimension padded to a multiple of 16 with zeros
struct DevMat {
    const void* p = nullptr;
    int64_t n = 0, d = 0, dp = 0, ld = 0;  // rows, real cols, padded cols, leading dimension (elements)
    int dtype = F32;
    DBuf owned;                             // empty when zero-copy
    bool zero_copy = false;                 // the caller's device buffer is streamed in place
};

// host algorithms (algo.cpp)
DevMat ingest(petal_ctx& c, const petal_matrix& x);
void   emit(petal_ctx& c, int dtype, const void* src, int64_t n, int64_t cols, int64_t ld, const petal_matrix& out, const double* scale = nullptr);   // scale: device, cols doubles (per-column factor on the way out)
void   allreduce_f64(petal_ctx& c, double* dev_buf, int64_t count, int op);

void rpca_fit(petal_ctx& c, const petal_matrix& x, int64_t k, int64_t n_oversample, int64_t n_iter, bool centering,
              const void* omega, void* components, void* means, void* singular, void* total_variance,
              const petal_matrix* y_out);
void pca_fit(petal_ctx& c, const petal_matrix& x, int64_t k, bool centering, void* components, void* means,
             void* singular, void* total_variance, const petal_matrix* y_out);
void transform(petal_ctx& c, const petal_matrix& x, const void* components, const void* means, int64_t k, int64_t d,
               bool centering, const petal_matrix& y_out);
void inverse_transform(petal_ctx& c, const petal_matrix& y, const void* components, const void* means, int64_t k,
                       int64_t d, bool centering, const petal_matrix& x_out);
void fastica_fit(petal_ctx& c, const petal_matrix& x, int64_t n_components, double tol, int64_t max_iter, int mode,
                 const void* w_init, void* components, void* means, int64_t* n_iter, const petal_matrix* y_out);
void ica_par(petal_ctx& c, const petal_matrix& x1, double tol, int64_t max_iter, int mode, const void* w_init,
             void* w_out, int64_t* n_iter);
void symmetric_decorrelation(petal_ctx& c, const void* w, int64_t nc, int dtype, int mode, void* out);
void logcosh(petal_ctx& c, const petal_matrix& x, const petal_matrix& g_out, void* gprime_out);
void svd_flip(petal_ctx& c, const petal_matrix& u, const petal_matrix& vt);
void gemm_xp(petal_ctx& c, const petal_matrix& x, const void* mu, const void* p, int64_t N, const void* bias,
             const petal_matrix& z_out);
void gemm_atb(petal_ctx& c, const petal_matrix& a, const void* mu_a, const petal_matrix* b, const void* mu_b, double* c_out);
void power_pass(petal_ctx& c, const petal_matrix& x, const void* mu, const void* p, int64_t N, double* y_out, const petal_matrix* z_out,
                int* fused_out);

}  // namespace petal
