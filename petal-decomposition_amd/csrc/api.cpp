// api.cpp -- the extern "C" boundary declared in include/petal_hip.h.  Every entry point catches all
// exceptions: nothing unwinds or aborts across the ABI (the reference panics at ica.rs:369 and
// linalg.rs:75/106/132; here those become PETAL_LINALG_ERROR / PETAL_INVALID_INPUT).
#include <cstdio>
#include <new>

#include "ctx.h"

using namespace petal;

namespace {

// makes the ctx's device current on the calling thread for the duration of an entry point (the ctx may live on a device
// other than the thread's current one, or be used from a thread that never called hipSetDevice) and restores the
// previous device afterwards
struct DeviceScope {
    petal::Dev* dev;
    int prev = -1;
    explicit DeviceScope(petal::Dev* d) : dev(d) { if (dev) prev = dev_push_current(dev); }
    ~DeviceScope() { if (dev) dev_pop_current(dev, prev); }
};

template <class F>
int guarded(petal_ctx* ctx, F&& f) {
    if (!ctx) return PETAL_INVALID_INPUT;
    try {
        ctx->err.clear();
        DeviceScope scope(ctx->dev);
        f();
        return PETAL_OK;
    } catch (const Error& e) {
        ctx->err = e.what();
        if (ctx->dev) dev_abort(ctx->dev);
        return e.code;
    } catch (const std::bad_alloc&) {
        ctx->err = "out of host memory";
        if (ctx->dev) dev_abort(ctx->dev);
        return PETAL_DEVICE_ERROR;
    } catch (const std::exception& e) {
        ctx->err = e.what();
        if (ctx->dev) dev_abort(ctx->dev);
        return PETAL_DEVICE_ERROR;
    } catch (...) {
        ctx->err = "unknown error";
        if (ctx->dev) dev_abort(ctx->dev);
        return PETAL_DEVICE_ERROR;
    }
}

void need(const void* p, const char* what) {
    if (!p) invalid_input(std::string(what) + " must not be null");
}

}  // namespace

extern "C" {

const char* petal_version(void) { return "petal-hip 0.1.0 (gfx950)"; }

int petal_ctx_create(int device, void* stream, petal_ctx** out) {
    if (!out) return PETAL_INVALID_INPUT;
    *out = nullptr;
    petal_ctx* c = new (std::nothrow) petal_ctx();
    if (!c) return PETAL_DEVICE_ERROR;
    char err[512] = {0};
    c->dev = dev_create(device, stream, err, sizeof(err));
    if (!c->dev) {
        std::fprintf(stderr, "petal_ctx_create: %s\n", err);
        delete c;
        return PETAL_DEVICE_ERROR;
    }
    c->force_collective = std::getenv("PETAL_FORCE_COLLECTIVE") != nullptr;   // (the default of PETAL_OPT_FORCE_COLLECTIVE)
    *out = c;
    return PETAL_OK;
}

void petal_ctx_destroy(petal_ctx* ctx) {
    if (!ctx) return;
    try { rccl_release(*ctx); } catch (...) {}
    if (ctx->dev) dev_destroy(ctx->dev);
    delete ctx;
}

int petal_rccl_unique_id(void* out128) {
    if (!out128) return PETAL_INVALID_INPUT;
    try { rccl_unique_id(out128); return PETAL_OK; } catch (const Error& e) { return e.code; } catch (...) { return PETAL_DEVICE_ERROR; }
}

int petal_ctx_init_rccl(petal_ctx* ctx, const void* unique_id128, int rank, int world_size) {
    return guarded(ctx, [&] { rccl_init(*ctx, unique_id128, rank, world_size); });
}

const char* petal_last_error(const petal_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int petal_ctx_set_collective(petal_ctx* ctx, petal_allreduce_fn fn, void* user, int rank, int world_size) {
    return guarded(ctx, [&] {
        if (world_size < 1 || rank < 0 || rank >= world_size) invalid_input("bad rank / world_size");
        if (world_size > 1 && !fn) invalid_input("world_size > 1 needs an all-reduce hook");
        rccl_release(*ctx);
        ctx->allreduce = fn;
        ctx->allreduce_user = user;
        ctx->rank = rank;
        ctx->world = world_size;
    });
}

int petal_ctx_set_profiling(petal_ctx* ctx, int profiling) {
    return guarded(ctx, [&] {
        ctx->profiling = profiling < 0 ? 0 : (profiling > 2 ? 2 : profiling);
        dev_set_profiling(ctx->dev, ctx->profiling);
    });
}

int petal_ctx_collective_info(const petal_ctx* ctx, int* kind, int* rank, int* world_size, int* comm_count, int* comm_device,
                              int* comm_rank) {
    if (!ctx) return PETAL_INVALID_INPUT;
    int cnt = -1, dev = -1, rk = -1;
    petal::rccl_info(*ctx, &cnt, &dev, &rk);
    if (kind) *kind = ctx->rccl ? 2 : (ctx->allreduce ? 1 : 0);
    if (rank) *rank = ctx->rank;
    if (world_size) *world_size = ctx->world;
    if (comm_count) *comm_count = cnt;
    if (comm_device) *comm_device = dev;
    if (comm_rank) *comm_rank = rk;
    return PETAL_OK;
}

int petal_ctx_set_gemm_mode(petal_ctx* ctx, int mode) {
    return guarded(ctx, [&] {
        if (mode != PETAL_GEMM_SPLIT_BF16X3 && mode != PETAL_GEMM_FP32_MFMA && mode != PETAL_GEMM_SPLIT_BF16X3_EXACT) invalid_input("unknown GEMM mode");
        dev_set_gemm_mode(ctx->dev, mode);
    });
}

int petal_ctx_set_option(petal_ctx* ctx, int option, double value) {
    return guarded(ctx, [&] {
        if (option == PETAL_OPT_FORCE_COLLECTIVE) { ctx->force_collective = value != 0; return; }
        if (option < 0 || option >= OPT_COUNT || !(value == value)) invalid_input("unknown ctx option or NaN value");
        dev_set_option(ctx->dev, option, value);
    });
}

int petal_ctx_get_option(const petal_ctx* ctx, int option, double* value) {
    if (!ctx || !value) return PETAL_INVALID_INPUT;
    if (option == PETAL_OPT_FORCE_COLLECTIVE) { *value = ctx->force_collective ? 1.0 : 0.0; return PETAL_OK; }
    if (option < 0 || option >= OPT_COUNT) return PETAL_INVALID_INPUT;
    *value = dev_option(ctx->dev, option);
    return PETAL_OK;
}

int petal_get_stats(const petal_ctx* ctx, petal_stats* out) {
    if (!ctx || !out) return PETAL_INVALID_INPUT;
    *out = ctx->stats;
    return PETAL_OK;
}

int petal_pca_fit(petal_ctx* ctx, const petal_matrix* x, int64_t k, int centering, void* components, void* means,
                  void* singular, void* total_variance, const petal_matrix* y_out) {
    return guarded(ctx, [&] {
        need(x, "x");
        pca_fit(*ctx, *x, k, centering != 0, components, means, singular, total_variance, y_out);
    });
}

int petal_rpca_fit(petal_ctx* ctx, const petal_matrix* x, int64_t k, int64_t n_oversample, int64_t n_iter,
                   int centering, const void* omega, void* components, void* means, void* singular,
                   void* total_variance, const petal_matrix* y_out) {
    return guarded(ctx, [&] {
        need(x, "x");
        rpca_fit(*ctx, *x, k, n_oversample, n_iter, centering != 0, omega, components, means, singular, total_variance,
                 y_out);
    });
}

int petal_transform(petal_ctx* ctx, const petal_matrix* x, const void* components, const void* means, int64_t k,
                    int64_t d, int centering, const petal_matrix* y_out) {
    return guarded(ctx, [&] {
        need(x, "x");
        need(y_out, "y_out");
        transform(*ctx, *x, components, means, k, d, centering != 0, *y_out);
    });
}

int petal_inverse_transform(petal_ctx* ctx, const petal_matrix* y, const void* components, const void* means,
                            int64_t k, int64_t d, int centering, const petal_matrix* x_out) {
    return guarded(ctx, [&] {
        need(y, "y");
        need(x_out, "x_out");
        inverse_transform(*ctx, *y, components, means, k, d, centering != 0, *x_out);
    });
}

int petal_fastica_fit(petal_ctx* ctx, const petal_matrix* x, int64_t n_components, double tol, int64_t max_iter,
                      int mode, const void* w_init, void* components, void* means, int64_t* n_iter,
                      const petal_matrix* y_out) {
    return guarded(ctx, [&] {
        need(x, "x");
        fastica_fit(*ctx, *x, n_components, tol, max_iter, mode, w_init, components, means, n_iter, y_out);
    });
}

int petal_ica_par(petal_ctx* ctx, const petal_matrix* x1, double tol, int64_t max_iter, int mode, const void* w_init,
                  void* w_out, int64_t* n_iter) {
    return guarded(ctx, [&] {
        need(x1, "x1");
        need(w_init, "w_init");
        need(w_out, "w_out");
        ica_par(*ctx, *x1, tol, max_iter, mode, w_init, w_out, n_iter);
    });
}

int petal_symmetric_decorrelation(petal_ctx* ctx, const void* w, int64_t nc, int32_t dtype, int mode, void* out) {
    return guarded(ctx, [&] {
        need(w, "w");
        need(out, "out");
        symmetric_decorrelation(*ctx, w, nc, dtype, mode, out);
    });
}

int petal_logcosh(petal_ctx* ctx, const petal_matrix* x, const petal_matrix* g_out, void* gprime_out) {
    return guarded(ctx, [&] {
        need(x, "x");
        need(g_out, "g_out");
        need(gprime_out, "gprime_out");
        logcosh(*ctx, *x, *g_out, gprime_out);
    });
}

int petal_svd_flip(petal_ctx* ctx, const petal_matrix* u, const petal_matrix* vt) {
    return guarded(ctx, [&] {
        need(u, "u");
        need(vt, "vt");
        svd_flip(*ctx, *u, *vt);
    });
}

int petal_gemm_xp(petal_ctx* ctx, const petal_matrix* x, const void* mu, const void* p, int64_t N, const void* bias,
                  const petal_matrix* z_out) {
    return guarded(ctx, [&] {
        need(x, "x");
        need(p, "p");
        need(z_out, "z_out");
        gemm_xp(*ctx, *x, mu, p, N, bias, *z_out);
    });
}

int petal_power_pass(petal_ctx* ctx, const petal_matrix* x, const void* mu, const void* p, int64_t N, double* y_out,
                     const petal_matrix* z_out, int* fused_out) {
    return guarded(ctx, [&] {
        need(x, "x");
        need(p, "p");
        need(y_out, "y_out");
        power_pass(*ctx, *x, mu, p, N, y_out, z_out, fused_out);
    });
}

int petal_gemm_atb(petal_ctx* ctx, const petal_matrix* a, const void* mu_a, const petal_matrix* b, const void* mu_b,
                   double* c_out) {
    return guarded(ctx, [&] {
        need(a, "a");
        need(c_out, "c_out");
        gemm_atb(*ctx, *a, mu_a, b, mu_b, c_out);
    });
}

}  // extern "C"
