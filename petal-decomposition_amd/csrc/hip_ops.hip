// hip_ops.hip -- gfx950 (MI355X / CDNA4) implementation of the device-operation layer (ops.h).
//
// Kernel inventory (DESIGN.md section 4 has the layout and the roofline of each; EXPERIMENTS.md what was tried on it):
//   split-product kernels (default GEMM mode "bf16x3": every fp32 operand split exactly into three bf16 planes, five or six piece
//   products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation)
//     k_xp3        K1  Z = (X - mu) . P        X streamed into A fragments, centred and split in registers, P planes through LDS
//     k_atb3       K2  Y = (X - mu)^T . Z      row chunks -> fp32 slabs, combined in fp64 in a fixed order (k_sum_parts*)
//     k_pow3       K3  Y' = Xc^T (Xc P)        the FUSED power iteration: one pass over X, P and Y' in registers, X planes through a
//                                              swizzled LDS image read back transposed (ds_read_b64_tr_b16); 512 features, l <= 80
//     k_gram5          C = Xc^T Xc (+ means)   256 x 256 tiles over row chunks, panels fetched in fragment order and split under the
//                                              MFMAs, upper sub-tiles only
//     k_ica3p      K7  fused FastICA step      on planes of X1 made once per loop (k_ica_planes); k_ica3: splits X1 every iteration
//   fp32 MFMA kernels (GEMM mode "fp32": v_mfma_f32_16x16x4_f32): k_xp_mfma / k_xp_pers (K1), k_atb_mfma (K2), k_ica_mfma (K7)
//   fp64 MFMA kernels (v_mfma_f64_16x16x4_f64): k_xp_f64, k_atb_f64 (K1 / K2 for fp64 data; the precise Gram matrix), k_syrk_f64,
//     k_trsm_pack, k_dgemm / k_gemm_nn_f64
//   one-workgroup fp64 small-matrix kernels: k_chol_rt4 (re-basing Cholesky, RT form, register-resident on four waves), k_chol_inv2 (blocked
//     Cholesky + explicit inverse), k_tridiag_r / k_tridiag_w +
//     k_trieig_r (symmetric eigenproblem up to order 138), k_jacobi_* (fallbacks and one-sided SVD), k_symdecorr / k_ica_tail
//     (symmetric decorrelation: scaled Newton-Schulz polar factor in LDS)
//   *_simple     generic (any shape, f32 / f64, fp64 accumulate) kernels for small / unaligned / f64 inputs
//
// wave = 64 lanes everywhere.  MFMA 16x16x4 f32 fragment maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][j = l & 15];
//   C/D: reg r of lane l is D[row = 4 (l >> 4) + r][col = l & 15].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "ops.h"

namespace petal {

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);  \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef PETAL_DEBUG_COUNTERS
__device__ int g_dbg[4];
__device__ long long g_cyc[32];
__device__ long long g_trace[8 * 16];   // k_pow3: absolute s_memtime of one workgroup's waves at the marks of one stage
#define DBG_T(i) do { if (threadIdx.x == 0) { long long _t = clock64(); g_cyc[i] += _t - _t0; _t0 = _t; } } while (0)
}  // namespace petal
// development builds only (-DPETAL_DEBUG_COUNTERS): read and clear the in-kernel phase counters
extern "C" void petal_debug_trace(long long* out128) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out128, HIP_SYMBOL(petal::g_trace), sizeof(long long) * 128);
}
extern "C" void petal_debug_counters(long long* cyc16, int* dbg4) {  // (32 counters)
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(cyc16, HIP_SYMBOL(petal::g_cyc), sizeof(long long) * 32);
    (void)hipMemcpyFromSymbol(dbg4, HIP_SYMBOL(petal::g_dbg), sizeof(int) * 4);
    long long z[32] = {0}; int zi[4] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(petal::g_cyc), z, sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(petal::g_dbg), zi, sizeof(zi));
}
namespace petal {
#else
#define DBG_T(i) do {} while (0)
#endif

// ================================================================================================
// Dev
// ================================================================================================
struct Dev {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // side stream for work that is independent of the main chain (dev_fork .. dev_fork_end .. dev_join)
    hipStream_t side = nullptr, main_saved = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool forked = false, on_side = false;
    std::vector<void*> deferred_free;   // blocks released while forked: back to the pool at the join
    int gemm_mode = [] { const char* e = getenv("PETAL_GEMM"); return (e && std::string(e) == "fp32") ? 1 : (e && std::string(e) == "bf16x3-exact") ? 2 : 0; }();
    // FastICA: W's bf16 planes for the split-product step kernel; the tail kernel refreshes them with the W it writes, so only
    // the first iteration of a fit runs the separate pack kernel
    void* ica_wpk3 = nullptr;
    size_t ica_wpk3_bytes = 0;
    const double* ica_wpk3_for = nullptr;
    int64_t ica_wpk3_nc = 0;
    bool ica_wpk3_valid = false;
    void* ica_x1pl = nullptr;             // the pre-split planes of the whitened data of the CURRENT fixed-point loop (op_ica_prepare)
    size_t ica_x1pl_bytes = 0;
    const void* ica_x1pl_for = nullptr;   // ... made from this X1T (nullptr: none)
    int64_t ica_x1pl_n = 0, ica_x1pl_ld = 0;
    // accepted ||W W^T - I||_F^2 of the decorrelation inside the loop: 1e-7 relative for fp32 data (whose outputs are fp32),
    // fp64 round-off otherwise; set by op_ica_step from the data type it is handed
    double ica_ortho_tol2 = 1e-26;
    int profiling = 0;  // 0 off, 1 = time ONE launch per tag and fit (rotating over the launches), 2 = every launch
    int tag = 0;
    int tag_seen[TAG_COUNT] = {};    // tagged launches so far in this fit
    int tag_count[TAG_COUNT] = {};   // tagged launches of the previous fit (period of the rotation)
    int tag_pick[TAG_COUNT] = {};    // index of the launch that is timed in this fit
    int tag_active = 0;              // level 1: the ONE tag that is sampled in this fit (rotates over the tags the last fit used)
    int fit_index = 0;
    // pinned staging for small device-to-host results: copies are queued back to back on the stream and handed to the
    // caller's (pageable) buffers at the next dev_sync, instead of one blocking staged copy each
    char* pin = nullptr;
    char* pin_dev = nullptr;         // the ring's address as the device sees it
    bool d2h_kernel = true;          // small results leave through a copy kernel (default) or hipMemcpyAsync: OPT_D2H_KERNEL
    double opt[OPT_COUNT] = {};      // dev_option / dev_set_option (defaults: dev_defaults_from_env, once, at dev_create)
    size_t pin_cap = 0, pin_used = 0;
    struct Pend { void* dst; size_t off, bytes; };
    std::vector<Pend> pend;
    // free blocks by size.  `free_list` ("hot"): released since the last synchronisation of the main stream -- their last users may
    // still be running there, which is fine for the main stream (stream order) and for a side stream that waits for it;
    // `free_cold`: released before it -- safe for anyone, in particular for a side stream that starts at once.
    std::multimap<size_t, void*> free_list, free_cold;
    bool side_nowait = false;
    std::unordered_map<void*, size_t> live;
    std::vector<hipEvent_t> ev_pool;
    struct Rec { int tag; hipEvent_t a, b; };
    std::vector<Rec> recs;
    KernelTiming acc;
    // per-DEVICE launch state (hipFuncSetAttribute is per device: a second ctx on another GPU of the same process needs its own)
    std::set<const void*> max_lds_set;
    int num_cu = 0;
    int* progress = nullptr;  // pinned, device-writable: FastICA's tail kernel reports {converged at, iterations done} here
};

// The ONLY place the product reads its environment knobs (besides PETAL_GEMM / PETAL_FORCE_COLLECTIVE / PETAL_DEBUG at ctx creation):
// defaults of the ctx options.  Everything after this goes through dev_option.
static void dev_defaults_from_env(Dev* d) {
    auto on = [](const char* name) { return getenv(name) != nullptr; };
    auto num = [](const char* name, double dflt) { const char* e = getenv(name); return e ? atof(e) : dflt; };
    d->opt[OPT_TWO_PLANE] = on("PETAL_NO_P2") ? 0 : 1;
    d->opt[OPT_TWO_PLANE_OMEGA] = on("PETAL_NO_P2_OMEGA") ? 0 : 1;
    d->opt[OPT_TWO_PLANE_ITERATE] = on("PETAL_NO_P2_ITERATE") ? 0 : 1;
    d->opt[OPT_STEERING] = on("PETAL_NO_POW3_FAST") ? 0 : 1;
    d->opt[OPT_FUSED_PASS] = on("PETAL_NO_POW3") ? 0 : 1;
    d->opt[OPT_FUSED_PASS_MIN_ROWS] = num("PETAL_POW3_MIN_ROWS", 8192);
    d->opt[OPT_VERDICT_THRESHOLD] = num("PETAL_P2_VERDICT_THR", 4e-6);
    d->opt[OPT_MEANS_FOLD_ROWS] = on("PETAL_NO_MEANS_FOLD") ? -1 : num("PETAL_MEANS_FOLD_ROWS", 200000);
    d->opt[OPT_GRAM_SPLIT] = on("PETAL_NO_GRAM3") ? 0 : 1;
    d->opt[OPT_GRAM_SPLIT_HOOK] = on("PETAL_GRAM_SPLIT") ? 1 : 0;
    d->opt[OPT_D2H_KERNEL] = on("PETAL_D2H_MEMCPY") ? 0 : 1;
    d->opt[OPT_ROW_PAD] = on("PETAL_NO_ROW_PAD") ? 0 : 1;
    d->opt[OPT_EIGH_JACOBI] = on("PETAL_EIGH_JACOBI") ? 1 : 0;
    d->opt[OPT_POISON] = on("PETAL_POISON") ? 1 : 0;
    d->d2h_kernel = d->opt[OPT_D2H_KERNEL] != 0;
}
void dev_set_option(Dev* d, int opt, double value) {
    if (opt < 0 || opt >= OPT_COUNT) throw std::invalid_argument("unknown ctx option");
    d->opt[opt] = value;
    if (opt == OPT_D2H_KERNEL) d->d2h_kernel = value != 0;
}
double dev_option(const Dev* d, int opt) {
    if (opt < 0 || opt >= OPT_COUNT) throw std::invalid_argument("unknown ctx option");
    return d->opt[opt];
}
static inline bool opt_on(const Dev* d, int opt) { return d->opt[opt] != 0; }

Dev* dev_create(int device, void* stream, char* err, size_t errlen) {
    auto fail = [&](const std::string& m) -> Dev* {
        if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str());
        return nullptr;
    };
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(std::string("no HIP device visible (") + hipGetErrorString(e) + "); this library has no CPU fallback");
    if (device < 0 || device >= count) return fail("device index out of range");
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail("hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
    Dev* d = new Dev();
    d->device = device;
    dev_defaults_from_env(d);
    if (stream) {
        d->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
            delete d;
            return fail("hipStreamCreate failed");
        }
        d->own_stream = true;
    }
    return d;
}

void dev_destroy(Dev* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipStreamSynchronize(d->stream);
    for (auto& kv : d->free_list) (void)hipFree(kv.second);
    for (auto& kv : d->free_cold) (void)hipFree(kv.second);
    for (auto& kv : d->live) (void)hipFree(kv.first);
    for (auto& r : d->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto ev : d->ev_pool) (void)hipEventDestroy(ev);
    if (d->pin) (void)hipHostFree(d->pin);
    if (d->progress) (void)hipHostFree(d->progress);
    if (d->side) { (void)hipStreamSynchronize(d->side); (void)hipStreamDestroy(d->side); (void)hipEventDestroy(d->ev_fork); (void)hipEventDestroy(d->ev_join); }
    if (d->own_stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

void* dev_stream(Dev* d) { return d->stream; }
volatile int* dev_host_progress(Dev* d) {
    if (!d->progress) {
        // coherent (uncached on the device side) + mapped: the kernel's system-scope stores must become visible to the polling
        // host while the stream is still running, whatever HIP_HOST_COHERENT / the platform default says
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d->progress), 64, hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(d->progress, 0, 64);
    }
    return d->progress;
}
void dev_make_current(Dev* d) { HIP_CHECK(hipSetDevice(d->device)); }
int dev_push_current(Dev* d) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != d->device) HIP_CHECK(hipSetDevice(d->device));
    return prev;
}
void dev_pop_current(Dev* d, int prev) {
    if (prev >= 0 && prev != d->device) (void)hipSetDevice(prev);
}

void* dev_alloc(Dev* d, size_t bytes) {
    const size_t sz = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
    void* p = nullptr;
    const bool cold_only = d->on_side && d->side_nowait;   // (a side stream that did not wait for the main stream's queue)
    auto it = cold_only ? d->free_list.end() : d->free_list.find(sz);
    auto ic = it == d->free_list.end() ? d->free_cold.find(sz) : d->free_cold.end();
    if (it != d->free_list.end()) {
        p = it->second;
        d->free_list.erase(it);
    } else if (ic != d->free_cold.end()) {
        p = ic->second;
        d->free_cold.erase(ic);
    } else {
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {  // drop the cache and retry once
            (void)hipStreamSynchronize(d->stream);
            if (d->side) (void)hipStreamSynchronize(d->side);
            for (auto& kv : d->free_list) (void)hipFree(kv.second);
            for (auto& kv : d->free_cold) (void)hipFree(kv.second);
            d->free_list.clear();
            d->free_cold.clear();
            HIP_CHECK(hipMalloc(&p, sz));
        }
    }
    d->live[p] = sz;
    // PETAL_POISON=1 (test aid): every block handed out is filled with 0xFF bytes (NaN as fp32 / fp64), so a kernel that
    // reads memory it never wrote produces a visible NaN instead of depending on what the block held before
    if (opt_on(d, OPT_POISON)) HIP_CHECK(hipMemsetAsync(p, 0xFF, sz, d->stream));
    return p;
}

void dev_free(Dev* d, void* p) {
    if (!p) return;
    auto it = d->live.find(p);
    if (it == d->live.end()) return;
    if (d->forked) { d->deferred_free.push_back(p); return; }   // two streams in flight: no reuse before they have joined
    // stream-ordered reuse: every consumer of this block was enqueued on d->stream before this call
    d->free_list.emplace(it->second, p);
    d->live.erase(it);
}
// Fork / join of a side stream.  Between dev_fork() and dev_fork_end() every launch and copy goes to the side stream, which
// starts behind everything the main stream holds at the fork; after dev_fork_end() work goes to the main stream again and the
// two run concurrently until dev_join() makes the main stream wait for the side stream's last operation.  The pool is
// stream-ordered for ONE stream, so no block released between fork and join is handed out again before the join.
// after_main = false: the side work needs nothing the main stream still has queued (only buffers whose earlier users have been
// synchronised with): it starts at once.
void dev_fork(Dev* d, bool after_main) {
    if (d->forked) throw std::logic_error("dev_fork: already forked");
    if (!d->side) {
        HIP_CHECK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming));
    }
    if (after_main) {
        HIP_CHECK(hipEventRecord(d->ev_fork, d->stream));
        HIP_CHECK(hipStreamWaitEvent(d->side, d->ev_fork, 0));
    }
    d->main_saved = d->stream;
    d->stream = d->side;
    d->forked = true;
    d->on_side = true;
    d->side_nowait = !after_main;
}
void dev_fork_end(Dev* d) {
    if (!d->on_side) return;
    HIP_CHECK(hipEventRecord(d->ev_join, d->stream));
    d->stream = d->main_saved;
    d->on_side = false;
}
void dev_join(Dev* d) {
    if (!d->forked) return;
    dev_fork_end(d);
    HIP_CHECK(hipStreamWaitEvent(d->stream, d->ev_join, 0));
    d->forked = false;
    for (void* p : d->deferred_free) dev_free(d, p);
    d->deferred_free.clear();
}
// error path: leave the fork whatever state it is in (both streams drained, blocks back in the pool)
void dev_fork_abort(Dev* d) {
    if (!d->forked) return;
    if (d->on_side) { d->stream = d->main_saved; d->on_side = false; }
    (void)hipStreamSynchronize(d->side);
    (void)hipStreamSynchronize(d->stream);
    d->forked = false;
    for (void* p : d->deferred_free) dev_free(d, p);
    d->deferred_free.clear();
}

void dev_memset(Dev* d, void* p, int v, size_t bytes) { if (bytes) HIP_CHECK(hipMemsetAsync(p, v, bytes, d->stream)); }
void dev_h2d(Dev* d, void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));  // callers pass short-lived pageable buffers
}
// recycle = false: hand the finished copies over but keep every ring slot (a side stream may still have a queued transfer that
// reads one: dev_sync between dev_fork_end and dev_join only waits for the main stream -- ADVICE round 4)
static void drain_pending(Dev* d, bool recycle = true) {
    for (auto& p : d->pend) std::memcpy(p.dst, d->pin + p.off, p.bytes);
    d->pend.clear();
    if (recycle) d->pin_used = 0;
}
constexpr size_t PIN_MAX_COPY = size_t(8) << 20, PIN_RING = size_t(32) << 20;
// Small results leave through a KERNEL that stores them straight into the pinned ring (device-visible host memory, posted writes over
// the link) instead of hipMemcpyAsync: a device-to-host copy is a blit kernel of 3-4 us behind a 6-12 us gap of runtime work (two
// of them close every RandomizedPca fit, four every Pca fit: profiles/r04_timeline_*), this one starts like any other launch.
template <class V>
__global__ __launch_bounds__(256) void k_copy_out(const V* __restrict__ src, V* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static void ensure_pin(Dev* d) {
    if (d->pin) return;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d->pin), PIN_RING, hipHostMallocDefault));
    d->pin_cap = PIN_RING;
    void* dp = nullptr;   // (the same address on this platform; asked for rather than assumed)
    d->pin_dev = (hipHostGetDevicePointer(&dp, d->pin, 0) == hipSuccess && dp) ? static_cast<char*>(dp) : nullptr;
}
static void copy_to_pin(Dev* d, size_t off, const void* src, size_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(src);
    if (!d->d2h_kernel || !d->pin_dev || (bytes & 3) || (a & 3)) {
        HIP_CHECK(hipMemcpyAsync(d->pin + off, src, bytes, hipMemcpyDeviceToHost, d->stream));
        return;
    }
    char* dst = d->pin_dev + off;   // (64-byte aligned slots)
    if (!((bytes | a) & 15)) {
        const size_t n = bytes / 16;
        hipLaunchKernelGGL(k_copy_out<uint4>, dim3((unsigned)std::min<size_t>(64, (n + 255) / 256)), dim3(256), 0, d->stream,
                           static_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n);
    } else if (!((bytes | a) & 7)) {
        const size_t n = bytes / 8;
        hipLaunchKernelGGL(k_copy_out<uint2>, dim3((unsigned)std::min<size_t>(64, (n + 255) / 256)), dim3(256), 0, d->stream,
                           static_cast<const uint2*>(src), reinterpret_cast<uint2*>(dst), n);
    } else {
        const size_t n = bytes / 4;
        hipLaunchKernelGGL(k_copy_out<unsigned>, dim3((unsigned)std::min<size_t>(64, (n + 255) / 256)), dim3(256), 0, d->stream,
                           static_cast<const unsigned*>(src), reinterpret_cast<unsigned*>(dst), n);
    }
    HIP_CHECK(hipGetLastError());
}
static void pin_make_room(Dev* d, size_t need) {
    ensure_pin(d);
    if (d->pin_used + need > d->pin_cap) {  // ring full: finish what is queued, hand it over, start again
        HIP_CHECK(hipStreamSynchronize(d->stream));
        if (d->forked) { HIP_CHECK(hipStreamSynchronize(d->side)); HIP_CHECK(hipStreamSynchronize(d->main_saved)); }
        drain_pending(d);
    }
}
static size_t pin_reserve(Dev* d, size_t bytes) {
    const size_t need = (bytes + 63) / 64 * 64;
    pin_make_room(d, need);
    const size_t off = d->pin_used;
    d->pin_used += need;
    return off;
}
void dev_d2h(Dev* d, void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    if (bytes > PIN_MAX_COPY) {  // large results go straight to the caller's buffer
        HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, d->stream));
        return;
    }
    const size_t off = pin_reserve(d, bytes);
    copy_to_pin(d, off, src, bytes);
    d->pend.push_back({dst, off, bytes});
}
// Several small results in ONE launch (a fit that ends with five separate copies pays five launches)
struct CopySegs { const unsigned* src[8]; unsigned* dst[8]; unsigned n[8]; };
__global__ __launch_bounds__(256) void k_copy_out_multi(CopySegs s) {
    const int seg = blockIdx.y;
    const unsigned* __restrict__ src = s.src[seg];
    unsigned* __restrict__ dst = s.dst[seg];
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < s.n[seg]; i += gridDim.x * 256) dst[i] = src[i];
}
void dev_d2h_multi(Dev* d, int nseg, void* const* dst, const void* const* src, const size_t* bytes) {
    ensure_pin(d);
    bool ok = d->d2h_kernel && d->pin_dev && nseg <= 8;
    size_t total = 0;
    for (int i = 0; i < nseg; ++i) {
        ok = ok && !(bytes[i] & 3) && !(reinterpret_cast<uintptr_t>(src[i]) & 3) && bytes[i] <= PIN_MAX_COPY;
        total += (bytes[i] + 63) / 64 * 64;
    }
    if (!ok || total > PIN_MAX_COPY) {
        for (int i = 0; i < nseg; ++i) dev_d2h(d, dst[i], src[i], bytes[i]);
        return;
    }
    pin_make_room(d, total);   // (all segments in one stretch of the ring)
    CopySegs cs{};
    int m = 0;
    unsigned nmax = 0;
    for (int i = 0; i < nseg; ++i) {
        if (!bytes[i]) continue;
        const size_t off = pin_reserve(d, bytes[i]);
        cs.src[m] = static_cast<const unsigned*>(src[i]);
        cs.dst[m] = reinterpret_cast<unsigned*>(d->pin_dev + off);
        cs.n[m] = (unsigned)(bytes[i] / 4);
        nmax = std::max(nmax, cs.n[m]);
        d->pend.push_back({dst[i], off, bytes[i]});
        ++m;
    }
    if (!m) return;
    hipLaunchKernelGGL(k_copy_out_multi, dim3(std::min<unsigned>(32, (nmax + 255) / 256), m), dim3(256), 0, d->stream, cs);
    HIP_CHECK(hipGetLastError());
}
// The same without a destination: the caller reads the result IN the pinned ring after its dev_sync (valid until the next copy is
// queued on this Dev) -- no hand-over copy, and a caller that has to transform the data on its way out does it in one pass.
const void* dev_d2h_view(Dev* d, const void* src, size_t bytes) {
    if (!bytes) return nullptr;
    if (bytes > PIN_MAX_COPY) throw std::logic_error("dev_d2h_view: larger than a ring slot");
    const size_t off = pin_reserve(d, bytes);
    copy_to_pin(d, off, src, bytes);
    return d->pin + off;
}
// Host-to-device without blocking the host: the bytes are copied into the pinned ring now (so the caller's buffer may be
// a short-lived pageable one) and the transfer is queued on the stream; the slot is recycled at the next dev_sync.
void dev_h2d_async(Dev* d, void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    if (bytes > PIN_MAX_COPY) { dev_h2d(d, dst, src, bytes); return; }
    const size_t off = pin_reserve(d, bytes);
    std::memcpy(d->pin + off, src, bytes);
    HIP_CHECK(hipMemcpyAsync(dst, d->pin + off, bytes, hipMemcpyHostToDevice, d->stream));
}
// The same without the transfer: the bytes are copied into the pinned ring and the ring's DEVICE-visible address is returned -- a
// kernel that reads its (small) operand once may as well read it over the link itself (valid until the next dev_sync).
const void* dev_h2d_view(Dev* d, const void* src, size_t bytes) {
    if (!bytes) return nullptr;
    ensure_pin(d);
    if (bytes > PIN_MAX_COPY || !d->pin_dev) return nullptr;
    const size_t off = pin_reserve(d, bytes);
    std::memcpy(d->pin + off, src, bytes);
    return d->pin_dev + off;
}
void dev_abort(Dev* d) {  // error path: the destinations of queued copies may be gone
    dev_fork_abort(d);
    (void)hipStreamSynchronize(d->stream);
    d->pend.clear();
    d->pin_used = 0;
}
void dev_d2d(Dev* d, void* dst, const void* src, size_t bytes) {
    if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, d->stream));
}
void dev_copy2d(Dev* d, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int kind) {
    if (!width || !height) return;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, k, d->stream));
    if (kind == 0) HIP_CHECK(hipStreamSynchronize(d->stream));
}
size_t dev_view_limit(Dev*) { return PIN_MAX_COPY; }
void dev_sync(Dev* d) {
    HIP_CHECK(hipStreamSynchronize(d->stream));
    drain_pending(d, /*recycle=*/!d->forked);
    if (!d->forked) {   // every block released so far has no user left anywhere
        for (auto& kv : d->free_list) d->free_cold.emplace(kv.first, kv.second);
        d->free_list.clear();
    }
}
void dev_set_profiling(Dev* d, int level) { d->profiling = level; }
void dev_set_gemm_mode(Dev* d, int mode) { d->gemm_mode = mode; }
int dev_gemm_mode(const Dev* d) { return d->gemm_mode; }
void dev_set_tag(Dev* d, int tag) { d->tag = tag; }

static hipEvent_t get_event(Dev* d) {
    if (!d->ev_pool.empty()) { hipEvent_t e = d->ev_pool.back(); d->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}
static void resolve_events(Dev* d) {
    for (auto& r : d->recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { d->acc.ms[r.tag] += ms; d->acc.launches[r.tag] += 1; }
        d->ev_pool.push_back(r.a);
        d->ev_pool.push_back(r.b);
    }
    d->recs.clear();
}
void dev_reset_timing(Dev* d) {
    if (!d->recs.empty()) { (void)hipStreamSynchronize(d->stream); resolve_events(d); }
    d->acc = KernelTiming{};
    // Level 1 samples ONE launch per fit in all: an event pair is a ~5 us bubble in the stream, and a RandomizedPca fit has two
    // tagged kinds (three when sharded).  The sampled kind rotates over the kinds the previous fit used, the launch index of a kind
    // advances each time the kind has had its turn: K fits still visit every launch position of every kind.
    int used[TAG_COUNT], nused = 0;
    for (int t = 1; t < TAG_COUNT; ++t) {
        if (d->tag_seen[t] > 0) d->tag_count[t] = d->tag_seen[t];
        if (d->tag_count[t] > 0) used[nused++] = t;
        d->tag_seen[t] = 0;
    }
    if (nused > 0 && d->tag_active > 0 && d->tag_count[d->tag_active] > 0)
        d->tag_pick[d->tag_active] = (d->tag_pick[d->tag_active] + 1) % d->tag_count[d->tag_active];
    ++d->fit_index;
    d->tag_active = nused > 0 ? used[d->fit_index % nused] : 0;   // 0: no history yet -- the first fit samples every kind
}
KernelTiming dev_timing(Dev* d) {
    if (!d->recs.empty()) { HIP_CHECK(hipStreamSynchronize(d->stream)); resolve_events(d); }
    return d->acc;
}
// brackets the dominant kernel of a tagged op with events on the launch stream
struct TagScope {
    Dev* d; bool on; hipEvent_t a{}, b{};
    explicit TagScope(Dev* dev) : d(dev), on(false) {
        if (d->profiling && d->tag > 0 && d->tag < TAG_COUNT) {
            const int idx = d->tag_seen[d->tag]++;
            on = d->profiling >= 2 || (idx == d->tag_pick[d->tag] && (d->tag_active == 0 || d->tag_active == d->tag));
        }
        if (on) { a = get_event(d); b = get_event(d); HIP_CHECK(hipEventRecord(a, d->stream)); }
    }
    void stop() {
        if (on) { HIP_CHECK(hipEventRecord(b, d->stream)); d->recs.push_back({d->tag, a, b}); on = false; }
    }
};

void* dev_span_begin(Dev* d, int tag) {
    if (!d->profiling || tag <= 0 || tag >= TAG_COUNT) return nullptr;
    const int idx = d->tag_seen[tag]++;
    // (collective calls are bracketed in EVERY profiled fit: ten event pairs on a fit of tens of milliseconds cost nothing, and a
    // three-step bench run is then sure to have timed them -- with four kinds in rotation it was a matter of phase)
    if (!(d->profiling >= 2 || tag == TAG_COMM || (idx == d->tag_pick[tag] && (d->tag_active == 0 || d->tag_active == tag)))) return nullptr;
    Dev::Rec* r = new Dev::Rec{tag, get_event(d), get_event(d)};
    HIP_CHECK(hipEventRecord(r->a, d->stream));
    return r;
}
void dev_span_end(Dev* d, void* token) {
    if (!token) return;
    Dev::Rec* r = static_cast<Dev::Rec*>(token);
    HIP_CHECK(hipEventRecord(r->b, d->stream));
    d->recs.push_back(*r);
    delete r;
}

static inline void launch_check() { HIP_CHECK(hipGetLastError()); }
static inline int cdiv(int64_t a, int64_t b) { return int((a + b - 1) / b); }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

#include "kernels/k_generic.inc"   // generic kernels: packing, column scans, slab combines, any-shape fp64-accumulate products
#include "kernels/k_gemm_fp32.inc"   // K1 on the fp32 matrix cores (k_xp_mfma, k_xp_pers)
#include "kernels/k_gemm_split_k1.inc"   // K1, split-product form (k_pack_p3, k_xp3)
#include "kernels/k_gemm_k2.inc"   // K2: k_atb_mfma (fp32 MFMA), k_atb3 (split-product)
#include "kernels/k_pow3.inc"   // K3: the fused power-iteration pass (k_pow3, k_pow3f)
#include "kernels/k_gemm_f64.inc"   // the fp64-MFMA products (k_atb_f64, k_xp_f64)
#include "kernels/k_ica_step.inc"   // K7: the fused FastICA step (k_ica_mfma, k_ica3, k_ica3p, k_ica_reduce)
#include "kernels/k_smallmat.inc"   // fp64 small-matrix kernels: k_dgemm, k_trsm_pack, k_trsm_left_pack, k_chol_inv2, k_chol_rt4, Jacobi solvers
#include "kernels/k_eigen.inc"   // the two-stage symmetric eigen-solver, the polar iteration / FastICA tail, result kernels
#include "kernels/host_gemm.inc"   // host-side launchers: packing, column scans, the X-streaming GEMM kernels
#include "kernels/host_small.inc"   // host-side launchers: FastICA step / tail, fp64 small-matrix ops, re-basing
#include "kernels/k_gram.inc"   // the split-product Gram kernel of the FastICA whitening (k_gram5) and its launcher
#include "kernels/host_pow3_eigh.inc"   // host-side launchers: the fused pass, Cholesky / eigen-solver / remaining small ops
}  // namespace petal
