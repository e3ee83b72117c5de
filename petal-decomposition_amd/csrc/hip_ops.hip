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

// ================================================================================================
// generic kernels (any shape, T = float | double, fp64 accumulation)
// ================================================================================================
template <class T>
__global__ void k_pack_strided(const T* __restrict__ src, int64_t n, int64_t d, int64_t rs, int64_t cs, T* __restrict__ dst,
                               int64_t ld, int64_t dpad) {
    const int64_t j = blockIdx.y * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.x;
    if (j >= dpad) return;
    dst[i * ld + j] = j < d ? src[i * rs + j * cs] : T(0);
}
template <class T>
__global__ void k_unpack_strided(const T* __restrict__ src, int64_t n, int64_t d, int64_t ld, T* __restrict__ dst, int64_t rs,
                                 int64_t cs) {
    const int64_t j = blockIdx.y * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.x;
    if (j >= d) return;
    dst[i * rs + j * cs] = src[i * ld + j];
}
// (flattened like k_unpack_scaled below: a device input whose width is not a multiple of 16 is copied into the padded layout)
template <class T>
__global__ __launch_bounds__(256) void k_pack_flat(const T* __restrict__ src, int64_t n, int64_t d, int64_t rs, int64_t cs, T* __restrict__ dst,
                                                   int64_t ld, int64_t dpad) {
    const int64_t total = n * dpad;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t e = ((int64_t)blockIdx.x * 4 + u) * 256 + threadIdx.x;
        if (e >= total) return;
        const int64_t i = e / dpad, j = e - i * dpad;
        dst[i * ld + j] = j < d ? src[i * rs + j * cs] : T(0);
    }
}
// the same over a flattened index -- a thread per element, consecutive threads along a row, 1024 elements per workgroup -- with
// an optional per-column factor (fit_transform's sigma_j and svd_flip sign, applied in fp64 as k_scale_cols does).  The row-per-
// workgroup form above launches n workgroups of mostly idle lanes: 1e6 x 64 took 0.25 ms for the scaling and the copy-out each.
template <class T>
__global__ __launch_bounds__(256) void k_unpack_scaled(const T* __restrict__ src, int64_t n, int64_t d, int64_t ld, T* __restrict__ dst,
                                                       int64_t rs, int64_t cs, const double* __restrict__ scale) {
    const int64_t total = n * d;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t e = ((int64_t)blockIdx.x * 4 + u) * 256 + threadIdx.x;
        if (e >= total) return;
        const int64_t i = e / d, j = e - i * d;
        const T v = src[i * ld + j];
        dst[i * rs + j * cs] = scale ? (T)((double)v * scale[j]) : v;
    }
}

// Column scans over a tall row-major matrix, first stage: block = 64 column lanes x 4 row lanes over `rows` rows (the
// loads of a thread are independent: many in flight), row lanes combined through LDS in fixed order.
constexpr int SCAN_RY = 4;
// rows per block of the column scans: ~256 row parts for wide matrices; a narrow one (few 64-column blocks) gets more parts so that
// the launch still puts ~2048 workgroups on the chip (1e6 x 64 on 256 workgroups streamed at 2.1 TB/s)
__host__ inline int64_t scan_rows_per_block(int64_t n, int64_t cols = 512) {
    const int64_t col_blocks = std::max<int64_t>(1, (cols + 63) / 64);
    const int64_t parts = std::min<int64_t>(1024, std::max<int64_t>(256, 2048 / col_blocks));   // (more parts cost the final combine more than they save)
    return std::max<int64_t>(256, (cdiv64(n, parts) + 3) / 4 * 4);
}
template <class T, bool SQ>  // SQ: also the column sums of squares, part = [nparts][2 d] = [sums | sums of squares]
__global__ __launch_bounds__(256) void k_colsum_part2(const T* __restrict__ X, int64_t n, int64_t d, int64_t ldx, int64_t rows,
                                                      double* __restrict__ part) {
    __shared__ double red[SCAN_RY][64], redq[SQ ? SCAN_RY : 1][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int64_t j = blockIdx.y * (int64_t)64 + cx;
    const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = min(n, r0 + rows);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (j < d) {
        int64_t i = r0 + ry;
        for (; i + 3 * SCAN_RY < r1; i += 4 * SCAN_RY) {
            const double v0 = (double)X[i * ldx + j], v1 = (double)X[(i + SCAN_RY) * ldx + j];
            const double v2 = (double)X[(i + 2 * SCAN_RY) * ldx + j], v3 = (double)X[(i + 3 * SCAN_RY) * ldx + j];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            if (SQ) { q0 += v0 * v0; q1 += v1 * v1; q2 += v2 * v2; q3 += v3 * v3; }
        }
        for (; i < r1; i += SCAN_RY) {
            const double v0 = (double)X[i * ldx + j];
            s0 += v0;
            if (SQ) q0 += v0 * v0;
        }
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    if (SQ) redq[ry][cx] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    if (ry == 0 && j < d) {
        const int64_t w = SQ ? 2 * d : d;
        part[(int64_t)blockIdx.x * w + j] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
        if (SQ) part[(int64_t)blockIdx.x * w + d + j] = (redq[0][cx] + redq[1][cx]) + (redq[2][cx] + redq[3][cx]);
    }
}
template <class T>
__global__ __launch_bounds__(256) void k_absmax_part2(const T* __restrict__ U, int64_t n, int64_t L, int64_t ldu, int64_t rows,
                                                      double* __restrict__ pmax, double* __restrict__ pidx,
                                                      double* __restrict__ psgn) {
    __shared__ double rm[SCAN_RY][64], ri[SCAN_RY][64], rs[SCAN_RY][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int64_t j = blockIdx.y * (int64_t)64 + cx;
    const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = min(n, r0 + rows);
    double best = -1.0, bi = INFINITY, bs = 1.0;
    if (j < L) {
#pragma unroll 4
        for (int64_t i = r0 + ry; i < r1; i += SCAN_RY) {  // ascending rows, strict '>' keeps the first maximum (pca.rs:830)
            const double v = (double)U[i * ldu + j], a = fabs(v);
            if (a > best) { best = a; bi = (double)i; bs = signbit(v) ? -1.0 : 1.0; }
        }
    }
    rm[ry][cx] = best; ri[ry][cx] = bi; rs[ry][cx] = bs;
    __syncthreads();
    if (ry == 0 && j < L) {
#pragma unroll
        for (int k = 1; k < SCAN_RY; ++k)
            if (rm[k][cx] > best || (rm[k][cx] == best && ri[k][cx] < bi)) { best = rm[k][cx]; bi = ri[k][cx]; bs = rs[k][cx]; }
        const int64_t o = (int64_t)blockIdx.x * L + j;
        pmax[o] = best; pidx[o] = bi; psgn[o] = bs;
    }
}
// out[(e / N) * ldc + e % N] = sum_p part[p * count + e]  (fp64 accumulation, fixed order => deterministic).
// block = 32 elements x 8 part-lanes.
template <class TP>
__global__ __launch_bounds__(256) void k_sum_parts2(const TP* __restrict__ part, int64_t nparts, int64_t count,
                                                    double* __restrict__ out, int64_t N, int64_t ldc, bool accumulate) {
    __shared__ double red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + tx;
    double s = 0;
    if (e < count)
        for (int64_t p = ty; p < nparts; p += 8) s += (double)part[p * count + e];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < count) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tx];
        double* o = out + (e / N) * ldc + (e % N);
        *o = accumulate ? *o + t : t;
    }
}
// the same for fp32 slabs whose count and row length are even: a block owns 128 consecutive outputs (512 B contiguous in
// every slab, one 8-B load per lane), its four waves each take a quarter of the slabs, eight loads in flight
__global__ __launch_bounds__(256) void k_sum_parts4(const float* __restrict__ part, int64_t nparts, int64_t count,
                                                    double* __restrict__ out, int64_t N, int64_t ldc) {
    __shared__ double red[3][64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = ((int64_t)blockIdx.x * 64 + lane) * 2;
    double s0 = 0, s1 = 0;
    if (e < count) {
        const float* src = part + e;
        int64_t p = wave;
        for (; p + 28 < nparts; p += 32) {
            f32x2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x2*>(src + (p + 4 * k) * count);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s0 += (double)v[k][0]; s1 += (double)v[k][1]; }
        }
        for (; p < nparts; p += 4) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(src + p * count);
            s0 += (double)v[0]; s1 += (double)v[1];
        }
    }
    if (wave > 0) { red[wave - 1][lane][0] = s0; red[wave - 1][lane][1] = s1; }
    __syncthreads();
    if (wave == 0 && e < count) {
#pragma unroll
        for (int w = 0; w < 3; ++w) { s0 += red[w][lane][0]; s1 += red[w][lane][1]; }
        double* o = out + (e / N) * ldc + (e % N);
        o[0] = s0; o[1] = s1;
    }
}
__global__ __launch_bounds__(256) void k_add_scalar_parts(const double* __restrict__ part, int64_t nparts, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0;
    for (int64_t p = threadIdx.x; p < nparts; p += 256) s += part[p];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += red[0];
}

template <class T>
__global__ void k_xp_simple(const T* __restrict__ X, int64_t n, int64_t K, int64_t ldx, const T* __restrict__ mu,
                            const double* __restrict__ P, int64_t N, int64_t ldp, const T* __restrict__ bias,
                            T* __restrict__ Z, int64_t ldz, double* __restrict__ ss_part) {
    const int64_t j = blockIdx.y * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.x;
    double acc = 0, ss = 0;
    const bool want_ss = ss_part != nullptr && blockIdx.y == 0 && threadIdx.x == 0;
    if (j < N || want_ss) {
        for (int64_t k = 0; k < K; ++k) {
            T xv = X[i * ldx + k];
            if (mu) xv = xv - mu[k];
            const double a = (double)xv;
            if (j < N) acc += a * (sizeof(T) == 4 ? (double)(float)P[k * ldp + j] : P[k * ldp + j]);
            ss += a * a;
        }
    }
    if (j < N) Z[i * ldz + j] = (T)(acc + (bias ? (double)bias[j] : 0.0));
    if (want_ss) ss_part[i] = ss;
}

constexpr int ATB_S_ROWS = 256;
template <class T>
__global__ void k_atb_simple(const T* __restrict__ A, int64_t lda, int64_t M, const T* __restrict__ muA,
                             const T* __restrict__ B, int64_t ldb, int64_t N, const T* __restrict__ muB, int64_t n,
                             double* __restrict__ part) {
    const int64_t j = blockIdx.y * (int64_t)blockDim.x + threadIdx.x;
    const int64_t m = blockIdx.z * (int64_t)blockDim.y + threadIdx.y;
    if (j >= N || m >= M) return;
    const int64_t r0 = (int64_t)blockIdx.x * ATB_S_ROWS, r1 = min(n, r0 + ATB_S_ROWS);
    const T ma = muA ? muA[m] : T(0), mb = muB ? muB[j] : T(0);
    double s = 0;
    for (int64_t i = r0; i < r1; ++i) s += (double)(T)(A[i * lda + m] - ma) * (double)(T)(B[i * ldb + j] - mb);
    part[((int64_t)blockIdx.x * M + m) * N + j] = s;
}
// one wave per column: lanes scan the row chunks in order (strict '>' keeps the first maximum, pca.rs:830),
// then a lexicographic (larger |u|, then smaller row) butterfly picks the winner.
__global__ __launch_bounds__(64) void k_absmax_final(const double* __restrict__ pmax, const double* __restrict__ pidx,
                                                     const double* __restrict__ psgn, int64_t nparts, int64_t L,
                                                     int64_t row_offset, double* __restrict__ omax, double* __restrict__ oidx,
                                                     double* __restrict__ osgn) {
    const int64_t j = blockIdx.x;
    const int lane = threadIdx.x;
    double best = -2.0, bi = INFINITY, bs = 1.0;
    for (int64_t p = lane; p < nparts; p += 64) {
        const double a = pmax[p * L + j];
        if (a > best) { best = a; bi = pidx[p * L + j]; bs = psgn[p * L + j]; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off, 64), oi = __shfl_down(bi, off, 64), os = __shfl_down(bs, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; bs = os; }
    }
    if (lane == 0) { omax[j] = best; oidx[j] = bi + (double)row_offset; osgn[j] = bs; }
}
template <class T>
__global__ void k_scale_cols(T* __restrict__ A, int64_t n, int64_t L, int64_t lda, const double* __restrict__ s) {
    const int64_t j = blockIdx.y * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.x;
    if (j < L) A[i * lda + j] = (T)((double)A[i * lda + j] * s[j]);
}
template <class T>
__global__ void k_logcosh_rows(const T* __restrict__ X, int64_t r, int64_t c, int64_t ldx, T* __restrict__ G, int64_t ldg,
                               double* __restrict__ gp) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= r) return;
    double s = 0;
    for (int64_t j = 0; j < c; ++j) {
        const double g = tanh((double)X[i * ldx + j]);
        G[i * ldg + j] = (T)g;
        s += 1.0 - g * g;
    }
    gp[i] = s;
}

// ================================================================================================
// K1: Z = (X - mu) . P + bias   -- fp32 MFMA, X streamed HBM -> VGPR -> MFMA
// ================================================================================================
// P is packed in B-fragment order: Ppk[((c * NTtot + nt) * 64 + lane) * 4 + s] = P[16 c + 4 (lane>>4) + s][16 nt + (lane&15)]
// so that one 16-B load per lane yields the B operands of the four MFMA k-steps of a 16-deep K chunk.
// The matching A fragment is one 16-B load per lane: X[row][16 c + 4 (lane>>4) + 0..3]  (k order inside the
// chunk is permuted identically on both operands).
__global__ void k_pack_p(const double* __restrict__ P, int64_t K, int64_t N, int64_t ldp, float* __restrict__ Ppk, int NTtot) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // one float4 slot per thread
    const int64_t total = (K / 16) * NTtot * 64;
    if (e >= total) return;
    const int lane = int(e & 63);
    const int64_t cn = e >> 6;
    const int nt = int(cn % NTtot);
    const int64_t c = cn / NTtot;
    const int q = lane >> 4, j = lane & 15;
    const int64_t col = 16 * (int64_t)nt + j;
    f32x4 v;
    for (int s = 0; s < 4; ++s) {
        const int64_t k = 16 * c + 4 * q + s;
        v[s] = (col < N && k < K) ? (float)P[k * ldp + col] : 0.0f;
    }
    reinterpret_cast<f32x4*>(Ppk)[e] = v;
}

// X is read exactly once per pass.  Non-temporal loads were tried here (to keep the re-read small operand in
// L2) and measured SLOWER on gfx950 (K1 at 1e6 x 512: 77 vs 88 TFLOP/s, FETCH_SIZE up 40 %), so plain loads stay.
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
#ifdef PETAL_NT_LOADS
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}

template <int RT, int NT, bool CENTER, bool SUMSQ>
__global__ __launch_bounds__(256, 2) void k_xp_mfma(const float* __restrict__ X, int64_t n, int K, int64_t ldx,
                                                    const float* __restrict__ mu, const float* __restrict__ Ppk, int NTtot,
                                                    int nt0, int N, const float* __restrict__ bias, float* __restrict__ Z,
                                                    int64_t ldz, double* __restrict__ ss_part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);
    if (row0 >= n) {
        if (SUMSQ && lane == 0) ss_part[(int64_t)blockIdx.x * 4 + wave] = 0.0;
        return;
    }
    f32x4 acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xrow[RT];
    bool rvalid[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t r = row0 + 16 * t + i;
        rvalid[t] = r < n;
#ifdef PETAL_EXP_AFIXED
        xrow[t] = X + (int64_t)(16 * t + i) * ldx + 4 * q;  // experiment: every wave re-reads the same 64 rows (cache resident)
#else
        xrow[t] = X + (rvalid[t] ? r : (n - 1)) * ldx + 4 * q;
#endif
    }
    const f32x4* pb = reinterpret_cast<const f32x4*>(Ppk) + (int64_t)nt0 * 64 + lane;
    const float* mup = mu + 4 * q;
    float ssq = 0.f;
    const int nchunk = K >> 4;

    // one 16-deep K chunk: RT + NT (+1) independent 16-B loads per lane, then 4 RT NT MFMAs
    auto load_chunk = [&](int c, f32x4(&a)[RT], f32x4(&b)[NT], f32x4& m) {
#pragma unroll
        for (int t = 0; t < RT; ++t) a[t] = ld_stream(xrow[t] + 16 * c);
#pragma unroll
#ifdef PETAL_EXP_BFIXED
        for (int u = 0; u < NT; ++u) b[u] = pb[((int64_t)(c & 1) * NTtot + u) * 64];  // experiment: P chunk always cache resident
#else
        for (int u = 0; u < NT; ++u) b[u] = pb[((int64_t)c * NTtot + u) * 64];
#endif
        if (CENTER) m = *reinterpret_cast<const f32x4*>(mup + 16 * c);
    };
    auto compute = [&](f32x4(&a)[RT], f32x4(&b)[NT], const f32x4& m) {
        if (CENTER) {
#pragma unroll
            for (int t = 0; t < RT; ++t) a[t] -= m;
        }
        if (SUMSQ) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
                if (rvalid[t]) ssq += a[t][0] * a[t][0] + a[t][1] * a[t][1] + a[t][2] * a[t][2] + a[t][3] * a[t][3];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][s], a[t][s], acc[t][u], 0, 0, 0);  // Z^T tile
    };
    // software pipeline, THREE register stages: the loads of chunks c+1 and c+2 are in flight while chunk c feeds the
    // MFMAs (two chunks of MFMA time, ~2 x 2560 cycles per wave, cover the loaded HBM latency at 2 waves / SIMD).
    // sched_barrier(0) pins the issue order (hipcc otherwise sinks the prefetch below the MFMAs and drains vmcnt(0)).
    f32x4 a0[RT], b0[NT], a1[RT], b1[NT], a2[RT], b2[NT];
    f32x4 m0 = f32x4{0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0;
    const int last = nchunk - 1;
    load_chunk(0, a0, b0, m0);
    load_chunk(last < 1 ? last : 1, a1, b1, m1);
    int c = 0;
    for (; c + 3 <= nchunk; c += 3) {
        load_chunk(c + 2 < last ? c + 2 : last, a2, b2, m2);
        __builtin_amdgcn_sched_barrier(0);
        compute(a0, b0, m0);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(c + 3 < last ? c + 3 : last, a0, b0, m0);
        __builtin_amdgcn_sched_barrier(0);
        compute(a1, b1, m1);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(c + 4 < last ? c + 4 : last, a1, b1, m1);
        __builtin_amdgcn_sched_barrier(0);
        compute(a2, b2, m2);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c < nchunk) compute(a0, b0, m0);
    if (c + 1 < nchunk) compute(a1, b1, m1);
    // epilogue.  The MFMA operands are swapped (P fragment as A, X fragment as B), so the accumulator tile is Z^T:
    // reg r of lane (i, q) is Z[row0 + 16 t + i][16 u + 4 q + r] -- one 16-B store per lane and tile instead of four
    // 4-B stores (the store tail is issue-bound: -10 % kernel time at 1e6 x 512).
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col = 16 * (nt0 + u) + 4 * q;
        if (col >= N) continue;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int64_t row = row0 + 16 * t + i;
#ifdef PETAL_EXP_NOSTORE
            if (row < n && acc[t][u][0] == 12345.678f)
#else
            if (row < n)
#endif
                *reinterpret_cast<f32x4*>(Z + row * ldz + col) = acc[t][u] + bv;
        }
    }
    if (SUMSQ) {
        double sred = (double)ssq;
        for (int off = 32; off > 0; off >>= 1) sred += __shfl_down(sred, off, 64);
        if (lane == 0) ss_part[(int64_t)blockIdx.x * 4 + wave] = sred;
    }
}

// ------------------------------------------------------------------------------------------------
// One register tile of K1 as a device function (used by the persistent form below): RT row-tiles x NT column tiles,
// one pass over K with a two- or three-stage register pipeline, 16-B epilogue stores.
// (A one-wave-per-SIMD form with 128-row register tiles built on it was measured slower and removed.)
template <int RT, int NT, bool CENTER, bool SUMSQ, int STAGES = 3>
__device__ __forceinline__ void xp_tile(const float* __restrict__ X, int64_t n, int K, int64_t ldx, const float* __restrict__ mu,
                                        const f32x4* __restrict__ pb, int NTtot, int nt0, int N, const float* __restrict__ bias,
                                        float* __restrict__ Z, int64_t ldz, int64_t row0, int lane, float& ssq) {
    const int i = lane & 15, q = lane >> 4;
    f32x4 acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xrow[RT];
    bool rvalid[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t r = row0 + 16 * t + i;
        rvalid[t] = r < n;
        xrow[t] = X + (rvalid[t] ? r : (n - 1)) * ldx + 4 * q;
    }
    const float* mup = mu + 4 * q;
    const int nchunk = K >> 4;
    auto load_chunk = [&](int c, f32x4(&a)[RT], f32x4(&b)[NT], f32x4& m) {
#pragma unroll
        for (int t = 0; t < RT; ++t) a[t] = ld_stream(xrow[t] + 16 * c);
#pragma unroll
        for (int u = 0; u < NT; ++u) b[u] = pb[((int64_t)c * NTtot + u) * 64];
        if (CENTER) m = *reinterpret_cast<const f32x4*>(mup + 16 * c);
    };
    auto compute = [&](f32x4(&a)[RT], f32x4(&b)[NT], const f32x4& m) {
        if (CENTER) {
#pragma unroll
            for (int t = 0; t < RT; ++t) a[t] -= m;
        }
        if (SUMSQ) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
                if (rvalid[t]) ssq += a[t][0] * a[t][0] + a[t][1] * a[t][1] + a[t][2] * a[t][2] + a[t][3] * a[t][3];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][s], a[t][s], acc[t][u], 0, 0, 0);  // Z^T tile
    };
    const int last = nchunk - 1;
    if constexpr (STAGES == 3) {
        f32x4 a0[RT], b0[NT], a1[RT], b1[NT], a2[RT], b2[NT];
        f32x4 m0 = f32x4{0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0;
        load_chunk(0, a0, b0, m0);
        load_chunk(last < 1 ? last : 1, a1, b1, m1);
        int c = 0;
        for (; c + 3 <= nchunk; c += 3) {
            load_chunk(c + 2 < last ? c + 2 : last, a2, b2, m2);
            __builtin_amdgcn_sched_barrier(0);
            compute(a0, b0, m0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(c + 3 < last ? c + 3 : last, a0, b0, m0);
            __builtin_amdgcn_sched_barrier(0);
            compute(a1, b1, m1);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(c + 4 < last ? c + 4 : last, a1, b1, m1);
            __builtin_amdgcn_sched_barrier(0);
            compute(a2, b2, m2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < nchunk) compute(a0, b0, m0);
        if (c + 1 < nchunk) compute(a1, b1, m1);
    } else {
        f32x4 a0[RT], b0[NT], a1[RT], b1[NT];
        f32x4 m0 = f32x4{0.f, 0.f, 0.f, 0.f}, m1 = m0;
        load_chunk(0, a0, b0, m0);
        int c = 0;
        for (; c + 2 <= nchunk; c += 2) {
            load_chunk(c + 1, a1, b1, m1);
            __builtin_amdgcn_sched_barrier(0);
            compute(a0, b0, m0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(c + 2 < last ? c + 2 : last, a0, b0, m0);
            __builtin_amdgcn_sched_barrier(0);
            compute(a1, b1, m1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < nchunk) compute(a0, b0, m0);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {  // Z^T accumulator tiles: one 16-B store per lane and tile (see k_xp_mfma)
        const int col = 16 * (nt0 + u) + 4 * q;
        if (col >= N) continue;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int64_t row = row0 + 16 * t + i;
            if (row < n) *reinterpret_cast<f32x4*>(Z + row * ldz + col) = acc[t][u] + bv;
        }
    }
}

// K1, persistent form (default): ONE 512-thread workgroup per CU (two waves per SIMD: waves w and w + 4 share one).
// All W = 8 * gridDim.x waves sweep the matrix together: in round j wave g owns the 64-row tile (j W + g), so the rows
// in flight at any time form one contiguous window (TLB / DRAM-page friendly, like a plain grid launch).  The last,
// partial round deals the remaining 16-row tiles out in balanced 0..4-tile pieces, the longer pieces to waves 0..3 of
// each workgroup (four different SIMDs): every SIMD ends within one 16-row tile of the average (7 vs 6.1 tiles at
// 100000 rows; the 64-row-per-wave grid form leaves it at 8).
#ifndef PETAL_PERS_STAGES
#define PETAL_PERS_STAGES 2
#endif
template <int NT, bool CENTER, bool SUMSQ>
__global__ __launch_bounds__(512, 2) void k_xp_pers(const float* __restrict__ X, int64_t n, int K, int64_t ldx,
                                                    const float* __restrict__ mu, const float* __restrict__ Ppk, int NTtot,
                                                    int nt0, int N, const float* __restrict__ bias, float* __restrict__ Z,
                                                    int64_t ldz, double* __restrict__ ss_part, int64_t ntiles16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* pb = reinterpret_cast<const f32x4*>(Ppk) + (int64_t)nt0 * 64 + lane;
    float ssq = 0.f;
    const int64_t W = (int64_t)gridDim.x * 8, g = (int64_t)blockIdx.x * 8 + wave;
    const int64_t nfull = ntiles16 / (4 * W);  // rounds in which every wave owns a full 64-row tile
    int64_t t = 0;
#define XP_TILE(R, S) xp_tile<R, NT, CENTER, SUMSQ, S>(X, n, K, ldx, mu, pb, NTtot, nt0, N, bias, Z, ldz, 16 * t, lane, ssq)
#ifdef PETAL_EXP_STAGGER
    {   // experiment: de-phase the workgroups (and the two waves of a SIMD) so the chip's load bursts do not coincide
        const int dly = (int)(blockIdx.x & 15) * 5 + (wave >= 4 ? 40 : 0);
        for (int k = 0; k < dly; ++k) __builtin_amdgcn_s_sleep(1);
    }
#endif
    for (int64_t j = 0; j < nfull; ++j) {
        t = (j * W + g) * 4;
        XP_TILE(4, PETAL_PERS_STAGES);
    }
    {
        const int64_t rem = ntiles16 - 4 * W * nfull;  // < 4 W tiles of 16 rows
        const int64_t base = rem / W, extra = rem - base * W;
        const int64_t rank = wave < 4 ? (int64_t)blockIdx.x * 4 + wave : W / 2 + (int64_t)blockIdx.x * 4 + (wave - 4);
        const int64_t cnt = base + (rank < extra ? 1 : 0);
        t = 4 * W * nfull + rank * base + (rank < extra ? rank : extra);
        if (cnt == 4) XP_TILE(4, PETAL_PERS_STAGES);
        else if (cnt == 3) XP_TILE(3, 2);
        else if (cnt == 2) XP_TILE(2, 2);
        else if (cnt == 1) XP_TILE(1, 2);
    }
#undef XP_TILE
    if (SUMSQ) {
        double sred = (double)ssq;
        for (int off = 32; off > 0; off >>= 1) sred += __shfl_down(sred, off, 64);
        if (lane == 0) ss_part[(int64_t)blockIdx.x * 8 + wave] = sred;
    }
}

// ================================================================================================
// K1, split-product form ("bf16x3"): every fp32 operand is split EXACTLY into three bf16 pieces, x = x_h + x_m + x_l
// (8 + 8 + 8 significant bits), and the product x p is formed from the six piece products of weight >= 2^-16,
//   x_h p_h + (x_h p_m + x_m p_h) + (x_h p_l + x_m p_m + x_l p_h),
// on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16: exact bf16 x bf16 products, fp32 accumulation).  The three dropped
// terms are <= 2^-24 |x p| relative -- below the rounding of ONE fp32 multiply -- so the result matches the fp32-MFMA
// kernel to within fp32 accumulation noise (measured: 4e-8 of the mean |z| vs 3e-6 accumulation error of a plain fp32
// GEMM at K = 512), while the matrix pipe does 6 bf16 MFMAs of 16 cycles per 16 x 16 x 32 block instead of 8 fp32 MFMAs of
// 32: 2.7x less matrix time, which turns K1 from MFMA-bound into HBM-bound.
// Layout: one wave = 64 rows x NT column tiles, K in 32-deep chunks.  X goes HBM -> VGPR (two 16-B loads per lane and
// row tile: X[row][32 c + 8 q .. + 7], which IS the 16x16x32 operand layout), is centred and split in registers; the
// three planes of P are pre-split by k_pack_p3 and staged chunk by chunk through LDS by the workgroup (double buffered,
// one barrier per chunk) so the four waves share them.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// written pair-wise so hipcc emits packed converts (12), shifts / masks (16) and packed subtracts (8) per 8 elements;
// the generic vector conversion costs one convert per ELEMENT
#ifndef PETAL_STEER_PIECES
#define PETAL_STEER_PIECES 4   // piece products of a STEERING product (both operands on two planes): 4, or 3 = without x_m p_m.  That term is 2^-18 of
                               // x_h p_h, the size of what the pass has dropped already -- but not below it: with three pieces configs[1] runs 0.96-0.98 ms
                               // for 0.99-1.02 and the converged fits keep their errors (2.8e-6 / 2.9e-6 / 3.0e-6 with three / four pieces / exact passes at
                               // 100000 x 512), while the LAST component of a fit that has not converged loses a factor two to three (20000 x 512, n_iter 3:
                               // 3.4e-5 / 1.1e-5 / 1.5e-5; 20000 x 1024, k = 128, n_iter 7: 1.04e-5 / 7.4e-6 / 7.2e-6).  Measured, not taken.
#endif
#ifndef PETAL_SPLIT_DOT2
#define PETAL_SPLIT_DOT2 1
#endif
// x - bf16 piece, for the two elements of a packed pair: one v_dot2c_f32_bf16 each ({h0, h1} . {-1, 0} + x0 and {h0, h1} . {0, -1} + x1;
// the products and the sum are exact, dev/micro_dot2.hip) instead of a shift or mask to unpack the piece and a packed subtract:
// 28 VALU instructions per eight elements instead of 36.  The selector constants sit in SGPRs (laundered: an inline constant on a
// packed-bf16 operand would have to mean the same to the assembler and to the hardware).
__device__ __forceinline__ void sub_pk_bf16(float& x0, float& x1, unsigned pk) {
#if PETAL_SPLIT_DOT2
    unsigned c0u, c1u;   // (not volatile: no inputs, so the two moves are common to every call site of a kernel and leave its loops)
    asm("s_mov_b32 %0, 0xbf80" : "=s"(c0u));
    asm("s_mov_b32 %0, 0xbf800000" : "=s"(c1u));
    x0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, c0u), x0, false);
    x1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, c1u), x1, false);
#else
    x0 -= __uint_as_float(pk << 16);
    x1 -= __uint_as_float(pk & 0xffff0000u);
#endif
}
__device__ __forceinline__ void split3(const f32x8 x, bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 hh, mm, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r0 = x[2 * e], r1 = x[2 * e + 1];
        hh[e] = cvt_pk_bf16(r0, r1);
        sub_pk_bf16(r0, r1, hh[e]);
        mm[e] = cvt_pk_bf16(r0, r1);
        sub_pk_bf16(r0, r1, mm[e]);
        ll[e] = cvt_pk_bf16(r0, r1);
    }
    h = __builtin_bit_cast(bf16x8, hh);
    m = __builtin_bit_cast(bf16x8, mm);
    l = __builtin_bit_cast(bf16x8, ll);
}
// the two leading pieces only (16 significant bits): the operands of the STEERING passes (k_pow3f, k_xp3<.., X2>, k_atb3<.., P4>)
__device__ __forceinline__ void split2(const f32x8 x, bf16x8& h, bf16x8& m) {
    u32x4 hh, mm;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r0 = x[2 * e], r1 = x[2 * e + 1];
        hh[e] = cvt_pk_bf16(r0, r1);
        sub_pk_bf16(r0, r1, hh[e]);
        mm[e] = cvt_pk_bf16(r0, r1);
    }
    h = __builtin_bit_cast(bf16x8, hh);
    m = __builtin_bit_cast(bf16x8, mm);
}
// Ppk3[((c NTtot + nt) 3 + plane) 64 + lane][e] = plane of (float)P[32 c + 8 (lane >> 4) + e][16 nt + (lane & 15)]
__global__ __launch_bounds__(256) void k_pack_p3(const double* __restrict__ P, int64_t K, int64_t N, int64_t ldp,
                                                 bf16x8* __restrict__ out, int NTtot, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t tile = idx >> 6;
    const int nt = (int)(tile % NTtot);
    const int64_t c = tile / NTtot;
    const int64_t col = 16 * nt + (lane & 15), k0 = 32 * c + 8 * (lane >> 4);
    f32x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (k0 + e < K && col < N) ? (float)P[(k0 + e) * ldp + col] : 0.f;
    bf16x8 h, m, l;
    split3(x, h, m, l);
    out[(tile * 3 + 0) * 64 + lane] = h;
    out[(tile * 3 + 1) * 64 + lane] = m;
    out[(tile * 3 + 2) * 64 + lane] = l;
}
#ifndef PETAL_XP3_RT
#define PETAL_XP3_RT 4      // row tiles per wave
#define PETAL_XP3_DEPTH 1   // raw X chunks in flight per wave (2 measured slower)
#define PETAL_XP3_OCC 2     // waves per SIMD the register budget is cut for
#endif
// NPL = planes of P that take part: 3, or 2 when the caller DEFINED P as the sum of its two leading bf16 pieces (the re-based
// iterate of the power iteration, k_trsm_pack<NB, true>): the product x_h p_l has nothing to multiply then -- five piece products
// instead of six, and a third less of P to stage through LDS.
// X2 (needs NPL = 2; a STEERING pass, see k_pow3f): X too is rounded to its two leading pieces -- four piece products per tile.
template <int RT, int NT, int DEPTH, bool CENTER, int WVK = 4, int OCC = PETAL_XP3_OCC, int NPL = 3, bool X2 = false>  // WVK = waves (row tiles of 16 RT rows) per workgroup
__global__ __launch_bounds__(64 * WVK, OCC) void k_xp3(const float* __restrict__ X, int64_t n, int K, int64_t ldx,
                                                            const float* __restrict__ mu, const bf16x8* __restrict__ Ppk3,
                                                            int NTtot, int nt0, int N, const float* __restrict__ bias,
                                                            float* __restrict__ Z, int64_t ldz, double* __restrict__ amax,
                                                            int64_t am_ld) {
    constexpr int PITEMS = NT * 64 * NPL;          // 16-B items of one P chunk (NT tiles x NPL planes x 64 lanes)
    constexpr int NTHR = 64 * WVK;
    constexpr int PI = (PITEMS + NTHR - 1) / NTHR;  // per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_xp3[];
    bf16x8* sP = reinterpret_cast<bf16x8*>(sm_xp3);                        // [2][PITEMS]
    float* sMu = reinterpret_cast<float*>(sm_xp3 + sizeof(bf16x8) * 2 * PITEMS);  // [32 nchunk] (zero padded)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * WVK + wave) * (16 * RT);
    const int nchunk = (K + 31) >> 5;
    if (CENTER)
        for (int k = tid; k < 32 * nchunk; k += NTHR) sMu[k] = k < K ? mu[k] : 0.f;
    f32x4 acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xrow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t r = row0 + 16 * t + i;
        xrow[t] = X + (r < n ? r : (n - 1)) * ldx + 8 * q;
    }
    // item j of the LDS image = (tile u, plane pl < NPL, lane): u = j / (64 NPL); in memory the tile keeps three planes
    const bf16x8* psrc = Ppk3 + (int64_t)nt0 * 192;
    int poff[PI];
#pragma unroll
    for (int it = 0; it < PI; ++it) {
        const int j = tid + NTHR * it;
        poff[it] = NPL == 3 ? j : (j / (64 * NPL)) * 192 + (j % (64 * NPL));
    }
    auto load_a = [&](int c, f32x8(&a)[RT]) {
        const bool in = 32 * c + 8 * q < K;  // K % 32 == 16: the upper half of the last chunk does not exist
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
            if (in) { lo = ld_stream(xrow[t] + 32 * c); hi = ld_stream(xrow[t] + 32 * c + 4); }
            a[t] = f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    auto load_p = [&](int c, bf16x8(&pn)[PI]) {
#pragma unroll
        for (int it = 0; it < PI; ++it)
            if (tid + NTHR * it < PITEMS) pn[it] = psrc[(int64_t)c * NTtot * 192 + poff[it]];
    };
    auto store_p = [&](int buf, const bf16x8(&pn)[PI]) {
#pragma unroll
        for (int it = 0; it < PI; ++it)
            if (tid + NTHR * it < PITEMS) sP[buf * PITEMS + tid + NTHR * it] = pn[it];
    };
    f32x8 a[DEPTH][RT];
    bf16x8 pn[PI];
    load_p(0, pn);
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load_a(s < nchunk ? s : nchunk - 1, a[s]);
    store_p(0, pn);
#ifdef PETAL_DEBUG_COUNTERS
    long long ph[6] = {0, 0, 0, 0, 0, 0};
    long long tq = __builtin_amdgcn_s_memtime();
#define XP3_STAMP(i) do { const long long _t = __builtin_amdgcn_s_memtime(); ph[i] += _t - tq; tq = _t; } while (0)
#else
#define XP3_STAMP(i) do {} while (0)
#endif
    for (int c0 = 0; c0 < nchunk; c0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int c = c0 + s;
            if (c >= nchunk) break;
            const int buf = c & 1;
            __syncthreads();  // chunk c is in sP[buf]; nobody still reads sP[buf ^ 1]
            XP3_STAMP(0);
#ifdef PETAL_DEBUG_COUNTERS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            XP3_STAMP(1);
#endif
            // the raw fragments are dead once split: their registers take the loads of chunk c + DEPTH, which then fly
            // under the MFMAs of DEPTH chunks
            bf16x8 ah[RT], am[RT], al[RT];
            {
                f32x8 m = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (CENTER) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(sMu + 32 * c + 8 * q), hi = *reinterpret_cast<const f32x4*>(sMu + 32 * c + 8 * q + 4);
                    m = f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    if (CENTER) a[s][t] -= m;
                    if (X2) { split2(a[s][t], ah[t], am[t]); al[t] = am[t]; }
                    else split3(a[s][t], ah[t], am[t], al[t]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            XP3_STAMP(2);
            // P first: its registers are waited for at the end of this chunk, and vmcnt retires in order -- X loads issued
            // BEFORE it would be drained by that wait, the ones issued after it stay in flight
            if (c + 1 < nchunk) load_p(c + 1, pn);
            if (c + DEPTH < nchunk) load_a(c + DEPTH, a[s]);
            __builtin_amdgcn_sched_barrier(0);
            XP3_STAMP(3);
            // P fragments one tile ahead of the MFMAs that use them (pinned: hoisting all 15 reads costs 48 more registers)
            const bf16x8* sPb = sP + buf * PITEMS + lane;
            bf16x8 bh = sPb[0], bm = sPb[64], bl = NPL == 3 ? sPb[128] : bm;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                bf16x8 nh = bh, nm = bm, nl = bl;
                if (u + 1 < NT) {
                    nh = sPb[((u + 1) * NPL) * 64]; nm = sPb[((u + 1) * NPL + 1) * 64];
                    if (NPL == 3) nl = sPb[((u + 1) * NPL + 2) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < RT; ++t) {  // P fragment as the A operand: the accumulator tile is Z^T (16-B stores below)
                    f32x4 c4 = acc[t][u];
                    if (NPL == 3) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[t], c4, 0, 0, 0);   // smallest terms first
                    if (!X2 || PETAL_STEER_PIECES >= 4) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am[t], c4, 0, 0, 0);
                    if (!X2) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[t], c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah[t], c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am[t], c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[t], c4, 0, 0, 0);
                    acc[t][u] = c4;
                }
                __builtin_amdgcn_sched_barrier(0);
                bh = nh; bm = nm; bl = nl;
            }
            __builtin_amdgcn_sched_barrier(0);
            XP3_STAMP(4);
            if (c + 1 < nchunk) store_p(buf ^ 1, pn);
            XP3_STAMP(5);
        }
    }
#ifdef PETAL_DEBUG_COUNTERS
    if (lane == 0) {
        for (int e = 0; e < 6; ++e) atomicAdd((unsigned long long*)&g_cyc[20 + e], (unsigned long long)ph[e]);
        atomicAdd((unsigned long long*)&g_cyc[26], 1ull);
    }
#endif
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int col = 16 * (nt0 + u) + 4 * q;
        if (col >= N) continue;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int64_t row = row0 + 16 * t + i;
            if (Z && row < n) *reinterpret_cast<f32x4*>(Z + row * ldz + col) = acc[t][u] + bv;   // (Z == nullptr: only the scan below is wanted)
        }
    }
    if (amax) {
        // svd_flip's scan (pca.rs:826-839: per column the first row of largest |u|) on the accumulators instead of a pass over the
        // stored product: one partial (max, row, sign) per workgroup and column, k_absmax_part2's format, reduced by
        // k_absmax_final.  Ordering key: |u|'s bits above (lowest row first, sign) -- a plain unsigned maximum.
        unsigned long long* sK = reinterpret_cast<unsigned long long*>(sm_xp3);   // [WVK][NT 16]
        __syncthreads();   // (the last chunk's P stage is dead)
        constexpr int NG = (NT * 4 + 15) / 16;     // column slots of a lane (NT tiles x 4), in groups of 16
        unsigned long long key[NG * 16];
#pragma unroll
        for (int sl = 0; sl < NG * 16; ++sl) key[sl] = 0;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            const int col = 16 * (nt0 + u) + 4 * q;
            if (bias && col < N) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const float v = acc[t][u][e] + bv[e];
                    const int lrow = wave * (16 * RT) + 16 * t + i;
                    if (row0 + 16 * t + i < n && v == v) {
                        const unsigned long long kk = ((unsigned long long)__float_as_uint(fabsf(v)) << 32) |
                                                      ((unsigned long long)(unsigned)(0x7fffffff - lrow) << 1) | (__float_as_uint(v) >> 31);
                        key[u * 4 + e] = kk > key[u * 4 + e] ? kk : key[u * 4 + e];
                    }
                }
            }
        }
        // maximum over the 16 row lanes of each slot as a transposing butterfly (15 exchanges per group of 16 slots instead of
        // 64): at distance h a lane keeps the half of its slots that matches its bit h and hands over the other half; after four
        // steps lane i holds slot i of the group
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
                for (int sl = 0; sl < h; ++sl) {
                    const unsigned long long a = key[16 * g + sl], b = key[16 * g + sl + h];
                    const bool up = (i & h) != 0;
                    const unsigned long long o = __shfl_xor(up ? a : b, h, 64), keep = up ? b : a;
                    key[16 * g + sl] = o > keep ? o : keep;
                }
            }
            const int sl = 16 * g + i;                      // = 4 u + e
            if (sl < NT * 4) sK[wave * (NT * 16) + (sl >> 2) * 16 + 4 * q + (sl & 3)] = key[16 * g];
        }
        __syncthreads();
        for (int c = tid; c < NT * 16; c += NTHR) {
            unsigned long long key = sK[c];
#pragma unroll
            for (int w = 1; w < WVK; ++w) { const unsigned long long o = sK[w * (NT * 16) + c]; key = o > key ? o : key; }
            const int col = 16 * nt0 + c;
            if (col < am_ld) {
                const int64_t o = (int64_t)blockIdx.x * am_ld + col, plane = (int64_t)gridDim.x * am_ld;
                const bool any = key != 0;
                const int lrow = 0x7fffffff - (int)((key & 0xffffffffull) >> 1);
                amax[o] = any ? (double)__uint_as_float((unsigned)(key >> 32)) : -1.0;
                amax[plane + o] = any ? (double)((int64_t)blockIdx.x * (WVK * 16 * RT) + lrow) : (double)INFINITY;
                amax[2 * plane + o] = (any && (key & 1)) ? -1.0 : 1.0;
            }
        }
    }
}

// ================================================================================================
// K2: C = (A - muA)^T . (B - muB)  -- split-K fp32 MFMA over row chunks, fp32 partial slabs
// ================================================================================================
// A wave owns 64 columns of A (4 m-tiles) x 16*NT columns of B and a contiguous row chunk.  One 16-B load
// per lane of A gives the A fragments of the 4 m-tiles of a 4-row k-step (tile t, row i  <->  m = m0 + 4 i + t);
// 16-B loads of B give col tiles in groups of four (tile 4 g + e, col j  <->  col = 64 g + 4 j + e) plus
// scalar loads for the NT % 4 remaining tiles.
template <int NT, bool CA, bool CB>
__global__ __launch_bounds__(256, 2) void k_atb_mfma(const float* __restrict__ A, int64_t lda, int M, const float* __restrict__ muA,
                                                     const float* __restrict__ B, int64_t ldb, int N, int n0col,
                                                     const float* __restrict__ muB, int64_t n, int64_t chunk,
                                                     float* __restrict__ part, int Npart) {
    constexpr int G4 = NT / 4, R1 = NT % 4;
    constexpr int G4n = G4 > 0 ? G4 : 1, R1n = R1 > 0 ? R1 : 1;
    constexpr int UN = 4;  // k-steps (of 4 rows) per pipeline stage: 16 rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    // XCD-aware launch (see k_atb3): grid = (8 hx, ceil(nsplit / 8)); the column groups of one row chunk share an XCD
    const int bx = blockIdx.x >> 3, by = blockIdx.y * 8 + (blockIdx.x & 7);
    if ((int64_t)by * chunk >= n) return;
    const int m0 = (bx * 4 + wave) * 64;
    if (m0 >= M) return;
    const int64_t rbeg = (int64_t)by * chunk, rend = min(n, rbeg + chunk);
    // Columns beyond M / N are never stored, so their loads are simply clamped into the row (any finite value
    // will do); rows beyond rend must contribute nothing: only the ragged tail stage masks them (B := 0).
    const float* ap = A + min(m0 + 4 * i, M - 4);
    const float* bp4[G4n];
    const float* bp1[R1n];
#pragma unroll
    for (int g = 0; g < G4; ++g) bp4[g] = B + min(n0col + 64 * g + 4 * i, N - 4);
#pragma unroll
    for (int e = 0; e < R1; ++e) bp1[e] = B + min(n0col + 64 * G4 + 16 * e + i, N - 1);
    f32x4 ma = f32x4{0.f, 0.f, 0.f, 0.f};
    if (CA) ma = *reinterpret_cast<const f32x4*>(muA + min(m0 + 4 * i, M - 4));
    f32x4 mb4[G4n];
    float mb1[R1n];
    if (CB) {
#pragma unroll
        for (int g = 0; g < G4; ++g) mb4[g] = *reinterpret_cast<const f32x4*>(muB + min(n0col + 64 * g + 4 * i, N - 4));
#pragma unroll
        for (int e = 0; e < R1; ++e) mb1[e] = muB[min(n0col + 64 * G4 + 16 * e + i, N - 1)];
    }
    f32x4 acc[4][NT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_stage = [&](int64_t r0, f32x4(&av)[UN], f32x4(&b4)[UN][G4n], float(&b1)[UN][R1n]) {
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            const int64_t r = r0 + 4 * s + q;
            av[s] = ld_stream(ap + r * lda);
#pragma unroll
            for (int g = 0; g < G4; ++g) b4[s][g] = *reinterpret_cast<const f32x4*>(bp4[g] + r * ldb);
#pragma unroll
            for (int e = 0; e < R1; ++e) b1[s][e] = bp1[e][r * ldb];
        }
    };
    auto compute_stage = [&](f32x4(&av)[UN], f32x4(&b4)[UN][G4n], float(&b1)[UN][R1n]) {
#pragma unroll
        for (int s = 0; s < UN; ++s) {
            if (CA) av[s] -= ma;
            if (CB) {
#pragma unroll
                for (int g = 0; g < G4; ++g) b4[s][g] -= mb4[g];
#pragma unroll
                for (int e = 0; e < R1; ++e) b1[s][e] -= mb1[e];
            }
        }
#pragma unroll
        for (int s = 0; s < UN; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int g = 0; g < G4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[t][4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], b4[s][g][e], acc[t][4 * g + e], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < R1; ++e)
                    acc[t][4 * G4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], b1[s][e], acc[t][4 * G4 + e], 0, 0, 0);
            }
    };
    // software pipeline over the full 16-row stages, two register stages (see k_xp_mfma)
    const int64_t nfull = (rend - rbeg) / (4 * UN);
    f32x4 av0[UN], b40[UN][G4n], av1[UN], b41[UN][G4n];
    float b10[UN][R1n], b11[UN][R1n];
    if (nfull > 0) {
        load_stage(rbeg, av0, b40, b10);
        int64_t st = 0;
        for (; st + 2 <= nfull; st += 2) {
            load_stage(rbeg + (st + 1) * 4 * UN, av1, b41, b11);
            __builtin_amdgcn_sched_barrier(0);
            compute_stage(av0, b40, b10);
            __builtin_amdgcn_sched_barrier(0);
            load_stage(rbeg + (st + 2 < nfull ? st + 2 : nfull - 1) * 4 * UN, av0, b40, b10);
            __builtin_amdgcn_sched_barrier(0);
            compute_stage(av1, b41, b11);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (st < nfull) compute_stage(av0, b40, b10);
    }
    {   // ragged tail (< 16 rows): rows past rend are read from a clamped row and masked to zero on the B side
        const int64_t r0 = rbeg + nfull * 4 * UN;
        for (int64_t rs = r0; rs < rend; rs += 4) {
            const int64_t r = rs + q;
            const bool rv = r < rend;
            const int64_t rc = rv ? r : rend - 1;
            f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + rc * lda);
            if (CA) a1 -= ma;
#pragma unroll
            for (int g = 0; g < G4; ++g) {
                f32x4 bv = *reinterpret_cast<const f32x4*>(bp4[g] + rc * ldb);
                if (CB) bv -= mb4[g];
                if (!rv) bv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[t][4 * g + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], bv[e], acc[t][4 * g + e], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < R1; ++e) {
                float bv = bp1[e][rc * ldb];
                if (CB) bv -= mb1[e];
                if (!rv) bv = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t][4 * G4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], bv, acc[t][4 * G4 + e], 0, 0, 0);
            }
        }
    }
    // D[row = 4 q + r][col = i] of tile (t, u):  m = m0 + 4 (4 q + r) + t;  col from the B grouping
    float* out = part + (int64_t)by * M * Npart;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * (4 * q + r) + t;
            if (m >= M) continue;
#pragma unroll
            for (int g = 0; g < G4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = n0col + 64 * g + 4 * i + e;
                    if (col < N) out[(int64_t)m * Npart + col] = acc[t][4 * g + e][r];
                }
#pragma unroll
            for (int e = 0; e < R1; ++e) {
                const int col = n0col + 64 * G4 + 16 * e + i;
                if (col < N) out[(int64_t)m * Npart + col] = acc[t][4 * G4 + e][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// K2, split-product form (see k_xp3 for the arithmetic): C = (A - muA)^T B over a row chunk, K = rows in 32-row stages.
// A wave owns 64 columns of A (4 m-tiles; lane (i, q) loads A[r0 + 8 q + e][m0 + 4 i .. + 3], e < 8: eight 16-B loads
// feed the eight k-values of all four tiles) and ALL NT column tiles of B.  The 32 x 16 NT stage of B is shared by the
// four waves: the workgroup converts it once -- thread <-> one operand item (tile u, lane (j, q): B[r0 + 8 q + e][16 u + j],
// e < 8), split into three bf16 planes -- into a double-buffered LDS image in MFMA operand order, one barrier per stage.
// n must be a multiple of 32 here: the host gives the ragged remainder to the fp32 kernel as one more slab.
// WV = waves per workgroup (4 or 8).  Every wave owns its own 64 columns of A but all of them share the workgroup's
// 32-row stage of B, which is read from memory and converted ONCE per workgroup: with 4-wave workgroups a 512-column A
// makes two workgroups per row chunk read and convert the same stage (K2's PMC traffic was 1.16-1.23x algorithmic, most
// of it this); 8-wave workgroups cover 512 columns with one.
// P4 (a STEERING pass, see k_pow3f): A and B are both rounded to their two leading pieces -- four piece products per tile instead of six.
template <int NT, bool CA, int WV, int MT = 4, bool P4 = false>  // MT = column tiles of A per wave (4: 64 columns, 16-B loads; 2: 32 columns, 8-B loads)
__global__ __launch_bounds__(64 * WV, 2) void k_atb3(const float* __restrict__ A, int64_t lda, int M, const float* __restrict__ muA,
                                                     const float* __restrict__ B, int64_t ldb, int N, int n0col, int64_t n,
                                                     int64_t chunk, float* __restrict__ part, int Npart, int n_tail) {
    constexpr int BITEMS = NT * 64;                       // operand items of one B stage (8 elements each)
    constexpr bool Z2 = BITEMS > 64 * WV;                 // the first NT - WV waves carry a second item (tile WV + wave)
    typedef float fvecm __attribute__((ext_vector_type(MT)));
    __shared__ bf16x8 sB[2][NT * 192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    // XCD-aware launch: grid = (8 hx, ceil(nsplit / 8)) with hx = column groups.  Workgroups are dealt to the 8 XCDs round-robin
    // by linear id, so the hx groups of ONE row chunk (blockIdx.x = s, s + 8, ...) share an XCD and its L2: the chunk's Z stage
    // is fetched once, not once per column group.
    const int bx = blockIdx.x >> 3, by = blockIdx.y * 8 + (blockIdx.x & 7);
    if ((int64_t)by * chunk >= n) return;   // (the last group of eight may be short; uniform per workgroup)
    const int m0 = (bx * WV + wave) * (16 * MT);
    const int64_t rbeg = (int64_t)by * chunk, rend = min(n, rbeg + chunk);
    const int nfull = (int)((rend - rbeg) >> 5);
    // n is a multiple of 32; the n_tail (< 32) ragged rows behind it ride along as ONE partial stage of the last row chunk:
    // its rows past the end are loaded from the last valid row (finite) on the A side and as zeros on the B side, so they add
    // nothing.  (Round 2 gave them to the fp32 kernel as an extra slab: two 8-us launches per product at 250000 rows.)
    const int tail = (rend == n) ? n_tail : 0;
    const int nstage = nfull + (tail > 0 ? 1 : 0);
    // uniform row bases (advance 32 rows per stage) + ONE 32-bit per-lane offset each: no per-load address arithmetic
    const float* abase = A + rbeg * lda;
    const float* zbase = B + rbeg * ldb;
    const unsigned aoff = (unsigned)((int64_t)(8 * q) * lda + min(m0 + MT * i, M - MT));  // columns beyond M are never stored
    const int zcol0 = n0col + 16 * wave + i, zcol1 = n0col + 16 * (WV + wave) + i;  // items tid and 64 WV + tid
    const bool z2w = Z2 && WV + wave < NT;                // this wave converts a second item
    const bool zon0 = wave < NT && zcol0 < N, zon1 = z2w && zcol1 < N;
    const unsigned zoff0 = (unsigned)((int64_t)(8 * q) * ldb + min(zcol0, N - 1));
    const unsigned zoff1 = (unsigned)((int64_t)(8 * q) * ldb + min(zcol1, N - 1));
    fvecm ma = fvecm(0.f);
    if (CA) ma = *reinterpret_cast<const fvecm*>(muA + min(m0 + MT * i, M - MT));
    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The ragged stage is PEELED out of the steady-state loop: the loop below sees whole 32-row stages only (a uniform base + one
    // per-lane offset per load), and the partial stage of the last row chunk runs behind it, unpipelined, with its own clamped
    // loads.  Round 3 selected between the two address forms with `st < nfull` INSIDE the loaders: both forms then stayed live
    // across the software-pipelined loop, 128 -> 190 VGPRs at <5, *, 8, 2> (4 -> 2 waves per SIMD) and 14 spilled registers at
    // <5, *, 8, 4> (tests/test_kernel_budgets.py now holds the budgets).
    auto load_a = [&](int st, fvecm(&av)[8]) {
        const float* p = abase + (int64_t)st * 32 * lda;
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = *reinterpret_cast<const fvecm*>(p + (int64_t)e * lda + aoff);
    };
    auto load_z1 = [&](int st, f32x8& zr, unsigned zoff) {
        const float* p = zbase + (int64_t)st * 32 * ldb;
#pragma unroll
        for (int e = 0; e < 8; ++e) zr[e] = p[(int64_t)e * ldb + zoff];  // always a valid address (column clamped)
    };
    auto stage_z1 = [&](int buf, f32x8 zr, int u, bool on) {
        if (!on) zr = f32x8{0, 0, 0, 0, 0, 0, 0, 0};  // columns beyond N contribute zeros
        bf16x8 h, m, l;
        if (P4) split2(zr, h, m); else split3(zr, h, m, l);
        sB[buf][(u * 3 + 0) * 64 + lane] = h;
        sB[buf][(u * 3 + 1) * 64 + lane] = m;
        if (!P4) sB[buf][(u * 3 + 2) * 64 + lane] = l;
    };
    auto split_a = [&](fvecm(&av)[8], bf16x8(&ah)[MT], bf16x8(&am)[MT], bf16x8(&al)[MT]) {
        if (CA) {  // centred in place (a centred COPY would keep 32 more registers alive across the four splits)
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] -= ma;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const f32x8 x = f32x8{av[0][t], av[1][t], av[2][t], av[3][t], av[4][t], av[5][t], av[6][t], av[7][t]};
            if (P4) { split2(x, ah[t], am[t]); al[t] = am[t]; }
            else split3(x, ah[t], am[t], al[t]);
        }
    };
    auto mfma_stage = [&](int buf, int lane, const bf16x8(&ah)[MT], const bf16x8(&am)[MT], const bf16x8(&al)[MT]) {
        // B fragments one tile ahead of the MFMAs that use them (pinned: hoisting all 15 reads costs 48 more registers)
        bf16x8 bh = sB[buf][0 * 64 + lane], bm = sB[buf][1 * 64 + lane], bl = P4 ? bm : sB[buf][2 * 64 + lane];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            bf16x8 nh = bh, nm = bm, nl = bl;
            if (u + 1 < NT) { nh = sB[buf][(u * 3 + 3) * 64 + lane]; nm = sB[buf][(u * 3 + 4) * 64 + lane]; nl = P4 ? nm : sB[buf][(u * 3 + 5) * 64 + lane]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                f32x4 c4 = acc[t][u];
                if (!P4) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh, c4, 0, 0, 0);   // smallest terms first
                if (!P4 || PETAL_STEER_PIECES >= 4) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t], bm, c4, 0, 0, 0);
                if (!P4) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl, c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t], bh, c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bm, c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh, c4, 0, 0, 0);
                acc[t][u] = c4;
            }
            __builtin_amdgcn_sched_barrier(0);
            bh = nh; bm = nm; bl = nl;
        }
    };
    fvecm av[8];
    f32x8 zr0, zr1 = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (nfull > 0) {
        if (wave < NT) load_z1(0, zr0, zoff0);
        if (z2w) load_z1(0, zr1, zoff1);
        load_a(0, av);
        if (wave < NT) stage_z1(0, zr0, wave, zon0);
        if (z2w) stage_z1(0, zr1, WV + wave, zon1);
    }
#ifdef PETAL_DEBUG_COUNTERS
    long long ph3[6] = {0, 0, 0, 0, 0, 0};
    long long tq3 = __builtin_amdgcn_s_memtime();
#define ATB3_STAMP(i) do { const long long _t = __builtin_amdgcn_s_memtime(); ph3[i] += _t - tq3; tq3 = _t; } while (0)
#else
#define ATB3_STAMP(i) do {} while (0)
#endif
    for (int st = 0; st < nfull; ++st) {
        const int buf = st & 1;
        __syncthreads();  // stage st of B is in sB[buf]; nobody still reads sB[buf ^ 1]
        ATB3_STAMP(0);
        bf16x8 ah[MT], am[MT], al[MT];
        split_a(av, ah, am, al);
        __builtin_amdgcn_sched_barrier(0);
        ATB3_STAMP(1);
        if (st + 1 < nfull) {  // fly under this stage's MFMAs; B first (vmcnt retires in order, see k_xp3)
            if (wave < NT) load_z1(st + 1, zr0, zoff0);
            if (z2w) load_z1(st + 1, zr1, zoff1);
            load_a(st + 1, av);
        }
        __builtin_amdgcn_sched_barrier(0);
        ATB3_STAMP(2);
        mfma_stage(buf, lane, ah, am, al);
        __builtin_amdgcn_sched_barrier(0);
        ATB3_STAMP(3);
        if (st + 1 < nfull) {
            if (wave < NT) stage_z1(buf ^ 1, zr0, wave, zon0);
            if (z2w) stage_z1(buf ^ 1, zr1, WV + wave, zon1);
        }
        ATB3_STAMP(4);
    }
    if (tail > 0) {  // the partial stage (uniform branch; only the last row chunk of a launch has one): nothing is in flight here
        // Every per-lane quantity of this block is re-derived from a laundered copy of the thread index, so that none of them
        // is a live range across the pipelined loop above (they cost 4-6 VGPRs there: 132 instead of 128 at <5, *, 8, 2>).
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int lane2 = tid2 & 63, wave2 = tid2 >> 6, i2 = lane2 & 15, q2 = lane2 >> 4;
        const int buf = nfull & 1;   // the buffer the last full stage did NOT read
        const float* zp = zbase + (int64_t)nfull * 32 * ldb;
        const float* ap = abase + (int64_t)nfull * 32 * lda + min((bx * WV + wave2) * (16 * MT) + MT * i2, M - MT);
        auto tail_z = [&](int u) {
            const int zcol = n0col + 16 * u + i2;
            f32x8 zr;
#pragma unroll
            for (int e = 0; e < 8; ++e) zr[e] = (8 * q2 + e < tail && zcol < N) ? zp[(int64_t)(8 * q2 + e) * ldb + min(zcol, N - 1)] : 0.f;
            bf16x8 h, m, l;
            if (P4) split2(zr, h, m); else split3(zr, h, m, l);
            sB[buf][(u * 3 + 0) * 64 + lane2] = h;
            sB[buf][(u * 3 + 1) * 64 + lane2] = m;
            if (!P4) sB[buf][(u * 3 + 2) * 64 + lane2] = l;
        };
        if (wave2 < NT) tail_z(wave2);
        if (Z2 && WV + wave2 < NT) tail_z(WV + wave2);
        // rows past the end -> the last valid row (finite; their B side is zero)
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = *reinterpret_cast<const fvecm*>(ap + (int64_t)min(8 * q2 + e, tail - 1) * lda);
        bf16x8 ah[MT], am[MT], al[MT];
        split_a(av, ah, am, al);
        __syncthreads();
        mfma_stage(buf, lane2, ah, am, al);
    }
#ifdef PETAL_DEBUG_COUNTERS
    if (lane == 0) {
        for (int e = 0; e < 5; ++e) atomicAdd((unsigned long long*)&g_cyc[10 + e], (unsigned long long)ph3[e]);
        atomicAdd((unsigned long long*)&g_cyc[15], (unsigned long long)nstage);
    }
#endif
    // D[row = 4 q + r][col = i] of tile (t, u):  m = m0 + MT (4 q + r) + t,  col = n0col + 16 u + i
    float* out = part + (int64_t)by * M * Npart;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + MT * (4 * q + r) + t;
            if (m >= M) continue;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int col = n0col + 16 * u + i;
                if (col < N) out[(int64_t)m * Npart + col] = acc[t][u][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// K3 "k_pow3": one power-iteration pass  Y' = Xc^T (Xc P)  (pca.rs:711 + 714) with ONE pass over X -- the two products K1 and K2
// form from two, and no Z in between (round 5).  For K = 512 features and N = 16 NT <= 80 columns; P is the re-based iterate
// DEFINED on two bf16 planes (k_trsm_pack<NB, true>) or a two-plane sketch matrix, so a wave's slice of it fits its registers.
//
// One persistent 8-wave workgroup per CU walks over 32-row stages of X.  Wave w owns the features [64 w, 64 w + 64) in BOTH
// products:
//   * its slice of P sits in registers for the whole launch as MFMA B operands (2 chunks x NT tiles x 2 planes = 80 VGPRs at
//     NT = 5), its slice of the Y' accumulators likewise (4 x NT tiles = 80 VGPRs);
//   * it loads its 32 x 64 piece of the stage in K1's layout (lane (i, q): row 16 t + i, features 32 c + 8 q .. + 7: two 16-B
//     loads), one stage ahead, centres and splits it ONCE into three bf16 planes -- which feed product 1 from registers and go to
//     a wave-private LDS image that product 2 reads back TRANSPOSED with ds_read_b64_tr_b16 (its A operand wants feature i with
//     eight samples per lane): X goes through the vector-memory pipeline once per power iteration;
//   * product 1 gives the wave a PARTIAL z (its 64 of the 512 features) per 16-row half of the stage, in accumulator layout
//     (lane (i, q): samples 4 q + r, column 16 u + i).  The eight partials meet in LDS; wave u < NT adds them for column tile u --
//     after both halves a lane holds the eight samples {4 q + r, 16 + 4 q + r} of one column, which IS a B-operand fragment of
//     product 2 once its k-slots are declared to be those samples (the X side reads its transposed blocks at the same rows) --
//     splits the sum into three planes and publishes the fragment; rows beyond n are zeroed here (their X rows are clamped
//     loads of the last row), and the last pass of a fit stores z as the iterate Z;
//   * product 2 accumulates Y'[64 w .. + 64][:] += Xc^T z over the stage (six piece products, smallest first, as K2).
// LDS: X planes 8 x 12 KB (16-B chunks XOR-swizzled by the row so that the row writes and the transposed reads are both
// conflict-free on 128-B rows), partial z 8 x NT KB, z fragments 3 NT KB, mu 2 KB: 153 KB at NT = 5 -- one workgroup per CU, two
// waves per SIMD, <= 256 registers.  Four barriers per stage (the partial-z buffer holds ONE half: there is no room for two).
// Every workgroup ends with its own 512 x 16 NT fp32 slab of Y', combined in fp64 in a fixed order by k_sum_parts4 like K2's.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 lds_tr2(const unsigned char* a0, const unsigned char* a1) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a1));
    return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
// byte offset of 16-B chunk `ch` (eight features) of row `row` inside one plane of a wave's image ([32 rows][128 B])
__device__ __forceinline__ int pow3_xoff(int row, int ch) { return row * 128 + 16 * (ch ^ ((2 * ((row >> 1) & 3)) ^ (row & 1))); }
// MEANS (the FIRST pass of a fit, round 5): `mu` is only a provisional centre mu0 (the means of a strided row sample), and the pass
// gathers what the separate means pass used to: the LAST column of the last tile -- zero padding of P, so z is 0 there -- is set
// to 1 for every valid row, which makes Y'[:, 16 NT - 1] = Xc0^T 1 the column sums about mu0 (product 2 forms them for free), and
// the splits accumulate sum (x - mu0)^2 over the valid rows (one partial per wave in ssq_part).  k_mean_fix then moves everything
// to the true centre mu = mu0 + delta, delta = sums / n:  Xc^T Xc P = Xc0^T Xc0 P - n delta (delta^T P),  sum (x - mu)^2 = ssq - n |delta|^2
// -- a rank-one correction of relative size (delta / sigma)^2, i.e. harmless, where the same identity about mu0 = 0 would cancel
// (mu / sigma)^2 of the product.
template <int NT, bool CENTER, bool STOREZ, bool MEANS = false>
__global__ __launch_bounds__(512) void k_pow3(const float* __restrict__ X, int64_t n, int64_t ldx, const float* __restrict__ mu,
                                              const bf16x8* __restrict__ Ppk3, int NTtot, float* __restrict__ part,
                                              float* __restrict__ Z, int64_t ldz, int64_t nstages, double* __restrict__ ssq_part) {
    constexpr int WV = 8, K = 512, XIMG = 3 * 32 * 128;          // bytes of one wave's plane image
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_pow3[];
    unsigned char* const sX = sm_pow3;                                           // [WV][3][32][128 B]
    f32x4* const sZp = reinterpret_cast<f32x4*>(sm_pow3 + WV * XIMG);            // [WV][NT][64]
    bf16x8* const sZB = reinterpret_cast<bf16x8*>(sm_pow3 + WV * XIMG + WV * NT * 1024);   // [NT][3][64]
    float* const sMu = reinterpret_cast<float*>(sm_pow3 + WV * XIMG + WV * NT * 1024 + NT * 3072);   // [K]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);           // (uniform: scalar registers)
    // Register budget: two waves per SIMD leave 256, and P (80) + Y' (80) + the raw stage in flight (32) are fixed.  Every per-lane
    // index below is therefore re-derived, phase by phase, from a LAUNDERED copy of the thread index -- left to itself hipcc hoists
    // some thirty loop-invariant LDS offsets and pointers into registers of their own and spills P fragments to pay for them
    // (each reload then waits behind the stage's X loads: vmcnt retires in order -- 7300 cycles for a 1300-cycle phase, measured).
#define POW3_LANE(ln) int ln = threadIdx.x & 63; asm volatile("" : "+v"(ln))
#ifndef PETAL_POW3_ZPF
#define PETAL_POW3_ZPF 0
#endif
#ifndef PETAL_POW3_DEPHASE
#define PETAL_POW3_DEPHASE 0
#endif
    if (CENTER)
        for (int k = threadIdx.x; k < K; k += 64 * WV) sMu[k] = mu[k];
    // this wave's slice of P: chunk c <-> features 64 wave + 32 c .. + 32, B-operand fragments of the packed planes (k_pack_p3's
    // layout: lane (j, q) holds P[32 c' + 8 q + e][16 u + j])
    bf16x8 ph[2][NT], pm[2][NT];
    {
        POW3_LANE(ln);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const bf16x8* src = Ppk3 + (((int64_t)(2 * wave + c) * NTtot + u) * 3) * 64 + ln;
                ph[c][u] = src[0];
                pm[c][u] = src[64];
            }
    }
    f32x4 acc2[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc2[m][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stage range of this workgroup (contiguous, balanced to one stage)
    const int64_t s0 = (int64_t)blockIdx.x * nstages / gridDim.x, s1 = (int64_t)(blockIdx.x + 1) * nstages / gridDim.x;
    f32x8 xa[2][2];
    auto load_x = [&](int64_t s) {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t r = min(s * 32 + 16 * t + li, n - 1);   // rows beyond n: the last row (finite); their z is zeroed below
            const float* p = X + r * ldx + 64 * wave + 8 * lq;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 lo = ld_stream(p + 32 * c), hi = ld_stream(p + 32 * c + 4);
                xa[t][c] = f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
    };
    if (s0 < s1) load_x(s0);
    unsigned char* const myX = sX + wave * XIMG;
    // product 1 of one 16-row half: operands read back from the wave's image (rows 16 h + i), X fragment as A (rows = samples), P
    // fragment as B; five piece products per tile, smallest first
    auto prod1 = [&](f32x4(&acc1)[NT], int h) {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned char* a = myX + pow3_xoff(16 * h + li, 4 * c + lq);
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(a), xm = *reinterpret_cast<const bf16x8*>(a + 4096),
                         xl = *reinterpret_cast<const bf16x8*>(a + 8192);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                f32x4 c4 = acc1[u];
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, ph[c][u], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, pm[c][u], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, ph[c][u], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, pm[c][u], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, ph[c][u], c4, 0, 0, 0);
                acc1[u] = c4;
            }
        }
    };
    auto park_partials = [&](const f32x4(&acc1)[NT]) {
        POW3_LANE(ln);
#pragma unroll
        for (int u = 0; u < NT; ++u) sZp[(wave * NT + u) * 64 + ln] = acc1[u];
    };
    // wave u < NT adds the eight partials of column tile u (fixed order)
    auto add_partials = [&]() {
        POW3_LANE(ln);
        f32x4 zs = sZp[(0 * NT + wave) * 64 + ln];
#pragma unroll
        for (int w = 1; w < WV; ++w) zs += sZp[(w * NT + wave) * 64 + ln];
        return zs;
    };
    // the raw fragments of feature chunk c (both row tiles): centred, split into three planes, parked in the wave's image; one
    // piece at a time (two splits in flight cost 28 registers)
    float ssq = 0.f;
    auto split_park = [&](int c, int64_t sidx) {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x8 x = xa[t][c];
            if (CENTER) {
                const float* mp = sMu + 64 * wave + 32 * c + 8 * lq;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(mp), hi = *reinterpret_cast<const f32x4*>(mp + 4);
                x -= f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (MEANS) {   // sum (x - mu0)^2 over the valid rows (rows beyond n are clamped loads of the last row)
                float q2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) q2 = fmaf(x[e], x[e], q2);
                ssq += sidx * 32 + 16 * t + li < n ? q2 : 0.f;
            }
            bf16x8 xh, xm, xl;
            split3(x, xh, xm, xl);
            unsigned char* a = myX + pow3_xoff(16 * t + li, 4 * c + lq);
            *reinterpret_cast<bf16x8*>(a) = xh;
            *reinterpret_cast<bf16x8*>(a + 4096) = xm;
            *reinterpret_cast<bf16x8*>(a + 8192) = xl;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    __syncthreads();   // mu
    // Software pipeline over the stages: the planes of stage s + 1 are produced UNDER product 2 of stage s -- its feature tile m
    // reads chunks 2 m, 2 m + 1 of every row, so once the fragments of tiles 0, 1 are in registers the image's first half (chunks
    // 0 .. 3) is free for the next stage's c = 0 pieces, and after those of tiles 2, 3 the second half: the split (180 VALU instructions and twelve
    // 16-B LDS stores per wave and stage) runs beside the other wave's MFMAs instead of in front of everybody's (phase stamps,
    // round 5: the split at the head of the stage was 1400 of its 16000 cycles with the matrix pipe idle).
    if (s0 < s1) {
        split_park(0, s0);
        split_park(1, s0);
        if (s0 + 1 < s1) load_x(s0 + 1);
    }
#ifdef PETAL_DEBUG_COUNTERS
    long long php[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tqp = __builtin_amdgcn_s_memtime();
#define POW3_STAMP(i) do { const long long _t = __builtin_amdgcn_s_memtime(); php[i] += _t - tqp; tqp = _t; \
        if (blockIdx.x == 7 && s == s0 + 5 && (threadIdx.x & 63) == 0) g_trace[wave * 16 + (i)] = _t; } while (0)
#else
#define POW3_STAMP(i) do {} while (0)
#endif
    for (int64_t s = s0; s < s1; ++s) {
        f32x4 zlo = f32x4{0.f, 0.f, 0.f, 0.f}, zhi = zlo;
        {
            f32x4 acc1[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            prod1(acc1, 0);
            POW3_STAMP(0);
            park_partials(acc1);                 // (ordered behind the previous stage's adders by that stage's last barrier)
            __syncthreads();
            POW3_STAMP(1);
            if (wave < NT) zlo = add_partials();
            POW3_STAMP(2);
        }
        {
            f32x4 acc1[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            prod1(acc1, 1);
            POW3_STAMP(3);
            __syncthreads();                     // the adders have read the first half's partials
            park_partials(acc1);
            __syncthreads();
            POW3_STAMP(4);
            if (wave < NT) zhi = add_partials();
        }
        if (wave < NT) {
            // lane (i, q): z[row 4 q + r][16 wave + i] (zlo), z[16 + 4 q + r][...] (zhi); rows beyond n are zeroed
            POW3_LANE(ln);
            const int li = ln & 15, lq = ln >> 4;
            const int64_t rb = s * 32 + 4 * lq;
            f32x8 z8;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z8[r] = rb + r < n ? zlo[r] : 0.f;
                z8[4 + r] = rb + 16 + r < n ? zhi[r] : 0.f;
            }
            if (MEANS && wave == NT - 1 && li == 15) {   // the all-ones column: Y'[:, 16 NT - 1] = Xc0^T 1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z8[r] = rb + r < n ? 1.f : 0.f;
                    z8[4 + r] = rb + 16 + r < n ? 1.f : 0.f;
                }
            }
            if (STOREZ) {
                float* zp = Z + rb * ldz + 16 * wave + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (rb + r < n) zp[r * ldz] = z8[r];
                    if (rb + 16 + r < n) zp[(16 + r) * ldz] = z8[4 + r];
                }
            }
            bf16x8 zh, zm, zl;
            split3(z8, zh, zm, zl);
            sZB[(wave * 3 + 0) * 64 + ln] = zh;
            sZB[(wave * 3 + 1) * 64 + ln] = zm;
            sZB[(wave * 3 + 2) * 64 + ln] = zl;
        }
        POW3_STAMP(5);
        __syncthreads();                         // the z fragments of this stage are published
        POW3_STAMP(6);
        // product 2: Y'[16 m + 4 q + r][16 u + i] += sum over the stage's samples; A = X^T (transposed reads), B = z fragments.
        // Transposed-read addresses (T10): lane 16 g + 4 q' + p of group g supplies block row q', columns 4 p .. 4 p + 3; the block
        // of k-slots e < 4 is rows 4 g + q', of e >= 4 rows 16 + 4 g + q' (g = q: the lane group IS the k group of the operand).
        {
            POW3_LANE(ln);
            const int trq = (ln >> 2) & 3, trp = ln & 3, lq = ln >> 4;
            // the twelve MFMAs per column tile of one pair of feature tiles (one set of z fragments; with ZPF the next tile's set is
            // fetched under them -- the raw registers of the pieces already split are free by then)
            auto mfma_pair = [&](const bf16x8(&ax)[2][3], int mp) {
#if PETAL_POW3_ZPF
                bf16x8 zc[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) zc[pl] = sZB[(0 * 3 + pl) * 64 + ln];
#endif
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
#if PETAL_POW3_ZPF
                    const bf16x8 zh = zc[0], zm = zc[1], zl = zc[2];
                    if (u + 1 < NT) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) zc[pl] = sZB[((u + 1) * 3 + pl) * 64 + ln];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#else
                    // (ONE set of z fragments, re-read for each pair of feature tiles: its latency is covered by the twelve MFMAs of the
                    // other wave of the SIMD)
                    const bf16x8 zh = sZB[(u * 3 + 0) * 64 + ln], zm = sZB[(u * 3 + 1) * 64 + ln], zl = sZB[(u * 3 + 2) * 64 + ln];
#endif
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        f32x4 c4 = acc2[2 * mp + mm][u];
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][2], zh, c4, 0, 0, 0);   // smallest terms first
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][1], zm, c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][0], zl, c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][1], zh, c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][0], zm, c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][0], zh, c4, 0, 0, 0);
                        acc2[2 * mp + mm][u] = c4;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {     // feature tiles in PAIRS: one set of z fragments feeds twelve MFMAs
                bf16x8 ax[2][3];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const unsigned char* a0 = myX + pow3_xoff(4 * lq + trq, 2 * (2 * mp + mm) + (trp >> 1)) + 8 * (trp & 1);   // (row + 16: + 2048, same swizzle)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) ax[mm][pl] = lds_tr2(a0 + pl * 4096, a0 + pl * 4096 + 2048);
                }
                // this half of the image has been read for the last time: the next stage's pieces go there.  The two waves of a SIMD
                // (w, w + 4) take the split and the MFMAs in OPPOSITE order (DEPHASE), so that one's VALU run meets the other's MFMAs
                // instead of its VALU run
#if PETAL_POW3_DEPHASE
                const bool split_first = wave < 4;
#else
                const bool split_first = true;
#endif
                if (split_first && s + 1 < s1) {
                    __builtin_amdgcn_sched_barrier(0);
                    split_park(mp, s + 1);
                }
                mfma_pair(ax, mp);
                if (!split_first && s + 1 < s1) {
                    __builtin_amdgcn_sched_barrier(0);
                    split_park(mp, s + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // (the next stage's first barrier orders these reads of the z fragments and of the partial buffer before they are rewritten)
        if (s + 2 < s1) load_x(s + 2);           // the raw registers are free again: three quarters of a stage ahead of their use
        POW3_STAMP(7);
    }
#ifdef PETAL_DEBUG_COUNTERS
    if ((threadIdx.x & 63) == 0) {
        for (int e = 0; e < 8; ++e) atomicAdd((unsigned long long*)&g_cyc[e], (unsigned long long)php[e]);
        atomicAdd((unsigned long long*)&g_cyc[8], (unsigned long long)(s1 - s0));
    }
#endif
    if (MEANS) {         // one partial of sum (x - mu0)^2 per wave, lanes added in a fixed order
        float v = ssq;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) ssq_part[(int64_t)blockIdx.x * WV + wave] = (double)v;
    }
    // this workgroup's slab: D[row = 4 q + r][col = i] of tile (m, u) is Y'[64 wave + 16 m + 4 q + r][16 u + i]
    {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
        float* out = part + (int64_t)blockIdx.x * K * (16 * NT);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* row = out + (int64_t)(64 * wave + 16 * m + 4 * lq + r) * (16 * NT);
#pragma unroll
                for (int u = 0; u < NT; ++u) row[16 * u + li] = acc2[m][u][r];
            }
    }
#undef POW3_LANE
}

// ---- k_pow3f: the fused pass for the INTERMEDIATE power iterations (late round 5).  A pass whose result only steers the iteration --
// every pass but the last one of a fit, whose Y' and Z become B and Q -- may round its operands like the iterate already is: X (after
// centring) and z to 16 significant bits, two bf16 planes each.  That is a perturbation of the same kind and size as the two-plane P
// (numpy model dev/x2_model.py: component errors against the oracle 1.1e-7 -> 3.8e-7 on planted spectra, unchanged on the slowly decaying
// ones, where the spectral verdict sends the fit to the exact pipeline anyway), and it changes the kernel's budget: FOUR piece
// products per tile in both products instead of five and six (160 MFMAs per wave and stage instead of 220), a third less split work, an
// X image of 64 KB instead of 96 -- which leaves room for the partial z of BOTH 16-row halves (80 KB), so a stage has TWO barriers
// instead of four: product 1 of both halves, park, barrier, add + publish z, barrier, product 2.
#define POW3_LANE(ln) int ln = threadIdx.x & 63; asm volatile("" : "+v"(ln))
#ifndef PETAL_POW3F_ZPF
#define PETAL_POW3F_ZPF 0
#endif
#ifndef PETAL_POW3F_DEPHASE
#define PETAL_POW3F_DEPHASE 0
#endif
// MEANS: as k_pow3's (the first pass of a fit about a provisional centre: the last column of z set to one gathers the column sums, the
// splits accumulate sum (x - mu0)^2 -- from the values BEFORE their rounding; the sums themselves are those of the 16-bit values, off the
// exact ones by 2^-17 sigma / sqrt(n) per column: 1e-8 sigma at the 200000 rows the fold starts from).
template <int NT, bool CENTER, bool MEANS = false>
__global__ __launch_bounds__(512) void k_pow3f(const float* __restrict__ X, int64_t n, int64_t ldx, const float* __restrict__ mu,
                                               const bf16x8* __restrict__ Ppk3, int NTtot, float* __restrict__ part, int64_t nstages,
                                               double* __restrict__ ssq_part) {
    constexpr int WV = 8, K = 512, XIMG = 2 * 32 * 128;          // bytes of one wave's two-plane image
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_pow3[];
    unsigned char* const sX = sm_pow3;                                           // [WV][2][32][128 B]
    f32x4* const sZp = reinterpret_cast<f32x4*>(sm_pow3 + WV * XIMG);            // [2][WV][NT][64]
    bf16x8* const sZB = reinterpret_cast<bf16x8*>(sm_pow3 + WV * XIMG + 2 * WV * NT * 1024);   // [NT][2][64]
    float* const sMu = reinterpret_cast<float*>(sm_pow3 + WV * XIMG + 2 * WV * NT * 1024 + NT * 2048);   // [K]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (CENTER)
        for (int k = threadIdx.x; k < K; k += 64 * WV) sMu[k] = mu[k];
    bf16x8 ph[2][NT], pm[2][NT];
    {
        POW3_LANE(ln);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const bf16x8* src = Ppk3 + (((int64_t)(2 * wave + c) * NTtot + u) * 3) * 64 + ln;
                ph[c][u] = src[0];
                pm[c][u] = src[64];
            }
    }
    f32x4 acc2[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc2[m][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t s0 = (int64_t)blockIdx.x * nstages / gridDim.x, s1 = (int64_t)(blockIdx.x + 1) * nstages / gridDim.x;
    f32x8 xa[2][2];
    auto load_x = [&](int64_t s) {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t r = min(s * 32 + 16 * t + li, n - 1);   // rows beyond n: the last row (finite); their z is zeroed below
            const float* p = X + r * ldx + 64 * wave + 8 * lq;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 lo = ld_stream(p + 32 * c), hi = ld_stream(p + 32 * c + 4);
                xa[t][c] = f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
    };
    if (s0 < s1) load_x(s0);
    unsigned char* const myX = sX + wave * XIMG;
    auto prod1 = [&](f32x4(&acc1)[NT], int h) {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned char* a = myX + pow3_xoff(16 * h + li, 4 * c + lq);
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(a), xm = *reinterpret_cast<const bf16x8*>(a + 4096);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                f32x4 c4 = acc1[u];
#if PETAL_STEER_PIECES >= 4
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, pm[c][u], c4, 0, 0, 0);   // smallest terms first
#endif
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, ph[c][u], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, pm[c][u], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, ph[c][u], c4, 0, 0, 0);
                acc1[u] = c4;
            }
        }
    };
    auto park_partials = [&](const f32x4(&acc1)[NT], int h) {
        POW3_LANE(ln);
#pragma unroll
        for (int u = 0; u < NT; ++u) sZp[((h * WV + wave) * NT + u) * 64 + ln] = acc1[u];
    };
    auto add_partials = [&](int h) {     // wave u < NT adds the eight partials of column tile u (fixed order)
        POW3_LANE(ln);
        f32x4 zs = sZp[((h * WV + 0) * NT + wave) * 64 + ln];
#pragma unroll
        for (int w = 1; w < WV; ++w) zs += sZp[((h * WV + w) * NT + wave) * 64 + ln];
        return zs;
    };
    float ssq = 0.f;
    auto split_park = [&](int c, int64_t sidx) {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x8 x = xa[t][c];
            if (CENTER) {
                const float* mp = sMu + 64 * wave + 32 * c + 8 * lq;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(mp), hi = *reinterpret_cast<const f32x4*>(mp + 4);
                x -= f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (MEANS) {   // sum (x - mu0)^2 over the valid rows (rows beyond n are clamped loads of the last row)
                float q2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) q2 = fmaf(x[e], x[e], q2);
                ssq += sidx * 32 + 16 * t + li < n ? q2 : 0.f;
            }
            bf16x8 xh, xm;
            split2(x, xh, xm);
            unsigned char* a = myX + pow3_xoff(16 * t + li, 4 * c + lq);
            *reinterpret_cast<bf16x8*>(a) = xh;
            *reinterpret_cast<bf16x8*>(a + 4096) = xm;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    __syncthreads();   // mu
    if (s0 < s1) {
        split_park(0, s0);
        split_park(1, s0);
        if (s0 + 1 < s1) load_x(s0 + 1);
    }
    for (int64_t s = s0; s < s1; ++s) {
        {
            f32x4 acc1[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            prod1(acc1, 0);
            park_partials(acc1, 0);              // (ordered behind the previous stage's adders by that stage's second barrier)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            prod1(acc1, 1);
            park_partials(acc1, 1);
        }
        __syncthreads();
        if (wave < NT) {
            POW3_LANE(ln);
            const int lq = ln >> 4;
            const int64_t rb = s * 32 + 4 * lq;
            f32x8 z8;
            {
                const f32x4 zlo = add_partials(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) z8[r] = rb + r < n ? zlo[r] : 0.f;
            }
            {
                const f32x4 zhi = add_partials(1);
#pragma unroll
                for (int r = 0; r < 4; ++r) z8[4 + r] = rb + 16 + r < n ? zhi[r] : 0.f;
            }
            if (MEANS && wave == NT - 1 && (ln & 15) == 15) {   // the all-ones column: Y'[:, 16 NT - 1] = Xc0^T 1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z8[r] = rb + r < n ? 1.f : 0.f;
                    z8[4 + r] = rb + 16 + r < n ? 1.f : 0.f;
                }
            }
            bf16x8 zh, zm;
            split2(z8, zh, zm);
            sZB[(wave * 2 + 0) * 64 + ln] = zh;
            sZB[(wave * 2 + 1) * 64 + ln] = zm;
        }
        __syncthreads();                         // the z fragments of this stage are published
        {
            POW3_LANE(ln);
            const int trq = (ln >> 2) & 3, trp = ln & 3, lq = ln >> 4;
            auto mfma_pair = [&](const bf16x8(&ax)[2][2], int mp) {
#if PETAL_POW3F_ZPF
                bf16x8 zc[2];
                zc[0] = sZB[(0 * 2 + 0) * 64 + ln]; zc[1] = sZB[(0 * 2 + 1) * 64 + ln];
#endif
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
#if PETAL_POW3F_ZPF
                    const bf16x8 zh = zc[0], zm = zc[1];
                    if (u + 1 < NT) { zc[0] = sZB[((u + 1) * 2 + 0) * 64 + ln]; zc[1] = sZB[((u + 1) * 2 + 1) * 64 + ln]; }
                    __builtin_amdgcn_sched_barrier(0);
#else
                    const bf16x8 zh = sZB[(u * 2 + 0) * 64 + ln], zm = sZB[(u * 2 + 1) * 64 + ln];
#endif
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        f32x4 c4 = acc2[2 * mp + mm][u];
#if PETAL_STEER_PIECES >= 4
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][1], zm, c4, 0, 0, 0);   // smallest terms first
#endif
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][1], zh, c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][0], zm, c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mm][0], zh, c4, 0, 0, 0);
                        acc2[2 * mp + mm][u] = c4;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
                bf16x8 ax[2][2];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const unsigned char* a0 = myX + pow3_xoff(4 * lq + trq, 2 * (2 * mp + mm) + (trp >> 1)) + 8 * (trp & 1);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) ax[mm][pl] = lds_tr2(a0 + pl * 4096, a0 + pl * 4096 + 2048);
                }
                // this half of the image has been read for the last time: the next stage's pieces go there -- in front of the pair's MFMAs
                // in one wave of a SIMD, behind them in the other (DEPHASE)
#if PETAL_POW3F_DEPHASE
                const bool split_first = wave < 4;
#else
                const bool split_first = true;
#endif
                if (split_first && s + 1 < s1) {
                    __builtin_amdgcn_sched_barrier(0);
                    split_park(mp, s + 1);
                }
                mfma_pair(ax, mp);
                if (!split_first && s + 1 < s1) {
                    __builtin_amdgcn_sched_barrier(0);
                    split_park(mp, s + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (s + 2 < s1) load_x(s + 2);
    }
    if (MEANS) {         // one partial of sum (x - mu0)^2 per wave, lanes added in a fixed order
        float v = ssq;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) ssq_part[(int64_t)blockIdx.x * WV + wave] = (double)v;
    }
    // this workgroup's slab: D[row = 4 q + r][col = i] of tile (m, u) is Y'[64 wave + 16 m + 4 q + r][16 u + i]
    {
        POW3_LANE(ln);
        const int li = ln & 15, lq = ln >> 4;
        float* out = part + (int64_t)blockIdx.x * K * (16 * NT);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* row = out + (int64_t)(64 * wave + 16 * m + 4 * lq + r) * (16 * NT);
#pragma unroll
                for (int u = 0; u < NT; ++u) row[16 * u + li] = acc2[m][u][r];
            }
    }
#undef POW3_LANE
}

// ------------------------------------------------------------------------------------------------
// K2p: the `precise` form of C = (A - muA)^T (B - muB): every product and the whole accumulation in fp64 on
// v_mfma_f64_16x16x4_f64 (exact Pca and FastICA whitening need the small eigenvalues of the Gram matrix, which an fp32
// accumulation would drown).  A wave owns 32 columns of A (2 m-tiles: one 8-B load per lane, tile t row i <-> m =
// m0 + 2 i + t) x 64 columns of B (4 tiles: one 16-B load, tile e col j <-> col = n0 + 4 j + e) and a row chunk.
// f64 C/D map: reg r of lane l is D[row = (l >> 4) + 4 r][col = l & 15].
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// T = float: the precise Gram of fp32 data; T = double: K2 for fp64 inputs (same kernel, 16-B / 32-B loads).
// (development knob: the minimum waves per SIMD asked of the compiler for the fp32-data instantiations.  Left alone the kernel
// takes 132 unified registers = 3 waves per SIMD; asked for 4 it fits 96 -- by serialising load -> convert -> MFMA -- and runs
// SLOWER: 3244 vs 2760 us at 500000 x 512, 449 vs 323 us at 200000 x 256, round 4, same box)
#ifndef PETAL_GRAM_MINWAVES
#define PETAL_GRAM_MINWAVES 1
#endif
template <class T, bool CA, bool CB, int NE = 4>
__global__ __launch_bounds__(256, (sizeof(T) == 4 && NE == 4 ? PETAL_GRAM_MINWAVES : 1)) void k_atb_f64(const T* __restrict__ A, int64_t lda, int M, const T* __restrict__ muA,
                                                 const T* __restrict__ B, int64_t ldb, int N, const T* __restrict__ muB,
                                                 int64_t n, int64_t chunk, double* __restrict__ part, int sym) {
    // NE = B tiles per wave: lane i holds the NE consecutive columns n0 + NE i .. (tile e, col j <-> col = n0 + NE j + e).
    // NE = 5 serves N = 80 (l = 74 padded) with one panel instead of two 64-wide ones, the second 75 % empty.
    typedef T tx2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    // Gram matrix (sym): only the (32 x 64) WAVE tiles that reach the diagonal or lie above it are computed -- the grid's x
    // index times four plus the wave enumerates them slice by slice (the 32-row slice a keeps the panels b >= a / 2); the rest
    // is mirrored afterwards.  (Round 2 enumerated 128 x 64 workgroup blocks: 75 % of the square at d = 256 against 62.5 % here,
    // 62.5 % against 56 % at d = 512.)
    int m0, n0;
    if (sym) {
        const int gy = (N + 63) / 64;
        int t = blockIdx.x * 4 + wave, a = 0;
        for (;;) {
            const int kept = gy - min(gy, a >> 1);
            if (kept == 0) return;              // past the last live tile (the last workgroup may be short)
            if (t < kept) break;
            t -= kept;
            ++a;
        }
        m0 = 32 * a;
        n0 = (min(gy, a >> 1) + t) * (16 * NE);
    } else {
        m0 = (blockIdx.x * 4 + wave) * 32;
        n0 = blockIdx.y * (16 * NE);
    }
    if (m0 >= M) return;
    const int64_t rbeg = (int64_t)blockIdx.z * chunk, rend = min(n, rbeg + chunk);
    const int mc = min(m0 + 2 * i, M - 2), ncl = min(n0 + NE * i, N - NE);  // clamped: out-of-range outputs are never stored
    const T* ap = A + mc;
    const T* bp = B + ncl;
    tx2 ma = tx2{T(0), T(0)};
    T mb[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) mb[e] = CB ? muB[ncl + e] : T(0);
    if (CA) ma = *reinterpret_cast<const tx2*>(muA + mc);
    f64x4 acc[2][NE];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[t][e] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int64_t r0 = rbeg; r0 < rend; r0 += 16) {
        tx2 av[4];
        T bv[4][NE];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t r = r0 + 4 * s + q;
            const int64_t rc = r < rend ? r : rend - 1;
            av[s] = *reinterpret_cast<const tx2*>(ap + rc * lda);
            if constexpr (NE == 4) {  // one 16-B / 32-B load
                typedef T tx4 __attribute__((ext_vector_type(4)));
                const tx4 v = *reinterpret_cast<const tx4*>(bp + rc * ldb);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[s][e] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) bv[s][e] = bp[rc * ldb + e];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool rv = (r0 + 4 * s + q) < rend;
            if (CA) av[s] -= ma;            // centred in the storage type, exactly like the crate's `input - &means`
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                if (CB) bv[s][e] -= mb[e];
                if (!rv) bv[s][e] = T(0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    acc[t][e] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[s][t], (double)bv[s][e], acc[t][e], 0, 0, 0);
        }
    }
    double* out = part + (int64_t)blockIdx.z * M * N;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 2 * (q + 4 * r) + t;
            if (m >= M) continue;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int col = n0 + NE * i + e;
                if (col < N) out[(int64_t)m * N + col] = acc[t][e][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// K1 for fp64 inputs on the fp64 matrix cores: Z = (X - mu) P (+ bias, + sum (X - mu)^2).  One wave = 32 rows x NT column
// tiles, K in 8-deep chunks: lane (i, q) loads X[row][8 c + 2 q, + 1] (16 B) -- the operands of two MFMA k-steps -- and
// the matching pair of the packed P (k order inside a chunk permuted identically); two register stages.
typedef double f64x2 __attribute__((ext_vector_type(2)));
// Ppk64[((c NTtot + nt) 64 + lane) 2 + s] = P[8 c + 2 (lane >> 4) + s][16 nt + (lane & 15)]
__global__ __launch_bounds__(256) void k_pack_p64(const double* __restrict__ P, int64_t K, int64_t N, int64_t ldp,
                                                  f64x2* __restrict__ out, int NTtot, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t tile = idx >> 6;
    const int nt = (int)(tile % NTtot);
    const int64_t c = tile / NTtot, col = 16 * nt + (lane & 15), k0 = 8 * c + 2 * (lane >> 4);
    f64x2 v;
    v[0] = (k0 < K && col < N) ? P[k0 * ldp + col] : 0.0;
    v[1] = (k0 + 1 < K && col < N) ? P[(k0 + 1) * ldp + col] : 0.0;
    out[idx] = v;
}
template <int NT, bool CENTER, bool SUMSQ>
__global__ __launch_bounds__(256) void k_xp_f64(const double* __restrict__ X, int64_t n, int K, int64_t ldx,
                                                const double* __restrict__ mu, const f64x2* __restrict__ Ppk, int NTtot, int nt0,
                                                int N, const double* __restrict__ bias, double* __restrict__ Z, int64_t ldz,
                                                double* __restrict__ ss_part) {
    constexpr int RT = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);
    double ssq = 0.0;
    if (row0 < n) {
        f64x4 acc[RT][NT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[t][u] = f64x4{0.0, 0.0, 0.0, 0.0};
        const double* xrow[RT];
        bool rvalid[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int64_t r = row0 + 16 * t + i;
            rvalid[t] = r < n;
            xrow[t] = X + (rvalid[t] ? r : (n - 1)) * ldx + 2 * q;
        }
        const f64x2* pb = Ppk + (int64_t)nt0 * 64 + lane;
        const double* mup = mu + 2 * q;
        const int nchunk = K >> 3;
        auto load_chunk = [&](int c, f64x2(&a)[RT], f64x2(&b)[NT], f64x2& m) {
#pragma unroll
            for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const f64x2*>(xrow[t] + 8 * c);
#pragma unroll
            for (int u = 0; u < NT; ++u) b[u] = pb[((int64_t)c * NTtot + u) * 64];
            if (CENTER) m = *reinterpret_cast<const f64x2*>(mup + 8 * c);
        };
        auto compute = [&](f64x2(&a)[RT], f64x2(&b)[NT], const f64x2& m) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if (CENTER) a[t] -= m;
                if (SUMSQ && rvalid[t]) ssq += a[t][0] * a[t][0] + a[t][1] * a[t][1];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][s2], b[u][s2], acc[t][u], 0, 0, 0);
        };
        f64x2 a0[RT], b0[NT], a1[RT], b1[NT], m0 = f64x2{0.0, 0.0}, m1 = m0;
        const int last = nchunk - 1;
        load_chunk(0, a0, b0, m0);
        int c = 0;
        for (; c + 2 <= nchunk; c += 2) {
            load_chunk(c + 1, a1, b1, m1);
            __builtin_amdgcn_sched_barrier(0);
            compute(a0, b0, m0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(c + 2 < last ? c + 2 : last, a0, b0, m0);
            __builtin_amdgcn_sched_barrier(0);
            compute(a1, b1, m1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < nchunk) compute(a0, b0, m0);
        // C/D of the f64 MFMA: register r of lane (i, q) is row q + 4 r of the A side (the X row), column i of the B side
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int col = 16 * (nt0 + u) + i;
            if (col >= N) continue;
            const double bv = bias ? bias[col] : 0.0;
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + 16 * t + q + 4 * r;
                    if (row < n) Z[row * ldz + col] = acc[t][u][r] + bv;
                }
        }
    }
    if (SUMSQ) {
        for (int off = 32; off > 0; off >>= 1) ssq += __shfl_down(ssq, off, 64);
        if (lane == 0) ss_part[(int64_t)blockIdx.x * 4 + wave] = ssq;
    }
}

// C[m][j] (m > j) <- C[j][m]: completes a symmetric product whose strictly-lower tiles were skipped
__global__ __launch_bounds__(256) void k_mirror_upper(double* __restrict__ C, int64_t M, int64_t ldc) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * M) return;
    const int64_t m = e / M, j = e - m * M;
    if (m > j) C[m * ldc + j] = C[j * ldc + m];
}

// ================================================================================================
// K7: fused FastICA step (ica.rs:332-333) -- per 16-sample tile: S = X1 . W^T (MFMA) -> tanh ->
// D += G^T . X1 (MFMA) and gp += sum(1 - g^2); partial D / gp per wave, combined in fp64.
// ================================================================================================
__device__ __forceinline__ float tanh_fast(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); saturates correctly at +-inf; |abs err| ~ 1e-7; tanh(0) = 0 exactly.
    // v_exp_f32 + v_rcp_f32 (1 ulp each): five VALU instructions instead of the ~14 of an IEEE division
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// Wpk[((kc * NT + nt) * 64 + lane) * 4 + s] = W[16 nt + (lane&15)][16 kc + 4 (lane>>4) + s]   (B = W^T)
__global__ void k_pack_w(const double* __restrict__ W, int nc, float* __restrict__ Wpk, int NT) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NT * NT * 64) return;
    const int lane = e & 63, cn = e >> 6, nt = cn % NT, kc = cn / NT, q = lane >> 4, j = lane & 15;
    f32x4 v;
    for (int s = 0; s < 4; ++s) {
        const int comp = 16 * nt + j, k = 16 * kc + 4 * q + s;
        v[s] = (comp < nc && k < nc) ? (float)W[comp * nc + k] : 0.f;
    }
    reinterpret_cast<f32x4*>(Wpk)[e] = v;
}

// one partial slab per WORKGROUP, deterministic: waves 0 and 1 store their tiles into two LDS slabs, waves 2 and 3 add
// theirs on top, and the sum (w0 + w2) + (w1 + w3) of the two slabs [NCP*NCP D | NCP gp] is written once, coalesced.
// s_slab holds 2 (NCP*NCP + NCP) floats.
template <int NT>
__device__ __forceinline__ void ica_write_slab(const f32x4 (&dacc)[NT][NT], const float (&gpa)[NT], float* s_slab,
                                               float* __restrict__ part) {
    constexpr int NCP = 16 * NT, SLAB = NCP * NCP + NCP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    float* mine = s_slab + (wave & 1) * SLAB;
    for (int ph = 0; ph < 2; ++ph) {
        if ((wave >> 1) == ph) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* dst = &mine[(16 * a + 4 * q + r) * NCP + 16 * b + i];
                        *dst = (ph == 0 ? 0.f : *dst) + dacc[a][b][r];
                    }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                float gsum = gpa[a];
                gsum += __shfl_xor(gsum, 16, 64);
                gsum += __shfl_xor(gsum, 32, 64);
                if (q == 0) {
                    float* dst = &mine[NCP * NCP + 16 * a + i];
                    *dst = (ph == 0 ? 0.f : *dst) + gsum;
                }
            }
        }
        __syncthreads();
    }
    float* out = part + (int64_t)blockIdx.x * SLAB;
    for (int e = threadIdx.x; e < SLAB; e += 256) out[e] = s_slab[e] + s_slab[SLAB + e];
}
template <int NT>
__global__ __launch_bounds__(256) void k_ica_mfma(const float* __restrict__ X1T, int64_t n, int64_t ld,
                                                  const float* __restrict__ Wpk, int64_t tiles_per_wave,
                                                  float* __restrict__ part, const int* __restrict__ state) {
    if (state && state[0]) return;
    constexpr int NCP = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    f32x4 wf[NT][NT];  // [kc][nt]
#pragma unroll
    for (int kc = 0; kc < NT; ++kc)
#pragma unroll
        for (int u = 0; u < NT; ++u) wf[kc][u] = reinterpret_cast<const f32x4*>(Wpk)[(kc * NT + u) * 64 + lane];
    f32x4 dacc[NT][NT];  // [component tile][x tile]
    float gpa[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        gpa[a] = 0.f;
#pragma unroll
        for (int b = 0; b < NT; ++b) dacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t t0 = wid * tiles_per_wave, t1 = min((n + 15) / 16, t0 + tiles_per_wave);
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t r0 = tile * 16;
        // A layout: lane (i, q) <- X1[r0 + i][16 kc + 4 q .. +3]
        const int64_t ra = r0 + i;
        const bool va = ra < n;
        f32x4 xa[NT];
#pragma unroll
        for (int kc = 0; kc < NT; ++kc)
            xa[kc] = va ? *reinterpret_cast<const f32x4*>(X1T + ra * ld + 16 * kc + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        // B layout for the second product: lane (j = i, q), k-step s <- X1[r0 + 4 q + s][16 b + j]
        float xb[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rb = r0 + 4 * q + s;
            const bool vb = rb < n;
#pragma unroll
            for (int b = 0; b < NT; ++b) xb[s][b] = vb ? X1T[rb * ld + 16 * b + i] : 0.f;
        }
        f32x4 sacc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NT; ++kc)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    sacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kc][s], wf[kc][u][s], sacc[u], 0, 0, 0);
        // sacc[u][r] = S[sample r0 + 4 q + r][component 16 u + i]
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = tanh_fast(sacc[u][r]);
                const bool v = (r0 + 4 * q + r) < n;
                sacc[u][r] = v ? g : 0.f;
                gpa[u] += v ? (1.0f - g * g) : 0.f;
            }
        // D[component][x] += sum_samples G[sample][component] X1[sample][x]:
        // A operand (i = component, k = q) of k-step s is G[r0 + 4 q + s][16 a + i] = sacc[a][s]
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    dacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(sacc[a][s], xb[s][b], dacc[a][b], 0, 0, 0);
    }
    __shared__ float s_slab[2 * (NCP * NCP + NCP)];
    ica_write_slab<NT>(dacc, gpa, s_slab, part);
}
// K7, split-product form: both products of the step on the bf16 matrix cores (six piece products each, see K1).  One
// wave handles 32 samples per pass.  The first product is laid out so that its OUTPUT is already the second product's
// A operand: C[row = sample][col = component] puts S[samples 4q+r of either 16-sample tile][component i] in lane (i, q),
// and since the k-slot <-> sample assignment of an MFMA is free as long as A and B agree, slot e of lane group q is
// declared to be sample (e < 4 ? 4q+e : 16+4q+e-4); X1 is then loaded in exactly that order for the B operand.  No
// cross-lane traffic between the two products.  W's three planes live in LDS (shared by the four waves).
// Wpk3[((kc NT + u) 3 + plane) 64 + lane][e] = plane of (float)W[16 u + (lane & 15)][32 kc + 8 (lane >> 4) + e]
__global__ __launch_bounds__(256) void k_pack_w3(const double* __restrict__ W, int nc, bf16x8* __restrict__ out, int NT,
                                                 int KCH) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= KCH * NT * 64) return;
    const int lane = idx & 63, tile = idx >> 6, u = tile % NT, kc = tile / NT;
    const int comp = 16 * u + (lane & 15), k0 = 32 * kc + 8 * (lane >> 4);
    f32x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (comp < nc && k0 + e < nc) ? (float)W[comp * nc + k0 + e] : 0.f;
    bf16x8 h, m, l;
    split3(x, h, m, l);
    out[(tile * 3 + 0) * 64 + lane] = h;
    out[(tile * 3 + 1) * 64 + lane] = m;
    out[(tile * 3 + 2) * 64 + lane] = l;
}

template <int NT>
__global__ __launch_bounds__(256, 2) void k_ica3(const float* __restrict__ X1T, int64_t n, int64_t ld,
                                                 const bf16x8* __restrict__ Wpk3, int64_t blocks_per_wave,
                                                 float* __restrict__ part, const int* __restrict__ state) {
    if (state && state[0]) return;
    constexpr int NCP = 16 * NT, KCH = (NCP + 31) / 32, WITEMS = KCH * NT * 192;
    constexpr int XP = NCP + 4;  // row pitch of the transposition buffer: 4 XP = 16 (mod 32) banks
    constexpr int XT_FLOATS = 4 * 32 * XP, SLAB = 2 * (NCP * NCP + NCP);
    __shared__ bf16x8 sW[WITEMS];
    __shared__ __attribute__((aligned(16))) float sX[XT_FLOATS > SLAB ? XT_FLOATS : SLAB];  // per wave [32 samples][XP]; the slab at the end
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    for (int e = threadIdx.x; e < WITEMS; e += 256) sW[e] = Wpk3[e];
    __syncthreads();
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    float* xt = sX + wave * 32 * XP;
    f32x4 dacc[NT][NT];  // [component tile][x tile]
    float gpa[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        gpa[a] = 0.f;
#pragma unroll
        for (int b = 0; b < NT; ++b) dacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t b0 = wid * blocks_per_wave, b1 = min((n + 31) / 32, b0 + blocks_per_wave);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // A operand of the first product: lane (i, q) <- X1[r0 + 16 t + i][32 kc + 8 q .. + 7]
    f32x4 xa[2][KCH][2];
    auto load_a = [&](int64_t blk) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t ra = blk * 32 + 16 * t + i;
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
                const bool v = ra < n && (32 * kc + 8 * q) < NCP;
                const float* src = X1T + ra * ld + 32 * kc + 8 * q;
                xa[t][kc][0] = v ? *reinterpret_cast<const f32x4*>(src) : z4;
                xa[t][kc][1] = v ? *reinterpret_cast<const f32x4*>(src + 4) : z4;
            }
        }
    };
    if (b0 < b1) load_a(b0);
    // Each group of MFMAs is written next to independent VALU work (chunk kc's MFMAs beside the split of chunk kc + 1 or
    // of the transposed rows; component tile a's MFMAs beside tanh + split of tile a + 1).  Measured (dev/micro_coissue.hip):
    // on a SIMD holding two such waves MFMA time and VALU time ADD rather than overlap, so the pass costs
    // ~3070 (192 MFMAs) + ~3600 (730 VALU, 64 of them quarter-rate transcendentals) cycles; the kernel runs within 25 % of that.
    for (int64_t blk = b0; blk < b1; ++blk) {
        const int64_t r0 = blk * 32;
        bf16x8 ah[2], am[2], al[2];
        auto split_a = [&](int kc) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x8 x = {xa[t][kc][0][0], xa[t][kc][0][1], xa[t][kc][0][2], xa[t][kc][0][3],
                                 xa[t][kc][1][0], xa[t][kc][1][1], xa[t][kc][1][2], xa[t][kc][1][3]};
                split3(x, ah[t], am[t], al[t]);
            }
        };
        // the raw rows also go to the wave's LDS buffer, from which the second product reads them transposed
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc)
                if ((32 * kc + 8 * q) < NCP) {
                    float* dst = xt + (16 * t + i) * XP + 32 * kc + 8 * q;
                    *reinterpret_cast<f32x4*>(dst) = xa[t][kc][0];
                    *reinterpret_cast<f32x4*>(dst + 4) = xa[t][kc][1];
                }
        bf16x8 nwh = sW[lane], nwm = sW[64 + lane], nwl = sW[128 + lane];
        split_a(0);
        __builtin_amdgcn_wave_barrier();
        f32x4 sacc[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) sacc[t][u] = z4;
        bf16x8 bh[NT], bm[NT], bl[NT];
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 ch[2], cm[2], cl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { ch[t] = ah[t]; cm[t] = am[t]; cl[t] = al[t]; }
            if (kc + 1 < KCH) {
                split_a(kc + 1);
            } else {
                // B operand: lane (j = i, q), slot e <- X1[r0 + (e < 4 ? 4 q + e : 16 + 4 q + e - 4)][16 b + j]
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    f32x8 xb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) xb[e] = xt[((e < 4 ? 4 * q + e : 12 + 4 * q + e)) * XP + 16 * b + i];
                    split3(xb, bh[b], bm[b], bl[b]);
                }
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const bf16x8 wh = nwh, wm = nwm, wl = nwl;
                if (kc * NT + u + 1 < KCH * NT) {  // W's pieces are read one tile ahead (LDS latency off the MFMA path)
                    const bf16x8* sw = sW + (kc * NT + u + 1) * 192 + lane;
                    nwh = sw[0], nwm = sw[64], nwl = sw[128];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 c4 = sacc[t][u];
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cl[t], wh, c4, 0, 0, 0);  // smallest terms first
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm[t], wm, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[t], wl, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cm[t], wh, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[t], wm, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[t], wh, c4, 0, 0, 0);
                    sacc[t][u] = c4;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < b1) load_a(blk + 1);  // next pass's rows land behind tanh and the second product
        // sacc[t][u][r] = S[sample r0 + 16 t + 4 q + r][component 16 u + i].  Rows past n were loaded as zeros: S = 0
        // and tanh(0) = 0 exactly, so only the g' sum needs masking (last pass).
        const bool tail = r0 + 32 > n;
        const float nvalid = tail ? (float)((n > r0 + 4 * q ? (int)min((int64_t)4, n - r0 - 4 * q) : 0) +
                                            (n > r0 + 16 + 4 * q ? (int)min((int64_t)4, n - r0 - 16 - 4 * q) : 0))
                                  : 8.0f;
        bf16x8 gh, gm, gl;
        auto make_g = [&](int u) {
            f32x8 g8;
            float gs = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = tanh_fast(sacc[t][u][r]);
                    g8[4 * t + r] = g;
                    gs = fmaf(-g, g, gs);
                }
            gpa[u] += gs + nvalid;
            split3(g8, gh, gm, gl);
        };
        make_g(0);
        // D[component][x] += sum_samples G[sample][component] X1[sample][x]
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 fh = gh, fm = gm, fl = gl;
            if (a + 1 < NT) make_g(a + 1);
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                f32x4 c4 = dacc[a][b];
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, bh[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm, bm[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bl[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm, bh[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bm[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bh[b], c4, 0, 0, 0);
                dacc[a][b] = c4;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // the slab aliases the transposition buffers
    ica_write_slab<NT>(dacc, gpa, sX, part);
}
// K7 on PRE-SPLIT whitened data (round 5).  X1 is constant over the 10 .. 200 iterations of a fit, and k_ica3 split it into bf16
// planes twice per iteration (row fragments for the first product, transposed fragments for the second): a third or more of the
// VALU instructions of a kernel that is paced by them (9.3 per MFMA, 16 % matrix-pipe busy: profiles/r04_pmc_fastica_*).  Here the
// planes are made ONCE per fit (k_ica_planes), in fragment order:
//     X1pl[(((b 2 + t) KCH + kc) 3 + plane) 64 + lane][e] = plane of X1[32 b + 16 t + (lane & 15)][32 kc + 8 (lane >> 4) + e]
// -- a wave's load of one fragment is 1 KB contiguous -- and the step kernel loads them as they are: they ARE the first product's A
// operand, and, parked in a wave-private row-major image (16-B chunks XOR-swizzled by the row, as k_pow3's), come back TRANSPOSED
// through ds_read_b64_tr_b16 as the second product's B operand (k-slots declared to be the samples {4 q + r, 16 + 4 q + r}, which is
// the first product's output layout: see k_ica3).  Only G = tanh(S) is split inside the loop.
template <int KCH>
__global__ __launch_bounds__(256) void k_ica_planes(const float* __restrict__ X1T, int64_t n, int64_t ld, int NCP, bf16x8* __restrict__ out,
                                                    int64_t nfrag) {
    const int64_t frag = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, t, kc)
    if (frag >= nfrag) return;
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int kc = (int)(frag % KCH);
    const int64_t bt = frag / KCH, row = 16 * bt + i;
    f32x8 x = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (row < n && 32 * kc + 8 * q < NCP) {
        const float* src = X1T + row * ld + 32 * kc + 8 * q;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
        x = f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    bf16x8 h, m, l;
    split3(x, h, m, l);
    bf16x8* o = out + (frag * 3) * 64 + lane;
    o[0] = h; o[64] = m; o[128] = l;
}
// byte offset of 16-B chunk `ch` of row `row` in one plane of a wave's image: 128-B rows (64 components) as k_pow3's; 64-B rows (32
// components): rows r and r + 4 share a 256-B bank row, so the upper four of each eight take the other half of it
template <int NT>
__device__ __forceinline__ int ica_xoff(int row, int ch) {
    return NT == 4 ? pow3_xoff(row, ch) : row * 64 + 16 * (ch ^ (2 * ((row >> 2) & 1)));
}
template <int NT>
__global__ __launch_bounds__(256, 2) void k_ica3p(const bf16x8* __restrict__ X1pl, int64_t n, const bf16x8* __restrict__ Wpk3,
                                                  int64_t blocks_per_wave, float* __restrict__ part, const int* __restrict__ state) {
    static_assert(NT == 2 || NT == 4, "whole 32-component chunks only");
    if (state && state[0]) return;
    constexpr int NCP = 16 * NT, KCH = NCP / 32, WITEMS = KCH * NT * 192;
    constexpr int ROWB = NCP * 2, PLANE = 32 * ROWB, IMG = 3 * PLANE, SLAB = 2 * (NCP * NCP + NCP);
    constexpr int SX = 4 * IMG > SLAB * 4 ? 4 * IMG : SLAB * 4;
    __shared__ bf16x8 sW[WITEMS];
    __shared__ __attribute__((aligned(16))) unsigned char sXb[SX];  // per wave [3 planes][32 samples][NCP bf16]; the slab at the end
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    for (int e = threadIdx.x; e < WITEMS; e += 256) sW[e] = Wpk3[e];
    __syncthreads();
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    unsigned char* const img = sXb + wave * IMG;
    f32x4 dacc[NT][NT];  // [component tile][x tile]
    float gpa[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        gpa[a] = 0.f;
#pragma unroll
        for (int b = 0; b < NT; ++b) dacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t b0 = wid * blocks_per_wave, b1 = min((n + 31) / 32, b0 + blocks_per_wave);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    bf16x8 xa[2][KCH][3];
    auto load_a = [&](int64_t blk) {
        const bf16x8* src = X1pl + (blk * 2 * KCH * 3) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xa[t][kc][pl] = src[((t * KCH + kc) * 3 + pl) * 64];
    };
    if (b0 < b1) load_a(b0);
    const int trq = (lane >> 2) & 3, trp = lane & 3;
    for (int64_t blk = b0; blk < b1; ++blk) {
        const int64_t r0 = blk * 32;
        // the planes go to the wave's image (row 16 t + i, chunk 4 kc + q), from which the second product reads them transposed
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
                unsigned char* a = img + ica_xoff<NT>(16 * t + i, 4 * kc + q);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(a + pl * PLANE) = xa[t][kc][pl];
            }
        bf16x8 nwh = sW[lane], nwm = sW[64 + lane], nwl = sW[128 + lane];
        f32x4 sacc[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) sacc[t][u] = z4;
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const bf16x8 wh = nwh, wm = nwm, wl = nwl;
                if (kc * NT + u + 1 < KCH * NT) {  // W's pieces are read one tile ahead (LDS latency off the MFMA path)
                    const bf16x8* sw = sW + (kc * NT + u + 1) * 192 + lane;
                    nwh = sw[0], nwm = sw[64], nwl = sw[128];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 c4 = sacc[t][u];
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t][kc][2], wh, c4, 0, 0, 0);  // smallest terms first
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t][kc][1], wm, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t][kc][0], wl, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t][kc][1], wh, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t][kc][0], wm, c4, 0, 0, 0);
                    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t][kc][0], wh, c4, 0, 0, 0);
                    sacc[t][u] = c4;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < b1) load_a(blk + 1);  // next pass's planes land behind tanh and the second product
        // B operand of the second product: lane (j = i, q), slot e <- X1[r0 + (e < 4 ? 4 q + e : 16 + 4 q + e - 4)][16 b + j], transposed
        // reads of the image (T10: lane 16 g + 4 q' + p supplies block row q', columns 4 p .. 4 p + 3)
        bf16x8 bh[NT], bm[NT], bl[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const unsigned char* a0 = img + ica_xoff<NT>(4 * q + trq, 2 * b + (trp >> 1)) + 8 * (trp & 1);   // (row + 16: + 16 rows, same swizzle)
            bh[b] = lds_tr2(a0, a0 + 16 * ROWB);
            bm[b] = lds_tr2(a0 + PLANE, a0 + PLANE + 16 * ROWB);
            bl[b] = lds_tr2(a0 + 2 * PLANE, a0 + 2 * PLANE + 16 * ROWB);
        }
        // sacc[t][u][r] = S[sample r0 + 16 t + 4 q + r][component 16 u + i].  Rows past n were stored as zero planes: S = 0
        // and tanh(0) = 0 exactly, so only the g' sum needs masking (last pass).
        const bool tail = r0 + 32 > n;
        const float nvalid = tail ? (float)((n > r0 + 4 * q ? (int)min((int64_t)4, n - r0 - 4 * q) : 0) +
                                            (n > r0 + 16 + 4 * q ? (int)min((int64_t)4, n - r0 - 16 - 4 * q) : 0))
                                  : 8.0f;
        bf16x8 gh, gm, gl;
        auto make_g = [&](int u) {
            f32x8 g8;
            float gs = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = tanh_fast(sacc[t][u][r]);
                    g8[4 * t + r] = g;
                    gs = fmaf(-g, g, gs);
                }
            gpa[u] += gs + nvalid;
            split3(g8, gh, gm, gl);
        };
        make_g(0);
        // D[component][x] += sum_samples G[sample][component] X1[sample][x]
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 fh = gh, fm = gm, fl = gl;
            if (a + 1 < NT) make_g(a + 1);
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                f32x4 c4 = dacc[a][b];
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, bh[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm, bm[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bl[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm, bh[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bm[b], c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bh[b], c4, 0, 0, 0);
                dacc[a][b] = c4;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // the slab aliases the images
    ica_write_slab<NT>(dacc, gpa, reinterpret_cast<float*>(sXb), part);
}
// combine the per-workgroup slabs in fp64 (fixed order) and drop the padding: GX_gp = [nc*nc | nc].
// block = 32 outputs x 32 part-lanes (the reduction is latency-bound: many short independent load chains).
__global__ __launch_bounds__(1024) void k_ica_reduce(const float* __restrict__ part, int64_t nparts, int NCP, int nc,
                                                     double* __restrict__ out, const int* __restrict__ state) {
    if (state && state[0]) return;
    __shared__ double red[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + tx;
    const int64_t slab = (int64_t)NCP * NCP + NCP;
    double sacc = 0;
    if (e < nc * nc + nc) {
        const int src = e < nc * nc ? (e / nc) * NCP + (e % nc) : NCP * NCP + (e - nc * nc);
        for (int64_t p = ty; p < nparts; p += 32) sacc += (double)part[p * slab + src];
    }
    red[ty][tx] = sacc;
    __syncthreads();
    if (ty == 0 && e < nc * nc + nc) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][tx];  // fixed order: deterministic
        out[e] = t;
    }
}

// generic FastICA step: one block per chunk of 64 samples, fp64
template <class T>
__global__ void k_ica_simple(const T* __restrict__ X1T, int64_t n, int nc, int64_t ld, const double* __restrict__ W,
                             double* __restrict__ part, const int* __restrict__ state) {
    if (state && state[0]) return;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* g = reinterpret_cast<double*>(smem_raw);  // [nc][64]
    const int64_t s0 = (int64_t)blockIdx.x * 64;
    const int ns = (int)min((int64_t)64, n - s0);
    for (int e = threadIdx.x; e < nc * 64; e += blockDim.x) {
        const int c = e / 64, s = e % 64;
        double v = 0;
        if (s < ns) {
            double wx = 0;
            for (int j = 0; j < nc; ++j) wx += (sizeof(T) == 4 ? (double)(float)W[c * nc + j] : W[c * nc + j]) * (double)X1T[(s0 + s) * ld + j];
            v = tanh(wx);
        }
        g[e] = v;
    }
    __syncthreads();
    double* out = part + (int64_t)blockIdx.x * (nc * nc + nc);
    for (int e = threadIdx.x; e < nc * nc + nc; e += blockDim.x) {
        double s = 0;
        if (e < nc * nc) {
            const int c = e / nc, j = e % nc;
            for (int t = 0; t < ns; ++t) s += g[c * 64 + t] * (double)X1T[(s0 + t) * ld + j];
        } else {
            const int c = e - nc * nc;
            for (int t = 0; t < ns; ++t) s += 1.0 - g[c * 64 + t] * g[c * 64 + t];
        }
        out[e] = s;
    }
}
__global__ void k_sum_parts_state(const double* __restrict__ part, int64_t nparts, int64_t count, double* __restrict__ out,
                                  const int* __restrict__ state) {
    if (state && state[0]) return;
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= count) return;
    double s = 0;
    for (int64_t p = 0; p < nparts; ++p) s += part[p * count + j];
    out[j] = s;
}

// ================================================================================================
// fp64 small-matrix kernels
// ================================================================================================
// C[M x N] = alpha op(A) op(B) + beta C, fp64, 16 x 16 outputs per block, 32-deep K steps; the global loads are
// coalesced for every transposition case (the transposed operand is read along its contiguous index and
// transposed on the way into LDS).
// Split-K form (gridDim.z > 1, used when K is long and the output small): slice z covers K-range [z kchunk, (z+1) kchunk)
// and writes its raw partial tile to part[z][M][N]; k_dgemm_reduce adds the slices in fixed order (deterministic).
__global__ __launch_bounds__(256) void k_dgemm(bool ta, bool tb, int64_t M, int64_t N, int64_t K, double alpha,
                                               const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                                               double beta, double* __restrict__ C, int64_t ldc, int64_t kchunk,
                                               double* __restrict__ part, const double* __restrict__ colscale,
                                               bf16x8* __restrict__ pk3, int NTtot) {
    constexpr int BK = 32;
    __shared__ double sa[16][BK + 1], sb[BK][17];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int64_t i0 = (int64_t)blockIdx.y * 16, j0 = (int64_t)blockIdx.x * 16;
    const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
    if (K > kbeg + kchunk) K = kbeg + kchunk;
    double acc = 0;
    for (int64_t k0 = kbeg; k0 < K; k0 += BK) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (ta) {  // A is K x M: contiguous in i
                const int kk = ty + 16 * h;
                const int64_t k = k0 + kk, i = i0 + tx;
                sa[tx][kk] = (i < M && k < K) ? A[k * lda + i] : 0.0;
            } else {   // A is M x K: contiguous in k
                const int kk = tx + 16 * h;
                const int64_t k = k0 + kk, i = i0 + ty;
                sa[ty][kk] = (i < M && k < K) ? A[i * lda + k] : 0.0;
            }
            if (tb) {  // B is N x K: contiguous in k
                const int kk = tx + 16 * h;
                const int64_t k = k0 + kk, j = j0 + ty;
                sb[kk][ty] = (j < N && k < K) ? B[j * ldb + k] : 0.0;
            } else {   // B is K x N: contiguous in j
                const int kk = ty + 16 * h;
                const int64_t k = k0 + kk, j = j0 + tx;
                sb[kk][tx] = (j < N && k < K) ? B[k * ldb + j] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) acc += sa[ty][k] * sb[k][tx];
        __syncthreads();
    }
    const int64_t i = i0 + ty, j = j0 + tx;
    if (i < M && j < N) {
        if (part) part[((int64_t)blockIdx.z * M + i) * N + j] = acc;
        else C[i * ldc + j] = alpha * acc * (colscale ? colscale[j] : 1.0) + (beta != 0.0 ? beta * C[i * ldc + j] : 0.0);
    }
    if (pk3) {
        // the product is the small operand of the next split-product GEMM (op_gemm_xp_prod): emit its three bf16 planes in
        // k_pack_p3's layout straight from this tile -- rows i0 .. i0 + 15 are two 8-row operand groups of chunk i0 / 32,
        // columns j0 .. j0 + 15 one column tile.  (M, N multiples of 16; beta = 0, no split-K.)
        __syncthreads();
        float* sv = reinterpret_cast<float*>(&sa[0][0]);  // [16][17] floats
        sv[ty * 17 + tx] = (i < M && j < N) ? (float)(alpha * acc) : 0.f;
        __syncthreads();
        if ((ty & 7) == 0) {
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = sv[(ty + e) * 17 + tx];
            bf16x8 h, m, l;
            split3(x, h, m, l);
            const int64_t c = i0 >> 5, tile = c * NTtot + (j0 >> 4);
            const int lane = tx + 16 * (int)(((i0 & 31) + ty) >> 3);
            pk3[(tile * 3 + 0) * 64 + lane] = h;
            pk3[(tile * 3 + 1) * 64 + lane] = m;
            pk3[(tile * 3 + 2) * 64 + lane] = l;
            if ((M & 31) == 16 && i0 + 16 == M) {   // the last 32-row chunk is half empty: its upper operand groups are zeros
                bf16x8 z; for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pk3[(tile * 3 + pl) * 64 + lane + 32] = z;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_dgemm_reduce(const double* __restrict__ part, int ks, int64_t M, int64_t N, double alpha,
                                                      double beta, double* __restrict__ C, int64_t ldc,
                                                      const double* __restrict__ colscale) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * N) return;
    double acc = 0;
    for (int z = 0; z < ks; ++z) acc += part[(int64_t)z * M * N + e];
    const int64_t i = e / N, j = e - i * N;
    C[i * ldc + j] = alpha * acc * (colscale ? colscale[j] : 1.0) + (beta != 0.0 ? beta * C[i * ldc + j] : 0.0);
}

// P = A . R^-1 for an upper-triangular R given in "RT form" (k_chol_inv2, rt_form: diagonal 16 x 16 blocks hold T_JJ = R_JJ^-1, the
// blocks above them R itself): blocked forward substitution P_J = (A_J - sum_{I < J} P_I R_IJ) T_JJ on the fp64 matrix cores, ONE
// WAVE per 16 rows of A (rows are independent), transposed so that results feed the next product without leaving the registers:
// X_J = A_J^T lives in the accumulator layout of v_mfma_f64_16x16x4 (register r of lane l = X[(l >> 4) + 4 r][l & 15]);
// P_I^T = T_II^T X_I and X_J -= R_IJ^T P_I^T take those registers directly as the B operands of MFMA r (the k index is summed
// over, so the accumulator's k' = (l >> 4) + 4 r order is as good as any when the A operand uses the same one).  The R / T
// operands come straight from L2 (16 consecutive doubles per row), none of them on the dependency chain, which is
// nb x 8 MFMAs long.  Epilogue as k_dgemm's: P (fp64) and the three bf16 operand planes of the next split-product GEMM (k_pack_p3's
// layout), through one LDS transpose.  M (the order of R) is a multiple of 16, <= 144; K a multiple of 16.
typedef double cf64x4 __attribute__((ext_vector_type(4)));
constexpr int TRSM_MAXM = 144;
__host__ __device__ inline size_t trsm_lds_bytes(int M) { return sizeof(double) * 16 * (size_t)(M + 2); }
// P2: the result is ROUNDED to the sum of its two leading bf16 pieces, P := bf16(p) + bf16(p - bf16(p)) (16 significant bits), in
// P_out too -- the re-based iterate may be any basis of range(A), so this rounding defines the iterate instead of perturbing a
// product (numpy model, configs[1] / configs[3] spectra: no change in the 5e-6 / 1.5e-5 component errors), and the next product
// needs five piece products instead of six (k_xp3<..., NPL = 2>).  The third plane is not written.
template <int NB, bool P2 = false>   // NB = M / 16 at compile time: straight-line code, so every operand load is issued ahead of the MFMA chain
__global__ __launch_bounds__(64) void k_trsm_pack(const double* __restrict__ A, int64_t lda, const double* __restrict__ RT, int64_t ldt,
                                                  int64_t K, double* __restrict__ P_out, int64_t ldpo,
                                                  bf16x8* __restrict__ pk3, int NTtot) {
    extern __shared__ __attribute__((aligned(16))) double sP[];   // [16][M + 2]
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    constexpr int M = 16 * NB, ldsp = M + 2;
    const int64_t i0 = (int64_t)blockIdx.x * 16;
    cf64x4 X[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) X[J][r] = A[(i0 + li) * lda + 16 * J + lk + 4 * r];
    }
    // Block row I + 1 of RT is requested while block row I is being used (two register sets): with every operand requested up
    // front the 45 tiles of NB = 9 need 360 registers and the compiler serialises them -- each MFMA then waits for its own L2
    // round trip (50 us at l = 138)
    cf64x4 rt[2][NB];
    auto load_row = [&](int I, cf64x4(&dst)[NB]) {
        const double* rrow = RT + (int64_t)(16 * I + lk) * ldt + li;   // row 16 I + lk (+ 4 r), column (16 J +) li
#pragma unroll
        for (int J = 0; J < NB; ++J)
            if (J >= I) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[J][r] = rrow[(int64_t)4 * r * ldt + 16 * J];
            }
    };
    load_row(0, rt[0]);
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        if (I + 1 < NB) load_row(I + 1, rt[(I + 1) & 1]);
        const cf64x4(&cur)[NB] = rt[I & 1];
        cf64x4 pt = cf64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][r], X[I][r], pt, 0, 0, 0);
#pragma unroll
        for (int J = I + 1; J < NB; ++J) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                X[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(-cur[J][r], pt[r], X[J], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sP[li * ldsp + 16 * I + lk + 4 * r] = pt[r];
        __builtin_amdgcn_sched_barrier(0);   // (keeps the loads of row I + 2 from being hoisted above this row's products)
    }
    __syncthreads();
    if (P_out && !P2)
        for (int e = lane; e < 16 * M; e += 64) {
            const int r = e / M, c = e - r * M;
            P_out[(i0 + r) * ldpo + c] = sP[r * ldsp + c];
        }
    if (pk3) {
        for (int idx = lane; idx < 2 * M; idx += 64) {   // item = 8 rows (group g) of column col
            const int g = idx / M, col = idx - g * M;
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)sP[(8 * g + e) * ldsp + col];
            bf16x8 h, m, l;
            split3(x, h, m, l);
            const int64_t c = i0 >> 5, tile = c * NTtot + (col >> 4);
            const int ln = (col & 15) + 16 * (int)(((i0 & 31) + 8 * g) >> 3);
            pk3[(tile * 3 + 0) * 64 + ln] = h;
            pk3[(tile * 3 + 1) * 64 + ln] = m;
            if (P2) {
                if (P_out) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) P_out[(i0 + 8 * g + e) * ldpo + col] = (double)(float)h[e] + (double)(float)m[e];
                }
            } else
            pk3[(tile * 3 + 2) * 64 + ln] = l;
            if ((K & 31) == 16 && i0 + 16 == K) {   // half-empty last chunk: zero operand groups for the rows that do not exist
                bf16x8 z; for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
#pragma unroll
                for (int pl = 0; pl < (P2 ? 2 : 3); ++pl) pk3[(tile * 3 + pl) * 64 + ln + 32] = z;
                if (P2 && P_out) {   // (those rows do not exist in P_out)
                }
            }
        }
    }
}

// X = R^-1 . B for an upper-triangular R in RT form (round 6: the final stage's U = Z (R^-1 Uh) without the explicit inverse -- with it
// the last factorisation of a fit can be k_chol_rt4's): blocked BACK substitution X_I = T_II (B_I - sum_{K > I} R_IK X_K), one wave per
// 16 COLUMNS of B (columns are independent), X_K in the accumulator layout (register r of lane (g, c): row g + 4 r, column c) -- the
// B operand of the products that use it; the A operands R_IK / T_II come from L2 with transposed addressing (A[i][k'] = block[i][g + 4 r]).
// Epilogue as k_trsm_pack's: X (fp64) and the three bf16 operand planes of the product that follows (k_pack_p3's layout), through
// one LDS transpose.  M (the order of R) = 16 NB <= 144 is also the row count of B; rows of the planes beyond M are zeros.
template <int NB>
__global__ __launch_bounds__(64) void k_trsm_left_pack(const double* __restrict__ RT, int64_t ldt, const double* __restrict__ B, int64_t ldb,
                                                       double* __restrict__ X_out, int64_t ldxo, bf16x8* __restrict__ pk3, int NTtot) {
    constexpr int M = 16 * NB, MP = (M + 31) / 32 * 32;
    __shared__ double sX[MP * 17];
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const int64_t j0 = (int64_t)blockIdx.x * 16;
    cf64x4 X[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[I][r] = B[(int64_t)(16 * I + lk + 4 * r) * ldb + j0 + li];
    // block row I of RT (its blocks K >= I), transposed addressing, two register sets (as k_trsm_pack: the next row is requested while
    // this one is used)
    cf64x4 rt[2][NB];
    auto load_row = [&](int I, cf64x4(&dst)[NB]) {
        const double* rrow = RT + (int64_t)(16 * I + li) * ldt + lk;
#pragma unroll
        for (int K = 0; K < NB; ++K)
            if (K >= I) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[K][r] = rrow[16 * K + 4 * r];
            }
    };
    load_row(NB - 1, rt[(NB - 1) & 1]);
#pragma unroll
    for (int I = NB - 1; I >= 0; --I) {
        if (I > 0) load_row(I - 1, rt[(I - 1) & 1]);
        const cf64x4(&cur)[NB] = rt[I & 1];
        cf64x4 acc = X[I];
#pragma unroll
        for (int K = I + 1; K < NB; ++K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-cur[K][r], X[K][r], acc, 0, 0, 0);
        }
        cf64x4 xi = cf64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) xi = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][r], acc[r], xi, 0, 0, 0);
        X[I] = xi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sX[(16 * I + lk + 4 * r) * 17 + li] = xi[r];
            if (X_out) X_out[(int64_t)(16 * I + lk + 4 * r) * ldxo + j0 + li] = xi[r];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MP > M)
        for (int e = lane; e < (MP - M) * 17; e += 64) sX[M * 17 + e] = 0.0;
    __syncthreads();
    if (pk3) {
#pragma unroll
        for (int cc = 0; cc < MP / 32; ++cc) {                    // item = 8 rows (group lk of the 32-row chunk) of column li
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)sX[(32 * cc + 8 * lk + e) * 17 + li];
            bf16x8 h, m, l;
            split3(x, h, m, l);
            const int64_t tile = (int64_t)cc * NTtot + blockIdx.x;
            pk3[(tile * 3 + 0) * 64 + lane] = h;
            pk3[(tile * 3 + 1) * 64 + lane] = m;
            pk3[(tile * 3 + 2) * 64 + lane] = l;
        }
    }
}

constexpr int CHOL_THREADS = 512;
// ---- fast Cholesky-inverse for L <= 140: R and T both packed COLUMN-major in LDS (cp(k, c) = c (c + 1) / 2 + k, k <= c),
// so every inner product below walks contiguous words with incremental addresses (no index arithmetic in the loops);
// the 16 x 16 diagonal block is factored AND inverted in the registers of one wave (column c in lane c, pivots and
// multipliers broadcast with v_readlane, reciprocal square roots instead of sqrt + divisions), and the panel solve
// becomes a product with the inverted diagonal block.  Same contract as k_chol_inv.
__device__ __forceinline__ double readlane_d(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
// 1 / sqrt(d) for a positive, normal d: the hardware estimate (v_rsq_f64) and two Newton steps y <- y + (y / 2)(1 - d y^2) -- ~10
// dependent instructions where the library routine's range handling makes it several times that, and it sits on the
// critical path of every Cholesky pivot
__device__ __forceinline__ double fast_rsqrt_pos(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-d * y, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    return y;
}
__device__ __forceinline__ double bperm_d(double x, int src_lane) {   // x of lane src_lane (per-lane index), through the LDS crossbar
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int cp(int k, int c) { return (c * (c + 1)) / 2 + k; }
typedef double cf64x4 __attribute__((ext_vector_type(4)));
// element (r, c) of an upper-triangular matrix kept column-packed; zero below the diagonal and outside the L x L matrix
__device__ __forceinline__ double ld_up(const double* P, int r, int c, int L) { return (r <= c && c < L) ? P[cp(r, c)] : 0.0; }
constexpr int CHOL2_MAXL = 140;
__host__ __device__ inline size_t chol2_lds_bytes(int L) {
    return sizeof(double) * ((size_t)L * (L + 1) + 2 * (size_t)L) + sizeof(int) * (size_t)L;
}
// gd_ref (nullable): the diagonal the pivots are compared with (rel_tol * gd_ref[j]) when G is a Schur complement of a
// larger matrix (blocked factorisation of L > 200: the dependence test stays relative to the ORIGINAL diagonal)
// The factorisation proper, on a column-packed upper triangle that is already in LDS (Rc filled, Tc zeroed, gd = the reference
// diagonal).  Collective over the workgroup; starts with a barrier.  (Split from its loader in round 4 for a kernel that fed it
// from in-launch partial sums -- measured slower than the separate launches, EXPERIMENTS.md -- and kept as the cleaner shape.)
__device__ __forceinline__ void chol2_body(double* sm_chol, int L, double* __restrict__ T, int64_t ldt, double rel_tol,
                                           int* __restrict__ ndead_out, int Lz, int rt_form, int ncount) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int np = L * (L + 1) / 2;
    double* Rc = sm_chol;
    double* Tc = Rc + np;
    double* gd = Tc + np;
    double* rinv = gd + L;
    int* dead = reinterpret_cast<int*>(rinv + L);
    for (int e = tid; e < Lz * Lz; e += nt) {  // zero padding of the output beyond the factored block
        const int r = e / Lz, c = e - r * Lz;
        if (r >= L || c >= L) T[(int64_t)r * ldt + c] = 0.0;
    }
    __syncthreads();
#ifdef PETAL_DEBUG_COUNTERS
    long long _t0 = clock64();
#endif
    const int nb = (L + 15) / 16;
    for (int J = 0; J < nb; ++J) {
        const int jb = 16 * J;
        // (the reference diagonal of this block's pivots, requested before the update phase so that its LDS latency is not paid
        // in the factoring wave's chain; broadcast reads, only wave 0 uses them)
        double gdv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gdv[i] = (tid < 64 && jb + i < L) ? gd[jb + i] : 0.0;
        // (1) block row J -= (finished rows above)^T (finished rows above), one 16 x 16 tile per wave pass on the fp64 matrix
        //     cores: C(J, Ct) -= sum_K R_KJ^T R_K,Ct.  MFMA 16x16x4 f64: lane l feeds A[l & 15][l >> 4], B[l >> 4][l & 15];
        //     register r of lane l is C[(l >> 4) + 4 r][l & 15].
        // The diagonal tile (Ct = J, always wave 0's first) does not go back to LDS: the accumulator layout -- register r of lane
        // (lk, li) = row lk + 4 r, column li -- IS the layout the factorisation below holds its block in, so wave 0 subtracts it
        // from the old block in registers and factors straight on while the other waves finish their tiles (no tile of theirs
        // is read by the factorisation, and none of them reads the diagonal tile): one barrier per block row instead of two.
        cf64x4 dacc = cf64x4{0.0, 0.0, 0.0, 0.0};
        if (jb > 0) {
            const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
            for (int Ct = J + (tid >> 6); Ct < nb; Ct += nt >> 6) {
                cf64x4 acc = cf64x4{0.0, 0.0, 0.0, 0.0};
                // (jb is a multiple of 16: the operands of a whole 16-row block are requested before its four MFMAs, so the
                // LDS latency is paid once per block instead of once per k-step)
                for (int k0 = 0; k0 < jb; k0 += 16) {
                    double pa[4], pb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {  // rows k0 + 4 u + lk of R: (R_KJ^T)[i][k] = R[k][jb + i]
                        pa[u] = ld_up(Rc, k0 + 4 * u + lk, jb + li, L);
                        pb[u] = ld_up(Rc, k0 + 4 * u + lk, 16 * Ct + li, L);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[u], pb[u], acc, 0, 0, 0);
                }
                if (Ct == J) { dacc = acc; continue; }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = jb + lk + 4 * r, cc = 16 * Ct + li;
                    if (rr <= cc && cc < L) Rc[cp(rr, cc)] -= acc[r];
                }
            }
        }
        DBG_T(8);
        // (2) diagonal block in the registers of wave 0, right-looking.  Lane (g, c) = (lane >> 4, lane & 15) holds the four rows
        //     g, g + 4, g + 8, g + 12 of column jb + c of the (symmetric) Schur complement, so a pivot step needs FIVE values from
        //     other lanes -- the scaled pivot row at its own column and at its four rows -- through the LDS crossbar
        //     (ds_bpermute: no SGPR round trip), where the one-column-per-lane form needed 15 - i multipliers by v_readlane pairs
        //     (~460 cycles per pivot, 40 % of this kernel).  Same operations on the same operands in the same order: the upper
        //     triangle comes out bit-identical.  Reciprocal square roots instead of sqrt + divisions.
        if (tid < 64) {
            const int c = tid & 15, g = tid >> 4, col = jb + c;
            double a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = g + 4 * m, row = jb + k;
                a[m] = (col < L && row < L) ? (k <= c ? Rc[cp(row, col)] : Rc[cp(col, row)]) - dacc[m] : 0.0;
            }
            double myinv = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gi = i & 3, mi = i >> 2;
                const double dii = readlane_d(a[mi], gi * 16 + i);
                const bool ok = (gdv[i] > 0.0) && (dii > rel_tol * gdv[i]);
                // the UNSCALED pivot row travels while the reciprocal square root is being computed (the exchange is off the
                // critical path) and is scaled where it arrives: the same products as scaling at the source, bit for bit
                const double su_c = bperm_d(a[mi], gi * 16 + c);  // S[i][my column]
                double su_k[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) su_k[m] = (4 * m + 3 > i) ? bperm_d(a[mi], gi * 16 + g + 4 * m) : 0.0;   // S[i][my row k]
                const double rs = fast_rsqrt_pos(ok ? dii : 1.0);
                const double inv = ok ? rs : 0.0;                 // (a select, not a branch: 40 cycles per pivot)
                const double r_own = a[mi] * inv;                 // row i of R, in the lanes that hold it (g == gi)
                const double rc = su_c * inv;                     // R[i][my column]
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int k = g + 4 * m;
                    const double rk = su_k[m] * inv;              // R[i][my row k]
                    if (m == mi) a[m] = (g == gi) ? r_own : (k > i ? a[m] - rk * rc : a[m]);
                    else if (4 * m + 3 > i) a[m] = (k > i) ? a[m] - rk * rc : a[m];
                }
                myinv = (tid == i) ? inv : myinv;                 // (kept in lane i: LDS writes inside the chain cost as much again)
            }
            if (tid < 16 && col < L) { rinv[col] = myinv; dead[col] = myinv > 0.0 ? 0 : 1; }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = g + 4 * m;
                if (k <= c && col < L) Rc[cp(jb + k, col)] = a[m];
            }
        }
        __syncthreads();
        DBG_T(9);
        // (3) panel right of the block by forward substitution, one thread per column (right-looking in registers; the
        //     multipliers R[jb + k][jb + i] are broadcast LDS reads, independent of the dependency chain)
        if (jb + 16 < L) {
            const int c = jb + 16 + tid;
            if (c < L) {
                double v[16];
                double* pc = Rc + (c * (c + 1)) / 2 + jb;
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = pc[k];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    v[k] *= rinv[jb + k];
#pragma unroll
                    for (int i = k + 1; i < 16; ++i) v[i] -= Rc[cp(jb + k, jb + i)] * v[k];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) pc[k] = v[k];
            }
            __syncthreads();
        }
        DBG_T(10);
    }
    // inverses of all diagonal blocks at once, one wave per block: column c of T_JJ in lane c, back substitution
    // right-looking (column k of R updates every partial sum: independent FMAs), multipliers as broadcast LDS reads
    for (int J = tid >> 6; J < nb; J += nt >> 6) {
        const int jb = 16 * J, ln = tid & 63, c = jb + ln;
        const int kcl = min(jb + 15, L - 1);
        double t[16], acc[16], rv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] = 0.0; rv[i] = rinv[min(jb + i, kcl)] * ((jb + i <= kcl) ? 1.0 : 0.0); }
#pragma unroll
        for (int k = 15; k >= 0; --k) {
            t[k] = (k == ln) ? rv[k] : (k < ln ? -rv[k] * acc[k] : 0.0);
            const double* pk_ = Rc + cp(jb, min(jb + k, kcl));  // R[jb + i][jb + k], contiguous in i
#pragma unroll
            for (int i = 0; i < k; ++i) acc[i] += pk_[i] * t[k];
        }
        if (ln < 16 && c < L) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i <= ln) Tc[cp(jb + i, c)] = t[i];
        }
    }
    __syncthreads();
    DBG_T(11);
    if (tid < 64 && ndead_out) {   // (wave 0 counts: a serial loop over L LDS reads by one thread was 2 us)
        int cdead = 0;
        for (int j = tid; j < min(L, ncount); j += 64) cdead += dead[j];   // (ncount: only the columns whose loss the caller minds)
        for (int off = 32; off > 0; off >>= 1) cdead += __shfl_down(cdead, off, 64);
        if (tid == 0 && cdead > *ndead_out) *ndead_out = cdead;
    }
    if (rt_form) {
        // "RT form": the caller applies R^-1 by a blocked triangular solve (k_trsm_pack) and only needs the inverses of the
        // diagonal blocks -- the explicit off-diagonal inverse below (30 % of this kernel) is skipped.  Output: diagonal 16 x 16
        // blocks = T_JJ = R_JJ^-1, blocks above them = R itself, zeros below.
        for (int e = tid; e < L * L; e += nt) {
            const int r = e / L, c = e - r * L;
            T[(int64_t)r * ldt + c] = c >= r ? (((r >> 4) == (c >> 4)) ? Tc[cp(r, c)] : Rc[cp(r, c)]) : 0.0;
        }
        return;
    }
    // ---- off-diagonal blocks of T = R^-1: T_IJ = -T_II sum_{K = I+1 .. J} R_IK T_KJ, as 16 x 16 block products on the fp64
    //      matrix cores.  First R~_IK = T_II R_IK for every off-diagonal block (in place: a block is read and written by
    //      one wave only), then block super-diagonal by block super-diagonal T_IJ = -sum_K R~_IK T_KJ.
    {
        const int lane = tid & 63, li = lane & 15, lk = lane >> 4, wv = tid >> 6, nw = nt >> 6;
        for (int blk = wv; blk < nb * (nb - 1) / 2; blk += nw) {
            int bI = blk, dl = 1;
            while (bI >= nb - dl) { bI -= nb - dl; ++dl; }
            const int bK = bI + dl;
            cf64x4 acc = cf64x4{0.0, 0.0, 0.0, 0.0};
            {
                double pa[4], pb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // A = T_II (upper triangular), B = R_IK
                    pa[u] = ld_up(Tc, 16 * bI + li, 16 * bI + 4 * u + lk, L);
                    pb[u] = ld_up(Rc, 16 * bI + 4 * u + lk, 16 * bK + li, L);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[u], pb[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * bI + lk + 4 * r, cc = 16 * bK + li;
                if (cc < L) Rc[cp(rr, cc)] = acc[r];
            }
        }
        __syncthreads();
        DBG_T(12);
        for (int dl = 1; dl < nb; ++dl) {
            for (int bI = wv; bI < nb - dl; bI += nw) {
                const int bJ = bI + dl;
                cf64x4 acc = cf64x4{0.0, 0.0, 0.0, 0.0};
                for (int bK = bI + 1; bK <= bJ; ++bK) {
                    double pa[4], pb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {  // A = R~_IK, B = T_KJ (upper triangular when K == J)
                        pa[u] = ld_up(Rc, 16 * bI + li, 16 * bK + 4 * u + lk, L);
                        pb[u] = ld_up(Tc, 16 * bK + 4 * u + lk, 16 * bJ + li, L);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[u], pb[u], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * bI + lk + 4 * r, cc = 16 * bJ + li;
                    if (cc < L) Tc[cp(rr, cc)] = -acc[r];
                }
            }
            __syncthreads();
        }
    }
    DBG_T(13);
    for (int e = tid; e < L * L; e += nt) {
        const int r = e / L, c = e - r * L;
        T[(int64_t)r * ldt + c] = c >= r ? Tc[cp(r, c)] : 0.0;
    }
}

__global__ __launch_bounds__(CHOL_THREADS) void k_chol_inv2(const double* __restrict__ G, int L, int64_t ldg, double* __restrict__ T,
                                                            int64_t ldt, double rel_tol, int* __restrict__ ndead_out, int Lz,
                                                            const double* __restrict__ gd_ref, int rt_form, int ncount) {
    extern __shared__ __attribute__((aligned(16))) double sm_chol[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int np = L * (L + 1) / 2;
    double* Rc = sm_chol;
    double* Tc = Rc + np;
    double* gd = Tc + np;
    // (loads in batches of four: a runtime-trip loop of load -> LDS-store pairs pays an L2 round trip per pair, eleven in a row
    // at l = 74)
    for (int e0 = tid; e0 < L * L; e0 += 4 * nt) {
        double gv[4];
        int rr[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nt;
            rr[u] = e / L; cc[u] = e - rr[u] * L;
            gv[u] = (e < L * L && cc[u] >= rr[u]) ? G[(int64_t)rr[u] * ldg + cc[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e0 + u * nt < L * L && cc[u] >= rr[u]) {
                Rc[cp(rr[u], cc[u])] = gv[u];
                Tc[cp(rr[u], cc[u])] = 0.0;
                if (cc[u] == rr[u]) gd[rr[u]] = gd_ref ? gd_ref[rr[u]] : gv[u];
            }
        }
    }
    chol2_body(sm_chol, L, T, ldt, rel_tol, ndead_out, Lz, rt_form, ncount);
}

// ---- k_chol_rt4<NB>: the re-basing Cholesky of the power iterations, register-resident on four waves (round 6) ----------------
// G = R^T R for an order L <= 16 NB, "RT form" output (diagonal 16 x 16 blocks: T_JJ = R_JJ^-1, blocks above them: R, zeros below:
// what k_trsm_pack reads).  k_chol_inv2 keeps the matrix in LDS and spends a third of its 28 us at l = 74 in barriers and LDS round
// trips between its phases.  Here block (I, J), I <= J, of the working matrix lives in REGISTERS in the accumulator layout of
// v_mfma_f64_16x16x4_f64 -- register m of lane (g, c) = (lane >> 4, lane & 15) is element (g + 4 m, c) of the block -- which is at once
//   * the layout the 16 x 16 diagonal factorisation wants: four rows of a column per lane; the multiplier of row k, S[k][i], sits in
//     column i of the symmetric block, i.e. in lane i of the 16-lane row that holds row k -- a DPP row broadcast (row_newbcast, the
//     one DPP control 64-bit operands have), where k_chol_inv2 sends five values per pivot through the LDS crossbar;
//   * the B operand of an MFMA for the block itself and the A operand for its TRANSPOSE, k-slots declared as k' = (l >> 4) + 4 r,
// so the panel R_JK = T_JJ^T S_JK and the trailing update S_KM -= R_JK^T R_JM take the registers as they are; T_JJ^T is what the
// elimination of [S_JJ | I] leaves, one 2-KB LDS transpose turns it into T_JJ.  Blocks not yet factored are kept NEGATED (N = -S):
// the trailing update is a plain accumulation.  One Newton step on v_rsq_f64 (1.5 ulp): the factor only has to keep the re-based
// iterate well conditioned (backward error 6e-16 either way, dev/chol1w.hip).
// A wave cannot overlap its own MFMAs with its own VALU work (a ONE-wave form of this kernel, dev/chol1w_kernel.h: 12.0 us over an
// empty kernel at l = 74, 41.7 at l = 138, and 78 in the fit, where its 75 KB of straight-line code arrive cold; interleaving the
// MFMAs between the pivots gains nothing), so the three other matrix pipes of the CU take the panel and the trailing update while
// wave 0 runs nothing but the pivot chain (280 cycles a pivot: readlane -> rsq -> Newton -> scale -> update, ~35 instructions):
//   wave 0   factors the diagonal block D_J, publishes -T_JJ, forms R_J,J+1 and N_J+1,J+1 += R_J,J+1^T R_J,J+1 from the two blocks
//            the owner of column J + 1 handed over, goes on;
//   wave h   (1..3) owns whole block COLUMNS M >= 2: after -T_JJ is out it forms its panel blocks R_JM, publishes them, and after every
//            panel is out updates its blocks N_KM += R_JK^T R_JM.
// Two workgroup barriers per block row; what crosses them goes through LDS in register order (lane-contiguous 8-byte words: no bank
// conflicts, no layout change).  Measured (dev/chol1w.hip, over an empty kernel): 9.5 us at l = 74 (k_chol_inv2: 25), 21.4 us at
// l = 138 (72).
// compile-time loops: every block index is a constant expression, so the blocks are registers (a #pragma unroll the optimizer gives
// up on turns the whole array into scratch memory)
template <int I0, int I1, class F>
__device__ __forceinline__ void chol_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        chol_static_for<I0 + 1, I1>(f);
    }
}
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

// ---- convergence of a Jacobi sweep, graded matrices included ------------------------------------------------------------
// Every matrix these solvers see is a Gram matrix (PSD, often with eigenvalues spread over many decades: B B^T of the
// randomized SVD, the covariance of exact Pca / FastICA whitening).  A stopping rule on ||off||_F / ||diag||_F, the
// usual one, declares such a matrix finished while the rows of its SMALL eigenvalues are still coupled -- at
// off = 1e-15 ||diag|| an eigenvalue of 1e-7 lambda_1 has lost half its digits (measured: singular values below
// 10^-3.5 sigma_1 wrong in the 5th digit).  Element (p, q) is therefore finished when
//     a_pq^2 <= tol^2 |a_pp a_qq|        (relative to ITS rows: the Demmel-Veselic criterion, which is what gives two-sided
//                                          Jacobi its high relative accuracy on positive definite matrices), or
//     a_pq^2 <= (1e-16 ||diag||_F)^2      (below the rounding noise of the matrix itself: exact zeros / rank deficiency).
// Returns the largest a_pq^2 / limit over the upper triangle: <= 1 means converged.  Reads r <= c only (the split solver keeps the upper triangle).
// Collective over the workgroup; s_red needs 64 doubles.
__device__ double wg_jacobi_violation(const double* A, int64_t ld, int L, double tol_rel, double* s_red) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = (nt + 63) >> 6;
    double dg = 0;
    for (int e = tid; e < L; e += nt) { const double v = A[e * ld + e]; dg += v * v; }
    for (int off = 32; off > 0; off >>= 1) dg += __shfl_down(dg, off, 64);
    if (lane == 0) s_red[wv] = dg;
    __syncthreads();
    double tdg = 0;
    for (int x = 0; x < nw; ++x) tdg += s_red[x];
    __syncthreads();
    if (!(tdg > 0.0) || !(tdg < 1e300)) return 0.0;  // zero or non-finite matrix: nothing a rotation could improve
    const double floor2 = 1e-32 * tdg, tol2 = tol_rel * tol_rel;
    double viol = 0;
    {
        int r = tid / L, c = tid - r * L;
        const int dr = nt / L, dc = nt - dr * L;
        for (int e = tid; e < L * L; e += nt) {
            if (c > r) {
                const double v = A[r * ld + c];
                if (v != 0.0) {
                    const double lim = fmax(tol2 * fabs(A[r * ld + r] * A[c * ld + c]), floor2);
                    viol = fmax(viol, v * v / lim);
                }
            }
            r += dr; c += dc;
            if (c >= L) { c -= L; ++r; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) viol = fmax(viol, __shfl_down(viol, off, 64));
    if (lane == 0) s_red[wv] = viol;
    __syncthreads();
    double tv = 0;
    for (int x = 0; x < nw; ++x) tv = fmax(tv, s_red[x]);
    __syncthreads();
    return tv;
}

// ---- workgroup-wide cyclic Jacobi eigen-solver (fp64) ----------------------------------------------
// A (L x L, lda) symmetric, destroyed; V (L x L, ldv) <- eigenvectors in columns.  Parallel ordering:
// round-robin tournament, L/2 disjoint rotations per round, three barriers per round.
__device__ void wg_jacobi(double* A, int64_t lda, double* V, int64_t ldv, int L, double* s_c, double* s_s, int* s_p, int* s_q,
                          double* s_red, double tol_rel) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < L * L; e += nt) V[(int64_t)(e / L) * ldv + (e % L)] = (e / L == e % L) ? 1.0 : 0.0;
    __syncthreads();
    if (L < 2) return;
    const int Le = (L + 1) & ~1, half = Le / 2, rounds = Le - 1;
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double viol = wg_jacobi_violation(A, lda, L, tol_rel, s_red);
        // (no "nearly there: one more sweep" shortcut: with clustered or tiny eigenvalues the step after |a_pq| ~ sqrt(tol) is
        // NOT at tol -- a_pq^2 / gap -- and stopping there left 3e-4 in the eigenvectors of a graded test matrix)
        if (!(viol > 1.0)) break;
        for (int rd = 0; rd < rounds; ++rd) {
            for (int k = tid; k < half; k += nt) {
                int p, q;
                if (k == 0) { p = Le - 1; q = rd; }
                else { p = (rd + k) % (Le - 1); q = (rd - k + (Le - 1)) % (Le - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, s = 0.0;
                if (q < L) {
                    const double apq = A[(int64_t)p * lda + q];
                    if (apq != 0.0) {
                        const double theta = (A[(int64_t)q * lda + q] - A[(int64_t)p * lda + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        s = t * c;
                    }
                } else { q = p; }  // bye
                s_p[k] = p; s_q[k] = q; s_c[k] = c; s_s[k] = s;
            }
            __syncthreads();
            for (int e = tid; e < L * half; e += nt) {  // columns: A <- A J, V <- V J
                const int r = e / half, k = e % half;
                const int p = s_p[k], q = s_q[k];
                if (p == q) continue;
                const double c = s_c[k], s = s_s[k];
                const double ap = A[(int64_t)r * lda + p], aq = A[(int64_t)r * lda + q];
                A[(int64_t)r * lda + p] = c * ap - s * aq;
                A[(int64_t)r * lda + q] = s * ap + c * aq;
                const double vp = V[(int64_t)r * ldv + p], vq = V[(int64_t)r * ldv + q];
                V[(int64_t)r * ldv + p] = c * vp - s * vq;
                V[(int64_t)r * ldv + q] = s * vp + c * vq;
            }
            __syncthreads();
            for (int e = tid; e < L * half; e += nt) {  // rows: A <- J^T A
                const int k = e / L, cc = e % L;
                const int p = s_p[k], q = s_q[k];
                if (p == q) continue;
                const double c = s_c[k], s = s_s[k];
                const double ap = A[(int64_t)p * lda + cc], aq = A[(int64_t)q * lda + cc];
                A[(int64_t)p * lda + cc] = c * ap - s * aq;
                A[(int64_t)q * lda + cc] = s * ap + c * aq;
            }
            __syncthreads();
        }
    }
}
// LDS-resident variant for L <= 16 MB: A and V are L x L with an odd leading dimension.  Two barriers per round:
//   (1) one thread per pair computes its rotation (c, s) from A[p][p], A[q][q], A[p][q];
//   (2) A <- J^T A J in ONE pass over 2 x 2 blocks -- the block at rows (p, q) of pair k and columns (p', q') of pair k'
//       is read once, rotated from both sides in registers and written back in place (no other thread touches it) --
//       and V <- V J column-wise in the same phase.
// Every thread gathers all its operands into registers first and scatters afterwards, so the LDS round trips overlap
// instead of serialising behind possibly-aliasing stores; (c, s) and (p, q) are packed for 16-B / 8-B loads.
template <int MB>
__device__ void wg_jacobi_fast(double* A, double* V, int L, double* s_c, double* s_s, int* s_p, int* s_q, double* s_red,
                               double tol_rel) {
    constexpr int MB2 = (MB + 1) / 2;  // 16-wide groups of pairs: half <= 8 MB
    const int LD = L | 1;  // odd leading dimension: column accesses (stride LD doubles) spread over all LDS banks
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    f64x2* s_cs = reinterpret_cast<f64x2*>(s_c);  // [half] (c, s)   (s_s == s_c + half: the two arrays are contiguous)
    i32x2* s_pq = reinterpret_cast<i32x2*>(s_p);  // [half] (p, q)   (s_q == s_p + half)
    (void)s_s; (void)s_q;
    for (int e = tid; e < L * LD; e += nt) V[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < L; e += nt) V[e * LD + e] = 1.0;
    __syncthreads();
    if (L < 2) return;
    const int Le = (L + 1) & ~1, half = Le / 2, rounds = Le - 1;
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double viol = wg_jacobi_violation(A, LD, L, tol_rel, s_red);
#ifdef PETAL_DEBUG_COUNTERS
        if (tid == 0) g_dbg[0] = sweep;
#endif
        // (no "nearly there: one more sweep" shortcut: with clustered or tiny eigenvalues the step after |a_pq| ~ sqrt(tol) is
        // NOT at tol -- a_pq^2 / gap -- and stopping there left 3e-4 in the eigenvectors of a graded test matrix)
        if (!(viol > 1.0)) break;
        for (int rd = 0; rd < rounds; ++rd) {
#ifdef PETAL_DEBUG_COUNTERS
            long long _t0 = clock64();
#endif
            if (tid < half) {
                const int k = tid;
                int p, q;
                if (k == 0) { p = Le - 1; q = rd; }
                else { p = rd + k; if (p >= Le - 1) p -= Le - 1; q = rd - k; if (q < 0) q += Le - 1; }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, sn = 0.0;
                if (q < L) {
                    const double apq = A[p * LD + q];
                    if (apq != 0.0) {
                        const double theta = (A[q * LD + q] - A[p * LD + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        sn = t * c;
                    }
                } else { q = p; }  // bye: identity rotation on a single index
                s_pq[k] = i32x2{p, q};
                s_cs[k] = f64x2{c, sn};
            }
            DBG_T(0);
            __syncthreads();
            DBG_T(1);
            {   // A <- J^T A J, one 2 x 2 block per (pair k, pair k')
                const int k0 = tid & 15;
                for (int k = tid >> 4; k < half; k += nt >> 4) {
                    const i32x2 pq = s_pq[k];
                    const f64x2 cs = s_cs[k];
                    double* rp = A + pq[0] * LD;
                    double* rq = A + pq[1] * LD;
                    double a[MB2], b[MB2], cc[MB2], d[MB2], c2[MB2], s2[MB2];
                    int p2[MB2], q2[MB2];
#pragma unroll
                    for (int m = 0; m < MB2; ++m) {
                        const int kp = min(k0 + 16 * m, half - 1);
                        const i32x2 pq2 = s_pq[kp];
                        const f64x2 cs2 = s_cs[kp];
                        p2[m] = pq2[0]; q2[m] = (k0 + 16 * m < half) ? pq2[1] : -1;
                        c2[m] = cs2[0]; s2[m] = cs2[1];
                        a[m] = rp[pq2[0]]; b[m] = rp[pq2[1]]; cc[m] = rq[pq2[0]]; d[m] = rq[pq2[1]];
                    }
#pragma unroll
                    for (int m = 0; m < MB2; ++m) {
                        if (q2[m] < 0) continue;
                        // right rotation (columns p', q'), then left rotation (rows p, q)
                        const double a1 = c2[m] * a[m] - s2[m] * b[m], b1 = s2[m] * a[m] + c2[m] * b[m];
                        const double c1 = c2[m] * cc[m] - s2[m] * d[m], d1 = s2[m] * cc[m] + c2[m] * d[m];
                        rp[p2[m]] = cs[0] * a1 - cs[1] * c1;
                        rp[q2[m]] = cs[0] * b1 - cs[1] * d1;
                        rq[p2[m]] = cs[1] * a1 + cs[0] * c1;
                        rq[q2[m]] = cs[1] * b1 + cs[0] * d1;
                    }
                }
            }
            {   // V <- V J (columns)
                const int kk = tid & 7;
                for (int r = tid >> 3; r < L; r += nt >> 3) {
                    double vp[MB], vq[MB], cc[MB], sn[MB];
                    int pp[MB], qq[MB];
                    double* vr = V + r * LD;
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const int k = min(kk + 8 * m, half - 1);
                        const i32x2 pq = s_pq[k];
                        const f64x2 cs = s_cs[k];
                        pp[m] = pq[0];
                        qq[m] = (kk + 8 * m < half && pq[0] != pq[1]) ? pq[1] : -1;
                        cc[m] = cs[0]; sn[m] = cs[1];
                        vp[m] = vr[pq[0]]; vq[m] = vr[pq[1]];
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        if (qq[m] >= 0) {
                            vr[pp[m]] = cc[m] * vp[m] - sn[m] * vq[m];
                            vr[qq[m]] = sn[m] * vp[m] + cc[m] * vq[m];
                        }
                    }
                }
            }
            DBG_T(2);
            __syncthreads();
            DBG_T(3);
        }
    }
}
// MB == 0: matrices in global memory (any L), generic loops; MB > 0: LDS-resident fast path for L <= 16 MB
template <int MB>
__device__ __forceinline__ void wg_jacobi_any(double* A, int64_t lda, double* V, int64_t ldv, int L, double* s_c, double* s_s,
                                              int* s_p, int* s_q, double* s_red, double tol_rel = 1e-15) {
    if constexpr (MB == 0) wg_jacobi(A, lda, V, ldv, L, s_c, s_s, s_p, s_q, s_red, tol_rel);
    else wg_jacobi_fast<MB>(A, V, L, s_c, s_s, s_p, s_q, s_red, tol_rel);
}
// sort eigenpairs descending: Vout[:, rank] = V[:, j], w[rank] = A[j][j]
__device__ void wg_sort_eig(const double* A, int64_t lda, const double* V, int64_t ldv, int L, double* Vout, int64_t ldo,
                            double* w, int* s_rank) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = tid; j < L; j += nt) {
        const double wj = A[(int64_t)j * lda + j];
        int rank = 0;
        for (int k = 0; k < L; ++k) {
            const double wk = A[(int64_t)k * lda + k];
            rank += (wk > wj || (wk == wj && k < j)) ? 1 : 0;
        }
        s_rank[j] = rank;
        w[rank] = wj;
    }
    __syncthreads();
    for (int e = tid; e < L * L; e += nt) {
        const int r = e / L, j = e % L;
        Vout[(int64_t)r * ldo + s_rank[j]] = V[(int64_t)r * ldv + j];
    }
    __syncthreads();
}

constexpr int EIG_MAXL = 8192;
struct JacWs { double* c; double* s; double* red; int* p; int* q; int* rank; };
__host__ __device__ inline size_t jac_ws_doubles(int L, int nthreads) {
    const int half = ((L + 1) & ~1) / 2;
    return (size_t)2 * half + 2 * (size_t)nthreads + (2 * (size_t)half + L + 1) / 2 + 1;
}
__device__ __forceinline__ JacWs jac_carve(double* base, int L, int nthreads) {
    const int half = ((L + 1) & ~1) / 2;
    JacWs w;
    w.c = base; w.s = w.c + half; w.red = w.s + half;
    w.p = reinterpret_cast<int*>(w.red + 2 * nthreads); w.q = w.p + half; w.rank = w.q + half;
    return w;
}
// MB > 0: A and the eigenvector accumulator live in LDS (2 L^2 doubles; L <= 16 MB, L <= 88); MB == 0: global memory
template <int MB>
__global__ __launch_bounds__(MB > 0 ? 768 : 1024) void k_eigh(double* A, int L, int64_t lda, double* Vtmp, double* V, int64_t ldv, double* w,
                                                                  double tol_rel, const int* flag) {
    if (flag && *flag == 0) return;
    extern __shared__ __attribute__((aligned(16))) double sm_eig[];
    const int tid = threadIdx.x, nt = blockDim.x;
    JacWs ws = jac_carve(sm_eig, L, nt);
    double* Aw = A; double* Vw = Vtmp; int64_t la = lda;
    if constexpr (MB > 0) {
        la = L | 1;
        Aw = sm_eig + jac_ws_doubles(L, nt); Vw = Aw + (size_t)L * la;
        for (int e = tid; e < L * L; e += nt) Aw[(e / L) * la + (e % L)] = A[(int64_t)(e / L) * lda + (e % L)];
        __syncthreads();
    }
    wg_jacobi_any<MB>(Aw, la, Vw, la, L, ws.c, ws.s, ws.p, ws.q, ws.red, tol_rel);
    wg_sort_eig(Aw, la, Vw, la, L, V, ldv, w, ws.rank);
}

// ---- split eigen-solver: rotations on A in one workgroup, eigenvector accumulation spread over the chip --------------
// The Jacobi rotations are decided by A alone; V <- V J only consumes (c, s) (the pairs (p, q) of a round are a pure
// function of the round number).  k_jacobi_a keeps A in LDS (so L up to 141 fits: no V beside it), runs the round-robin
// sweeps and LOGS every round's (c, s) to global memory; k_apply_rot then replays the log on the rows of V = I, one row
// per wave (rows are independent, so the replay runs on L waves in parallel instead of inside the single Jacobi
// workgroup), and scatters the columns into descending-eigenvalue order.
typedef double jf64x2 __attribute__((ext_vector_type(2)));
constexpr int JACA_MAX_SWEEPS = 24;
__host__ __device__ inline size_t jaca_lds_bytes(int L) {
    const int half = ((L + 1) & ~1) / 2;
    return sizeof(double) * ((size_t)L * (L | 1) + 2 * (size_t)half + 64);
}
// workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations (lgkmcnt) but leaves its global
// stores in flight, unlike __syncthreads(), whose fence also drains vmcnt
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// round-robin tournament: pair k of round rd (Le players, Le - 1 rounds); p < q, or p == q for a bye (odd L)
__device__ __forceinline__ void pair_pq(int k, int rd, int Le, int L, int& p, int& q) {
    if (k == 0) { p = Le - 1; q = rd; }
    else { p = rd + k; if (p >= Le - 1) p -= Le - 1; q = rd - k; if (q < 0) q += Le - 1; }
    if (p > q) { const int t = p; p = q; q = t; }
    if (q >= L) q = p;
}
// the same pairs, advanced round by round without divisions: raw players (rd + k, rd - k) mod (Le - 1)
struct PairIt {
    int pr, qr;
    __device__ __forceinline__ void init(int k, int rd, int Le) {
        const int m = Le - 1;
        pr = (rd + k) % m; qr = ((rd - k) % m + m) % m;
    }
    __device__ __forceinline__ void next(int Le) {
        if (++pr >= Le - 1) pr = 0;
        if (++qr >= Le - 1) qr = 0;
    }
    __device__ __forceinline__ void get(int k, int Le, int L, int& p, int& q) const {
        p = (k == 0) ? Le - 1 : pr;
        q = (k == 0) ? pr : qr;  // pair 0 is (Le - 1, rd) and pr == rd for k == 0
        if (p > q) { const int t = p; p = q; q = t; }
        if (q >= L) q = p;
    }
};
__device__ __forceinline__ void jacobi_angle(double app, double aqq, double apq, double& c, double& s, double& t) {
    const double d = aqq - app;
    const double m = fmax(fabs(d), fabs(apq));
    if (m > 1e-140 && m < 1e140) {
        // t = sgn(d) 2 apq / (|d| + sqrt(d^2 + 4 apq^2)); hardware rsq / rcp seeds with Newton steps instead of the
        // IEEE sqrt / divide sequences.  t must be the root to full precision, not just c: the caller sets A[p][q] = 0 and
        // updates the diagonal in closed form, so an error delta in t leaves an unrecorded residual ~ delta ||A|| per
        // rotation (one Newton step each left 2e-13: eigenvalues below 1e-7 lambda_1 lost half their digits -- the
        // singular-value cliff at 10^-3.5 sigma_1 that tests/test_gpu_parity.py::test_gram_route_singular_value_floor found)
        const double x = d * d + 4.0 * apq * apq;
        double rs = __builtin_amdgcn_rsq(x);
        rs = rs * (1.5 - 0.5 * x * rs * rs);
        rs = rs * (1.5 - 0.5 * x * rs * rs);
        const double den = fabs(d) + x * rs;
        double r = __builtin_amdgcn_rcp(den);
        r = r * (2.0 - den * r);
        r = r * (2.0 - den * r);
        t = (d >= 0.0 ? 2.0 : -2.0) * apq * r;
    } else {
        const double theta = d / (2.0 * apq);
        t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    }
    const double y = t * t + 1.0;  // in [1, 2]
    double cr = __builtin_amdgcn_rsq(y);
    cr = cr * (1.5 - 0.5 * y * cr * cr);
    cr = cr * (1.5 - 0.5 * y * cr * cr);
    c = cr;
    s = t * cr;
}
// A is symmetric and stays symmetric, so only its UPPER triangle is kept and rotated: element {i, j} lives at
// (min, max).  A round touches every unordered pair of rotation pairs {k, k'} once (a 2 x 2 block, rotated from the left
// with k's and from the right with k''s angle), and the 2 x 2 diagonal block of each pair is updated in closed form by
// the thread that computes its angle.  Two thread roles:
//   angle threads (tid < half, the first PW threads): read (app, aqq, apq), compute (c, s), update the diagonal block;
//   block threads (tid >= PW): 16-lane group g owns the row pairs k = g and k = half-1-g, which together have exactly
//     half-1 partners k' > k.  Their operands do not depend on this round's angles, so addresses and the 2 x 2 blocks are
//     fetched BEFORE the barrier, under the angle threads' latency chain; after it only (c, s) are read and applied.
template <int MB2, int GW>  // GW-wide batches of partner pairs per lane: half - 1 <= GW MB2
__global__ __launch_bounds__(1024) void k_jacobi_a(const double* __restrict__ Ain, int L, int64_t lda, jf64x2* __restrict__ log_cs,
                                                   int* __restrict__ nrounds_out, double* __restrict__ w,
                                                   int* __restrict__ rank_out, int PW, double tol_rel, const int* __restrict__ flag) {
    if (flag && *flag == 0) return;  // the two-stage solver in front of this launch has already delivered (k_trieig_verdict)
    extern __shared__ __attribute__((aligned(16))) double sm_ja[];
    const int LD = L | 1;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    const int Le = (L + 1) & ~1, half = Le / 2, rounds = Le - 1;
    jf64x2* s_cs = reinterpret_cast<jf64x2*>(sm_ja);
    double* s_red = sm_ja + 2 * half;
    double* A = s_red + 64;
    {
        int r = tid / L, c = tid - r * L;
        const int dr = nt / L, dc = nt - dr * L;
        for (int e = tid; e < L * L; e += nt) {
            if (c >= r) A[r * LD + c] = Ain[(int64_t)r * lda + c];
            r += dr; c += dc;
            if (c >= L) { c -= L; ++r; }
        }
    }
    __syncthreads();
    // this block thread's slots j = k0 + 16 m: slot j < n1 -> (k = g, k' = g + 1 + j), else (k = gb, k' = gb + 1 + j - n1)
    const int bt = tid - PW, g = bt / GW, k0 = bt - g * GW;  // GW lanes per group of two row pairs
    const int gb = half - 1 - g, n1 = half - 1 - g;
    const int nslots = (bt < 0) ? 0 : ((g < gb) ? half - 1 : (g == gb ? n1 : 0));
    int R = 0;  // rounds logged so far
    for (int sweep = 0; sweep < JACA_MAX_SWEEPS && L >= 2; ++sweep) {
        const double viol = wg_jacobi_violation(A, LD, L, tol_rel, s_red);
        // (no "nearly there: one more sweep" shortcut: with clustered or tiny eigenvalues the step after |a_pq| ~ sqrt(tol) is
        // NOT at tol -- a_pq^2 / gap -- and stopping there left 3e-4 in the eigenvectors of a graded test matrix)
        if (!(viol > 1.0)) break;
#ifdef PETAL_DEBUG_COUNTERS
        if (tid == 0) g_dbg[0] = sweep + 1;
        long long _t0 = clock64();
#endif
        PairIt ita, itb, itk[MB2], itme;
        ita.init(max(g, 0), 0, Le); itb.init(max(gb, 0) % half, 0, Le); itme.init(min(tid, half - 1), 0, Le);
#pragma unroll
        for (int m = 0; m < MB2; ++m) {
            const int j = k0 + GW * m;
            itk[m].init(min(max(j < n1 ? g + 1 + j : gb + 1 + (j - n1), 0), half - 1), 0, Le);
        }
        for (int rd = 0; rd < rounds; ++rd, ++R) {
            double e00[MB2], e01[MB2], e10[MB2], e11[MB2];
            int i00[MB2], i01[MB2], i10[MB2], i11[MB2], kpv[MB2];
            if (tid < half) {
                int p, q;
                itme.get(tid, Le, L, p, q);
                double c = 1.0, sn = 0.0;
                if (p != q) {
                    // (all three operands requested at once: a dependent second LDS round trip costs ~100 cycles of the
                    // angle chain, which is the longest leg of a round)
                    const double apq = A[p * LD + q], app = A[p * LD + p], aqq = A[q * LD + q];
                    if (apq != 0.0) {
                        double t;
                        jacobi_angle(app, aqq, apq, c, sn, t);
                        A[p * LD + p] = app - t * apq;
                        A[q * LD + q] = aqq + t * apq;
                        A[p * LD + q] = 0.0;
                    }
                }
                s_cs[tid] = jf64x2{c, sn};
                log_cs[(size_t)R * half + tid] = jf64x2{c, sn};
            } else if (nslots > 0) {
                int pa, qa, pb, qb;
                ita.get(g, Le, L, pa, qa);
                itb.get(gb, Le, L, pb, qb);
#pragma unroll
                for (int m = 0; m < MB2; ++m) {
                    const int j = k0 + GW * m;
                    const bool first = j < n1;
                    const int kp = min(first ? g + 1 + j : gb + 1 + (j - n1), half - 1);
                    kpv[m] = kp;
                    const int p = first ? pa : pb, q = first ? qa : qb;
                    int p2, q2;
                    itk[m].get(kp, Le, L, p2, q2);
                    i00[m] = min(p, p2) * LD + max(p, p2);
                    i01[m] = min(p, q2) * LD + max(p, q2);
                    i10[m] = min(q, p2) * LD + max(q, p2);
                    i11[m] = min(q, q2) * LD + max(q, q2);
                    e00[m] = A[i00[m]]; e01[m] = A[i01[m]]; e10[m] = A[i10[m]]; e11[m] = A[i11[m]];
                }
            }
            DBG_T(0);
            lds_barrier();  // NOT __syncthreads(): that also waits for the global log stores (vmcnt), ~1 us per round
            DBG_T(1);
            if (nslots > 0) {
                const jf64x2 csa = s_cs[g], csb = s_cs[gb];
                jf64x2 cs2[MB2];
#pragma unroll
                for (int m = 0; m < MB2; ++m) cs2[m] = s_cs[kpv[m]];
#pragma unroll
                for (int m = 0; m < MB2; ++m) {
                    const int j = k0 + GW * m;
                    if (j >= nslots) continue;
                    const jf64x2 cs = (j < n1) ? csa : csb;
                    // right rotation (columns p', q') with the partner's angle, then left rotation (rows p, q) with ours
                    const double a1 = cs2[m][0] * e00[m] - cs2[m][1] * e01[m], b1 = cs2[m][1] * e00[m] + cs2[m][0] * e01[m];
                    const double c1 = cs2[m][0] * e10[m] - cs2[m][1] * e11[m], d1 = cs2[m][1] * e10[m] + cs2[m][0] * e11[m];
                    A[i00[m]] = cs[0] * a1 - cs[1] * c1;
                    A[i01[m]] = cs[0] * b1 - cs[1] * d1;
                    A[i10[m]] = cs[1] * a1 + cs[0] * c1;
                    A[i11[m]] = cs[1] * b1 + cs[0] * d1;
                }
            }
            DBG_T(2);
            lds_barrier();
            DBG_T(3);
            ita.next(Le); itb.next(Le); itme.next(Le);
#pragma unroll
            for (int m = 0; m < MB2; ++m) itk[m].next(Le);
        }
    }
    if (tid == 0) *nrounds_out = R;
    for (int j = tid; j < L; j += nt) {  // descending order: rank[j] = position of eigenvalue j
        const double wj = A[j * LD + j];
        int rank = 0;
        for (int k = 0; k < L; ++k) {
            const double wk = A[k * LD + k];
            rank += (wk > wj || (wk == wj && k < j)) ? 1 : 0;
        }
        rank_out[j] = rank;
        w[rank] = wj;
    }
}

// ================================================================================================
// two-stage symmetric eigen-solver: Householder tridiagonalisation, then per eigenpair (one wave each, spread over the
// chip) multisection on the Sturm count, the eigenvector from the twisted factorisation, and the back-transformation.
// ================================================================================================
// The Jacobi solvers above spend ~1 us per round on a chain of on-chip latencies, (L - 1) rounds per sweep, 4-5 sweeps:
// 0.36 ms at L = 74, 1.8 ms at L = 138, all on ONE workgroup.  Here the only serial part is the reduction to tridiagonal
// form T = Q^T A Q (L - 2 Householder steps on one workgroup); everything after it is independent per eigenpair:
//   lambda_j   64-way multisection of the Sturm count of T (6 bits per pass, 10 passes to fp64 resolution),
//   z_j        twisted factorisation (Parlett-Dhillon): the two Sturm recurrences from the top and from the bottom meet at
//              the index of smallest |gamma|; one sweep each way gives the eigenvector of T without iteration,
//   v_j        = H_0 ... H_{L-3} z_j, the reflectors applied to that one column.
// Accuracy is that of a backward-stable dense method, eps ||A|| absolute in the eigenvalues.  Eigenvectors of eigenvalues
// closer than 1e-10 ||A|| (rank deficiency, exact multiplicities) are not guaranteed orthogonal by this route: k_trieig
// leaves the verdict in `flag` and the Jacobi solver, launched behind it, runs only then (it returns at once otherwise).
// ---- the same reduction for orders beyond the LDS (139 ... 2048), ONE LAUNCH PER STEP on the whole chip (round 4).  k_tridiag
// walks such a matrix with one workgroup and its working copy in global memory: 29 ms at order 512, which is where an exact Pca of
// data without a spectral gap behind its k components ends up (the subspace iteration cannot converge there and the d x d
// covariance is eigen-decomposed in full).  Launch k applies the PENDING rank-2 update of step k - 1 to the rows below row k and, in
// the same pass over each row, forms that row's entry of p_k = tau_k A22 v_k: one pass over the trailing block per step.  Every
// workgroup first rebuilds, redundantly and bit-identically, what the pass needs -- w_{k-1} = p_{k-1} + K v_{k-1} from the vectors
// the previous launch left in global memory, row k as the pending update leaves it, and from it reflector k -- so nothing but
// a launch boundary orders the steps.  Vectors are indexed by absolute column (zeros outside their range).
constexpr int TMW_NV = 8;                                          // orders up to 256 * TMW_NV
__global__ __launch_bounds__(256) void k_tridiag_mw(double* __restrict__ W, int64_t ld, int L, int k, const double* __restrict__ vp_prev,
                                                    double* __restrict__ vp_cur, double* __restrict__ dd, double* __restrict__ ee,
                                                    double* __restrict__ HV, double* __restrict__ tau) {
    extern __shared__ __attribute__((aligned(16))) double sm_mw[];
    double* svp = sm_mw;            // v_{k-1}
    double* swp = svp + L;          // w_{k-1}
    double* svn = swp + L;          // v_k
    double* s_red = svn + L;        // 2 x 4 partial sums
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int red_slot = 0;
    auto block_sum = [&](double a) {                               // to every thread; alternating slots: one barrier per sum
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        double* slot = s_red + 4 * red_slot;
        red_slot ^= 1;
        if (lane == 0) slot[wv] = a;
        __syncthreads();
        return (slot[0] + slot[1]) + (slot[2] + slot[3]);
    };
    // (a) w_{k-1} = p_{k-1} + K v_{k-1}, K = -tau_{k-1} (p^T v) / 2
    double vq[TMW_NV], pq[TMW_NV];
    double pv = 0;
#pragma unroll
    for (int t = 0; t < TMW_NV; ++t) {
        const int j = tid + 256 * t;
        vq[t] = (k > 0 && j < L) ? vp_prev[j] : 0.0;
        pq[t] = (k > 0 && j < L) ? vp_prev[L + j] : 0.0;
        pv = fma(pq[t], vq[t], pv);
    }
    const double tkp = k > 0 ? tau[k - 1] : 0.0;
    const double Kp = -0.5 * tkp * block_sum(pv);
#pragma unroll
    for (int t = 0; t < TMW_NV; ++t) {
        const int j = tid + 256 * t;
        pq[t] = fma(Kp, vq[t], pq[t]);                            // w_{k-1}
        if (j < L) { svp[j] = vq[t]; swp[j] = pq[t]; }
    }
    __syncthreads();
    // (b) row k as the pending update leaves it (v_{k-1}[k] = 1), and reflector k from it
    const double wk = k > 0 ? swp[k] : 0.0, vk = k > 0 ? svp[k] : 0.0;
    double rq[TMW_NV];
    double sq = 0;
#pragma unroll
    for (int t = 0; t < TMW_NV; ++t) {
        const int j = tid + 256 * t;
        rq[t] = (j >= k && j < L) ? fma(-vk, pq[t], fma(-wk, vq[t], W[(size_t)k * ld + j])) : 0.0;
        sq = fma(j >= k + 2 ? rq[t] : 0.0, j >= k + 2 ? rq[t] : 0.0, sq);
    }
    // (row k's own entries k and k + 1 sit in whichever thread owns those columns: they travel through LDS)
#pragma unroll
    for (int t = 0; t < TMW_NV; ++t) {
        const int j = tid + 256 * t;
        if (j == k) s_red[8] = rq[t];
        if (j == k + 1) s_red[9] = rq[t];
    }
    const double sigma = block_sum(sq);                           // (its barrier also publishes s_red[8 .. 9])
    const double dk = s_red[8], alpha = k + 1 < L ? s_red[9] : 0.0;
    double beta = alpha, tk = 0.0, scale = 0.0;
    if (k + 2 < L && sigma > 0.0) {
        beta = -copysign(sqrt(fma(alpha, alpha, sigma)), alpha);
        tk = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    const bool refl = k + 2 < L;
#pragma unroll
    for (int t = 0; t < TMW_NV; ++t) {
        const int j = tid + 256 * t;
        const double vj = !refl ? 0.0 : (j == k + 1 ? 1.0 : (j >= k + 2 ? rq[t] * scale : 0.0));
        if (j < L) {
            svn[j] = vj;
            if (blockIdx.x == 0) {
                vp_cur[j] = vj;
                if (refl) HV[(size_t)k * L + j] = vj;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) { dd[k] = dk; ee[k] = k + 1 < L ? beta : 0.0; tau[k] = tk; }
    __syncthreads();
    // (c) rows below k: the pending update, and p_k = tau_k A22 v_k from the updated row -- one wave per row
    for (int i = k + 1 + blockIdx.x * 4 + wv; i < L; i += 4 * gridDim.x) {
        double* row = W + (size_t)i * ld;
        const double vi = svp[i], wi = swp[i];
        double acc = 0;
        for (int j = k + 1 + lane; j < L; j += 64) {
            const double nv = fma(-vi, swp[j], fma(-wi, svp[j], row[j]));
            if (k > 0) row[j] = nv;
            acc = fma(nv, svn[j], acc);
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) vp_cur[L + i] = tk * acc;
    }
}

// one wave per eigenpair j (descending): lambda_j, z_j, v_j.  d, e in LDS; q+ / q- / z per wave in LDS.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_trieig(const double* __restrict__ dd, const double* __restrict__ ee,
                                                     const double* __restrict__ HV, const double* __restrict__ tau, int L,
                                                     double* __restrict__ w, double* __restrict__ V, int64_t ldv) {
    extern __shared__ __attribute__((aligned(16))) double sm_te[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double* sd = sm_te;                 // d
    double* se = sd + L;                // e (sub-diagonal), e[L - 1] = 0
    double* s2 = se + L;                // e^2
    double* qp = s2 + L + (size_t)wv * 3 * L;   // per wave: q+, q-, z
    double* qm = qp + L;
    double* z = qm + L;
    for (int i = tid; i < L; i += 64 * WPB) { const double ev = ee[i]; sd[i] = dd[i]; se[i] = ev; s2[i] = ev * ev; }
    __syncthreads();
    const int j = blockIdx.x * WPB + wv;                         // descending index
    if (j >= L) return;
    // Gershgorin interval and the scale of T
    double glo = 1e300, ghi = -1e300;
    for (int i = lane; i < L; i += 64) {
        const double r = (i > 0 ? fabs(se[i - 1]) : 0.0) + (i + 1 < L ? fabs(se[i]) : 0.0);
        glo = fmin(glo, sd[i] - r); ghi = fmax(ghi, sd[i] + r);
    }
    for (int off = 32; off > 0; off >>= 1) { glo = fmin(glo, __shfl_xor(glo, off, 64)); ghi = fmax(ghi, __shfl_xor(ghi, off, 64)); }
    const double tnorm = fmax(fabs(glo), fabs(ghi));
    const double tiny = fmax(tnorm, 1e-300) * 1e-30;            // stands for an exact zero pivot in the recurrences
    const int kth = L - 1 - j;                                    // 0-based ascending index of the wanted eigenvalue
    auto sturm = [&](double x) {                                  // number of eigenvalues of T below x
        double q = sd[0] - x;
        if (fabs(q) < tiny) q = -tiny;
        int cnt = q < 0.0 ? 1 : 0;
        for (int i = 1; i < L; ++i) {
            q = sd[i] - x - s2[i - 1] / q;
            if (fabs(q) < tiny) q = -tiny;
            cnt += q < 0.0 ? 1 : 0;
        }
        return cnt;
    };
    double lo = glo - 1e-12 * tnorm - tiny, hi = ghi + 1e-12 * tnorm + tiny;
    if (!(tnorm < 1e300)) { lo = 0; hi = 0; }                     // non-finite input: leave a NaN-free nonsense, the flag catches it
    for (int pass = 0; pass < 11; ++pass) {                       // 65-way split per pass
        const double h = (hi - lo) * (1.0 / 65.0);
        const double x = lo + h * (lane + 1);
        const int c = sturm(x);
        const unsigned long long mask = __ballot(c > kth);        // lanes whose point has more than kth eigenvalues below it
        if (mask) {
            const int first = __ffsll((long long)mask) - 1;
            const double nhi = lo + h * (first + 1), nlo = lo + h * first;
            hi = nhi; lo = nlo;
        } else {
            lo = lo + h * 64;
        }
    }
    const double lam = 0.5 * (lo + hi);
    if (lane == 0) w[j] = lam;
    // eigenvector of T: twisted factorisation at the index of smallest |gamma| (two independent recurrences: one lane each)
    if (lane == 0) {
        double q = sd[0] - lam;
        if (fabs(q) < tiny) q = tiny;
        qp[0] = q;
        for (int i = 1; i < L; ++i) { q = sd[i] - lam - s2[i - 1] / q; if (fabs(q) < tiny) q = tiny; qp[i] = q; }
    } else if (lane == 1) {
        double q = sd[L - 1] - lam;
        if (fabs(q) < tiny) q = tiny;
        qm[L - 1] = q;
        for (int i = L - 2; i >= 0; --i) { q = sd[i] - lam - s2[i] / q; if (fabs(q) < tiny) q = tiny; qm[i] = q; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double gbest = 1e300;
    int rbest = 0;
    for (int i = lane; i < L; i += 64) {
        const double g = fabs(qp[i] + qm[i] - (sd[i] - lam));
        if (g < gbest) { gbest = g; rbest = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double og = __shfl_xor(gbest, off, 64);
        const int orr = __shfl_xor(rbest, off, 64);
        if (og < gbest || (og == gbest && orr < rbest)) { gbest = og; rbest = orr; }
    }
    if (lane == 0) {
        z[rbest] = 1.0;
        double zi = 1.0;
        for (int i = rbest - 1; i >= 0; --i) { zi = -se[i] * zi / qp[i]; z[i] = zi; }
    } else if (lane == 1) {
        double zi = 1.0;
        for (int i = rbest + 1; i < L; ++i) { zi = -se[i - 1] * zi / qm[i]; z[i] = zi; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // normalise (scaled against overflow), then v = H_0 ... H_{L-3} z: reflector k acts on entries k + 1 ..
    double zmax = 0;
    for (int i = lane; i < L; i += 64) zmax = fmax(zmax, fabs(z[i]));
    for (int off = 32; off > 0; off >>= 1) zmax = fmax(zmax, __shfl_xor(zmax, off, 64));
    const double zs = zmax > 0.0 && zmax < 1e300 ? 1.0 / zmax : 1.0;
    double nrm = 0;
    for (int i = lane; i < L; i += 64) { const double v = z[i] * zs; z[i] = v; nrm += v * v; }
    for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor(nrm, off, 64);
    const double rn = nrm > 0.0 ? 1.0 / sqrt(nrm) : 0.0;
    for (int i = lane; i < L; i += 64) z[i] *= rn;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = L - 3; k >= 0; --k) {
        const double tk = tau[k];
        if (tk == 0.0) continue;
        const double* hv = HV + (size_t)k * L;
        double dot = 0;
        for (int i = k + 1 + lane; i < L; i += 64) dot += hv[i] * z[i];
        for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        const double f = tk * dot;
        for (int i = k + 1 + lane; i < L; i += 64) z[i] -= f * hv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    for (int i = lane; i < L; i += 64) V[(int64_t)i * ldv + j] = z[i];
}
// verdict of the two-stage route, read by the Jacobi kernels launched behind it: flag = 1 (run Jacobi) when some eigenvalue
// is not finite or two neighbours are closer than 1e-10 of the largest magnitude (their vectors need not be orthogonal)
__global__ __launch_bounds__(1024) void k_trieig_verdict(const double* __restrict__ w, const double* __restrict__ ee, int L, double gap_tol,
                                                         const double* __restrict__ A, int64_t lda, int* __restrict__ flag, int ncheck) {
    __shared__ int bad;
    __shared__ double red[3][16];
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const double scale = fmax(fabs(w[0]), fabs(w[L - 1]));
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        const double a = w[j];
        bool b = !(fabs(a) < 1e300);
        if (j + 1 < L) b = b || (j < ncheck && !(a - w[j + 1] > gap_tol * scale)) || !(fabs(ee[j]) > 1e-14 * scale);   // (or the matrix decouples)
        if (b) bad = 1;
    }
    // two invariants of the similarity transform, checked against the INPUT matrix (k_tridiag works on a copy): the trace and
    // the squared Frobenius norm.  A defect anywhere in the reduction or the eigenvalue search shows up here instead of in
    // the caller's results (orders above the workgroup size once went through a truncated Householder step unnoticed).
    double tr = 0, fr = 0, sw = 0, sw2 = 0;
    for (int64_t e = threadIdx.x; e < (int64_t)L * L; e += blockDim.x) {
        const int64_t r = e / L, c = e - r * L;
        const double v = A[r * lda + c];
        fr += v * v;
        if (r == c) tr += v;
    }
    for (int j = threadIdx.x; j < L; j += blockDim.x) { const double a = w[j]; sw += a; sw2 += a * a; }
    double d0 = tr - sw, d1 = fr - sw2, d2 = fr;
    for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); d1 += __shfl_down(d1, off, 64); d2 += __shfl_down(d2, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d0; red[1][threadIdx.x >> 6] = d1; red[2][threadIdx.x >> 6] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int x = 0; x < (int)(blockDim.x >> 6); ++x) { t0 += red[0][x]; t1 += red[1][x]; t2 += red[2][x]; }
        const double fn = sqrt(fmax(t2, 0.0));
        const bool off_inv = !(fabs(t0) <= 1e-9 * fmax(fn, 1e-300) * sqrt((double)L)) || !(fabs(t1) <= 1e-9 * fmax(t2, 1e-300));
        *flag = (bad || off_inv || !(scale > 0.0)) ? 1 : 0;
    }
}

// ---- register-resident variants for L <= 141 (the l = k + 10 of the randomized fits, the nc x nc problems of FastICA) ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, the same bits in every lane: four DPP butterfly steps inside each row of 16 lanes, then the four
// row sums through v_readlane
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ double rcp_nr(double x) {     // 1 / x to fp64 rounding error: v_rcp_f64 and two Newton steps
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rsqrt_nr(double x) {   // 1 / sqrt(x) for normal x > 0
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}
__device__ __forceinline__ void wave_sum2_f64(double& a, double& b) {  // two sums at once: the two chains interleave
    a += dpp_f64<0xB1>(a);  b += dpp_f64<0xB1>(b);
    a += dpp_f64<0x4E>(a);  b += dpp_f64<0x4E>(b);
    a += dpp_f64<0x141>(a); b += dpp_f64<0x141>(b);
    a += dpp_f64<0x140>(a); b += dpp_f64<0x140>(b);
    const double a0 = readlane_d(a, 0) + readlane_d(a, 16), b0 = readlane_d(b, 0) + readlane_d(b, 16);
    const double a1 = readlane_d(a, 32) + readlane_d(a, 48), b1 = readlane_d(b, 32) + readlane_d(b, 48);
    a = a0 + a1;
    b = b0 + b1;
}
__device__ __forceinline__ double oct_sum_f64(double v) {  // sum over each group of 8 consecutive lanes
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    return v;
}

// Householder tridiagonalisation with two barriers per step and no global-memory traffic inside the loop.  The kernel is
// bound by instruction issue and by fp64 latency on ONE compute unit, so it runs one wave per SIMD and keeps the per-wave
// instruction count down.  Thread (g, sub) = (tid / 8, tid % 8): the eight lanes of a group share a row, lane sub owns the
// columns j = sub (mod 8) of the trailing block and keeps v_j (and later w_j) of ITS columns in registers, read from row k
// of the symmetric working copy -- every group computes sigma, beta, tau and p^T v redundantly (bit-identically) from
// those, so nothing but p has to cross a barrier before the rank-2 update.  The step is a template on the number T of
// column slots per lane (trailing order m <= 8 T) and the kernel walks down a ladder of T as the block shrinks; the working
// copy carries PAD zero columns on the right, at least the ladder's overshoot 8 T - m, so no column needs a bound check
// (v and w are exactly zero there and the update leaves the zeros alone).  Reflector k is parked in column k of the working
// copy (dead by then), beta and tau in row k; one pass at the end writes d, e, tau and the reflector rows out.
#ifndef PETAL_TRR_THREADS
#define PETAL_TRR_THREADS 512   // two waves per SIMD: 256 and 1024 threads measured 6 % and 12 % slower at L = 74
#endif
constexpr int TRR_THREADS = PETAL_TRR_THREADS;
template <int T>
__device__ __forceinline__ void tri_step(double* __restrict__ W, int ld, int L, int k, double* __restrict__ sv,
                                         double* __restrict__ sp, long long& _t0) {
    constexpr int NG = TRR_THREADS / 8;                          // row groups
    constexpr int RB = (8 * T + NG - 1) / NG;                    // rows per group: the whole trailing block in one unrolled pass
    const int tid = threadIdx.x, g = tid >> 3, sub = tid & 7;
    const int m = L - k - 1;
    double* rowk = W + k * ld + k + 1;                           // x_j = W[k][k + 1 + j] (= W[k + 1 + j][k]); zeros from j = m on
    double xv[T], wj[T];
    double sq0 = 0, sq1 = 0;
    const double alpha = rowk[0];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        xv[t] = rowk[sub + 8 * t];
        if (t == 0) sq0 = sub > 0 ? xv[0] * xv[0] : 0.0;
        else if (t & 1) sq1 = fma(xv[t], xv[t], sq1);
        else sq0 = fma(xv[t], xv[t], sq0);
    }
    const double sigma = oct_sum_f64(sq0 + sq1);
    double beta = alpha, tk = 0.0, scale = 0.0;
    if (sigma > 0.0) {                                            // v_rsq / v_rcp + Newton, the reciprocal seeded from the raw rsq
        const double n2 = fma(alpha, alpha, sigma);
        double rs = __builtin_amdgcn_rsq(n2);
        double rc = __builtin_amdgcn_rcp(fma(n2, rs, fabs(alpha)));   // ~ 1 / (|alpha| + ||x||), refined below
        rs = rs * fma(-0.5 * n2 * rs, rs, 1.5);
        rs = rs * fma(-0.5 * n2 * rs, rs, 1.5);
        const double nrm = n2 * rs;
        beta = -copysign(nrm, alpha);
        const double den = fabs(alpha) + nrm;                     // |alpha - beta|
        rc = fma(fma(-den, rc, 1.0), rc, rc);
        rc = fma(fma(-den, rc, 1.0), rc, rc);
        scale = copysign(rc, alpha);
        tk = den * rs;                                            // (beta - alpha) / beta
    }
#pragma unroll
    for (int t = 0; t < T; ++t) xv[t] *= scale;                   // v_j (exactly 0 from j = m on)
    if (sub == 0) xv[0] = 1.0;
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = sub + 8 * t;
            sv[j] = xv[t];
            if (j < m) W[(k + 1 + j) * ld + k] = xv[t];
        }
    }
    DBG_T(0);
    if (tk != 0.0) {                                              // (uniform: every thread holds the same bits)
        double acc[RB][2];                                        // p = tau A22 v: the rows of a group side by side (ILP)
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = g + NG * r;
            const double* row = W + (k + 1 + min(i, m - 1)) * ld + k + 1 + sub;
            acc[r][0] = 0; acc[r][1] = 0;
#pragma unroll
            for (int t = 0; t < T; ++t) acc[r][t & 1] = fma(row[8 * t], xv[t], acc[r][t & 1]);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = g + NG * r;
            const double a = oct_sum_f64(acc[r][0] + acc[r][1]);
            if (sub == 0 && i < 8 * T) sp[i] = i < m ? tk * a : 0.0;  // (p is exactly 0 from i = m on, like v)
        }
    }
    DBG_T(1);
    __syncthreads();                                              // every read of row k is done: it now keeps beta and tau
    DBG_T(2);
    if (tid == 0) { rowk[0] = beta; rowk[1] = tk; }
    if (tk != 0.0) {
        double pv0 = 0, pv1 = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            wj[t] = sp[sub + 8 * t];
            if (t & 1) pv1 = fma(wj[t], xv[t], pv1);
            else pv0 = fma(wj[t], xv[t], pv0);
        }
        const double K = -0.5 * tk * oct_sum_f64(pv0 + pv1);
#pragma unroll
        for (int t = 0; t < T; ++t) wj[t] = fma(K, xv[t], wj[t]);  // w = p + K v
        DBG_T(3);
#pragma unroll
        for (int r = 0; r < RB; ++r) {                            // A22 -= v w^T + w v^T
            const int i = g + NG * r;
            if (i < m) {
                const double vi = sv[i], wi = fma(K, vi, sp[i]);
                double* row = W + (k + 1 + i) * ld + k + 1 + sub;
#pragma unroll
                for (int t = 0; t < T; ++t) row[8 * t] = fma(-vi, wj[t], fma(-wi, xv[t], row[8 * t]));
            }
        }
    }
    DBG_T(4);
    __syncthreads();
    DBG_T(5);
}
// the steps whose trailing order m = L - k - 1 needs T slots (m > 8 (T - STEP)), then down the ladder
template <int T, int STEP>
__device__ __forceinline__ void tri_ladder(double* __restrict__ W, int ld, int L, int& k, double* __restrict__ sv,
                                           double* __restrict__ sp, long long& _t0) {
    for (const int kend = min(L - 2, L - 1 - 8 * (T - STEP)); k < kend; ++k) tri_step<T>(W, ld, L, k, sv, sp, _t0);
    if constexpr (T - STEP > 0) tri_ladder<T - STEP, STEP>(W, ld, L, k, sv, sp, _t0);
}
// row pitch of the working copy: at least L + PAD, and 8 or 24 (mod 32) doubles where the LDS has room -- the four rows that a
// 32-lane half reads (8 lanes each, 8 consecutive doubles) then fall on four different quarters of the bank row (ds_read_b64
// banks: (a / 4) mod 64); an odd pitch, the round-3 choice, leaves two of them overlapping: 85.4 -> 80.5 us at l = 74
__host__ __device__ constexpr int tri_ld(int L, int step) {
    const int lo = L + 8 * step, r = lo % 32;
    const int v = r <= 8 ? lo + (8 - r) : (r <= 24 ? lo + (24 - r) : lo + (40 - r));
    return (size_t)8 * ((size_t)L * v + 288) <= 160 * 1024 ? v : (step == 1 ? L + 8 : (lo | 1));
}
__host__ __device__ constexpr int tri_sp_len(int tmax) { return 8 * tmax; }
template <int TMAX, int STEP>  // L <= 8 TMAX; PAD = 8 STEP zero columns
__global__ __launch_bounds__(TRR_THREADS) void k_tridiag_r(const double* __restrict__ A, int L, int64_t lda, double* __restrict__ dd,
                                                           double* __restrict__ ee, double* __restrict__ HV,
                                                           double* __restrict__ tau, double* __restrict__ gg,
                                                           int* __restrict__ flag, double* __restrict__ V, int64_t ldv, int Lz, int reset_flag) {
    extern __shared__ __attribute__((aligned(16))) double sm_tri[];
    const int tid = threadIdx.x;
    const int ld = tri_ld(L, STEP);
    for (int e = tid; e < Lz * Lz; e += TRR_THREADS) {           // the caller's zero padding of V (rows / columns L .. Lz - 1)
        const int r = e / Lz, c = e - r * Lz;
        if (r >= L || c >= L) V[(int64_t)r * ldv + c] = 0.0;
    }
    double* W = sm_tri;
    double* sv = W + (size_t)L * ld;                              // 8 TMAX entries
    double* sp = sv + 8 * TMAX;                                   // tri_sp_len(TMAX) entries
    for (int e0 = tid; e0 < L * ld; e0 += 8 * TRR_THREADS) {     // (eight loads in flight per thread: a load -> LDS-store loop pays
        double t8[8];                                            //  an L2 round trip per trip, thirteen in a row at L = 74)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * TRR_THREADS, r = e / ld, c = e - r * ld;
            t8[u] = (e < L * ld && c < L) ? A[(int64_t)r * lda + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * TRR_THREADS; if (e < L * ld) W[e] = t8[u]; }
    }
    if (tid == 0 && reset_flag) *flag = 0;   // (a caller-owned verdict word accumulates: it is not reset here)
    __syncthreads();
    long long _t0 = 0;
#ifdef PETAL_DEBUG_COUNTERS
    _t0 = clock64();
    const long long _w0 = wall_clock64();
#endif
    int k = 0;
    tri_ladder<TMAX, STEP>(W, ld, L, k, sv, sp, _t0);
#ifdef PETAL_DEBUG_COUNTERS
    if (tid == 0) g_cyc[25] += wall_clock64() - _w0;
#endif
    for (int e = tid; e < (L - 2) * L; e += TRR_THREADS) {       // reflector k: row k of HV, entries k + 1 .. L - 1
        const int kk = e / L, c = e - kk * L;
        HV[e] = c > kk ? W[c * ld + kk] : 0.0;
    }
    for (int kk = tid; kk < L; kk += TRR_THREADS) {
        dd[kk] = W[kk * ld + kk];
        ee[kk] = kk + 2 < L ? W[kk * ld + kk + 1] : (kk + 2 == L ? W[(L - 1) * ld + L - 2] : 0.0);
        tau[kk] = kk + 2 < L ? W[kk * ld + kk + 2] : 0.0;
    }
    // gg[k] = v_k . v_{k-1} (lets the back-transformation apply two reflectors per reduction round), eight lanes per k
    for (int kk = tid >> 3; kk < L; kk += TRR_THREADS / 8) {
        double acc = 0;
        if (kk >= 1 && kk + 2 < L)
            for (int c = kk + 1 + (tid & 7); c < L; c += 8) acc += W[c * ld + kk] * W[c * ld + kk - 1];
        acc = oct_sum_f64(acc);
        if ((tid & 7) == 0) gg[kk] = acc;
    }
}

// ---- the same reduction on 16-byte LDS accesses (round 4).  With two waves per SIMD the LDS hands out 8-byte reads at a fraction
// of its rate (MI355X_MICROARCH.md, LDS: ds_read_b64 needs ~4 waves per SIMD, ds_read_b128 one), and a step of k_tridiag_r is paced
// by the number of LDS instructions between its two barriers (measured with the look-ahead experiment, EXPERIMENTS.md round 4).
// Here lane `sub` owns the ABSOLUTE column pairs {2 sub, 2 sub + 1} (mod 16): its slots mean the same columns in every step and
// start on 16-byte boundaries, so every row access is one ds_read_b128 / ds_write_b128 per 16 columns -- half the LDS
// instructions of the 8-byte form.  Columns that have left the trailing block keep v = w = 0: the update writes their old
// values back unchanged, so nothing needs a guard; reflector k is parked in ROW k (dead once the step has read it), contiguous,
// with beta in the slot of its unit entry and tau in the dead column k below it.
template <int TM, int T>
__device__ __forceinline__ void triw_step(double* __restrict__ W, int ld, int L, int off, int k, double* __restrict__ sv,
                                          double* __restrict__ sp) {
    constexpr int NG = TRR_THREADS / 8;
    constexpr int RB = (16 * T + NG - 1) / NG;
    constexpr int A0 = TM - T;
    const int tid = threadIdx.x, g = tid >> 3, sub = tid & 7;
    const int c0 = 2 * sub - off;                                 // column of element 0 of slot a: c0 + 16 a
    // ---- reflector k from row k ----
    const double* rowk = W + k * ld;
    const double alpha = rowk[k + 1];
    f64x2 x[TM];
    double sq0 = 0, sq1 = 0;
#pragma unroll
    for (int a = A0; a < TM; ++a) {
        const int c = c0 + 16 * a;
        x[a] = *reinterpret_cast<const f64x2*>(rowk + c);
        if (a == A0) {                                            // (only the rung's leading slot holds columns that have left the block)
            x[a].x = c > k + 1 ? x[a].x : 0.0;
            x[a].y = c + 1 > k + 1 ? x[a].y : 0.0;
        }
        sq0 = fma(x[a].x, x[a].x, sq0);
        sq1 = fma(x[a].y, x[a].y, sq1);
    }
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
#pragma unroll
    for (int a = A0; a < TM; ++a) {
        const int c = c0 + 16 * a;
        x[a].x = (a == A0 && c == k + 1) ? 1.0 : x[a].x * scale;
        x[a].y = (a == A0 && c + 1 == k + 1) ? 1.0 : x[a].y * scale;
    }
    if (g == 0) {
#pragma unroll
        for (int a = A0; a < TM; ++a) *reinterpret_cast<f64x2*>(sv + 2 * sub + 16 * a) = x[a];
        // tau is parked in the dead column k of the row below -- BEFORE the barrier: the update behind it writes the old values
        // of dead columns back, so it has to find tau there (the product in front of it multiplies it by v = 0 either way)
        if (sub == 0) W[(k + 1) * ld + k] = tk;
    }
    if (tk != 0.0) {                                              // (uniform)
        // ---- p = tau A22 v ----
        double acc[RB][2];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = k + 1 + g + NG * r;
            const double* row = W + min(i, L - 1) * ld + c0;
            acc[r][0] = 0; acc[r][1] = 0;
#pragma unroll
            for (int a = A0; a < TM; ++a) {
                const f64x2 rv = *reinterpret_cast<const f64x2*>(row + 16 * a);
                acc[r][0] = fma(rv.x, x[a].x, acc[r][0]);
                acc[r][1] = fma(rv.y, x[a].y, acc[r][1]);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = k + 1 + g + NG * r;
            const double s = oct_sum_f64(acc[r][0] + acc[r][1]);
            if (sub == 0 && i < L) sp[i + off] = tk * s;
        }
    }
    __syncthreads();
    // reflector k is parked now that every group has read row k (no update touches row k): v, with beta where its unit entry sits
    // (by a group of the second wave: the first has the most rows to update)
    if (g == 8) {
        double* rk = W + k * ld;
#pragma unroll
        for (int a = A0; a < TM; ++a) {
            const int c = c0 + 16 * a;
            if (c + 1 > k) {
                f64x2 pv = x[a];
                if (c == k + 1) pv.x = beta;
                if (c + 1 == k + 1) pv.y = beta;
                if (c == k) pv.x = rk[k];                          // (the diagonal entry stays)
                *reinterpret_cast<f64x2*>(rk + c) = pv;
            }
        }
    }
    if (tk != 0.0) {
        f64x2 w[TM];
        double pv0 = 0, pv1 = 0;
#pragma unroll
        for (int a = A0; a < TM; ++a) {
            const int c = c0 + 16 * a;
            w[a] = *reinterpret_cast<const f64x2*>(sp + 2 * sub + 16 * a);
            if (a == A0) {                                        // (entries of p from rows that have left the block are stale)
                w[a].x = c > k ? w[a].x : 0.0;
                w[a].y = c + 1 > k ? w[a].y : 0.0;
            }
            pv0 = fma(w[a].x, x[a].x, pv0);
            pv1 = fma(w[a].y, x[a].y, pv1);
        }
        const double K = -0.5 * tk * oct_sum_f64(pv0 + pv1);
#pragma unroll
        for (int a = A0; a < TM; ++a) { w[a].x = fma(K, x[a].x, w[a].x); w[a].y = fma(K, x[a].y, w[a].y); }
#pragma unroll
        for (int r = 0; r < RB; ++r) {                            // A22 -= v w^T + w v^T
            const int i = k + 1 + g + NG * r;
            if (i < L) {
                const double vi = sv[i + off], wi = fma(K, vi, sp[i + off]);
                double* row = W + i * ld + c0;
#pragma unroll
                for (int a = A0; a < TM; ++a) {
                    f64x2 rv = *reinterpret_cast<const f64x2*>(row + 16 * a);
                    rv.x = fma(-vi, w[a].x, fma(-wi, x[a].x, rv.x));
                    rv.y = fma(-vi, w[a].y, fma(-wi, x[a].y, rv.y));
                    *reinterpret_cast<f64x2*>(row + 16 * a) = rv;
                }
            }
        }
    }
    __syncthreads();
}
template <int TM, int T>
__device__ __forceinline__ void triw_ladder(double* __restrict__ W, int ld, int L, int S2, int off, int& k, double* __restrict__ sv,
                                            double* __restrict__ sp) {
    if (S2 >= T) {                                                // (a rung with more slots than the matrix has is skipped)
        const int kend = T > 1 ? min(L - 2, 16 * (S2 - T + 1) - 1) : L - 2;
        for (; k < kend; ++k) triw_step<TM, T>(W, ld, L, off, k, sv, sp);
    }
    if constexpr (T > 1) triw_ladder<TM, T - 1>(W, ld, L, S2, off, k, sv, sp);
}
// row pitch: every 16-column slot addressable; 16 (mod 32) doubles where the LDS has room -- the four 8-double pieces of a
// ds_read_b128 lane group (two rows x two half slots) then fall on four different quarters of the bank row
__host__ __device__ inline int triw_ld(int L, int tm) {
    const int s16 = 16 * ((L + 15) / 16);
    const int want = (s16 % 32 == 16) ? s16 : s16 + 16;
    return sizeof(double) * ((size_t)L * want + 32 * tm) <= 160 * 1024 ? want : s16;
}
template <int TM>  // L <= 16 TM
__global__ __launch_bounds__(TRR_THREADS) void k_tridiag_w(const double* __restrict__ A, int L, int64_t lda, double* __restrict__ dd,
                                                           double* __restrict__ ee, double* __restrict__ HV,
                                                           double* __restrict__ tau, double* __restrict__ gg,
                                                           int* __restrict__ flag, double* __restrict__ V, int64_t ldv, int Lz, int reset_flag) {
    extern __shared__ __attribute__((aligned(16))) double sm_tri[];
    const int tid = threadIdx.x;
    const int S2 = (L + 15) >> 4, off = 16 * (TM - S2);
    const int ld = triw_ld(L, TM);
    for (int e = tid; e < Lz * Lz; e += TRR_THREADS) {           // the caller's zero padding of V (rows / columns L .. Lz - 1)
        const int r = e / Lz, c = e - r * Lz;
        if (r >= L || c >= L) V[(int64_t)r * ldv + c] = 0.0;
    }
    double* W = sm_tri;
    double* sv = W + (size_t)L * ld;                              // 16 TM entries, indexed by column + off
    double* sp = sv + 16 * TM;                                    // 16 TM entries
    for (int e0 = tid; e0 < L * ld; e0 += 8 * TRR_THREADS) {
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * TRR_THREADS, r = e / ld, c = e - r * ld;
            t8[u] = (e < L * ld && c < L) ? A[(int64_t)r * lda + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * TRR_THREADS; if (e < L * ld) W[e] = t8[u]; }
    }
    for (int e = tid; e < 32 * TM; e += TRR_THREADS) sv[e] = 0.0;
    if (tid == 0 && reset_flag) *flag = 0;   // (a caller-owned verdict word accumulates: it is not reset here)
    __syncthreads();
    int k = 0;
    triw_ladder<TM, TM>(W, ld, L, S2, off, k, sv, sp);
    for (int e = tid; e < (L - 2) * L; e += TRR_THREADS) {       // reflector k: row k of HV, entries k + 1 .. L - 1
        const int kk = e / L, c = e - kk * L;
        HV[e] = c > kk + 1 ? W[kk * ld + c] : (c == kk + 1 ? 1.0 : 0.0);
    }
    for (int kk = tid; kk < L; kk += TRR_THREADS) {
        dd[kk] = W[kk * ld + kk];
        ee[kk] = kk + 2 < L ? W[kk * ld + kk + 1] : (kk + 2 == L ? W[(L - 1) * ld + L - 2] : 0.0);
        tau[kk] = kk + 2 < L ? W[(kk + 1) * ld + kk] : 0.0;
    }
    // gg[k] = v_k . v_{k-1} (lets the back-transformation apply two reflectors per reduction round), eight lanes per k: v_k is row k
    // from column k + 2 on and 1 at column k + 1
    for (int kk = tid >> 3; kk < L; kk += TRR_THREADS / 8) {
        double acc = 0;
        if (kk >= 1 && kk + 2 < L)
            for (int c = kk + 1 + (tid & 7); c < L; c += 8) acc += (c == kk + 1 ? 1.0 : W[kk * ld + c]) * W[(kk - 1) * ld + c];
        acc = oct_sum_f64(acc);
        if ((tid & 7) == 0) gg[kk] = acc;
    }
}

// one wave per eigenpair.  The Sturm count runs on the three-term recurrence of the leading minors, renormalised by their
// exponent every step (v_frexp_mant / v_ldexp: no division on the chain); d and e^2 are fetched eight at a time so the LDS
// latency is paid once per eight steps.  The twisted factorisation keeps both recurrences in one instruction stream
// (lanes 0 and 1) and stores the multipliers, so the two sweeps of z are products only; z then lives in registers and the
// reflector rows come from an LDS stage filled while the eigenvalue search runs.  An eigenvalue with a neighbour inside
// gap_tol ||T|| (its vector is then only eps / gap_tol accurate) or a non-finite result raises *flag for the Jacobi solver
// launched behind this kernel.
template <int WPB, int QMAX>  // L <= 64 QMAX
__global__ __launch_bounds__(64 * WPB) void k_trieig_r(const double* __restrict__ dd, const double* __restrict__ ee,
                                                       const double* __restrict__ HV, const double* __restrict__ tau,
                                                       const double* __restrict__ gg, int L, double gap_tol, int hv_rows,
                                                       double* __restrict__ w, double* __restrict__ V, int64_t ldv,
                                                       int* __restrict__ flag, int ncheck) {
    extern __shared__ __attribute__((aligned(16))) double sm_te[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef PETAL_DEBUG_COUNTERS
    long long _t0 = clock64();
    const long long _w0 = wall_clock64();
#define DBG_E(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { long long _t = clock64(); g_cyc[i] += _t - _t0; _t0 = _t; } } while (0)
#else
#define DBG_E(i) do {} while (0)
#endif
    const int LP = ((L + 7) & ~7) + 8;          // d, e, e^2 padded: whole batches of eight may run past L without a bound check
    double* sd = sm_te;
    double* se = sd + LP;
    double* s2 = se + LP;
    double* st = s2 + LP;                       // tau
    double* sdr = st + LP;                      // d and e^2 in reversed order (the recurrence from the bottom)
    double* s2r = sdr + LP;
    double* sg = s2r + LP;                      // v_k . v_{k-1}
    double* qp = sg + LP + (size_t)wv * 4 * LP; // per wave: minors from the top / bottom and their predecessors; later l+, u-, z
    double* qm = qp + LP;
    double* lp = qm + LP;
    double* lm = lp + LP;
    double* shv = sg + LP + (size_t)WPB * 4 * LP;   // hv_rows reflector rows, staged for the back-transformation
    // the last hv_rows reflectors (the first ones applied) start their way into LDS now
    int kbase = max(L - 2 - hv_rows, 0);
    auto stage = [&](int k0, int nrows) {                          // eight loads per thread in flight: the rows come from HBM / MALL
        const int ne = nrows * L;
        const double* src = HV + (size_t)k0 * L;
        for (int e0 = tid; e0 < ne; e0 += 64 * WPB * 8) {
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * 64 * WPB; t8[u] = e < ne ? src[e] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * 64 * WPB; if (e < ne) shv[e] = t8[u]; }
        }
    };
    stage(kbase, L - 2 - kbase);
    // Gershgorin interval and the scale of T (every wave, from global memory: L values)
    double glo = 1e300, ghi = -1e300;
    for (int i = lane; i < L; i += 64) {
        const double r = (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i + 1 < L ? fabs(ee[i]) : 0.0);
        glo = fmin(glo, dd[i] - r); ghi = fmax(ghi, dd[i] + r);
    }
    for (int off = 32; off > 0; off >>= 1) { glo = fmin(glo, __shfl_xor(glo, off, 64)); ghi = fmax(ghi, __shfl_xor(ghi, off, 64)); }
    const double tnorm = fmax(fabs(glo), fabs(ghi));
    const bool sane = tnorm > 1e-140 && tnorm < 1e140;
    const double inv = sane ? 1.0 / tnorm : 0.0;
    for (int i = tid; i < LP; i += 64 * WPB) {
        const double ev = i < L ? ee[i] * inv : 0.0;
        sd[i] = i < L ? dd[i] * inv : 4.0;                        // padding: d - x > 0 and e = 0 there, no sign change is added
        se[i] = ev;
        s2[i] = fmax(ev * ev, 1e-280);                            // never an exact split: the minors cannot stick at zero
        // a (numerically) decoupled matrix: the twisted factorisation's choice of r compares pivots that the 1e-30 clamp has
        // flattened (an exactly diagonal A with lambda hit to the last bit picked a neighbour's unit vector) -- Jacobi's case
        if (i + 1 < L && !(fabs(ev) > 1e-14)) atomicOr(flag, 1);
        st[i] = i < L ? tau[i] : 0.0;
        sg[i] = i < L ? gg[i] : 0.0;
        const int ir = L - 1 - i;                                 // reversed: sdr[t] = d_{L-1-t}, s2r[u] = e^2_{L-2-u}
        sdr[i] = ir >= 0 ? dd[ir] * inv : 4.0;
        const double evr = ir >= 1 ? ee[ir - 1] * inv : 0.0;
        s2r[i] = fmax(evr * evr, 1e-280);
    }
    __syncthreads();
    DBG_E(16);
    const int j = min(blockIdx.x * WPB + wv, L - 1);             // descending index (a surplus wave repeats the last one)
    const int kth = L - 1 - j;
    // (p0, p1) = two consecutive leading minors of T - x on a common power-of-two scale.  The scale follows the exponent of
    // the OLDER of the two (integer instructions on the exponent field, off the dependent chain), so the chain per step is
    // one fma and one multiply; the magnitudes stay within a few decades of 1 because a step grows a minor at most 3 x.
    auto step_scale = [](double p1) {
        const int ef = max((__double2hiint(p1) >> 20) & 0x7ff, 423);
        return __hiloint2double((2045 - ef) << 20, 0);            // 2^(1022 - ef), at most 2^599 (exact zero, denormals)
    };
    auto sturm = [&](double x) {                                  // number of eigenvalues of T / ||T|| below x
        double p0 = 1.0, p1 = sd[0] - x;
        unsigned sprev = (unsigned)__double2hiint(p1) >> 31, cnt = sprev;
        for (int i0 = 1; i0 < L; i0 += 8) {
            double dv[8], e2v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dv[u] = sd[i0 + u]; e2v[u] = s2[i0 + u - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {                         // (runs into the padding: no bound check per step)
                const double raw = fma(dv[u] - x, p1, -e2v[u] * p0);
                const unsigned sg = (unsigned)__double2hiint(raw) >> 31;
                cnt += sg ^ sprev;                                // sign change (a zero minor: either sign)
                sprev = sg;
                if (u & 1) {                                      // renormalise every second step: two steps move a minor by
                    const double sc = step_scale(p1);             // at most 1e32 either way
                    p0 = p1 * sc;
                    p1 = raw * sc;
                } else {
                    p0 = p1;
                    p1 = raw;
                }
            }
        }
        return (int)cnt;
    };
    double lo = glo * inv - 1e-12, hi = ghi * inv + 1e-12;
    for (int pass = 0; pass < 9; ++pass) {                        // 65-way split per pass: (2 + 2e-12) / 65^9 < 1e-16 of ||T||
        const double h = (hi - lo) * (1.0 / 65.0);
        const int c = sturm(lo + h * (lane + 1));
        const unsigned long long mask = __ballot(c > kth);
        if (mask) {
            const int first = __ffsll((long long)mask) - 1;
            const double nhi = lo + h * (first + 1), nlo = lo + h * first;
            hi = nhi; lo = nlo;
        } else {
            lo = lo + h * 64;
        }
    }
    const double lam = 0.5 * (lo + hi);
    DBG_E(17);
    if (lane == 0) w[j] = lam * tnorm;
    // neighbours inside gap_tol? (counts at lam -+ gap_tol: exactly this one eigenvalue should lie between)
    {
        const int c = sturm((lane & 1) ? lam + gap_tol : lam - gap_tol);
        const int c0 = __builtin_amdgcn_readlane(c, 0), c1 = __builtin_amdgcn_readlane(c, 1);
        // (ncheck: the caller only uses the leading ncheck pairs -- a cluster further down keeps its eps / gap vectors)
        if (lane == 0 && ((c1 - c0 != 1 && j < ncheck) || !sane)) atomicOr(flag, 1);
    }
    DBG_E(18);
    // twisted factorisation.  The pivots are ratios of consecutive minors: q+_i = p_{i+1} / p_i from the top (lane 0), q-_i
    // likewise from the bottom (lane 1), both on the division-free recurrence of the Sturm count in ONE instruction stream:
    // lane 1 walks the REVERSED copies of d and e^2 and leaves its results in reversed order, so both lanes address
    // base + step.  The divisions, gamma and the multipliers are then lane-parallel.
    const bool fw = lane == 0;
    if (lane < 2) {
        const double* bd = fw ? sd : sdr;
        const double* be = fw ? s2 : s2r;
        double* narr = fw ? qp : qm;
        double* darr = fw ? lp : lm;
        double p0 = 1.0, p1 = bd[0] - lam;
        narr[0] = p1;
        darr[0] = 1.0;
        for (int t0 = 1; t0 < L; t0 += 8) {                       // (steps past L - 1 read and write the padding)
            double dv[8], e2v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dv[u] = bd[t0 + u]; e2v[u] = be[t0 + u - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double raw = fma(dv[u] - lam, p1, -e2v[u] * p0);
                narr[t0 + u] = raw;
                darr[t0 + u] = p1;
                if (u & 1) {
                    const double sc = step_scale(p1);
                    p0 = p1 * sc;
                    p1 = raw * sc;
                } else {
                    p0 = p1;
                    p1 = raw;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    DBG_E(19);
    double gbest = 1e300;
    int rbest = 0;
    {
        double qpv[QMAX], qmv[QMAX];
        auto pivot = [](double num, double den) {                 // a vanishing or non-finite pivot is replaced, as dstein does
            double q = num / den;
            if (!(fabs(q) >= 1e-30)) q = 1e-30;
            if (!(fabs(q) <= 1e300)) q = copysign(1e300, num) * (den < 0.0 ? -1.0 : 1.0);
            return q;
        };
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            const int i = lane + 64 * q;
            if (i < L) {
                qpv[q] = pivot(qp[i], lp[i]);
                qmv[q] = pivot(qm[L - 1 - i], lm[L - 1 - i]);
                const double gm = fabs(qpv[q] + qmv[q] - (sd[i] - lam));
                if (gm < gbest) { gbest = gm; rbest = i; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // multipliers, laid out for the two sweeps away from r: lp[L - 2 - i] = l+_i = e_i / q+_i (lane 0 walks i downwards,
        // the array upwards), lm[i] = u-_{i-1} = e_{i-1} / q-_i
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            const int i = lane + 64 * q;
            if (i < L) {
                if (i + 1 < L) lp[L - 2 - i] = se[i] * rcp_nr(qpv[q]);
                if (i > 0) lm[i] = se[i - 1] * rcp_nr(qmv[q]);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double og = __shfl_xor(gbest, off, 64);
        const int orr = __shfl_xor(rbest, off, 64);
        if (og < gbest || (og == gbest && orr < rbest)) { gbest = og; rbest = orr; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    DBG_E(20);
    // z_r = 1, z_i = -l+_i z_{i+1} below r (lane 0: kept reversed in qm), z_i = -u-_{i-1} z_{i-1} above (lane 1: in qp)
    if (lane < 2) {
        const double* bm = fw ? lp + (L - 2 - rbest) : lm + rbest;   // the multiplier of step t sits at bm[t]
        double* bz = fw ? qm + (L - 1 - rbest) : qp + rbest;         // z of step t goes to bz[t]
        const int tcap = (fw ? rbest : L - 1 - rbest) + 8;           // steps past a lane's own end land in the padding
        const int nmax = max(rbest, L - 1 - rbest);
        double zi = 1.0;
        bz[0] = 1.0;
        for (int t0 = 1; t0 <= nmax; t0 += 8) {
            double mv[8];
            int tt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { tt[u] = min(t0 + u, tcap); mv[u] = bm[tt[u]]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { zi = -mv[u] * zi; bz[tt[u]] = zi; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    DBG_E(21);
    double zq[QMAX];
    double zmax = 0;
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int i = lane + 64 * q;
        zq[q] = i < L ? (i < rbest ? qm[L - 1 - i] : qp[i]) : 0.0;
        zmax = fmax(zmax, fabs(zq[q]));
    }
    for (int off = 32; off > 0; off >>= 1) zmax = fmax(zmax, __shfl_xor(zmax, off, 64));
    const double zs = zmax > 0.0 && zmax < 1e300 ? 1.0 / zmax : 1.0;
    double nrm = 0;
#pragma unroll
    for (int q = 0; q < QMAX; ++q) { zq[q] *= zs; nrm += zq[q] * zq[q]; }
    nrm = wave_sum_f64(nrm);
    const double rn = nrm > 0.0 ? 1.0 / sqrt(nrm) : 0.0;
    if (lane == 0 && !(nrm > 0.0 && nrm < 1e300)) atomicOr(flag, 1);
#pragma unroll
    for (int q = 0; q < QMAX; ++q) zq[q] *= rn;
    DBG_E(22);
    // v = H_0 ... H_{L-3} z from the staged rows, hv_rows at a time (one stage when they all fit)
    for (int ktop = L - 3; ktop >= 0;) {
        for (int k = ktop; k >= kbase; k -= 4) {
            double hv[4][QMAX], tk[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int kk = max(k - b, kbase);
                tk[b] = k - b >= kbase ? st[kk] : 0.0;
#pragma unroll
                for (int q = 0; q < QMAX; ++q) {
                    const int i = lane + 64 * q;
                    hv[b][q] = i < L ? shv[(kk - kbase) * L + i] : 0.0;
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b += 2) {                      // reflectors k - b, then k - b - 1: one reduction round for both
                double da = 0, db = 0;
#pragma unroll
                for (int q = 0; q < QMAX; ++q) { da += hv[b][q] * zq[q]; db += hv[b + 1][q] * zq[q]; }
                wave_sum2_f64(da, db);
                const double fa = tk[b] * da;
                const double fb = tk[b + 1] * (db - fa * sg[max(k - b, 0)]);   // v_b . (z - fa v_a) = db - fa (v_a . v_b)
#pragma unroll
                for (int q = 0; q < QMAX; ++q) zq[q] -= fa * hv[b][q] + fb * hv[b + 1][q];
            }
        }
        ktop = kbase - 1;
        if (ktop < 0) break;
        __syncthreads();                                          // (uniform: every wave of the block walks the same stages)
        kbase = max(ktop + 1 - hv_rows, 0);
        stage(kbase, ktop + 1 - kbase);
        __syncthreads();
    }
    DBG_E(23);
#ifdef PETAL_DEBUG_COUNTERS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_cyc[24] += wall_clock64() - _w0;
#endif
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int i = lane + 64 * q;
        if (i < L) V[(int64_t)i * ldv + j] = zq[q];
    }
}

// replay of the rotation log on V = I: wave <-> row r of V (kept in LDS), lanes <-> the disjoint pairs of a round.
// The angles of JR_BATCH rounds are fetched ahead (independent loads) so the global-memory latency is paid once per batch.
constexpr int JR_WAVES = 4, JR_BATCH = 8;
template <int HP>  // pairs per lane: half <= 64 HP
__global__ __launch_bounds__(64 * JR_WAVES) void k_apply_rot(const jf64x2* __restrict__ log_cs, const int* __restrict__ nrounds,
                                                             const int* __restrict__ rank, int L, double* __restrict__ V, int64_t ldv,
                                                             const int* __restrict__ flag) {
    if (flag && *flag == 0) return;
    extern __shared__ __attribute__((aligned(16))) double sm_jr[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Le = (L + 1) & ~1, half = Le / 2, rounds = Le - 1;
    const int r = blockIdx.x * JR_WAVES + wv;
    double* row = sm_jr + (size_t)wv * Le;
    for (int j = lane; j < Le; j += 64) row[j] = (j == r) ? 1.0 : 0.0;
    const int nR = *nrounds;
    PairIt it[HP];
#pragma unroll
    for (int h = 0; h < HP; ++h) it[h].init(min(lane + 64 * h, half - 1), 0, Le);
    for (int R0 = 0; R0 < nR; R0 += JR_BATCH) {
        jf64x2 cs[JR_BATCH][HP];
#pragma unroll
        for (int u = 0; u < JR_BATCH; ++u) {
            const int Ru = min(R0 + u, nR - 1);
#pragma unroll
            for (int h = 0; h < HP; ++h) cs[u][h] = log_cs[(size_t)Ru * half + min(lane + 64 * h, half - 1)];
        }
#pragma unroll
        for (int u = 0; u < JR_BATCH; ++u) {
            if (R0 + u < nR) {
                double vp[HP], vq[HP];
                int p[HP], q[HP];
#pragma unroll
                for (int h = 0; h < HP; ++h) {
                    it[h].get(min(lane + 64 * h, half - 1), Le, L, p[h], q[h]);
                    it[h].next(Le);  // the sweeps restart at round 0 after Le - 1 rounds: the raw players wrap with them
                    vp[h] = row[p[h]]; vq[h] = row[q[h]];
                }
#pragma unroll
                for (int h = 0; h < HP; ++h) {
                    if (lane + 64 * h < half && p[h] != q[h]) {
                        row[p[h]] = cs[u][h][0] * vp[h] - cs[u][h][1] * vq[h];
                        row[q[h]] = cs[u][h][1] * vp[h] + cs[u][h][0] * vq[h];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (r < L)
        for (int j = lane; j < L; j += 64) V[(int64_t)r * ldv + rank[j]] = row[j];
}

#ifndef PETAL_NS_CAP
#define PETAL_NS_CAP 1.25
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {  // v as seen through the DPP lane pattern CTRL (VALU speed, no LDS)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// Wout = symmetric_decorrelation(Win) (ica.rs:363-381).  S (= W W^T, then destroyed) and the eigenvector accumulator
// Zt are the Jacobi-hot matrices (LDS when nc <= 64); Z, Mm, w are scratch in global memory.
template <int MB>
__device__ void wg_symdecorr(const double* Win, double* Wout, int nc, int mode, double* S, double* Zt, double* Z, double* Mm,
                             double* w, JacWs& ws) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ls = MB > 0 ? (nc | 1) : nc;  // the LDS-resident matrices use an odd leading dimension
    for (int e = tid; e < nc * nc; e += nt) {  // S = W W^T (ica.rs:369)
        const int i = e / nc, j = e % nc;
        double acc = 0;
        for (int k = 0; k < nc; ++k) acc += Win[i * nc + k] * Win[j * nc + k];
        S[i * ls + j] = acc;
    }
    __syncthreads();
    wg_jacobi_any<MB>(S, ls, Zt, ls, nc, ws.c, ws.s, ws.p, ws.q, ws.red);
    wg_sort_eig(S, ls, Zt, ls, nc, Z, nc, w, ws.rank);
    // textbook (W W^T)^(-1/2) = Z D Z^T.  literal crate arithmetic: Z_asc^T D Z_asc with LAPACK's
    // ascending order; for nc == 2 LAPACK's dlaev2 path returns a symmetric Z for PSD input, where both agree.
    // The literal form is not invariant under eigenvector sign flips, and LAPACK's raw signs are an artefact of the
    // backend (MKL in the crate's CI, OpenBLAS elsewhere): eigenvectors are therefore sign-NORMALISED -- the first
    // component of largest magnitude made positive -- which the oracle's literal mode can be asked to do as well.
    const bool literal = mode == 1 && nc > 2;
    if (literal) {
        for (int c = tid; c < nc; c += nt) {
            double best = -1.0, sg = 1.0;
            for (int k = 0; k < nc; ++k) {
                const double v = Z[k * nc + c];
                if (fabs(v) > best) { best = fabs(v); sg = v < 0.0 ? -1.0 : 1.0; }
            }
            ws.red[c] = sg;
        }
        __syncthreads();
    }
    for (int e = tid; e < nc * nc; e += nt) {
        const int i = e / nc, j = e % nc;
        double acc = 0;
        for (int k = 0; k < nc; ++k) {
            if (literal) acc += Z[k * nc + (nc - 1 - i)] * (1.0 / sqrt(w[nc - 1 - k])) * Z[k * nc + (nc - 1 - j)];
            else acc += Z[i * nc + k] * (1.0 / sqrt(w[k])) * Z[j * nc + k];
        }
        if (literal) acc *= ws.red[nc - 1 - i] * ws.red[nc - 1 - j];
        Mm[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < nc * nc; e += nt) {
        const int i = e / nc, j = e % nc;
        double acc = 0;
        for (int k = 0; k < nc; ++k) acc += Mm[i * nc + k] * Win[k * nc + j];
        Wout[e] = acc;
    }
    __syncthreads();
}
// Orthogonal polar factor of D (= (D D^T)^(-1/2) D, the symmetric decorrelation of ica.rs:363-381) by the
// Newton-Schulz iteration X <- 1.5 X - 0.5 (X X^T) X from X0 = D / ||D||_F, entirely in LDS: two nc^3 products and
// three barriers per step instead of a full Jacobi eigen-solve (305 us -> tens of us at nc = 32).  Converges
// quadratically once the singular values are O(1); returns false (caller falls back to the eigen-solver) if D is
// singular / non-finite or 60 steps do not reach ||X X^T - I||_F <= 1e-13.
// On entry X (LDS, leading dimension nc | 1) holds D; returns the LDS buffer holding the polar factor, or nullptr.
// Leading dimension of the three LDS matrices of the iteration.  The fp64 MFMA operands are read as 16 rows x 4 consecutive doubles per
// instruction (lane (i, k): row i, column k0 + k): with the odd pitch of the Jacobi solvers (nc | 1: 130 dwords = 2 mod 64 banks) the
// lanes (i, k) and (i + 1, k - 1) share a bank pair -- every operand read of X X^T a 2- to 4-way conflict, and the products are LDS time,
// not matrix time (7400 clock ticks for 32 MFMAs per wave at nc = 64).  nc + 2 puts a row 4 (mod 8) dwords on: conflict-free for
// the row-wise operands (both of X X^T, T of T X).
__host__ __device__ constexpr int polar_ld(int nc) { return (nc & 15) ? (nc | 1) : nc + 2; }
// NTC: the number of 16-wide tiles per side when it is known at compile time (nc = 16 NTC: the products' k loops unroll completely and
// a tile's operand reads are all in flight before its first MFMA -- a 4-deep unroll exposed the LDS latency four times per tile), 0 otherwise.
template <int NTC>
__device__ double* wg_polar_ns(int nc_rt, double* X_, double* T_, double* Y_, double* s_red_, double tol2 = 1e-26) {
    // The four buffers are LDS.  Through generic pointers -- this function is not inlined into its callers -- every access was a FLAT
    // instruction (165 of them at nc = 64, each waiting on vmcnt AND lgkmcnt; the products looked like LDS-latency chains whatever was
    // done to them): the address space is stated here.  64 components: 48.8 -> 32.7 us per FastICA iteration's tail.
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) float lds_f32;
    lds_f64* X = (lds_f64*)X_;
    lds_f64* T = (lds_f64*)T_;
    lds_f64* Y = (lds_f64*)Y_;
    lds_f64* const s_red = (lds_f64*)s_red_;
    const int nc = NTC ? 16 * NTC : nc_rt;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    const int ld = polar_ld(nc);
    // nc a multiple of 16: the two nc^3 products of a step run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, operands
    // straight from LDS); the scalar loops below are LDS-bound -- two reads per FMA -- and cost 20 us per product at nc = 64
    const bool use_mfma = (nc & 15) == 0;
    const int ntile = nc >> 4;
    // ONE barrier per sum: consecutive calls alternate between two slots of s_red, so a call's writes cannot overtake the
    // reads of the call before it (whose slot is rewritten only two calls later, with the call in between's barrier behind them)
    int sum_slot = 0;
    auto block_sum = [&](double v) {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        lds_f64* slot = s_red + 16 * sum_slot;
        sum_slot ^= 1;
        if (lane == 0) slot[wv] = v;
        __syncthreads();
        // (ONE LDS read per lane and four DPP steps inside each row of 16 lanes -- every row forms the same sum, the same bits in
        // every lane -- instead of nw dependent reads: 8 x ~130 cycles, once per step)
        double t = (lane & 15) < nw ? slot[lane & 15] : 0.0;
        t += dpp_f64<0xB1>(t);
        t += dpp_f64<0x4E>(t);
        t += dpp_f64<0x141>(t);
        t += dpp_f64<0x140>(t);
        return t;
    };
    double ss = 0;
    for (int e = tid; e < nc * nc; e += nt) { const double v = X[(e / nc) * ld + (e % nc)]; ss += v * v; }
    const double fro2 = block_sum(ss);
    if (!(fro2 > 0.0) || !(fro2 < 1e300)) return nullptr;
    const double inv = 1.0 / sqrt(fro2);
    for (int e = tid; e < nc * nc; e += nt) X[(e / nc) * ld + (e % nc)] *= inv;
    __syncthreads();
    bool ok = false;
#ifdef PETAL_DEBUG_COUNTERS
    long long _t0 = clock64();
    if (tid == 0) g_dbg[2] += 1;   // polar factors formed / Newton-Schulz steps taken (dev/tail_phases.py)
#endif
    for (int it = 0; it < 60; ++it) {
        double err = 0;
#ifdef PETAL_DEBUG_COUNTERS
        DBG_T(16);
        if (tid == 0) g_dbg[1] += 1;
#endif
        if (use_mfma) {
            // T = X X^T: tile (ti, tj) per wave pass; A[i][k] = X[16 ti + i][k], B[k][j] = X[16 tj + j][k];
            // C/D: reg r of lane l is row (l >> 4) + 4 r, column l & 15
            for (int tile = wv; tile < ntile * ntile; tile += nw) {
                const int ti = tile / ntile, tj = tile - ti * ntile;
                const lds_f64* pa = X + (16 * ti + (lane & 15)) * ld + (lane >> 4);
                const lds_f64* pb = X + (16 * tj + (lane & 15)) * ld + (lane >> 4);
                f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
                if constexpr (NTC > 0) {
#pragma unroll
                    for (int k0 = 0; k0 < 16 * NTC; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], acc, 0, 0, 0);
                } else {
#pragma unroll 4
                    for (int k0 = 0; k0 < nc; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ti + (lane >> 4) + 4 * r, j = 16 * tj + (lane & 15);
                    T[i * ld + j] = acc[r];
                    const double dlt = acc[r] - (i == j ? 1.0 : 0.0);
                    err += dlt * dlt;
                }
            }
        } else {
            for (int e = tid; e < nc * nc; e += nt) {
                const int i = e / nc, j = e - i * nc;
                double acc = 0;
                for (int k = 0; k < nc; ++k) acc += X[i * ld + k] * X[j * ld + k];
                T[i * ld + j] = acc;
                const double dlt = acc - (i == j ? 1.0 : 0.0);
                err += dlt * dlt;
            }
        }
#ifdef PETAL_DEBUG_COUNTERS
        DBG_T(17);
#endif
        const double terr = block_sum(err);  // (its barriers also publish T)
#ifdef PETAL_DEBUG_COUNTERS
        DBG_T(18);
#endif
        if (!(terr == terr)) return nullptr;
        if (terr <= tol2) { ok = true; break; }  // tol2 = the square of the accepted ||X X^T - I||_F
        // Scaled step: g = max_i sum_j |T_ij| >= lambda_max(T) = sigma_max(X)^2 (Gershgorin), so X / sqrt(g) still has all
        // singular values <= 1 (the iteration stays monotone) but the largest one is pushed towards 1.  For a nearly
        // orthogonal D -- the FastICA case: D ~ beta W -- T is nearly diagonal, the bound is tight, and the iteration
        // turns quadratic after one step instead of crawling up from sigma = 1 / sqrt(nc) by factors of 1.5.
        // (LDS latency is ~110 cycles: a serial loop over nc entries costs microseconds, so the row sums use eight lanes per
        // row and the maximum is a per-wave DPP reduction; fp32 is plenty for a scaling bound)
        // Once ||T - I||_F <= 0.3 every singular value of X lies in [0.83, 1.15]: the plain step X (3 I - T) / 2 converges
        // quadratically from there (errors 0.3 -> 0.1 -> 0.02 -> 5e-4 -> 4e-7 -> ...) and the bounds below -- one more pass
        // over T and a barrier per step -- would only reproduce a ~ 1, g ~ 1.
        double ca = 1.5, cb = 0.5;
        if (terr > 0.09) {
        lds_f32* s_rs = (lds_f32*)(s_red + 64);
        for (int row = tid >> 3; row < nc; row += nt >> 3) {
            float rs = 0.f;
            for (int j = tid & 7; j < nc; j += 8) rs += (float)fabs(T[row * ld + j]);
            rs += dpp_f32<0xB1>(rs);   // quad_perm [1,0,3,2]
            rs += dpp_f32<0x4E>(rs);   // quad_perm [2,3,0,1]
            rs += dpp_f32<0x141>(rs);  // row_half_mirror
            if ((tid & 7) == 0) { s_rs[row] = rs * 1.000001f; s_rs[64 + row] = 2.f * (float)T[row * ld + row] - rs * 1.000001f; }
        }
        __syncthreads();
        float gm = lane < nc ? s_rs[lane] : 0.f;
        gm = fmaxf(gm, dpp_f32<0xB1>(gm));
        gm = fmaxf(gm, dpp_f32<0x4E>(gm));
        gm = fmaxf(gm, dpp_f32<0x141>(gm));
        gm = fmaxf(gm, dpp_f32<0x140>(gm));  // row_mirror: every lane of a 16-lane row holds the row maximum
        auto row_max = [&](int l0) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gm), l0)); };
        const double g = (double)fmaxf(fmaxf(row_max(0), row_max(16)), fmaxf(row_max(32), row_max(48)));
        // ... and a lower bound on lambda_min(T): Gershgorin's min_i (T_ii - sum_{j != i} |T_ij|), or 1 - ||T - I||_F
        float gl = lane < nc ? s_rs[64 + lane] : 3.0e38f;
        gl = fminf(gl, dpp_f32<0xB1>(gl));
        gl = fminf(gl, dpp_f32<0x4E>(gl));
        gl = fminf(gl, dpp_f32<0x141>(gl));
        gl = fminf(gl, dpp_f32<0x140>(gl));
        auto row_min = [&](int l0) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), l0)); };
        const double glo = fmax(fmax((double)fminf(fminf(row_min(0), row_min(16)), fminf(row_min(32), row_min(48))), 1.0 - sqrt(terr)), 0.0);
        // Over-relaxed step (Chen & Chow's scaled Newton-Schulz): with the singular values of X / sqrt(g) in [l, 1] the
        // polynomial a x (3 - a^2 x^2) / 2 with a = sqrt(3 / (1 + l + l^2)) maps both ends to the same value and lifts the
        // small ones by up to 1.5 a per step instead of 1.5.  l comes from the lower bound above (0 while T is far from
        // diagonally dominant); a is capped at 1.25 (measured: 1.35 costs steps at nc = 64) so that an underestimated l cannot push the large singular values down
        // by more than 10 % (a -> 1 as l -> 1: the last steps are the plain quadratic iteration).
        const double gs = (g > 0.0 && g < 1.0) ? g : 1.0;
        const double l = sqrt(fmin(glo / gs, 1.0));
        const double al = fmin(PETAL_NS_CAP, sqrt(3.0 / (1.0 + l + l * l)));
        const double rg = 1.0 / sqrt(gs);
        ca = 1.5 * al * rg; cb = 0.5 * al * al * al * rg / gs;
        }
        if (use_mfma) {
            // Y = ca X - cb T X: A[i][k] = T[16 ti + i][k], B[k][j] = X[k][16 tj + j]
            for (int tile = wv; tile < ntile * ntile; tile += nw) {
                const int ti = tile / ntile, tj = tile - ti * ntile;
                const lds_f64* pa = T + (16 * ti + (lane & 15)) * ld + (lane >> 4);
                const lds_f64* pb = X + (lane >> 4) * ld + 16 * tj + (lane & 15);
                f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
                if constexpr (NTC > 0) {
#pragma unroll
                    for (int k0 = 0; k0 < 16 * NTC; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0 * ld], acc, 0, 0, 0);
                } else {
#pragma unroll 4
                    for (int k0 = 0; k0 < nc; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0 * ld], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ti + (lane >> 4) + 4 * r, j = 16 * tj + (lane & 15);
                    Y[i * ld + j] = ca * X[i * ld + j] - cb * acc[r];
                }
            }
        } else {
            for (int e = tid; e < nc * nc; e += nt) {
                const int i = e / nc, j = e - i * nc;
                double acc = 0;
                for (int k = 0; k < nc; ++k) acc += T[i * ld + k] * X[k * ld + j];
                Y[i * ld + j] = ca * X[i * ld + j] - cb * acc;
            }
        }
        __syncthreads();
        lds_f64* sw = X; X = Y; Y = sw;
    }
    return ok ? (double*)X : nullptr;
}

constexpr int ICA_TAIL_THREADS = 512;
template <int MB>
__global__ __launch_bounds__(ICA_TAIL_THREADS) void k_symdecorr(const double* Win, double* Wout, int nc, int mode, double* scratch,
                                                                int* zero2) {
    if (zero2 && threadIdx.x < 2) zero2[threadIdx.x] = 0;   // (the loop's {converged, iterations} state, cleared by the way)
    constexpr bool use_lds = MB > 0;
    extern __shared__ __attribute__((aligned(16))) double sm_sd[];
    JacWs ws = jac_carve(sm_sd, nc, blockDim.x);
    double* S = scratch; double* Zt = S + nc * nc; double* Z = Zt + nc * nc; double* Mm = Z + nc * nc; double* w = Mm + nc * nc;
    if (use_lds) { S = sm_sd + jac_ws_doubles(nc, blockDim.x); Zt = S + nc * (nc | 1); }
    // textbook (W W^T)^(-1/2) W is the orthogonal polar factor of W: scaled Newton-Schulz first, eigen-solver as fallback
    if constexpr (MB > 0) {
        if (mode == 0 || nc <= 2) {
            const int ld = polar_ld(nc);
            double* const P1 = S + nc * ld;   // (the polar iteration's own carving of the matrices' space: its pitch is not the solver's)
            for (int e = threadIdx.x; e < nc * nc; e += blockDim.x) S[(e / nc) * ld + (e % nc)] = Win[e];
            __syncthreads();
            constexpr int NTS = MB <= 4 ? MB : 0;   // (80 / 96 components: the unrolled products spill)
            const double* res = nc == 16 * NTS ? wg_polar_ns<NTS>(nc, S, P1, P1 + nc * ld, ws.red) : wg_polar_ns<0>(nc, S, P1, P1 + nc * ld, ws.red);
            if (res) {
                for (int e = threadIdx.x; e < nc * nc; e += blockDim.x) Wout[e] = res[(e / nc) * ld + (e % nc)];
                return;
            }
            __syncthreads();
        }
    }
    wg_symdecorr<MB>(Win, Wout, nc, mode, S, Zt, Z, Mm, w, ws);
}
template <int MB>
__global__ __launch_bounds__(ICA_TAIL_THREADS) void k_ica_tail(int nc, double n_total, double* W, const double* GX_gp, int mode,
                                                               double tol, int* state, int iter, double* scratch,
                                                               bf16x8* __restrict__ wpk3, double ortho_tol2, int* progress) {
    // (the loop's "converged" word is only LOOKED AT in front of the first write: its load, a memory round trip, travels with the
    // operands' instead of in front of them; a launch queued behind the converging one computes for nothing and leaves no trace)
    const int done = __builtin_nontemporal_load(state);
    constexpr bool use_lds = MB > 0;
    extern __shared__ __attribute__((aligned(16))) double sm_tail[];
    const int tid = threadIdx.x, nt = blockDim.x;
    JacWs ws = jac_carve(sm_tail, nc, nt);
    double* S = scratch; double* Zt = S + nc * nc; double* Z = Zt + nc * nc; double* Mm = Z + nc * nc; double* w = Mm + nc * nc;
    double* D = w + nc; double* W1 = D + nc * nc;
    if (use_lds) { S = sm_tail + jac_ws_doubles(nc, nt); Zt = S + nc * (nc | 1); }
    const double* GX = GX_gp; const double* gp = GX_gp + nc * nc;
    const double pinv = 1.0 / n_total;
    const int ldl = polar_ld(nc);
    double* const P1 = S + nc * ldl;   // (the polar iteration's own carving of the matrices' space)
    double* Wl = nullptr;        // LDS copy of W (fast path): one global read of W, none of W1
    const double* res = nullptr; // LDS result of the polar iteration
    if constexpr (MB > 0) {
        if (mode == 0 || nc <= 2) {
            Wl = P1 + 2 * nc * ldl;
            // D = GX / n - diag(g') W / n (ica.rs:334-342), straight into LDS; the loads of four trips are in flight together (a
            // load -> LDS-store loop pays a memory round trip per trip: eight in a row at nc = 64)
            for (int e0 = tid; e0 < nc * nc; e0 += 4 * nt) {
                double wv4[4], gx4[4], gp4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = min(e0 + u * nt, nc * nc - 1);
                    wv4[u] = W[e]; gx4[u] = GX[e]; gp4[u] = gp[e / nc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * nt;
                    if (e < nc * nc) {
                        const int i = e / nc, j = e - i * nc;
                        Wl[i * ldl + j] = wv4[u];
                        S[i * ldl + j] = gx4[u] * pinv - gp4[u] * pinv * wv4[u];
                    }
                }
            }
            __syncthreads();
            constexpr int NTS = MB <= 4 ? MB : 0;   // (80 / 96 components: the unrolled products spill)
            res = nc == 16 * NTS ? wg_polar_ns<NTS>(nc, S, P1, P1 + nc * ldl, ws.red, ortho_tol2)
                                 : wg_polar_ns<0>(nc, S, P1, P1 + nc * ldl, ws.red, ortho_tol2);  // ica.rs:343
        }
    }
    typedef __attribute__((address_space(3))) double lds_f64;   // (the polar factor and the copy of W are LDS: no FLAT accesses)
    const lds_f64* const resl = (const lds_f64*)res;
    const lds_f64* const Wll = (const lds_f64*)Wl;
    double lim = 0;  // ica.rs:344-354
    if (res) {
        // (eight lanes per row: one thread per row walked nc dependent LDS reads, ~4 us at 64 components)
        for (int i = tid >> 3; i < nc; i += nt >> 3) {
            double dot = 0;
            for (int j = tid & 7; j < nc; j += 8) dot += resl[i * ldl + j] * (mode == 1 ? Wll[j * ldl + i] : Wll[i * ldl + j]);  // ica.rs:345-349
            dot = oct_sum_f64(dot);
            lim = fmax(lim, fabs(fabs(dot) - 1.0));
        }
    } else {
        __syncthreads();
        for (int e = tid; e < nc * nc; e += nt) D[e] = GX[e] * pinv - gp[e / nc] * pinv * W[e];
        __syncthreads();
        wg_symdecorr<MB>(D, W1, nc, mode, S, Zt, Z, Mm, w, ws);
        for (int i = tid; i < nc; i += nt) {
            double dot = 0;
            for (int j = 0; j < nc; ++j) dot += W1[i * nc + j] * (mode == 1 ? W[j * nc + i] : W[i * nc + j]);
            lim = fmax(lim, fabs(fabs(dot) - 1.0));
        }
    }
    // max over the workgroup: inside each wave by shuffles, across the waves through LDS (one barrier pair; a tree over the 512
    // threads was nine)
    for (int off = 32; off > 0; off >>= 1) lim = fmax(lim, __shfl_xor(lim, off, 64));
    __syncthreads();                                              // (ws.red's last readers are done)
    if ((tid & 63) == 0) ws.red[tid >> 6] = lim;
    __syncthreads();
    double tl = 0;
    for (int w = 0; w < (nt >> 6); ++w) tl = fmax(tl, ws.red[w]);
    if (done) return;                                             // (uniform)
    if (res) { for (int e = tid; e < nc * nc; e += nt) W[e] = resl[(e / nc) * ldl + (e % nc)]; }
    else { for (int e = tid; e < nc * nc; e += nt) W[e] = W1[e]; }
    if (wpk3) {  // the next step kernel's operand planes (k_pack_w3's layout), straight from the new W
        const int NT = (nc + 15) >> 4, KCH = (16 * NT + 31) >> 5;
        for (int idx = tid; idx < KCH * NT * 64; idx += nt) {
            const int lane = idx & 63, tile = idx >> 6, u = tile % NT, kc = tile / NT;
            const int comp = 16 * u + (lane & 15), k0 = 32 * kc + 8 * (lane >> 4);
            f32x8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                x[e] = (comp < nc && k0 + e < nc) ? (float)(res ? resl[comp * ldl + k0 + e] : W1[comp * nc + k0 + e]) : 0.f;
            bf16x8 h, m, l;
            split3(x, h, m, l);
            wpk3[(tile * 3 + 0) * 64 + lane] = h;
            wpk3[(tile * 3 + 1) * 64 + lane] = m;
            wpk3[(tile * 3 + 2) * 64 + lane] = l;
        }
    }
    if (tid == 0 && (tl < tol)) { state[0] = 1; state[1] = iter + 1; }  // ica.rs:355-357
    if (tid == 0 && progress) {  // host-visible progress: plain stores to pinned memory, the host polls without a sync
        if (tl < tol) __hip_atomic_store(progress, iter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(progress + 1, iter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void k_dscal(double* x, int64_t count, double alpha) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < count) x[e] *= alpha;
}
__global__ void k_daxpy(int64_t count, double alpha, const double* x, double* y) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < count) y[e] += alpha * x[e];
}
__global__ void k_dvec(int mode, const double* x, double* y, int64_t count, double thr) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= count) return;
    if (mode == 0) y[e] = sqrt(fmax(x[e], 0.0));
    else if (mode == 2) y[e] = x[e] * x[e];
    else y[e] = (x[e] > thr * x[0] && x[e] > 0.0) ? 1.0 / x[e] : 0.0;
}
__global__ void k_sigma_inv(const double* __restrict__ lam, double* __restrict__ sig, double* __restrict__ inv, int64_t count, double thr) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= count) return;
    const double s = sqrt(fmax(lam[e], 0.0)), s0 = sqrt(fmax(lam[0], 0.0));
    sig[e] = s;
    inv[e] = (s > thr * s0 && s > 0.0) ? 1.0 / s : 0.0;
}
// The fitted components straight in the caller's layout and type (pca.rs:543: rows of V^T): comp[j][i] = (T)(B^T u_j)[i] / sigma_j
// for j < k, from Bt (d x l fp64), the eigenvectors Uh (columns) and eigenvalues lam of B B^T.  1 / sigma_j is formed here
// (sigma_j = sqrt(lam_j); 0 below thr sigma_0, the rule of k_sigma_inv), so the fit ends with one small launch and a k x d copy in
// the output type instead of a scaling launch, an fp64 d x l product and a d x l fp64 copy.  Block = 64 i's x 16 j's.
template <class T>
__global__ __launch_bounds__(256) void k_components_out(const double* __restrict__ Bt, int64_t ldb, const double* __restrict__ Uh,
                                                        int64_t ldu, const double* __restrict__ lam, double thr, int64_t d, int L,
                                                        int64_t k, T* __restrict__ comp) {
    __shared__ double sB[64][65];       // [i][l] of the current 64-wide slice of l
    __shared__ double sU[64][16];       // [l][j]
    const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 16;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int l0 = 0; l0 < L; l0 += 64) {
        // (all of a slice's loads are issued before the first LDS store waits for one: a runtime-trip loop of load -> store
        // pairs pays an L2 round trip per pair)
        double vb[16], vu[4];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = (tid >> 6) + 4 * u, l = l0 + (tid & 63);
            vb[u] = (i0 + r < d && l < L) ? Bt[(i0 + r) * ldb + l] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = l0 + (tid >> 4) + 16 * u;
            vu[u] = (l < L && j0 + ti < k) ? Uh[(int64_t)l * ldu + j0 + ti] : 0.0;
        }
        __syncthreads();   // (the previous slice's readers are done)
#pragma unroll
        for (int u = 0; u < 16; ++u) sB[(tid >> 6) + 4 * u][tid & 63] = vb[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) sU[(tid >> 4) + 16 * u][ti] = vu[u];
        __syncthreads();
#pragma unroll 8
        for (int l = 0; l < 64; ++l) {
            const double u = sU[l][tj];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += sB[ti + 16 * r][l] * u;
        }
    }
    const int64_t j = j0 + tj;
    if (j < k) {
        const double s = sqrt(fmax(lam[j], 0.0)), s0 = sqrt(fmax(lam[0], 0.0));
        const double inv = (s > thr * s0 && s > 0.0) ? 1.0 / s : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + ti + 16 * r;
            if (i < d) comp[j * d + i] = (T)(acc[r] * inv);
        }
    }
}
// exact Pca's small-matrix tail, two one-launch helpers:
// P[i][j] = j < r ? V[i][j] * inv[j] : 0 (rows x rp, the right-hand side of U = Xc V / sigma, zero padded) ...
__global__ void k_scale_pad_cols(const double* __restrict__ V, int64_t ldv, const double* __restrict__ inv, int64_t rows, int64_t r,
                                 int64_t rp, double* __restrict__ P) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * rp) return;
    const int64_t i = e / rp, j = e - i * rp;
    P[e] = j < r ? V[i * ldv + j] * inv[j] : 0.0;
}
// ... and the components in the caller's layout and type: comp[j][i] = (T)V[i][j] for j < k, i < d (32 x 32 tiles through LDS)
template <class T>
__global__ __launch_bounds__(256) void k_transpose_out(const double* __restrict__ V, int64_t ldv, int64_t d, int64_t k, T* __restrict__ comp) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + ty + 8 * u, j = j0 + tx;
        tile[ty + 8 * u][tx] = (i < d && j < k) ? V[i * ldv + j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t j = j0 + ty + 8 * u, i = i0 + tx;
        if (j < k && i < d) comp[j * d + i] = (T)tile[tx][ty + 8 * u];
    }
}
// column means from the per-block partial sums of k_colsum_part2, in one launch: fixed-order fp64 sum of the parts,
// mu64[j] = sum / n_total for j < d (the sums of squares at d <= j < w stay unscaled), muT[j] = (T)mu64[j]
template <class T>
__global__ __launch_bounds__(256) void k_colmean_final(const double* __restrict__ part, int64_t nparts, int64_t w, int64_t d,
                                                       double inv_n, double* __restrict__ mu64, T* __restrict__ muT) {
    __shared__ double red[32][9];
    const int e = threadIdx.x & 7, pl = threadIdx.x >> 3;  // 8 outputs x 32 part-lanes per block
    const int64_t j = (int64_t)blockIdx.x * 8 + e;
    double acc = 0;
    if (j < w)
        for (int64_t p = pl; p < nparts; p += 32) acc += part[p * w + j];
    red[pl][e] = acc;
    __syncthreads();
    if (pl == 0 && j < w) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][e];
        if (j < d) { t *= inv_n; muT[j] = (T)t; }
        mu64[j] = t;
    }
}
__global__ void k_dscale_cols(double* A, int64_t M, int64_t N, int64_t lda, const double* s) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < M * N) A[(e / N) * lda + (e % N)] *= s[e % N];
}
template <class T>
__global__ void k_cvt_from_f64(T* dst, const double* src, int64_t count) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < count) dst[e] = (T)src[e];
}
// dst (rows_p x cols_p fp64, zero padded) <- the leading rows x cols block of src (row-major, leading dimension lds)
template <class T>
__global__ void k_pad_to_f64(double* __restrict__ dst, int64_t rows_p, int64_t cols_p, const T* __restrict__ src, int64_t rows,
                             int64_t cols, int64_t lds, double* __restrict__ zero_ptr, int64_t zero_count) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < zero_count) zero_ptr[e] = 0.0;
    if (e >= rows_p * cols_p) return;
    const int64_t r = e / cols_p, c = e - r * cols_p;
    dst[e] = (r < rows && c < cols) ? (double)src[r * lds + c] : 0.0;
}
template <class T>
__global__ void k_cvt_to_f64(double* dst, const T* src, int64_t count) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e < count) dst[e] = (double)src[e];
}

// ---- FastICA step with more than 64 components: the pieces around the two GEMM kernels -------------------------------
__global__ void k_transpose_pad(const double* __restrict__ W, int64_t nc, double* __restrict__ WT, int64_t ncp) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ncp * ncp) return;
    const int64_t k = e / ncp, i = e - k * ncp;  // WT[k][i] = W[i][k]
    WT[e] = (k < nc && i < nc) ? W[i * nc + k] : 0.0;
}
template <class T>
__global__ void k_tanh_inplace(T* __restrict__ x, int64_t count) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) x[e] = (T)tanh((double)x[e]);
}
// GX_gp = [ GXp[:nc, :nc] | n - sum_s g_si^2 ]   (sum_s (1 - g^2) = n - sum g^2, fp64)
__global__ void k_ica_big_out(const double* __restrict__ GXp, const double* __restrict__ sumsq, double n, int64_t nc, int64_t ncp,
                              double* __restrict__ GX_gp, const int* __restrict__ state) {
    if (state && state[0]) return;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nc * nc) GX_gp[e] = GXp[(e / nc) * ncp + (e % nc)];
    else if (e < nc * nc + nc) GX_gp[e] = n - sumsq[e - nc * nc];
}

// ================================================================================================
// host-side launchers
// ================================================================================================
#define DISPATCH_T(dt, ...)                              \
    do {                                                 \
        if ((dt) == F64) { using T = double; __VA_ARGS__; } \
        else { using T = float; __VA_ARGS__; }           \
    } while (0)

void op_pack_strided(Dev* d, int dt, const void* src, int64_t n, int64_t dd, int64_t rs, int64_t cs, void* dst, int64_t ld,
                     int64_t dpad) {
    if (n == 0 || dpad == 0) return;
    const int64_t blocks = (n * dpad + 1023) / 1024;
    if (blocks < (int64_t(1) << 31)) {
        DISPATCH_T(dt, hipLaunchKernelGGL(k_pack_flat<T>, dim3((unsigned)blocks), dim3(256), 0, d->stream, (const T*)src, n, dd, rs, cs,
                                          (T*)dst, ld, dpad));
    } else {
        DISPATCH_T(dt, hipLaunchKernelGGL(k_pack_strided<T>, dim3((unsigned)n, cdiv(dpad, 256)), dim3(256), 0, d->stream,
                                          (const T*)src, n, dd, rs, cs, (T*)dst, ld, dpad));
    }
    launch_check();
}
void op_unpack_strided(Dev* d, int dt, const void* src, int64_t n, int64_t dd, int64_t ld, void* dst, int64_t rs, int64_t cs,
                       const double* scale) {
    if (n == 0 || dd == 0) return;
    const int64_t blocks = (n * dd + 1023) / 1024;
    if (blocks < (int64_t(1) << 31)) {
        DISPATCH_T(dt, hipLaunchKernelGGL(k_unpack_scaled<T>, dim3((unsigned)blocks), dim3(256), 0, d->stream, (const T*)src, n, dd, ld,
                                          (T*)dst, rs, cs, scale));
    } else {
        if (scale) op_scale_cols(d, dt, const_cast<void*>(src), n, dd, ld, scale);
        DISPATCH_T(dt, hipLaunchKernelGGL(k_unpack_strided<T>, dim3((unsigned)n, cdiv(dd, 256)), dim3(256), 0, d->stream,
                                          (const T*)src, n, dd, ld, (T*)dst, rs, cs));
    }
    launch_check();
}

// first stage of the column sums (a 16-byte-per-lane form measured the same: EXPERIMENTS.md round 4)
static void launch_colsum_parts(Dev* d, int dt, const void* X, int64_t n, int64_t dd, int64_t ldx, int64_t rows, int64_t nparts,
                                double* part, bool with_sq) {
    const dim3 grid((unsigned)nparts, cdiv(dd, 64));
    if (with_sq) {
        DISPATCH_T(dt, hipLaunchKernelGGL((k_colsum_part2<T, true>), grid, dim3(256), 0, d->stream, (const T*)X, n, dd, ldx, rows, part));
    } else {
        DISPATCH_T(dt, hipLaunchKernelGGL((k_colsum_part2<T, false>), grid, dim3(256), 0, d->stream, (const T*)X, n, dd, ldx, rows, part));
    }
    launch_check();
}
void op_colsum(Dev* d, int dt, const void* X, int64_t n, int64_t dd, int64_t ldx, double* out, bool with_sq) {
    if (dd == 0) return;
    const int64_t w = with_sq ? 2 * dd : dd;
    if (n == 0) { dev_memset(d, out, 0, sizeof(double) * w); return; }
    const int64_t rows = scan_rows_per_block(n, dd), nparts = cdiv(n, rows);
    double* part = (double*)dev_alloc(d, sizeof(double) * nparts * w);
    launch_colsum_parts(d, dt, X, n, dd, ldx, rows, nparts, part, with_sq);
    hipLaunchKernelGGL(k_sum_parts2<double>, dim3(cdiv(w, 32)), dim3(256), 0, d->stream, part, nparts, w, out, w, w, false);
    launch_check();
    dev_free(d, part);
}

void op_colmean(Dev* d, int dt, const void* X, int64_t n, int64_t dd, int64_t ldx, double n_total, double* mu64, void* muT, bool with_sq) {
    if (dd == 0) return;
    const int64_t w = with_sq ? 2 * dd : dd;
    if (n == 0) { dev_memset(d, mu64, 0, sizeof(double) * w); dev_memset(d, muT, 0, dtype_size(dt) * dd); return; }
    const int64_t rows = scan_rows_per_block(n, dd), nparts = cdiv(n, rows);
    double* part = (double*)dev_alloc(d, sizeof(double) * nparts * w);
    {
        TagScope ts(d);   // (the pass over X; bracketed when the caller tagged it)
        launch_colsum_parts(d, dt, X, n, dd, ldx, rows, nparts, part, with_sq);
        ts.stop();
    }
    DISPATCH_T(dt, hipLaunchKernelGGL(k_colmean_final<T>, dim3(cdiv(w, 8)), dim3(256), 0, d->stream, part, nparts, w, dd, 1.0 / n_total, mu64, (T*)muT));
    launch_check();
    dev_free(d, part);
}
void op_sigma_inv(Dev* d, const double* lam, double* sig, double* inv, int64_t count, double thr) {
    if (!count) return;
    hipLaunchKernelGGL(k_sigma_inv, dim3(cdiv(count, 256)), dim3(256), 0, d->stream, lam, sig, inv, count, thr);
    launch_check();
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// split-product (bf16x3, see k_xp3) kernels unless the ctx asks for the fp32-MFMA ones (petal_ctx_set_gemm_mode /
// PETAL_GEMM=fp32)
static bool gemm_split_product(const Dev* d) { return d->gemm_mode != 1; }   // (mode 2: the same kernels, three-plane operands only: algo.cpp)
// kernels that ask for more than 64 KB of dynamic LDS: the attribute is set once per device (ctx)
static void set_max_lds(Dev* d, const void* fn) {
    if (d->max_lds_set.count(fn)) return;
    HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    d->max_lds_set.insert(fn);
}
void op_scale_pad_cols(Dev* d, const double* V, int64_t ldv, const double* inv, int64_t rows, int64_t r, int64_t rp, double* P) {
    if (rows == 0 || rp == 0) return;
    hipLaunchKernelGGL(k_scale_pad_cols, dim3(cdiv(rows * rp, 256)), dim3(256), 0, d->stream, V, ldv, inv, rows, r, rp, P);
    launch_check();
}
void op_transpose_out(Dev* d, int dt, const double* V, int64_t ldv, int64_t dd, int64_t k, void* comp) {
    if (dd == 0 || k == 0) return;
    const dim3 grid(cdiv(dd, 32), cdiv(k, 32));
    if (dt == F32) hipLaunchKernelGGL(k_transpose_out<float>, grid, dim3(256), 0, d->stream, V, ldv, dd, k, (float*)comp);
    else hipLaunchKernelGGL(k_transpose_out<double>, grid, dim3(256), 0, d->stream, V, ldv, dd, k, (double*)comp);
    launch_check();
}
void op_components_out(Dev* d, int dt, const double* Bt, int64_t ldb, const double* Uh, int64_t ldu, const double* lam, double thr,
                       int64_t dd, int64_t L, int64_t k, void* comp) {
    if (k == 0 || dd == 0) return;
    const dim3 grid(cdiv(dd, 64), cdiv(k, 16));
    if (dt == F32)
        hipLaunchKernelGGL(k_components_out<float>, grid, dim3(256), 0, d->stream, Bt, ldb, Uh, ldu, lam, thr, dd, (int)L, k, (float*)comp);
    else
        hipLaunchKernelGGL(k_components_out<double>, grid, dim3(256), 0, d->stream, Bt, ldb, Uh, ldu, lam, thr, dd, (int)L, k, (double*)comp);
    launch_check();
}
static int num_cus(Dev* d) {
    if (!d->num_cu) { hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, d->device)); d->num_cu = prop.multiProcessorCount; }
    return d->num_cu;
}

template <int RT, int NT>
static void launch_xp(Dev* d, const float* X, int64_t n, int K, int64_t ldx, const float* mu, const float* Ppk, int NTtot,
                      int nt0, int N, const float* bias, float* Z, int64_t ldz, double* ss_part, int blocks) {
    const bool center = mu != nullptr, ss = ss_part != nullptr;
#define XP_ARGS X, n, K, ldx, mu, Ppk, NTtot, nt0, N, bias, Z, ldz, ss_part
    if (center && ss) hipLaunchKernelGGL((k_xp_mfma<RT, NT, true, true>), dim3(blocks), dim3(256), 0, d->stream, XP_ARGS);
    else if (center) hipLaunchKernelGGL((k_xp_mfma<RT, NT, true, false>), dim3(blocks), dim3(256), 0, d->stream, XP_ARGS);
    else if (ss) hipLaunchKernelGGL((k_xp_mfma<RT, NT, false, true>), dim3(blocks), dim3(256), 0, d->stream, XP_ARGS);
    else hipLaunchKernelGGL((k_xp_mfma<RT, NT, false, false>), dim3(blocks), dim3(256), 0, d->stream, XP_ARGS);
#undef XP_ARGS
    launch_check();
}

// prod_A != nullptr: the small operand is the product prod_A (K x prod_M) . P (prod_M x N) (op_gemm_xp_prod); only the
// split-product path forms it inside its pack kernel, every other path gets it from a GEMM launch first
// (split-product path only) svd_flip's column scan of the product, from the kernel's accumulators: cols columns, results as op_col_absmax's
struct AbsmaxReq { int64_t cols, row_offset; double *absmax, *idx, *sign; };
static void gemm_xp_impl(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N,
                         int64_t ldp, const void* bias, void* Z, int64_t ldz, double* sumsq, const double* prod_A, int64_t prod_M,
                         int64_t prod_lda, double* prod_out, int64_t prod_ldo, int prod_rt, const AbsmaxReq* am = nullptr,
                         bool p2_hint = false, bool steering = false);
void op_gemm_xp(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N,
                int64_t ldp, const void* bias, void* Z, int64_t ldz, double* sumsq, int p_planes, bool steering) {
    gemm_xp_impl(d, dt, X, n, K, ldx, mu, P, N, ldp, bias, Z, ldz, sumsq, nullptr, 0, 0, nullptr, 0, false, nullptr, p_planes == 2, steering);
}
void op_gemm_xp_prod(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* A, int64_t M,
                     int64_t lda, const double* T, int64_t N, int64_t ldt, double* P_out, int64_t ldpo, void* Z, int64_t ldz) {
    const bool fused = dt == F32 && gemm_split_product(d) && n >= 64 && K % 16 == 0 && K > 0 && N % 16 == 0 && ldx % 4 == 0 &&
                       aligned16(X) && (!mu || aligned16(mu)) && ldz % 4 == 0 && aligned16(Z) && K < (1 << 24) && N < (1 << 24);
    if (fused) {
        double* tmpo = P_out ? nullptr : (double*)dev_alloc(d, sizeof(double) * K * N);
        gemm_xp_impl(d, dt, X, n, K, ldx, mu, T, N, ldt, nullptr, Z, ldz, nullptr, A, M, lda, P_out ? P_out : tmpo, P_out ? ldpo : N, false);
        if (tmpo) dev_free(d, tmpo);
        return;
    }
    double* tmp = nullptr;
    double* P = P_out;
    int64_t ldp = ldpo;
    if (!P) { tmp = (double*)dev_alloc(d, sizeof(double) * K * N); P = tmp; ldp = N; }
    op_dgemm(d, false, false, K, N, M, 1.0, A, lda, T, ldt, 0.0, P, ldp);
    op_gemm_xp(d, dt, X, n, K, ldx, mu, P, N, ldp, nullptr, Z, ldz, nullptr);
    if (tmp) dev_free(d, tmp);
}
void op_gemm_xp_prod_absmax(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* A, int64_t M,
                            int64_t lda, const double* T, int64_t N, int64_t ldt, double* P_out, int64_t ldpo, void* Z, int64_t ldz,
                            int64_t row_offset, double* absmax, double* idx, double* sign, bool store_product, bool a_rt) {
    const bool fused = dt == F32 && gemm_split_product(d) && n >= 64 && K % 16 == 0 && K > 0 && N % 16 == 0 && N > 0 && ldx % 4 == 0 &&
                       aligned16(X) && (!mu || aligned16(mu)) && ldz % 4 == 0 && aligned16(Z) && K < (1 << 24) && N < (1 << 24) &&
                       n < (int64_t(1) << 31) && P_out != nullptr;
    if (a_rt && !(fused && M == K && M % 16 == 0 && M <= TRSM_MAXM)) throw std::logic_error("op_gemm_xp_prod_absmax: RT-form operand outside the fused path");
    if (!fused) {
        op_gemm_xp_prod(d, dt, X, n, K, ldx, mu, A, M, lda, T, N, ldt, P_out, ldpo, Z, ldz);
        op_col_absmax(d, dt, Z, n, N, ldz, row_offset, absmax, idx, sign);
        return;
    }
    const AbsmaxReq am{N, row_offset, absmax, idx, sign};
    // (store_product = false: the caller only wants svd_flip's scan -- the product stays in the accumulators, 26 MB less to write at
    // configs[1])
    gemm_xp_impl(d, dt, X, n, K, ldx, mu, T, N, ldt, nullptr, store_product ? Z : nullptr, ldz, nullptr, A, M, lda, P_out, ldpo, a_rt ? 2 : 0, &am);
}
// Z = (X - mu) P with svd_flip's column scan (first row of largest |z| per column, its sign) taken from the product kernel's
// accumulators where that kernel runs -- no second pass over Z, and no store of Z at all when store_product is false.
void op_gemm_xp_absmax(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N,
                       int64_t ldp, void* Z, int64_t ldz, int64_t row_offset, double* absmax, double* idx, double* sign, bool store_product) {
    const bool fused = dt == F32 && gemm_split_product(d) && n >= 64 && K % 16 == 0 && K > 0 && N % 16 == 0 && N > 0 && ldx % 4 == 0 &&
                       aligned16(X) && (!mu || aligned16(mu)) && ldz % 4 == 0 && aligned16(Z) && K < (1 << 24) && N < (1 << 24) &&
                       n < (int64_t(1) << 31);
    if (!fused) {
        op_gemm_xp(d, dt, X, n, K, ldx, mu, P, N, ldp, nullptr, Z, ldz, nullptr);
        op_col_absmax(d, dt, Z, n, N, ldz, row_offset, absmax, idx, sign);
        return;
    }
    const AbsmaxReq am{N, row_offset, absmax, idx, sign};
    gemm_xp_impl(d, dt, X, n, K, ldx, mu, P, N, ldp, nullptr, store_product ? Z : nullptr, ldz, nullptr, nullptr, 0, 0, nullptr, 0, false, &am);
}
static void gemm_xp_impl(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N,
                         int64_t ldp, const void* bias, void* Z, int64_t ldz, double* sumsq, const double* prod_A, int64_t prod_M,
                         int64_t prod_lda, double* prod_out, int64_t prod_ldo, int prod_rt, const AbsmaxReq* am, bool p2_hint, bool steering) {
    if (n == 0 || N == 0) return;
    const bool mfma = dt == F32 && K % 16 == 0 && K > 0 && ldx % 4 == 0 && aligned16(X) && (!mu || aligned16(mu)) && n >= 64 &&
                      K < (1 << 24) && N < (1 << 24) && N % 16 == 0 && ldz % 4 == 0 && aligned16(Z) && (!bias || aligned16(bias));
    const bool mfma64 = dt == F64 && K % 8 == 0 && K > 0 && ldx % 2 == 0 && aligned16(X) && (!mu || aligned16(mu)) && n >= 64 &&
                        K < (1 << 24) && N < (1 << 24) && N % 16 == 0;
    if (am && !(mfma && gemm_split_product(d) && !sumsq)) throw std::logic_error("gemm_xp_impl: abs-max epilogue outside the split-product path");
    if (mfma64) {  // fp64 inputs: the fp64 matrix cores, 32-row wave tiles, column panels of <= 5 tiles
        const int NTtot = cdiv(N, 16);
        const int64_t total = (K / 8) * (int64_t)NTtot * 64;
        f64x2* Ppk = (f64x2*)dev_alloc(d, sizeof(f64x2) * total);
        hipLaunchKernelGGL(k_pack_p64, dim3(cdiv(total, 256)), dim3(256), 0, d->stream, P, K, N, ldp, Ppk, NTtot, total);
        launch_check();
        const int blocks = cdiv(n, 128);
        double* ssp = sumsq ? (double*)dev_alloc(d, sizeof(double) * blocks * 4) : nullptr;
        const double* Xd = (const double*)X; const double* mud = (const double*)mu; const double* bd = (const double*)bias; double* Zd = (double*)Z;
        TagScope ts(d);
        for (int nt0 = 0; nt0 < NTtot;) {
            const int rem = NTtot - nt0;
            const int w = rem >= 5 ? 5 : rem;
            double* sp = nt0 == 0 ? ssp : nullptr;
#define XP64_ARGS Xd, n, (int)K, ldx, mud, Ppk, NTtot, nt0, (int)N, bd, Zd, ldz, sp
#define XP64_LAUNCH(NTv)                                                                                                            \
            do {                                                                                                                    \
                if (mud && sp) hipLaunchKernelGGL((k_xp_f64<NTv, true, true>), dim3(blocks), dim3(256), 0, d->stream, XP64_ARGS);    \
                else if (mud) hipLaunchKernelGGL((k_xp_f64<NTv, true, false>), dim3(blocks), dim3(256), 0, d->stream, XP64_ARGS);    \
                else if (sp) hipLaunchKernelGGL((k_xp_f64<NTv, false, true>), dim3(blocks), dim3(256), 0, d->stream, XP64_ARGS);     \
                else hipLaunchKernelGGL((k_xp_f64<NTv, false, false>), dim3(blocks), dim3(256), 0, d->stream, XP64_ARGS);            \
            } while (0)
            switch (w) {
                case 5: XP64_LAUNCH(5); break;
                case 4: XP64_LAUNCH(4); break;
                case 3: XP64_LAUNCH(3); break;
                case 2: XP64_LAUNCH(2); break;
                default: XP64_LAUNCH(1); break;
            }
#undef XP64_LAUNCH
#undef XP64_ARGS
            launch_check();
            nt0 += w;
        }
        ts.stop();
        if (sumsq) {
            hipLaunchKernelGGL(k_add_scalar_parts, dim3(1), dim3(256), 0, d->stream, ssp, (int64_t)blocks * 4, sumsq);
            launch_check();
            dev_free(d, ssp);
        }
        dev_free(d, Ppk);
        return;
    }
    if (!mfma) {
        double* ssp = sumsq ? (double*)dev_alloc(d, sizeof(double) * n) : nullptr;
        TagScope ts(d);
        DISPATCH_T(dt, hipLaunchKernelGGL(k_xp_simple<T>, dim3((unsigned)n, cdiv(N, 64)), dim3(64), 0, d->stream, (const T*)X, n,
                                          K, ldx, (const T*)mu, P, N, ldp, (const T*)bias, (T*)Z, ldz, ssp));
        launch_check();
        ts.stop();
        if (sumsq) {
            hipLaunchKernelGGL(k_add_scalar_parts, dim3(1), dim3(256), 0, d->stream, ssp, n, sumsq);
            launch_check();
            dev_free(d, ssp);
        }
        return;
    }
    const int NTtot = cdiv(N, 16);
    if (gemm_split_product(d) && !sumsq) {
        // split-product (bf16x3) form: grid of 64-row wave tiles, column panels of <= 5 tiles
        const int64_t nch = (K + 31) / 32, total = nch * NTtot * 64;
        bf16x8* Ppk3 = (bf16x8*)dev_alloc(d, sizeof(bf16x8) * total * 3);
        // two-plane P (five piece products) where P is the re-based iterate of a power iteration: k_trsm_pack rounds it so
        // (development / test knobs: PETAL_NO_P2 keeps three planes everywhere, PETAL_NO_P2_ITERATE for the re-based iterate only,
        // PETAL_NO_P2_OMEGA for the sketch matrix only)
        const bool no_p2 = !opt_on(d, OPT_TWO_PLANE);
        const bool no_p2_it = no_p2 || !opt_on(d, OPT_TWO_PLANE_ITERATE), no_p2_om = no_p2 || !opt_on(d, OPT_TWO_PLANE_OMEGA);
        const bool p2 = p2_hint && ((prod_A && prod_rt && !no_p2_it) || (!prod_A && !no_p2_om)) && !am;
        const bool x2 = p2 && steering && opt_on(d, OPT_STEERING);   // (a steering pass: X on two planes too; the wide form only)
        if (prod_A) {
            // P = prod_A . P: the fp64 GEMM kernel writes the product (prod_out) AND its operand planes from its epilogue
            if (prod_rt == 2) {   // prod_A is R in RT form and the operand is R^-1 P: blocked back substitution, one wave per 16 columns
                switch ((int)(prod_M / 16)) {
#define PETAL_TRSML_CASE(NB)                                                                                                       \
    case NB:                                                                                                                       \
        hipLaunchKernelGGL((k_trsm_left_pack<NB>), dim3(cdiv(N, 16)), dim3(64), 0, d->stream, prod_A, prod_lda, P, ldp, prod_out, prod_ldo, Ppk3, NTtot); \
        break
                    PETAL_TRSML_CASE(1); PETAL_TRSML_CASE(2); PETAL_TRSML_CASE(3); PETAL_TRSML_CASE(4); PETAL_TRSML_CASE(5);
                    PETAL_TRSML_CASE(6); PETAL_TRSML_CASE(7); PETAL_TRSML_CASE(8); PETAL_TRSML_CASE(9);
#undef PETAL_TRSML_CASE
                    default: throw std::runtime_error("k_trsm_left_pack: order out of range");
                }
            } else if (prod_rt)   // P is R in RT form: blocked triangular solve instead of the product with the explicit inverse
                switch ((int)(N / 16)) {
#define PETAL_TRSM_CASE(NB)                                                                                                        \
    case NB:                                                                                                                       \
        if (p2) hipLaunchKernelGGL((k_trsm_pack<NB, true>), dim3(cdiv(K, 16)), dim3(64), trsm_lds_bytes(16 * NB), d->stream, prod_A, prod_lda, P,  \
                                   ldp, K, prod_out, prod_ldo, Ppk3, NTtot);                                                       \
        else hipLaunchKernelGGL((k_trsm_pack<NB, false>), dim3(cdiv(K, 16)), dim3(64), trsm_lds_bytes(16 * NB), d->stream, prod_A, prod_lda, P,  \
                           ldp, K, prod_out, prod_ldo, Ppk3, NTtot);                                                               \
        break
                    PETAL_TRSM_CASE(1); PETAL_TRSM_CASE(2); PETAL_TRSM_CASE(3); PETAL_TRSM_CASE(4); PETAL_TRSM_CASE(5);
                    PETAL_TRSM_CASE(6); PETAL_TRSM_CASE(7); PETAL_TRSM_CASE(8); PETAL_TRSM_CASE(9);
#undef PETAL_TRSM_CASE
                    default: throw std::runtime_error("k_trsm_pack: order out of range");
                }
            else
            hipLaunchKernelGGL(k_dgemm, dim3(cdiv(N, 16), cdiv(K, 16)), dim3(16, 16), 0, d->stream, false, false, K, N, prod_M, 1.0,
                               prod_A, prod_lda, P, ldp, 0.0, prod_out, prod_ldo, prod_M, (double*)nullptr, (const double*)nullptr,
                               Ppk3, NTtot);
        } else {
            hipLaunchKernelGGL(k_pack_p3, dim3(cdiv(total, 256)), dim3(256), 0, d->stream, P, K, N, ldp, Ppk3, NTtot, total);
        }
        launch_check();
        constexpr int RTv = PETAL_XP3_RT, DPv = PETAL_XP3_DEPTH;
        const float* Xf = (const float*)X; const float* muf = (const float*)mu; const float* bf = (const float*)bias; float* Zf = (float*)Z;
        static_assert(RTv == 4, "both workgroup shapes below cover 256 rows: one abs-max partial per 256 rows");
        const int64_t am_parts = cdiv(n, 256), am_ld = am ? am->cols : 0;
        double* am_part = am ? (double*)dev_alloc(d, sizeof(double) * 3 * am_parts * am_ld) : nullptr;
        TagScope ts(d);
        // Column panels: as few passes over X as 9-tile panels allow, the tiles spread evenly over them.  A panel of <= 5 tiles
        // runs on 64-row wave tiles (RT = 4), one of 6 .. 9 tiles on 32-row wave tiles (RT = 2: 72 accumulator registers at 9
        // tiles, eight waves per workgroup) -- every pass reads and splits X again, so l = 138 in one 9-tile pass instead of
        // 5 + 4 takes 0.41 instead of 0.55 ms at 250000 x 1024 (0.47 with four waves per workgroup, which stage the P chunk
        // twice as often; a one-wave-per-SIMD 64-row form had measured slower; at <= 5 tiles the 32-row form loses, 64 vs 54 us).
        const int npass = cdiv(NTtot, 9);
        for (int nt0 = 0, pass = 0; nt0 < NTtot; ++pass) {
            const int w = (NTtot - nt0 + (npass - pass) - 1) / (npass - pass);
            const size_t lds = sizeof(bf16x8) * 2 * w * 64 * (p2 ? 2 : 3) + sizeof(float) * 32 * nch;
#define XP3_LAUNCH8P(NTv, NPLv)                                                                                                           \
            do {                                                                                                                            \
                const int blocksw = cdiv(n, 256);                                                                                           \
                if (lds > 64 * 1024) set_max_lds(d, muf ? reinterpret_cast<const void*>(k_xp3<2, NTv, DPv, true, 8, PETAL_XP3_OCC, NPLv>) : reinterpret_cast<const void*>(k_xp3<2, NTv, DPv, false, 8, PETAL_XP3_OCC, NPLv>)); \
                if (muf) hipLaunchKernelGGL((k_xp3<2, NTv, DPv, true, 8, PETAL_XP3_OCC, NPLv>), dim3(blocksw), dim3(512), lds, d->stream, Xf, n, (int)K, ldx, muf, Ppk3, NTtot, nt0, (int)N, bf, Zf, ldz, am_part, am_ld); \
                else hipLaunchKernelGGL((k_xp3<2, NTv, DPv, false, 8, PETAL_XP3_OCC, NPLv>), dim3(blocksw), dim3(512), lds, d->stream, Xf, n, (int)K, ldx, muf, Ppk3, NTtot, nt0, (int)N, bf, Zf, ldz, am_part, am_ld); \
            } while (0)
#define XP3_LAUNCH8X(NTv)                                                                                                                 \
            do {                                                                                                                            \
                const int blocksw = cdiv(n, 256);                                                                                           \
                if (lds > 64 * 1024) set_max_lds(d, muf ? reinterpret_cast<const void*>(k_xp3<2, NTv, DPv, true, 8, PETAL_XP3_OCC, 2, true>) : reinterpret_cast<const void*>(k_xp3<2, NTv, DPv, false, 8, PETAL_XP3_OCC, 2, true>)); \
                if (muf) hipLaunchKernelGGL((k_xp3<2, NTv, DPv, true, 8, PETAL_XP3_OCC, 2, true>), dim3(blocksw), dim3(512), lds, d->stream, Xf, n, (int)K, ldx, muf, Ppk3, NTtot, nt0, (int)N, bf, Zf, ldz, am_part, am_ld); \
                else hipLaunchKernelGGL((k_xp3<2, NTv, DPv, false, 8, PETAL_XP3_OCC, 2, true>), dim3(blocksw), dim3(512), lds, d->stream, Xf, n, (int)K, ldx, muf, Ppk3, NTtot, nt0, (int)N, bf, Zf, ldz, am_part, am_ld); \
            } while (0)
#define XP3_LAUNCH8(NTv) do { if (p2 && x2) XP3_LAUNCH8X(NTv); else if (p2) XP3_LAUNCH8P(NTv, 2); else XP3_LAUNCH8P(NTv, 3); } while (0)
            if (w >= 6) {   // eight 32-row waves per workgroup: the P chunk is staged once per 256 rows, as in the 64-row form
                switch (w) {
                    case 9: XP3_LAUNCH8(9); break;
                    case 8: XP3_LAUNCH8(8); break;
                    case 7: XP3_LAUNCH8(7); break;
                    default: XP3_LAUNCH8(6); break;
                }
                launch_check();
                nt0 += w;
                continue;
            }
#define XP3_LAUNCHP(RTw, NTv, NPLv)                                                                                                        \
            do {                                                                                                                            \
                const int blocksw = cdiv(n, 64 * RTw);                                                                                      \
                if (lds > 64 * 1024) set_max_lds(d, muf ? reinterpret_cast<const void*>(k_xp3<RTw, NTv, DPv, true, 4, PETAL_XP3_OCC, NPLv>) : reinterpret_cast<const void*>(k_xp3<RTw, NTv, DPv, false, 4, PETAL_XP3_OCC, NPLv>)); \
                if (muf) hipLaunchKernelGGL((k_xp3<RTw, NTv, DPv, true, 4, PETAL_XP3_OCC, NPLv>), dim3(blocksw), dim3(256), lds, d->stream, Xf, n, (int)K, ldx, muf, Ppk3, NTtot, nt0, (int)N, bf, Zf, ldz, am_part, am_ld); \
                else hipLaunchKernelGGL((k_xp3<RTw, NTv, DPv, false, 4, PETAL_XP3_OCC, NPLv>), dim3(blocksw), dim3(256), lds, d->stream, Xf, n, (int)K, ldx, muf, Ppk3, NTtot, nt0, (int)N, bf, Zf, ldz, am_part, am_ld); \
            } while (0)
#define XP3_LAUNCH(RTw, NTv) do { if (p2) XP3_LAUNCHP(RTw, NTv, 2); else XP3_LAUNCHP(RTw, NTv, 3); } while (0)
            switch (w) {
                case 5: XP3_LAUNCH(RTv, 5); break;
                case 4: XP3_LAUNCH(RTv, 4); break;
                case 3: XP3_LAUNCH(RTv, 3); break;
                case 2: XP3_LAUNCH(RTv, 2); break;
                default: XP3_LAUNCH(RTv, 1); break;
            }
#undef XP3_LAUNCH
#undef XP3_LAUNCH8
#undef XP3_LAUNCHP
#undef XP3_LAUNCH8P
            launch_check();
            nt0 += w;
        }
        ts.stop();
        if (am) {
            hipLaunchKernelGGL(k_absmax_final, dim3((unsigned)am->cols), dim3(64), 0, d->stream, am_part, am_part + am_parts * am_ld,
                               am_part + 2 * am_parts * am_ld, am_parts, am_ld, am->row_offset, am->absmax, am->idx, am->sign);
            launch_check();
            dev_free(d, am_part);
        }
        dev_free(d, Ppk3);
        return;
    }
    float* Ppk = (float*)dev_alloc(d, sizeof(float) * (K / 16) * NTtot * 64 * 4);
    {
        const int64_t total = (K / 16) * (int64_t)NTtot * 64;
        hipLaunchKernelGGL(k_pack_p, dim3(cdiv(total, 256)), dim3(256), 0, d->stream, P, K, N, ldp, Ppk, NTtot);
        launch_check();
    }
    constexpr bool use_classic = false, force_pers = false;
    const int num_cu = num_cus(d);
    // Form selection (measured, MI355X): with fewer than two 64-row tiles per wave slot the grid form leaves SIMDs
    // a whole 64-row tile apart (100000 x 512: 74 TFLOP/s) and the balanced persistent form wins (84); with many tiles
    // per slot the hardware dispatcher balances the grid form dynamically and it is the faster one (1e6 x 512: 95 vs 76).
    const bool small = (n + 63) / 64 < (int64_t)num_cu * 8 * 2;
    if (!use_classic && (small || force_pers)) {
        // persistent form: one 512-thread workgroup per CU, balanced ranges of 16-row tiles, column panels of <= 5 tiles
        const int64_t ntiles16 = (n + 15) / 16;
        const int blocks = (int)std::min<int64_t>(num_cu, (ntiles16 + 7) / 8);
        double* ssp = sumsq ? (double*)dev_alloc(d, sizeof(double) * blocks * 8) : nullptr;
        TagScope ts(d);
        for (int nt0 = 0; nt0 < NTtot;) {
            const int rem = NTtot - nt0;
            const int w = rem >= 5 ? 5 : rem;
            double* sp = nt0 == 0 ? ssp : nullptr;
            const float* Xf = (const float*)X; const float* muf = (const float*)mu; const float* bf = (const float*)bias; float* Zf = (float*)Z;
            const bool center = muf != nullptr, ss = sp != nullptr;
#define XPP_ARGS Xf, n, (int)K, ldx, muf, Ppk, NTtot, nt0, (int)N, bf, Zf, ldz, sp, ntiles16
#define XPP_LAUNCH(NTv)                                                                                                      \
            do {                                                                                                             \
                if (center && ss) hipLaunchKernelGGL((k_xp_pers<NTv, true, true>), dim3(blocks), dim3(512), 0, d->stream, XPP_ARGS);   \
                else if (center) hipLaunchKernelGGL((k_xp_pers<NTv, true, false>), dim3(blocks), dim3(512), 0, d->stream, XPP_ARGS);   \
                else if (ss) hipLaunchKernelGGL((k_xp_pers<NTv, false, true>), dim3(blocks), dim3(512), 0, d->stream, XPP_ARGS);       \
                else hipLaunchKernelGGL((k_xp_pers<NTv, false, false>), dim3(blocks), dim3(512), 0, d->stream, XPP_ARGS);              \
            } while (0)
            switch (w) {
                case 5: XPP_LAUNCH(5); break;
                case 4: XPP_LAUNCH(4); break;
                case 3: XPP_LAUNCH(3); break;
                case 2: XPP_LAUNCH(2); break;
                default: XPP_LAUNCH(1); break;
            }
#undef XPP_LAUNCH
#undef XPP_ARGS
            launch_check();
            nt0 += w;
        }
        ts.stop();
        if (sumsq) {
            hipLaunchKernelGGL(k_add_scalar_parts, dim3(1), dim3(256), 0, d->stream, ssp, (int64_t)blocks * 8, sumsq);
            launch_check();
            dev_free(d, ssp);
        }
        dev_free(d, Ppk);
        return;
    }
    {
        // classic form: 64-row wave tiles, two waves per SIMD, column panels of <= 5 tiles
        const int blocks = cdiv(n, 256);
        double* ssp = sumsq ? (double*)dev_alloc(d, sizeof(double) * blocks * 4) : nullptr;
        TagScope ts(d);
        for (int nt0 = 0; nt0 < NTtot;) {
            const int rem = NTtot - nt0;
            const int w = rem >= 5 ? 5 : rem;
            double* sp = nt0 == 0 ? ssp : nullptr;
            const float* Xf = (const float*)X; const float* muf = (const float*)mu; const float* bf = (const float*)bias; float* Zf = (float*)Z;
            switch (w) {
                case 5: launch_xp<4, 5>(d, Xf, n, (int)K, ldx, muf, Ppk, NTtot, nt0, (int)N, bf, Zf, ldz, sp, blocks); break;
                case 4: launch_xp<4, 4>(d, Xf, n, (int)K, ldx, muf, Ppk, NTtot, nt0, (int)N, bf, Zf, ldz, sp, blocks); break;
                case 3: launch_xp<4, 3>(d, Xf, n, (int)K, ldx, muf, Ppk, NTtot, nt0, (int)N, bf, Zf, ldz, sp, blocks); break;
                case 2: launch_xp<4, 2>(d, Xf, n, (int)K, ldx, muf, Ppk, NTtot, nt0, (int)N, bf, Zf, ldz, sp, blocks); break;
                default: launch_xp<4, 1>(d, Xf, n, (int)K, ldx, muf, Ppk, NTtot, nt0, (int)N, bf, Zf, ldz, sp, blocks); break;
            }
            nt0 += w;
        }
        ts.stop();
        if (sumsq) {
            hipLaunchKernelGGL(k_add_scalar_parts, dim3(1), dim3(256), 0, d->stream, ssp, (int64_t)blocks * 4, sumsq);
            launch_check();
            dev_free(d, ssp);
        }
        dev_free(d, Ppk);
        return;
    }
}

template <int NT>
static void launch_atb(Dev* d, const float* A, int64_t lda, int M, const float* muA, const float* B, int64_t ldb, int N,
                       int n0col, const float* muB, int64_t n, int64_t chunk, float* part, int nsplit) {
    const dim3 grid(8 * cdiv(M, 256), cdiv(nsplit, 8)), block(256);
#define ATB_ARGS A, lda, M, muA, B, ldb, N, n0col, muB, n, chunk, part, N
    if (muA && muB) hipLaunchKernelGGL((k_atb_mfma<NT, true, true>), grid, block, 0, d->stream, ATB_ARGS);
    else if (muA) hipLaunchKernelGGL((k_atb_mfma<NT, true, false>), grid, block, 0, d->stream, ATB_ARGS);
    else if (muB) hipLaunchKernelGGL((k_atb_mfma<NT, false, true>), grid, block, 0, d->stream, ATB_ARGS);
    else hipLaunchKernelGGL((k_atb_mfma<NT, false, false>), grid, block, 0, d->stream, ATB_ARGS);
#undef ATB_ARGS
    launch_check();
}

void op_gemm_atb(Dev* d, int dt, const void* A, int64_t lda, int64_t M, const void* muA, const void* B, int64_t ldb, int64_t N,
                 const void* muB, int64_t n, double* C, int64_t ldc, bool precise, bool steering) {
    if (M == 0 || N == 0) return;
    const bool p4 = steering && !precise && opt_on(d, OPT_STEERING);   // (a steering pass: both operands on two planes; the wide form only)
    if (n == 0) { HIP_CHECK(hipMemset2DAsync(C, ldc * sizeof(double), 0, N * sizeof(double), M, d->stream)); return; }
    const bool mfma = !precise && dt == F32 && M % 16 == 0 && N % 16 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned16(A) && aligned16(B) &&
                      (!muA || aligned16(muA)) && (!muB || aligned16(muB)) && n >= 64 && M < (1 << 24) && N < (1 << 24);
    // fp64 matrix cores: the precise Gram of fp32 data, and every product of fp64 data (8-byte elements: rows 16-B aligned
    // with an even leading dimension)
    const bool mfma64 = ((precise && dt == F32 && lda % 4 == 0 && ldb % 4 == 0) || (dt == F64 && lda % 2 == 0 && ldb % 2 == 0)) &&
                        M % 16 == 0 && N % 16 == 0 && aligned16(A) && aligned16(B) && (!muA || aligned16(muA)) &&
                        (!muB || aligned16(muB)) && n >= 64 && M < (1 << 24) && N < (1 << 24);
    if (mfma64) {
        const int sym = (A == B && muA == muB && lda == ldb && M == N) ? 1 : 0;  // Gram matrix: upper tiles only, then mirror
        const bool ne5 = !sym && N % 80 == 0;  // l = 74 -> 80 columns: one 5-tile panel instead of two 4-tile ones
        const int mslices = cdiv(M, 32), npanels = ne5 ? (int)(N / 80) : cdiv(N, 64);
        const int gx = cdiv(M, 128);
        int active = gx * npanels;  // workgroups per row chunk
        if (sym) {   // live 32 x 64 wave tiles, four to a workgroup
            int tiles = 0;
            for (int a = 0; a < mslices; ++a) tiles += npanels - std::min(npanels, a / 2);
            active = cdiv(tiles, 4);
        }
        // row split: workgroups are dealt round-robin to the 256 CUs (up to three resident on each: the kernel's register
        // budget), so the launch takes ceil(active * ns / 256) / ns of the single-split time t1 -- one workgroup's four 32 x 64
        // tiles over all n rows, ~0.1 us a row (20000 x 256 in 15 splits: 119 us) -- plus what the ns fp64 slabs cost to write and
        // to combine (M N 16 bytes each at ~4 TB/s); pick the ns that minimises the sum.  (Round 4: a hard cap "slabs <= 40 % of
        // the input traffic" stood here and left 20000 x 256 on 75 workgroups: 119 us where 250 take half that.)
        const int64_t ns_max = std::min<int64_t>(128, std::max<int64_t>(1, n / 64));
        const double t1 = 0.1 * double(n), slab_us = double(M) * double(N) * 16.0 / 4.0e6;
        auto launch_us = [&](int64_t ns) { return t1 * double((active * ns + 255) / 256) / double(ns) * (active * ns <= 256 ? 1.25 : 1.0); };
        // (a single workgroup per CU leaves one wave per SIMD and the load latency exposed: measured 312 vs 265 us)
        int64_t nsplit = 1;
        double best = 1e30;
        for (int64_t ns = 1; ns <= ns_max; ++ns) {
            const double cost = launch_us(ns) + slab_us * double(ns);
            if (cost < best * 0.97) { best = cost; nsplit = ns; }
        }
        // among the splits that tie with the best one, the finest up to 96: whole "rounds" of workgroups cost the same in this
        // model, but a finer split keeps three workgroups on a CU and shortens the tail of the launch (500000 x 512: 2909 us at
        // 28 splits, 2863 at 42, 2837 at 56, 2757 at 84 -- against 0.4 us of k_sum_parts2 per extra slab) -- while the slabs stay
        // small change (3 % of the launch)
        {
            const double base = launch_us(nsplit);
            for (int64_t ns = nsplit + 1; ns <= std::min<int64_t>(ns_max, 96); ++ns)
                if (launch_us(ns) <= base * 1.005 && slab_us * double(ns) <= 0.03 * base) nsplit = ns;
        }
        (void)mslices;
        const int64_t chunk = ((n + nsplit - 1) / nsplit + 15) / 16 * 16;
        nsplit = (n + chunk - 1) / chunk;
        double* part = (double*)dev_alloc(d, sizeof(double) * nsplit * M * N);
        const dim3 grid(sym ? active : gx, sym ? 1 : npanels, (unsigned)nsplit), block(256);
        TagScope ts(d);
#define ATB64_LAUNCH(TT, NEv)                                                                                                     \
        do {                                                                                                                      \
            const TT* Af = (const TT*)A; const TT* Bf = (const TT*)B; const TT* ma = (const TT*)muA; const TT* mb = (const TT*)muB; \
            if (ma && mb) hipLaunchKernelGGL((k_atb_f64<TT, true, true, NEv>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, mb, n, chunk, part, sym); \
            else if (ma) hipLaunchKernelGGL((k_atb_f64<TT, true, false, NEv>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, mb, n, chunk, part, sym); \
            else if (mb) hipLaunchKernelGGL((k_atb_f64<TT, false, true, NEv>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, mb, n, chunk, part, sym); \
            else hipLaunchKernelGGL((k_atb_f64<TT, false, false, NEv>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, mb, n, chunk, part, sym); \
        } while (0)
        if (dt == F64) { if (ne5) ATB64_LAUNCH(double, 5); else ATB64_LAUNCH(double, 4); }
        else { if (ne5) ATB64_LAUNCH(float, 5); else ATB64_LAUNCH(float, 4); }
#undef ATB64_LAUNCH
        launch_check();
        ts.stop();
        hipLaunchKernelGGL(k_sum_parts2<double>, dim3(cdiv(M * N, 32)), dim3(256), 0, d->stream, part, nsplit, M * N, C, N, ldc, false);
        launch_check();
        if (sym) {
            hipLaunchKernelGGL(k_mirror_upper, dim3(cdiv(M * M, 256)), dim3(256), 0, d->stream, C, M, ldc);
            launch_check();
        }
        dev_free(d, part);
        return;
    }
    if (!mfma) {
        const int64_t nparts = cdiv(n, ATB_S_ROWS);
        double* part = (double*)dev_alloc(d, sizeof(double) * nparts * M * N);
        TagScope ts(d);
        DISPATCH_T(dt, hipLaunchKernelGGL(k_atb_simple<T>, dim3((unsigned)nparts, cdiv(N, 16), cdiv(M, 16)), dim3(16, 16), 0,
                                          d->stream, (const T*)A, lda, M, (const T*)muA, (const T*)B, ldb, N, (const T*)muB, n, part));
        launch_check();
        ts.stop();
        hipLaunchKernelGGL(k_sum_parts2<double>, dim3(cdiv(M * N, 32)), dim3(256), 0, d->stream, part, nparts, M * N, C, N, ldc, false);
        launch_check();
        dev_free(d, part);
        return;
    }
    // split the rows so that every SIMD gets one wave (fp32-MFMA kernel: MFMA-bound, 256 workgroups) or two (split-product
    // kernel: paced by loads in flight -- 512 workgroups = two per CU measured 67 vs 75 us at 100000 x 512)
    const int mslices = cdiv(M, 64);
    constexpr int waves_env = 0;
    const int num_cu2 = num_cus(d);
    // (at 100000 rows the extra 128 slabs cost k_sum_parts2 what the kernel gains: two per CU only for long row ranges)
    const int waves_target = waves_env > 0 ? waves_env : num_cu2 * ((gemm_split_product(d) && !muB && n >= 400000) ? 8 : 4);
    int64_t nsplit = std::max<int64_t>(1, waves_target / mslices);
    // A NARROW A (at most 128 columns) takes smaller workgroups -- four waves of 32 columns, two for 64 columns -- instead of eight
    // waves of which most would re-load and re-multiply the clamped last columns (1e6 x 64: 2.3 TB/s); more row chunks keep the
    // chip filled with them (their slabs are small)
    const int NTall = int(N / 16);
    const int narrow_wv = (gemm_split_product(d) && !muB)
                              ? ((M <= 64 && NTall <= 4) ? 2 : ((M <= 128 && NTall <= 8) ? 4 : 0)) : 0;
    nsplit = std::min<int64_t>(nsplit, narrow_wv ? 256 * (8 / narrow_wv) : 256);  // bounds the partial-slab traffic of narrow (Gram) products
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, n / 64));
    const bool split3_mode = gemm_split_product(d) && !muB && n >= 32 * nsplit;
    const int64_t cq = split3_mode ? 32 : 16;  // rows per pipeline stage
    const int64_t n_main = split3_mode ? n / cq * cq : n, n_tail = n - n_main;  // split-product kernels: whole stages
    const int64_t chunk = ((n_main + nsplit - 1) / nsplit + cq - 1) / cq * cq;
    nsplit = (n_main + chunk - 1) / chunk;
    const int64_t nslab = nsplit;   // (the split-product kernels take the ragged rows along in their last row chunk)
    float* part = (float*)dev_alloc(d, sizeof(float) * nslab * M * N);
    const int NTtot = int(N / 16);
    TagScope ts(d);
    // split-product kernels: as few column panels (passes over A) as 9-tile panels allow, the tiles spread evenly; a panel of
    // 6 .. 9 tiles runs on waves that own 32 columns of A (two m-tiles, 8-B loads: 72 accumulator registers at 9 tiles)
    const int npass3 = cdiv(NTtot, 9);
    for (int nt0 = 0, pass = 0; nt0 < NTtot; ++pass) {
        const int rem = NTtot - nt0;
        const int w = split3_mode ? (rem + (npass3 - pass) - 1) / std::max(npass3 - pass, 1) : (rem >= 5 ? 5 : rem);
        const float* Af = (const float*)A; const float* Bf = (const float*)B;
        const float* ma = (const float*)muA; const float* mb = (const float*)muB;
        // 8-wave workgroups of 64-column waves (one B stage per 512 columns of A) only where the row split already puts two waves
        // on every SIMD (long row ranges); everywhere else, and for every panel of more than 5 tiles, 8 waves of 32 columns:
        // at 100000 x 512 that form takes 55.4 us where four 64-column waves took 57.4 (at 1e6 rows it is the slower one,
        // 0.544 vs 0.533 ms)
        const bool wv8 = M >= 512 && nsplit * mslices >= (int64_t)num_cu2 * 8;
        if (split3_mode && narrow_wv) {
            const dim3 grid(8 * cdiv(M, 32 * narrow_wv), (unsigned)cdiv(nsplit, 8)), block(64 * narrow_wv);
#define ATB3N_LAUNCH(NTv, WVv)                                                                                                        \
            do {                                                                                                                      \
                if (ma) hipLaunchKernelGGL((k_atb3<NTv, true, WVv, 2>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
                else hipLaunchKernelGGL((k_atb3<NTv, false, WVv, 2>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
            } while (0)
            if (narrow_wv == 2) {
                switch (w) {
                    case 4: ATB3N_LAUNCH(4, 2); break;
                    case 3: ATB3N_LAUNCH(3, 2); break;
                    case 2: ATB3N_LAUNCH(2, 2); break;
                    default: ATB3N_LAUNCH(1, 2); break;
                }
            } else {
                switch (w) {
                    case 8: ATB3N_LAUNCH(8, 4); break;
                    case 7: ATB3N_LAUNCH(7, 4); break;
                    case 6: ATB3N_LAUNCH(6, 4); break;
                    case 5: ATB3N_LAUNCH(5, 4); break;
                    case 4: ATB3N_LAUNCH(4, 4); break;
                    case 3: ATB3N_LAUNCH(3, 4); break;
                    case 2: ATB3N_LAUNCH(2, 4); break;
                    default: ATB3N_LAUNCH(1, 4); break;
                }
            }
#undef ATB3N_LAUNCH
            launch_check();
            nt0 += w;
            continue;
        }
        if (split3_mode && (w >= 6 || !wv8)) {
            const dim3 grid(8 * cdiv(M, 256), (unsigned)cdiv(nsplit, 8)), block(512);   // 8 waves x 32 columns per workgroup
#define ATB3W_LAUNCH(NTv)                                                                                                             \
            do {                                                                                                                      \
                if (p4 && ma) hipLaunchKernelGGL((k_atb3<NTv, true, 8, 2, true>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
                else if (p4) hipLaunchKernelGGL((k_atb3<NTv, false, 8, 2, true>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
                else if (ma) hipLaunchKernelGGL((k_atb3<NTv, true, 8, 2>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
                else hipLaunchKernelGGL((k_atb3<NTv, false, 8, 2>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
            } while (0)
            switch (w) {
                case 9: ATB3W_LAUNCH(9); break;
                case 8: ATB3W_LAUNCH(8); break;
                case 7: ATB3W_LAUNCH(7); break;
                case 6: ATB3W_LAUNCH(6); break;
                case 5: ATB3W_LAUNCH(5); break;
                case 4: ATB3W_LAUNCH(4); break;
                case 3: ATB3W_LAUNCH(3); break;
                case 2: ATB3W_LAUNCH(2); break;
                default: ATB3W_LAUNCH(1); break;
            }
#undef ATB3W_LAUNCH
            launch_check();
            nt0 += w;
            continue;
        }
        if (split3_mode) {
            const dim3 grid(8 * cdiv(M, 512), (unsigned)cdiv(nsplit, 8)), block(512);
#define ATB3_LAUNCH(NTv)                                                                                                              \
            do {                                                                                                                      \
                if (ma) hipLaunchKernelGGL((k_atb3<NTv, true, 8>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
                else hipLaunchKernelGGL((k_atb3<NTv, false, 8>), grid, block, 0, d->stream, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, n_main, chunk, part, (int)N, (int)n_tail); \
            } while (0)
            switch (w) {
                case 5: ATB3_LAUNCH(5); break;
                case 4: ATB3_LAUNCH(4); break;
                case 3: ATB3_LAUNCH(3); break;
                case 2: ATB3_LAUNCH(2); break;
                default: ATB3_LAUNCH(1); break;
            }
#undef ATB3_LAUNCH
            launch_check();
        }
        if (split3_mode) {
            nt0 += w;
            continue;
        }
        switch (w) {
            case 5: launch_atb<5>(d, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, mb, n, chunk, part, (int)nsplit); break;
            case 4: launch_atb<4>(d, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, mb, n, chunk, part, (int)nsplit); break;
            case 3: launch_atb<3>(d, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, mb, n, chunk, part, (int)nsplit); break;
            case 2: launch_atb<2>(d, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, mb, n, chunk, part, (int)nsplit); break;
            default: launch_atb<1>(d, Af, lda, (int)M, ma, Bf, ldb, (int)N, 16 * nt0, mb, n, chunk, part, (int)nsplit); break;
        }
        nt0 += w;
    }
    ts.stop();
    if ((M * N) % 2 == 0 && N % 2 == 0)
        hipLaunchKernelGGL(k_sum_parts4, dim3(cdiv(M * N, 128)), dim3(256), 0, d->stream, part, nslab, M * N, C, N, ldc);
    else
        hipLaunchKernelGGL(k_sum_parts2<float>, dim3(cdiv(M * N, 32)), dim3(256), 0, d->stream, part, nslab, M * N, C, N, ldc, false);
    launch_check();
    dev_free(d, part);
}

__global__ void k_flip_key(const double* __restrict__ t, double* __restrict__ key, int64_t L, const int* __restrict__ flag) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == L && flag) key[L] = flag[0] != 0 ? 2.0 : (flag[1] != 0 ? 1.0 : 0.0);   // (MAX over the ranks: the stronger redo wins)
    if (j >= L) return;
    const double a = t[j] < 0 ? 0.0 : t[j];
    unsigned long long bits = (unsigned long long)__double_as_longlong(a);
    const double rw = t[L + j];
    const unsigned long long row = rw < (double)(1ll << 28) ? (unsigned long long)rw : (1ull << 28) - 1;
    const unsigned long long payload = ((((1ull << 28) - 1) - row) << 1) | (t[2 * L + j] < 0 ? 1ull : 0ull);
    bits = (bits & ~((1ull << 29) - 1)) | (t[j] < 0 ? 0ull : payload);
    key[j] = __longlong_as_double((long long)bits);
}
void op_flip_key(Dev* d, const double* triple, double* key, int64_t L, const int* flag) {
    if (L == 0 && !flag) return;
    hipLaunchKernelGGL(k_flip_key, dim3(cdiv(L + 1, 256)), dim3(256), 0, d->stream, triple, key, L, flag);
    launch_check();
}
// Was the two-plane (16-bit) rounding of the sketch matrix and of the re-based iterates harmless for THIS spectrum?  (ops.h)
__global__ __launch_bounds__(256) void k_tail_verdict(const double* __restrict__ lam, int L, int k, const double* __restrict__ mu_sq, int dp,
                                                       int d, double n_total, const double* __restrict__ tvp, double eps2, double thr,
                                                       int* __restrict__ flag2) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0;
    if (mu_sq) for (int j = tid; j < d; j += 256) s += fmax(0.0, mu_sq[dp + j] - n_total * mu_sq[j] * mu_sq[j]);
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    const double tv = mu_sq ? red[0] : tvp[0];
    __syncthreads();
    s = 0;
    for (int j = tid; j < L; j += 256) s += fmax(lam[j], 0.0);
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    const double tail = fmax(tv - red[0], 0.0);
    const double T = sqrt(fmax(lam[L - 1], 0.0) * tail / (double)d);
    int bad = 0;
    for (int j = tid; j < k && j < L; j += 256) {
        const double lj = lam[j];
        if (!(lj > 0.0)) continue;                       // (a sigma = 0 component of rank-deficient data: nothing to perturb)
        double gap = j + 1 < L ? lj - fmax(lam[j + 1], 0.0) : lj;
        if (j > 0) gap = fmin(gap, lam[j - 1] - lj);
        const double relgap = fmax(gap / lj, 1e-3);
        if (eps2 * T / lj / relgap > thr) bad = 1;
    }
    if (__syncthreads_or(bad) && tid == 0) flag2[1] = 1;
}
void op_tail_verdict(Dev* d, const double* lam, int64_t L, int64_t k, const double* mu_sq, int64_t dp, int64_t dd, double n_total,
                     const double* tv, double eps2, double thr, int* flag2) {
    if (L <= 0 || k <= 0) return;
    hipLaunchKernelGGL(k_tail_verdict, dim3(1), dim3(256), 0, d->stream, lam, (int)L, (int)k, mu_sq, (int)dp, (int)dd, n_total, tv, eps2, thr, flag2);
    launch_check();
}
void op_col_absmax(Dev* d, int dt, const void* U, int64_t n, int64_t L, int64_t ldu, int64_t row_offset, double* absmax,
                   double* idx, double* sign) {
    if (L == 0) return;
    const int64_t rows = scan_rows_per_block(n, L), nparts = std::max<int64_t>(1, cdiv(n, rows));
    double* part = (double*)dev_alloc(d, sizeof(double) * 3 * nparts * L);
    double *pm = part, *pi = part + nparts * L, *ps = part + 2 * nparts * L;
    if (n == 0) {
        std::vector<double> h(3 * L);
        for (int64_t j = 0; j < L; ++j) { h[j] = -1.0; h[L + j] = INFINITY; h[2 * L + j] = 1.0; }
        dev_h2d(d, absmax, h.data(), sizeof(double) * L);
        dev_h2d(d, idx, h.data() + L, sizeof(double) * L);
        dev_h2d(d, sign, h.data() + 2 * L, sizeof(double) * L);
        dev_free(d, part);
        return;
    }
    DISPATCH_T(dt, hipLaunchKernelGGL(k_absmax_part2<T>, dim3((unsigned)nparts, cdiv(L, 64)), dim3(256), 0, d->stream, (const T*)U,
                                      n, L, ldu, rows, pm, pi, ps));
    launch_check();
    hipLaunchKernelGGL(k_absmax_final, dim3((unsigned)L), dim3(64), 0, d->stream, pm, pi, ps, nparts, L, row_offset, absmax, idx, sign);
    launch_check();
    dev_free(d, part);
}

void op_scale_cols(Dev* d, int dt, void* A, int64_t n, int64_t L, int64_t lda, const double* s) {
    if (n == 0 || L == 0) return;
    DISPATCH_T(dt, hipLaunchKernelGGL(k_scale_cols<T>, dim3((unsigned)n, cdiv(L, 64)), dim3(64), 0, d->stream, (T*)A, n, L, lda, s));
    launch_check();
}
void op_logcosh_rows(Dev* d, int dt, const void* X, int64_t r, int64_t c, int64_t ldx, void* G, int64_t ldg, double* gp) {
    if (r == 0) return;
    DISPATCH_T(dt, hipLaunchKernelGGL(k_logcosh_rows<T>, dim3(cdiv(r, 64)), dim3(64), 0, d->stream, (const T*)X, r, c, ldx, (T*)G, ldg, gp));
    launch_check();
}

// once per fixed-point loop: X1 is constant over its iterations, so its bf16 planes are made here (fp32 data, split-product modes,
// 32 or 64 padded components) and every op_ica_step of the loop reads them instead of splitting X1 twice per iteration
void op_ica_prepare(Dev* d, int dt, const void* X1T, int64_t n, int64_t nc, int64_t ld) {
    constexpr bool off = false;
    d->ica_x1pl_for = nullptr;
    const int NT = int((nc + 15) / 16);
    if (off || dt != F32 || !gemm_split_product(d) || (NT != 2 && NT != 4) || n < 256 || ld % 4 != 0 || ld < 16 * NT || !aligned16(X1T)) return;
    const int KCH = NT / 2;
    const int64_t nfrag = cdiv(n, 32) * 2 * KCH;
    const size_t need = sizeof(bf16x8) * (size_t)nfrag * 3 * 64;
    if (!d->ica_x1pl || d->ica_x1pl_bytes < need) {
        if (d->ica_x1pl) dev_free(d, d->ica_x1pl);
        d->ica_x1pl = dev_alloc(d, need);
        d->ica_x1pl_bytes = need;
    }
    if (KCH == 1) hipLaunchKernelGGL(k_ica_planes<1>, dim3((unsigned)cdiv(nfrag, 4)), dim3(256), 0, d->stream, (const float*)X1T, n, ld, 16 * NT, (bf16x8*)d->ica_x1pl, nfrag);
    else hipLaunchKernelGGL(k_ica_planes<2>, dim3((unsigned)cdiv(nfrag, 4)), dim3(256), 0, d->stream, (const float*)X1T, n, ld, 16 * NT, (bf16x8*)d->ica_x1pl, nfrag);
    launch_check();
    d->ica_x1pl_for = X1T; d->ica_x1pl_n = n; d->ica_x1pl_ld = ld;
}
void op_ica_step(Dev* d, int dt, const void* X1T, int64_t n, int64_t nc, int64_t ld, const double* W, double* GX_gp,
                 const int* state) {
    const int64_t cnt = nc * nc + nc;
    const bool planes_current = d->ica_wpk3_valid && d->ica_wpk3_for == W && d->ica_wpk3_nc == nc;
    d->ica_ortho_tol2 = dt == F32 ? 1e-14 : 1e-26;
    d->ica_wpk3_for = nullptr;  // (only the split-product path below asks the tail kernel to keep the planes current)
    if (n == 0) { dev_memset(d, GX_gp, 0, sizeof(double) * cnt); return; }
    const bool mfma = dt == F32 && nc <= 64 && ld % 4 == 0 && ld >= (nc + 15) / 16 * 16 && aligned16(X1T) && n >= 256;
    if (nc > 64) {
        // more than 64 components (the crate's default is min(n, d) of them, src/ica.rs:173): the step as it stands in the
        // reference -- S = X1^T W^T (ica.rs:332), g = tanh (ica.rs:388-396), G^T X1 (ica.rs:333) -- on the two X-streaming
        // GEMM kernels, with g' from the column sums of g^2.  Unfused: S makes one round trip through HBM.
        const int64_t ncp = (nc + 15) / 16 * 16;
        if (ld < ncp) throw std::runtime_error("ica_step: leading dimension below the padded component count");
        double* WT = (double*)dev_alloc(d, sizeof(double) * ncp * ncp);
        hipLaunchKernelGGL(k_transpose_pad, dim3(cdiv(ncp * ncp, 256)), dim3(256), 0, d->stream, W, nc, WT, ncp);
        launch_check();
        void* S = dev_alloc(d, dtype_size(dt) * (size_t)n * ncp);
        op_gemm_xp(d, dt, X1T, n, ncp, ld, nullptr, WT, ncp, ncp, nullptr, S, ncp, nullptr);
        const int saved_tag = d->tag;
        d->tag = TAG_NONE;  // (only the first product is bracketed as "the step kernel")
        DISPATCH_T(dt, hipLaunchKernelGGL(k_tanh_inplace<T>, dim3(cdiv((int64_t)n * ncp, 256)), dim3(256), 0, d->stream, (T*)S, (int64_t)n * ncp));
        launch_check();
        double* cs = (double*)dev_alloc(d, sizeof(double) * 2 * ncp);
        op_colsum(d, dt, S, n, ncp, ncp, cs, true);
        double* GXp = (double*)dev_alloc(d, sizeof(double) * ncp * ncp);
        op_gemm_atb(d, dt, S, ncp, ncp, nullptr, X1T, ld, ncp, nullptr, n, GXp, ncp);
        hipLaunchKernelGGL(k_ica_big_out, dim3(cdiv(cnt, 256)), dim3(256), 0, d->stream, GXp, cs + ncp, (double)n, nc, ncp, GX_gp, state);
        launch_check();
        d->tag = saved_tag;
        dev_free(d, GXp); dev_free(d, cs); dev_free(d, S); dev_free(d, WT);
        return;
    }
    if (!mfma) {
        const int64_t nparts = cdiv(n, 64);
        double* part = (double*)dev_alloc(d, sizeof(double) * nparts * cnt);
        TagScope ts(d);
        DISPATCH_T(dt, hipLaunchKernelGGL(k_ica_simple<T>, dim3((unsigned)nparts), dim3(256), sizeof(double) * nc * 64, d->stream,
                                          (const T*)X1T, n, (int)nc, ld, W, part, state));
        launch_check();
        ts.stop();
        hipLaunchKernelGGL(k_sum_parts_state, dim3(cdiv(cnt, 256)), dim3(256), 0, d->stream, part, nparts, cnt, GX_gp, state);
        launch_check();
        dev_free(d, part);
        return;
    }
    const int NT = int((nc + 15) / 16), NCP = 16 * NT;
    const int64_t slab = (int64_t)NCP * NCP + NCP;
    if (gemm_split_product(d)) {
        const int KCH = (NCP + 31) / 32;
        const size_t need = sizeof(bf16x8) * KCH * NT * 192;
        if (!d->ica_wpk3 || d->ica_wpk3_bytes < need) {
            if (d->ica_wpk3) dev_free(d, d->ica_wpk3);
            d->ica_wpk3 = dev_alloc(d, need);
            d->ica_wpk3_bytes = need;
            d->ica_wpk3_valid = false;
        }
        bf16x8* Wpk3 = (bf16x8*)d->ica_wpk3;
        if (!planes_current) {
            hipLaunchKernelGGL(k_pack_w3, dim3(cdiv(KCH * NT * 64, 256)), dim3(256), 0, d->stream, W, (int)nc, Wpk3, NT, KCH);
            launch_check();
        }
        d->ica_wpk3_for = W; d->ica_wpk3_nc = nc; d->ica_wpk3_valid = false;  // valid again once the tail has refreshed them
        const int64_t nblk = (n + 31) / 32;
        int64_t waves = std::min<int64_t>(2048, nblk);
        const int64_t bpw = (nblk + waves - 1) / waves;
        waves = (nblk + bpw - 1) / bpw;
        const int blocks = cdiv(waves, 4);
        float* part = (float*)dev_alloc(d, sizeof(float) * blocks * slab);
        TagScope ts(d);
        const bool pre = d->ica_x1pl_for == X1T && d->ica_x1pl_n == n && d->ica_x1pl_ld == ld && (NT == 2 || NT == 4);
        if (pre) {
            if (NT == 2) hipLaunchKernelGGL(k_ica3p<2>, dim3(blocks), dim3(256), 0, d->stream, (const bf16x8*)d->ica_x1pl, n, Wpk3, bpw, part, state);
            else hipLaunchKernelGGL(k_ica3p<4>, dim3(blocks), dim3(256), 0, d->stream, (const bf16x8*)d->ica_x1pl, n, Wpk3, bpw, part, state);
        } else
        switch (NT) {
            case 1: hipLaunchKernelGGL(k_ica3<1>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk3, bpw, part, state); break;
            case 2: hipLaunchKernelGGL(k_ica3<2>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk3, bpw, part, state); break;
            case 3: hipLaunchKernelGGL(k_ica3<3>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk3, bpw, part, state); break;
            default: hipLaunchKernelGGL(k_ica3<4>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk3, bpw, part, state); break;
        }
        launch_check();
        ts.stop();
        hipLaunchKernelGGL(k_ica_reduce, dim3(cdiv(cnt, 32)), dim3(1024), 0, d->stream, part, (int64_t)blocks, NCP, (int)nc, GX_gp, state);
        launch_check();
        dev_free(d, part);
        return;
    }
    float* Wpk = (float*)dev_alloc(d, sizeof(float) * NT * NT * 64 * 4);
    hipLaunchKernelGGL(k_pack_w, dim3(cdiv(NT * NT * 64, 256)), dim3(256), 0, d->stream, W, (int)nc, Wpk, NT);
    launch_check();
    const int64_t tiles = (n + 15) / 16;
    int64_t waves = std::min<int64_t>(2048, tiles);
    const int64_t tpw = (tiles + waves - 1) / waves;
    waves = (tiles + tpw - 1) / tpw;
    const int blocks = cdiv(waves, 4);
    const int64_t nparts = (int64_t)blocks;  // one slab per workgroup
    float* part = (float*)dev_alloc(d, sizeof(float) * nparts * slab);
    TagScope ts(d);
    switch (NT) {
        case 1: hipLaunchKernelGGL(k_ica_mfma<1>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk, tpw, part, state); break;
        case 2: hipLaunchKernelGGL(k_ica_mfma<2>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk, tpw, part, state); break;
        case 3: hipLaunchKernelGGL(k_ica_mfma<3>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk, tpw, part, state); break;
        default: hipLaunchKernelGGL(k_ica_mfma<4>, dim3(blocks), dim3(256), 0, d->stream, (const float*)X1T, n, ld, Wpk, tpw, part, state); break;
    }
    launch_check();
    ts.stop();
    hipLaunchKernelGGL(k_ica_reduce, dim3(cdiv(cnt, 32)), dim3(1024), 0, d->stream, part, nparts, NCP, (int)nc, GX_gp, state);
    launch_check();
    dev_free(d, part);
    dev_free(d, Wpk);
}


#define MB_DISPATCH(mb, CALL)            \
    switch (mb) {                        \
        case 0: { constexpr int MBv = 0; CALL; } break; \
        case 1: { constexpr int MBv = 1; CALL; } break; \
        case 2: { constexpr int MBv = 2; CALL; } break; \
        case 3: { constexpr int MBv = 3; CALL; } break; \
        case 4: { constexpr int MBv = 4; CALL; } break; \
        case 5: { constexpr int MBv = 5; CALL; } break; \
        default: { constexpr int MBv = 6; CALL; } break; \
    }

void op_ica_tail(Dev* d, int64_t nc, double n_total, double* W, const double* GX_gp, int mode, double tol, int* state, int iter,
                 int* progress) {
    double* scratch = (double*)dev_alloc(d, sizeof(double) * (6 * nc * nc + nc));
    // refresh the step kernel's planes of W if the last step used them for this W
    bf16x8* wpk3 = (d->ica_wpk3 && d->ica_wpk3_for == W && d->ica_wpk3_nc == nc) ? (bf16x8*)d->ica_wpk3 : nullptr;
    const int mb = nc <= 64 ? (int)((nc + 15) / 16) : 0;
    const size_t lds = sizeof(double) * (jac_ws_doubles((int)nc, ICA_TAIL_THREADS) + (mb ? 4 * nc * polar_ld((int)nc) : 0));
    MB_DISPATCH(mb, {
        set_max_lds(d, reinterpret_cast<const void*>(k_ica_tail<MBv>));
        hipLaunchKernelGGL(k_ica_tail<MBv>, dim3(1), dim3(ICA_TAIL_THREADS), lds, d->stream, (int)nc, n_total, W, GX_gp, mode, tol,
                           state, iter, scratch, wpk3, d->ica_ortho_tol2, progress);
    });
    launch_check();
    if (wpk3) d->ica_wpk3_valid = true;
    dev_free(d, scratch);
}
void op_symdecorr(Dev* d, int64_t nc, const double* Win, double* Wout, int mode, int* zero2) {
    d->ica_wpk3_valid = false;
    double* scratch = (double*)dev_alloc(d, sizeof(double) * (4 * nc * nc + nc));
    const int mb = nc <= 64 ? (int)((nc + 15) / 16) : 0;
    const size_t lds = sizeof(double) * (jac_ws_doubles((int)nc, ICA_TAIL_THREADS) + (mb ? 3 * nc * polar_ld((int)nc) : 0));
    MB_DISPATCH(mb, {
        set_max_lds(d, reinterpret_cast<const void*>(k_symdecorr<MBv>));
        hipLaunchKernelGGL(k_symdecorr<MBv>, dim3(1), dim3(ICA_TAIL_THREADS), lds, d->stream, Win, Wout, (int)nc, mode, scratch, zero2);
    });
    launch_check();
    dev_free(d, scratch);
}

// G = A^T A for a tall-skinny fp64 A (K x Mn, Mn a multiple of 16): the re-basing Gram matrices (l x l from a d x l
// iterate).  One workgroup per upper 16 x 16 tile, its four waves take a quarter of K each on the fp64 matrix cores and
// add through LDS; the mirror tile is written too.  (The general kernel needs a split-K launch plus a reduce launch here.)
// SYM = false: C = A^T B for two such matrices (every tile computed; H = P^T Yp of the one-pass thin QR).
template <bool SYM>
__global__ __launch_bounds__(256) void k_syrk_f64(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                                                  int Mn, int64_t K, double* __restrict__ C, int64_t ldc) {
    __shared__ double red[3][4][64];
    const int nt = Mn >> 4;
    int t = blockIdx.x, ti = 0;
    if (SYM) { while (t >= nt - ti) { t -= nt - ti; ++ti; } t += ti; }
    else { ti = t / nt; t -= ti * nt; }
    const int tj = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const int64_t kq = ((K + 3) / 4 + 3) / 4 * 4, kbeg = wave * kq, kend = min(K, kbeg + kq);
    const double* pa = A + 16 * ti + i;
    const double* pb = B + 16 * tj + i;
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
    // the kernel is pure load latency (a wave's 32 MFMAs take 1 us): all operands of a 128-row batch are requested at once
    for (int64_t kb = kbeg; kb < kend; kb += 128) {
        double a[32], b[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int64_t k = kb + 4 * u + q;
            const bool in = k < kend;
            a[u] = in ? pa[k * lda] : 0.0;
            b[u] = in ? pb[k * ldb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
            const int row = 16 * ti + q + 4 * r, col = 16 * tj + i;
            C[(int64_t)row * ldc + col] = v;
            if (SYM && ti != tj) C[(int64_t)col * ldc + row] = v;
        }
    }
}

// C = A B for small fp64 matrices whose shapes are whole 16 x 16 tiles (A: M x K, B: K x N, both row-major, not transposed): one
// workgroup per output tile, its four waves take a quarter of K each on the fp64 matrix cores and add through LDS -- ONE launch
// where the generic kernel needs a split-K launch and a reduce launch (the subspace iteration's C Q products: 256 x 256 x 48)
// or walks K in 16-deep LDS stages (Yp T: 512 x 80 x 80).  Like k_syrk_f64 the kernel is load latency: a wave's operands of a
// 64-deep batch are all requested before its 16 MFMAs.
__global__ __launch_bounds__(256) void k_gemm_nn_f64(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                                                     int64_t K, double* __restrict__ C, int64_t ldc, int ntn) {
    __shared__ double red[3][4][64];
    const int ti = blockIdx.x / ntn, tj = blockIdx.x - ti * ntn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const int64_t kq = ((K + 3) / 4 + 3) / 4 * 4, kbeg = wave * kq, kend = min(K, kbeg + kq);
    const double* pa = A + (int64_t)(16 * ti + i) * lda;     // A[row][k]
    const double* pb = B + 16 * tj + i;                        // B[k][col]
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int64_t kb = kbeg; kb < kend; kb += 64) {
        double a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t k = kb + 4 * u + q;
            const bool in = k < kend;
            a[u] = in ? pa[k] : 0.0;
            b[u] = in ? pb[k * ldb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            C[(int64_t)(16 * ti + q + 4 * r) * ldc + 16 * tj + i] = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    }
}

void op_dgemm(Dev* d, bool ta, bool tb, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
              const double* B, int64_t ldb, double beta, double* C, int64_t ldc, const double* colscale) {
    if (M == 0 || N == 0) return;
    if (!colscale && ta && !tb && M == N && M % 16 == 0 && M <= 256 && K >= 64 && alpha == 1.0 && beta == 0.0) {
        const int nt = (int)(M / 16);
        if (A == B && lda == ldb)
            hipLaunchKernelGGL(k_syrk_f64<true>, dim3(nt * (nt + 1) / 2), dim3(256), 0, d->stream, A, lda, B, ldb, (int)M, K, C, ldc);
        else
            hipLaunchKernelGGL(k_syrk_f64<false>, dim3(nt * nt), dim3(256), 0, d->stream, A, lda, B, ldb, (int)M, K, C, ldc);
        launch_check();
        return;
    }
    constexpr bool no_nn = false;
    if (!no_nn && !colscale && !ta && !tb && M % 16 == 0 && N % 16 == 0 && K >= 64 && K <= 4096 && alpha == 1.0 && beta == 0.0 &&
        (M / 16) * (N / 16) <= 4096) {
        hipLaunchKernelGGL(k_gemm_nn_f64, dim3((unsigned)((M / 16) * (N / 16))), dim3(256), 0, d->stream, A, lda, B, ldb, K, C, ldc, (int)(N / 16));
        launch_check();
        return;
    }
    const int64_t tiles = (int64_t)cdiv(N, 16) * cdiv(M, 16);
    int ks = 1;
    if (K >= 256 && tiles < 256) ks = (int)std::min<int64_t>(16, std::min<int64_t>(K / 64, (512 + tiles - 1) / tiles));
    if (ks > 1) {  // long reduction, few output tiles: spread K over the chip, then add the slices in order
        const int64_t kchunk = (cdiv(K, ks) + 31) / 32 * 32;
        ks = cdiv(K, kchunk);
        double* part = (double*)dev_alloc(d, sizeof(double) * ks * M * N);
        hipLaunchKernelGGL(k_dgemm, dim3(cdiv(N, 16), cdiv(M, 16), ks), dim3(16, 16), 0, d->stream, ta, tb, M, N, K, alpha, A, lda, B,
                           ldb, beta, C, ldc, kchunk, part, colscale, (bf16x8*)nullptr, 0);
        launch_check();
        hipLaunchKernelGGL(k_dgemm_reduce, dim3(cdiv(M * N, 256)), dim3(256), 0, d->stream, part, ks, M, N, alpha, beta, C, ldc, colscale);
        launch_check();
        dev_free(d, part);
        return;
    }
    hipLaunchKernelGGL(k_dgemm, dim3(cdiv(N, 16), cdiv(M, 16)), dim3(16, 16), 0, d->stream, ta, tb, M, N, K, alpha, A, lda, B, ldb,
                       beta, C, ldc, K, (double*)nullptr, colscale, (bf16x8*)nullptr, 0);
    launch_check();
}
__global__ void k_copy_diag(const double* __restrict__ G, int64_t ldg, int64_t L, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) out[j] = G[j * ldg + j];
}
// L beyond the one-workgroup kernels (k + n_oversample > 200): blocked right-looking factorisation on the chip-wide fp64
// GEMM kernel, 128-wide diagonal blocks by k_chol_inv2 (which hands back T_JJ = R_JJ^-1 directly):
//   G_J,J.. -= sum_{K<J} R_KJ^T R_K,J..;  T_JJ = chol_inv(G_JJ);  R_J,rest = T_JJ^T G_J,rest;
//   T_IJ = -T_II sum_{K=I+1..J} R_IK T_KJ, block super-diagonal by block super-diagonal.
static void chol_inv_blocked(Dev* d, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead,
                             int64_t Lz) {
    constexpr int64_t B = 128;
    const int64_t nb = (L + B - 1) / B;
    double* W = (double*)dev_alloc(d, sizeof(double) * L * L);      // working copy of G (Schur complements of the diagonal blocks)
    double* R = (double*)dev_alloc(d, sizeof(double) * L * L);      // off-diagonal blocks of the factor
    double* gd = (double*)dev_alloc(d, sizeof(double) * L);
    double* tmp = (double*)dev_alloc(d, sizeof(double) * B * B);
    HIP_CHECK(hipMemcpy2DAsync(W, L * sizeof(double), G, ldg * sizeof(double), L * sizeof(double), L, hipMemcpyDeviceToDevice, d->stream));
    hipLaunchKernelGGL(k_copy_diag, dim3(cdiv(L, 256)), dim3(256), 0, d->stream, G, ldg, L, gd);
    launch_check();
    HIP_CHECK(hipMemset2DAsync(T, ldt * sizeof(double), 0, Lz * sizeof(double), Lz, d->stream));
    set_max_lds(d, reinterpret_cast<const void*>(k_chol_inv2));
    for (int64_t J = 0; J < nb; ++J) {
        const int64_t j0 = J * B, bj = std::min(B, L - j0), rest = L - j0 - bj;
        if (j0 > 0)  // block row J of the Schur complement (its diagonal block and everything right of it)
            op_dgemm(d, true, false, bj, L - j0, j0, -1.0, R + j0, L, R + j0, L, 1.0, W + j0 * L + j0, L);
        hipLaunchKernelGGL(k_chol_inv2, dim3(1), dim3(CHOL_THREADS), chol2_lds_bytes((int)bj), d->stream, W + j0 * L + j0, (int)bj, L,
                           T + j0 * ldt + j0, ldt, rel_tol, ndead, (int)bj, (const double*)(gd + j0), 0, (int)bj);
        launch_check();
        if (rest > 0)
            op_dgemm(d, true, false, bj, rest, bj, 1.0, T + j0 * ldt + j0, ldt, W + j0 * L + j0 + bj, L, 0.0, R + j0 * L + j0 + bj, L);
    }
    for (int64_t dl = 1; dl < nb; ++dl)
        for (int64_t I = 0; I + dl < nb; ++I) {
            const int64_t Jb = I + dl, i0 = I * B, bi = std::min(B, L - i0), j0 = Jb * B, bj = std::min(B, L - j0);
            const int64_t k0 = i0 + bi, kk = j0 + bj - k0;  // rows k0 .. j0 + bj of column block J of T are final
            op_dgemm(d, false, false, bi, bj, kk, 1.0, R + i0 * L + k0, L, T + k0 * ldt + j0, ldt, 0.0, tmp, bj);
            op_dgemm(d, false, false, bi, bj, bi, -1.0, T + i0 * ldt + i0, ldt, tmp, bj, 0.0, T + i0 * ldt + j0, ldt);
        }
    dev_free(d, tmp); dev_free(d, gd); dev_free(d, R); dev_free(d, W);
}

// the re-basing factorisation in RT form (what k_trsm_pack reads): k_chol_rt4 for M = 16 NB <= 144
static void launch_chol_rt(Dev* d, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead, int64_t M,
                           int64_t ncount = 0) {
    switch ((int)(M / 16)) {
#define PETAL_CHOL_RT_CASE(NB)                                                                                                      \
    case NB:                                                                                                                        \
        hipLaunchKernelGGL((k_chol_rt4<NB>), dim3(1), dim3(256), 0, d->stream, G, (int)L, ldg, T, ldt, rel_tol, ndead, (int)(ncount > 0 ? ncount : L)); \
        break
        PETAL_CHOL_RT_CASE(1); PETAL_CHOL_RT_CASE(2); PETAL_CHOL_RT_CASE(3); PETAL_CHOL_RT_CASE(4); PETAL_CHOL_RT_CASE(5);
        PETAL_CHOL_RT_CASE(6); PETAL_CHOL_RT_CASE(7); PETAL_CHOL_RT_CASE(8); PETAL_CHOL_RT_CASE(9);
#undef PETAL_CHOL_RT_CASE
        default: throw std::logic_error("launch_chol_rt: order out of range");
    }
    launch_check();
}

// One re-basing step of the power iteration (see ops.h).  With the split-product kernels and L <= 140 the inverse of R is never
// formed: k_chol_inv2 stops after the diagonal-block inverses ("RT form") and k_trsm_pack applies R^-1 by blocked substitution
// while it packs the operand planes of the product that follows.
void op_rebase_xp(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* G, int64_t L,
                  int64_t ldg, double rel_tol, int* ndead, const double* A, int64_t M, int64_t lda, double* T, int64_t ldt,
                  double* P_out, int64_t ldpo, void* Z, int64_t ldz, int p_planes, bool steering) {
    const bool no_rt = false;
    const bool fused = dt == F32 && gemm_split_product(d) && n >= 64 && K % 16 == 0 && K > 0 && M % 16 == 0 && ldx % 4 == 0 &&
                       aligned16(X) && (!mu || aligned16(mu)) && ldz % 4 == 0 && aligned16(Z) && K < (1 << 24) && M < (1 << 24);
    if (!fused || no_rt || L == 0 || L > CHOL2_MAXL || M > TRSM_MAXM || M < L || P_out == nullptr) {
        op_chol_inv(d, G, L, ldg, T, ldt, rel_tol, ndead, M);
        op_gemm_xp_prod(d, dt, X, n, K, ldx, mu, A, M, lda, T, M, ldt, P_out, ldpo, Z, ldz);
        return;
    }
    launch_chol_rt(d, G, L, ldg, T, ldt, rel_tol, ndead, M);
    gemm_xp_impl(d, dt, X, n, K, ldx, mu, T, M, ldt, nullptr, Z, ldz, nullptr, A, M, lda, P_out, ldpo, true, nullptr, p_planes == 2, steering);
}

// ---- split-product Gram matrix C = (X - mu)^T (X - mu) for FastICA's whitening (round 5) --------------------------------------
// The fp64-MFMA Gram (k_atb_f64) is 60 % of a FastICA fit at the configs[4] share; the whitening only keeps the top eigenpairs of a
// covariance whose wanted eigenvalues lie within a few decades, which fp32-accumulated exact products deliver (the fit checks the
// spectrum it finds and falls back to the fp64 Gram otherwise: algo.cpp).  Operand layout of a product that sums over ROWS: fragment
// order, feature-major -- plane[((b FT + ft) 3 + plane) 64 + lane][e] = plane of (X - mu)[32 b + 8 (lane >> 4) + e][16 ft + (lane & 15)].
// Round 5 built three forms (256 x 128 tiles on planes pre-split by a pass of their own, 256 x 256 tiles on those planes, 256 x 256
// tiles split on the fly); the last one, k_gram5, is the product's, the other two were retired in round 6 (EXPERIMENTS.md round 5).
static inline int64_t round_up_i64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
// ---- k_gram5: k_gram4 WITHOUT the pre-split pass.  The planes were a third of the Gram path (0.5 ms of 1.65 at 500000 x 512: 1 GB
// read, 1.5 GB written, then read 2 x) for a split that costs a workgroup ~110 VALU instructions per wave and stage beside 192 MFMAs.
// Here every wave fetches its two feature tiles of the next stage's panels as fp32 straight in fragment order (lane (i, q): feature
// 16 ft + i, samples 8 q .. + 7: eight dword loads per fragment, 64-B segments, both halves of a line by the same wave), centres and
// splits them under this stage's MFMAs and stores the planes into the LDS panels k_gram4 filled by DMA: the A panel of stage b + 1 into
// the other A buffer at any time, the B panel once the barrier behind the B-fragment reads has passed.  X is read d / 256 times
// (tiles above the diagonal read two panels, tiles on it one) and nothing is written but the slabs.
// DIAG: a tile ON the diagonal.  Its B panel is its A panel (one fetch, one park, one barrier per stage, room for all four sets of B
// fragments), and of its 256 16 x 16 sub-tiles only the 136 that reach the upper triangle are computed: the sub-tiles are dealt out
// CYCLICALLY (wave (wm, wn): row tiles wm + 2 a, column tiles wn + 4 c) and the waves paired on the SIMDs so that every SIMD gets
// 32 - 36 of them (contiguous 128 x 64 blocks would leave one SIMD with 58 of its 64 and two waves with nothing).
#ifndef PETAL_G5_NOPARK   // (timing experiment, never a product build: the diagonal tiles' MFMA / LDS-read structure alone, on stale planes --
#define PETAL_G5_NOPARK 0  //  200000 x 256: 70.5 against 90.7 us, i.e. fetch + split + park cost 22 % and the rest runs at 0.53 of the matrix pipe)
#endif
// SUMS (tiles on the diagonal of a fit whose `mu` is only a provisional centre mu0, the means of a row sample): the column sums of
// X - mu0 over the chunk come out beside the tile (every 256-feature panel is the A panel of exactly one diagonal tile), one fp32
// partial per feature and chunk; k_gram5_centre turns them into delta = sums / n and the true means, and the reduction subtracts
// n delta delta^T: C = sum (x - mu0)(x - mu0)^T - n delta delta^T, a correction of relative size (delta / sigma)^2.  The separate
// pass over X for the column means (0.18 ms at 500000 x 512) is gone.
template <bool CENTER, bool DIAG, bool SUMS>
__device__ __forceinline__ void gram5_body(unsigned char* sm_g5, const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                           const float* __restrict__ mu, int mi, int nj, int64_t b0, int64_t b1, float* __restrict__ out,
                                           float* __restrict__ sums_out) {
    constexpr int PANEL = 48 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (DIAG: the pairing (0,2)+(1,0), (0,3)+(1,1), (1,3)+(0,0), (0,1)+(1,2) of waves w, w + 4)
    const int wm = DIAG ? (180 >> wave) & 1 : wave >> 2, wn = DIAG ? (33918 >> (2 * wave)) & 3 : wave & 3;
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's two feature tiles (2 wave, 2 wave + 1) of a panel: raw fragments of one stage, and where their planes go.
    // Addresses: a wave-uniform base per stage and panel (scalar registers) + a 32-bit per-lane offset that is re-derived at every use
    // from a laundered lane index -- kept live across the stage, the sixteen 64-bit row pointers of the two panels cost 20 spilled
    // registers beside the 128 accumulators.
    const int ldxi = (int)ldx;
    float muA[2], muB[2];
    {
        const int li = lane & 15;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int fa = 256 * mi + 32 * wave + 16 * t + li, fb = 256 * nj + 32 * wave + 16 * t + li;
            muA[t] = (CENTER && fa < d) ? mu[fa] : 0.f;
            muB[t] = (CENTER && !DIAG && fb < d) ? mu[fb] : 0.f;
        }
    }
    auto fetch = [&](int64_t b, int blk, f32x8(&raw)[2]) {
        int ln = threadIdx.x & 63;
        asm volatile("" : "+v"(ln));
        const int li = ln & 15, lq = ln >> 4;
        const float* Xb = X + 32 * b * ldx;                          // (uniform)
        const int left = (int)min((int64_t)32, n - 32 * b);         // valid rows of the stage (uniform)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int f = 256 * blk + 32 * wave + 16 * t + li;
            const int o = 8 * lq * ldxi + min(f, d - 1);
            if (left == 32) {
#pragma unroll
                for (int e = 0; e < 8; ++e) raw[t][e] = Xb[o + e * ldxi];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) raw[t][e] = 8 * lq + e < left ? Xb[o + e * ldxi] : 0.f;
            }
        }
    };
    float csum[2] = {0.f, 0.f};
    auto park = [&](const f32x8(&raw)[2], int blk, const float(&m2)[2], unsigned char* panel, int64_t b, bool is_a) {
        int ln = threadIdx.x & 63;
        asm volatile("" : "+v"(ln));
        const int li = ln & 15, lq = ln >> 4;
        const int left = (int)min((int64_t)32, n - 32 * b);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x8 x = raw[t];
            const bool live = 256 * blk + 32 * wave + 16 * t + li < d;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (live && 8 * lq + e < left) ? x[e] - m2[t] : 0.f;
            if (SUMS && DIAG && is_a) csum[t] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
            bf16x8 h, m, l;
            split3(x, h, m, l);
            bf16x8* dst = reinterpret_cast<bf16x8*>(panel) + ((2 * wave + t) * 3) * 64 + ln;
            dst[0] = h; dst[64] = m; dst[128] = l;
        }
    };
    auto mfma6 = [&](f32x4& c4, const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm, const bf16x8 bl) {
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c4, 0, 0, 0);   // smallest terms first
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c4, 0, 0, 0);
    };
    unsigned char* const sBp = sm_g5 + 2 * PANEL;
    f32x8 rawA[2], rawB[2];
    fetch(b0, mi, rawA);
    if (!DIAG) fetch(b0, nj, rawB);
    park(rawA, mi, muA, sm_g5, b0, true);
    if (!DIAG) park(rawB, nj, muB, sBp, b0, false);
    if (b0 + 1 < b1) { fetch(b0 + 1, mi, rawA); if (!DIAG) fetch(b0 + 1, nj, rawB); }
    const int park_at = wave < 4 ? 1 : 5;   // (a different row tile in the two waves of a SIMD: one's ~100 VALU instructions meet the
                                            // other's MFMAs and not its VALU run -- in step, both park with the pipe idle: 2300 cycles per stage)
    for (int64_t b = b0; b < b1; ++b) {
        unsigned char* const sAp = sm_g5 + (int)((b - b0) & 1) * PANEL;
        unsigned char* const sAn = sm_g5 + (int)(((b - b0) & 1) ^ 1) * PANEL;
        __syncthreads();                                            // stage b's planes are in LDS; nobody still reads the other A buffer
        const bool more = b + 1 < b1;
        if constexpr (DIAG) {
            const bf16x8* sB = reinterpret_cast<const bf16x8*>(sAp) + lane;
            const bf16x8* sA = reinterpret_cast<const bf16x8*>(sAp) + lane;
            bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const bf16x8* q = sB + (wn + 4 * c) * 192; bh[c] = q[0]; bm[c] = q[64]; bl[c] = q[128]; }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const bf16x8* q = sA + (wm + 2 * a) * 192;
                const bf16x8 ah = q[0], am = q[64], al = q[128];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (wm + 2 * a <= wn + 4 * c) mfma6(acc[a][c], ah, am, al, bh[c], bm[c], bl[c]);   // (uniform: sub-tiles below the diagonal are skipped)
                if (a == park_at && more && !PETAL_G5_NOPARK) {
                    __builtin_amdgcn_sched_barrier(0);
                    // (... and the raw registers go straight back into flight for the stage after it: a whole stage of slack for HBM)
                    park(rawA, mi, muA, sAn, b + 1, true);
                    if (b + 2 < b1) fetch(b + 2, mi, rawA);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            const bf16x8* sB = reinterpret_cast<const bf16x8*>(sBp) + (4 * wn) * 192 + lane;
            const bf16x8* sA = reinterpret_cast<const bf16x8*>(sAp) + (8 * wm) * 192 + lane;
            bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { bh[c] = sB[c * 192]; bm[c] = sB[c * 192 + 64]; bl[c] = sB[c * 192 + 128]; }
            __syncthreads();                                        // every wave holds its B fragments: the B buffer is free
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const bf16x8 ah = sA[a * 192], am = sA[a * 192 + 64], al = sA[a * 192 + 128];
#pragma unroll
                for (int c = 0; c < 4; ++c) mfma6(acc[a][c], ah, am, al, bh[c], bm[c], bl[c]);
                // the next stage's two panels at two different row tiles, and at different ones in the two waves of a SIMD
                if (a == (park_at >> 1) && more) {                  // a = 0 / 2
                    __builtin_amdgcn_sched_barrier(0);
                    park(rawA, mi, muA, sAn, b + 1, true);
                    if (b + 2 < b1) fetch(b + 2, mi, rawA);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (a == 4 + (park_at >> 1) && more) {              // a = 4 / 6
                    __builtin_amdgcn_sched_barrier(0);
                    park(rawB, nj, muB, sBp, b + 1, false);
                    if (b + 2 < b1) fetch(b + 2, nj, rawB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if constexpr (SUMS && DIAG) {   // lanes (i, 0 .. 3) hold the four row groups' shares of feature 32 wave + 16 t + i
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = csum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) sums_out[32 * wave + 16 * t + lane] = v;
        }
    }
    // slab [256][256]: D[row = 4 q + r][col = i] of sub-tile (R, C) -> row 16 R + 4 q + r, column 16 C + i
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R = DIAG ? wm + 2 * a : 8 * wm + a;
            float* row = out + (16 * R + 4 * q + r) * 256 + i;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int C = DIAG ? wn + 4 * c : 4 * wn + c;
                if (!DIAG || R <= C) row[16 * C] = acc[a][c][r];
            }
        }
}
template <bool CENTER, bool SUMS>
__global__ __launch_bounds__(512) void k_gram5(const float* __restrict__ X, int64_t n, int d, int64_t ldx, const float* __restrict__ mu,
                                               int64_t nblocks, const int* __restrict__ tile_bpc, int ntiles, const int* __restrict__ tile_mi,
                                               const int* __restrict__ tile_nj, float* __restrict__ slab, float* __restrict__ sums) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_g5[];   // A[2][48 KB] (16 feature tiles x 3 planes), B[48 KB]
    const int tile = blockIdx.x >> 3;
    const int64_t chunk = (int64_t)blockIdx.y * 8 + (blockIdx.x & 7);
    if (tile >= ntiles) return;                                     // (uniform per workgroup)
    const int64_t blocks_per_chunk = tile_bpc[tile];                // (per tile: a tile above the diagonal costs more per stage, so its chunks are shorter)
    const int64_t b0 = chunk * blocks_per_chunk, b1 = min(nblocks, b0 + blocks_per_chunk);
    if (b0 >= b1) return;
    const int mi = tile_mi[tile], nj = tile_nj[tile];
    float* out = slab + ((int64_t)chunk * ntiles + tile) * (256 * 256);
    float* so = SUMS ? sums + ((int64_t)chunk * ntiles + tile) * 256 : nullptr;
    if (mi == nj) gram5_body<CENTER, true, SUMS>(sm_g5, X, n, d, ldx, mu, mi, nj, b0, b1, out, so);
    else gram5_body<CENTER, false, SUMS>(sm_g5, X, n, d, ldx, mu, mi, nj, b0, b1, out, so);
}
// delta[f] = (sum over the chunks of feature f's diagonal tile of the partial column sums) / n; mu64[f] = mu0[f] + delta[f]; muT = (float) mu64.
// One workgroup per 16 features, sixteen threads per feature (thread p adds the chunks p, p + 16, ...: one thread per feature walked
// 256 dependent-latency loads, 57 us at d = 256), combined in a fixed order.
__global__ __launch_bounds__(256) void k_gram5_centre(const float* __restrict__ sums, const int* __restrict__ tile_nch, int ntiles,
                                                      const int* __restrict__ tile_mi, const int* __restrict__ tile_nj, int d, int dp, double n_total,
                                                      double* __restrict__ mu64, float* __restrict__ muT, double* __restrict__ delta) {
    __shared__ double sp[16][17];
    const int fl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int f = blockIdx.x * 16 + fl;
    double sacc = 0;
    if (f < d) {
        int tile = -1;
        for (int t = 0; t < ntiles; ++t)
            if (tile_mi[t] == tile_nj[t] && tile_mi[t] == (f >> 8)) tile = t;
        const float* src = sums + (int64_t)max(tile, 0) * 256 + (f & 255);
        const int nch = tile >= 0 ? tile_nch[tile] : 0;   // (every 256-feature panel below d has its diagonal tile in the list)
#pragma unroll 4
        for (int k = part; k < nch; k += 16) sacc += (double)src[(int64_t)k * ntiles * 256];
    }
    sp[part][fl] = sacc;
    __syncthreads();
    if (part != 0 || f >= dp) return;
    double tot = 0;
#pragma unroll
    for (int p2 = 0; p2 < 16; ++p2) tot += sp[p2][fl];
    const double dl = tot / n_total;
    delta[f] = dl;
    const double m = f < d ? (double)muT[f] + dl : 0.0;
    mu64[f] = m;
    muT[f] = (float)m;
}
__global__ __launch_bounds__(256) void k_gram4_reduce(const float* __restrict__ slab, const int* __restrict__ tile_nch, int ntiles,
                                                      const int* __restrict__ tile_mi, const int* __restrict__ tile_nj, int d,
                                                      double* __restrict__ C, int64_t ldc, const double* __restrict__ delta, double n_total) {
    const int tile = blockIdx.y;
    const int64_t nchunks = tile_nch[tile];
    const int e = blockIdx.x * 256 + threadIdx.x;                   // element of the 256 x 256 tile
    const int r = e >> 8, c = e & 255;
    const int f = 256 * tile_mi[tile] + r, g = 256 * tile_nj[tile] + c;
    if (f >= d || g >= d || g < f) return;
    double sacc = 0;
    const float* src = slab + (int64_t)tile * (256 * 256) + e;
#pragma unroll 8
    for (int64_t k = 0; k < nchunks; ++k) sacc += (double)src[k * ntiles * (256 * 256)];   // (fixed order; eight loads in flight)
    if (delta) sacc -= n_total * delta[f] * delta[g];              // (the move from the provisional centre to the true one)
    C[(int64_t)f * ldc + g] = sacc;
    C[(int64_t)g * ldc + f] = sacc;
}
// C (d x d fp64, ldc; rows / columns d .. dp zero) = (X - mu)^T (X - mu), fp32 data; false: shape not covered, nothing done
// mu64_fold != NULL (single-rank fits): the column means are NOT known yet -- this call forms them too.  mu (device float[dp]) and
// mu64_fold (device double[dp]) receive the true means; on the way mu holds the provisional centre (a row sample's means).
bool op_gram_split(Dev* d, const void* X, int64_t n, int64_t dd, int64_t dp, int64_t ldx, const void* mu, double* C, int64_t ldc,
                   double* mu64_fold, double n_total) {
    if (!opt_on(d, OPT_GRAM_SPLIT) || d->gemm_mode == 1 || n < 4096 || dd < 64 || dp > 4096) return false;
    if (mu64_fold && (d->opt[OPT_MEANS_FOLD_ROWS] < 0 || !mu)) return false;
    // 256 x 256 tiles, split on the fly (k_gram5).  (The forms it replaced -- k_gram3 / k_gram4 on planes pre-split by k_presplit_t -- were
    // retired in round 6; EXPERIMENTS.md round 5 has their measurements.)
    constexpr int form = 5;
    constexpr bool wide = true;
    if (mu64_fold) {   // the provisional centre: the means of a strided sample of the rows (one small pass)
        const int64_t ns = std::min<int64_t>(n, 4096), stride = n / ns;
        op_colmean(d, F32, X, ns, dp, ldx * stride, double(ns), mu64_fold, const_cast<void*>(mu), false);
    }
    const int FT = (int)(round_up_i64(dp, 256) / 16);              // feature tiles, padded to whole 256-feature tile rows
    const int64_t nblocks = cdiv(n, 32);
    // tiles that reach the upper triangle
    std::vector<int> h;
    const int MT = FT / 16, NTl = wide ? FT / 16 : FT / 8, TN = wide ? 256 : 128;
    std::vector<int> tmi, tnj;
    for (int mi = 0; mi < MT; ++mi)
        for (int nj = 0; nj < NTl; ++nj)
            if (TN * nj + TN > 256 * mi && 256 * mi < dd && TN * nj < dd) { tmi.push_back(mi); tnj.push_back(nj); }
    const int ntiles = (int)tmi.size();
    // row chunks: one workgroup per CU (144 KB of LDS), about two rounds of them
    const int ncu = num_cus(d);
    // (whole rounds: 2 ncu / ntiles rounded DOWN -- 86 chunks x 6 tiles = 516 workgroups on 256 CUs ran a third round for four of them;
    // a chunk of at least 16 stages: a workgroup's 128 / 256 KB slab is written once and read once per chunk)
    constexpr int rounds = 2;
    // (... and WHOLE rounds where the chunks would get shorter than that: 368 chunks of one tile on 256 CUs take as long as 512)
    const int64_t min_stages = wide ? 16 : 8;
    int64_t nsplit = (rounds * (int64_t)ncu) / ntiles;
    for (int r = rounds - 1; r >= 1 && nsplit > nblocks / min_stages; --r) nsplit = (r * (int64_t)ncu) / ntiles;
    nsplit = std::max<int64_t>(1, std::min<int64_t>(nsplit, nblocks / min_stages));
    const int64_t bpc = cdiv(nblocks, nsplit);
    nsplit = cdiv(nblocks, bpc);                                    // (never more than asked for)
    // k_gram5: a tile above the diagonal computes 256 sub-tiles and fetches two panels per stage where a diagonal one computes 136 and
    // fetches one (launched alone at 500000 x 512: 390 us for the one tile above the diagonal, 215 us for each of the two on it).  Its
    // chunks are shorter -- by more than that ratio: the sweep 1.36 / 2.0 / 2.6 / 3.0 / 3.6 / 4.5 / 6 gave 1079 / 1102 / 968 / 926 / 885 /
    // 914 / 984 us for the whole kernel, i.e. the workgroups that start last should be the short diagonal ones
    std::vector<int> tbpc(ntiles, (int)bpc), tnch(ntiles, (int)nsplit);
    int64_t max_split = nsplit;
    if (form == 5 && ntiles > 1 && nsplit > 1) {
        constexpr double w_off = 3.6;
        double wsum = 0;
        for (int t = 0; t < ntiles; ++t) wsum += tmi[t] == tnj[t] ? 1.0 : w_off;
        const double total = double(nsplit) * ntiles;
        for (int t = 0; t < ntiles; ++t) {
            int64_t ns = std::max<int64_t>(1, (int64_t)(total * (tmi[t] == tnj[t] ? 1.0 : w_off) / wsum));
            ns = std::min<int64_t>(ns, std::max<int64_t>(1, nblocks / min_stages));
            tbpc[t] = (int)cdiv(nblocks, ns);
            tnch[t] = (int)cdiv(nblocks, (int64_t)tbpc[t]);
            max_split = std::max<int64_t>(max_split, tnch[t]);
        }
    }
    int* tiles_dev = (int*)dev_alloc(d, sizeof(int) * 4 * ntiles);
    h = tmi; h.insert(h.end(), tnj.begin(), tnj.end()); h.insert(h.end(), tbpc.begin(), tbpc.end()); h.insert(h.end(), tnch.begin(), tnch.end());
    dev_h2d_async(d, tiles_dev, h.data(), sizeof(int) * h.size());
    float* slab = (float*)dev_alloc(d, sizeof(float) * (size_t)max_split * ntiles * 256 * TN);
    float* sums = mu64_fold ? (float*)dev_alloc(d, sizeof(float) * (size_t)max_split * ntiles * 256) : nullptr;
    double* delta = mu64_fold ? (double*)dev_alloc(d, sizeof(double) * (size_t)dp) : nullptr;
    const dim3 grid(8 * ntiles, (unsigned)cdiv(max_split, 8));
    {
        TagScope ts(d);
        if (form == 5) {
#define PETAL_G5(CE, SU)                                                                                                                     \
    do {                                                                                                                                     \
        set_max_lds(d, reinterpret_cast<const void*>(k_gram5<CE, SU>));                                                                      \
        hipLaunchKernelGGL((k_gram5<CE, SU>), grid, dim3(512), 144 * 1024, d->stream, (const float*)X, n, (int)dd, ldx, (const float*)mu, nblocks, \
                           tiles_dev + 2 * ntiles, ntiles, tiles_dev, tiles_dev + ntiles, slab, sums);                                        \
    } while (0)
            if (mu64_fold) PETAL_G5(true, true);
            else if (mu) PETAL_G5(true, false);
            else PETAL_G5(false, false);
#undef PETAL_G5
        }
        launch_check();
        ts.stop();
    }
    HIP_CHECK(hipMemset2DAsync(C, ldc * sizeof(double), 0, dp * sizeof(double), dp, d->stream));
    if (mu64_fold) {
        hipLaunchKernelGGL(k_gram5_centre, dim3((unsigned)cdiv(dp, 16)), dim3(256), 0, d->stream, sums, tiles_dev + 3 * ntiles, ntiles, tiles_dev,
                           tiles_dev + ntiles, (int)dd, (int)dp, n_total, mu64_fold, (float*)const_cast<void*>(mu), delta);
        launch_check();
    }
    hipLaunchKernelGGL(k_gram4_reduce, dim3(256, ntiles), dim3(256), 0, d->stream, slab, tiles_dev + 3 * ntiles, ntiles, tiles_dev, tiles_dev + ntiles,
                       (int)dd, C, ldc, (const double*)delta, n_total);
    launch_check();
    dev_free(d, slab); dev_free(d, tiles_dev);
    if (sums) dev_free(d, sums);
    if (delta) dev_free(d, delta);
    return true;
}

// ---- the fused power-iteration pass (k_pow3) ------------------------------------------------------------------------------
static bool pow3_ok(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, int64_t N, const void* Z, int64_t ldz) {
    const int64_t min_rows = (int64_t)d->opt[OPT_FUSED_PASS_MIN_ROWS];
    return opt_on(d, OPT_FUSED_PASS) && dt == F32 && d->gemm_mode == 0 && K == 512 && N % 16 == 0 && N >= 16 && N <= 80 && n >= min_rows &&
           n < (int64_t(1) << 40) && ldx % 4 == 0 && aligned16(X) && (!mu || aligned16(mu)) && (!Z || (ldz >= N && aligned16(Z)));
}
// (the knobs that keep an operand at three planes also keep its product off the fused kernel, whose P is a two-plane one by construction)
static bool pow3_knob_off(const Dev* d, int opt) { return !opt_on(d, OPT_TWO_PLANE) || !opt_on(d, opt); }
bool op_power_pass_applies(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, int64_t N) {
    return pow3_ok(d, dt, X, n, K, ldx, mu, N, nullptr, 0);
}
// Ppk3: the packed planes of the K x N small operand (k_pack_p3's layout; planes 0 and 1 are read)
// ssq_parts_out (nullable; needs mu and no Z): the MEANS form -- mu is a provisional centre, Y's last column comes back as the column
// sums about it and *ssq_parts_out as a device array of *nparts_out partials of sum (x - mu)^2 (see k_pow3)
// `steering`: the pass only steers the iteration (not the last one of a fit): it may take the four-piece kernel k_pow3f
static void launch_pow3(Dev* d, const float* X, int64_t n, int64_t ldx, const float* mu, const bf16x8* Ppk3, int64_t N, float* Z, int64_t ldz,
                        double* Y, int64_t ldy, double** ssq_parts_out = nullptr, int* nparts_out = nullptr, bool steering = false) {
    const bool ssq_out = ssq_parts_out != nullptr;
    const bool no_fast = !opt_on(d, OPT_STEERING);
    if (steering && !no_fast && !Z) {
        const int NT = (int)(N / 16);
        const int64_t nstages = cdiv(n, 32);
        const int grid = (int)std::min<int64_t>(num_cus(d), nstages);
        float* part = (float*)dev_alloc(d, sizeof(float) * (size_t)grid * 512 * N);
        double* ssq_part = ssq_out ? (double*)dev_alloc(d, sizeof(double) * (size_t)grid * 8) : nullptr;
        const size_t lds = (size_t)8 * 8192 + (size_t)2 * 8 * NT * 1024 + (size_t)NT * 2048 + 2048;
#define POW3F_GO(NTv, CE, ME)                                                                                                      \
    do {                                                                                                                           \
        set_max_lds(d, reinterpret_cast<const void*>(k_pow3f<NTv, CE, ME>));                                                       \
        hipLaunchKernelGGL((k_pow3f<NTv, CE, ME>), dim3(grid), dim3(512), lds, d->stream, X, n, ldx, mu, Ppk3, NT, part, nstages, ssq_part); \
    } while (0)
#define POW3F_NT(NTv) do { if (ssq_out) POW3F_GO(NTv, true, true); else if (mu) POW3F_GO(NTv, true, false); else POW3F_GO(NTv, false, false); } while (0)
        {
            TagScope ts(d);
            switch (NT) {
                case 5: POW3F_NT(5); break;
                case 4: POW3F_NT(4); break;
                case 3: POW3F_NT(3); break;
                case 2: POW3F_NT(2); break;
                default: POW3F_NT(1); break;
            }
            launch_check();
            ts.stop();
        }
#undef POW3F_NT
#undef POW3F_GO
        hipLaunchKernelGGL(k_sum_parts4, dim3(cdiv(512 * N, 128)), dim3(256), 0, d->stream, part, (int64_t)grid, (int64_t)512 * N, Y, N, ldy);
        launch_check();
        dev_free(d, part);
        if (ssq_parts_out) { *ssq_parts_out = ssq_part; *nparts_out = grid * 8; }   // (the caller's next kernel adds them; it frees the block)
        return;
    }
    const int NT = (int)(N / 16);
    const int64_t nstages = cdiv(n, 32);
    const int grid = (int)std::min<int64_t>(num_cus(d), nstages);
    float* part = (float*)dev_alloc(d, sizeof(float) * (size_t)grid * 512 * N);
    double* ssq_part = ssq_out ? (double*)dev_alloc(d, sizeof(double) * (size_t)grid * 8) : nullptr;
    const size_t lds = (size_t)8 * 12288 + (size_t)8 * NT * 1024 + (size_t)NT * 3072 + 2048;
#define POW3_GO(NTv, CE, SZ, ME)                                                                                                  \
    do {                                                                                                                          \
        set_max_lds(d, reinterpret_cast<const void*>(k_pow3<NTv, CE, SZ, ME>));                                                   \
        hipLaunchKernelGGL((k_pow3<NTv, CE, SZ, ME>), dim3(grid), dim3(512), lds, d->stream, X, n, ldx, mu, Ppk3, NT, part, Z, ldz, nstages, ssq_part); \
    } while (0)
#define POW3_NT(NTv)                                                                                                              \
    do {                                                                                                                          \
        if (ssq_out) POW3_GO(NTv, true, false, true);                                                                             \
        else if (mu && Z) POW3_GO(NTv, true, true, false); else if (mu) POW3_GO(NTv, true, false, false);                         \
        else if (Z) POW3_GO(NTv, false, true, false); else POW3_GO(NTv, false, false, false);                                     \
    } while (0)
    {
        TagScope ts(d);
        switch (NT) {
            case 5: POW3_NT(5); break;
            case 4: POW3_NT(4); break;
            case 3: POW3_NT(3); break;
            case 2: POW3_NT(2); break;
            default: POW3_NT(1); break;
        }
        launch_check();
        ts.stop();
    }
#undef POW3_NT
#undef POW3_GO
    hipLaunchKernelGGL(k_sum_parts4, dim3(cdiv(512 * N, 128)), dim3(256), 0, d->stream, part, (int64_t)grid, (int64_t)512 * N, Y, N, ldy);
    launch_check();
    dev_free(d, part);
    if (ssq_parts_out) { *ssq_parts_out = ssq_part; *nparts_out = grid * 8; }   // (the caller's next kernel adds them; it frees the block)
}
// the move from the provisional centre to the true one (see k_pow3, MEANS).  One workgroup per FOUR columns of Y:
//   delta_f = Y[f][c1] / n;  t_j = sum_f delta_f P[f][j];  Y[f][j] -= n delta_f t_j (j < L);  Y[f][c1] = 0;
// workgroup 0 also: mu64 = mu0 + delta (+ muT, the data type's copy) and *tv = (sum of the waves' partials of sum (x - mu0)^2) - n |delta|^2.
// (every workgroup reads the sums column c1, so nobody may zero it in the same launch: that is the second launch, k_mean_fix<true>)
template <bool FINISH>
__global__ __launch_bounds__(256) void k_mean_fix(double* __restrict__ Y, int K, int d, int N, int64_t ldy, int c1, int L, double n_total,
                                                  const double* __restrict__ P, int64_t ldp, double* __restrict__ mu64, float* __restrict__ muT,
                                                  const double* __restrict__ ssq_part, int nparts, double* __restrict__ tv) {
    __shared__ double sd[512];
    __shared__ double sr[256];
    __shared__ double st[4];
    const int tid = threadIdx.x;
    if (FINISH) {        // (second launch, one workgroup: the sums column is no longer needed)
        for (int f = tid; f < K; f += 256) Y[(int64_t)f * ldy + c1] = 0.0;
        return;
    }
    for (int f = tid; f < K; f += 256) sd[f] = f < d ? Y[(int64_t)f * ldy + c1] / n_total : 0.0;
    __syncthreads();
    const int j0 = blockIdx.x * 4, jj = tid & 3, fl = tid >> 2;      // 64 row lanes x 4 columns
    double acc = 0;
    // (the fused pass multiplied by the TWO-PLANE P: the rank-one correction uses the same rounded values -- bf16(p) + bf16(p - bf16(p)),
    // round-to-nearest-even each, what v_cvt_pk_bf16_f32 and the host simulation oracle/cpu_ops.cpp two_plane() do)
    auto two_plane = [](double v) {
        auto bf = [](float f) {
            unsigned u = __float_as_uint(f);
            u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
            return __uint_as_float(u);
        };
        const float f = (float)v, h = bf(f), m = bf(f - h);
        return (double)h + (double)m;
    };
    if (j0 + jj < L)
        for (int f = fl; f < K; f += 64) acc += sd[f] * two_plane(P[(int64_t)f * ldp + j0 + jj]);
    sr[tid] = acc;
    __syncthreads();
    if (tid < 4) {
        double t = 0;
        for (int g = 0; g < 64; ++g) t += sr[g * 4 + tid];
        st[tid] = t;
    }
    __syncthreads();
    if (j0 + jj < L && j0 + jj != c1)
        for (int f = fl; f < K; f += 64) Y[(int64_t)f * ldy + j0 + jj] -= n_total * sd[f] * st[jj];
    if (blockIdx.x == 0) {
        double q = 0;
        for (int f = tid; f < K; f += 256) q += sd[f] * sd[f];
        double sp = 0;
        for (int e = tid; e < nparts; e += 256) sp += ssq_part[e];
        __syncthreads();
        sr[tid] = q;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) { if (tid < h) sr[tid] += sr[tid + h]; __syncthreads(); }
        const double q2 = sr[0];
        __syncthreads();
        sr[tid] = sp;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) { if (tid < h) sr[tid] += sr[tid + h]; __syncthreads(); }
        if (tid == 0) tv[0] = fmax(sr[0] - n_total * q2, 0.0);
        for (int f = tid; f < K; f += 256) {
            const double m = mu64[f] + sd[f];
            mu64[f] = m;
            muT[f] = (float)m;
        }
    }
}
bool op_power_pass_means(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t dcols, int64_t ldx, double n_total, const double* P,
                         int64_t N, int64_t ldp, int64_t L, double* Y, int64_t ldy, double* mu64, void* muT, double* ssq_scratch, double* tv) {
    const bool off = d->opt[OPT_MEANS_FOLD_ROWS] < 0 || pow3_knob_off(d, OPT_TWO_PLANE_OMEGA);
    if (off || L >= N || !pow3_ok(d, dt, X, n, K, ldx, muT, N, nullptr, 0)) return false;
    // the provisional centre: the means of a strided sample of the rows (one small pass)
    const int64_t ns = std::min<int64_t>(n, 4096), stride = n / ns;
    op_colmean(d, dt, X, ns, K, ldx * stride, double(ns), mu64, muT, false);
    const int64_t total = (K / 32) * (N / 16) * 64;
    bf16x8* Ppk3 = (bf16x8*)dev_alloc(d, sizeof(bf16x8) * total * 3);
    hipLaunchKernelGGL(k_pack_p3, dim3(cdiv(total, 256)), dim3(256), 0, d->stream, P, K, N, ldp, Ppk3, (int)(N / 16), total);
    launch_check();
    double* parts = nullptr;
    int nparts = 0;
    launch_pow3(d, (const float*)X, n, ldx, (const float*)muT, Ppk3, N, nullptr, 0, Y, ldy, &parts, &nparts, /*steering=*/true);
    dev_free(d, Ppk3);
    (void)ssq_scratch;
    const int c1 = (int)(N - 1);
    hipLaunchKernelGGL(k_mean_fix<false>, dim3((unsigned)cdiv(L, 4)), dim3(256), 0, d->stream, Y, (int)K, (int)dcols, (int)N, ldy, c1, (int)L, n_total, P, ldp,
                       mu64, (float*)muT, (const double*)parts, nparts, tv);
    hipLaunchKernelGGL(k_mean_fix<true>, dim3(1), dim3(256), 0, d->stream, Y, (int)K, (int)dcols, (int)N, ldy, c1, (int)L, n_total, P, ldp, mu64,
                       (float*)muT, (const double*)parts, nparts, tv);
    launch_check();
    dev_free(d, parts);
    return true;
}
bool op_power_pass(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* P, int64_t N, int64_t ldp,
                   void* Z, int64_t ldz, double* Y, int64_t ldy, bool steering) {
    const bool knob = pow3_knob_off(d, OPT_TWO_PLANE_OMEGA);
    if (knob || !pow3_ok(d, dt, X, n, K, ldx, mu, N, Z, ldz)) return false;
    const int64_t total = (K / 32) * (N / 16) * 64;
    bf16x8* Ppk3 = (bf16x8*)dev_alloc(d, sizeof(bf16x8) * total * 3);
    hipLaunchKernelGGL(k_pack_p3, dim3(cdiv(total, 256)), dim3(256), 0, d->stream, P, K, N, ldp, Ppk3, (int)(N / 16), total);
    launch_check();
    launch_pow3(d, (const float*)X, n, ldx, (const float*)mu, Ppk3, N, (float*)Z, ldz, Y, ldy, nullptr, nullptr, steering && Z == nullptr);
    dev_free(d, Ppk3);
    return true;
}
bool op_rebase_power_pass(Dev* d, int dt, const void* X, int64_t n, int64_t K, int64_t ldx, const void* mu, const double* G, int64_t L,
                          int64_t ldg, double rel_tol, int* ndead, const double* A, int64_t M, int64_t lda, double* T, int64_t ldt,
                          double* P_out, int64_t ldpo, void* Z, int64_t ldz, double* Y, int64_t ldy, bool steering) {
    const bool no_rt = pow3_knob_off(d, OPT_TWO_PLANE_ITERATE);
    if (!pow3_ok(d, dt, X, n, K, ldx, mu, M, Z, ldz) || no_rt || L == 0 || L > CHOL2_MAXL || M > TRSM_MAXM || M < L || P_out == nullptr) return false;
    launch_chol_rt(d, G, L, ldg, T, ldt, rel_tol, ndead, M);
    const int NTtot = (int)(M / 16);
    const int64_t total = (K / 32) * (int64_t)NTtot * 64;
    bf16x8* Ppk3 = (bf16x8*)dev_alloc(d, sizeof(bf16x8) * total * 3);
    switch (NTtot) {   // P_out = A R^-1, ROUNDED to two planes, and its operand planes
#define PETAL_TRSM_CASE(NB)                                                                                                            \
    case NB:                                                                                                                           \
        hipLaunchKernelGGL((k_trsm_pack<NB, true>), dim3(cdiv(K, 16)), dim3(64), trsm_lds_bytes(16 * NB), d->stream, A, lda, T, ldt, K, P_out, \
                           ldpo, Ppk3, NTtot);                                                                                         \
        break
        PETAL_TRSM_CASE(1); PETAL_TRSM_CASE(2); PETAL_TRSM_CASE(3); PETAL_TRSM_CASE(4); PETAL_TRSM_CASE(5);
#undef PETAL_TRSM_CASE
        default: throw std::runtime_error("op_rebase_power_pass: order out of range");
    }
    launch_check();
    launch_pow3(d, (const float*)X, n, ldx, (const float*)mu, Ppk3, M, (float*)Z, ldz, Y, ldy, nullptr, nullptr, steering && Z == nullptr);
    dev_free(d, Ppk3);
    return true;
}

void op_chol_inv(Dev* d, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead, int64_t Lz,
                 int64_t ndead_cols) {
    if (L == 0) return;
    if (Lz < L) Lz = L;
    // k_chol_inv2 takes the orders up to 140; beyond, the blocked right-looking factorisation on the chip-wide fp64 GEMM kernel.
    // (Orders 141 .. 200 once ran an older one-workgroup kernel, k_chol_inv, whose build-T-in-global-memory mode no test reached and
    // whose factors were WRONG -- RandomizedPca with 132 <= k <= 190, found by dev/fuzz_rpca.py in round 3; retired in round 6.)
    if (L > CHOL2_MAXL) { chol_inv_blocked(d, G, L, ldg, T, ldt, rel_tol, ndead, Lz); return; }
    set_max_lds(d, reinterpret_cast<const void*>(k_chol_inv2));
    hipLaunchKernelGGL(k_chol_inv2, dim3(1), dim3(CHOL_THREADS), chol2_lds_bytes((int)L), d->stream, G, (int)L, ldg, T, ldt, rel_tol,
                       ndead, (int)Lz, (const double*)nullptr, 0, (int)(ndead_cols > 0 ? ndead_cols : L));
    launch_check();
}
bool op_chol_rt(Dev* d, int dt, int64_t n, const double* G, int64_t L, int64_t ldg, double* T, int64_t ldt, double rel_tol, int* ndead,
                int64_t Lz, int64_t ndead_cols) {
    if (Lz < L) Lz = L;
    const bool rt = dt == F32 && gemm_split_product(d) && n >= 64 && n < (int64_t(1) << 31) && L > 0 && L <= CHOL2_MAXL && Lz % 16 == 0 &&
                    Lz <= TRSM_MAXM;
    if (!rt) { op_chol_inv(d, G, L, ldg, T, ldt, rel_tol, ndead, Lz, ndead_cols); return false; }
    launch_chol_rt(d, G, L, ldg, T, ldt, rel_tol, ndead, Lz, ndead_cols);
    return true;
}
void op_trsm_right(Dev* d, const double* A, int64_t rows, int64_t lda, const double* RT, int64_t M, int64_t ldt, double* out, int64_t ldo) {
    if (rows % 16 != 0 || M % 16 != 0 || M > TRSM_MAXM) throw std::logic_error("op_trsm_right: shape outside the kernel");
    switch ((int)(M / 16)) {
#define PETAL_TRSMR_CASE(NB)                                                                                                         \
    case NB:                                                                                                                         \
        hipLaunchKernelGGL((k_trsm_pack<NB, false>), dim3(cdiv(rows, 16)), dim3(64), trsm_lds_bytes(16 * NB), d->stream, A, lda, RT, ldt, rows, \
                           out, ldo, (bf16x8*)nullptr, NB);                                                                          \
        break
        PETAL_TRSMR_CASE(1); PETAL_TRSMR_CASE(2); PETAL_TRSMR_CASE(3); PETAL_TRSMR_CASE(4); PETAL_TRSMR_CASE(5);
        PETAL_TRSMR_CASE(6); PETAL_TRSMR_CASE(7); PETAL_TRSMR_CASE(8); PETAL_TRSMR_CASE(9);
#undef PETAL_TRSMR_CASE
        default: throw std::logic_error("op_trsm_right: order out of range");
    }
    launch_check();
}
void op_eigh(Dev* d, double* A, int64_t L, int64_t lda, double* V, int64_t ldv, double* w, double tol_rel, bool clustered, int64_t Lz, int64_t ncheck,
             int* verdict, bool verdict_fresh, double gap_tol_override) {
    if (L == 0) { if (verdict && verdict_fresh) HIP_CHECK(hipMemsetAsync(verdict, 0, sizeof(int), d->stream)); return; }
    const bool pad_done = Lz <= L;   // else: rows / columns L .. Lz - 1 of V are to be zeroed here
    if (L > EIG_MAXL) throw std::runtime_error("eigh: matrix too large for the one-workgroup Jacobi solver");
    // Two-stage solver first (tridiagonalisation on one workgroup, then one wave per eigenpair over the chip); the Jacobi
    // launches below run behind it and return at once unless its verdict flags eigenvalues too close for its vectors.
    const bool jacobi_only = opt_on(d, OPT_EIGH_JACOBI);
    int* flag = nullptr;
    char* ts = nullptr;
    const bool two_stage = !jacobi_only && !clustered && L >= 3 && L <= 2048;
    // verdict given (and the order is one the register-resident kernels take): the two-stage result is delivered as it is and
    // the verdict goes to the caller's device word, who decides what to do about a flagged spectrum (RandomizedPca redoes the
    // fit on its robust path) -- the two fallback launches, which return at once on every separated spectrum, are not issued
    const bool ext_verdict = verdict != nullptr && two_stage && L <= 138;
    // (verdict_fresh: the caller's word is cleared first -- by the tridiagonalisation kernel itself where it runs)
    if (verdict && verdict_fresh && !ext_verdict) HIP_CHECK(hipMemsetAsync(verdict, 0, sizeof(int), d->stream));
    // k_tridiag_r (orders up to 138) writes the padding itself; every other route gets one 2-D clear of the Lz x Lz frame first
    if (!pad_done && !(two_stage && L <= 138)) HIP_CHECK(hipMemset2DAsync(V, sizeof(double) * ldv, 0, sizeof(double) * Lz, Lz, d->stream));
    if (two_stage) {
        // eigenvectors of eigenvalues closer than gap_tol ||A|| come out only eps / gap_tol accurate: fp32 results carry
        // 2e-8, fp64 results 2e-11; anything closer goes to Jacobi
        const double gap_tol = gap_tol_override > 0 ? gap_tol_override : (tol_rel >= 1e-9 ? 1e-8 : 1e-5);
        const int64_t ldw = L | 1;
        const size_t lds_in = sizeof(double) * ((size_t)L * ldw + 2 * L + 32);
        const bool inlds = lds_in <= 160 * 1024 - 256;
        const bool regs = L <= 138;                               // (the padded working copy of k_tridiag_r fits the LDS)
        const size_t bytes = sizeof(double) * (4 * L + L * L + (inlds ? 0 : L * ldw)) + 64;
        ts = (char*)dev_alloc(d, bytes);
        double* dd = reinterpret_cast<double*>(ts);
        double* ee = dd + L;
        double* tau = ee + L;
        double* gg = tau + L;
        double* HV = gg + L;
        double* Wg = inlds ? nullptr : HV + L * L;
        flag = reinterpret_cast<int*>(ts + bytes - 64);
        if (ext_verdict) flag = verdict;   // (the closeness verdict ORs into the caller's word; no Jacobi launches behind it)
        if (regs) {
#define PETAL_TRI_LAUNCH(TM, ST)                                                                                                   \
    do {                                                                                                                           \
        const size_t lds_r = sizeof(double) * ((size_t)L * tri_ld((int)L, ST) + 8 * TM + tri_sp_len(TM));                          \
        set_max_lds(d, reinterpret_cast<const void*>(k_tridiag_r<TM, ST>));                                                        \
        hipLaunchKernelGGL((k_tridiag_r<TM, ST>), dim3(1), dim3(TRR_THREADS), lds_r, d->stream, A, (int)L, lda, dd, ee, HV, tau, gg, flag, V, ldv, (int)(pad_done ? 0 : Lz), (ext_verdict && !verdict_fresh) ? 0 : 1); \
    } while (0)
#define PETAL_TRIW_LAUNCH(TM)                                                                                                      \
    do {                                                                                                                           \
        const size_t lds_r = sizeof(double) * ((size_t)L * triw_ld((int)L, TM) + 32 * TM);                                         \
        set_max_lds(d, reinterpret_cast<const void*>(k_tridiag_w<TM>));                                                            \
        hipLaunchKernelGGL((k_tridiag_w<TM>), dim3(1), dim3(TRR_THREADS), lds_r, d->stream, A, (int)L, lda, dd, ee, HV, tau, gg, flag, V, ldv, (int)(pad_done ? 0 : Lz), (ext_verdict && !verdict_fresh) ? 0 : 1); \
    } while (0)
            // l <= 80: the 8-byte form (78.5 us at l = 74, where a step is a chain of latencies and the 16-byte form measures 83.6);
            // above: the 16-byte form (238 us at l = 138 against 290).  (Round 6 built a form with the MATRIX in registers, only vectors
            // through LDS: parity-green and no faster -- 78.2 us at l = 74: the step is a chain of dependent latencies, not LDS
            // bandwidth; dev/tridiag_q_kernel.h, EXPERIMENTS.md.)
            if (L <= 80) PETAL_TRI_LAUNCH(10, 2);
            else PETAL_TRIW_LAUNCH(9);
#undef PETAL_TRI_LAUNCH
#undef PETAL_TRIW_LAUNCH
            launch_check();
            const int hv_rows = (int)std::min<int64_t>(L - 2, (96 * 1024) / (8 * L));
            const size_t lds_e = sizeof(double) * (23 * (((L + 7) & ~7) + 8) + (size_t)hv_rows * L);
            if (L <= 128) {
                set_max_lds(d, reinterpret_cast<const void*>(k_trieig_r<4, 2>));
                hipLaunchKernelGGL((k_trieig_r<4, 2>), dim3(cdiv(L, 4)), dim3(256), lds_e, d->stream, dd, ee, HV, tau, gg, (int)L, gap_tol, hv_rows, w, V, ldv, flag, (int)(ncheck > 0 ? ncheck : L));
            } else {
                set_max_lds(d, reinterpret_cast<const void*>(k_trieig_r<4, 3>));
                hipLaunchKernelGGL((k_trieig_r<4, 3>), dim3(cdiv(L, 4)), dim3(256), lds_e, d->stream, dd, ee, HV, tau, gg, (int)L, gap_tol, hv_rows, w, V, ldv, flag, (int)(ncheck > 0 ? ncheck : L));
            }
            launch_check();
        } else {
            {
            // one launch per Householder step, the whole chip per launch (k_tridiag_mw): 29 -> ~5 ms at order 512
            const int64_t ldm = L | 1;
            double* mw = (double*)dev_alloc(d, sizeof(double) * ((size_t)L * ldm + 4 * L));
            double* vp = mw + (size_t)L * ldm;
            HIP_CHECK(hipMemcpy2DAsync(mw, ldm * sizeof(double), A, lda * sizeof(double), L * sizeof(double), L, hipMemcpyDeviceToDevice, d->stream));
            HIP_CHECK(hipMemsetAsync(vp, 0, sizeof(double) * 4 * L, d->stream));
            HIP_CHECK(hipMemsetAsync(HV, 0, sizeof(double) * L * L, d->stream));
            const size_t lds_mw = sizeof(double) * (3 * L + 16);
            for (int64_t k = 0; k < L; ++k) {
                const int64_t m = L - k - 1;
                const unsigned grid = (unsigned)std::min<int64_t>(512, std::max<int64_t>(1, (m + 3) / 4));
                hipLaunchKernelGGL(k_tridiag_mw, dim3(grid), dim3(256), lds_mw, d->stream, mw, ldm, (int)L, (int)k, vp + ((k + 1) & 1) * 2 * L,
                                   vp + (k & 1) * 2 * L, dd, ee, HV, tau);
            }
            launch_check();
            dev_free(d, mw);
            }
            launch_check();
            if (L <= 512) {
                hipLaunchKernelGGL(k_trieig<4>, dim3(cdiv(L, 4)), dim3(256), sizeof(double) * 15 * L, d->stream, dd, ee, HV, tau, (int)L, w, V, ldv);
            } else {
                set_max_lds(d, reinterpret_cast<const void*>(k_trieig<1>));
                hipLaunchKernelGGL(k_trieig<1>, dim3((unsigned)L), dim3(64), sizeof(double) * 6 * L, d->stream, dd, ee, HV, tau, (int)L, w, V, ldv);
            }
            launch_check();
            hipLaunchKernelGGL(k_trieig_verdict, dim3(1), dim3(1024), 0, d->stream, w, ee, (int)L, gap_tol, A, lda, flag, (int)(ncheck > 0 ? ncheck : L));
            launch_check();
        }
    }
    struct Cleanup { Dev* d; char* p; ~Cleanup() { if (p) dev_free(d, p); } } cleanup{d, ts};
    if (ext_verdict) return;
    if (jaca_lds_bytes((int)L) <= 160 * 1024 - 256) {
        // split solver: A in LDS + rotation log, eigenvectors replayed on L waves
        const int Le = (int)((L + 1) & ~1), half = Le / 2, rounds = Le - 1;
        const size_t nlog = (size_t)JACA_MAX_SWEEPS * rounds * half;
        char* buf = (char*)dev_alloc(d, nlog * 16 + 16 + sizeof(int) * (L + 4));
        jf64x2* log_cs = reinterpret_cast<jf64x2*>(buf);
        int* nrounds = reinterpret_cast<int*>(buf + nlog * 16);
        int* rank = nrounds + 4;
        // block threads: groups of GW lanes own two row pairs each; 32-lane groups (fewer blocks per thread: the rounds are
        // paced by the per-thread work on both sides of the barrier) while the workgroup still fits 1024 threads
        const int groups = (half + 1) / 2;                                // two row pairs each
        const int pw = (half + 63) / 64 * 64;                             // angle threads (whole waves)
        const int gw = (pw + 32 * groups <= 1024) ? 32 : 16;
        const int mb2 = std::max(1, (half - 1 + gw - 1) / gw);            // partner pairs per lane
        const int threads = std::min(1024, (pw + gw * groups + 63) / 64 * 64);
        const size_t lds = jaca_lds_bytes((int)L);
#define JACA_CASE(M, G)                                                                                                  \
    case 10 * M + (G == 32 ? 1 : 0): {                                                                                   \
        set_max_lds(d, reinterpret_cast<const void*>(k_jacobi_a<M, G>));                        \
        hipLaunchKernelGGL((k_jacobi_a<M, G>), dim3(1), dim3(threads), lds, d->stream, A, (int)L, lda, log_cs, nrounds, w, rank, pw, tol_rel, (const int*)flag); \
    } break;
        switch (10 * std::min(mb2, 5) + (gw == 32 ? 1 : 0)) {
            JACA_CASE(1, 16) JACA_CASE(2, 16) JACA_CASE(3, 16) JACA_CASE(4, 16) JACA_CASE(5, 16)
            JACA_CASE(1, 32) JACA_CASE(2, 32) JACA_CASE(3, 32)
            default: throw std::runtime_error("eigh: unsupported size for the split Jacobi solver");
        }
#undef JACA_CASE
        launch_check();
        const int hp = (half + 63) / 64;
        const dim3 grid((unsigned)cdiv(L, JR_WAVES));
        const size_t lds2 = sizeof(double) * JR_WAVES * Le;
        if (hp <= 1) hipLaunchKernelGGL(k_apply_rot<1>, grid, dim3(64 * JR_WAVES), lds2, d->stream, log_cs, nrounds, rank, (int)L, V, ldv, (const int*)flag);
        else hipLaunchKernelGGL(k_apply_rot<2>, grid, dim3(64 * JR_WAVES), lds2, d->stream, log_cs, nrounds, rank, (int)L, V, ldv, (const int*)flag);
        launch_check();
        dev_free(d, buf);
        return;
    }
    double* Vtmp = (double*)dev_alloc(d, sizeof(double) * L * lda);  // (the kernel indexes it with A's leading dimension)
    const size_t lds = sizeof(double) * jac_ws_doubles((int)L, 1024);
    {
        set_max_lds(d, reinterpret_cast<const void*>(k_eigh<0>));
        hipLaunchKernelGGL(k_eigh<0>, dim3(1), dim3(1024), lds, d->stream, A, (int)L, lda, Vtmp, V, ldv, w, tol_rel, (const int*)flag);
    }
    launch_check();
    dev_free(d, Vtmp);
}
// One-sided (Hestenes) Jacobi on the ROWS of M (L x L), G accumulates the rotations from the identity: one workgroup, 32 lanes
// per row pair, the L / 2 disjoint pairs of a round-robin round in flight together; a round ends with one barrier.  M and G live in
// LDS when both fit (L <= 96), else in global memory (L2-resident).  Built for accuracy, not speed: it only runs for fp64
// data whose wanted singular values fall below the Gram route's 10^-3.5 sigma_1 accuracy floor.
constexpr int HJ_THREADS = 1024, HJ_GROUP = 32;
template <bool INLDS>
__global__ __launch_bounds__(HJ_THREADS) void k_jacobi_svd_rows(double* __restrict__ A, int L, int64_t lda, double* __restrict__ Gg,
                                                                double* __restrict__ U, int64_t ldu, double* __restrict__ s_inv, int* __restrict__ nonconv) {
    extern __shared__ __attribute__((aligned(16))) double sm_hj[];   // [row norms (L) | flag | M, G when INLDS]: all dynamic
    const int tid = threadIdx.x, gl = tid & (HJ_GROUP - 1), grp = tid / HJ_GROUP, ngrp = HJ_THREADS / HJ_GROUP;
    const int ld = INLDS ? (L | 1) : L;
    double* s_nrm = sm_hj;
    volatile int* s_rotp = reinterpret_cast<volatile int*>(sm_hj + L);
#define s_rot (*s_rotp)
    double* M = INLDS ? sm_hj + L + 2 : A;
    double* G = INLDS ? sm_hj + L + 2 + (size_t)L * ld : Gg;
    const int64_t ldm = INLDS ? ld : lda;
    for (int e = tid; e < L * L; e += HJ_THREADS) {
        const int r = e / L, c = e - r * L;
        if (INLDS) M[r * ld + c] = A[(int64_t)r * lda + c];
        G[(size_t)r * ld + c] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    const int n = (L + 1) & ~1, half = n >> 1;   // round-robin over n players (a dummy when L is odd)
    // a pair is orthogonal when |p . q| <= tol |p| |q|, tol = 4 eps sqrt(L): the rounding noise of an L-term fp64 dot product
    // (a fixed 1e-15 sat below it, so the flag rarely cleared and the kernel usually ran all 40 sweeps: ADVICE round 3)
    const double otol = 8.9e-16 * sqrt((double)L);
    bool converged = false;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int r = 0; r < n - 1; ++r) {
            for (int slot = grp; slot < half; slot += ngrp) {
                int p = slot == 0 ? n - 1 : (r + slot) % (n - 1);
                int q = (n - 1 - slot + r) % (n - 1);
                if (p > q) { const int t = p; p = q; q = t; }
                if (q >= L) continue;                      // the dummy sits out
                double* mp = M + (size_t)p * ldm;
                double* mq = M + (size_t)q * ldm;
                double al = 0, be = 0, ga = 0;
                for (int j = gl; j < L; j += HJ_GROUP) { const double a = mp[j], b = mq[j]; al += a * a; be += b * b; ga += a * b; }
                for (int off = HJ_GROUP / 2; off > 0; off >>= 1) {
                    al += __shfl_xor(al, off, 64); be += __shfl_xor(be, off, 64); ga += __shfl_xor(ga, off, 64);
                }
                if (!(fabs(ga) > otol * sqrt(al * be))) continue;   // (uniform over the 32 lanes of the pair)
                if (gl == 0) s_rot = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                double* gp = G + (size_t)p * ld;
                double* gq = G + (size_t)q * ld;
                for (int j = gl; j < L; j += HJ_GROUP) {
                    const double a = mp[j], b = mq[j];
                    mp[j] = cs * a - sn * b; mq[j] = sn * a + cs * b;
                    const double u = gp[j], v = gq[j];
                    gp[j] = cs * u - sn * v; gq[j] = sn * u + cs * v;
                }
            }
            __syncthreads();
        }
        const int any = s_rot;
        __syncthreads();
        if (!any) { converged = true; break; }
    }
    if (nonconv && tid == 0 && !converged) *nonconv = 1;   // 40 sweeps were not enough: the caller keeps its other result
    // singular values = row norms; ascending order (ties: lower row first), 1 / s out
    for (int i = tid; i < L; i += HJ_THREADS) {
        double a = 0;
        for (int j = 0; j < L; ++j) { const double v = M[(size_t)i * ldm + j]; a += v * v; }
        s_nrm[i] = sqrt(a);
    }
    __syncthreads();
    for (int i = tid; i < L; i += HJ_THREADS) {
        const double mine = s_nrm[i];
        int rank = 0;
        for (int k2 = 0; k2 < L; ++k2) { const double o = s_nrm[k2]; rank += (o < mine || (o == mine && k2 < i)) ? 1 : 0; }
        s_inv[rank] = mine > 0.0 ? 1.0 / mine : 0.0;
        for (int j = 0; j < L; ++j) U[(int64_t)j * ldu + rank] = G[(size_t)i * ld + j];
    }
#undef s_rot
}
void op_jacobi_svd_rows(Dev* d, double* A, int64_t L, int64_t lda, double* U, int64_t ldu, double* s_inv, int* nonconv) {
    if (L == 0) return;
    if (L > 1024) throw std::runtime_error("jacobi_svd_rows: order above 1024");
    const size_t lds = sizeof(double) * (L + 2 + 2 * (size_t)L * (L | 1));
    if (lds <= 150 * 1024) {
        set_max_lds(d, reinterpret_cast<const void*>(k_jacobi_svd_rows<true>));
        hipLaunchKernelGGL(k_jacobi_svd_rows<true>, dim3(1), dim3(HJ_THREADS), lds, d->stream, A, (int)L, lda, (double*)nullptr, U, ldu, s_inv, nonconv);
        launch_check();
        return;
    }
    double* G = (double*)dev_alloc(d, sizeof(double) * L * L);
    hipLaunchKernelGGL(k_jacobi_svd_rows<false>, dim3(1), dim3(HJ_THREADS), sizeof(double) * (L + 2), d->stream, A, (int)L, lda, G, U, ldu, s_inv, nonconv);
    launch_check();
    dev_free(d, G);
}
// ONE block: lane <-> column (row-major reads stay coalesced), the 16 waves split the rows and add through LDS in a fixed order;
// the largest squared residual, theta_0 and the bad flag (a non-finite residual, or the eigen-solver's closeness verdict `flag`
// when given) are written directly -- no clear, no atomics
__global__ __launch_bounds__(1024) void k_ritz_residual(const double* __restrict__ CV, int64_t ldc, const double* __restrict__ Vr, int64_t ldv,
                                                        int64_t rows, int nc, const double* __restrict__ theta, const int* __restrict__ flag,
                                                        double* __restrict__ out3, double* __restrict__ w_out) {
    __shared__ double red[16][65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double worst = 0.0, bad = 0.0;   // (meaningful in wave 0)
    for (int jb = 0; jb < nc; jb += 64) {
        const int j = jb + lane;
        const bool on = j < nc;
        const double th = on ? theta[j] : 0.0;
        double s2 = 0;
        if (on)
            for (int64_t i = wv; i < rows; i += 16) { const double v = CV[i * ldc + j] - th * Vr[i * ldv + j]; s2 += v * v; }
        red[wv][lane] = s2;
        __syncthreads();
        if (wv == 0 && on) {
            double tot = 0;
            for (int w = 0; w < 16; ++w) tot += red[w][lane];
            if (!(tot < 1e300)) bad = 1.0;
            else worst = fmax(worst, tot);
            if (w_out) w_out[j] = th;   // the nc wanted Ritz values, delivered to the caller's array by the way
        }
        __syncthreads();
    }
    if (wv == 0) {
        for (int off = 32; off > 0; off >>= 1) { worst = fmax(worst, __shfl_down(worst, off, 64)); bad = fmax(bad, __shfl_down(bad, off, 64)); }
        if (lane == 0) {
            if (flag && *flag != 0) bad = 1.0;
            out3[0] = worst; out3[1] = nc > 0 ? theta[0] : 0.0; out3[2] = bad;
        }
    }
}
void op_ritz_residual(Dev* d, const double* CV, int64_t ldc, const double* Vr, int64_t ldv, int64_t rows, int64_t nc, const double* theta,
                      const int* flag, double* out3, double* w_out) {
    hipLaunchKernelGGL(k_ritz_residual, dim3(1), dim3(1024), 0, d->stream, CV, ldc, Vr, ldv, rows, (int)nc, theta, flag, out3, w_out);
    launch_check();
}
// one block per column j: the eigenvector's sign is NORMALISED first (its first component of largest magnitude made positive).
// An eigen-solver returns v or -v at its whim -- the two-stage solver's choice can flip under a last-bit perturbation of the
// matrix, and a flipped whitening row sends the fixed-point iteration from the same w_init down another path (a sharded fit
// and the single-process fit of the same data then stop at different iteration counts).
__global__ __launch_bounds__(256) void k_whiten_k(const double* __restrict__ U, int64_t ldu, const double* __restrict__ lam, int64_t rows,
                                                  int64_t nc, int64_t ncp, double scale, double* __restrict__ KT, double* __restrict__ KTs) {
    __shared__ double rm[256];
    __shared__ int64_t ri[256];
    const int64_t j = blockIdx.x;
    if (j >= nc) {
        for (int64_t i = threadIdx.x; i < rows; i += 256) { KT[i * ncp + j] = 0.0; KTs[i * ncp + j] = 0.0; }
        return;
    }
    double best = -1.0;
    int64_t bi = 0;
    for (int64_t i = threadIdx.x; i < rows; i += 256) {   // ascending rows per thread: strict '>' keeps the first maximum
        const double a = fabs(U[i * ldu + j]);
        if (a > best) { best = a; bi = i; }
    }
    rm[threadIdx.x] = best; ri[threadIdx.x] = bi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            const double ob = rm[threadIdx.x + st];
            const int64_t oi = ri[threadIdx.x + st];
            if (ob > rm[threadIdx.x] || (ob == rm[threadIdx.x] && oi < ri[threadIdx.x])) { rm[threadIdx.x] = ob; ri[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    const double sgn = (rm[0] > 0.0 && U[ri[0] * ldu + j] < 0.0) ? -1.0 : 1.0;
    const double sg = sqrt(fmax(lam[j], 0.0));
    const double f = sg > 0.0 ? sgn / sg : 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += 256) {
        const double v = U[i * ldu + j] * f;
        KT[i * ncp + j] = v;
        KTs[i * ncp + j] = v * scale;
    }
}
void op_whiten_k(Dev* d, const double* U, int64_t ldu, const double* lam, int64_t rows, int64_t nc, int64_t ncp, double scale,
                 double* KT, double* KTs) {
    if (rows * ncp == 0) return;
    hipLaunchKernelGGL(k_whiten_k, dim3((unsigned)ncp), dim3(256), 0, d->stream, U, ldu, lam, rows, nc, ncp, scale, KT, KTs);
    launch_check();
}
__global__ __launch_bounds__(256) void k_refill_zero_cols(double* __restrict__ Y, int64_t rows, int64_t ldy, const double* __restrict__ Src, int64_t lds) {
    __shared__ int nz;
    const int64_t j = blockIdx.x;
    if (threadIdx.x == 0) nz = 0;
    __syncthreads();
    bool any = false;
    for (int64_t i = threadIdx.x; i < rows; i += 256) any = any || (Y[i * ldy + j] != 0.0);
    if (any) nz = 1;
    __syncthreads();
    if (nz) return;
    for (int64_t i = threadIdx.x; i < rows; i += 256) Y[i * ldy + j] = Src[i * lds + j];
}
void op_refill_zero_cols(Dev* d, double* Y, int64_t rows, int64_t cols, int64_t ldy, const double* Src, int64_t lds) {
    if (rows == 0 || cols == 0) return;
    hipLaunchKernelGGL(k_refill_zero_cols, dim3((unsigned)cols), dim3(256), 0, d->stream, Y, rows, ldy, Src, lds);
    launch_check();
}
void op_dscal(Dev* d, double* x, int64_t count, double alpha) {
    if (!count) return;
    hipLaunchKernelGGL(k_dscal, dim3(cdiv(count, 256)), dim3(256), 0, d->stream, x, count, alpha);
    launch_check();
}
void op_daxpy(Dev* d, int64_t count, double alpha, const double* x, double* y) {
    if (!count) return;
    hipLaunchKernelGGL(k_daxpy, dim3(cdiv(count, 256)), dim3(256), 0, d->stream, count, alpha, x, y);
    launch_check();
}
void op_dvec(Dev* d, int mode, const double* x, double* y, int64_t count, double thr) {
    if (!count) return;
    hipLaunchKernelGGL(k_dvec, dim3(cdiv(count, 256)), dim3(256), 0, d->stream, mode, x, y, count, thr);
    launch_check();
}
void op_dscale_cols(Dev* d, double* A, int64_t M, int64_t N, int64_t lda, const double* s) {
    if (M * N == 0) return;
    hipLaunchKernelGGL(k_dscale_cols, dim3(cdiv(M * N, 256)), dim3(256), 0, d->stream, A, M, N, lda, s);
    launch_check();
}
void op_cvt_from_f64(Dev* d, int dt, void* dst, const double* src, int64_t count) {
    if (!count) return;
    DISPATCH_T(dt, hipLaunchKernelGGL(k_cvt_from_f64<T>, dim3(cdiv(count, 256)), dim3(256), 0, d->stream, (T*)dst, src, count));
    launch_check();
}
void op_pad_to_f64(Dev* d, int dt, double* dst, int64_t rows_p, int64_t cols_p, const void* src, int64_t rows, int64_t cols, int64_t lds,
                   double* zero_ptr, int64_t zero_count) {
    if (rows_p * cols_p == 0 && zero_count == 0) return;
    DISPATCH_T(dt, hipLaunchKernelGGL(k_pad_to_f64<T>, dim3(cdiv(std::max(rows_p * cols_p, zero_count), 256)), dim3(256), 0, d->stream, dst,
                                      rows_p, cols_p, (const T*)src, rows, cols, lds, zero_ptr, zero_count));
    launch_check();
}
void op_cvt_to_f64(Dev* d, int dt, double* dst, const void* src, int64_t count) {
    if (!count) return;
    DISPATCH_T(dt, hipLaunchKernelGGL(k_cvt_to_f64<T>, dim3(cdiv(count, 256)), dim3(256), 0, d->stream, dst, (const T*)src, count));
    launch_check();
}

}  // namespace petal
