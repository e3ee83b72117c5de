/* petal_hip.h -- C ABI of the MI355X-native (gfx950) implementation of petal-decomposition's
 * dense hot path: Pca / RandomizedPca / FastIca  fit / transform / inverse_transform.
 *
 * The reference (petabi/petal-decomposition v0.9.0, Rust) has no FFI for this path: its boundary is
 * the crate's public generic API (src/lib.rs:17-18).  Every entry point below names the reference
 * interface it replaces (file:line relative to the crate root); a Rust facade binds them 1:1
 * (INTEGRATION.md shows the `extern "C"` block and the ndarray glue).
 *
 * Conventions
 *  - plain pointers and sizes only; `petal_matrix` is ndarray's (ptr, shape, strides) triple
 *    (strides in ELEMENTS, like ndarray) plus a dtype tag and the memory space of `data`.
 *  - small results (components, means, singular values, ...) are written to HOST memory owned by the
 *    caller, in the dtype of the input; large results (n x k) go to a caller-described petal_matrix.
 *  - the library never frees or mutates caller memory that is passed as `const`.
 *  - every call is synchronous: it returns after the device work has completed.
 *  - return value: PETAL_OK or one of the error codes; petal_last_error() holds the reference's
 *    message text (e.g. "every dimension should be at least 3", src/pca.rs:200-203).
 *  - a ctx is used by one thread at a time; several ctxs may coexist (one per GPU / process).
 *  - there is NO CPU fallback: without a usable gfx950 device petal_ctx_create fails.
 */
#ifndef PETAL_HIP_H
#define PETAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct petal_ctx petal_ctx;

/* DecompositionError (src/lib.rs:22-28) + a device/collective failure class.
 * PETAL_LINALG_ERROR: where the crate's LAPACK calls report info != 0 (src/linalg.rs:58, 84, 115) -- in practice a NaN or an infinity in
 * the input: petal_pca_fit / petal_rpca_fit return it with "did not converge", petal_fastica_fit with "cannot compute eigenvalues"
 * (petal_last_error has the text); all-zero and constant matrices are legal inputs (zero singular values). */
enum { PETAL_OK = 0, PETAL_INVALID_INPUT = 1, PETAL_LINALG_ERROR = 2, PETAL_DEVICE_ERROR = 3 };
enum { PETAL_F32 = 0, PETAL_F64 = 1 };
enum { PETAL_HOST = 0, PETAL_DEVICE = 1 };
enum { PETAL_SUM = 0, PETAL_MAX = 1, PETAL_MIN = 2 };
/* FastICA semantics: TEXTBOOK = (W W^T)^(-1/2) W and rows.rows convergence test;
 * REFERENCE_LITERAL = the crate's arithmetic as written (src/ica.rs:345-349, 369-380). */
enum { PETAL_ICA_TEXTBOOK = 0, PETAL_ICA_REFERENCE_LITERAL = 1 };

typedef struct petal_matrix {
    void*   data;
    int64_t rows, cols;
    int64_t row_stride, col_stride; /* in elements */
    int32_t dtype;                  /* PETAL_F32 | PETAL_F64 */
    int32_t space;                  /* PETAL_HOST | PETAL_DEVICE */
} petal_matrix;

/* Sample-sharded multi-GPU: every rank holds a row block of X; the library calls this hook to sum
 * (or max/min) small replicated buffers across ranks.  `buf` is DEVICE memory of `count` elements
 * of `dtype`; the reduction must be enqueued on / ordered with `stream` (a hipStream_t).  With
 * torch.distributed this is one all_reduce on the RCCL process group. */
typedef int (*petal_allreduce_fn)(void* user, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream);

typedef struct petal_stats {
    double  fit_ms;            /* wall time of the last fit call (host clock, incl. final sync)        */
    /* the two X-streaming power-iteration GEMM kernels (hipEvent time on the ctx stream, profiling on) */
    double  xp_ms;             /* K1  Z = Xc . P   : summed kernel time                                */
    int64_t xp_launches;
    double  atb_ms;            /* K2  Y = Xc^T . Z : summed kernel time (main kernel, not its reducer) */
    int64_t atb_launches;
    double  pass_flops;        /* algorithmic flops of ONE such launch: 2 n d l, l = k + n_oversample  */
    double  pass_bytes;        /* algorithmic bytes of ONE such launch: 4 (n d + n l + d l)            */
    double  ica_step_ms;       /* fused FastICA step kernel: summed kernel time                       */
    int64_t ica_step_launches;
    double  ica_step_flops;    /* per launch: 4 nc^2 n                                                 */
    double  ica_step_bytes;    /* per launch: 4 nc n                                                   */
    int64_t n_iter;            /* FastICA iterations of the last fit                                  */
    /* sample-sharded fits: what went through the collective during the last fit (separates comm from compute) */
    int64_t allreduce_calls;   /* all-reduce calls issued (RandomizedPca: n_iter + 3)                  */
    double  allreduce_bytes;   /* payload bytes summed over those calls                                */
    double  allreduce_ms;      /* stream time of the all-reduce calls that were bracketed (profiling on: one call per fit at
                                  level 1, the index rotating from fit to fit; all at level 2)                              */
    int64_t allreduce_timed;   /* ... and how many were: allreduce_ms / allreduce_timed * allreduce_calls estimates a fit's  */
    /* how the last fit held X on the device */
    int64_t x_row_pitch_bytes; /* row pitch of X as the kernels streamed it (a host or strided input is copied with 128 B of
                                  padding per row when its natural pitch is a multiple of 1 KiB: spreads the rows over the
                                  memory channels; a device input that is streamed in place keeps the caller's pitch)      */
    int64_t x_zero_copy;       /* 1: the caller's device buffer was streamed in place                                       */
    /* RandomizedPca: how the last fit ran */
    int64_t rpca_redo;         /* 0: the optimistic run stood; 1: redone with three-plane operands (heavy-tailed spectrum: the
                                  16-bit rounding of the sketch matrix / iterates was not harmless); 3: redone with the sketch re-based on the
                                  tall side before its product with Xc^T (a pivot was lost in the first run: on full-rank data that is
                                  the un-rebased first product pair's, sigma_1 / sigma_l beyond ~5e3); 2: redone on the robust
                                  path (a pivot lost again: rank deficiency)                                                       */
    double  pow_ms;            /* fused power-iteration pass Y' = Xc^T (Xc P) (one pass over X): summed kernel time           */
    int64_t pow_launches;
    double  stream_ms;         /* the other row-streaming kernels of a RandomizedPca fit (means pass, U = Z (T Uh)): with profiling at
                                  level 2, fit time - (xp + atb + pow + stream + allreduce) is the replicated small-matrix chain   */
    int64_t stream_launches;
    /* FastIca: how the last fit ran */
    int64_t ica_redo;          /* 0: the optimistic run stood (two-product subspace iteration; for fp32 data of >= 256 padded features the
                                  covariance from the split-product Gram kernel); 1: redone with the residual-controlled iteration
                                  on the fp64-MFMA covariance (failed residual verdict, or kept eigenvalues spread over > 1 decade)  */
    int64_t ica_gram_split;    /* 1: the covariance that reached the result came from the split-product Gram kernels              */
    int64_t means_folded;      /* 1: the column means of the last fit were gathered INSIDE its first pass over X (RandomizedPca: the
                                  first fused power-iteration pass; FastIca: the split-product Gram pass), 0: a means pass of its own */
    int64_t eigh_redo;         /* RandomizedPca: 1: the small eigen-solve of the last fit (B B^T, order l) was repeated with the Jacobi solver --
                                  wanted eigenvalues too close, relative to the largest, for the two-stage solver's vectors; the passes over
                                  X were NOT repeated (round 6; until then this verdict redid the whole fit on the robust path)        */
} petal_stats;

/* ---- context ------------------------------------------------------------------------------- */
/* `stream`: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream) or NULL to let
 * the ctx create its own.  Device-resident inputs (space = 1) are read on that stream: the caller makes sure whatever
 * produces them has finished or is ordered before the call (the Python mirror synchronises torch's stream itself);
 * every entry point returns after its own work is complete. */
int         petal_ctx_create(int device, void* stream, petal_ctx** out);
void        petal_ctx_destroy(petal_ctx* ctx);
const char* petal_last_error(const petal_ctx* ctx);
const char* petal_version(void);
int         petal_ctx_set_collective(petal_ctx* ctx, petal_allreduce_fn fn, void* user, int rank, int world_size);
/* Built-in collective ("RCCL ncclAllReduce on one communicator"): rank 0 obtains a 128-byte ncclUniqueId,
 * the host distributes it (MPI / torch.distributed / a file), then EVERY rank calls petal_ctx_init_rccl collectively;
 * from then on the ctx all-reduces with ncclAllReduce on its own stream, no host callback in the loop.  RCCL is bound
 * with dlopen at the first call (the copy already loaded in the process wins), error 3 when it cannot be found. */
int         petal_rccl_unique_id(void* out128);
int         petal_ctx_init_rccl(petal_ctx* ctx, const void* unique_id128, int rank, int world_size);
/* What the collective of this ctx is: *kind = 0 none, 1 a caller's hook (petal_ctx_set_collective), 2 the built-in RCCL communicator;
 * for the built-in one, what RCCL ITSELF reports -- ncclCommCount / ncclCommCuDevice / ncclCommUserRank (-1 where unavailable) -- so a
 * scaling record can show that the communicator spanned N ranks (rank / world_size as the ctx was told them come back too). */
int         petal_ctx_collective_info(const petal_ctx* ctx, int* kind, int* rank, int* world_size, int* comm_count, int* comm_device,
                                      int* comm_rank);
/* profiling: 0 off; 1 = bracket ONE launch per fit with hipEvents (the kernel kind -- K1, K2, the FastICA step, the all-reduce --
 * and the launch index within the kind rotate from fit to fit, so K fits sample every launch position of every kind; an event
 * pair is a ~5 us bubble in the stream); 2 = every launch.
 * petal_stats.*_ms / *_launches count the bracketed launches only. */
int         petal_ctx_set_profiling(petal_ctx* ctx, int profiling);
/* How the two X-streaming GEMM kernels of fp32 fits form their products (results agree to fp32 accumulation noise):
 *   PETAL_GEMM_SPLIT_BF16X3 (default): every fp32 operand is split exactly into three bf16 pieces and the product is the
 *     sum of the six piece products of weight >= 2^-16, on the bf16 matrix cores with fp32 accumulation (dropped terms
 *     <= 2^-24 relative, below one fp32 rounding); 2.7x less matrix-pipe time, the kernels become HBM-bound;
 *     RandomizedPca runs OPTIMISTICALLY in this mode: the sketch matrix and the re-based iterates are taken on two bf16 planes (16
 *     significant bits), which makes the fused one-pass power iteration possible, and a verdict formed from the spectrum the fit
 *     found sends heavy-tailed data back through the pipeline with three-plane operands (petal_stats.rpca_redo = 1: twice the time);
 *   PETAL_GEMM_FP32_MFMA: v_mfma_f32_16x16x4_f32, exact fp32 products.  Env PETAL_GEMM=fp32 selects it at ctx creation;
 *   PETAL_GEMM_SPLIT_BF16X3_EXACT: the split-product kernels with three-plane operands from the start -- for callers who know their
 *     spectra decay slowly (no optimistic run, no redo).  Env PETAL_GEMM=bf16x3-exact. */
#define PETAL_GEMM_SPLIT_BF16X3 0
#define PETAL_GEMM_FP32_MFMA 1
#define PETAL_GEMM_SPLIT_BF16X3_EXACT 2
int         petal_ctx_set_gemm_mode(petal_ctx* ctx, int mode);
/* Options of a ctx.  Every switch that selects WHICH arithmetic or kernel form a fit runs is an option of the ctx: its default comes
 * from the environment variable named with it, read ONCE at petal_ctx_create; afterwards only petal_ctx_set_option changes it (no
 * library call reads the environment).  Options 0 - 8 change numerics within the documented parity bar; 9 - 13 are test / tooling aids.
 * Unknown option: PETAL_INVALID_INPUT. */
#define PETAL_OPT_TWO_PLANE_OPERANDS 0   /* (1) optimistic first run of an fp32 RandomizedPca fit: sketch matrix and re-based iterates on two bf16
                                            planes behind the spectrum verdict; 0 = three planes throughout            PETAL_NO_P2=1 -> 0 */
#define PETAL_OPT_TWO_PLANE_OMEGA 1      /* (1) ... the sketch matrix alone                                             PETAL_NO_P2_OMEGA=1 -> 0 */
#define PETAL_OPT_TWO_PLANE_ITERATE 2    /* (1) ... the re-based iterates alone                                         PETAL_NO_P2_ITERATE=1 -> 0 */
#define PETAL_OPT_STEERING_PASSES 3      /* (1) every pass before the last one rounds Xc and z to 16 bits as well        PETAL_NO_POW3_FAST=1 -> 0 */
#define PETAL_OPT_FUSED_PASS 4           /* (1) Y' = Xc^T (Xc P) in one pass over X where the kernel exists              PETAL_NO_POW3=1 -> 0 */
#define PETAL_OPT_FUSED_PASS_MIN_ROWS 5  /* (8192) fewer rows: the two GEMM kernels                                      PETAL_POW3_MIN_ROWS */
#define PETAL_OPT_VERDICT_THRESHOLD 6    /* (4e-6) the verdict redoes a fit whose estimated component error exceeds it   PETAL_P2_VERDICT_THR */
#define PETAL_OPT_MEANS_FOLD_ROWS 7      /* (200000) single-rank fp32 fits gather the column means inside their first pass from this many
                                            rows on; negative: never                         PETAL_MEANS_FOLD_ROWS, PETAL_NO_MEANS_FOLD=1 -> -1 */
#define PETAL_OPT_GRAM_SPLIT 8           /* (1) FastICA whitening of fp32 data: optimistic split-product covariance; 0 = fp64 products
                                                                                                                         PETAL_NO_GRAM3=1 -> 0 */
#define PETAL_OPT_GRAM_SPLIT_HOOK 9      /* (0) test hook: petal_gemm_atb sends Gram products to the split-product Gram kernel  PETAL_GRAM_SPLIT=1 */
#define PETAL_OPT_D2H_KERNEL 10          /* (1) small results leave through a copy kernel into the pinned ring; 0 = hipMemcpyAsync
                                                                                                                         PETAL_D2H_MEMCPY=1 -> 0 */
#define PETAL_OPT_ROW_PAD 11             /* (1) copied inputs get 128 B of row padding when the natural pitch is a multiple of 1 KiB
                                                                                                                         PETAL_NO_ROW_PAD=1 -> 0 */
#define PETAL_OPT_EIGH_JACOBI 12         /* (0) symmetric eigenproblems go straight to the Jacobi solvers                 PETAL_EIGH_JACOBI=1 */
#define PETAL_OPT_POISON 13              /* (0) every workspace block is filled with NaN patterns when handed out          PETAL_POISON=1 */
#define PETAL_OPT_FORCE_COLLECTIVE 14    /* (0) a one-rank ctx with a collective installed still takes the sharded code path
                                                                                                                         PETAL_FORCE_COLLECTIVE=1 */
int         petal_ctx_set_option(petal_ctx* ctx, int option, double value);
int         petal_ctx_get_option(const petal_ctx* ctx, int option, double* value);
int         petal_get_stats(const petal_ctx* ctx, petal_stats* out);

/* ---- Pca<A>::fit / fit_transform  (src/pca.rs:116-122, 153-167, 195-231) ---------------------- */
/* components: k x d, means: d, singular: k, total_variance: 1 (all host, dtype of x).
 * y_out (nullable): n x k = U[:, :k] * sigma  (transform_with_u, src/pca.rs:758-779). */
int petal_pca_fit(petal_ctx* ctx, const petal_matrix* x, int64_t k, int centering,
                  void* components, void* means, void* singular, void* total_variance,
                  const petal_matrix* y_out);

/* ---- RandomizedPca<A,R>::fit / fit_transform  (src/pca.rs:430-436, 467-481, 509-550, 668-718) -- */
/* omega: HOST, d x (k + n_oversample) row-major, dtype of x: the StandardNormal matrix the crate
 * draws at src/pca.rs:701-705 from the model's RNG (the facade draws it and passes it in).
 * Reference constants: n_oversample = 10 (src/pca.rs:679), n_iter = 7 (src/pca.rs:680). */
int petal_rpca_fit(petal_ctx* ctx, const petal_matrix* x, int64_t k, int64_t n_oversample, int64_t n_iter,
                   int centering, const void* omega,
                   void* components, void* means, void* singular, void* total_variance,
                   const petal_matrix* y_out);

/* ---- transform / inverse_transform  (src/pca.rs:726-750, 788-811; src/ica.rs:120-131) --------- */
/* y_out = (x - means) . components^T   (means ignored when centering == 0) */
int petal_transform(petal_ctx* ctx, const petal_matrix* x, const void* components, const void* means,
                    int64_t k, int64_t d, int centering, const petal_matrix* y_out);
/* x_out = y . components + means */
int petal_inverse_transform(petal_ctx* ctx, const petal_matrix* y, const void* components, const void* means,
                            int64_t k, int64_t d, int centering, const petal_matrix* x_out);

/* ---- FastIca<A,R>::fit / fit_transform  (src/ica.rs:105-112, 147-157, 167-221) ---------------- */
/* n_components == 0 -> min(n, d) as the crate does (src/ica.rs:173).  w_init: HOST nc x nc row-major
 * StandardNormal draw (src/ica.rs:210-214).  Reference constants tol = 1e-4, max_iter = 200
 * (src/ica.rs:216).  components: nc x d, means: d (host).  y_out (nullable): n x nc sources. */
int petal_fastica_fit(petal_ctx* ctx, const petal_matrix* x, int64_t n_components, double tol, int64_t max_iter,
                      int mode, const void* w_init, void* components, void* means, int64_t* n_iter,
                      const petal_matrix* y_out);

/* ---- the crate-private kernels that carry known-answer tests ----------------------------------- */
/* ica_par (src/ica.rs:319-361): x1 is nc x n (whitened); w_init / w_out HOST nc x nc. */
int petal_ica_par(petal_ctx* ctx, const petal_matrix* x1, double tol, int64_t max_iter, int mode,
                  const void* w_init, void* w_out, int64_t* n_iter);
/* symmetric_decorrelation (src/ica.rs:363-381): w, out HOST nc x nc (dtype). */
int petal_symmetric_decorrelation(petal_ctx* ctx, const void* w, int64_t nc, int32_t dtype, int mode, void* out);
/* logcosh (src/ica.rs:383-398): x is r x c; g_out r x c = tanh(x); gprime_out HOST r = mean(1-g^2). */
int petal_logcosh(petal_ctx* ctx, const petal_matrix* x, const petal_matrix* g_out, void* gprime_out);
/* svd_flip (src/pca.rs:815-850): flips columns of u (n x m) and rows of vt (m' x d) in place. */
int petal_svd_flip(petal_ctx* ctx, const petal_matrix* u, const petal_matrix* vt);

/* ---- the two X-streaming GEMM kernels of the power iteration, callable on their own ---------------
 * (parity tests with exact-integer data, and the roofline measurement of bench.py).
 * z_out[n x N] = (x[n x K] - mu) . p + bias     (src/pca.rs:707, 714, 745, 806: `input.dot(&pl)`)
 *   mu (nullable): HOST K values; p: HOST K x N row-major; bias (nullable): HOST N values; all in x's dtype. */
int petal_gemm_xp(petal_ctx* ctx, const petal_matrix* x, const void* mu, const void* p, int64_t N, const void* bias,
                  const petal_matrix* z_out);
/* c_out[M x N] (HOST, row-major, fp64) = (a[n x M] - mu_a)^T . (b[n x N] - mu_b)   (src/pca.rs:711, 681:
 * `input.t().dot(&pl)`, `q.t().dot(input)`);  mu_a / mu_b nullable HOST vectors in a's dtype.
 * b == NULL means b = a (Gram matrix / covariance, src/ica.rs:189 equivalent). */
int petal_gemm_atb(petal_ctx* ctx, const petal_matrix* a, const void* mu_a, const petal_matrix* b, const void* mu_b,
                   double* c_out);

/* One power iteration of the range finder as ONE pass over x (src/pca.rs:711 + 714: `input.t().dot(&pl)` of `input.dot(&pl)`):
 *   y_out[K x N] (HOST, row-major, fp64) = (x - mu)^T . ((x - mu) . p),   z_out (nullable) [n x N] = (x - mu) . p
 * with p taken as the sum of its two leading bf16 pieces (16 significant bits: exact for the small integers of the parity tests).
 * *fused_out (nullable) = 1 when the fused kernel ran (fp32 data, split-product mode, K = 512, N <= 80), 0 when the two GEMM
 * kernels formed the same products one after the other (every other shape / mode: p then counts in full). */
int petal_power_pass(petal_ctx* ctx, const petal_matrix* x, const void* mu, const void* p, int64_t N, double* y_out,
                     const petal_matrix* z_out, int* fused_out);

#ifdef __cplusplus
}
#endif
#endif /* PETAL_HIP_H */
