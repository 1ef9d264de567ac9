/*
 * aa_hip.h -- C ABI of libaa_hip.so: the MI355X (gfx950) implementation of the
 * archetypal-analysis / GPNH-convex-coding inner solver of `convex_dim_red`.
 *
 * The reference (azedarach/matrix-factorization-case-studies) has NO native/FFI layer:
 * its hot path is numba-JIT Python behind the module surface of
 * src/convex_dim_red.  This header is therefore the boundary a maintainer would
 * bind with ctypes from that package; every entry point cites
 * the reference code it replaces (paths relative to src/convex_dim_red/).
 *
 * Conventions
 *   - every function returns 0 on success, a negative AA_ERR_* code otherwise;
 *     aa_last_error() returns a message for the calling thread's last failure;
 *   - nothing throws across the ABI; plain pointers and sizes only;
 *   - host buffers are caller-owned, row-major, float64 unless said otherwise,
 *     and are only read or filled during the call; device memory is owned by
 *     the context; a context is NOT thread-safe (one per estimator call);
 *   - "n" = samples (rows of X), "p" = features, "k" = components (k <= 64).
 */
#ifndef AA_HIP_H
#define AA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define AA_OK            0
#define AA_ERR_ARG      -1   /* bad argument / unsupported size                 */
#define AA_ERR_HIP      -2   /* a HIP runtime call failed (no GPU, OOM, fault)   */
#define AA_ERR_STATE    -3   /* call order violated (e.g. update before set_data)*/
#define AA_ERR_COMM     -4   /* RCCL failure                                     */

#define AA_F32 0             /* X stored + contracted in float32 (MFMA f32)      */
#define AA_F64 1             /* X stored + contracted in float64 (reference dtype)*/

#define AA_FORM_DATA   0     /* ArchetypalAnalysis: data matrix X (n x p)        */
#define AA_FORM_KERNEL 1     /* KernelAA: kernel matrix K (n x n)                */

#define AA_MAX_K 64

typedef struct aa_ctx aa_ctx;   /* opaque */

/* Parameters of the per-sample simplex QP solver.
 * Replaces the keyword arguments of quad_simplex_spg (spg.py:286-291) as forwarded
 * by _update_kernel_aa_weights (archetypal_analysis.py:372-383) and
 * _update_gpnh_weights (gpnh_convex_coding.py:257-268). */
typedef struct {
    double gamma;
    int    memory;
    double sigma_one, sigma_two, lambda_min;
    double alpha0, alpha_min, alpha_max;
    double epsilon_one, epsilon_two;
    int    max_iterations, max_feval;
} aa_qp_params;

/* Parameters of the dictionary SPG solver: keyword arguments of spg()
 * (spg.py:46-51).  alpha0 < 0 means "None" (derive from the first projected
 * gradient step, spg.py:178-189). */
typedef struct {
    double gamma;
    int    memory;
    double sigma_one, sigma_two, lambda_min;
    double alpha0, alpha_min, alpha_max;
    double epsilon_one, epsilon_two;
    int    use_infinity_norm;
    int    max_iterations, max_feval;
} aa_spg_params;

#define AA_SPG_FLAG_CONVERGED      1   /* spg.py:259-266                         */
#define AA_SPG_FLAG_LAMBDA_MIN     2   /* warning at spg.py:225-229              */
#define AA_SPG_FLAG_MAX_FEVAL      4   /* warning at spg.py:272-276              */
#define AA_SPG_FLAG_MAX_ITER       8   /* warning at spg.py:278-281              */
#define AA_SPG_FLAG_PROJ_UNCONV   16   /* internal: projection pass cap reached  */

typedef struct {
    double f;          /* spg() return value 2: f at the returned point          */
    int    n_iter;     /* spg() return value 3: 0-based index of the last pass   */
    int    n_feval;    /* spg() return value 4                                   */
    int    flags;      /* AA_SPG_FLAG_*                                          */
    double res_norm;   /* ||P(x - g) - x||_2 at exit                             */
} aa_spg_stats;

typedef struct {
    long   total_passes;   /* sum over samples of SPG loop passes                */
    int    max_passes;     /* largest per-sample pass count                      */
    int    reserved;
} aa_qp_stats;

/* ------------------------------------------------------------------ misc */
const char *aa_last_error(void);
int aa_version(void);
int aa_device_count(int *count);
/* Process-wide tuning knobs (results are identical up to rounding for every setting):
 *   "row_local_variant" -1..9  float32 row-local GEMM: -1 (default) chosen by size (from 32768 rows
 *                               per GPU: 9 for k <= 32, 8 for k <= 64; else 4), 0 direct, 1 wave-private
 *                               LDS, 2..7 block-tiled (2: 64-column tiles, 3: 64 double-buffered,
 *                               4: 128, 5: 32 double-buffered, 6: 128 double-buffered, 7: 32),
 *                               8 wave-streaming (wave-private X tiles, shared B slabs, register
 *                               staged), 9 wave-streaming with both operands by LDS-DMA and float64
 *                               sums of 32-column fp32 pieces
 *   "row_local_acc64"   0..2   float32 row-local pass: 1 (default) = every fp32 accumulation chain ends
 *                               after 32 columns and the pieces are summed in float64 (block-tiled
 *                               kernels and variant 9; what keeps long float32 runs on the float64
 *                               trajectory); 2 = also in variant 8 (half speed);
 *                               0 = one fp32 chain per column chunk (round 2)
 *   "row_local_split"   0|1    1 (default): block-tiled kernels split the contraction over column
 *                               chunks when there are few row blocks
 *   "row_local_ring"    0|8..12 variant 9: LDS pieces in a wave's ring (default 8; 0: what LDS allows)
 *   "row_local_nt"      0|1    variant 9: non-temporal hint on the X stream (default 0)
 *   "f64_mfma"          0..3   float64 data: pass kernels on the f64 matrix cores (1, default: the
 *                               row-local one wave-streaming from 32768 rows per GPU, else
 *                               block-tiled; 2 / 3 force either) or on the f64 VALU (0)
 *   "row_local_waves"   0..16  waves per block of variant 8 (0: one block per CU)
 *   "proj_mode"         0|1    column simplex projection: 0 candidate lists, 1 iterative full
 *                               passes (also the fallback of a rank whose list overflows)
 *   "proj_small"        0|1    1 (default): columns of at most 8192 entries (single rank) find their
 *                               projection threshold in one kernel instead of four
 *   "proj_list_cap"     1..2048 multi-rank: most candidates per rank and column that travel in
 *                               the list all-reduce of a projection (effective: min(this,
 *                               2048 / ranks), so that the union fits the solver's LDS)
 *   "use_graph"         0|1    1: aa_outer_iterations captures two outer iterations in a
 *                               hipGraph and replays it (single rank, data form, one SPG
 *                               iteration per dictionary update, >= 8 iterations).  Default 0:
 *                               bit-identical and measured neutral on ROCm 7.2
 *   "qp_pass_cap"       >= 1   SPG passes a sample spends in the lane-per-sample QP kernel
 *                               before it moves to the wave-per-sample kernel
 *   "qp_refill_min"     1..64  idle lanes of a wave that trigger pulling new samples (default
 *                               64: a wave works off 64 samples at a time)
 *   "qp_sort"           0|1    1 (default): the lane-per-sample kernel takes the samples in the
 *                               order of their pass counts in the previous weights update,
 *                               longest first
 *   "qp_waves"          >= 1   most waves (64 samples each) the lane-per-sample kernel runs
 *                               at once; samples beyond that are pulled in as lanes free up
 *   "qp_overlap_tail"   0|1    1: the wave-per-sample kernel of the stragglers runs on a side
 *                               stream while Z'X is accumulated (float32 data); the rows it
 *                               changes enter Z'X as a rank-m correction.  Default 0: measured
 *                               neutral, the stragglers run 2x slower next to the GEMM
 *   "qp_mode"           0..4   0 (default), k <= 32: four lanes per sample in the matrix-core operand
 *                               layout (16 samples per wave) followed by the wave-per-sample kernel
 *                               for the stragglers; one lane per sample when max_iterations <= 4
 *                               (nothing diverges); k > 32: one wave per sample.
 *                               1: one wave per sample; 2: lane-per-sample + wave; 3: row kernel
 *                               (16 lanes per sample, samples run to completion); 4: four lanes per
 *                               sample + wave
 *   "qp_quad_cap"       >= 0   SPG passes after which the four-lane kernel parks a sample for the
 *                               wave-per-sample kernel (default 0: 32 from 65 536 samples per GPU,
 *                               24 below)
 *   "qp_quad_waves"     >= 1   most waves (16 samples each) of the four-lane kernel (default 8192:
 *                               up to 131 072 samples every wave takes one batch)
 *   "qp_quad_occ"       2..4   register budget of the four-lane kernel in waves per SIMD (default 3)
 *   "qp_quad_refill"    1..16  idle sample slots of a wave that trigger a refill when a wave
 *                               takes several batches (default 16)
 *   "qp_row_waves"      >= 1   most waves the row kernel runs with (default 2048: 2 per SIMD)
 *   "qp_row_hot"        >= 0   SPG passes after which the wave of a sample raises its issue
 *                               priority (also when the previous update needed twice as many)
 *   "qp_row_chunk"      0..4096 0 (default): the row kernel's waves own fixed, interleaved slices of
 *                               the longest-first sample list (no queue); > 0: a global queue,
 *                               this many tickets per atomic
 *   "qp_row_long"       0..63  samples that needed at least this many passes in the previous
 *                               update are solved by the wave-per-sample kernel on a side stream,
 *                               concurrently with the row kernel (default 0: no side stream; measured
 *                               slower -- the two kernels take each other's issue slots)
 *   "qp_row_cap"        >= 1   passes after which the row kernel hands a sample to the
 *                               wave-per-sample kernel (default: never)
 *   "qp_tail_cap"       >= 0   with qp_overlap_tail: only samples beyond this many passes go to the
 *                               side stream (default 96; 0: all parked ones)
 *   "qp_wave_blocks"    1..8192 grid of the wave-per-sample continuation launch (default 1024)
 *   "qp_live"           0|1    1: the four-lane kernel hands parked samples to a consumer launch of the
 *                               wave-per-sample kernel that is resident beside it on CUs of its own
 *                               (bit-identical results; default 0: measured slower);
 *                               "qp_live_blocks" 1..128 = CUs given to the consumers (default 48),
 *                               "qp_live_occ" 2..4 = register budget of the four-lane kernel beside them
 *   "proj_res_side"     0|1    1 (default): the residual projection of a one-iteration dictionary SPG
 *                               (convergence flags only) runs on a side stream beside the weights QP
 *   "outer_nosync"      0|1    measurement only (tools/interleave_probe.py): aa_outer_iterations called
 *                               WITHOUT a cost buffer returns with its work in flight
 *   "fuse_finalize"     0|1    1 (default): fewer, fatter launches in the dictionary update (set-up
 *                               kernel, scalar stages inside the finalize kernels, two-launch line
 *                               search, x update inside the gradient kernel)
 * Round 4 -- none of these changes a bit of any result (tests/test_gpu_longrun.py:
 * test_schedule_knobs_do_not_change_a_bit):
 *   "qp_fused_order"    0|1    1 (default): the sample order of the NEXT weights update is formed by extra
 *                               blocks of this update's continuation launch instead of two launches in
 *                               front of the four-lane kernel
 *   "qp_wave_lazy"      0|1    1 (default): the wave-per-sample kernel decides the stopping test of a pass at
 *                               the top of the next one and skips it when <d, d> proves it negative
 *   "qp_quad_lazy"      0|1    the same in the four-lane kernel (default 0: no gain there)
 *   "fin_in_last"       0|1    1 (default): the two reductions of a projection are finalized by the last
 *                               block of their pass (write-through partials), not by a launch of their own
 *   "setup_in_grad"     0|1    1 (default): the dictionary update's set-up block is block 0 of its first
 *                               gradient launch
 *   "gram_side"         0|1    Z'Z of the refresh after a weights update on the side stream (default 0)
 *   "grad_side"         0|1    1 (default): the tail of a one-iteration dictionary SPG (g_new, x += lambda d,
 *                               BB stage, residual projection) on the side stream beside the weights QP
 *   "pack_comm"         0|1    multi-rank, 1 (default): four small reductions ride in the tail of the
 *                               all-reduce that follows them (10 collectives per outer iteration, not 14)
 *   "pq_mfma"           0|1    1 (default): the two wide Grams of the line search on the f64 matrix cores (another
 *                               summation order than the LDS / VALU kernel: results differ at rounding level)
 *   "proj_small"        0|1|2  1 (default): columns of <= 8192 rows are projected by one block each;
 *                               2: up to 16 384 rows (slower at 12 500) */
int aa_set_option(const char *name, int value);

/* -------------------------------------------- stateless ops (unit-test surface) */

/* Euclidean projection of every row of `in` (rows x cols) onto the unit simplex.
 * Replaces simplex_project_rows (simplex_projection.py:40-47) /
 * simplex_project_vector (:13-27). */
int aa_simplex_project_rows(int device, const double *in, double *out, long rows, long cols);

/* Z[t] = argmin_{z in simplex} 0.5 z'Az + b_t'z,  b_t[j] = -B[j*stride_j + t*stride_t],
 * started from Z0[t]; t = 0..n-1.  A: k x k.  iters (nullable): loop passes per sample.
 * Replaces _gu_update_kernel_aa_weights (archetypal_analysis.py:344-366; stride_j = n,
 * stride_t = 1) and _gu_update_gpnh_weights (gpnh_convex_coding.py:229-251; stride_j = 1,
 * stride_t = k), i.e. n calls of quad_simplex_spg (spg.py:286-398). */
int aa_quad_simplex_spg_batch(int device, const double *A, const double *B,
                              long stride_j, long stride_t,
                              const double *Z0, double *Zout, long n, int k,
                              const aa_qp_params *params, int *iters);

/* ------------------------------------------------------ resident solver context */
int aa_ctx_create(aa_ctx **ctx, int device, int dtype);
int aa_ctx_destroy(aa_ctx *ctx);

/* Multi-GPU (one process per GPU, rows of X sharded): rank 0 calls
 * aa_comm_get_unique_id and hands the 128 bytes to the other ranks out of band;
 * every rank then calls aa_ctx_comm_init.  Collectives (RCCL all-reduce, sum/max)
 * run only on the small k x k / k x p Gram products and packed scalars. */
int aa_comm_get_unique_id(void *id128);
int aa_ctx_comm_init(aa_ctx *ctx, const void *id128, int rank, int world);
/* The second transport: a one-shot peer-to-peer all-reduce for the path's
 * latency-bound messages (<= 1 MiB; larger ones go in pieces) -- every rank stores its vector
 * straight into a receive buffer of every peer (mapped with hipIpcOpenMemHandle), raises a flag per
 * block, and reduces the world slots in rank order once the peers' flags are up: one kernel per rank
 * and collective, identical bits on every rank, no RCCL (csrc/comm.hip).  Every rank calls
 * aa_ctx_p2p_export (allocates the buffer; 64-byte IPC handle out), the handles travel out of band
 * like the RCCL unique id, then every rank calls aa_ctx_p2p_init with all `world` handles in rank
 * order; all collectives of the context then use this transport.  At most 8 ranks (one node). */
int aa_ctx_p2p_export(aa_ctx *ctx, int world, void *handle64);
int aa_ctx_p2p_init(aa_ctx *ctx, const void *handles, int rank, int world);
/* all-reduce of a small host vector through the context's communicator (used by
 * bench.py for the barrier + max-over-ranks timing); op: 0 = sum, 1 = max. */
int aa_ctx_allreduce_host(aa_ctx *ctx, double *buf, int count, int op);

/* Upload this rank's row block of the data matrix (form DATA: n x p) or kernel
 * matrix (form KERNEL: n x n, p == n; single rank only).  host_dtype: AA_F32/AA_F64
 * element type of `X` (converted to the context dtype on upload).  The matrix stays
 * resident across restarts.  `n_global`/`row_offset` describe the shard.
 * Replaces the `X`/`K` argument of _iterate_aa / _iterate_kernel_aa
 * (archetypal_analysis.py:534, :399) and `X` of _iterate_gpnh_convex_coding
 * (gpnh_convex_coding.py:282). */
int aa_set_data(aa_ctx *ctx, const void *X, int host_dtype, long n, long p, long ld,
                int form, long n_global, long row_offset);
/* ||X||_F^2 (form DATA; = trace(X X'), archetypal_analysis.py:552 without the n x n
 * temporary) or trace(K) (form KERNEL, :412), summed over all ranks. */
int aa_data_trace(aa_ctx *ctx, double *trace);

/* Factors: C (k x n, this rank's COLUMN block k x n_local, row-major with leading
 * dimension ldc), Z (n_local x k), alpha (k).  Replaces the weights/dictionary/alpha
 * arguments and return values of _iterate_aa (archetypal_analysis.py:534-536,669). */
int aa_set_state(aa_ctx *ctx, int k, const double *C, long ldc, const double *Z, const double *alpha);
int aa_get_state(aa_ctx *ctx, double *C, long ldc, double *Z, double *alpha);
int aa_set_alpha(aa_ctx *ctx, const double *alpha);

/* Recompute every Gram product from (X, C, Z, alpha) and return the cost
 * 0.5 (tr XX' - 2 tr(D C XX' Z) + tr(D Z'Z D C XX' C')) / n.
 * Replaces archetypal_analysis.py:543-558 (:403-418 for the kernel form). */
int aa_prepare(aa_ctx *ctx, double *cost);
/* Cost from the current Gram products and alpha (archetypal_analysis.py:623-627). */
int aa_cost(aa_ctx *ctx, double *cost);
/* k x k Gram products for the host-side scale-factor update
 * (archetypal_analysis.py:243-258): ZtZ, C XX' C', C XX' Z, and the data trace. */
int aa_get_grams(aa_ctx *ctx, double *ZtZ, double *CKCt, double *CKZ, double *trace);
/* Override the dictionary-update inputs with caller-supplied values:
 * KZ = XX'Z or KZ (n_local x k), ZtZ (k x k), trace.  Needed to mirror the
 * test-visible _update_kernel_aa_dictionary(K, C, alpha, trace_K, KZ, ZtZ)
 * (archetypal_analysis.py:304) / _update_aa_dictionary (:324). */
int aa_set_dictionary_inputs(aa_ctx *ctx, const double *KZ, const double *ZtZ, double trace);

/* SPG update of the dictionary with row-simplex constraint, then refresh of
 * CX, C XX', C XX' C', C XX' Z.  Replaces _update_aa_dictionary
 * (archetypal_analysis.py:324-341, cost :261-270, gradient :293-301), or
 * _update_kernel_aa_dictionary (:304-321, :273-290) in the kernel form, plus the
 * refresh at :618-621 (:484-486). */
int aa_dictionary_update(aa_ctx *ctx, const aa_spg_params *params, aa_spg_stats *stats);
/* Per-sample QP update of the weights, then refresh of Z'Z, X'Z, XX'Z, C XX' Z.
 * Replaces _update_kernel_aa_weights (archetypal_analysis.py:369-396) plus the
 * refresh at :640-643 (:501-503). */
int aa_weights_update(aa_ctx *ctx, const aa_qp_params *params, aa_qp_stats *stats);

/* Restarts (bin/run_hadisst_aa.py:158-172 fits n_init models on the same data):
 * `ctx` takes the resident data matrix of `owner` (same device, same dtype, single rank) without
 * a copy -- the solver only reads it -- so several contexts, each with its own factors, streams
 * and scratch, work on ONE copy of X concurrently.  `owner` must outlive `ctx` or give it new
 * data first (aa_set_data / aa_share_data release the alias, never the owner's memory). */
int aa_share_data(aa_ctx *ctx, const aa_ctx *owner);

/* KernelAA on the implicit linear kernel K = X X' (SURVEY 8(f4)): with `on` != 0 the resident
 * DATA matrix X (n x p) stands in for the n x n kernel matrix of _iterate_kernel_aa
 * (archetypal_analysis.py:399-531), which is never formed: every product with K runs as two
 * skinny passes over X, exactly as in the data form, and the dictionary gradient follows the
 * kernel form's convention (divided by n_components, _kernel_aa_dictionary_gradient :281-288,
 * where _aa_dictionary_gradient :291-300 divides by n_samples) -- the one place where the two
 * forms of the reference differ for K = X X'.  Reset by aa_set_data. */
int aa_set_linear_kernel(aa_ctx *ctx, int on);
/* KernelAA on an implicit RBF kernel (archetypal_analysis.py:673-910 with the n x n kernel
 * matrix K_ij = exp(-gamma ||x_i - x_j||^2) NEVER formed): uploads the n x p feature matrix (float64) and
 * switches the context to the kernel form of the algorithm; every product C K / K Z is one fused
 * distance + exp + multiply pass over the features (csrc/kernels_gemm.hip: k_rbf_kv), tr K = n, and
 * aa_distance_column returns sqrt(2 - 2 K_ij).  Single rank, float64 context.  Everything downstream
 * (aa_set_state, aa_prepare, aa_iterate, aa_get_state ...) is the explicit kernel form's. */
int aa_set_rbf_features(aa_ctx *ctx, const double *X, long n, long p, long ld, double gamma);

/* Driver-side preprocessing on the device (bin/run_hadisst_aa.py:133-146 weight_and_flatten_data,
 * :196-209 NaN-column removal and training / validation split; the same steps in
 * bin/run_jra55_pca_gpnh.py).  `raw` is the flattened field, n_total x p_full, row-major
 * (float64 or float32, NaN = missing), `col_weight[p_full]` the latitude weight of every flattened
 * column (NULL: 1).  Columns with a NaN in ANY of the n_total rows are dropped, the others are
 * multiplied by their weight, and rows [row0, row0 + n) become the context's data matrix exactly
 * as after aa_set_data (data form).  valid[p_full] receives 1 / 0 per column, *p_valid the
 * number of columns kept.  The raw field crosses PCIe once; weighting, the NaN scan and the
 * compaction run on the GPU. */
int aa_set_data_weighted(aa_ctx *ctx, const void *raw, int host_dtype, long n_total, long p_full,
                         long ld, const double *col_weight, long row0, long n,
                         unsigned char *valid, long *p_valid);
/* The resident data matrix back on the host as float64 (n x p, row stride ld): tests, and the
 * few host-side uses of the data the estimators have (GPNH 'random' initialisation). */
int aa_get_data(aa_ctx *ctx, double *out, long ld);

/* The alternating loop of _iterate_aa / _iterate_kernel_aa (archetypal_analysis.py:586-663,
 * :455-524) for delta == 0, with the monotonicity check (:167-174) and the stopping rule
 * (:177-197, :663) evaluated ON THE DEVICE after every iteration: the host enqueues
 * `check_every` iterations at a time and reads one small status record per batch instead of
 * synchronising three times per iteration (delta != 0 included: the k-vector scale-factor
 * update of :243-258 is one more small kernel).  When the rule fires at iteration j the factors of
 * iteration j are kept (a conditional device-side snapshot), the iterations that were already
 * enqueued behind it are discarded, and the context is left consistent with iteration j. */
typedef struct {
    int    max_outer;          /* archetypal_analysis.py:586 `max_iterations`             */
    double tolerance;
    int    criterion;          /* 0: abs_delta_f, 1: rel_delta_f (:177-197)                */
    int    require_monotonic;  /* :167-174                                                 */
    double mono_tolerance;     /* tolerance of the monotonicity check: the reference uses
                                  `tolerance`; float32 data pass max(tolerance, the float32
                                  noise floor of the trace-form cost)                       */
    int    update_dictionary, update_weights;
    int    check_every;        /* iterations per host poll (>= 1)                          */
    double delta;              /* != 0 (with a scale-factor solver passed to aa_iterate): the
                                  scale-factor update of archetypal_analysis.py:590-609 opens
                                  every iteration, as a k-dimensional SPG in one small kernel  */
} aa_iter_params;

typedef struct {
    int    n_iter;             /* 0-based index of the last iteration (reference n_iter)   */
    int    converged;          /* the stopping rule fired                                  */
    int    error_stage;        /* 0 none, 1 / 2 / 3: cost increased after the dictionary / weights /
                                  scale-factor update                                        */
    int    error_iter;
    int    spg_flags;          /* OR of the dictionary SPG's AA_SPG_FLAG_* over iterations */
    int    reserved;
    double cost;               /* cost after the last iteration kept                       */
} aa_iter_stats;

/* cost0: cost of the prepared state (aa_prepare); costs: 2 * max_outer entries, filled for the
 * iterations kept.  Returns AA_OK also when error_stage != 0 (the caller raises). */
int aa_iterate(aa_ctx *ctx, const aa_iter_params *it, const aa_spg_params *spg,
               const aa_qp_params *qp, const aa_spg_params *scale_spg /* NULL: no scale factors */,
               double cost0, double *costs, aa_iter_stats *stats);

/* The alternating loop of _iterate_gpnh_convex_coding (gpnh_convex_coding.py:282-402) on the
 * device, same loop control as aa_iterate.  Per iteration: Z'X (reduce-over-rows GEMM), the
 * regularised normal equations (Z'Z/n + lambda_W GW) W' = Z'X/n (:213-226) solved by a Cholesky
 * factorisation in one small kernel (the reference calls lstsq on the same k x k system; when
 * the factorisation meets a pivot below 1e-13 of the largest diagonal entry the call returns
 * with stats->error_stage = 3 and the caller falls back to its host lstsq path), X W
 * (row-local GEMM), W'W, the GPNH penalty (:179-196), the cost (:317-330) and the n per-sample
 * QPs (:254-279).  Factors set with aa_gpnh_set_factors; the dictionary is read back with
 * aa_gpnh_get_dictionary. */
typedef struct {
    double lambda_W;
    aa_iter_params loop;
} aa_gpnh_params;
int aa_gpnh_iterate(aa_ctx *ctx, const aa_gpnh_params *params, const aa_qp_params *qp,
                    double *cost0, double *costs, aa_iter_stats *stats);
int aa_gpnh_get_dictionary(aa_ctx *ctx, double *Wt /* k x p, row-major */, long ld);

/* n_outer full outer iterations (dictionary, weights) without returning to the
 * caller; costs[2*i], costs[2*i+1] = cost after the dictionary / weights update of
 * iteration i.  delta == 0 only (no scale-factor update).  Replaces the loop body
 * archetypal_analysis.py:586-657 for benchmarking. */
int aa_outer_iterations(aa_ctx *ctx, int n_outer, const aa_spg_params *spg,
                        const aa_qp_params *qp, double *costs);

/* 0.5 ||X - Z diag(alpha) C X||_F^2 / n evaluated in residual form (no trace
 * cancellation); the reconstruction error reported by bench.py. */
int aa_reconstruction_cost(aa_ctx *ctx, double *cost);

/* archetypes = diag(alpha?) C X (k x p): archetypal_analysis.py:1144. */
int aa_get_archetypes(aa_ctx *ctx, double *CX, long ld);

/* FurthestSum support (furthest_sum.py:23-127 needs column j of the n x n
 * dissimilarity matrix, archetypal_analysis.py:95-100): d[i] =
 * sqrt(K_ii - 2 K_ij + K_jj) for this rank's rows i, K = XX' never formed
 * (form DATA) or the stored K (form KERNEL).  j is a GLOBAL row index. */
int aa_distance_column(aa_ctx *ctx, long j, double *d);
/* The whole FurthestSum selection (furthest_sum.py:23-127) on the device, single rank: running
 * distance sums and the candidate flags stay in device memory, every pick is one distance-column
 * kernel and one single-block step, the picked index is passed on in device memory -- no host
 * round trip per pick.  selected[k] out.  *tie != 0: at some pick the largest running sum was
 * shared by several candidates; the reference then takes the one its stable sorts left last, a rule
 * that depends on the history of the list -- the caller repeats the selection through
 * aa_distance_column and the host's list logic (convex_dim_red/furthest_sum.py).  Distances in the
 * arithmetic of aa_distance_column, sums updated in the same order: the same picks. */
int aa_furthest_sum(aa_ctx *ctx, int k, long start_index, const int *exclude, int n_exclude, int extra_steps,
                    int *selected, int *tie);

/* GPNH restarts side by side (the drivers' n_init loop, bin/run_jra55_pca_gpnh.py:112-138):
 * R independent fits of k components each share one set of device arrays (R k <= 64) and every launch
 * of an outer iteration -- restart r lives in component slots [r k, (r + 1) k).  Every restart gets
 * the bits it gets from aa_gpnh_iterate on its own.
 *   aa_gpnh_slots_begin   R empty slots; gp / qp as for aa_gpnh_iterate (both updates on)
 *   aa_gpnh_slots_load    start factors of a restart into slot r (W': k x p, leading dimension ld;
 *                         Z: n x k), its initial cost
 *   aa_gpnh_slots_run     n_iters outer iterations of every slot; status[R] out.  A slot stops by its
 *                         own stopping rule / monotonicity check / iteration cap; its factors of that
 *                         iteration are kept while the others go on
 *   aa_gpnh_slots_fetch   factors, cost record (2 (stop_iter + 1) values) and initial cost of a
 *                         stopped slot; the slot can then be loaded again
 *   aa_slots_end          (shared with the AA slots below) leaves the slot mode: the context must be
 *                         ended -- or destroyed -- before it is used for single fits again; a begin
 *                         call of either slot mode resets whatever an earlier, un-ended one left */
typedef struct {
    int stop, converged, error_stage, stop_iter;
    int not_spd;            /* the slot's normal equations were not positive definite */
    int iterations_run;     /* outer iterations this slot has been through since it was loaded */
} aa_slot_status;
int aa_gpnh_slots_begin(aa_ctx *ctx, int R, int k, const aa_gpnh_params *gp, const aa_qp_params *qp);
int aa_gpnh_slots_load(aa_ctx *ctx, int r, const double *Wt, long ld, const double *Z);
int aa_gpnh_slots_run(aa_ctx *ctx, int n_iters, aa_slot_status *status);
int aa_gpnh_slots_fetch(aa_ctx *ctx, int r, double *Wt, long ld, double *Z, double *costs, double *cost0);

/* The same for archetypal analysis (bin/run_hadisst_aa.py:149-174; archetypal_analysis.py:534-670 per
 * restart): R fits of k components in the component slots of one set of arrays (R k <= 32), production
 * settings only (data form, one SPG iteration per dictionary update, fewer than 65 536 samples,
 * k <= 16, single rank).  aa_slots_begin + aa_slots_load x R start the first restarts together;
 * aa_slots_run iterates every slot and reports its status (a stopped slot's factors of that iteration
 * are kept while the others go on); aa_slots_fetch returns a stopped slot's factors, cost record and
 * C X -- recomputed from the fetched C, or (carried != 0) as the loop carried it at the stopping
 * iteration, which is what aa_get_archetypes returns after an aa_iterate that stopped on the last
 * iteration of a batch; aa_slots_reload puts the next restart into the freed slot (its first dictionary
 * update is the cold one of a fit, the running slots keep the products they carry); aa_slots_finish
 * (optional, all slots stopped) restores the stopping-iteration factors and rebuilds the products;
 * aa_slots_end returns the context to single fits (from either slot mode: it also ends a run begun with
 * aa_gpnh_slots_begin).  status[r].not_spd carries the slot's SPG warning
 * flags (AA_SPG_FLAG_*).  Every restart gets the bits aa_iterate gives it on its own. */
int aa_slots_begin(aa_ctx *ctx, int R, int k, const aa_iter_params *loop, const aa_spg_params *spg,
                   const aa_qp_params *qp, const aa_spg_params *scale_spg /* loop->delta != 0 */);
int aa_slots_load(aa_ctx *ctx, int r, const double *C, long ldc, const double *Z, const double *alpha /* k, or NULL: ones */);
int aa_slots_run(aa_ctx *ctx, int n_iters, aa_slot_status *status);
/* a new restart into slot r of a RUNNING group (its previous occupant has stopped and been fetched): the
 * other slots keep the products they carry; the slot's next dictionary update is the cold one of a fit */
int aa_slots_reload(aa_ctx *ctx, int r, const double *C, long ldc, const double *Z, const double *alpha);
int aa_slots_finish(aa_ctx *ctx);     /* all slots stopped: stopping-iteration factors restored, products rebuilt */
int aa_slots_fetch(aa_ctx *ctx, int r, double *C, long ldc, double *Z, double *CX, long ldx, int carried,
                   double *costs, double *cost0, double *alpha /* k scale factors out, nullable */);
int aa_slots_end(aa_ctx *ctx);

/* ------------------------------------------------------------------ GPNH */
/* Dictionary W (p x k) is passed transposed, Wt (k x p, leading dimension ld).
 * aa_gpnh_set_factors uploads Wt and/or Z (either may be NULL to keep the current
 * one) and, when Wt is given, computes XW = X W (n x k).
 * Replaces gpnh_convex_coding.py:292,352 (WtXt) and :270 (XW). */
int aa_gpnh_set_factors(aa_ctx *ctx, int k, const double *Wt, long ld, const double *Z);
int aa_gpnh_get_weights(aa_ctx *ctx, double *Z);
/* Z'X (k x p, summed over ranks), Z'Z (k x k), and tr(W'X'Z) = sum(XW * Z).
 * Replaces gpnh_convex_coding.py:219 (ZtX), :293,:374 (ZtZ), :303,:355,:376. */
int aa_gpnh_reduce(aa_ctx *ctx, double *ZtX, long ld, double *ZtZ, double *trace_WtXtZ);
/* QP update of the weights: A = WtW (k x k, host-supplied), b_t = -XW[t].
 * Replaces _update_gpnh_weights (gpnh_convex_coding.py:254-279). */
int aa_gpnh_weights_update(aa_ctx *ctx, const double *WtW, const aa_qp_params *params,
                           aa_qp_stats *stats);
/* 0.5 ||X - Z W'||_F^2 / n in residual form: _gpnh_cost (gpnh_convex_coding.py:199-210)
 * without the penalty term (added on the host). */
int aa_gpnh_residual_cost(aa_ctx *ctx, double *cost);

/* ------------------------------------------------------------ the two passes, on their own */
/* The two contractions against the resident matrix that every update is built from, callable
 * with caller-supplied small operands (unit tests of the pass kernels against X.dot(...) in
 * float64; the reference's C.dot(X), X.T.dot(Z) at archetypal_analysis.py:614,627 and
 * CX.dot(X.T), X.dot(XtZ) at :615,628):
 *   aa_pass_reduce_rows: out[k][p] (ld = ldo) = A' X,  A host n x k row-major (tall operand);
 *   aa_pass_row_local:   out[n][k]            = X B',  B host k x p row-major (ld = ldb).
 * Both size the context's factor buffers for k components (like aa_set_state) and leave the
 * solver state invalid (aa_set_state / aa_prepare must follow before any update). */
int aa_pass_reduce_rows(aa_ctx *ctx, int k, const double *A, double *out, long ldo);
int aa_pass_row_local(aa_ctx *ctx, int k, const double *B, long ldb, double *out);

/* Diagnostics: the scalar state of the latest dictionary SPG iteration (device-resident, see
 * csrc/aa_internal.h: ScalarSlot), copied to out[0..AA_SPG_SCALARS).  Layout: 0 tr, 1 tr(C HD),
 * 2 tr(M C K C'), 3 f_old, 4 f_new, 5 alpha (BB / first step), 6 alpha_set, 7 lambda,
 * 8 <d,g>, 9 <d,d>, 10 tr(D HD), 11 a1, 12 a2, 13 <d,g_new>, 14 ||res||^2, 15 ||res||_inf,
 * 16 n_feval, 17 flags, 18 projection multiplier, 19 max|P(x-g)-x|, 20 divisor of f. */
#define AA_SPG_SCALARS 21
int aa_get_spg_scalars(aa_ctx *ctx, double *out);

/* ------------------------------------------------------------ measurement */
/* Time `reps` launches of one hot-path GEMM kernel with HIP events on the
 * context's stream.  which: 0 = reduce-over-rows (C X, X'Z; k x p out),
 * 1 = row-local (CX X', X X'Z; n x k out), 2..5 = a plain streaming read of X (the read
 * bandwidth the memory system delivers; reference point for the roofline) with 4 / 8 / 16 / 8
 * loads in flight per thread on 4096 / 2048 / 1024 / 8192 blocks; 6 / 7 = the load pattern
 * of the row-local kernel alone, from the row-major matrix / from a tiled view of the same bytes.
 * ms_avg = average duration of one launch in milliseconds. */
int aa_time_kernel(aa_ctx *ctx, int which, int reps, double *ms_avg);
/* In-context timing of the two pass kernels: while enabled, every launch of the
 * reduce-over-rows / row-local GEMM kernel is bracketed by a HIP event pair on the
 * context's stream.  Each call returns the average duration (ms) and the number of
 * launches recorded since the previous call, clears them, and sets the mode to `enable`. */
int aa_gemm_timing(aa_ctx *ctx, int enable, double *ms_reduce_rows, int *n_reduce_rows,
                   double *ms_row_local, int *n_row_local);
/* Names of the kernels the context's most recent reduce-over-rows and row-local passes were
 * launched with, as "reduce;row_local" (e.g. "k_reduce_rows_f32<1,4>;k_row_local_f32_dma<8>"):
 * the kernels are picked by shard size and dtype (csrc/kernels_gemm.hip: launch_reduce_rows,
 * launch_row_local), and a parity test has to know that it exercised the ones a benchmark times
 * (tests/test_gpu_headline.py).  The reference has no counterpart (NumPy picks its BLAS kernels
 * silently).  Empty names before the first pass.  Returns AA_ERR_ARG if `len` is too small. */
int aa_pass_kernels(aa_ctx *ctx, char *buf, int len);

#ifdef __cplusplus
}
#endif
#endif /* AA_HIP_H */
