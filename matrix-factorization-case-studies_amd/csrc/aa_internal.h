// aa_internal.h -- shared declarations of libaa_hip.so (gfx950 only).
//
// Data layout in HBM (all device buffers are zero-initialised, padded, and owned
// by the context):
//   X        [n_pad][ldx]   T (float|double)  row-major data (or kernel) matrix;
//                            n_pad = roundup(n,128), ldx = p_pad = roundup(p,128);
//                            padding rows/columns are ZERO so GEMM kernels need no
//                            bounds checks and padding never contributes to a sum.
//   "tall"   [n_pad][KP]    double            per-sample x per-component arrays:
//                            Ct (dictionary, TRANSPOSED: component on the fast axis),
//                            Z, D (search direction), G (C XX' or C K, transposed),
//                            g (gradient), H (XX'Z or KZ), XW ...   KP = 32 or 64.
//   "wide"   [KP][p_pad]    double (+ a T copy used as MFMA operand)
//                            P = C X, Q = D X, ZtX = Z'X, Wt.
//   small    [KP][KP]       double            Gram products.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/aa_hip.h"

namespace aa {

void set_error(const char *fmt, ...);

#define AA_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            aa::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                          hipGetErrorString(e__));                                      \
            return AA_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define AA_CHECK(expr)                                                                  \
    do {                                                                                \
        int rc__ = (expr);                                                              \
        if (rc__ != AA_OK) return rc__;                                                 \
    } while (0)

#define AA_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            aa::set_error(__VA_ARGS__);                                                 \
            return code;                                                                \
        }                                                                               \
    } while (0)

inline long round_up(long v, long m) { return (v + m - 1) / m * m; }

// zero rows allocated past n_pad in X and in every tall array, so software-pipelined
// kernels may prefetch past the end of their row range without a guard
#define AA_SLACK_ROWS 64

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool borrowed = false;     // aliases another context's buffer (aa_share_data): never freed here
    int alloc(size_t b)
    {
        if (b <= bytes && p && !borrowed) return AA_OK;
        release();
        if (b == 0) b = 16;
        AA_CHECK_HIP(hipMalloc(&p, b));
        bytes = b;
        // zero fill on the null stream, waited for: the contexts' streams are non-blocking (nothing
        // they run is ordered against the null stream), and allocations are rare
        AA_CHECK_HIP(hipMemsetAsync(p, 0, b, nullptr));
        AA_CHECK_HIP(hipStreamSynchronize(nullptr));
        return AA_OK;
    }
    void release()
    {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        borrowed = false;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ------------------------------------------------------------------ scalars
// Device-resident scalar state of the dictionary SPG solver and the projection
// passes.  Written by single-thread "scalar stage" kernels so the host does not
// have to synchronise inside an SPG iteration.
enum ScalarSlot {
    SC_TRACE = 0,   // tr(XX') or tr(K)
    SC_S1,          // tr(C * HD)            H = XX'Z or KZ
    SC_A0,          // tr(M * C K C')
    SC_F_OLD,
    SC_F_NEW,
    SC_ALPHA,       // BB step
    SC_ALPHA_SET,   // 0 => derive alpha from the first projected-gradient step
    SC_LAMBDA,
    SC_DELTA,       // <d, g>
    SC_DD,          // <d, d>
    SC_S1D,         // tr(D * HD)
    SC_A1,
    SC_A2,
    SC_DGN,         // <d, g_new>
    SC_RES2,        // ||res||^2
    SC_RESINF,
    SC_NFEVAL,
    SC_FLAGS,
    SC_PROJ_A,      // multiplier of g inside the current projection: w = x - a*g
    SC_AINV,        // max |P(x-g)-x|
    SC_FNORM,       // divisor of f (k in both forms)
    SC_FMEM0,       // f_mem[0..15]
    SC_COUNT = SC_FMEM0 + 16
};

struct ProjState {          // per projection, device memory
    double t[AA_MAX_K];
    double cnt[AA_MAX_K];
    int done;
    int passes;
    double mx[AA_MAX_K];          // column maxima of the current projection
    int shrunk[AA_MAX_K];         // the support has started to shrink (oscillation guard)
    double warm[4][AA_MAX_K];     // final thresholds of the previous projection of each kind
    // multi-rank list projection: a rank whose candidate list did not fit, or a union too long
    // for the solver, leaves a column unconverged.  The host checks that right away only while
    // the lists of that kind of projection are not known to be short; otherwise the check is
    // deferred to the next point where the host reads device state anyway (sticky flag).
    int overflow_sticky;          // set when any column of any projection was left unconverged
    int list_max[4];              // longest gathered list (over the columns) of the last projection per kind
    int fin_arrived;              // blocks of the running first / finish pass that have written their partials
                                  // (FinTail: the last one finalizes and puts it back to 0)
};

#define AA_SC_STRIDE 64      // doubles per slot of the per-slot scalar blocks (>= SC_COUNT)

struct IterState {          // aa_iterate: device-side loop status
    int stop, converged, error_stage, stop_iter, last_iter, spg_flags, pad0, pad1;
};

// ------------------------------------------------------------------ context
struct Comm;   // RCCL wrapper (comm.hip)
struct P2P;    // one-shot peer-to-peer all-reduce over IPC-mapped buffers (comm.hip)

struct Ctx {
    int device = 0;
    int dtype = AA_F64;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;             // side stream: the QP's straggler kernel
    hipStream_t stream3 = nullptr;             // the live consumers of parked QP samples (qp_live): a queue of
                                               // their own, stream2 may still hold the residual projection
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    // GPNH restarts side by side (aa_gpnh_slots_*): per-slot cost records, counters, status, initial costs
    DevBuf slotCosts, slotCounters, slotStates, slotCost0;
    int slots_R = 0, slots_k = 0, slots_stride = 0, slots_max_outer = 0;
    // AA restarts side by side (aa_slots_*): the launchers of the coupled steps pick their per-slot
    // forms while this is set; per-slot SPG scalars [R][AA_SC_STRIDE]
    bool slots_aa = false, slots_started = false;
    unsigned slots_cold = 0;                   // slots loaded since the last iteration: their next dictionary
                                               // update is the cold one of a fit (projection of the caller's
                                               // factors, products recomputed); slots_cold_cols: their columns
    unsigned slots_cold_cols = 0xffffffffu;
    DevBuf slotSaveP, slotSaveGr;              // C X and (C X X')' of the running slots across a reload
    DevBuf slotScal, slotSnapP;
    aa_iter_params slots_ip;
    aa_spg_params slots_sp, slots_scale_sp;
    aa_gpnh_params slots_gp;
    aa_qp_params slots_qp;
    DevBuf qpLive;                             // ready[cap] | done[cap] flags of the live hand-over (QpLive)
    long qp_live_cap = 0;
    int qp_live_epoch = 0;
    char pass_names[2][48] = {{0}, {0}};       // last reduce-over-rows / row-local kernel launched (aa_pass_kernels)
    bool qp_tail_pending = false;              // stragglers run on stream2, results in tmpTall by slot
    // the residual projection of the dictionary SPG (spg.py:250-276: convergence flags only) runs on
    // the side stream beside the weights QP, on its own scratch set (launch_proj_side / join_side)
    hipEvent_t evFork2 = nullptr, evJoin2 = nullptr;
    bool qp_perm_ready = false;                // the previous QP launch of this context left the order of the next one in qpPerm (k_qp_wave_ord)
    long qp_perm_n = 0;
    bool side_pending = false;
    DevBuf tmpTall2, redPartial2, redOut2, proj2, projList2, projSegCnt2;
    int projPassHint2[4] = {0, 0, 0, 0};
    bool projWarm2[4] = {false, false, false, false};
    // measurement (aa_gemm_timing): HIP event pairs around every launch of the two pass kernels
    bool time_gemm = false;
    std::vector<hipEvent_t> gemmEvents[2];     // [0] reduce-over-rows, [1] row-local: start, stop, ...
    const int *qp_tail_rows = nullptr;         // device: overflow slot -> row
    const unsigned int *qp_tail_count = nullptr;   // device: number of overflow slots
    Comm *comm = nullptr;
    P2P *p2p = nullptr;
    int rank = 0, world = 1;
    bool force_comm = false;                   // AA_FORCE_RCCL=1: use RCCL even with one rank
    // multi-rank: small reductions that ride in the tail of the next all-reduce instead of having one of
    // their own (pack_comm).  ride_dst: where the rider's values go (the tail of a wide buffer, or the
    // second region of redGather); ride_count: doubles there; ride: the k_post step owed after the all-reduce
    struct RidePost {
        bool on = false;
        int kind = 0, mode = 0, NV = 0, slot = 0, gated = 0, stage_after = -1;
        unsigned max_mask = 0u;
        aa_spg_params sp;
        double *red = nullptr;
        ProjState *ps = nullptr;
        double *gather = nullptr;              // where its [world][NV][KP] values sit
        long count = 0;
    };
    RidePost ride, ride_grad;
    double *ride_dst = nullptr;                // explicit tail content of a wide buffer (Z'Z behind Z'X): its start
    long ride_count = 0;                       // ... and length
    double *ride_gather_next = nullptr;        // the next closing reduction of a projection goes here and waits (ride)
    bool weights_follow = false;               // a weights update follows this dictionary update in the same iteration
    bool ride_grad_next = false;               // the next launch_grad's dot rides with the projection that follows it

    // data
    int form = AA_FORM_DATA;
    long n = 0, n_pad = 0, p = 0, p_pad = 0, n_global = 0, row_offset = 0;
    DevBuf X;
    bool have_data = false;
    double trace = 0.0;
    bool have_trace = false;

    // problem
    int k = 0, KP = 0;
    bool have_state = false, grams_valid = false, gpnh_valid = false;
    std::vector<double> alpha;                 // host copy
    std::vector<double> ZtZ, CKCt, CKZ;        // host copies (k x k, dense k), fetched on demand
    bool dict_inputs_overridden = false;

    // tall arrays
    DevBuf Ct, Zt, Dt, Gr, Gn, gk, gn, H, tmpTall;
    // wide arrays
    DevBuf P, Q, ZtX, Pw, Qw;                  // Pw/Qw: T-typed MFMA operand copies
    DevBuf wideScratch;                        // KP x max(p_pad, n_pad) double
    // reduction scratch
    DevBuf partial;                            // GEMM split-row partials
    DevBuf rlPartial;                          // row-local GEMM split over column chunks (short, wide data)
    DevBuf redPartial;                         // tall/wide reduction partials
    DevBuf gramOut;                            // up to 4 KPxKP results
    DevBuf gramState;                          // [3][KP*KP] on the device: Z'Z | C K C' | C K Z
    DevBuf costDev;                            // costs recorded by aa_outer_iterations
    DevBuf costSlot;                           // device counter: next free slot of costDev
    bool host_grams_valid = false;             // the host copies below match gramState
    DevBuf redOut;                             // finalized [NV][KP] reduction results
    DevBuf redGather;                          // multi-rank: [world][NV][KP] per-rank results
    DevBuf listGather;                         // multi-rank: [world][KP][cap + 1] candidate lists
    DevBuf scalars;                            // SC_COUNT doubles
    DevBuf proj;                               // ProjState
    DevBuf projList, projSegCnt;               // candidate lists of the column projection
    DevBuf Mdev, alphaDev;                     // KP*KP, KP
    DevBuf iterState, snapC, snapZ, snapAlpha; // aa_iterate: status record, factors at the stopping iteration
    DevBuf qpIters;                            // n ints: pass counts of the latest weights update
    DevBuf qpPerm;                             // n ints: sample order of the lane kernel
    DevBuf feat, featNorm;                     // implicit RBF kernel: features [n_pad][feat_ld] (float64) and their squared norms
    long feat_p = 0, feat_ld = 0;
    int implicit_kernel = 0;                   // 0: none; 1: K_ij = exp(-rbf_gamma ||x_i - x_j||^2), never formed (aa_set_rbf_features)
    double rbf_gamma = 0.0;
    DevBuf fsScratch;                          // FurthestSum on the device: running sums, one distance column, state, alive flags
    bool qp_iters_valid = false;               // qpIters belongs to the current rows / state
    bool linear_kernel = false;                // data form, KernelAA conventions: K = X X' implicit (aa_set_linear_kernel)
    DevBuf qpStats;                            // 2 long long
    void *hostPinned = nullptr;                // small pinned staging
    size_t hostPinnedBytes = 0;

    long nslab = 0, rows_per_slab = 0;         // reduce-over-rows decomposition
    int tallBlocks = 0;                        // blocks of the tall reductions
    int projPassHint[4] = {0, 0, 0, 0};        // Michelot passes the last projection of each kind needed
    bool projWarm[4] = {false, false, false, false};   // ProjState::warm[kind] is valid
    bool projListShort[4] = {false, false, false, false};   // multi-rank: gathered lists of this kind were <= a quarter of the solver's capacity at the last poll
    bool x_feasible = false;                   // dictionary known to be on the simplex
    bool products_valid = false;               // P (= CX) and Gr (= C XX' or C K) match Ct
    bool ckz_valid = false;                    // gramState's C K Z matches Ct and H
};

// ------------------------------------------------------------------ kernels_gemm.hip
// out[KP][p_pad] (double) = sum_r A[r][i] * X[r][c];  A tall double, X T.
// Also refreshes the T-typed operand copy outT (may alias out when T == double).
int launch_reduce_rows(Ctx *c, const double *A_tall, double *out_wide, void *outT,
                       bool main_only = false);
// second stage of the split-row reduction (after a main_only launch): sums the slab partials
// plus `extra_slabs` further slabs appended by launch_reduce_rows_fixup
int launch_reduce_rows_finish(Ctx *c, double *out_wide, void *outT, int extra_slabs);
// appends sum_s (znew[s] - Z[rows[s]]) x_{rows[s]}' as QP_FIX_SLABS slabs of partials
int launch_reduce_rows_fixup(Ctx *c, const unsigned int *count_dev, const int *rows_dev,
                             const double *zslot, const double *Ztall);
// out[n_pad][KP] (double) = sum_c X[r][c] * B[i][c];  B wide, T-typed.
int launch_row_local(Ctx *c, const void *B_wideT, double *out_tall);
int launch_stream_probe(Ctx *c, int variant);   // measurement: one streaming read of X

// ------------------------------------------------------------------ kernels_tall.hip
int tall_setup(Ctx *c);
int launch_proj(Ctx *c, const double *x, const double *g, double a_const, int a_slot, int mode,
                const aa_spg_params *sp = nullptr, int stage_after = -1);
enum { PROJ_FEAS = 0, PROJ_ALPHA = 1, PROJ_DIR = 2, PROJ_RES = 3 };
int launch_grad(Ctx *c, const double *Graw, const double *H, double *gout, double scale,
                const double *d_for_dot /*nullable*/, int dot_slot, double *xupd = nullptr,
                const aa_spg_params *sp = nullptr, int stage_after = -1,
                const aa_spg_params *setup_sp = nullptr /* the update's set-up as block 0 */, double setup_fnorm = 0.0);
int launch_tall_axpy_lambda(Ctx *c, double *x, const double *d);             // x += lambda * d
int launch_tall_dot_scaled(Ctx *c, const double *x, const double *H, const double *alpha_dev,
                           int slot);   // sum x*H*alpha (alpha_dev nullable => 1)
int launch_gram_tall(Ctx *c, const double *A, const double *B, double *out_dev, bool local_only = false);
int launch_ride_post(Ctx *c, Ctx::RidePost *rp, const double *gather);   // the k_post step a rider is owed
#define AA_WIDE_TAIL(KP) ((size_t)(KP) * (KP) + (size_t)64 * 4 * (KP))   // doubles behind a wide buffer for riders // A'B  (KPxKP)
int launch_gram_wide(Ctx *c, const double *A, const double *B, double *out_dev); // A B' (KPxKP)
int launch_scale_gram(Ctx *c, double *dst, const double *src);      // dst = D src D
// the outer iteration's judge, run by the cost kernel that records the iteration's last cost
struct GpnhJudge {
    int on, it;
    double cost0, tol, mono_tol;
    int criterion, require, upd_dict, upd_w;
    IterState *st;
    int track_spg;              // AA: fold the dictionary SPG's flags into the status record
};
int launch_aa_cost(Ctx *c, double *out_dev, int *slot_counter_dev = nullptr, const GpnhJudge *judge = nullptr);   // cost from gramState;
                                              // with a counter: out_dev[(*counter)++]
int launch_set_scalars(Ctx *c, double trace, double fnorm);          // SC_TRACE, SC_FNORM
int launch_wide_axpy_lambda(Ctx *c, double *P, const double *Q, void *PT);   // P += lambda*Q; PT = T(P)
int launch_wide_to_T(Ctx *c, const double *src, void *dstT);
int launch_transpose_wide_to_tall(Ctx *c, const double *wide, double *tall); // [KP][p_pad] -> [n_pad][KP] (kernel form)
int launch_transpose_tall_to_wide(Ctx *c, const double *tall, double *wide, void *wideT);
int launch_scalar_stage(Ctx *c, int stage, const aa_spg_params *sp, int it);
int launch_linesearch_fused(Ctx *c, const aa_spg_params *sp, double *cost_out, int *cost_slot);
int launch_dict_setup(Ctx *c, const aa_spg_params *sp, double fnorm, unsigned slotmask = 0xffffffffu);
int launch_iter_judge(Ctx *c, int it, double cost0, const double *costs, IterState *st,
                      const aa_iter_params *ip, bool judged = false);
int launch_cost_carry(Ctx *c, double *costs, int *slot, double cost0);
int launch_scale_factors(Ctx *c, const aa_spg_params *sp, double delta_box, int it, double cost0,
                         const double *costs, const int *slot, IterState *st, double mono_tol, int require);
int launch_col_has_nan(Ctx *c, const void *raw_dev, int host_dtype, long ld, long n_total, long p_full,
                       const double *w_dev /*nullable*/, unsigned char *flags_dev);
int launch_gather_weight(Ctx *c, const void *raw_dev, int host_dtype, long ld, long row0, long n,
                         const int *idx_dev, long p_valid, const double *w_dev);
int launch_data_to_double(Ctx *c, double *out_dev);
int launch_gpnh_solve(Ctx *c, double lambda, int *ok_dev);
// R restarts side by side (kernels_tall.hip: GpnhSlots)
int launch_gpnh_solve_slots(Ctx *c, double lambda);
int launch_gpnh_cost_slots(Ctx *c, double lambda, unsigned mask, int what, const aa_iter_params *ip, bool form_gram);
int launch_gpnh_snap_slots(Ctx *c);
int launch_aa_cost_slots(Ctx *c, int what, const aa_iter_params *ip);
int launch_aa_snap_slots(Ctx *c);
int launch_qp_slots_aa(Ctx *c, const aa_qp_params *p);      // kernels_qp.hip: quad + wave kernels, grid.y = slot
int launch_qp_slots(Ctx *c, int R, int k, const double *gram_dev, const aa_qp_params *p);   // kernels_qp.hip
bool gpnh_cost_can_gram(const Ctx *c);
int launch_gpnh_cost(Ctx *c, double lambda, double *out_dev, int *slot_counter, bool from_wide = false,
                     bool gram_w = false, const GpnhJudge *judge = nullptr);
int launch_gpnh_judge(Ctx *c, int it, double cost0, const double *costs, IterState *st,
                      const aa_iter_params *ip, bool judged = false);
enum { ST_INIT_F = 0, ST_ALPHA = 1, ST_LINESEARCH = 2, ST_BB = 3, ST_CONV = 4 };
int launch_row_sqnorm_sum(Ctx *c, double *trace_out_host);
int launch_distance_column(Ctx *c, long j_local, int owner_has_row, const double *xj_host, double *d_host);
int launch_implicit_kv(Ctx *c, const double *V_tall, double *out_tall);   // out = K V for the implicit kernel (tall in, tall out)
int launch_rbf_norms(Ctx *c);
int launch_furthest_sum(Ctx *c, int k, int start, const int *exclude_host, int n_ex, int extra_steps,
                        int *selected_host, int *tie_host);
int launch_row_broadcast(Ctx *c, long j_local, bool own);
int launch_proj_side(Ctx *c, const double *x, const double *g, double a_const, int a_slot, int mode,
                     const aa_spg_params *sp, int stage_after);   // launch_proj on the side stream / scratch set
int join_side(Ctx *c);             // the main stream waits for a pending side-stream projection
// Copies and fills ORDERED ON THE CONTEXT'S STREAM.  The streams of a context are created
// hipStreamNonBlocking (round 4): the legacy null stream orders nothing for them any more, and a
// null-stream hipMemcpy in one context is no longer a barrier across every context of the process
// (worker threads of fit_restarts, stream capture).  ctx_memcpy: the side stream joined, the copy
// enqueued on the main stream, the host waits for it (the semantics the synchronous hipMemcpy had
// for this context).  ctx_memset: enqueued only -- whatever reads the buffer is enqueued behind it.
inline hipError_t ctx_memcpy(Ctx *c, void *dst, const void *src, size_t nbytes, hipMemcpyKind kind)
{
    if (join_side(c) != AA_OK) return hipErrorUnknown;
    hipError_t e = hipMemcpyAsync(dst, src, nbytes, kind, c->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(c->stream);
}
inline hipError_t ctx_memcpy2d(Ctx *c, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width,
                               size_t height, hipMemcpyKind kind)
{
    if (join_side(c) != AA_OK) return hipErrorUnknown;
    hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, c->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(c->stream);
}
inline hipError_t ctx_memset(Ctx *c, void *dst, int value, size_t nbytes)
{
    return hipMemsetAsync(dst, value, nbytes, c->stream);
}
bool side_available(const Ctx *c);  // side stream + second scratch set usable (single rank, fused stages)
int side_begin(Ctx *c);            // launch_* calls go to the side stream (own scratch set) until side_end
int side_begin_behind(Ctx *c);
int side_end(Ctx *c);
extern int g_proj_res_side, g_grad_side;
int proj_poll_multirank(Ctx *c);   // multi-rank: read the deferred overflow flag / list lengths (host sync point)   // multi-rank: row j -> wideScratch on every rank
int launch_residual_cost(Ctx *c, const double *Ztall, const double *Wwide, const double *alpha_dev,
                         double *out_host);

// ------------------------------------------------------------------ kernels_qp.hip
// A_host (k x k) and bscale_host (k or null) come from the host, or -- A_host == nullptr --
// the Hessian is D G D with G = gram_dev (KP x KP, device) and D = bscale = alphaDev, set
// up by a kernel (no host synchronisation).
int launch_qp(Ctx *c, const double *A_host, const double *Btall, long stride_j, long stride_t,
              const double *bscale_host /*k or null*/, double *Ztall, int ldz, long n, int k,
              const aa_qp_params *p, int *iters_dev, aa_qp_stats *stats,
              const double *gram_dev = nullptr, bool defer_tail = false);
// defer_tail: the wave-per-sample kernel of the stragglers runs on the side stream and leaves
// its results in tmpTall, indexed by overflow slot (c->qp_tail_pending); the caller overlaps
// the reduce-over-rows pass of Z'X with it, then calls launch_qp_tail_fixup: it waits for
// the stragglers, adds sum_s (z_new - z_old)_s x_s' as QP_FIX_SLABS extra slabs of the
// split-row partials (device-side count, no host synchronisation) and commits z_new to Z.
#define QP_FIX_SLABS 32
int launch_qp_tail_fixup(Ctx *c, double *Ztall);
int launch_simplex_rows_generic(hipStream_t s, const double *in, double *out, long rows, long cols);

extern int g_use_graph;           // solver.hip
extern int g_proj_mode;           // kernels_tall.hip
extern int g_proj_small, g_pq_blocks;           // kernels_tall.hip
extern int g_fuse_finalize;       // kernels_tall.hip
extern int g_proj_check_always;   // kernels_tall.hip
extern int g_proj_list_cap;       // kernels_tall.hip
extern int g_row_local_variant;   // kernels_gemm.hip
extern int g_row_local_waves;     // kernels_gemm.hip
extern int g_row_local_split, g_row_local_chunk, g_row_local_acc64, g_row_local_ring, g_row_local_nt, g_row_local_prio, g_row_local_early, g_row_local_reverse;     // kernels_gemm.hip
extern int g_row_local_stagger;   // kernels_gemm.hip
extern int g_reduce_rows_unroll;  // kernels_gemm.hip
extern int g_f64_mfma;            // kernels_gemm.hip
extern int g_reduce_rows_blocks;  // kernels_gemm.hip
extern int g_qp_pass_cap;         // kernels_qp.hip
extern int g_qp_mode;             // kernels_qp.hip
extern int g_qp_wave_blocks;
extern int g_qp_quad_waves, g_qp_quad_refill, g_qp_quad_cap, g_qp_quad_occ;   // kernels_qp.hip
extern int g_qp_row_waves;        // kernels_qp.hip
extern int g_qp_matvec;           // kernels_qp.hip
extern int g_qp_row_hot;          // kernels_qp.hip
extern int g_qp_row_chunk;        // kernels_qp.hip
extern int g_qp_row_long;         // kernels_qp.hip
extern int g_qp_row_cap;          // kernels_qp.hip
extern int g_qp_refill_min;       // kernels_qp.hip
extern int g_qp_waves;            // kernels_qp.hip
extern int g_qp_sort;             // kernels_qp.hip
extern int g_qp_profile;          // kernels_qp.hip
extern int g_qp_wave_mem1, g_qp_fused_order, g_qp_wave_lazy, g_fin_in_last, g_gram_side, g_setup_in_grad, g_pack_comm, g_qp_quad_lazy, g_qp_wave_queue, g_pq_mfma;
extern int g_qp_overlap_tail, g_qp_tail_cap, g_qp_live, g_qp_live_blocks, g_qp_live_occ;     // kernels_qp.hip

// ------------------------------------------------------------------ comm.hip
int comm_unique_id(void *id128);
int comm_init(Ctx *c, const void *id128, int rank, int world);
void comm_destroy(Ctx *c);
int comm_allreduce(Ctx *c, double *dev, long count, int op);   // in place, on c->stream
int p2p_export(Ctx *c, int world, void *handle64);   // this rank's receive buffer as an IPC handle
int p2p_init(Ctx *c, const void *handles, int rank, int world);   // handles: world x 64 bytes, in rank order
int p2p_check(Ctx *c);             // a kernel of the peer-to-peer all-reduce gave up waiting?

}  // namespace aa
