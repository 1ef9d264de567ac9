// solver.hip -- host drivers of the device-resident solver and the extern "C" surface
// declared in include/aa_hip.h.
//
// Dictionary update = the reference's spg() (spg.py:46-283) specialised to the
// dictionary objective (archetypal_analysis.py:261-341), restated so that one SPG
// iteration costs TWO passes over X instead of the reference's seven:
//   * C X is linear in C:       (C + lam D) X = CX + lam DX            (one pass: DX)
//   * f is quadratic in C:      f(C + lam D) follows from tr(C H D), tr(D H D) and the
//                               k x k Grams of CX, DX -> every line-search trial is
//                               scalar arithmetic (stage kernels, kernels_tall.hip)
//   * the gradient at the accepted point needs (CX + lam DX) X'         (one pass)
//     and doubles as the C XX' the weights update consumes (archetypal_analysis.py:619)
//   * the gradient at the top of the next iteration (spg.py:176) is the one just
//     computed (spg.py:233).
// The reference's quirks are kept: f divides by k while the data-form gradient divides
// by n (archetypal_analysis.py:265 vs :297), f_mem starts as zeros (spg.py:153),
// sigma_one is an absolute bound (spg.py:28).
#include "aa_internal.h"

#include <mutex>

namespace aa {

int g_outer_nosync = 0;
int g_use_graph = 0;     // 1: aa_outer_iterations replays a captured pair of iterations (measured neutral)

static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

static inline size_t esize(const Ctx *c) { return c->dtype == AA_F32 ? 4 : 8; }

// T-typed MFMA operand copy of a wide double array (aliases the array itself in f64)
static inline void *operandT(Ctx *c, DevBuf &dbl, DevBuf &shadow)
{
    return c->dtype == AA_F32 ? shadow.p : dbl.p;
}

static int ensure_problem(Ctx *c, int k)
{
    AA_REQUIRE(c->have_data, AA_ERR_STATE, "set_data must precede the factors");
    AA_REQUIRE(k >= 1 && k <= AA_MAX_K, AA_ERR_ARG,
               "n_components = %d unsupported by the HIP backend (1..%d)", k, AA_MAX_K);
    const int KP = k <= 32 ? 32 : 64;
    if (c->k == k && c->KP == KP && c->Ct.p) return AA_OK;
    c->k = k;
    c->KP = KP;
    const size_t tall = (size_t)(c->n_pad + AA_SLACK_ROWS) * KP * sizeof(double);
    const size_t wide = ((size_t)KP * c->p_pad + AA_WIDE_TAIL(KP)) * sizeof(double);   // (tail: riders of the all-reduce)
    DevBuf *talls[] = {&c->Ct, &c->Zt, &c->Dt, &c->Gr, &c->Gn, &c->gk, &c->gn, &c->H, &c->tmpTall};
    for (DevBuf *b : talls) {
        b->release();
        AA_CHECK(b->alloc(tall));
    }
    DevBuf *wides[] = {&c->P, &c->Q, &c->ZtX, &c->wideScratch};
    for (DevBuf *b : wides) {
        b->release();
        AA_CHECK(b->alloc(wide));
    }
    c->Pw.release();
    c->Qw.release();
    if (c->dtype == AA_F32) {
        AA_CHECK(c->Pw.alloc((size_t)KP * c->p_pad * sizeof(float)));
        AA_CHECK(c->Qw.alloc((size_t)KP * c->p_pad * sizeof(float)));
    }
    // split-row decomposition of the reduce-over-rows GEMM
    const long colgroups = c->dtype == AA_F32 ? (c->p_pad + 511) / 512 : (c->p_pad + 255) / 256;
    long nslab = (g_reduce_rows_blocks + colgroups - 1) / colgroups;   // 2 blocks per CU: 64 slabs at p = 4096
    if (nslab > 256) nslab = 256;
    if (nslab < 1) nslab = 1;
    long rps = round_up((c->n_pad + nslab - 1) / nslab, 32);
    nslab = (c->n_pad + rps - 1) / rps;
    c->nslab = nslab;
    c->rows_per_slab = rps;
    c->partial.release();
    AA_CHECK(c->partial.alloc((size_t)(nslab + QP_FIX_SLABS) * KP * c->p_pad * esize(c)));
    AA_CHECK(tall_setup(c));
    c->alpha.assign(k, 1.0);
    c->ZtZ.assign((size_t)k * k, 0.0);
    c->CKCt.assign((size_t)k * k, 0.0);
    c->CKZ.assign((size_t)k * k, 0.0);
    c->grams_valid = false;
    c->host_grams_valid = false;
    c->gpnh_valid = false;
    c->have_state = false;
    c->dict_inputs_overridden = false;
    return AA_OK;
}

static int upload_alpha(Ctx *c)
{
    std::vector<double> a(c->KP, 0.0);
    for (int i = 0; i < c->k; ++i) a[i] = c->alpha[i];
    AA_CHECK_HIP(ctx_memcpy(c, c->alphaDev.p, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
    return AA_OK;
}

// host k x k (dense) <- device KP x KP
static int fetch_gram(Ctx *c, const double *dev, std::vector<double> &out)
{
    std::vector<double> tmp((size_t)c->KP * c->KP);
    AA_CHECK_HIP(hipMemcpyAsync(tmp.data(), dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    out.resize((size_t)c->k * c->k);
    for (int i = 0; i < c->k; ++i)
        for (int j = 0; j < c->k; ++j) out[(size_t)i * c->k + j] = tmp[(size_t)i * c->KP + j];
    return AA_OK;
}

static int upload_tall(Ctx *c, DevBuf &dst, const double *src, long ld_row, long ld_col, long rows, int cols)
{
    // dst[r][i] = src[r*ld_row + i*ld_col], zero padded to [n_pad][KP]
    std::vector<double> tmp((size_t)c->n_pad * c->KP, 0.0);
    for (long r = 0; r < rows; ++r)
        for (int i = 0; i < cols; ++i) tmp[(size_t)r * c->KP + i] = src[r * ld_row + i * ld_col];
    AA_CHECK_HIP(ctx_memcpy(c, dst.p, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    return AA_OK;
}

static int download_tall(Ctx *c, const DevBuf &src, double *dst, long ld_row, long ld_col, long rows, int cols)
{
    std::vector<double> tmp((size_t)c->n_pad * c->KP);
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_CHECK_HIP(ctx_memcpy(c, tmp.data(), src.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (long r = 0; r < rows; ++r)
        for (int i = 0; i < cols; ++i) dst[r * ld_row + i * ld_col] = tmp[(size_t)r * c->KP + i];
    return AA_OK;
}

// device-resident Gram state [Z'Z | C K C' | C K Z]; the host copies follow on demand
static inline double *dev_ZtZ(Ctx *c) { return c->gramState.as<double>(); }
static inline double *dev_CKCt(Ctx *c) { return c->gramState.as<double>() + (size_t)c->KP * c->KP; }
static inline double *dev_CKZ(Ctx *c) { return c->gramState.as<double>() + (size_t)2 * c->KP * c->KP; }

// C K Z = C' H (k x k) is only needed by the host (scale-factor update, aa_get_grams) and
// by the Gram form of the cost: after a dictionary update it is recomputed on demand
static int ensure_ckz(Ctx *c)
{
    if (c->ckz_valid) return AA_OK;
    AA_CHECK(launch_gram_tall(c, c->Ct.as<double>(), c->H.as<double>(), dev_CKZ(c)));
    c->ckz_valid = true;
    c->host_grams_valid = false;
    return AA_OK;
}

static int sync_host_grams(Ctx *c)
{
    AA_CHECK(ensure_ckz(c));
    if (c->host_grams_valid) return AA_OK;
    const size_t GS = (size_t)c->KP * c->KP;
    std::vector<double> tmp(3 * GS);
    AA_CHECK_HIP(hipMemcpyAsync(tmp.data(), c->gramState.p, tmp.size() * sizeof(double),
                                hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> *dst[3] = {&c->ZtZ, &c->CKCt, &c->CKZ};
    for (int m = 0; m < 3; ++m) {
        dst[m]->resize((size_t)c->k * c->k);
        for (int i = 0; i < c->k; ++i)
            for (int j = 0; j < c->k; ++j) (*dst[m])[(size_t)i * c->k + j] = tmp[m * GS + (size_t)i * c->KP + j];
    }
    c->host_grams_valid = true;
    return AA_OK;
}

// cost of the current state (archetypal_analysis.py:553-556), evaluated on the device
static int device_cost(Ctx *c, double *cost)
{
    AA_CHECK(c->costDev.alloc(64 * sizeof(double)));
    AA_CHECK(ensure_ckz(c));
    AA_CHECK(launch_aa_cost(c, c->costDev.as<double>()));
    AA_CHECK_HIP(hipMemcpyAsync(cost, c->costDev.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    return AA_OK;
}

static int ensure_trace(Ctx *c)
{
    if (c->have_trace) return AA_OK;
    if (c->implicit_kernel) {                    // RBF: every diagonal entry is exp(0) = 1
        c->trace = (double)c->n_global;
        c->have_trace = true;
        return AA_OK;
    }
    AA_CHECK(launch_row_sqnorm_sum(c, &c->trace));
    c->have_trace = true;
    return AA_OK;
}

// ----------------------------------------------------------------- Gram refresh
static int refresh_after_dictionary(Ctx *c, bool recompute_products, bool ckct_done = false)
{
    double *gpp = dev_CKCt(c);
    if (c->form == AA_FORM_DATA) {
        if (recompute_products) {
            AA_CHECK(launch_reduce_rows(c, c->Ct.as<double>(), c->P.as<double>(), operandT(c, c->P, c->Pw)));
            AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gr.as<double>()));
        }
        // the fused line search leaves (P + lam Q)(P + lam Q)' in the Gram state and the host
        // fetches C K Z on demand (ensure_ckz)
        if (ckct_done) {
            c->products_valid = true;
            c->ckz_valid = false;
            c->host_grams_valid = false;
            return AA_OK;
        }
        AA_CHECK(launch_gram_wide(c, c->P.as<double>(), c->P.as<double>(), gpp));
    } else {
        if (recompute_products && c->implicit_kernel) {
            AA_CHECK(launch_implicit_kv(c, c->Ct.as<double>(), c->Gr.as<double>()));      // (C K)' = K C'
        } else if (recompute_products) {
            AA_CHECK(launch_reduce_rows(c, c->Ct.as<double>(), c->wideScratch.as<double>(), nullptr));
            AA_CHECK(launch_transpose_wide_to_tall(c, c->wideScratch.as<double>(), c->Gr.as<double>()));
        }
        AA_CHECK(launch_gram_tall(c, c->Gr.as<double>(), c->Ct.as<double>(), gpp));
    }
    c->products_valid = true;
    AA_CHECK(launch_gram_tall(c, c->Ct.as<double>(), c->H.as<double>(), dev_CKZ(c)));
    c->ckz_valid = true;
    c->host_grams_valid = false;
    return AA_OK;
}

static int refresh_after_weights(Ctx *c)
{
    if (c->form == AA_FORM_DATA) {
        // Z'X.  When the QP's stragglers are still running on the side stream, the pass
        // over X starts with the Z of the lane kernel and the rows that change afterwards
        // enter as a rank-m correction (launch_qp_tail_fixup), so the latency-bound tail of
        // the weights update hides behind an HBM-bound pass.
        const bool tail = c->qp_tail_pending;
        // Z'Z needs the weights only: on the side stream, beside the HBM-bound pass, instead of 16 us of
        // two small launches between the two passes (joined below, with the dictionary's side work)
        const bool multi = c->world > 1 || c->force_comm;
        const bool pack = multi && g_pack_comm;         // Z'Z rides in the tail of the Z'X all-reduce
        const bool gram_on_side = g_gram_side && !tail && !c->slots_aa && side_available(c);
        if (gram_on_side) {
            AA_CHECK(side_begin_behind(c));
            const int rc = launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), dev_ZtZ(c));
            AA_CHECK(side_end(c));
            AA_CHECK(rc);
        }
        AA_CHECK(launch_reduce_rows(c, c->Zt.as<double>(), c->ZtX.as<double>(), operandT(c, c->ZtX, c->Qw), true));
        if (tail) AA_CHECK(launch_qp_tail_fixup(c, c->Zt.as<double>()));
        if (pack) {
            double *tailp = c->ZtX.as<double>() + (size_t)c->KP * c->p_pad;
            AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), tailp, true));
            c->ride_dst = tailp;
            c->ride_count = (long)c->KP * c->KP;
        }
        AA_CHECK(launch_reduce_rows_finish(c, c->ZtX.as<double>(), operandT(c, c->ZtX, c->Qw),
                                           tail ? QP_FIX_SLABS : 0));
        if (pack)
            AA_CHECK_HIP(hipMemcpyAsync(dev_ZtZ(c), c->ZtX.as<double>() + (size_t)c->KP * c->p_pad,
                                        (size_t)c->KP * c->KP * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        else if (!gram_on_side) AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), dev_ZtZ(c)));
        AA_CHECK(join_side(c));                  // the side stream's gradient kernel reads the old H and updates C
        AA_CHECK(launch_row_local(c, operandT(c, c->ZtX, c->Qw), c->H.as<double>()));
    } else {
        if (c->qp_tail_pending) AA_CHECK(launch_qp_tail_fixup(c, c->Zt.as<double>()));   // not used
        AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), dev_ZtZ(c)));
        if (c->implicit_kernel) {
            AA_CHECK(launch_implicit_kv(c, c->Zt.as<double>(), c->H.as<double>()));       // K Z
        } else {
            AA_CHECK(launch_transpose_tall_to_wide(c, c->Zt.as<double>(), c->ZtX.as<double>(),
                                                   operandT(c, c->ZtX, c->Qw)));
            AA_CHECK(launch_row_local(c, operandT(c, c->ZtX, c->Qw), c->H.as<double>()));
        }
    }
    AA_CHECK(launch_gram_tall(c, c->Ct.as<double>(), c->H.as<double>(), dev_CKZ(c)));
    c->ckz_valid = true;
    c->host_grams_valid = false;
    c->dict_inputs_overridden = false;
    return AA_OK;
}

static int prepare(Ctx *c, double *cost)
{
    AA_REQUIRE(c->have_state, AA_ERR_STATE, "set_state must precede prepare");
    AA_CHECK(ensure_trace(c));
    AA_CHECK(upload_alpha(c));
    AA_CHECK(refresh_after_weights(c));          // ZtZ, H = XX'Z (or KZ), CKZ (overwritten below)
    AA_CHECK(refresh_after_dictionary(c, true)); // CX, C XX', C XX' C', C XX' Z
    c->grams_valid = true;
    if (cost) AA_CHECK(device_cost(c, cost));
    return AA_OK;
}

// ----------------------------------------------------------------- restart slots: mixed cold / warm groups
// A slot that has just been loaded (slots_cold) goes through the COLD dictionary update of a fit --
// factors projected, C X, (C X)(C X)' and C X X' recomputed -- while the other slots of the group
// are in the middle of theirs and must keep the products they carry (C X is updated as P + lambda Q:
// recomputing it gives other bits).  The recomputing kernels run on the stacked arrays; these
// helpers put the warm slots' parts back afterwards.
static unsigned slots_all_mask(const Ctx *c) { return c->slots_R >= 32 ? 0xffffffffu : ((1u << c->slots_R) - 1u); }
static bool slots_mixed(const Ctx *c) { return c->slots_aa && c->slots_cold != 0 && c->slots_cold != slots_all_mask(c); }

static int slots_save_P(Ctx *c)
{
    const size_t wide = (size_t)c->KP * c->p_pad * sizeof(double);
    AA_CHECK(c->slotSaveP.alloc(wide));
    AA_CHECK_HIP(hipMemcpyAsync(c->slotSaveP.p, c->P.p, wide, hipMemcpyDeviceToDevice, c->stream));
    return AA_OK;
}
static int slots_restore_P_warm(Ctx *c)
{
    const int k = c->slots_k;
    for (int r = 0; r < c->slots_R; ++r) {
        if ((c->slots_cold >> r) & 1u) continue;
        const size_t off = (size_t)r * k * c->p_pad;
        AA_CHECK_HIP(hipMemcpyAsync(c->P.as<double>() + off, c->slotSaveP.as<double>() + off,
                                    (size_t)k * c->p_pad * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->dtype == AA_F32) AA_CHECK(launch_wide_to_T(c, c->P.as<double>(), operandT(c, c->P, c->Pw)));
    return AA_OK;
}

// ----------------------------------------------------------------- dictionary SPG
// cost_out / cost_slot (device, nullable): where the cost after the update is recorded.
// *cost_recorded tells the caller whether that happened inside the update (fused line search,
// one SPG iteration) or is still to be done with launch_aa_cost.
static int dictionary_update(Ctx *c, const aa_spg_params *sp, aa_spg_stats *st, bool refresh,
                             double *cost_out = nullptr, int *cost_slot = nullptr,
                             bool *cost_recorded = nullptr)
{
    AA_REQUIRE(c->have_state && (c->grams_valid || c->dict_inputs_overridden), AA_ERR_STATE,
               "dictionary_update needs prepare() or set_dictionary_inputs() first");
    AA_REQUIRE(sp->max_iterations >= 1, AA_ERR_ARG, "spg max_iterations must be >= 1");
    AA_REQUIRE(sp->memory <= 16, AA_ERR_ARG,
               "spg memory = %d exceeds the HIP backend limit of 16", sp->memory);
    const int k = c->k, KP = c->KP;
    const bool data = c->form == AA_FORM_DATA;
    if (cost_recorded) *cost_recorded = false;
    AA_CHECK(join_side(c));

    double *x = c->Ct.as<double>();
    double *gram = c->gramOut.as<double>();
    const size_t GS = (size_t)KP * KP;
    // the data form divides the gradient by n (:297), the kernel form by k (:288); a data matrix
    // standing in for the kernel K = X X' (aa_set_linear_kernel) follows the kernel form
    const double gscale = (data && !c->linear_kernel) ? 1.0 / (double)c->n_global : 1.0 / (double)k;

    // spg.py:148 projects the start point.  A dictionary that came out of our own update
    // (x_old + lambda d, a convex combination of simplex points) is feasible to rounding,
    // and projecting it would move it by ~1e-17: skipped.  Caller-supplied factors are
    // always projected.
    // The same holds for its products: CX (P) and C XX' / C K (Gr) computed by the
    // previous update's refresh (archetypal_analysis.py:618-619) are the f / gradient
    // ingredients of this one (spg.py:156,176) -- the weights update in between does not
    // touch the dictionary -- so two passes over X are saved (`warm`).
    const bool warm = c->x_feasible && c->products_valid;
    // fused small stages (fewer, fatter launches; aa_set_option("fuse_finalize", 0) restores the
    // one-kernel-per-step sequence): with everything the update needs already in the Gram
    // state, one block sets up M, the scalars and f(x)
    const bool fused = data && g_fuse_finalize;
    const bool fast = fused && warm && c->grams_valid && c->ckz_valid && !c->dict_inputs_overridden;
    // (single fit: the set-up block rides in the gradient launch below, beside its row blocks)
    const bool setup_in_grad = fast && g_setup_in_grad && !c->slots_aa;
    if (fast) {
        if (!setup_in_grad) AA_CHECK(launch_dict_setup(c, sp, (double)k));      // spg.py:153-157
    } else {
        // M = D Z'Z D  (archetypal_analysis.py:310,330), formed on the device
        AA_CHECK(launch_scale_gram(c, c->Mdev.as<double>(), dev_ZtZ(c)));
        AA_CHECK(launch_set_scalars(c, c->trace, (double)k));   // archetypal_analysis.py:265,277
        if (!c->x_feasible) AA_CHECK(launch_proj(c, x, nullptr, 0.0, -1, PROJ_FEAS));
        if (!warm) {
            if (data) {
                const bool mixed = slots_mixed(c);
                if (mixed) AA_CHECK(slots_save_P(c));
                AA_CHECK(launch_reduce_rows(c, x, c->P.as<double>(), operandT(c, c->P, c->Pw)));
                if (mixed) AA_CHECK(slots_restore_P_warm(c));
                AA_CHECK(launch_gram_wide(c, c->P.as<double>(), c->P.as<double>(), gram));
            } else {
                if (c->implicit_kernel) {
                    AA_CHECK(launch_implicit_kv(c, x, c->Gr.as<double>()));
                } else {
                    AA_CHECK(launch_reduce_rows(c, x, c->wideScratch.as<double>(), nullptr));
                    AA_CHECK(launch_transpose_wide_to_tall(c, c->wideScratch.as<double>(), c->Gr.as<double>()));
                }
                AA_CHECK(launch_gram_tall(c, c->Gr.as<double>(), x, gram));
            }
        } else {
            AA_CHECK_HIP(hipMemcpyAsync(gram, dev_CKCt(c), GS * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
        AA_CHECK(launch_tall_dot_scaled(c, x, c->H.as<double>(), c->alphaDev.as<double>(), SC_S1));
        AA_CHECK(launch_scalar_stage(c, ST_INIT_F, sp, 0));                     // spg.py:156
        // the slots that are in the middle of their fits start this update the way they always do
        if (slots_mixed(c)) AA_CHECK(launch_dict_setup(c, sp, (double)k, slots_all_mask(c) & ~c->slots_cold));
        if (data && !warm) AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gr.as<double>()));
    }
    c->products_valid = false;
    AA_CHECK(launch_grad(c, c->Gr.as<double>(), c->H.as<double>(), c->gk.as<double>(), gscale, nullptr, 0, nullptr,
                         nullptr, -1, setup_in_grad ? sp : nullptr, (double)k));

    std::vector<double> sc(SC_COUNT, 0.0);
    int n_iter = -1, flags = 0;
    bool grad_on_side = false;
    for (int it = 0; it < sp->max_iterations; ++it) {
        n_iter = it;
        if (it == 0 && sp->alpha0 < 0.0) {                                      // spg.py:178-189
            AA_CHECK(launch_proj(c, x, c->gk.as<double>(), 1.0, -1, PROJ_ALPHA, sp, ST_ALPHA));
        }
        // multi-rank: the closing reduction of this projection (<d,g>, <d,d>, ...: read by the line search
        // only) rides with the all-reduce of Q = D'X
        const bool multi = c->world > 1 || c->force_comm;
        if (multi && g_pack_comm && data && !c->slots_aa) c->ride_gather_next = c->Q.as<double>() + (size_t)KP * c->p_pad;
        AA_CHECK(launch_proj(c, x, c->gk.as<double>(), 0.0, SC_ALPHA, PROJ_DIR)); // spg.py:191-194,206
        if (data) {
            AA_CHECK(launch_reduce_rows(c, c->Dt.as<double>(), c->Q.as<double>(), nullptr));
            if (fused) {
                // (P Q'), (Q Q'), the line search, the Gram of the accepted point and -- with one
                // SPG iteration per update -- the cost after the update: one launch
                const bool rec = cost_out && sp->max_iterations == 1;
                AA_CHECK(launch_linesearch_fused(c, sp, rec ? cost_out : nullptr, rec ? cost_slot : nullptr));
                if (rec && cost_recorded) *cost_recorded = true;
            } else {
                AA_CHECK(launch_gram_wide(c, c->P.as<double>(), c->Q.as<double>(), gram + GS));
                AA_CHECK(launch_gram_wide(c, c->Q.as<double>(), c->Q.as<double>(), gram + 2 * GS));
                AA_CHECK(launch_scalar_stage(c, ST_LINESEARCH, sp, 1));
            }
        } else {
            if (c->implicit_kernel) {
                AA_CHECK(launch_implicit_kv(c, c->Dt.as<double>(), c->Gn.as<double>()));  // (D K)' = K D'
            } else {
                AA_CHECK(launch_reduce_rows(c, c->Dt.as<double>(), c->wideScratch.as<double>(), nullptr));
                AA_CHECK(launch_transpose_wide_to_tall(c, c->wideScratch.as<double>(), c->Gn.as<double>()));
            }
            AA_CHECK(launch_gram_tall(c, c->Gn.as<double>(), x, gram + GS));
            AA_CHECK(launch_gram_tall(c, c->Gn.as<double>(), c->Dt.as<double>(), gram + 2 * GS));
            AA_CHECK(launch_gram_tall(c, c->Gr.as<double>(), c->Dt.as<double>(), gram + 3 * GS));
            AA_CHECK(launch_scalar_stage(c, ST_LINESEARCH, sp, 0));
        }
        if (data) {
            // x = x_old + lam d happens inside the gradient kernel (it reads d for <d, g_new>),
            // the BB stage (spg.py:232-244) in the last block of that kernel
            AA_CHECK(launch_wide_axpy_lambda(c, c->P.as<double>(), c->Q.as<double>(), operandT(c, c->P, c->Pw)));
            AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gn.as<double>()));
            // With one SPG iteration per update nobody on the main stream waits for g_new, the step
            // x += lambda d, the BB stage or the residual projection before the update of the
            // weights has run (the QP reads C XX' and the Gram of the accepted point only): the whole
            // tail of the SPG iteration goes to the side stream and runs beside the QP (round 4:
            // 36 + 14 us of gradient kernel and finalize step off the critical path).  Joined in
            // refresh_after_weights before H = X (X'Z)' is overwritten (the gradient reads the old
            // H) and before C K Z is formed from the updated C.
            grad_on_side = g_grad_side && sp->max_iterations == 1 && !st && side_available(c);
            if (grad_on_side) AA_CHECK(side_begin(c));
            // multi-rank: <d, g_new> rides with the first reduction of the residual projection below
            if (multi && g_pack_comm && !c->slots_aa && g_proj_mode == 0 && g_fuse_finalize) c->ride_grad_next = true;
            AA_CHECK(launch_grad(c, c->Gn.as<double>(), c->H.as<double>(), c->gn.as<double>(), gscale,
                                 c->Dt.as<double>(), SC_DGN, x, sp, ST_BB));
        } else {
            AA_CHECK(launch_tall_axpy_lambda(c, x, c->Dt.as<double>()));        // x = x_old + lam d
            AA_CHECK(launch_tall_axpy_lambda(c, c->Gr.as<double>(), c->Gn.as<double>()));
            AA_CHECK(launch_grad(c, c->Gr.as<double>(), c->H.as<double>(), c->gn.as<double>(), gscale,
                                 c->Dt.as<double>(), SC_DGN, nullptr, sp, ST_BB));
        }
        // multi-rank, a weights update next: the closing sums of the residual projection (convergence flags
        // only: read by the judge behind the weights update) wait for the Z'X all-reduce and ride behind Z'Z
        if (multi && g_pack_comm && data && !c->slots_aa && c->weights_follow && sp->max_iterations == 1 && !st &&
            g_fuse_finalize)
            c->ride_gather_next = c->ZtX.as<double>() + (size_t)KP * c->p_pad + (size_t)KP * KP;
        // spg.py:250-276; with one SPG iteration per update nobody waits for the flags: side stream
        if (grad_on_side) {
            const int rc = launch_proj(c, x, c->gn.as<double>(), 1.0, -1, PROJ_RES, sp, ST_CONV);
            AA_CHECK(side_end(c));
            AA_CHECK(rc);
        } else if (sp->max_iterations == 1 && !st && data)
            AA_CHECK(launch_proj_side(c, x, c->gn.as<double>(), 1.0, -1, PROJ_RES, sp, ST_CONV));
        else
            AA_CHECK(launch_proj(c, x, c->gn.as<double>(), 1.0, -1, PROJ_RES, sp, ST_CONV));
        if (data) std::swap(c->Gr, c->Gn);
        std::swap(c->gk, c->gn);
        // the host needs the flags only to decide on a further iteration or to report
        if (st || it + 1 < sp->max_iterations) {
            AA_CHECK_HIP(hipMemcpyAsync(sc.data(), c->scalars.p, SC_COUNT * sizeof(double),
                                        hipMemcpyDeviceToHost, c->stream));
            AA_CHECK_HIP(hipStreamSynchronize(c->stream));
            AA_CHECK(proj_poll_multirank(c));
            flags = (int)sc[SC_FLAGS];
            if (flags & (AA_SPG_FLAG_CONVERGED | AA_SPG_FLAG_MAX_FEVAL)) break;
        }
    }
    if (n_iter == sp->max_iterations - 1 && !(flags & AA_SPG_FLAG_CONVERGED))
        flags |= AA_SPG_FLAG_MAX_ITER;                                          // spg.py:278-281
    if (st) {
        st->f = sc[SC_F_OLD];
        st->n_iter = n_iter;
        st->n_feval = (int)sc[SC_NFEVAL];
        st->flags = flags;
        st->res_norm = sqrt(sc[SC_RES2]);
    }
    c->x_feasible = true;
    if (refresh) AA_CHECK(refresh_after_dictionary(c, false, fused));
    return AA_OK;
}

static int weights_update(Ctx *c, const aa_qp_params *qp, aa_qp_stats *stats)
{
    AA_REQUIRE(c->have_state && c->grams_valid, AA_ERR_STATE, "weights_update needs prepare() first");
    // Hessian D C K C' D (archetypal_analysis.py:387) and b-scale D are set up on the device
    const bool defer = g_qp_overlap_tail && c->form == AA_FORM_DATA;
    AA_CHECK(c->qpIters.alloc((size_t)c->n * sizeof(int)));
    if (c->slots_aa) {                           // restarts side by side: one QP per sample and slot
        AA_CHECK(launch_qp_slots_aa(c, qp));
        c->qp_iters_valid = false;
        AA_CHECK(refresh_after_weights(c));
        AA_CHECK(join_side(c));
        return AA_OK;
    }
    AA_CHECK(launch_qp(c, nullptr, c->Gr.as<double>(), 1, c->KP, nullptr, c->Zt.as<double>(), c->KP, c->n,
                       c->k, qp, c->qpIters.as<int>(), stats, dev_CKCt(c), defer));
    c->qp_iters_valid = true;
    AA_CHECK(refresh_after_weights(c));
    AA_CHECK(join_side(c));                      // the judge behind this update reads the SPG flags
    return AA_OK;
}

}  // namespace aa

// ===========================================================================
// extern "C"
// ===========================================================================
using namespace aa;

struct aa_ctx {
    Ctx c;
};

extern "C" {

const char *aa_last_error(void) { return g_err.c_str(); }
int aa_version(void) { return 100; }

int aa_set_option(const char *name, int value)
{
    AA_REQUIRE(name != nullptr, AA_ERR_ARG, "null option name");
    if (!strcmp(name, "row_local_variant")) {
        AA_REQUIRE(value >= -1 && value <= 9, AA_ERR_ARG, "row_local_variant must be in -1..9");
        g_row_local_variant = value;
    } else if (!strcmp(name, "qp_pass_cap")) {
        AA_REQUIRE(value >= 1, AA_ERR_ARG, "qp_pass_cap must be >= 1");
        g_qp_pass_cap = value;
    } else if (!strcmp(name, "qp_refill_min")) {
        AA_REQUIRE(value >= 1 && value <= 64, AA_ERR_ARG, "qp_refill_min must be in 1..64");
        g_qp_refill_min = value;
    } else if (!strcmp(name, "row_local_waves")) {
        AA_REQUIRE(value == 0 || (value >= 8 && value <= 16), AA_ERR_ARG, "row_local_waves must be 0 or 8..16");
        g_row_local_waves = value;
    } else if (!strcmp(name, "row_local_acc64")) {
        AA_REQUIRE(value >= 0 && value <= 2, AA_ERR_ARG, "row_local_acc64 must be 0, 1 or 2");
        g_row_local_acc64 = value;
    } else if (!strcmp(name, "row_local_ring")) {
        AA_REQUIRE(value == 0 || (value >= 8 && value <= 12), AA_ERR_ARG, "row_local_ring must be 0 or 8..12");
        g_row_local_ring = value;
    } else if (!strcmp(name, "row_local_nt")) {
        g_row_local_nt = value != 0;
    } else if (!strcmp(name, "row_local_early")) {
        g_row_local_early = value != 0;
    } else if (!strcmp(name, "row_local_prio")) {
        g_row_local_prio = value != 0;
    } else if (!strcmp(name, "row_local_reverse")) {
        g_row_local_reverse = value != 0;
    } else if (!strcmp(name, "row_local_chunk")) {
        AA_REQUIRE(value >= 0, AA_ERR_ARG, "row_local_chunk must be >= 0");
        g_row_local_chunk = value;
    } else if (!strcmp(name, "row_local_split")) {
        g_row_local_split = value != 0;
    } else if (!strcmp(name, "f64_mfma")) {
        AA_REQUIRE(value >= 0 && value <= 3, AA_ERR_ARG, "f64_mfma must be in 0..3");
        g_f64_mfma = value;
    } else if (!strcmp(name, "reduce_rows_unroll")) {
        AA_REQUIRE(value == 4 || value == 8, AA_ERR_ARG, "reduce_rows_unroll must be 4 or 8");
        g_reduce_rows_unroll = value;
    } else if (!strcmp(name, "reduce_rows_blocks")) {
        AA_REQUIRE(value >= 1, AA_ERR_ARG, "reduce_rows_blocks must be >= 1");
        g_reduce_rows_blocks = value;      // takes effect at the next aa_set_state
    } else if (!strcmp(name, "row_local_stagger")) {
        AA_REQUIRE(value >= 0, AA_ERR_ARG, "row_local_stagger must be >= 0");
        g_row_local_stagger = value;
    } else if (!strcmp(name, "proj_list_cap")) {
        AA_REQUIRE(value >= 1 && value <= 2048, AA_ERR_ARG, "proj_list_cap must be in 1..2048");
        g_proj_list_cap = value;
    } else if (!strcmp(name, "proj_mode")) {
        AA_REQUIRE(value == 0 || value == 1, AA_ERR_ARG, "proj_mode must be 0 or 1");
        g_proj_mode = value;
    } else if (!strcmp(name, "proj_check")) {
        g_proj_check_always = value != 0;
    } else if (!strcmp(name, "fuse_finalize")) {
        g_fuse_finalize = value != 0;
    } else if (!strcmp(name, "qp_wave_queue")) {
        g_qp_wave_queue = value != 0;
    } else if (!strcmp(name, "qp_quad_lazy")) {
        g_qp_quad_lazy = value != 0;
    } else if (!strcmp(name, "pack_comm")) {
        g_pack_comm = value != 0;
    } else if (!strcmp(name, "setup_in_grad")) {
        g_setup_in_grad = value != 0;
    } else if (!strcmp(name, "gram_side")) {
        g_gram_side = value != 0;
    } else if (!strcmp(name, "fin_in_last")) {
        g_fin_in_last = value != 0;
    } else if (!strcmp(name, "qp_fused_order")) {
        g_qp_fused_order = value != 0;
    } else if (!strcmp(name, "grad_side")) {
        g_grad_side = value != 0;
    } else if (!strcmp(name, "qp_wave_lazy")) {
        g_qp_wave_lazy = value != 0;
    } else if (!strcmp(name, "qp_wave_mem1")) {
        g_qp_wave_mem1 = value != 0;
    } else if (!strcmp(name, "qp_overlap_tail")) {
        g_qp_overlap_tail = value != 0;
    } else if (!strcmp(name, "qp_tail_cap")) {
        AA_REQUIRE(value >= 0, AA_ERR_ARG, "qp_tail_cap must be >= 0");
        g_qp_tail_cap = value;
    } else if (!strcmp(name, "use_graph")) {
        g_use_graph = value != 0;
    } else if (!strcmp(name, "pq_mfma")) {
        g_pq_mfma = value != 0;
    } else if (!strcmp(name, "pq_blocks")) {
        AA_REQUIRE(value >= 1 && value <= 1024, AA_ERR_ARG, "pq_blocks must be in 1..1024");
        g_pq_blocks = value;
    } else if (!strcmp(name, "proj_res_side")) {
        g_proj_res_side = value != 0;      // takes effect at the next aa_set_state of a new problem size
    } else if (!strcmp(name, "proj_small")) {
        g_proj_small = value;
    } else if (!strcmp(name, "qp_profile")) {
        g_qp_profile = value != 0;
    } else if (!strcmp(name, "qp_sort")) {
        g_qp_sort = value != 0;
    } else if (!strcmp(name, "qp_waves")) {
        AA_REQUIRE(value >= 1, AA_ERR_ARG, "qp_waves must be >= 1");
        g_qp_waves = value;
    } else if (!strcmp(name, "qp_mode")) {
        AA_REQUIRE(value >= 0 && value <= 4, AA_ERR_ARG, "qp_mode must be in 0..4");
        g_qp_mode = value;
    } else if (!strcmp(name, "qp_quad_waves")) {
        AA_REQUIRE(value >= 1, AA_ERR_ARG, "qp_quad_waves must be >= 1");
        g_qp_quad_waves = value;
    } else if (!strcmp(name, "qp_quad_refill")) {
        AA_REQUIRE(value >= 1 && value <= 16, AA_ERR_ARG, "qp_quad_refill must be in 1..16");
        g_qp_quad_refill = value;
    } else if (!strcmp(name, "qp_wave_blocks")) {
        AA_REQUIRE(value >= 1 && value <= 8192, AA_ERR_ARG, "qp_wave_blocks must be in 1..8192");
        g_qp_wave_blocks = value;
    } else if (!strcmp(name, "outer_nosync")) {
        g_outer_nosync = value != 0;
    } else if (!strcmp(name, "qp_live")) {
        g_qp_live = value ? 1 : 0;
    } else if (!strcmp(name, "qp_live_occ")) {
        AA_REQUIRE(value >= 2 && value <= 4, AA_ERR_ARG, "qp_live_occ must be 2, 3 or 4");
        g_qp_live_occ = value;
    } else if (!strcmp(name, "qp_live_blocks")) {
        AA_REQUIRE(value >= 1 && value <= 128, AA_ERR_ARG, "qp_live_blocks must be in 1..128");
        g_qp_live_blocks = value;
    } else if (!strcmp(name, "qp_quad_occ")) {
        AA_REQUIRE(value >= 2 && value <= 4, AA_ERR_ARG, "qp_quad_occ must be 2, 3 or 4");
        g_qp_quad_occ = value;
    } else if (!strcmp(name, "qp_quad_cap")) {
        AA_REQUIRE(value >= 0, AA_ERR_ARG, "qp_quad_cap must be >= 0");
        g_qp_quad_cap = value;
    } else if (!strcmp(name, "qp_row_waves")) {
        AA_REQUIRE(value >= 1, AA_ERR_ARG, "qp_row_waves must be >= 1");
        g_qp_row_waves = value;
    } else if (!strcmp(name, "qp_row_chunk")) {
        AA_REQUIRE(value >= 0 && value <= 4096, AA_ERR_ARG, "qp_row_chunk must be in 0..4096");
        g_qp_row_chunk = value;
    } else if (!strcmp(name, "qp_matvec")) {
        AA_REQUIRE(value == 0 || value == 1, AA_ERR_ARG, "qp_matvec must be 0 or 1");
        g_qp_matvec = value;
    } else if (!strcmp(name, "qp_row_long")) {
        AA_REQUIRE(value >= 0 && value < 64, AA_ERR_ARG, "qp_row_long must be in 0..63");
        g_qp_row_long = value;
    } else if (!strcmp(name, "qp_row_cap")) {
        AA_REQUIRE(value >= 1, AA_ERR_ARG, "qp_row_cap must be >= 1");
        g_qp_row_cap = value;
    } else if (!strcmp(name, "qp_row_hot")) {
        AA_REQUIRE(value >= 0, AA_ERR_ARG, "qp_row_hot must be >= 0");
        g_qp_row_hot = value;
    } else {
        set_error("unknown option '%s'", name);
        return AA_ERR_ARG;
    }
    return AA_OK;
}

int aa_device_count(int *count)
{
    int n = 0;
    AA_CHECK_HIP(hipGetDeviceCount(&n));
    *count = n;
    return AA_OK;
}

int aa_ctx_create(aa_ctx **out, int device, int dtype)
{
    AA_REQUIRE(out != nullptr, AA_ERR_ARG, "null ctx pointer");
    AA_REQUIRE(dtype == AA_F32 || dtype == AA_F64, AA_ERR_ARG, "bad dtype %d", dtype);
    int n = 0;
    AA_CHECK_HIP(hipGetDeviceCount(&n));
    AA_REQUIRE(n > 0, AA_ERR_HIP, "no HIP device visible");
    AA_REQUIRE(device >= 0 && device < n, AA_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    AA_CHECK_HIP(hipSetDevice(device));
    aa_ctx *h = new aa_ctx();
    h->c.device = device;
    h->c.dtype = dtype;
    // non-blocking streams: no implicit synchronisation with the legacy null stream, i.e. with the
    // other contexts of the process (every copy and fill of this library goes through a stream:
    // ctx_memcpy / ctx_memset)
    hipError_t e = hipStreamCreateWithFlags(&h->c.stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        int lo = 0, hi = 0;                       // numerically lower = higher priority
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&h->c.stream2, hipStreamNonBlocking, hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->c.stream3, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->c.evFork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->c.evJoin, hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(e));
        delete h;
        return AA_ERR_HIP;
    }
    *out = h;
    return AA_OK;
}

int aa_ctx_destroy(aa_ctx *h)
{
    if (!h) return AA_OK;
    Ctx *c = &h->c;
    (void)hipSetDevice(c->device);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_destroy(c);
    DevBuf *all[] = {&c->X, &c->Ct, &c->Zt, &c->Dt, &c->Gr, &c->Gn, &c->gk, &c->gn, &c->H, &c->tmpTall,
                     &c->P, &c->Q, &c->ZtX, &c->Pw, &c->Qw, &c->wideScratch, &c->partial, &c->rlPartial, &c->redPartial,
                     &c->gramOut, &c->gramState, &c->costDev, &c->costSlot, &c->redOut, &c->redGather, &c->listGather, &c->scalars, &c->proj, &c->projList, &c->projSegCnt, &c->Mdev, &c->alphaDev, &c->iterState, &c->snapC, &c->snapZ, &c->snapAlpha, &c->qpIters, &c->qpPerm, &c->fsScratch, &c->feat, &c->featNorm,
                     &c->qpStats, &c->qpLive, &c->slotCosts, &c->slotCounters, &c->slotStates, &c->slotCost0, &c->slotSnapP, &c->slotSaveP, &c->slotSaveGr, &c->tmpTall2, &c->redPartial2, &c->redOut2, &c->proj2, &c->projList2, &c->projSegCnt2};
    for (DevBuf *b : all) b->release();
    if (c->evFork2) (void)hipEventDestroy(c->evFork2);
    if (c->evJoin2) (void)hipEventDestroy(c->evJoin2);
    for (int w = 0; w < 2; ++w)
        for (hipEvent_t e : c->gemmEvents[w]) (void)hipEventDestroy(e);
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    if (c->evJoin) (void)hipEventDestroy(c->evJoin);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete h;
    return AA_OK;
}

int aa_set_linear_kernel(aa_ctx *h, int on)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    AA_REQUIRE(!on || h->c.form == AA_FORM_DATA, AA_ERR_STATE, "aa_set_linear_kernel needs a data matrix");
    h->c.linear_kernel = on != 0;
    return AA_OK;
}

int aa_set_rbf_features(aa_ctx *h, const double *X, long n, long p, long ld, double gamma)
{
    AA_REQUIRE(h && X, AA_ERR_ARG, "null argument");
    AA_REQUIRE(n >= 1 && p >= 1 && ld >= p && p < (1L << 30), AA_ERR_ARG, "bad shape n=%ld p=%ld ld=%ld", n, p, ld);
    AA_REQUIRE(gamma > 0.0 && gamma < 1e300, AA_ERR_ARG, "gamma must be positive");
    Ctx *c = &h->c;
    AA_REQUIRE(c->world == 1 && !c->force_comm, AA_ERR_STATE, "the implicit RBF kernel is single-rank");
    AA_CHECK_HIP(hipSetDevice(c->device));
    c->form = AA_FORM_KERNEL;                    // the kernel form of the algorithm (/k conventions, tr K)
    c->linear_kernel = false;
    c->implicit_kernel = 1;
    c->rbf_gamma = gamma;
    c->n = n;
    c->p = n;                                    // the (virtual) kernel matrix is n x n
    c->n_pad = round_up(n, 128);
    c->p_pad = c->n_pad;
    c->n_global = n;
    c->row_offset = 0;
    c->X.release();                              // nothing stands in for K
    c->feat_p = p;
    c->feat_ld = round_up(p, 2);
    AA_CHECK(c->feat.alloc((size_t)(c->n_pad + AA_SLACK_ROWS) * c->feat_ld * sizeof(double)));
    AA_CHECK_HIP(ctx_memset(c, c->feat.p, 0, c->feat.bytes));
    AA_CHECK_HIP(ctx_memcpy2d(c, c->feat.p, (size_t)c->feat_ld * sizeof(double), X, (size_t)ld * sizeof(double),
                              (size_t)p * sizeof(double), (size_t)n, hipMemcpyHostToDevice));
    AA_CHECK(c->featNorm.alloc((size_t)(c->n_pad + AA_SLACK_ROWS) * sizeof(double)));
    AA_CHECK(launch_rbf_norms(c));
    c->have_data = true;
    c->have_trace = false;
    c->k = 0;
    c->KP = 0;
    c->have_state = false;
    c->grams_valid = false;
    return AA_OK;
}

int aa_comm_get_unique_id(void *id128) { return comm_unique_id(id128); }

int aa_ctx_comm_init(aa_ctx *h, const void *id128, int rank, int world)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    return comm_init(&h->c, id128, rank, world);
}

int aa_ctx_p2p_export(aa_ctx *h, int world, void *handle64)
{
    AA_REQUIRE(h && handle64, AA_ERR_ARG, "null argument");
    return p2p_export(&h->c, world, handle64);
}

int aa_ctx_p2p_init(aa_ctx *h, const void *handles, int rank, int world)
{
    AA_REQUIRE(h && handles, AA_ERR_ARG, "null argument");
    return p2p_init(&h->c, handles, rank, world);
}

int aa_ctx_allreduce_host(aa_ctx *h, double *buf, int count, int op)
{
    AA_REQUIRE(h && buf && count >= 0, AA_ERR_ARG, "bad arguments");
    Ctx *c = &h->c;
    if ((c->world <= 1 && !c->force_comm) || count == 0) return AA_OK;
    AA_CHECK_HIP(hipSetDevice(c->device));
    DevBuf tmp;
    AA_CHECK(tmp.alloc((size_t)count * sizeof(double)));
    AA_CHECK_HIP(hipMemcpyAsync(tmp.p, buf, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = comm_allreduce(c, tmp.as<double>(), count, op);
    if (rc == AA_OK) {
        hipError_t e = hipMemcpyAsync(buf, tmp.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            set_error("allreduce_host copy back: %s", hipGetErrorString(e));
            rc = AA_ERR_HIP;
        }
    }
    tmp.release();
    return rc;
}

int aa_set_data(aa_ctx *h, const void *X, int host_dtype, long n, long p, long ld, int form,
                long n_global, long row_offset)
{
    AA_REQUIRE(h && X, AA_ERR_ARG, "null argument");
    AA_REQUIRE(n >= 1 && p >= 1 && ld >= p, AA_ERR_ARG, "bad shape n=%ld p=%ld ld=%ld", n, p, ld);
    AA_REQUIRE(host_dtype == AA_F32 || host_dtype == AA_F64, AA_ERR_ARG, "bad host dtype");
    AA_REQUIRE(form == AA_FORM_DATA || form == AA_FORM_KERNEL, AA_ERR_ARG, "bad form");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    if (form == AA_FORM_KERNEL)
        AA_REQUIRE((c->world == 1 && !c->force_comm) && n == p && n_global == n && row_offset == 0, AA_ERR_ARG,
                   "kernel form needs a square matrix on a single rank");
    AA_REQUIRE(n_global >= n && row_offset >= 0 && row_offset + n <= n_global, AA_ERR_ARG, "bad shard");
    c->form = form;
    c->linear_kernel = false;
    c->implicit_kernel = 0;
    c->n = n;
    c->p = p;
    c->n_pad = round_up(n, 128);
    c->p_pad = round_up(p, 128);
    if (form == AA_FORM_KERNEL) c->p_pad = c->n_pad;
    c->n_global = n_global;
    c->row_offset = row_offset;
    const size_t es = esize(c);
    c->X.release();
    AA_CHECK(c->X.alloc((size_t)(c->n_pad + AA_SLACK_ROWS) * c->p_pad * es));
    const size_t hes = host_dtype == AA_F32 ? 4 : 8;
    if (host_dtype == c->dtype) {
        AA_CHECK_HIP(ctx_memcpy2d(c, c->X.p, (size_t)c->p_pad * es, X, (size_t)ld * hes, (size_t)p * es,
                                 (size_t)n, hipMemcpyHostToDevice));
    } else {
        const long chunk = 2048;
        std::vector<unsigned char> tmp((size_t)chunk * p * es);
        for (long r0 = 0; r0 < n; r0 += chunk) {
            const long rows = (n - r0 < chunk) ? n - r0 : chunk;
            if (c->dtype == AA_F32) {
                const double *src = reinterpret_cast<const double *>(X);
                float *dst = reinterpret_cast<float *>(tmp.data());
                for (long r = 0; r < rows; ++r)
                    for (long q = 0; q < p; ++q) dst[r * p + q] = (float)src[(r0 + r) * ld + q];
            } else {
                const float *src = reinterpret_cast<const float *>(X);
                double *dst = reinterpret_cast<double *>(tmp.data());
                for (long r = 0; r < rows; ++r)
                    for (long q = 0; q < p; ++q) dst[r * p + q] = (double)src[(r0 + r) * ld + q];
            }
            AA_CHECK_HIP(ctx_memcpy2d(c, reinterpret_cast<unsigned char *>(c->X.p) + (size_t)r0 * c->p_pad * es,
                                     (size_t)c->p_pad * es, tmp.data(), (size_t)p * es, (size_t)p * es,
                                     (size_t)rows, hipMemcpyHostToDevice));
        }
    }
    c->have_data = true;
    c->have_trace = false;
    c->k = 0;   // forces (re)allocation of the factor buffers
    c->KP = 0;
    c->have_state = false;
    c->grams_valid = false;
    return AA_OK;
}

int aa_share_data(aa_ctx *h, const aa_ctx *owner)
{
    AA_REQUIRE(h && owner && h != owner, AA_ERR_ARG, "two different contexts needed");
    Ctx *c = &h->c;
    const Ctx *o = &owner->c;
    AA_REQUIRE(!o->implicit_kernel, AA_ERR_STATE, "aa_share_data: implicit kernels are not shared");
    AA_REQUIRE(o->have_data, AA_ERR_STATE, "the owner holds no data matrix");
    AA_REQUIRE(o->device == c->device && o->dtype == c->dtype, AA_ERR_ARG,
               "aa_share_data: same device and same data type needed");
    AA_REQUIRE(c->world == 1 && !c->force_comm && o->world == 1 && !o->force_comm, AA_ERR_ARG,
               "aa_share_data is single-rank");
    AA_CHECK_HIP(hipSetDevice(c->device));
    c->X.release();
    c->X.p = o->X.p;                    // read-only everywhere in the solver
    c->X.bytes = o->X.bytes;
    c->X.borrowed = true;
    c->form = o->form;
    c->linear_kernel = false;
    c->n = o->n;
    c->p = o->p;
    c->n_pad = o->n_pad;
    c->p_pad = o->p_pad;
    c->n_global = o->n_global;
    c->row_offset = o->row_offset;
    c->have_data = true;
    c->have_trace = o->have_trace;
    c->trace = o->trace;
    c->k = 0;
    c->KP = 0;
    c->have_state = false;
    c->grams_valid = false;
    return AA_OK;
}

int aa_set_data_weighted(aa_ctx *h, const void *raw, int host_dtype, long n_total, long p_full, long ld,
                         const double *col_weight, long row0, long n, unsigned char *valid, long *p_valid)
{
    AA_REQUIRE(h && raw && valid && p_valid, AA_ERR_ARG, "null argument");
    AA_REQUIRE(host_dtype == AA_F32 || host_dtype == AA_F64, AA_ERR_ARG, "bad host dtype");
    AA_REQUIRE(n_total >= 1 && p_full >= 1 && ld >= p_full && p_full < (1L << 31), AA_ERR_ARG,
               "bad shape n_total=%ld p_full=%ld ld=%ld", n_total, p_full, ld);
    AA_REQUIRE(row0 >= 0 && n >= 1 && row0 + n <= n_total, AA_ERR_ARG, "bad row block [%ld, %ld) of %ld", row0,
               row0 + n, n_total);
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->world == 1 && !c->force_comm, AA_ERR_ARG, "preprocessing entry point is single-rank");
    const size_t hes = host_dtype == AA_F32 ? 4 : 8;
    DevBuf draw, dflag, didx, dw;
    int rc = draw.alloc((size_t)n_total * p_full * hes);
    if (rc == AA_OK) rc = dflag.alloc((size_t)p_full);
    if (rc == AA_OK && col_weight) rc = dw.alloc((size_t)p_full * sizeof(double));
    if (rc != AA_OK) { draw.release(); dflag.release(); dw.release(); return rc; }
    hipError_t e = ctx_memcpy2d(c, draw.p, (size_t)p_full * hes, raw, (size_t)ld * hes, (size_t)p_full * hes,
                               (size_t)n_total, hipMemcpyHostToDevice);
    if (e == hipSuccess && col_weight)
        e = ctx_memcpy(c, dw.p, col_weight, (size_t)p_full * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // the mask is taken on the weighted field (run_hadisst_aa.py:133,201)
        rc = launch_col_has_nan(c, draw.p, host_dtype, p_full, n_total, p_full,
                                col_weight ? dw.as<double>() : (const double *)nullptr, dflag.as<unsigned char>());
        if (rc == AA_OK) e = hipStreamSynchronize(c->stream);
    }
    std::vector<unsigned char> flags((size_t)p_full);
    if (rc == AA_OK && e == hipSuccess) e = ctx_memcpy(c, flags.data(), dflag.p, (size_t)p_full, hipMemcpyDeviceToHost);
    std::vector<int> idx;
    if (rc == AA_OK && e == hipSuccess) {
        idx.reserve((size_t)p_full);
        for (long q = 0; q < p_full; ++q) {
            valid[q] = flags[(size_t)q] ? 0 : 1;
            if (!flags[(size_t)q]) idx.push_back((int)q);
        }
        *p_valid = (long)idx.size();
        if (idx.empty()) {
            set_error("every column of the field holds a NaN");
            rc = AA_ERR_ARG;
        }
    }
    if (rc == AA_OK && e == hipSuccess) {
        const long p = (long)idx.size();
        c->form = AA_FORM_DATA;
        c->linear_kernel = false;
        c->n = n;
        c->p = p;
        c->n_pad = round_up(n, 128);
        c->p_pad = round_up(p, 128);
        c->n_global = n;
        c->row_offset = 0;
        c->X.release();
        rc = c->X.alloc((size_t)(c->n_pad + AA_SLACK_ROWS) * c->p_pad * esize(c));
        if (rc == AA_OK) rc = didx.alloc(idx.size() * sizeof(int));
        if (rc == AA_OK) e = ctx_memcpy(c, didx.p, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice);
        if (rc == AA_OK && e == hipSuccess)
            rc = launch_gather_weight(c, draw.p, host_dtype, p_full, row0, n, didx.as<int>(), p,
                                      col_weight ? dw.as<double>() : (const double *)nullptr);
        if (rc == AA_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        c->have_data = rc == AA_OK && e == hipSuccess;
        c->have_trace = false;
        c->k = 0;
        c->KP = 0;
        c->have_state = false;
        c->grams_valid = false;
    }
    draw.release();
    dflag.release();
    didx.release();
    dw.release();
    if (rc == AA_OK && e != hipSuccess) {
        set_error("aa_set_data_weighted: %s", hipGetErrorString(e));
        rc = AA_ERR_HIP;
    }
    return rc;
}

int aa_get_data(aa_ctx *h, double *out, long ld)
{
    AA_REQUIRE(h && out, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(!c->implicit_kernel, AA_ERR_STATE, "aa_get_data: no stored matrix behind an implicit kernel");
    AA_REQUIRE(c->have_data, AA_ERR_STATE, "no data");
    AA_REQUIRE(ld >= c->p, AA_ERR_ARG, "ld < p");
    AA_CHECK_HIP(hipSetDevice(c->device));
    DevBuf tmp;
    AA_CHECK(tmp.alloc((size_t)c->n * c->p * sizeof(double)));
    int rc = launch_data_to_double(c, tmp.as<double>());
    hipError_t e = hipSuccess;
    if (rc == AA_OK) e = hipStreamSynchronize(c->stream);
    if (rc == AA_OK && e == hipSuccess)
        e = ctx_memcpy2d(c, out, (size_t)ld * sizeof(double), tmp.p, (size_t)c->p * sizeof(double),
                        (size_t)c->p * sizeof(double), (size_t)c->n, hipMemcpyDeviceToHost);
    tmp.release();
    if (rc == AA_OK && e != hipSuccess) {
        set_error("aa_get_data: %s", hipGetErrorString(e));
        rc = AA_ERR_HIP;
    }
    return rc;
}

int aa_data_trace(aa_ctx *h, double *trace)
{
    AA_REQUIRE(h && trace, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_data, AA_ERR_STATE, "no data");
    AA_CHECK_HIP(hipSetDevice(c->device));
    if (!c->redPartial.p) {   // trace asked before any factors: set up minimal scratch
        AA_CHECK(c->redPartial.alloc(8192 * sizeof(double)));
        AA_CHECK(c->redOut.alloc(512 * sizeof(double)));
    }
    AA_CHECK(ensure_trace(c));
    *trace = c->trace;
    return AA_OK;
}

int aa_set_state(aa_ctx *h, int k, const double *C, long ldc, const double *Z, const double *alpha)
{
    AA_REQUIRE(h && C && Z, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK(join_side(c));
    AA_CHECK(ensure_problem(c, k));
    AA_REQUIRE(ldc >= c->n, AA_ERR_ARG, "ldc < n");
    AA_CHECK(upload_tall(c, c->Ct, C, 1, ldc, c->n, k));     // Ct[r][i] = C[i][r]
    AA_CHECK(upload_tall(c, c->Zt, Z, k, 1, c->n, k));
    for (int i = 0; i < k; ++i) c->alpha[i] = alpha ? alpha[i] : 1.0;
    AA_CHECK(upload_alpha(c));
    c->have_state = true;
    c->qp_iters_valid = false;
    c->grams_valid = false;
    c->dict_inputs_overridden = false;
    c->x_feasible = false;
    c->products_valid = false;
    c->ckz_valid = false;
    for (int m = 0; m < 4; ++m) {
        c->projWarm[m] = c->projWarm2[m] = false;
        c->projPassHint[m] = c->projPassHint2[m] = 0;
        c->projListShort[m] = false;
    }
    return AA_OK;
}

int aa_get_state(aa_ctx *h, double *C, long ldc, double *Z, double *alpha)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_state, AA_ERR_STATE, "no state");
    AA_CHECK_HIP(hipSetDevice(c->device));
    if (C) AA_CHECK(download_tall(c, c->Ct, C, 1, ldc, c->n, c->k));
    if (Z) AA_CHECK(download_tall(c, c->Zt, Z, c->k, 1, c->n, c->k));
    if (alpha)
        for (int i = 0; i < c->k; ++i) alpha[i] = c->alpha[i];
    return AA_OK;
}

int aa_set_alpha(aa_ctx *h, const double *alpha)
{
    AA_REQUIRE(h && alpha, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->k > 0, AA_ERR_STATE, "no problem");
    AA_CHECK_HIP(hipSetDevice(c->device));
    for (int i = 0; i < c->k; ++i) c->alpha[i] = alpha[i];
    return upload_alpha(c);
}

int aa_prepare(aa_ctx *h, double *cost)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    AA_CHECK_HIP(hipSetDevice(h->c.device));
    return prepare(&h->c, cost);
}

int aa_cost(aa_ctx *h, double *cost)
{
    AA_REQUIRE(h && cost, AA_ERR_ARG, "null argument");
    AA_REQUIRE(h->c.grams_valid, AA_ERR_STATE, "Gram products not valid");
    AA_CHECK_HIP(hipSetDevice(h->c.device));
    return device_cost(&h->c, cost);
}

int aa_get_grams(aa_ctx *h, double *ZtZ, double *CKCt, double *CKZ, double *trace)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    Ctx *c = &h->c;
    AA_REQUIRE(c->grams_valid, AA_ERR_STATE, "Gram products not valid");
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK(sync_host_grams(c));
    const size_t kk = (size_t)c->k * c->k;
    if (ZtZ) memcpy(ZtZ, c->ZtZ.data(), kk * sizeof(double));
    if (CKCt) memcpy(CKCt, c->CKCt.data(), kk * sizeof(double));
    if (CKZ) memcpy(CKZ, c->CKZ.data(), kk * sizeof(double));
    if (trace) *trace = c->trace;
    return AA_OK;
}

int aa_set_dictionary_inputs(aa_ctx *h, const double *KZ, const double *ZtZ, double trace)
{
    AA_REQUIRE(h && KZ && ZtZ, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_state, AA_ERR_STATE, "set_state first");
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK(upload_tall(c, c->H, KZ, c->k, 1, c->n, c->k));
    {   // Z'Z goes to the device Gram state (padded to KP x KP)
        std::vector<double> pad((size_t)c->KP * c->KP, 0.0);
        for (int i = 0; i < c->k; ++i)
            for (int j = 0; j < c->k; ++j) pad[(size_t)i * c->KP + j] = ZtZ[(size_t)i * c->k + j];
        AA_CHECK_HIP(ctx_memcpy(c, dev_ZtZ(c), pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    c->host_grams_valid = false;
    c->trace = trace;
    c->have_trace = true;
    c->dict_inputs_overridden = true;
    return AA_OK;
}

int aa_dictionary_update(aa_ctx *h, const aa_spg_params *params, aa_spg_stats *stats)
{
    AA_REQUIRE(h && params, AA_ERR_ARG, "null argument");
    AA_CHECK_HIP(hipSetDevice(h->c.device));
    return dictionary_update(&h->c, params, stats, true);
}

int aa_weights_update(aa_ctx *h, const aa_qp_params *params, aa_qp_stats *stats)
{
    AA_REQUIRE(h && params, AA_ERR_ARG, "null argument");
    AA_CHECK_HIP(hipSetDevice(h->c.device));
    return weights_update(&h->c, params, stats);
}

int aa_outer_iterations(aa_ctx *h, int n_outer, const aa_spg_params *spg, const aa_qp_params *qp,
                        double *costs)
{
    AA_REQUIRE(h && spg && qp, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    // costs are evaluated and kept on the device; the host waits once, at the end
    AA_CHECK(c->costDev.alloc((size_t)(2 * n_outer + 64) * sizeof(double)));
    AA_CHECK(c->costSlot.alloc(64));
    int *slot = c->costSlot.as<int>();
    double *cd = c->costDev.as<double>();
    AA_CHECK_HIP(hipMemsetAsync(slot, 0, sizeof(int), c->stream));
    auto one_iteration = [&]() -> int {
        bool recorded = false;
        c->weights_follow = true;
        const int rcd = dictionary_update(c, spg, nullptr, true, costs ? cd : nullptr, slot, &recorded);
        c->weights_follow = false;
        AA_CHECK(rcd);
        if (costs && !recorded) {
            AA_CHECK(ensure_ckz(c));
            AA_CHECK(launch_aa_cost(c, cd, slot));
        }
        AA_CHECK(weights_update(c, qp, nullptr));
        if (costs) AA_CHECK(launch_aa_cost(c, cd, slot));
        return AA_OK;
    };
    // An outer iteration with one SPG iteration per dictionary update is a fixed sequence
    // of ~56 launches without host decisions, two thirds of them small dependent kernels:
    // after two eager iterations (warm state, every lazy allocation done) TWO iterations --
    // the gradient / Gram buffer pairs swap once per iteration -- are captured into a
    // hipGraph and replayed.  Single rank only (RCCL stays outside graphs).
    const bool graph = g_use_graph && c->world <= 1 && !c->force_comm && c->form == AA_FORM_DATA &&
                       spg->max_iterations == 1 && n_outer >= 8 && !g_qp_overlap_tail && !g_qp_live &&
                       !c->time_gemm;      // (timing events recorded inside a capture cannot be read)
    int i = 0;
    const int eager = graph ? 2 : n_outer;
    for (; i < eager; ++i) AA_CHECK(one_iteration());
    if (graph) {
        const int pairs = (n_outer - i) / 2;
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        AA_CHECK_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int rc = one_iteration();
        if (rc == AA_OK) rc = one_iteration();
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc != AA_OK || e != hipSuccess || !g) {
            if (rc == AA_OK) set_error("hipStreamEndCapture: %s (set option use_graph=0)", hipGetErrorString(e));
            if (g) (void)hipGraphDestroy(g);
            return rc != AA_OK ? rc : AA_ERR_HIP;   // host state has advanced: not recoverable
        }
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int p = 0; p < pairs && e == hipSuccess; ++p) e = hipGraphLaunch(ge, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);    // the replays are still in flight
        if (ge) (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) {
            set_error("hipGraph: %s (set option use_graph=0)", hipGetErrorString(e));
            return AA_ERR_HIP;
        }
        i += 2 * pairs;
        for (; i < n_outer; ++i) AA_CHECK(one_iteration());
    }
    // measurement only (tools/interleave_probe.py: several contexts enqueued round-robin by one host
    // thread): return with the work in flight; the next blocking call of this context waits for it
    if (g_outer_nosync && !costs) return AA_OK;
    if (costs && n_outer > 0)
        AA_CHECK_HIP(hipMemcpyAsync(costs, c->costDev.p, (size_t)2 * n_outer * sizeof(double),
                                    hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_CHECK(proj_poll_multirank(c));
    return AA_OK;
}

int aa_iterate(aa_ctx *h, const aa_iter_params *ip, const aa_spg_params *spg, const aa_qp_params *qp,
               const aa_spg_params *scale_spg, double cost0, double *costs, aa_iter_stats *stats)
{
    AA_REQUIRE(h && ip && spg && qp && costs && stats, AA_ERR_ARG, "null argument");
    AA_REQUIRE(ip->max_outer >= 1 && ip->check_every >= 1, AA_ERR_ARG, "bad iteration counts");
    AA_REQUIRE(ip->criterion == 0 || ip->criterion == 1, AA_ERR_ARG, "bad stopping criterion");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->have_state && c->grams_valid, AA_ERR_STATE, "aa_iterate needs aa_prepare first");
    const int n_max = ip->max_outer;
    const size_t tall_bytes = (size_t)c->n_pad * c->KP * sizeof(double);
    AA_CHECK(c->costDev.alloc((size_t)(2 * n_max + 64) * sizeof(double)));
    AA_CHECK(c->costSlot.alloc(64));
    AA_CHECK(c->iterState.alloc(sizeof(IterState)));
    AA_CHECK(c->snapC.alloc(tall_bytes));
    AA_CHECK(c->snapZ.alloc(tall_bytes));
    AA_CHECK(c->snapAlpha.alloc(64 * sizeof(double)));
    int *slot = c->costSlot.as<int>();
    double *cd = c->costDev.as<double>();
    IterState *st = c->iterState.as<IterState>();
    AA_CHECK_HIP(hipMemsetAsync(slot, 0, sizeof(int), c->stream));
    AA_CHECK_HIP(hipMemsetAsync(st, 0, sizeof(IterState), c->stream));
    const bool scale = scale_spg != nullptr && ip->delta != 0.0;
    if (scale) AA_REQUIRE(scale_spg->memory <= 16, AA_ERR_ARG, "spg memory > 16 unsupported");
    IterState hs;
    memset(&hs, 0, sizeof(hs));
    int done = 0;
    while (done < n_max) {
        const int batch = n_max - done < ip->check_every ? n_max - done : ip->check_every;
        for (int b = 0; b < batch; ++b) {
            if (scale) {
                // archetypal_analysis.py:590-609: the Gram state of the previous weights refresh
                // (Z'Z, C K C', C K Z) is what the k-vector problem is made of
                AA_CHECK(ensure_ckz(c));
                AA_CHECK(launch_scale_factors(c, scale_spg, ip->delta, done + b, cost0, cd, slot, st,
                                              ip->mono_tolerance, ip->require_monotonic));
            }
            if (ip->update_dictionary) {
                bool recorded = false;
                c->weights_follow = ip->update_weights != 0;
                const int rcd = dictionary_update(c, spg, nullptr, true, cd, slot, &recorded);
                c->weights_follow = false;
                AA_CHECK(rcd);
                if (!recorded) {
                    AA_CHECK(ensure_ckz(c));
                    AA_CHECK(launch_aa_cost(c, cd, slot));
                }
            } else {
                AA_CHECK(launch_cost_carry(c, cd, slot, cost0));
            }
            if (ip->update_weights) {
                AA_CHECK(weights_update(c, qp, nullptr));
                GpnhJudge jd;                          // the judge rides in the cost kernel
                memset(&jd, 0, sizeof(jd));
                jd.on = 1;
                jd.it = done + b;
                jd.cost0 = cost0;
                jd.tol = ip->tolerance;
                jd.mono_tol = ip->mono_tolerance;
                jd.criterion = ip->criterion;
                jd.require = ip->require_monotonic;
                jd.upd_dict = ip->update_dictionary;
                jd.upd_w = ip->update_weights;
                jd.st = st;
                jd.track_spg = 1;
                AA_CHECK(launch_aa_cost(c, cd, slot, &jd));
                AA_CHECK(launch_iter_judge(c, done + b, cost0, cd, st, ip, true));
            } else {
                AA_CHECK(join_side(c));                // the snapshot reads the dictionary the side stream steps
                AA_CHECK(launch_cost_carry(c, cd, slot, cost0));
                AA_CHECK(launch_iter_judge(c, done + b, cost0, cd, st, ip));
            }
        }
        done += batch;
        AA_CHECK_HIP(hipMemcpyAsync(&hs, st, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
        AA_CHECK(proj_poll_multirank(c));
        if (hs.stop) break;
    }
    const int last = hs.stop ? hs.stop_iter : n_max - 1;
    AA_CHECK_HIP(ctx_memcpy(c, costs, cd, (size_t)2 * (last + 1) * sizeof(double), hipMemcpyDeviceToHost));
    stats->n_iter = last;
    stats->converged = hs.converged;
    stats->error_stage = hs.error_stage;
    stats->error_iter = hs.error_stage ? hs.stop_iter : -1;
    stats->spg_flags = hs.spg_flags;
    stats->reserved = done;                      /* iterations enqueued (>= n_iter + 1) */
    stats->cost = costs[2 * last + 1];
    const bool restore = hs.stop && hs.stop_iter < done - 1 && !hs.error_stage;
    if (scale) {                                 /* the host copy of alpha follows the device's */
        std::vector<double> a(c->KP, 1.0);
        AA_CHECK_HIP(ctx_memcpy(c, a.data(), restore ? c->snapAlpha.p : c->alphaDev.p, (size_t)c->KP * sizeof(double),
                               hipMemcpyDeviceToHost));
        for (int i = 0; i < c->k; ++i) c->alpha[i] = a[i];
    }
    if (hs.stop && hs.stop_iter < done - 1 && !hs.error_stage) {
        // iterations behind the stopping one have run: restore its factors and rebuild the
        // products from them (four passes over the data, once per fit)
        AA_CHECK_HIP(ctx_memcpy(c, c->Ct.p, c->snapC.p, tall_bytes, hipMemcpyDeviceToDevice));
        AA_CHECK_HIP(ctx_memcpy(c, c->Zt.p, c->snapZ.p, tall_bytes, hipMemcpyDeviceToDevice));
        c->products_valid = false;
        c->grams_valid = false;
        c->qp_iters_valid = false;
        AA_CHECK(prepare(c, nullptr));
        c->x_feasible = true;                    // came out of our own update
    }
    return AA_OK;
}

int aa_reconstruction_cost(aa_ctx *h, double *cost)
{
    AA_REQUIRE(h && cost, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_state && c->grams_valid && c->form == AA_FORM_DATA, AA_ERR_STATE,
               "reconstruction cost needs prepared data-form state");
    AA_CHECK_HIP(hipSetDevice(c->device));
    double s = 0.0;
    AA_CHECK(launch_residual_cost(c, c->Zt.as<double>(), c->P.as<double>(), c->alphaDev.as<double>(), &s));
    *cost = 0.5 * s / (double)c->n_global;
    return AA_OK;
}

int aa_get_archetypes(aa_ctx *h, double *CX, long ld)
{
    AA_REQUIRE(h && CX, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->grams_valid && c->form == AA_FORM_DATA, AA_ERR_STATE, "no C X available");
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_CHECK_HIP(ctx_memcpy2d(c, CX, (size_t)ld * sizeof(double), c->P.p, (size_t)c->p_pad * sizeof(double),
                             (size_t)c->p * sizeof(double), (size_t)c->k, hipMemcpyDeviceToHost));
    return AA_OK;
}

int aa_distance_column(aa_ctx *h, long j, double *d)
{
    AA_REQUIRE(h && d, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_data, AA_ERR_STATE, "no data");
    AA_REQUIRE(j >= 0 && j < c->n_global, AA_ERR_ARG, "row %ld out of range", j);
    AA_CHECK_HIP(hipSetDevice(c->device));
    const size_t es = esize(c);
    if (!c->tmpTall.p || c->tmpTall.bytes < (size_t)c->n * sizeof(double))
        AA_CHECK(c->tmpTall.alloc((size_t)c->n_pad * 32 * sizeof(double)));
    if (c->form == AA_FORM_DATA) {
        if (!c->wideScratch.p || c->wideScratch.bytes < (size_t)c->p_pad * sizeof(double))
            AA_CHECK(c->wideScratch.alloc((size_t)32 * c->p_pad * sizeof(double)));
        const long jl = j - c->row_offset;
        const bool own = jl >= 0 && jl < c->n;
        if ((c->world == 1 && !c->force_comm)) {
            AA_CHECK_HIP(hipMemcpyAsync(c->wideScratch.p,
                                        reinterpret_cast<unsigned char *>(c->X.p) + (size_t)jl * c->p_pad * es,
                                        (size_t)c->p_pad * es, hipMemcpyDeviceToDevice, c->stream));
        } else {
            // the owner publishes the row through a sum all-reduce of zero-filled buffers, all
            // on the device (no host staging)
            AA_CHECK(launch_row_broadcast(c, jl, own));
        }
        return launch_distance_column(c, jl, own ? 1 : 0, nullptr, d);
    }
    return launch_distance_column(c, j, 1, nullptr, d);
}

int aa_furthest_sum(aa_ctx *h, int k, long start_index, const int *exclude, int n_exclude, int extra_steps,
                    int *selected, int *tie)
{
    AA_REQUIRE(h && selected && tie, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_data, AA_ERR_STATE, "no data");
    AA_REQUIRE(c->world <= 1 && !c->force_comm, AA_ERR_STATE,
               "aa_furthest_sum is single-rank (row-sharded runs pull distance columns: aa_distance_column)");
    AA_REQUIRE(!c->implicit_kernel, AA_ERR_STATE, "aa_furthest_sum: implicit kernels pull distance columns (aa_distance_column)");
    AA_REQUIRE(k >= 1 && k <= AA_MAX_K, AA_ERR_ARG, "k = %d out of range", k);
    AA_REQUIRE(start_index >= 0 && start_index < c->n, AA_ERR_ARG, "start index %ld out of range", start_index);
    AA_REQUIRE(n_exclude >= 0 && (n_exclude == 0 || exclude), AA_ERR_ARG, "bad exclude list");
    for (int e = 0; e < n_exclude; ++e)
        AA_REQUIRE(exclude[e] >= 0 && exclude[e] < c->n && exclude[e] != start_index, AA_ERR_ARG,
                   "excluded index %d out of range or equal to the start index", exclude[e]);
    AA_REQUIRE((long)k <= c->n - n_exclude, AA_ERR_ARG, "too few points for %d components", k);
    AA_CHECK_HIP(hipSetDevice(c->device));
    return launch_furthest_sum(c, k, (int)start_index, exclude, n_exclude, extra_steps < 0 ? 0 : extra_steps,
                               selected, tie);
}

// ------------------------------------------------------------------ GPNH
int aa_gpnh_set_factors(aa_ctx *h, int k, const double *Wt, long ld, const double *Z)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->form == AA_FORM_DATA, AA_ERR_STATE, "GPNH needs a data matrix");
    AA_CHECK(ensure_problem(c, k));
    if (Wt) {
        AA_REQUIRE(ld >= c->p, AA_ERR_ARG, "ld < p");
        std::vector<double> tmp((size_t)c->KP * c->p_pad, 0.0);
        for (int i = 0; i < k; ++i)
            for (long q = 0; q < c->p; ++q) tmp[(size_t)i * c->p_pad + q] = Wt[(size_t)i * ld + q];
        AA_CHECK_HIP(ctx_memcpy(c, c->P.p, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
        AA_CHECK(launch_wide_to_T(c, c->P.as<double>(), operandT(c, c->P, c->Pw)));
        AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gr.as<double>()));   // XW
        c->gpnh_valid = true;
    }
    if (Z) {
        AA_CHECK(upload_tall(c, c->Zt, Z, k, 1, c->n, k));
        c->have_state = true;
    }
    return AA_OK;
}

int aa_gpnh_get_weights(aa_ctx *h, double *Z)
{
    AA_REQUIRE(h && Z, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_state, AA_ERR_STATE, "no weights");
    AA_CHECK_HIP(hipSetDevice(c->device));
    return download_tall(c, c->Zt, Z, c->k, 1, c->n, c->k);
}

int aa_gpnh_reduce(aa_ctx *h, double *ZtX, long ld, double *ZtZ, double *trace_WtXtZ)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_state, AA_ERR_STATE, "no weights");
    AA_CHECK_HIP(hipSetDevice(c->device));
    if (ZtX) {
        AA_CHECK(launch_reduce_rows(c, c->Zt.as<double>(), c->ZtX.as<double>(), nullptr));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
        AA_CHECK_HIP(ctx_memcpy2d(c, ZtX, (size_t)ld * sizeof(double), c->ZtX.p, (size_t)c->p_pad * sizeof(double),
                                 (size_t)c->p * sizeof(double), (size_t)c->k, hipMemcpyDeviceToHost));
    }
    if (ZtZ) {
        AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), c->gramOut.as<double>()));
        std::vector<double> g;
        AA_CHECK(fetch_gram(c, c->gramOut.as<double>(), g));
        memcpy(ZtZ, g.data(), g.size() * sizeof(double));
    }
    if (trace_WtXtZ) {
        AA_REQUIRE(c->gpnh_valid, AA_ERR_STATE, "no dictionary");
        AA_CHECK(launch_tall_dot_scaled(c, c->Gr.as<double>(), c->Zt.as<double>(), nullptr, SC_S1));
        AA_CHECK_HIP(hipMemcpyAsync(trace_WtXtZ, c->scalars.as<double>() + SC_S1, sizeof(double),
                                    hipMemcpyDeviceToHost, c->stream));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    return AA_OK;
}

int aa_gpnh_weights_update(aa_ctx *h, const double *WtW, const aa_qp_params *params, aa_qp_stats *stats)
{
    AA_REQUIRE(h && WtW && params, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_state && c->gpnh_valid, AA_ERR_STATE, "set_factors first");
    AA_CHECK_HIP(hipSetDevice(c->device));
    return launch_qp(c, WtW, c->Gr.as<double>(), 1, c->KP, nullptr, c->Zt.as<double>(), c->KP, c->n, c->k,
                     params, nullptr, stats);
}

int aa_gpnh_get_dictionary(aa_ctx *h, double *Wt, long ld)
{
    AA_REQUIRE(h && Wt, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->gpnh_valid, AA_ERR_STATE, "no dictionary");
    AA_REQUIRE(ld >= c->p, AA_ERR_ARG, "ld < p");
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_CHECK_HIP(ctx_memcpy2d(c, Wt, (size_t)ld * sizeof(double), c->P.p, (size_t)c->p_pad * sizeof(double),
                             (size_t)c->p * sizeof(double), (size_t)c->k, hipMemcpyDeviceToHost));
    return AA_OK;
}

int aa_gpnh_iterate(aa_ctx *h, const aa_gpnh_params *gp, const aa_qp_params *qp, double *cost0_out,
                    double *costs, aa_iter_stats *stats)
{
    AA_REQUIRE(h && gp && qp && costs && stats, AA_ERR_ARG, "null argument");
    const aa_iter_params *ip = &gp->loop;
    AA_REQUIRE(ip->max_outer >= 1 && ip->check_every >= 1, AA_ERR_ARG, "bad iteration counts");
    AA_REQUIRE(ip->criterion == 0 || ip->criterion == 1, AA_ERR_ARG, "bad stopping criterion");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->have_state && c->gpnh_valid && c->form == AA_FORM_DATA, AA_ERR_STATE,
               "aa_gpnh_iterate needs aa_gpnh_set_factors (dictionary and weights) first");
    const int n_max = ip->max_outer, KP = c->KP;
    const double lambda = gp->lambda_W;
    size_t snap_bytes = (size_t)c->n_pad * KP * sizeof(double);
    if ((size_t)KP * c->p_pad * sizeof(double) > snap_bytes) snap_bytes = (size_t)KP * c->p_pad * sizeof(double);
    AA_CHECK(c->costDev.alloc((size_t)(2 * n_max + 64) * sizeof(double)));
    AA_CHECK(c->costSlot.alloc(64));
    AA_CHECK(c->iterState.alloc(sizeof(IterState)));
    AA_CHECK(c->snapC.alloc(snap_bytes));
    AA_CHECK(c->snapZ.alloc(snap_bytes));
    AA_CHECK(c->qpIters.alloc((size_t)c->n * sizeof(int)));
    AA_CHECK(ensure_trace(c));
    for (int i = 0; i < c->k; ++i) c->alpha[i] = 1.0;            // the QP set-up scales by alpha
    AA_CHECK(upload_alpha(c));
    int *slot = c->costSlot.as<int>();
    double *cd = c->costDev.as<double>();
    IterState *st = c->iterState.as<IterState>();
    AA_CHECK_HIP(hipMemsetAsync(slot, 0, sizeof(int), c->stream));
    AA_CHECK_HIP(hipMemsetAsync(st, 0, sizeof(IterState), c->stream));
    // Gram state: [0] = Z'Z, [1] = W'W; scal[SC_S1] = tr(W'X'Z) = <XW, Z>
    AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), dev_ZtZ(c)));
    AA_CHECK(launch_gram_wide(c, c->P.as<double>(), c->P.as<double>(), dev_CKCt(c)));
    // tr(W'X'Z) = <Z'X, W'>: with Z'X of the current weights at hand (the dictionary solve needs it
    // anyway) the cost takes no pass over the n rows at all
    AA_CHECK(launch_reduce_rows(c, c->Zt.as<double>(), c->ZtX.as<double>(), nullptr));
    bool ztx_current = true;
    AA_CHECK(launch_gpnh_cost(c, lambda, cd + 2 * n_max + 8, nullptr, true));
    double cost0 = 0.0;
    AA_CHECK_HIP(hipMemcpyAsync(&cost0, cd + 2 * n_max + 8, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (cost0_out) *cost0_out = cost0;
    c->grams_valid = false;          // the Gram state now holds GPNH products, not AA ones
    c->ckz_valid = false;
    c->host_grams_valid = false;

    IterState hs;
    memset(&hs, 0, sizeof(hs));
    int done = 0;
    while (done < n_max) {
        const int batch = n_max - done < ip->check_every ? n_max - done : ip->check_every;
        for (int b = 0; b < batch; ++b) {
            if (ip->update_dictionary) {
                if (!ztx_current) AA_CHECK(launch_reduce_rows(c, c->Zt.as<double>(), c->ZtX.as<double>(), nullptr));
                ztx_current = true;
                AA_CHECK(launch_gpnh_solve(c, lambda, &st->pad0));                      // W'
                AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gr.as<double>()));   // X W
                const bool gram_in_cost = gpnh_cost_can_gram(c);
                if (!gram_in_cost) AA_CHECK(launch_gram_wide(c, c->P.as<double>(), c->P.as<double>(), dev_CKCt(c)));
                AA_CHECK(launch_gpnh_cost(c, lambda, cd, slot, true, gram_in_cost));
            } else {
                AA_CHECK(launch_cost_carry(c, cd, slot, cost0));
            }
            if (ip->update_weights) {
                AA_CHECK(launch_qp(c, nullptr, c->Gr.as<double>(), 1, KP, nullptr, c->Zt.as<double>(), KP, c->n,
                                   c->k, qp, c->qpIters.as<int>(), nullptr, dev_CKCt(c)));
                c->qp_iters_valid = true;
                AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), dev_ZtZ(c)));
                AA_CHECK(launch_reduce_rows(c, c->Zt.as<double>(), c->ZtX.as<double>(), nullptr));   // also the next solve's
                ztx_current = true;
                GpnhJudge jd;
                memset(&jd, 0, sizeof(jd));
                jd.on = 1;
                jd.it = done + b;
                jd.cost0 = cost0;
                jd.tol = ip->tolerance;
                jd.mono_tol = ip->mono_tolerance;
                jd.criterion = ip->criterion;
                jd.require = ip->require_monotonic;
                jd.upd_dict = ip->update_dictionary;
                jd.upd_w = ip->update_weights;
                jd.st = st;
                AA_CHECK(launch_gpnh_cost(c, lambda, cd, slot, true, false, &jd));
                AA_CHECK(launch_gpnh_judge(c, done + b, cost0, cd, st, ip, true));
            } else {
                AA_CHECK(launch_cost_carry(c, cd, slot, cost0));
                AA_CHECK(launch_gpnh_judge(c, done + b, cost0, cd, st, ip));
            }
        }
        done += batch;
        AA_CHECK_HIP(hipMemcpyAsync(&hs, st, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (hs.stop || hs.pad0) break;
    }
    memset(stats, 0, sizeof(*stats));
    stats->reserved = done;
    if (hs.pad0) {                    // the normal equations were not positive definite
        stats->error_stage = 3;
        stats->n_iter = -1;
        stats->cost = cost0;
        return AA_OK;
    }
    const int last = hs.stop ? hs.stop_iter : n_max - 1;
    AA_CHECK_HIP(ctx_memcpy(c, costs, cd, (size_t)2 * (last + 1) * sizeof(double), hipMemcpyDeviceToHost));
    stats->n_iter = last;
    stats->converged = hs.converged;
    stats->error_stage = hs.error_stage;
    stats->error_iter = hs.error_stage ? hs.stop_iter : -1;
    stats->cost = costs[2 * last + 1];
    if (hs.stop && hs.stop_iter < done - 1 && !hs.error_stage) {
        // iterations behind the stopping one have run: restore its factors, rebuild X W
        AA_CHECK_HIP(ctx_memcpy(c, c->Zt.p, c->snapZ.p, (size_t)c->n_pad * KP * sizeof(double), hipMemcpyDeviceToDevice));
        AA_CHECK_HIP(ctx_memcpy(c, c->P.p, c->snapC.p, (size_t)KP * c->p_pad * sizeof(double), hipMemcpyDeviceToDevice));
        AA_CHECK(launch_wide_to_T(c, c->P.as<double>(), operandT(c, c->P, c->Pw)));
        AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gr.as<double>()));
        c->qp_iters_valid = false;
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    return AA_OK;
}

// ------------------------------------------------------------------ AA restarts side by side
// (bin/run_hadisst_aa.py:149-174).  R independent fits of k components each in the
// component slots of ONE set of device arrays (restart r: columns [r k, (r + 1) k) of C' and Z): the
// passes over X, the Gram kernels, the gradient kernel (block-diagonal M) and the column
// projections are the single fit's launches on the stacked arrays -- an output column of theirs
// depends on its own column only -- and everything that couples the components of a fit (the SPG
// scalars, the line search, the QP Hessian, the cost, the judge) exists once per slot (ctx->slots_aa
// makes the launchers pick those forms; dictionary_update / weights_update themselves are the
// single fit's).  Every slot gets the bits it gets from aa_iterate on its own.  Production
// settings only: data form, one SPG iteration per dictionary update, delta = 0, single rank,
// fewer than 65 536 samples, k <= 16.
int aa_slots_begin(aa_ctx *h, int R, int k, const aa_iter_params *ip, const aa_spg_params *spg,
                   const aa_qp_params *qp, const aa_spg_params *scale_spg)
{
    AA_REQUIRE(h && ip && spg && qp, AA_ERR_ARG, "null argument");
    AA_REQUIRE(ip->delta == 0.0 || (scale_spg && scale_spg->memory <= 16), AA_ERR_ARG,
               "slots: delta != 0 needs the scale-factor SPG parameters (memory <= 16)");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->have_data && c->form == AA_FORM_DATA && !c->linear_kernel, AA_ERR_STATE, "AA slots need a data matrix");
    AA_REQUIRE(c->world <= 1 && !c->force_comm, AA_ERR_STATE, "restart slots are single-rank");
    // R k <= 32: the tall kernels sum a column's rows in KP-dependent interleaved chains, so the slots
    // must run the kernels a single fit of k <= 16 components runs (KP = 32) to get its bits
    AA_REQUIRE(R >= 1 && k >= 1 && k <= 16 && R * k <= 32, AA_ERR_ARG,
               "slots: R = %d restarts of k = %d components do not fit 32 component slots", R, k);
    AA_REQUIRE(c->n < 65536, AA_ERR_ARG, "AA slots: fewer than 65 536 samples");
    AA_REQUIRE(ip->max_outer >= 1 && ip->update_dictionary && ip->update_weights, AA_ERR_ARG, "slots: both updates");
    AA_REQUIRE(spg->max_iterations == 1 && spg->memory <= 16, AA_ERR_ARG, "slots: one SPG iteration per dictionary update");
    AA_REQUIRE(g_fuse_finalize && g_proj_mode == 0, AA_ERR_STATE, "slots: default projection options");
    c->slots_aa = false;
    c->k = 0;                                         // fresh, zeroed factor arrays
    AA_CHECK(ensure_problem(c, R * k));
    AA_CHECK(ensure_trace(c));
    for (int i = 0; i < c->k; ++i) c->alpha[i] = 1.0;
    AA_CHECK(upload_alpha(c));
    c->slots_R = R;
    c->slots_k = k;
    c->slots_max_outer = ip->max_outer;
    c->slots_stride = 2 * ip->max_outer + 64;
    c->slots_ip = *ip;
    c->slots_sp = *spg;
    c->slots_qp = *qp;
    if (scale_spg) c->slots_scale_sp = *scale_spg;
    AA_CHECK(c->slotCosts.alloc((size_t)R * c->slots_stride * sizeof(double)));
    AA_CHECK(c->slotCounters.alloc(64 * sizeof(int)));
    AA_CHECK(c->slotStates.alloc(32 * sizeof(IterState)));
    AA_CHECK(c->slotCost0.alloc(32 * sizeof(double)));
    AA_CHECK(c->snapAlpha.alloc(64 * sizeof(double)));
    c->scalars.release();
    AA_CHECK(c->scalars.alloc((size_t)(R + 1) * AA_SC_STRIDE * sizeof(double)));      // one block per slot
    const size_t tall_bytes = (size_t)c->n_pad * c->KP * sizeof(double);
    AA_CHECK(c->snapC.alloc(tall_bytes));
    AA_CHECK(c->snapZ.alloc(tall_bytes));
    AA_CHECK(c->slotSnapP.alloc((size_t)c->KP * c->p_pad * sizeof(double)));
    AA_CHECK_HIP(ctx_memset(c, c->snapC.p, 0, tall_bytes));
    AA_CHECK_HIP(ctx_memset(c, c->snapZ.p, 0, tall_bytes));
    std::vector<IterState> st(32);
    memset(st.data(), 0, st.size() * sizeof(IterState));
    for (int r = 0; r < 32; ++r) st[r].stop = 1;      // empty
    AA_CHECK_HIP(ctx_memcpy(c, c->slotStates.p, st.data(), st.size() * sizeof(IterState), hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memset(c, c->slotCounters.p, 0, 64 * sizeof(int)));
    AA_CHECK_HIP(ctx_memset(c, c->Ct.p, 0, tall_bytes));
    AA_CHECK_HIP(ctx_memset(c, c->Zt.p, 0, tall_bytes));
    AA_CHECK_HIP(ctx_memset(c, c->Mdev.p, 0, (size_t)c->KP * c->KP * sizeof(double)));
    AA_CHECK_HIP(ctx_memset(c, c->gramOut.p, 0, (size_t)4 * c->KP * c->KP * sizeof(double)));
    c->slots_aa = true;
    c->have_state = true;
    c->grams_valid = false;
    c->products_valid = false;
    c->ckz_valid = false;
    c->dict_inputs_overridden = false;
    c->x_feasible = false;
    c->qp_iters_valid = false;
    for (int m = 0; m < 4; ++m) {                     // as aa_set_state: every fit starts its projections cold
        c->projWarm[m] = c->projWarm2[m] = false;
        c->projPassHint[m] = c->projPassHint2[m] = 0;
        c->projListShort[m] = false;
    }
    c->slots_started = false;
    return AA_OK;
}

// start factors of a restart into slot r (C: k x n, leading dimension ldc; Z: n x k), the products of
// the stacked state rebuilt (aa_prepare's passes), the slot's initial cost
static int slots_set_alpha(Ctx *c, int r, const double *alpha, bool running)
{
    // the host copy follows the device's (the scale-factor kernel moves the running slots' factors)
    if (running) {
        std::vector<double> a(c->KP, 1.0);
        AA_CHECK_HIP(ctx_memcpy(c, a.data(), c->alphaDev.p, (size_t)c->KP * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < c->k; ++i) c->alpha[i] = a[i];
    }
    for (int i = 0; i < c->slots_k; ++i) c->alpha[r * c->slots_k + i] = alpha ? alpha[i] : 1.0;
    return upload_alpha(c);
}

int aa_slots_load(aa_ctx *h, int r, const double *C, long ldc, const double *Z, const double *alpha)
{
    AA_REQUIRE(h && C && Z, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_aa && r >= 0 && r < c->slots_R, AA_ERR_ARG, "slot %d out of range", r);
    AA_REQUIRE(ldc >= c->n, AA_ERR_ARG, "ldc < n");
    const int k = c->slots_k, o = r * k;
    AA_CHECK(join_side(c));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> ct((size_t)c->n * k);
    for (long row = 0; row < c->n; ++row)
        for (int i = 0; i < k; ++i) ct[(size_t)row * k + i] = C[(size_t)i * ldc + row];
    AA_CHECK_HIP(ctx_memcpy2d(c, c->Ct.as<double>() + o, (size_t)c->KP * sizeof(double), ct.data(), (size_t)k * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memcpy2d(c, c->Zt.as<double>() + o, (size_t)c->KP * sizeof(double), Z, (size_t)k * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyHostToDevice));
    IterState zero;
    memset(&zero, 0, sizeof(zero));
    AA_CHECK_HIP(ctx_memcpy(c, c->slotStates.as<IterState>() + r, &zero, sizeof(zero), hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memset(c, c->slotCounters.as<int>() + r, 0, sizeof(int)));
    AA_CHECK(slots_set_alpha(c, r, alpha, false));
    c->products_valid = false;
    c->grams_valid = false;
    c->slots_started = false;
    return AA_OK;
}

// a new restart into slot r of a RUNNING group (its previous occupant has stopped and been fetched):
// the factors, the products aa_prepare computes -- for this slot; the other slots keep the products
// they carry -- its initial cost; its next dictionary update is the cold one of a fit.
int aa_slots_reload(aa_ctx *h, int r, const double *C, long ldc, const double *Z, const double *alpha)
{
    AA_REQUIRE(h && C && Z, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_aa && c->slots_started && r >= 0 && r < c->slots_R, AA_ERR_ARG, "slot %d out of range", r);
    AA_REQUIRE(ldc >= c->n, AA_ERR_ARG, "ldc < n");
    const int k = c->slots_k, o = r * k, KP = c->KP;
    AA_CHECK(join_side(c));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> ct((size_t)c->n * k);
    for (long row = 0; row < c->n; ++row)
        for (int i = 0; i < k; ++i) ct[(size_t)row * k + i] = C[(size_t)i * ldc + row];
    AA_CHECK_HIP(ctx_memcpy2d(c, c->Ct.as<double>() + o, (size_t)KP * sizeof(double), ct.data(), (size_t)k * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memcpy2d(c, c->Zt.as<double>() + o, (size_t)KP * sizeof(double), Z, (size_t)k * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyHostToDevice));
    IterState zero;
    memset(&zero, 0, sizeof(zero));
    AA_CHECK_HIP(ctx_memcpy(c, c->slotStates.as<IterState>() + r, &zero, sizeof(zero), hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memset(c, c->slotCounters.as<int>() + r, 0, sizeof(int)));
    AA_CHECK(slots_set_alpha(c, r, alpha, true));
    // a fit starts its projections cold (aa_set_state): a warm threshold of +inf selects nothing, which
    // is the cold start of k_proj_small -- for this slot's columns, in both projection states
    {
        std::vector<double> inf(k, INFINITY);
        DevBuf *states[2] = {&c->proj, &c->proj2};
        for (DevBuf *b : states) {
            if (!b->p) continue;
            ProjState *ps = b->as<ProjState>();
            for (int m = 1; m < 4; ++m)
                AA_CHECK_HIP(ctx_memcpy(c, &ps->warm[m][o], inf.data(), (size_t)k * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    // products: aa_prepare's passes on the stacked state; what the RUNNING slots carry (C X, (C X X')',
    // (C X)(C X)') is put back afterwards -- the rest (Z'Z, X X'Z, C X X'Z) comes out as it was
    const size_t wide = (size_t)KP * c->p_pad * sizeof(double), tall = (size_t)c->n_pad * KP * sizeof(double);
    const size_t GS = (size_t)KP * KP;
    AA_CHECK(c->slotSaveP.alloc(wide));
    AA_CHECK(c->slotSaveGr.alloc(tall + GS * sizeof(double)));
    AA_CHECK_HIP(ctx_memcpy(c, c->slotSaveP.p, c->P.p, wide, hipMemcpyDeviceToDevice));
    AA_CHECK_HIP(ctx_memcpy(c, c->slotSaveGr.p, c->Gr.p, tall, hipMemcpyDeviceToDevice));
    double *saveCKCt = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(c->slotSaveGr.p) + tall);
    AA_CHECK_HIP(ctx_memcpy(c, saveCKCt, dev_CKCt(c), GS * sizeof(double), hipMemcpyDeviceToDevice));
    c->products_valid = false;
    c->grams_valid = false;
    AA_CHECK(prepare(c, nullptr));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    const unsigned loaded = c->slots_cold | (1u << r);
    for (int q = 0; q < c->slots_R; ++q) {
        if ((loaded >> q) & 1u) continue;                 // freshly loaded slots keep what prepare computed
        const int oq = q * k;
        AA_CHECK_HIP(ctx_memcpy(c, c->P.as<double>() + (size_t)oq * c->p_pad, c->slotSaveP.as<double>() + (size_t)oq * c->p_pad,
                               (size_t)k * c->p_pad * sizeof(double), hipMemcpyDeviceToDevice));
        AA_CHECK_HIP(ctx_memcpy2d(c, c->Gr.as<double>() + oq, (size_t)KP * sizeof(double),
                                 c->slotSaveGr.as<double>() + oq, (size_t)KP * sizeof(double),
                                 (size_t)k * sizeof(double), (size_t)c->n_pad, hipMemcpyDeviceToDevice));
        AA_CHECK_HIP(ctx_memcpy2d(c, dev_CKCt(c) + (size_t)oq * KP + oq, (size_t)KP * sizeof(double),
                                 saveCKCt + (size_t)oq * KP + oq, (size_t)KP * sizeof(double),
                                 (size_t)k * sizeof(double), (size_t)k, hipMemcpyDeviceToDevice));
    }
    if (c->dtype == AA_F32) AA_CHECK(launch_wide_to_T(c, c->P.as<double>(), operandT(c, c->P, c->Pw)));
    c->slots_cold = loaded;
    c->slots_cold_cols = 0u;
    for (int q = 0; q < c->slots_R; ++q)
        if ((loaded >> q) & 1u)
            for (int i = 0; i < k; ++i) c->slots_cold_cols |= 1u << (q * k + i);
    // initial cost of the new slot only (launch_aa_cost_slots takes all slots: the others' cost0 is
    // rewritten with the same bits -- their Gram blocks are what they were)
    std::vector<double> c0(32);
    AA_CHECK_HIP(ctx_memcpy(c, c0.data(), c->slotCost0.p, 32 * sizeof(double), hipMemcpyDeviceToHost));
    AA_CHECK(launch_aa_cost_slots(c, 0, nullptr));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (int q = 0; q < c->slots_R; ++q)
        if (q != r)
            AA_CHECK_HIP(ctx_memcpy(c, c->slotCost0.as<double>() + q, &c0[q], sizeof(double), hipMemcpyHostToDevice));
    return AA_OK;
}

// n_iters outer iterations of every slot; status[R] out.  The first call after the loads prepares the
// stacked state the way aa_set_state + aa_prepare prepare a single fit (all slots of a group start
// together: the first dictionary update of a fit projects the caller's factors and recomputes the
// products, later ones start from the previous update's -- a host-side choice that the slots share,
// which is why a group is loaded as a whole and a finished slot waits for the others).
int aa_slots_run(aa_ctx *h, int n_iters, aa_slot_status *status)
{
    AA_REQUIRE(h && status && n_iters >= 1, AA_ERR_ARG, "bad argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_aa && c->slots_R > 0, AA_ERR_STATE, "aa_slots_begin first");
    const int R = c->slots_R;
    if (!c->slots_started) {
        c->x_feasible = false;
        AA_CHECK(prepare(c, nullptr));
        AA_CHECK(launch_aa_cost_slots(c, 0, nullptr));
        c->slots_started = true;
        c->slots_cold = slots_all_mask(c);
        c->slots_cold_cols = 0xffffffffu;
    }
    bool recorded = false;
    for (int it = 0; it < n_iters; ++it) {
        if (c->slots_ip.delta != 0.0) {           // archetypal_analysis.py:590-609, once per slot
            AA_CHECK(ensure_ckz(c));
            AA_CHECK(launch_scale_factors(c, &c->slots_scale_sp, c->slots_ip.delta, 0, 0.0, nullptr, nullptr, nullptr,
                                          c->slots_ip.mono_tolerance, c->slots_ip.require_monotonic));
        }
        if (c->slots_cold) {                      // (aa_slots_reload) the cold update of the freshly loaded slots
            c->x_feasible = false;
            c->products_valid = false;
        }
        AA_CHECK(dictionary_update(c, &c->slots_sp, nullptr, true, c->slotCosts.as<double>(),
                                   c->slotCounters.as<int>(), &recorded));
        AA_REQUIRE(recorded, AA_ERR_STATE, "AA slots: the fused line search did not record the cost");
        c->slots_cold = 0;
        c->slots_cold_cols = 0xffffffffu;
        AA_CHECK(weights_update(c, &c->slots_qp, nullptr));
        AA_CHECK(launch_aa_cost_slots(c, 2, &c->slots_ip));
        AA_CHECK(launch_aa_snap_slots(c));
    }
    std::vector<IterState> st(R);
    std::vector<int> cnt(R);
    AA_CHECK_HIP(hipMemcpyAsync(st.data(), c->slotStates.p, (size_t)R * sizeof(IterState), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipMemcpyAsync(cnt.data(), c->slotCounters.p, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (int r = 0; r < R; ++r) {
        status[r].stop = st[r].stop;
        status[r].converged = st[r].converged;
        status[r].error_stage = st[r].error_stage;
        status[r].stop_iter = st[r].stop_iter;
        status[r].not_spd = st[r].spg_flags;          // AA: the SPG warning flags of the slot
        status[r].iterations_run = cnt[r] / 2;
    }
    return AA_OK;
}

// every slot has stopped: the factors of the stopping iterations back into the working arrays and the
// products rebuilt from them (what aa_iterate does for a fit that ran past its stopping iteration)
int aa_slots_finish(aa_ctx *h)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_aa && c->slots_started, AA_ERR_STATE, "nothing to finish");
    AA_CHECK(join_side(c));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    const size_t tall_bytes = (size_t)c->n_pad * c->KP * sizeof(double);
    AA_CHECK_HIP(ctx_memcpy(c, c->Ct.p, c->snapC.p, tall_bytes, hipMemcpyDeviceToDevice));
    AA_CHECK_HIP(ctx_memcpy(c, c->Zt.p, c->snapZ.p, tall_bytes, hipMemcpyDeviceToDevice));
    c->products_valid = false;
    c->grams_valid = false;
    AA_CHECK(prepare(c, nullptr));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    return AA_OK;
}

// factors (C: k x n with leading dimension ldc, Z: n x k), C X (k x p, leading dimension ldx: recomputed
// from the fetched C after aa_slots_finish, or -- carried != 0 -- as the loop carried it at the stopping
// iteration), cost record and initial cost of a stopped slot
int aa_slots_fetch(aa_ctx *h, int r, double *C, long ldc, double *Z, double *CX, long ldx, int carried,
                   double *costs, double *cost0, double *alpha)
{
    AA_REQUIRE(h && C && Z && costs && cost0 && CX, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_aa && r >= 0 && r < c->slots_R, AA_ERR_ARG, "slot %d out of range", r);
    AA_REQUIRE(ldc >= c->n, AA_ERR_ARG, "ldc < n");
    AA_CHECK(join_side(c));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    IterState st;
    AA_CHECK_HIP(ctx_memcpy(c, &st, c->slotStates.as<IterState>() + r, sizeof(st), hipMemcpyDeviceToHost));
    AA_REQUIRE(st.stop, AA_ERR_STATE, "slot %d has not stopped", r);
    const int k = c->slots_k, o = r * k;
    std::vector<double> ct((size_t)c->n * k);
    AA_CHECK_HIP(ctx_memcpy2d(c, ct.data(), (size_t)k * sizeof(double), c->snapC.as<double>() + o, (size_t)c->KP * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyDeviceToHost));
    for (long row = 0; row < c->n; ++row)
        for (int i = 0; i < k; ++i) C[(size_t)i * ldc + row] = ct[(size_t)row * k + i];
    AA_CHECK_HIP(ctx_memcpy2d(c, Z, (size_t)k * sizeof(double), c->snapZ.as<double>() + o, (size_t)c->KP * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyDeviceToHost));
    AA_CHECK_HIP(ctx_memcpy(c, costs, c->slotCosts.as<double>() + (size_t)r * c->slots_stride,
                           (size_t)2 * (st.stop_iter + 1) * sizeof(double), hipMemcpyDeviceToHost));
    AA_CHECK_HIP(ctx_memcpy(c, cost0, c->slotCost0.as<double>() + r, sizeof(double), hipMemcpyDeviceToHost));
    if (alpha)                                                // the scale factors of the stopping iteration
        AA_CHECK_HIP(ctx_memcpy(c, alpha, c->snapAlpha.as<double>() + o, (size_t)k * sizeof(double), hipMemcpyDeviceToHost));
    AA_REQUIRE(ldx >= c->p, AA_ERR_ARG, "ldx < p");
    if (!carried) {
        // C X recomputed from the stopping iteration's dictionary: the pass aa_prepare runs (on the
        // scratch arrays of the dictionary update, free between iterations)
        AA_CHECK_HIP(ctx_memcpy(c, c->Dt.p, c->snapC.p, (size_t)c->n_pad * c->KP * sizeof(double), hipMemcpyDeviceToDevice));
        AA_CHECK(launch_reduce_rows(c, c->Dt.as<double>(), c->Q.as<double>(), nullptr));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    const double *src = (carried ? c->slotSnapP.as<double>() : c->Q.as<double>()) + (size_t)o * c->p_pad;
    AA_CHECK_HIP(ctx_memcpy2d(c, CX, (size_t)ldx * sizeof(double), src, (size_t)c->p_pad * sizeof(double),
                             (size_t)c->p * sizeof(double), (size_t)k, hipMemcpyDeviceToHost));
    return AA_OK;
}

// leaves the slot mode (the context can be used for single fits again)
int aa_slots_end(aa_ctx *h)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK(join_side(c));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    c->slots_aa = false;
    c->slots_R = 0;
    c->slots_started = false;
    c->k = 0;                                         // the next aa_set_state sizes the arrays afresh
    c->have_state = false;
    c->gpnh_valid = false;                            // (the GPNH slot mode ends here too)
    c->grams_valid = false;
    c->ckz_valid = false;
    c->products_valid = false;
    c->qp_iters_valid = false;
    return AA_OK;
}

// ------------------------------------------------------------------ GPNH restarts side by side
// (kernels_tall.hip: GpnhSlots; convex_dim_red/restarts.py drives it).  begin: R empty slots of k
// components each in one set of arrays; load: a restart's start factors into a slot, its initial
// cost; run: outer iterations for all slots, status of every slot back; fetch: factors and cost
// record of a slot that has stopped.  Single rank.
int aa_gpnh_slots_begin(aa_ctx *h, int R, int k, const aa_gpnh_params *gp, const aa_qp_params *qp)
{
    AA_REQUIRE(h && gp && qp, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->have_data && c->form == AA_FORM_DATA, AA_ERR_STATE, "GPNH needs a data matrix");
    AA_REQUIRE(c->world <= 1 && !c->force_comm, AA_ERR_STATE, "restart slots are single-rank");
    AA_REQUIRE(R >= 1 && k >= 1 && k <= 32 && R * k <= AA_MAX_K && R <= 32, AA_ERR_ARG,
               "slots: R = %d restarts of k = %d components do not fit", R, k);
    const aa_iter_params *ip = &gp->loop;
    AA_REQUIRE(ip->max_outer >= 1 && ip->update_dictionary && ip->update_weights, AA_ERR_ARG,
               "slots: both updates, max_outer >= 1");
    AA_REQUIRE(ip->criterion == 0 || ip->criterion == 1, AA_ERR_ARG, "bad stopping criterion");
    AA_REQUIRE(qp->max_iterations >= 1 && qp->memory <= 8, AA_ERR_ARG, "slots: QP max_iterations >= 1, memory <= 8");
    AA_REQUIRE(qp->max_iterations <= 4 || (qp->memory <= 1 && k <= 16 && c->n < 65536), AA_ERR_ARG,
               "slots: QPs of more than four passes need memory 1, k <= 16, fewer than 65 536 samples");
    c->k = 0;                                         // force fresh, zeroed factor arrays
    c->slots_aa = false;                              // (an AA slot run that was never ended)
    c->slots_started = false;
    c->products_valid = false;
    AA_CHECK(ensure_problem(c, R * k));
    AA_CHECK(ensure_trace(c));
    for (int i = 0; i < c->k; ++i) c->alpha[i] = 1.0;
    AA_CHECK(upload_alpha(c));
    c->slots_R = R;
    c->slots_k = k;
    c->slots_max_outer = ip->max_outer;
    c->slots_stride = 2 * ip->max_outer + 64;
    c->slots_gp = *gp;
    c->slots_qp = *qp;
    AA_CHECK(c->slotCosts.alloc((size_t)R * c->slots_stride * sizeof(double)));
    AA_CHECK(c->slotCounters.alloc(64 * sizeof(int)));
    AA_CHECK(c->slotStates.alloc(32 * sizeof(IterState)));
    AA_CHECK(c->slotCost0.alloc(32 * sizeof(double)));
    size_t snap_bytes = (size_t)c->n_pad * c->KP * sizeof(double);
    if ((size_t)c->KP * c->p_pad * sizeof(double) > snap_bytes) snap_bytes = (size_t)c->KP * c->p_pad * sizeof(double);
    AA_CHECK(c->snapC.alloc(snap_bytes));
    AA_CHECK(c->snapZ.alloc(snap_bytes));
    std::vector<IterState> st(32);
    memset(st.data(), 0, st.size() * sizeof(IterState));
    for (int r = 0; r < 32; ++r) st[r].stop = 1;      // empty: the judge leaves it alone
    AA_CHECK_HIP(ctx_memcpy(c, c->slotStates.p, st.data(), st.size() * sizeof(IterState), hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memset(c, c->slotCounters.p, 0, 64 * sizeof(int)));
    AA_CHECK_HIP(ctx_memset(c, c->Zt.p, 0, (size_t)c->n_pad * c->KP * sizeof(double)));
    AA_CHECK_HIP(ctx_memset(c, c->P.p, 0, (size_t)c->KP * c->p_pad * sizeof(double)));
    AA_CHECK_HIP(ctx_memset(c, c->gramState.p, 0, (size_t)3 * c->KP * c->KP * sizeof(double)));
    c->gpnh_valid = true;
    c->have_state = true;
    c->grams_valid = false;
    c->ckz_valid = false;
    c->host_grams_valid = false;
    c->qp_iters_valid = false;
    return AA_OK;
}

int aa_gpnh_slots_load(aa_ctx *h, int r, const double *Wt, long ld, const double *Z)
{
    AA_REQUIRE(h && Wt && Z, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_R > 0 && r >= 0 && r < c->slots_R, AA_ERR_ARG, "slot %d out of range", r);
    AA_REQUIRE(ld >= c->p, AA_ERR_ARG, "ld < p");
    const int k = c->slots_k, o = r * k;
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_CHECK_HIP(ctx_memcpy2d(c, c->P.as<double>() + (size_t)o * c->p_pad, (size_t)c->p_pad * sizeof(double), Wt,
                             (size_t)ld * sizeof(double), (size_t)c->p * sizeof(double), (size_t)k,
                             hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memcpy2d(c, c->Zt.as<double>() + o, (size_t)c->KP * sizeof(double), Z, (size_t)k * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyHostToDevice));
    IterState zero;
    memset(&zero, 0, sizeof(zero));
    AA_CHECK_HIP(ctx_memcpy(c, c->slotStates.as<IterState>() + r, &zero, sizeof(zero), hipMemcpyHostToDevice));
    AA_CHECK_HIP(ctx_memset(c, c->slotCounters.as<int>() + r, 0, sizeof(int)));
    // the products of the stacked factors (the other slots' parts come out as they were) and this
    // slot's initial cost, the way aa_gpnh_iterate forms it
    AA_CHECK(launch_wide_to_T(c, c->P.as<double>(), operandT(c, c->P, c->Pw)));
    AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gr.as<double>()));          // X W
    AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), dev_ZtZ(c)));
    AA_CHECK(launch_gram_wide(c, c->P.as<double>(), c->P.as<double>(), dev_CKCt(c)));
    AA_CHECK(launch_reduce_rows(c, c->Zt.as<double>(), c->ZtX.as<double>(), nullptr));
    AA_CHECK(launch_gpnh_cost_slots(c, c->slots_gp.lambda_W, 1u << r, 0, nullptr, false));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    return AA_OK;
}

int aa_gpnh_slots_run(aa_ctx *h, int n_iters, aa_slot_status *status)
{
    AA_REQUIRE(h && status && n_iters >= 1, AA_ERR_ARG, "bad argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_R > 0, AA_ERR_STATE, "aa_gpnh_slots_begin first");
    const int R = c->slots_R, k = c->slots_k;
    const double lambda = c->slots_gp.lambda_W;
    const aa_iter_params *ip = &c->slots_gp.loop;
    const unsigned all = R >= 32 ? 0xffffffffu : ((1u << R) - 1u);
    // the single fit forms W'W inside its cost kernel when the factor is small (gpnh_cost_can_gram
    // with ITS k); the slots follow the same rule so that every restart sees the same bits
    const bool gram_in_cost = k * k <= 256 && (long)k * c->p_pad <= 4096;
    for (int it = 0; it < n_iters; ++it) {
        AA_CHECK(launch_gpnh_solve_slots(c, lambda));                                      // W'
        AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gr.as<double>()));       // X W
        if (!gram_in_cost) AA_CHECK(launch_gram_wide(c, c->P.as<double>(), c->P.as<double>(), dev_CKCt(c)));
        AA_CHECK(launch_gpnh_cost_slots(c, lambda, all, 1, ip, gram_in_cost));
        // at most four SPG passes per QP: the lane-per-sample kernel, as in a single fit; more: the
        // four-lane and wave-per-sample kernels of the AA slots (the Hessian blocks are W'W's)
        if (c->slots_qp.max_iterations <= 4) AA_CHECK(launch_qp_slots(c, R, k, dev_CKCt(c), &c->slots_qp));
        else AA_CHECK(launch_qp_slots_aa(c, &c->slots_qp));
        AA_CHECK(launch_gram_tall(c, c->Zt.as<double>(), c->Zt.as<double>(), dev_ZtZ(c)));
        AA_CHECK(launch_reduce_rows(c, c->Zt.as<double>(), c->ZtX.as<double>(), nullptr));
        AA_CHECK(launch_gpnh_cost_slots(c, lambda, all, 2, ip, false));
        AA_CHECK(launch_gpnh_snap_slots(c));
    }
    std::vector<IterState> st(R);
    std::vector<int> cnt(R);
    AA_CHECK_HIP(hipMemcpyAsync(st.data(), c->slotStates.p, (size_t)R * sizeof(IterState), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipMemcpyAsync(cnt.data(), c->slotCounters.p, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (int r = 0; r < R; ++r) {
        status[r].stop = st[r].stop;
        status[r].converged = st[r].converged;
        status[r].error_stage = st[r].error_stage;
        status[r].stop_iter = st[r].stop_iter;
        status[r].not_spd = st[r].pad0;
        status[r].iterations_run = cnt[r] / 2;
    }
    return AA_OK;
}

int aa_gpnh_slots_fetch(aa_ctx *h, int r, double *Wt, long ld, double *Z, double *costs, double *cost0)
{
    AA_REQUIRE(h && Wt && Z && costs && cost0, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_REQUIRE(c->slots_R > 0 && r >= 0 && r < c->slots_R, AA_ERR_ARG, "slot %d out of range", r);
    AA_REQUIRE(ld >= c->p, AA_ERR_ARG, "ld < p");
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    IterState st;
    AA_CHECK_HIP(ctx_memcpy(c, &st, c->slotStates.as<IterState>() + r, sizeof(st), hipMemcpyDeviceToHost));
    AA_REQUIRE(st.stop, AA_ERR_STATE, "slot %d has not stopped", r);
    const int k = c->slots_k, o = r * k;
    // the factors of the stopping iteration (k_gpnh_snap_slots)
    AA_CHECK_HIP(ctx_memcpy2d(c, Wt, (size_t)ld * sizeof(double), c->snapC.as<double>() + (size_t)o * c->p_pad,
                             (size_t)c->p_pad * sizeof(double), (size_t)c->p * sizeof(double), (size_t)k,
                             hipMemcpyDeviceToHost));
    AA_CHECK_HIP(ctx_memcpy2d(c, Z, (size_t)k * sizeof(double), c->snapZ.as<double>() + o, (size_t)c->KP * sizeof(double),
                             (size_t)k * sizeof(double), (size_t)c->n, hipMemcpyDeviceToHost));
    AA_CHECK_HIP(ctx_memcpy(c, costs, c->slotCosts.as<double>() + (size_t)r * c->slots_stride,
                           (size_t)2 * (st.stop_iter + 1) * sizeof(double), hipMemcpyDeviceToHost));
    AA_CHECK_HIP(ctx_memcpy(c, cost0, c->slotCost0.as<double>() + r, sizeof(double), hipMemcpyDeviceToHost));
    return AA_OK;
}

int aa_gpnh_residual_cost(aa_ctx *h, double *cost)
{
    AA_REQUIRE(h && cost, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->have_state && c->gpnh_valid, AA_ERR_STATE, "set_factors first");
    AA_CHECK_HIP(hipSetDevice(c->device));
    double s = 0.0;
    AA_CHECK(launch_residual_cost(c, c->Zt.as<double>(), c->P.as<double>(), nullptr, &s));
    *cost = 0.5 * s / (double)c->n_global;
    return AA_OK;
}

// ------------------------------------------------------------------ stateless ops
int aa_simplex_project_rows(int device, const double *in, double *out, long rows, long cols)
{
    AA_REQUIRE(in && out && rows >= 0 && cols >= 0, AA_ERR_ARG, "bad arguments");
    if (rows == 0 || cols == 0) return AA_OK;
    int n = 0;
    AA_CHECK_HIP(hipGetDeviceCount(&n));
    AA_REQUIRE(n > 0 && device >= 0 && device < n, AA_ERR_HIP, "no usable HIP device");
    AA_CHECK_HIP(hipSetDevice(device));
    DevBuf a, b;
    const size_t bytes = (size_t)rows * cols * sizeof(double);
    AA_CHECK(a.alloc(bytes));
    int rc = b.alloc(bytes);
    if (rc == AA_OK) {
        // (everything of this stateless entry point runs on the null stream, waited for by the host)
        hipError_t e = hipMemcpy(a.p, in, bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            rc = launch_simplex_rows_generic(nullptr, a.as<double>(), b.as<double>(), rows, cols);
            if (rc == AA_OK) e = hipDeviceSynchronize();
            if (rc == AA_OK && e == hipSuccess) e = hipMemcpy(out, b.p, bytes, hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) {
            set_error("simplex_project_rows: %s", hipGetErrorString(e));
            rc = AA_ERR_HIP;
        }
    }
    a.release();
    b.release();
    return rc;
}

int aa_quad_simplex_spg_batch(int device, const double *A, const double *B, long stride_j, long stride_t,
                              const double *Z0, double *Zout, long n, int k, const aa_qp_params *params,
                              int *iters)
{
    AA_REQUIRE(A && B && Z0 && Zout && params, AA_ERR_ARG, "null argument");
    AA_REQUIRE(n >= 0 && k >= 1 && k <= AA_MAX_K, AA_ERR_ARG, "bad n=%ld k=%d", n, k);
    AA_REQUIRE(stride_j >= 1 && stride_t >= 1, AA_ERR_ARG, "bad strides");
    if (n == 0) return AA_OK;
    // one scratch context per device for the stateless entry points, kept for the life of the
    // process: creating and destroying a context (two streams, events, allocations) per call
    // cost 8 ms -- more than the QPs of every unit-test-sized problem
    // (ctypes releases the GIL: calls from several threads take turns on the scratch context; it
    // is never destroyed -- HIP may already be gone when static destructors run)
    static aa_ctx *scratch[64] = {nullptr};
    static std::mutex scratch_mu;
    AA_REQUIRE(device >= 0 && device < 64, AA_ERR_ARG, "device %d out of range", device);
    std::lock_guard<std::mutex> hold(scratch_mu);
    if (!scratch[device]) AA_CHECK(aa_ctx_create(&scratch[device], device, AA_F64));
    aa_ctx *h = scratch[device];
    AA_CHECK_HIP(hipSetDevice(device));
    Ctx *c = &h->c;
    c->qp_iters_valid = false;
    DevBuf dB, dZ, dI;
    const size_t extent = (size_t)((k - 1) * stride_j + (n - 1) * stride_t + 1);
    int rc = dB.alloc(extent * sizeof(double));
    if (rc == AA_OK) rc = dZ.alloc((size_t)n * k * sizeof(double));
    if (rc == AA_OK) rc = dI.alloc((size_t)n * sizeof(int));
    if (rc == AA_OK) {
        hipError_t e = ctx_memcpy(c, dB.p, B, extent * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = ctx_memcpy(c, dZ.p, Z0, (size_t)n * k * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("qp batch upload: %s", hipGetErrorString(e));
            rc = AA_ERR_HIP;
        }
    }
    aa_qp_stats prof_stats;
    if (rc == AA_OK)
        rc = launch_qp(c, A, dB.as<double>(), stride_j, stride_t, nullptr, dZ.as<double>(), k, n, k, params,
                       dI.as<int>(), g_qp_profile ? &prof_stats : (aa_qp_stats *)nullptr);
    if (rc == AA_OK) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = ctx_memcpy(c, Zout, dZ.p, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess && iters) e = ctx_memcpy(c, iters, dI.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error("qp batch: %s", hipGetErrorString(e));
            rc = AA_ERR_HIP;
        }
    }
    dB.release();
    dZ.release();
    dI.release();
    return rc;
}

int aa_get_spg_scalars(aa_ctx *h, double *out)
{
    AA_REQUIRE(h && out, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(c->scalars.p, AA_ERR_STATE, "no solver state");
    AA_CHECK(join_side(c));
    static_assert(AA_SPG_SCALARS == SC_FMEM0, "include/aa_hip.h documents the ScalarSlot layout");
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_CHECK_HIP(ctx_memcpy(c, out, c->scalars.p, AA_SPG_SCALARS * sizeof(double), hipMemcpyDeviceToHost));
    return AA_OK;
}

// ------------------------------------------------------------------ the two passes, on their own
static int invalidate_solver_state(Ctx *c)
{
    c->have_state = false;
    c->grams_valid = false;
    c->products_valid = false;
    c->ckz_valid = false;
    c->gpnh_valid = false;
    c->x_feasible = false;
    c->qp_iters_valid = false;
    return AA_OK;
}

int aa_pass_reduce_rows(aa_ctx *h, int k, const double *A, double *out, long ldo)
{
    AA_REQUIRE(h && A && out, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(!c->implicit_kernel, AA_ERR_STATE, "aa_pass_reduce_rows: no stored matrix behind an implicit kernel");
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK(ensure_problem(c, k));
    AA_REQUIRE(ldo >= (c->form == AA_FORM_KERNEL ? c->n : c->p), AA_ERR_ARG, "ldo too small");
    AA_CHECK(invalidate_solver_state(c));
    AA_CHECK(upload_tall(c, c->Dt, A, k, 1, c->n, k));
    AA_CHECK(launch_reduce_rows(c, c->Dt.as<double>(), c->Q.as<double>(), nullptr));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_CHECK_HIP(ctx_memcpy2d(c, out, (size_t)ldo * sizeof(double), c->Q.p, (size_t)c->p_pad * sizeof(double),
                             (size_t)c->p * sizeof(double), (size_t)k, hipMemcpyDeviceToHost));
    return AA_OK;
}

int aa_pass_row_local(aa_ctx *h, int k, const double *B, long ldb, double *out)
{
    AA_REQUIRE(h && B && out, AA_ERR_ARG, "null argument");
    Ctx *c = &h->c;
    AA_REQUIRE(!c->implicit_kernel, AA_ERR_STATE, "aa_pass_row_local: no stored matrix behind an implicit kernel");
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK(ensure_problem(c, k));
    AA_REQUIRE(ldb >= c->p, AA_ERR_ARG, "ldb < p");
    AA_CHECK(invalidate_solver_state(c));
    std::vector<double> tmp((size_t)c->KP * c->p_pad, 0.0);
    for (int i = 0; i < k; ++i)
        for (long q = 0; q < c->p; ++q) tmp[(size_t)i * c->p_pad + q] = B[(size_t)i * ldb + q];
    AA_CHECK_HIP(ctx_memcpy(c, c->P.p, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    AA_CHECK(launch_wide_to_T(c, c->P.as<double>(), operandT(c, c->P, c->Pw)));
    AA_CHECK(launch_row_local(c, operandT(c, c->P, c->Pw), c->Gn.as<double>()));
    return download_tall(c, c->Gn, out, k, 1, c->n, k);
}

// ------------------------------------------------------------------ measurement
int aa_pass_kernels(aa_ctx *h, char *buf, int len)
{
    AA_REQUIRE(h && buf, AA_ERR_ARG, "null argument");
    const Ctx *c = &h->c;
    const int need = snprintf(nullptr, 0, "%s;%s", c->pass_names[0], c->pass_names[1]) + 1;
    AA_REQUIRE(len >= need, AA_ERR_ARG, "aa_pass_kernels: buffer of %d bytes, %d needed", len, need);
    snprintf(buf, (size_t)len, "%s;%s", c->pass_names[0], c->pass_names[1]);
    return AA_OK;
}

int aa_gemm_timing(aa_ctx *h, int enable, double *ms_reduce_rows, int *n_reduce_rows,
                   double *ms_row_local, int *n_row_local)
{
    AA_REQUIRE(h, AA_ERR_ARG, "null ctx");
    Ctx *c = &h->c;
    AA_CHECK_HIP(hipSetDevice(c->device));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    double *ms_out[2] = {ms_reduce_rows, ms_row_local};
    int *n_out[2] = {n_reduce_rows, n_row_local};
    for (int w = 0; w < 2; ++w) {
        std::vector<hipEvent_t> &ev = c->gemmEvents[w];
        double total = 0.0;
        int n = 0;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) {
                total += ms;
                ++n;
            }
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
        if (ms_out[w]) *ms_out[w] = n ? total / n : 0.0;
        if (n_out[w]) *n_out[w] = n;
    }
    c->time_gemm = enable != 0;
    return AA_OK;
}

int aa_time_kernel(aa_ctx *h, int which, int reps, double *ms_avg)
{
    AA_REQUIRE(h && ms_avg && reps >= 1, AA_ERR_ARG, "bad arguments");
    Ctx *c = &h->c;
    AA_REQUIRE(!c->implicit_kernel, AA_ERR_STATE, "aa_time_kernel: no stored matrix behind an implicit kernel");
    AA_REQUIRE(c->have_state && c->form == AA_FORM_DATA, AA_ERR_STATE, "needs data-form state");
    AA_CHECK_HIP(hipSetDevice(c->device));
    hipEvent_t e0, e1;
    AA_CHECK_HIP(hipEventCreate(&e0));
    AA_CHECK_HIP(hipEventCreate(&e1));
    int rc = AA_OK;
    // one untimed launch first
    for (int phase = 0; phase < 2 && rc == AA_OK; ++phase) {
        const int nrep = phase == 0 ? 1 : reps;
        if (phase == 1) (void)hipEventRecord(e0, c->stream);
        for (int r = 0; r < nrep && rc == AA_OK; ++r) {
            if (which == 0)
                rc = launch_reduce_rows(c, c->Ct.as<double>(), c->Q.as<double>(), nullptr, true);
            else if (which == 1)
                rc = launch_row_local(c, operandT(c, c->P, c->Pw), c->Gn.as<double>());
            else if (which >= 2 && which <= 7)
                rc = launch_stream_probe(c, which - 2);
            else {
                set_error("aa_time_kernel: unknown kernel %d", which);
                rc = AA_ERR_ARG;
            }
        }
        if (phase == 1) (void)hipEventRecord(e1, c->stream);
    }
    if (rc == AA_OK) {
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            set_error("aa_time_kernel: %s", hipGetErrorString(e));
            rc = AA_ERR_HIP;
        }
        *ms_avg = (double)ms / reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
