// kernels_tall.hip -- the HBM/latency-bound float64 kernels around the GEMMs:
// column-wise simplex projection of the transposed dictionary (the reference's
// simplex_project_rows on the k x n dictionary, simplex_projection.py:40-47, called at
// spg.py:148,184,193,250), gradient assembly (archetypal_analysis.py:284-301), SPG
// reductions (spg.py:206,237-238,252), small Gram products (archetypal_analysis.py:
// 543-556,618-621,640-643), and the single-thread "scalar stage" kernels that carry
// the SPG control arithmetic (spg.py:19-43,153-266) on the device.
//
// "tall" arrays are [n_pad][KP] doubles, component on the fast axis: thread t of a
// 256-thread block handles component t % KP of row (t / KP) + RS*step, so every wave
// touches whole 256-byte rows (coalesced).  Every reduction is two-stage with a fixed
// summation order (per-block partials, then one finalize block) => deterministic,
// and the finalize step is where the multi-GPU all-reduce is spliced in.
//
// The simplex projection does not sort: per column it runs Michelot's fixed point
//   S <- {w > t},  t <- (sum_S w - 1)/|S|,
// which ends at exactly the support the reference's sorted scan finds and the same
// closed form t = (sum of the m largest - 1)/m (simplex_projection.py:23).  Default form
// ("candidate lists", below): one pass for a lower bound of t, one pass that copies the few
// candidates above it, the fixed point on that short list, one pass that applies t.  The
// iterative form (one full pass per Michelot round) is the fallback of the multi-rank path
// and stays selectable (aa_set_option("proj_mode", 1)).
#include "aa_internal.h"

#include <utility>

namespace aa {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- helpers
__device__ __forceinline__ void store_agent(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// agent: the block's results go out with agent-scope stores, drained before the closing barrier (FinTail)
template <int KP, int NV, int NT = 256>
__device__ __forceinline__ void block_col_combine(const double (&v)[NV], unsigned max_mask,
                                                  double *sm, double *dst, bool agent = false)
{
    constexpr int RS = NT / KP;
    const int t = threadIdx.x, comp = t % KP, rsub = t / KP;
#pragma unroll
    for (int a = 0; a < NV; ++a) sm[a * NT + t] = v[a];
    __syncthreads();
    if (rsub < NV) {                       // value a = rsub is combined by row-subgroup a
        const int a = rsub;
        double s = sm[a * NT + comp];
        if ((max_mask >> a) & 1u) {
            for (int q = 1; q < RS; ++q) s = fmax(s, sm[a * NT + q * KP + comp]);
        } else {
            for (int q = 1; q < RS; ++q) s += sm[a * NT + q * KP + comp];
        }
        if (agent) {
            store_agent(&dst[a * KP + comp], s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            dst[a * KP + comp] = s;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ double load_a(double a_const, const double *scal, int a_slot)
{
    return a_slot >= 0 ? scal[a_slot] : a_const;
}
// restarts side by side (kslot > 0: components per slot): column `comp` takes the scalar of ITS slot
__device__ __forceinline__ double load_a(double a_const, const double *scal, int a_slot, int comp, int kslot)
{
    return a_slot >= 0 ? scal[(kslot > 0 ? (comp / kslot) * AA_SC_STRIDE : 0) + a_slot] : a_const;
}

#define PROJ_NT 1024       // threads per block of the first / finish passes of a projection

// second stage + scalar step, defined further down
enum { POST_NONE = -1, POST_COLMAX = 0, POST_MICHELOT, POST_FIN, POST_SCALAR_SUM, POST_FIRST };
#define FIN_NT 1024
__device__ void post_step(int kind, int mode, const double *__restrict__ red, int KP, int k,
                          ProjState *__restrict__ ps, double *__restrict__ scal, int slot);
template <bool AGENT = false>
__device__ __forceinline__ void finalize_sum_block(const double *partial, int nb, int NV,
                                                   int KP, unsigned max_mask, double *__restrict__ red,
                                                   double *sm, double *__restrict__ gather, int rank,
                                                   int world);
__device__ void scalar_stage_simple(int stage, double *__restrict__ sc, const aa_spg_params &sp);
__device__ void post_step_slots(int kind, int mode, const double *__restrict__ red, int KP, int kslot, int R,
                                ProjState *__restrict__ ps, double *__restrict__ scal, int slot, int stage_after,
                                const aa_spg_params &sp);

// The second reduction stage of a pass (k_finalize_sum's work) done by the LAST block of the pass to
// arrive instead of by a launch of its own: one launch, ~6 us of dependent-launch latency, less per
// reduction -- four of them sit in front of the Z'X pass of every dictionary update.  Single rank only
// (the multi-rank stage has a collective in the middle).  Every block writes its partials with
// agent-scope stores and drains them (the XCDs' L2s are not coherent with each other; a full release
// fence -- an L2 write-back per block, of the pass's whole output -- cost 100 us per pass), counts
// itself in; the block that counts last reads all partials with agent-scope loads in the fixed order
// of finalize_sum_block, so the bits do not depend on which block that was.
struct FinTail {
    int on;                 // 0: the caller launches k_finalize_sum
    int NV;
    unsigned max_mask;
    double *red;
    int kind, mode, k;
    ProjState *ps;          // (its fin_arrived is the counter)
    double *scal;
    int slot, stage_after;
    aa_spg_params sp;
    int kslot, R;
};

template <int KP>
__device__ __forceinline__ void fin_tail(const FinTail &ft, const double *__restrict__ partial, double *sm /* 4 * FIN_NT */)
{
    // (block_col_combine(agent) has drained this block's partials and closed with a barrier)
    __shared__ int fin_last;
    if (threadIdx.x == 0)
        fin_last = __hip_atomic_fetch_add(&ft.ps->fin_arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                   (int)gridDim.x - 1;
    __syncthreads();
    if (!fin_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(&ft.ps->fin_arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    finalize_sum_block<true>(partial, (int)gridDim.x, ft.NV, KP, ft.max_mask, ft.red, sm, (double *)nullptr, 0, 1);
    if (ft.kslot > 0 && (ft.kind == POST_FIN || ft.kind == POST_SCALAR_SUM)) {
        post_step_slots(ft.kind, ft.mode, ft.red, KP, ft.kslot, ft.R, ft.ps, ft.scal, ft.slot, ft.stage_after, ft.sp);
        return;
    }
    if (ft.kind != POST_NONE) post_step(ft.kind, ft.mode, ft.red, KP, ft.k, ft.ps, ft.scal, ft.slot);
    if (ft.stage_after >= 0) {
        __syncthreads();
        if (threadIdx.x == 0) scalar_stage_simple(ft.stage_after, ft.scal, ft.sp);
    }
}

// ---------------------------------------------------------------- projection passes
// w[r][i] = x[r][i] - a * g[r][i]   (g == nullptr => w = x)
// first pass of a projection: column maxima, and w = x - a*g written out once so the
// Michelot passes stream one array instead of two
template <int KP>
__global__ __launch_bounds__(256) void k_proj_colmax(const double *__restrict__ x,
                                                     const double *__restrict__ g, double a_const,
                                                     const double *__restrict__ scal, int a_slot,
                                                     long n, long rows_pb, int k,
                                                     double *__restrict__ wout,
                                                     double *__restrict__ partial)
{
    __shared__ double sm[256];
    constexpr int RS = 256 / KP;
    const int t = threadIdx.x, comp = t % KP, rsub = t / KP;
    const double a = load_a(a_const, scal, a_slot);
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    if (re > n) re = n;
    double m[1] = {-INFINITY};
    if (comp < k)
        for (long r = rb + rsub; r < re; r += RS) {
            const double w = g ? x[r * KP + comp] - a * g[r * KP + comp] : x[r * KP + comp];
            if (wout) wout[r * KP + comp] = w;
            m[0] = fmax(m[0], w);
        }
    block_col_combine<KP, 1>(m, 1u, sm, partial + (size_t)blockIdx.x * KP);
}

template <int KP>
__global__ __launch_bounds__(256) void k_proj_pass(const double *__restrict__ x,
                                                   const double *__restrict__ g, double a_const,
                                                   const double *__restrict__ scal, int a_slot,
                                                   long n, long rows_pb, int k,
                                                   const ProjState *__restrict__ ps,
                                                   double *__restrict__ partial)
{
    if (ps->done) return;   // uniform: every thread reads the same word
    __shared__ double sm[2 * 256];
    constexpr int RS = 256 / KP;
    const int t = threadIdx.x, comp = t % KP, rsub = t / KP;
    const double a = load_a(a_const, scal, a_slot);
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    if (re > n) re = n;
    double v[2] = {0.0, 0.0};
    if (comp < k) {
        const double th = ps->t[comp];
        for (long r = rb + rsub; r < re; r += RS) {
            const double w = g ? x[r * KP + comp] - a * g[r * KP + comp] : x[r * KP + comp];
            if (w > th) {
                v[0] += w;
                v[1] += 1.0;
            }
        }
    }
    block_col_combine<KP, 2>(v, 0u, sm, partial + (size_t)blockIdx.x * 2 * KP);
}

// mode: PROJ_FEAS  out = max(w - t, 0)                                   (spg.py:148)
//       PROJ_ALPHA v3 = max |max(w-t,0) - x|                             (spg.py:184)
//       PROJ_DIR   out = d = max(w-t,0) - x; v0 = <d,g>, v1 = <d,d>, v2 = <d, H alpha>
//       PROJ_RES   v0 = sum res^2, v3 = max |res|                        (spg.py:250-263)
template <int KP>
__global__ __launch_bounds__(PROJ_NT) void k_proj_finish(int mode, const double *__restrict__ x,
                                                     const double *__restrict__ g, double a_const,
                                                     const double *__restrict__ scal, int a_slot,
                                                     const double *__restrict__ H,
                                                     const double *__restrict__ alpha, long n,
                                                     long rows_pb, int k,
                                                     const ProjState *ps,
                                                     double *__restrict__ out,
                                                     double *__restrict__ partial, int kslot,
                                                     unsigned colmask, FinTail ft)
{
    static_assert(PROJ_NT == FIN_NT, "fin_tail: the last block runs finalize_sum_block");
    __shared__ double sm[4 * PROJ_NT];
    constexpr int RS = PROJ_NT / KP;
    const int t = threadIdx.x, comp = t % KP, rsub = t / KP;
    const double a = load_a(a_const, scal, a_slot, comp, kslot);
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    if (re > n) re = n;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (comp < k) {
        const double th = ps->t[comp];
        const double al = alpha ? alpha[comp] : 1.0;
#pragma unroll 4
        for (long r = rb + rsub; r < re; r += RS) {
            const long e = r * KP + comp;
            const double xe = x[e];
            const double ge = g ? g[e] : 0.0;
            const double w = g ? xe - a * ge : xe;
            const double pr = fmax(w - th, 0.0);
            if (mode == PROJ_FEAS) {
                // (restart slots: only the columns of freshly loaded slots are projected -- colmask)
                if ((colmask >> (comp & 31)) & 1u) out[e] = pr;
            } else {
                const double d = pr - xe;
                if (mode == PROJ_DIR) {
                    out[e] = d;
                    v[0] += d * ge;
                    v[1] += d * d;
                    v[2] += d * H[e] * al;
                } else if (mode == PROJ_RES) {
                    v[0] += d * d;
                    v[3] = fmax(v[3], fabs(d));
                } else {
                    v[3] = fmax(v[3], fabs(d));
                }
            }
        }
    }
    block_col_combine<KP, 4, PROJ_NT>(v, 8u, sm, partial + (size_t)blockIdx.x * 4 * KP, ft.on != 0);
    if (ft.on) fin_tail<KP>(ft, partial, sm);
}

// ---------------------------------------------------------------- candidate-list projection
// Single-rank fast path.  Every Michelot iterate from ANY threshold with a non-empty
// support is a Newton step on the convex, decreasing phi(t) = sum max(w - t, 0) - 1 and
// therefore a lower bound of the root t*.  So after ONE pass (column maxima plus the
// Newton step from the previous projection's threshold, fused) the candidates
// {w > t_lower} contain the support; they are copied once into per-thread segments (no
// atomics, fixed order => deterministic) and one block per column finishes the fixed point
// on that short list.  3 passes over the array + 1 tiny kernel, no host synchronisation,
// instead of 2 + (Michelot passes) full passes with a host check.
//
// first pass: w = x - a*g (written once), v0 = max w, v1/v2 = sum / count of {w > warm t}
template <int KP>
__global__ __launch_bounds__(PROJ_NT) void k_proj_first(const double *__restrict__ x,
                                                    const double *__restrict__ g, double a_const,
                                                    const double *__restrict__ scal, int a_slot,
                                                    long n, long rows_pb, int k, int warm_slot,
                                                    const ProjState *ps,
                                                    double *__restrict__ wout,
                                                    double *__restrict__ partial, int kslot, FinTail ft)
{
    __shared__ double sm[4 * PROJ_NT];
    constexpr int RS = PROJ_NT / KP;
    const int t = threadIdx.x, comp = t % KP, rsub = t / KP;
    const double a = load_a(a_const, scal, a_slot, comp, kslot);
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    if (re > n) re = n;
    double v[3] = {-INFINITY, 0.0, 0.0};
    if (comp < k) {
        const double th = warm_slot > 0 ? ps->warm[warm_slot][comp] : INFINITY;
#pragma unroll 4
        for (long r = rb + rsub; r < re; r += RS) {
            const double w = g ? x[r * KP + comp] - a * g[r * KP + comp] : x[r * KP + comp];
            if (wout) wout[r * KP + comp] = w;
            v[0] = fmax(v[0], w);
            if (w > th) {
                v[1] += w;
                v[2] += 1.0;
            }
        }
    }
    block_col_combine<KP, 3, PROJ_NT>(v, 1u, sm, partial + (size_t)blockIdx.x * 3 * KP, ft.on != 0);
    if (ft.on) fin_tail<KP>(ft, partial, sm);
}

// candidates {w > t_lower}: thread (rsub, comp) of block b appends its own rows, in row
// order, to segment (comp, b*RS + rsub) of `list` ([KP][nseg][segcap]); counts in segcnt
template <int KP>
__global__ __launch_bounds__(256) void k_proj_collect(const double *__restrict__ w, long n,
                                                      long rows_pb, int k,
                                                      const ProjState *__restrict__ ps,
                                                      double *__restrict__ list,
                                                      int *__restrict__ segcnt)
{
    constexpr int RS = 256 / KP;
    const int t = threadIdx.x, comp = t % KP, rsub = t / KP;
    if (comp >= k) return;
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    if (re > n) re = n;
    const long segcap = rows_pb / RS;
    const long nseg = (long)gridDim.x * RS;
    const long seg = (long)blockIdx.x * RS + rsub;
    double *dst = list + ((long)comp * nseg + seg) * segcap;
    const double th = ps->t[comp];
    int cnt = 0;
    // eight rows per step with their loads issued together (the conditional store would
    // otherwise serialise one memory latency per row)
    long r = rb + rsub;
    for (; r + 7 * RS < re; r += 8 * RS) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w[(r + u * RS) * KP + comp];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] > th) dst[cnt++] = v[u];
    }
    for (; r < re; r += RS) {
        const double v = w[r * KP + comp];
        if (v > th) dst[cnt++] = v;
    }
    segcnt[(long)comp * nseg + seg] = cnt;
}

// one block per column: Michelot's fixed point on the candidate list, started from the
// lower bound.  Lists of <= PROJ_LDS_CAP candidates are gathered into LDS first.
#define PROJ_LDS_CAP 2048
#define PROJ_SPT 8          // segments per thread the register-blocked gather handles

// fixed-order block sum of (s, m) over 256 threads: xor-shuffle tree inside each wave, then
// the four wave totals in order => deterministic, two barriers
__device__ __forceinline__ void block_sum_sm(double &s, int &m, double *rs, int *rm)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        m += __shfl_xor(m, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        rs[w] = s;
        rm[w] = m;
    }
    __syncthreads();
    s = ((rs[0] + rs[1]) + rs[2]) + rs[3];
    m = rm[0] + rm[1] + rm[2] + rm[3];
    __syncthreads();
}

// Michelot's fixed point from the lower bound `th` on the `total` candidates in LDS (u);
// all 256 threads; returns the threshold, the support size and whether it converged
__device__ __forceinline__ void proj_solve_lds(const double *u, int total, double &th, int &prev,
                                               int &conv, double *rs, int *rm)
{
    const int t = threadIdx.x;
    for (int it = 0; it < 200 && !conv; ++it) {
        double s = 0.0;
        int m = 0;
        for (int i = t; i < total; i += 256) {
            const double v = u[i];
            if (v > th) { s += v; m += 1; }
        }
        block_sum_sm(s, m, rs, rm);
        // supports shrink monotonically from a lower bound; a repeat (or a last-bit
        // regrowth) is the fixed point
        if (m == prev || (prev > 0 && m > prev) || m == 0) conv = 1;
        if (m > 0 && m != prev) th = (s - 1.0) / (double)m;
        if (m > 0) prev = m;
    }
}

// PACK = false: one block per column finishes the projection on this rank's candidate
// list.  PACK = true (multi-rank): the block only copies the list, in its fixed order, into
// this rank's slot of the gather buffer ([world][KP][cap + 2]: candidates, element cap = the true
// count, element cap + 1 = the largest candidate left out when the list is longer than the slot);
// k_proj_solve_gathered continues after the all-reduce.
template <bool PACK>
__global__ __launch_bounds__(256) void k_proj_solve(const double *__restrict__ list,
                                                    const int *__restrict__ segcnt, long nseg,
                                                    long segcap, ProjState *__restrict__ ps,
                                                    double *__restrict__ pack_slot, int pack_cap,
                                                    int pack_stride)
{
    __shared__ double u[PROJ_LDS_CAP];
    __shared__ double rs[4];
    __shared__ int rm[4];
    __shared__ int scan[256];
    const int comp = blockIdx.x, t = threadIdx.x;
    const int spt = (int)((nseg + 255) / 256);          // consecutive segments per thread
    const long s0 = (long)t * spt;
    const int *mycnt = segcnt + (long)comp * nseg;
    const double *mylist = list + (long)comp * nseg * segcap;

    // counts of this thread's segments (all loads in flight together)
    int cq[PROJ_SPT];
    int mine = 0;
    const bool blocked = spt <= PROJ_SPT;
    if (blocked) {
#pragma unroll
        for (int q = 0; q < PROJ_SPT; ++q) cq[q] = (q < spt && s0 + q < nseg) ? mycnt[s0 + q] : 0;
#pragma unroll
        for (int q = 0; q < PROJ_SPT; ++q) mine += cq[q];
    } else {
        for (int q = 0; q < spt; ++q)
            if (s0 + q < nseg) mine += mycnt[s0 + q];
    }
    scan[t] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {            // inclusive Hillis-Steele scan
        const int add = t >= off ? scan[t - off] : 0;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    const int total = scan[255];
    const bool in_lds = total <= (PACK ? pack_cap : PROJ_LDS_CAP);
    double *dst = PACK ? pack_slot + (size_t)comp * pack_stride : u;
    if (in_lds) {
        int pos = scan[t] - mine;
        if (blocked) {
            // level i copies the i-th candidate of each of the thread's segments: the
            // PROJ_SPT loads of a level are independent, so one memory latency per level
            // (segments hold a handful of candidates) instead of one per candidate
            int base[PROJ_SPT], maxc = 0;
#pragma unroll
            for (int q = 0; q < PROJ_SPT; ++q) {
                base[q] = pos;
                pos += cq[q];
                maxc = cq[q] > maxc ? cq[q] : maxc;
            }
            for (int i = 0; i < maxc; ++i) {
                double v[PROJ_SPT];
#pragma unroll
                for (int q = 0; q < PROJ_SPT; ++q)
                    v[q] = i < cq[q] ? mylist[(s0 + q) * segcap + i] : 0.0;
#pragma unroll
                for (int q = 0; q < PROJ_SPT; ++q)
                    if (i < cq[q]) dst[base[q] + i] = v[q];
            }
        } else {
            for (int q = 0; q < spt; ++q)
                if (s0 + q < nseg) {
                    const int c = mycnt[s0 + q];
                    const double *src = mylist + (s0 + q) * segcap;
                    for (int i = 0; i < c; ++i) dst[pos++] = src[i];
                }
        }
    }
    if constexpr (PACK) {
        // slot layout: [0, cap) candidates, [cap] = this rank's TRUE candidate count, [cap + 1] = the
        // largest candidate that was NOT sent (-inf: all were sent; +inf: none selected, list too long)
        if (in_lds) {
            if (t == 0) {
                dst[pack_cap] = (double)total;
                dst[pack_cap + 1] = -INFINITY;
            }
            return;
        }
        // More candidates than the slot holds (they are the entries above a LOWER BOUND of the
        // threshold, usually far more than the support): send the pack_cap largest, in their list
        // order, and the largest one left out as a witness.  If the threshold the ranks then find on
        // the union of these lists is >= every rank's witness, nothing left out belongs to the
        // support and the threshold is exact (k_proj_solve_gathered); the device decides, the host
        // does not have to look.  The (cap + 1)-th largest key by radix selection, bit by bit, on the
        // order-preserving integer image of the doubles.
        if (total > 64 * pack_cap) {                     // hopeless (dense start): leave it to the fallback
            if (t == 0) {
                dst[pack_cap] = (double)total;
                dst[pack_cap + 1] = INFINITY;
            }
            return;
        }
        auto key_of = [](double v) -> unsigned long long {
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
        };
        auto for_each_mine = [&](auto &&fn) {
            for (int q = 0; q < spt; ++q)
                if (s0 + q < nseg) {
                    const int cq2 = mycnt[s0 + q];
                    const double *src = mylist + (s0 + q) * segcap;
                    for (int i = 0; i < cq2; ++i) fn(src[i]);
                }
        };
        unsigned long long kw = 0ull;
        for (int bit = 63; bit >= 0; --bit) {
            const unsigned long long trial = kw | (1ull << bit);
            double dummy = 0.0;
            int m = 0;
            for_each_mine([&](double v) { m += key_of(v) >= trial ? 1 : 0; });
            block_sum_sm(dummy, m, rs, rm);
            if (m >= pack_cap + 1) kw = trial;               // uniform: every thread holds the block total
        }
        // kw = key of the (cap + 1)-th largest candidate: everything strictly above it is sent
        int sel = 0;
        for_each_mine([&](double v) { sel += key_of(v) > kw ? 1 : 0; });
        __syncthreads();
        scan[t] = sel;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = t >= off ? scan[t - off] : 0;
            __syncthreads();
            scan[t] += add;
            __syncthreads();
        }
        int pos = scan[t] - sel;
        double witness = -INFINITY;
        for_each_mine([&](double v) {
            if (key_of(v) > kw) dst[pos++] = v;
            else witness = fmax(witness, v);
        });
        double dummy = 0.0;
        int sent = scan[255];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) witness = fmax(witness, __shfl_xor(witness, o, 64));
        if ((t & 63) == 0) rs[t >> 6] = witness;
        __syncthreads();
        if (t == 0) {
            // (zero-fill between the candidates sent and the slot's end: the buffer was zeroed)
            dst[pack_cap] = (double)total;
            dst[pack_cap + 1] = fmax(fmax(rs[0], rs[1]), fmax(rs[2], rs[3]));
            (void)sent;
            (void)dummy;
        }
        return;
    }
    __syncthreads();

    double th = ps->t[comp];
    int prev = -1, conv = 0;
    if (total == 0) {
        conv = 1;                                        // cannot happen (max w > t_lower)
    } else if (in_lds) {
        proj_solve_lds(u, total, th, prev, conv, rs, rm);
    } else {
        for (int it = 0; it < 200 && !conv; ++it) {
            double s = 0.0;
            int m = 0;
            if (blocked) {
                // long lists (cold start / dense columns) stay in global memory: level i
                // reads the i-th candidate of the thread's segments, loads independent
                int maxc = 0;
#pragma unroll
                for (int q = 0; q < PROJ_SPT; ++q) maxc = cq[q] > maxc ? cq[q] : maxc;
                for (int i = 0; i < maxc; ++i) {
                    double v[PROJ_SPT];
#pragma unroll
                    for (int q = 0; q < PROJ_SPT; ++q)
                        v[q] = i < cq[q] ? mylist[(s0 + q) * segcap + i] : -INFINITY;
#pragma unroll
                    for (int q = 0; q < PROJ_SPT; ++q)
                        if (v[q] > th) { s += v[q]; m += 1; }
                }
            } else {
                for (int q = 0; q < spt; ++q)
                    if (s0 + q < nseg) {
                        const int c = mycnt[s0 + q];
                        const double *src = mylist + (s0 + q) * segcap;
                        for (int i = 0; i < c; ++i) {
                            const double v = src[i];
                            if (v > th) { s += v; m += 1; }
                        }
                    }
            }
            block_sum_sm(s, m, rs, rm);
            if (m == prev || (prev > 0 && m > prev) || m == 0) conv = 1;
            if (m > 0 && m != prev) th = (s - 1.0) / (double)m;
            if (m > 0) prev = m;
        }
    }
    if (t == 0) {
        ps->t[comp] = th;
        ps->cnt[comp] = (double)prev;
        ps->shrunk[comp] = conv;                         // list mode: "column converged"
    }
}

// Short columns (n <= 16 384, single rank): the whole threshold search of a column in
// ONE block -- its entries in registers, the lower bound (max - 1, or the Newton step from the
// previous projection's threshold) and the Michelot fixed point of k_proj_solve on them -- instead
// of the first pass, its finalize, the candidate lists and their solver (four launches of ~6 us
// each; three projections per outer iteration are a fifth of the HadISST-shaped problem's time).
// PROJ_SMALL_RPT: entries per thread, 32 (n <= 8192) or 64 (n <= 16 384: the 12 500-row shards of the
// headline problem on 8 GPUs, when run as problems of their own)
template <int PROJ_SMALL_RPT>
__global__ __launch_bounds__(256) void k_proj_small(const double *__restrict__ x,
                                                    const double *__restrict__ g, double a_const,
                                                    const double *__restrict__ scal, int a_slot, long n,
                                                    int KP, int warm_slot, ProjState *__restrict__ ps,
                                                    int kslot = 0)
{
    __shared__ double rs[4];
    __shared__ int rm[4];
    const int comp = blockIdx.x, t = threadIdx.x;
    const double a = load_a(a_const, scal, a_slot, comp, kslot);
    double w[PROJ_SMALL_RPT];
    double mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < PROJ_SMALL_RPT; ++q) {
        const long r = t + 256L * q;
        w[q] = -INFINITY;
        if (r < n) w[q] = g ? x[r * KP + comp] - a * g[r * KP + comp] : x[r * KP + comp];
        mx = fmax(mx, w[q]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) rs[t >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(rs[0], rs[1]), fmax(rs[2], rs[3]));
    __syncthreads();
    // lower bound of t* (POST_FIRST): max - 1, or the Newton step from the warm threshold
    double th = mx - 1.0;
    if (warm_slot > 0) {
        const double tw = ps->warm[warm_slot][comp];
        double s = 0.0;
        int m = 0;
#pragma unroll
        for (int q = 0; q < PROJ_SMALL_RPT; ++q)
            if (w[q] > tw) { s += w[q]; m += 1; }
        block_sum_sm(s, m, rs, rm);
        if (m > 0) {
            const double t1 = (s - 1.0) / (double)m;
            if (t1 > th && t1 < mx) th = t1;
        }
    }
    int prev = -1, conv = 0;
    for (int it = 0; it < 200 && !conv; ++it) {          // the loop of k_proj_solve
        double s = 0.0;
        int m = 0;
#pragma unroll
        for (int q = 0; q < PROJ_SMALL_RPT; ++q)
            if (w[q] > th) { s += w[q]; m += 1; }
        block_sum_sm(s, m, rs, rm);
        if (m == prev || (prev > 0 && m > prev) || m == 0) conv = 1;
        if (m > 0 && m != prev) th = (s - 1.0) / (double)m;
        if (m > 0) prev = m;
    }
    if (t == 0) {
        ps->mx[comp] = mx;
        ps->t[comp] = th;
        ps->cnt[comp] = (double)prev;
        ps->shrunk[comp] = conv;
        if (comp == 0) {
            ps->done = 0;
            ps->passes = -1;                  // list mode: POST_FIN derives `done`
        }
    }
}

// multi-rank: the ranks' candidate lists side by side (gathered [world][KP][cap + 1], summed
// all-reduce of per-rank slots) -> the same fixed point on every rank, lists concatenated in
// rank order.  A rank with more candidates than its slot holds has sent its largest ones and the
// largest it left out (k_proj_solve<true>); if that witness lies above the threshold found here the
// column is left unconverged (sticky flag; the host falls back to the iterative passes from the lower
// bound when it checks at once, or stops with an error at its next poll).
__global__ __launch_bounds__(256) void k_proj_solve_gathered(const double *__restrict__ gathered,
                                                             int world, int cap, int KP,
                                                             ProjState *__restrict__ ps, int mode)
{
    __shared__ double u[PROJ_LDS_CAP];
    __shared__ double rs[4];
    __shared__ int rm[4];
    const int comp = blockIdx.x, t = threadIdx.x;
    const size_t stride = (size_t)cap + 2;
    // per rank: its true candidate count (the first min(count, cap) slots hold what it sent) and the
    // largest candidate it left out
    int total = 0, true_total = 0;
    double witness = -INFINITY;
    for (int r = 0; r < world; ++r) {                    // uniform: every thread reads the counts
        const double *src = gathered + ((size_t)r * KP + comp) * stride;
        const int c = (int)src[cap];
        true_total += c;
        total += c < cap ? c : cap;
        witness = fmax(witness, src[cap + 1]);
    }
    bool overflow = total > PROJ_LDS_CAP || witness == INFINITY;
    double th = ps->t[comp];
    int prev = -1, conv = 0;
    if (!overflow) {
        int pos = 0;
        for (int r = 0; r < world; ++r) {
            const double *src = gathered + ((size_t)r * KP + comp) * stride;
            const int c0 = (int)src[cap];
            const int c = c0 < cap ? c0 : cap;
            for (int i = t; i < c; i += 256) u[pos + i] = src[i];
            pos += c;
        }
        __syncthreads();
        if (total == 0) conv = 1;
        else proj_solve_lds(u, total, th, prev, conv, rs, rm);
        // a candidate that was left out and lies above the threshold belongs to the support: the lists
        // were too short for THIS column (its support, not just its candidate list, outgrew a slot)
        if (conv && witness > th) {
            conv = 0;
            overflow = true;
        }
    }
    if (t == 0) {
        ps->t[comp] = overflow ? ps->t[comp] : th;       // an unconverged column keeps its lower bound for the fallback
        ps->cnt[comp] = (double)prev;
        ps->shrunk[comp] = conv;
        if (!conv) atomicOr(&ps->overflow_sticky, 1);
        atomicMax(&ps->list_max[mode & 3], overflow ? (1 << 30) : true_total);
    }
}

// fallback after an overflowing list: continue with iterative passes from the lower bounds
__global__ void k_proj_fallback_init(ProjState *__restrict__ ps, int k)
{
    const int i = threadIdx.x;
    if (i < k) {
        ps->cnt[i] = 0.0;
        ps->shrunk[i] = 0;
    }
    if (i == 0) {
        ps->done = 0;
        ps->passes = 0;
    }
}

// ---------------------------------------------------------------- finalize
// partial [nb][NV][KP] -> red [NV][KP] (fixed order), then (single rank) the post step
// in the same launch.  One block of 256 threads.
__device__ void post_step(int kind, int mode, const double *__restrict__ red, int KP, int k,
                          ProjState *__restrict__ ps, double *__restrict__ scal, int slot)
{
    const int i = threadIdx.x;
    __shared__ int all_conv;
    if (kind == POST_COLMAX) {
        // slot > 0: warm start from the previous projection of kind `slot` (the Michelot
        // map converges from any threshold with a non-empty support: one step from above
        // lands below t*); otherwise t = max - 1 <= t*.
        if (i < k) {
            const double cold = red[i] - 1.0;
            double t0 = cold;
            if (slot > 0) {
                const double tw = ps->warm[slot][i];
                if (tw < red[i] && tw > cold) t0 = tw;
            }
            ps->mx[i] = red[i];
            ps->t[i] = t0;
            ps->cnt[i] = 0.0;
            ps->shrunk[i] = 0;
        }
        if (i == 0) {
            ps->done = 0;
            ps->passes = 0;
        }
    } else if (kind == POST_FIRST) {
        // red = [max | sum of {w > warm t} | count]: lower bound of t* = the larger of
        // max - 1 and the Newton step from the warm threshold
        if (i < k) {
            const double cold = red[i] - 1.0, cnt = red[2 * KP + i];
            double t0 = cold;
            if (cnt > 0.0) {
                const double t1 = (red[KP + i] - 1.0) / cnt;
                if (t1 > cold && t1 < red[i]) t0 = t1;
            }
            ps->mx[i] = red[i];
            ps->t[i] = t0;
            ps->cnt[i] = 0.0;
            ps->shrunk[i] = 0;
        }
        if (i == 0) {
            ps->done = 0;
            ps->passes = -1;                  // list mode: POST_FIN derives `done`
        }
    } else if (kind == POST_MICHELOT) {
        if (i == 0) all_conv = 1;
        __syncthreads();
        if (i < k) {
            const double s = red[i], cnt = red[KP + i], prev = ps->cnt[i];
            // converged when the support repeats; a support that grows again after it has
            // started to shrink is a last-bit oscillation of the threshold
            const bool conv = (prev > 0.0) && (cnt == prev || (ps->shrunk[i] && cnt > prev));
            if (prev > 0.0 && cnt < prev) ps->shrunk[i] = 1;
            if (cnt > 0.0) ps->t[i] = (s - 1.0) / cnt;
            else ps->t[i] = ps->mx[i] - 1.0;           // empty support: restart from below
            ps->cnt[i] = cnt;
            if (!conv) atomicAnd(&all_conv, 0);
        }
        __syncthreads();
        if (i == 0) {
            ps->passes += 1;
            if (all_conv) ps->done = 1;
        }
    } else if (kind == POST_FIN) {
        if (i < k && mode > 0 && mode < 4) ps->warm[mode][i] = ps->t[i];
        if (i < 64) {                       // wave 0: the k column results -> scalars (xor tree)
            const bool col = i < k;
            double s0 = col ? red[i] : 0.0, s1 = col ? red[KP + i] : 0.0;
            double s2 = col ? red[2 * KP + i] : 0.0, m3 = col ? red[3 * KP + i] : 0.0;
            const bool conv_all = __ballot(col && !ps->shrunk[i]) == 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s0 += __shfl_xor(s0, o, 64);
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
                m3 = fmax(m3, __shfl_xor(m3, o, 64));
            }
            if (i == 0) {
                if (ps->passes < 0) ps->done = conv_all ? 1 : 0;
                if (mode == PROJ_DIR) {
                    scal[SC_DELTA] = s0;
                    scal[SC_DD] = s1;
                    scal[SC_S1D] = s2;
                } else if (mode == PROJ_RES) {
                    scal[SC_RES2] = s0;
                    scal[SC_RESINF] = m3;
                } else if (mode == PROJ_ALPHA) {
                    scal[SC_AINV] = m3;
                }
                const bool done = ps->passes < 0 ? conv_all : (ps->done != 0);
                if (!done)
                    scal[SC_FLAGS] = (double)((int)scal[SC_FLAGS] | AA_SPG_FLAG_PROJ_UNCONV);
            }
        }
    } else if (kind == POST_SCALAR_SUM) {
        if (i < 64) {
            double s = i < k ? red[i] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (i == 0) scal[slot] = s;
        }
    }
}

// Restarts side by side: the scalar part of POST_FIN / POST_SCALAR_SUM once per slot -- wave w takes
// the slots w, w + 16, ...; the slot's kslot column results sit in lanes 0 .. kslot - 1 of the xor
// tree exactly as a single fit's k columns do (same association, same bits) -- and the scalar
// stage that follows, once per slot on the slot's scalar block.
__device__ void post_step_slots(int kind, int mode, const double *__restrict__ red, int KP, int kslot, int R,
                                ProjState *__restrict__ ps, double *__restrict__ scal, int slot,
                                int stage_after, const aa_spg_params &sp)
{
    const int i = threadIdx.x, lane = i & 63, nw = (int)blockDim.x >> 6;
    if (kind == POST_FIN && i < kslot * R && mode > 0 && mode < 4) ps->warm[mode][i] = ps->t[i];
    for (int r = i >> 6; r < R; r += nw) {
        double *sc = scal + (size_t)r * AA_SC_STRIDE;
        const bool col = lane < kslot;
        const int idx = r * kslot + lane;
        if (kind == POST_FIN) {
            double s0 = col ? red[idx] : 0.0, s1 = col ? red[KP + idx] : 0.0;
            double s2 = col ? red[2 * KP + idx] : 0.0, m3 = col ? red[3 * KP + idx] : 0.0;
            const bool conv_all = __ballot(col && !ps->shrunk[col ? idx : 0]) == 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s0 += __shfl_xor(s0, o, 64);
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
                m3 = fmax(m3, __shfl_xor(m3, o, 64));
            }
            if (lane == 0) {
                if (mode == PROJ_DIR) {
                    sc[SC_DELTA] = s0;
                    sc[SC_DD] = s1;
                    sc[SC_S1D] = s2;
                } else if (mode == PROJ_RES) {
                    sc[SC_RES2] = s0;
                    sc[SC_RESINF] = m3;
                } else if (mode == PROJ_ALPHA) {
                    sc[SC_AINV] = m3;
                }
                if (!conv_all) sc[SC_FLAGS] = (double)((int)sc[SC_FLAGS] | AA_SPG_FLAG_PROJ_UNCONV);
            }
        } else if (kind == POST_SCALAR_SUM) {
            double s = col ? red[idx] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) sc[slot] = s;
        }
        if (stage_after >= 0 && lane == 0) scalar_stage_simple(stage_after, sc, sp);
    }
    if (kind == POST_FIN && i == 0 && ps->passes < 0) ps->done = 1;
}

// partial [nb][NV][KP] -> red [NV][KP], fixed order; all FIN_NT threads of one block; sm holds
// 4 * FIN_NT doubles.  `gather`: multi-rank slot buffer (see k_finalize_sum).
template <bool AGENT>     // AGENT: the partials were written by blocks of THIS launch (FinTail): agent-scope loads
__device__ __forceinline__ void finalize_sum_block(const double *partial, int nb, int NV,
                                                   int KP, unsigned max_mask, double *__restrict__ red,
                                                   double *sm, double *__restrict__ gather, int rank,
                                                   int world)
{
    const int RS = FIN_NT / KP;                 // 32 or 16 partial groups, all NV values at once
    const int t = threadIdx.x, comp = t % KP, part = t / KP;
    double acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = ((max_mask >> a) & 1u) ? -INFINITY : 0.0;
    // batches of 8 partial rows per thread with all their loads issued before the first use
    // (a rolled loop waits one memory latency per row)
    for (int b0 = part; b0 < nb; b0 += 8 * RS) {
        double val[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u * RS;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bool is_max = (max_mask >> a) & 1u;
                if (a < NV && b < nb)
                    val[u][a] = AGENT ? load_agent(&partial[((size_t)b * NV + a) * KP + comp])
                                      : partial[((size_t)b * NV + a) * KP + comp];
                else
                    val[u][a] = is_max ? -INFINITY : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                acc[a] = ((max_mask >> a) & 1u) ? fmax(acc[a], val[u][a]) : acc[a] + val[u][a];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) sm[a * FIN_NT + t] = acc[a];
    __syncthreads();
    if (part < NV) {                            // group a finishes value a (fixed order)
        const int a = part;
        const bool is_max = (max_mask >> a) & 1u;
        double s = sm[a * FIN_NT + comp];
        for (int q = 1; q < RS; ++q) {
            const double val = sm[a * FIN_NT + q * KP + comp];
            s = is_max ? fmax(s, val) : s + val;
        }
        red[a * KP + comp] = s;
        // multi-rank: this rank's slot of the gather buffer, zeros in the other slots; a sum
        // all-reduce then leaves every rank's values side by side on every rank
        if (gather)
            for (int r = 0; r < world; ++r) gather[((size_t)r * NV + a) * KP + comp] = r == rank ? s : 0.0;
    }
    __syncthreads();
}

__global__ __launch_bounds__(FIN_NT) void k_finalize_sum(const double *__restrict__ partial, int nb,
                                                         int NV, int KP, unsigned max_mask,
                                                         double *__restrict__ red,
                                                         const ProjState *__restrict__ ps_gate,
                                                         int kind, int mode, int k,
                                                         ProjState *__restrict__ ps,
                                                         double *__restrict__ scal, int slot,
                                                         double *__restrict__ gather, int rank,
                                                         int world, int stage_after, aa_spg_params sp,
                                                         int kslot = 0, int R = 0)
{
    if (ps_gate && ps_gate->done) return;
    __shared__ double sm[4 * FIN_NT];
    finalize_sum_block(partial, nb, NV, KP, max_mask, red, sm, gather, rank, world);
    if (kslot > 0 && (kind == POST_FIN || kind == POST_SCALAR_SUM)) {      // restarts side by side
        post_step_slots(kind, mode, red, KP, kslot, R, ps, scal, slot, stage_after, sp);
        return;
    }
    if (kind != POST_NONE) post_step(kind, mode, red, KP, k, ps, scal, slot);
    // the scalar stage that consumes these reductions (ST_ALPHA / ST_BB / ST_CONV) rides along
    // instead of being its own ~5 us launch
    if (stage_after >= 0) {
        __syncthreads();
        if (threadIdx.x == 0) scalar_stage_simple(stage_after, scal, sp);
    }
}

// red [NV][KP] -> projection state / scalars, as its own launch (multi-rank: runs after
// the all-reduce of `red`).  One block of 256 threads.
__global__ __launch_bounds__(256) void k_post(int kind, int mode, double *__restrict__ red,
                                              int KP, int k, ProjState *__restrict__ ps,
                                              double *__restrict__ scal, int slot, int gated,
                                              const double *__restrict__ gather, int world, int NV,
                                              unsigned max_mask, int stage_after, aa_spg_params sp)
{
    if (gated && ps->done) return;
    // gathered [world][NV][KP] -> red [NV][KP], ranks combined in rank order on every rank
    // (identical bits everywhere, whatever the all-reduce algorithm)
    const int t = threadIdx.x;
    if (t < NV * KP) {
        const int a = t / KP;
        const bool is_max = (max_mask >> a) & 1u;
        double s = gather[t];
        for (int r = 1; r < world; ++r) {
            const double v = gather[(size_t)r * NV * KP + t];
            s = is_max ? fmax(s, v) : s + v;
        }
        red[t] = s;
    }
    __syncthreads();
    post_step(kind, mode, red, KP, k, ps, scal, slot);
    if (stage_after >= 0) {
        __syncthreads();
        if (threadIdx.x == 0) scalar_stage_simple(stage_after, scal, sp);
    }
}

// ---------------------------------------------------------------- dictionary set-up (one block)
// Everything a dictionary update needs from the Gram state (after the previous weights
// update leaves: Z'Z, C K C', C K Z in the Gram state): M = D Z'Z D
// (archetypal_analysis.py:310,330), gram[0] = C K C', the scalars, tr(C H D) = sum_i
// alpha_i (C K Z)_ii, and f(x) (spg.py:153-157) -- one block instead of seven launches.  Its own
// launch (k_dict_setup) or block 0 of the update's first gradient launch (k_grad, DictSetup).
__device__ double block_trace_MG(const double *__restrict__ M, const double *__restrict__ G, int k,
                                 int KP, bool transposed, double *sm);
struct DictSetup {
    int on;
    const double *state;     // ZtZ | CKCt | CKZ
    double trace, fnorm;
    double *Mout, *gram, *sc;
    aa_spg_params sp;
};
__device__ __forceinline__ void dict_setup_body(const double *__restrict__ state, const double *__restrict__ alpha,
                                                int k, int KP, double trace, double fnorm, double *__restrict__ Mout,
                                                double *__restrict__ gram, double *__restrict__ sc,
                                                const aa_spg_params &sp, double *sm /* 256 */)
{
    const int GS = KP * KP, t = threadIdx.x;
    const double *ZtZ = state, *CKCt = state + GS, *CKZ = state + 2 * GS;
    for (int e = t; e < GS; e += 256) {
        const int i = e / KP, j = e % KP;
        Mout[e] = (i < k && j < k) ? alpha[i] * ZtZ[e] * alpha[j] : 0.0;
        gram[e] = CKCt[e];
    }
    __syncthreads();
    const double a0 = block_trace_MG(Mout, gram, k, KP, false, sm);
    sm[t] = t < k ? alpha[t] * CKZ[t * KP + t] : 0.0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) sm[t] += sm[t + o];
        __syncthreads();
    }
    if (t == 0) {
        sc[SC_TRACE] = trace;
        sc[SC_FNORM] = fnorm;
        sc[SC_S1] = sm[0];
        sc[SC_A0] = a0;
        sc[SC_F_OLD] = 0.5 * (trace - 2.0 * sm[0] + a0) / fnorm;
        sc[SC_NFEVAL] = 1.0;
        sc[SC_FLAGS] = 0.0;
        for (int i = 0; i < 16; ++i) sc[SC_FMEM0 + i] = 0.0;   // f_mem = zeros (spg.py:153)
        sc[SC_ALPHA_SET] = (sp.alpha0 >= 0.0) ? 1.0 : 0.0;
        sc[SC_ALPHA] = sp.alpha0;
    }
}

// ---------------------------------------------------------------- gradient
// g[r][i] = (sum_j M[i][j] * Graw[r][j] - H[r][i] * alpha[i]) * scale
//   data form:   Graw = (C X X')', H = XX'Z, scale = 1/n   (archetypal_analysis.py:293-301)
//   kernel form: Graw = (C K)',    H = K Z,  scale = 1/k   (archetypal_analysis.py:284-290)
// optional: v0 = <d, g>.
// The k x k product runs on v_mfma_f64_16x16x4_f64: a wave owns 16 rows; A-operand lane l
// = Graw[r0 + (l&15)][4s + (l>>4)], B-operand = M[16*ti + (l&15)][4s + (l>>4)] (held in
// registers for the whole kernel); D[row = (l>>4) + 4*reg][col = l&15] is written as four
// 128-byte row segments per register.
template <int KP>
__global__ __launch_bounds__(256) void k_grad(const double *__restrict__ Graw,
                                              const double *__restrict__ H,
                                              const double *__restrict__ M,
                                              const double *__restrict__ alpha, double scale,
                                              long n, long rows_pb, int k,
                                              double *__restrict__ gout,
                                              const double *__restrict__ d,
                                              double *__restrict__ partial,
                                              double *__restrict__ xupd,
                                              const double *__restrict__ scalw, int kslot, int nslot,
                                              DictSetup ds)
{
    constexpr int T = KP / 16;       // component tiles
    constexpr int S = KP / 4;        // contraction steps
    __shared__ double sm[256];
    // ds.on: the update's set-up (M, the scalars, f(x)) is block 0 of this launch, beside the row blocks,
    // which form their operand tiles of M = D Z'Z D from the Gram state themselves (the same products)
    if (ds.on && blockIdx.x == 0) {
        dict_setup_body(ds.state, alpha, k, KP, ds.trace, ds.fnorm, ds.Mout, ds.gram, ds.sc, ds.sp, sm);
        return;
    }
    const long bx = (long)blockIdx.x - (ds.on ? 1 : 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane >> 4, lc = lane & 15;
    double mreg[T][S];
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = 16 * ti + lc, j = 4 * s + lr;
            if (ds.on) mreg[ti][s] = (i < k && j < k) ? alpha[i] * ds.state[i * KP + j] * alpha[j] : 0.0;
            else mreg[ti][s] = M[i * KP + j];
        }
    double al[T];
#pragma unroll
    for (int ti = 0; ti < T; ++ti) al[ti] = alpha[16 * ti + lc];
    const long rb = bx * rows_pb;      // rows_pb is a multiple of 64
    long re = rb + rows_pb;
    double dot = 0.0;
    // xupd: x <- x + lambda d for the rows of this block (the accepted SPG step, spg.py:208-222;
    // d is read for <d, g_new> anyway)
    // (restarts side by side: the step of the slot a column belongs to)
    double lamv[T];
#pragma unroll
    for (int ti = 0; ti < T; ++ti) {
        const int sl = kslot > 0 ? (16 * ti + lc) / kslot : 0;
        lamv[ti] = xupd ? scalw[(size_t)(sl < nslot ? sl : 0) * AA_SC_STRIDE + SC_LAMBDA] : 0.0;
    }
    // Two 16-row tiles (r0 and r0 + 64) per step, every load of both -- the A operands, and H,
    // d, x of the epilogue -- issued before the first MFMA: with one tile in flight a wave waited
    // out a full memory latency three times per 16 rows (35 us per launch).
    for (long r0 = rb + 16 * wave; r0 < re; r0 += 128) {
        if (r0 >= n) break;                           // wave-uniform
        const bool two = (r0 + 64 < re) && (r0 + 64 < n);   // wave-uniform
        double gv[2][S], hv[2][T][4], dv[2][T][4], xv[2][T][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const long ru = r0 + 64 * u;
#pragma unroll
            for (int s = 0; s < S; ++s) gv[u][s] = Graw[(ru + lc) * KP + 4 * s + lr];   // rows < n_pad
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long e = (ru + lr + 4 * reg) * KP + 16 * ti + lc;   // < (n_pad + slack) * KP
                    hv[u][ti][reg] = H[e];
                    dv[u][ti][reg] = d ? d[e] : 0.0;
                    xv[u][ti][reg] = xupd ? xupd[e] : 0.0;
                }
        }
        f64x4 acc[2][T];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ti = 0; ti < T; ++ti) acc[u][ti] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int ti = 0; ti < T; ++ti)
                    acc[u][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[u][s], mreg[ti][s], acc[u][ti], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const long ru = r0 + 64 * u;
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long r = ru + lr + 4 * reg;
                    const int comp = 16 * ti + lc;
                    if (r < n && comp < k) {
                        const long e = r * KP + comp;
                        const double ge = (acc[u][ti][reg] - hv[u][ti][reg] * al[ti]) * scale;
                        gout[e] = ge;
                        if (d) {
                            const double de = dv[u][ti][reg];
                            dot += de * ge;
                            if (xupd) xupd[e] = xv[u][ti][reg] + lamv[ti] * de;
                        }
                    }
                }
        }
    }
    if (partial) {
        // fixed-order block sum of the per-thread dot products -> one value per block,
        // stored in component slot 0 (the finalize step adds the k slots)
        sm[threadIdx.x] = dot;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x < KP) partial[(size_t)bx * KP + threadIdx.x] = threadIdx.x == 0 ? sm[0] : 0.0;
    }
}

// restarts side by side: <d, g_new> of slot blockIdx.y, summed in EXACTLY the order k_grad sums it for
// a single fit of k components (same grid, same thread -> element map with the slot's columns in the
// place of columns 0 .. k-1, same tree): the BB step of every slot gets the bits it gets alone.
// partial [blocks][KP]: the slot's dot in component slot `r k`, zeros in the rest of its columns.
template <int KP>
__global__ __launch_bounds__(256) void k_grad_dot_slots(const double *__restrict__ g, const double *__restrict__ d,
                                                        long n, long rows_pb, int k, int R,
                                                        double *__restrict__ partial)
{
    constexpr int T = KP / 16;
    __shared__ double sm[256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane >> 4, lc = lane & 15;
    const int r = blockIdx.y, o = r * k;
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    double dot = 0.0;
    for (long r0 = rb + 16 * wave; r0 < re; r0 += 128) {
        if (r0 >= n) break;
        const bool two = (r0 + 64 < re) && (r0 + 64 < n);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const long ru = r0 + 64 * u;
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long row = ru + lr + 4 * reg;
                    const int comp = 16 * ti + lc;
                    if (row < n && comp < k) {
                        const long e = row * KP + o + comp;
                        dot += d[e] * g[e];
                    }
                }
        }
    }
    sm[threadIdx.x] = dot;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if ((int)threadIdx.x < q) sm[threadIdx.x] += sm[threadIdx.x + q];
        __syncthreads();
    }
    double *dst = partial + (size_t)blockIdx.x * KP;
    if ((int)threadIdx.x < k) dst[o + threadIdx.x] = threadIdx.x == 0 ? sm[0] : 0.0;
    if (r == 0)
        for (int i = R * k + threadIdx.x; i < KP; i += 256) dst[i] = 0.0;
}

// v0 = sum x * H * alpha   (tr(C * H D), archetypal_analysis.py:267,279)
template <int KP>
__global__ __launch_bounds__(256) void k_tall_dot_scaled(const double *__restrict__ x,
                                                         const double *__restrict__ H,
                                                         const double *__restrict__ alpha, long n,
                                                         long rows_pb, int k,
                                                         double *__restrict__ partial)
{
    __shared__ double sm[256];
    constexpr int RS = 256 / KP;
    const int t = threadIdx.x, comp = t % KP, rsub = t / KP;
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    if (re > n) re = n;
    double v[1] = {0.0};
    if (comp < k) {
        const double al = alpha ? alpha[comp] : 1.0;
        for (long r = rb + rsub; r < re; r += RS) v[0] += x[r * KP + comp] * H[r * KP + comp] * al;
    }
    block_col_combine<KP, 1>(v, 0u, sm, partial + (size_t)blockIdx.x * KP);
}

// x += lambda * d   (rows < n only; padding stays zero)
__global__ __launch_bounds__(256) void k_tall_axpy(double *__restrict__ x,
                                                   const double *__restrict__ d,
                                                   const double *__restrict__ scal, long elems)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < elems) x[i] = x[i] + scal[SC_LAMBDA] * d[i];
}

template <typename T>
__global__ __launch_bounds__(256) void k_wide_axpy(double *__restrict__ P,
                                                   const double *__restrict__ Q,
                                                   const double *__restrict__ scal, long elems,
                                                   T *__restrict__ PT, long ld = 0, int kslot = 0, int R = 0)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < elems) {
        if (kslot > 0) {                     // restarts side by side: row i / ld belongs to slot row / kslot
            const int r = (int)((i / ld) / kslot);
            scal += (size_t)(r < R ? r : 0) * AA_SC_STRIDE;
        }
        const double v = P[i] + scal[SC_LAMBDA] * Q[i];
        P[i] = v;
        if (PT) PT[i] = (T)v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_wide_to_T(const double *__restrict__ src, long elems,
                                                   T *__restrict__ dst)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < elems) dst[i] = (T)src[i];
}

// ---------------------------------------------------------------- Gram products
// tall:  out[i][j] = sum_r A[r][i] * B[r][j]      (Z'Z, C (XX'Z), (CK) C')
// v_mfma_f64_16x16x4_f64 with the contraction over 4 rows per instruction: lane l
// supplies A[r + (l>>4)][16*ti + (l&15)] and B[r + (l>>4)][16*tj + (l&15)] (four whole
// 128-byte row segments per load); D[row = (l>>4) + 4*reg][col = l&15].  The 4 waves of
// a block take interleaved 4-row groups and are combined through LDS in a fixed order.
// Rows in [n, n_pad) are zero in every tall array, so no row guard is needed.
template <int KP>
__global__ __launch_bounds__(256) void k_gram_tall(const double *__restrict__ A,
                                                   const double *__restrict__ B, long n_pad,
                                                   long rows_pb, double *__restrict__ partial)
{
    constexpr int T = KP / 16;
    __shared__ double comb[3][KP * KP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane >> 4, lc = lane & 15;
    const long rb = (long)blockIdx.x * rows_pb;
    long re = rb + rows_pb;
    if (re > n_pad) re = n_pad;
    f64x4 acc[T][T];
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = 0; tj < T; ++tj) acc[ti][tj] = (f64x4){0.0, 0.0, 0.0, 0.0};
    // 4 row groups (64 block-rows) per step: 8*T loads in flight before the first MFMA
    constexpr int UG = 4;
    for (long r = rb + 4 * wave; r < re; r += 16 * UG) {
        double av[UG][T], bv[UG][T];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const long rr = r + 16 * u + lr;           // < n_pad + AA_SLACK_ROWS (zero rows)
#pragma unroll
            for (int ti = 0; ti < T; ++ti) {
                av[u][ti] = A[rr * KP + 16 * ti + lc];
                bv[u][ti] = B[rr * KP + 16 * ti + lc];
            }
        }
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int tj = 0; tj < T; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][ti], bv[u][tj], acc[ti][tj], 0, 0, 0);
    }
    // combine the 4 waves: waves 1..3 park their tiles in LDS, wave 0 adds them in order
    if (wave > 0) {
#pragma unroll
        for (int ti = 0; ti < T; ++ti)
#pragma unroll
            for (int tj = 0; tj < T; ++tj)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    comb[wave - 1][(16 * ti + lr + 4 * reg) * KP + 16 * tj + lc] = acc[ti][tj][reg];
    }
    __syncthreads();
    if (wave == 0) {
        double *dst = partial + (size_t)blockIdx.x * KP * KP;
#pragma unroll
        for (int ti = 0; ti < T; ++ti)
#pragma unroll
            for (int tj = 0; tj < T; ++tj)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int e = (16 * ti + lr + 4 * reg) * KP + 16 * tj + lc;
                    dst[e] = ((acc[ti][tj][reg] + comb[0][e]) + comb[1][e]) + comb[2][e];
                }
    }
}

// wide:  out[i][j] = sum_c A[i][c] * B[j][c]      ((CX)(CX)', (CX)(DX)', (DX)(DX)')
template <int KP>
__global__ __launch_bounds__(256) void k_gram_wide(const double *__restrict__ A,
                                                   const double *__restrict__ B, int ld,
                                                   double *__restrict__ partial)
{
    constexpr int JT = KP * KP / 256;
    constexpr int TPI = KP / JT;
    constexpr int CW = 32;
    __shared__ double As[KP][CW + 1], Bs[KP][CW + 1];
    const int t = threadIdx.x, i = t / TPI, j0 = (t % TPI) * JT;
    const int c0 = blockIdx.x * 128;
    double acc[JT];
#pragma unroll
    for (int q = 0; q < JT; ++q) acc[q] = 0.0;
    for (int cc = 0; cc < 128; cc += CW) {
        for (int e = t; e < KP * CW; e += 256) {
            As[e / CW][e % CW] = A[(long)(e / CW) * ld + c0 + cc + e % CW];
            Bs[e / CW][e % CW] = B[(long)(e / CW) * ld + c0 + cc + e % CW];
        }
        __syncthreads();
        for (int c = 0; c < CW; ++c) {
            const double a = As[i][c];
#pragma unroll
            for (int q = 0; q < JT; ++q) acc[q] = fma(a, Bs[j0 + q][c], acc[q]);
        }
        __syncthreads();
    }
    double *dst = partial + (size_t)blockIdx.x * KP * KP;
#pragma unroll
    for (int q = 0; q < JT; ++q) dst[i * KP + j0 + q] = acc[q];
}

__global__ __launch_bounds__(256) void k_gram_finalize(const double *__restrict__ partial, int nb,
                                                       int elems, double *__restrict__ out)
{
    // block = 64 elements x 4 partial groups (group g sums b = g, g+4, ...), combined in a
    // fixed order
    __shared__ double sm[256];
    const int t = threadIdx.x, g = t >> 6;
    const int e = blockIdx.x * 64 + (t & 63);
    double s = 0.0;
    if (e < elems) {
#pragma unroll 8
        for (int b = g; b < nb; b += 4) s += partial[(size_t)b * elems + e];
    }
    sm[t] = s;
    __syncthreads();
    if (g == 0 && e < elems) out[e] = ((sm[t] + sm[t + 64]) + sm[t + 128]) + sm[t + 192];
}

// ---------------------------------------------------------------- transposes (kernel form)
// wide [KP][ld] -> tall [n_pad][KP]
__global__ __launch_bounds__(256) void k_wide_to_tall(const double *__restrict__ wide, int ld,
                                                      int KP, long n_pad, double *__restrict__ tall)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const long r0 = (long)blockIdx.x * 32;
    const int i0 = blockIdx.y * 32;
    for (int q = ty; q < 32; q += 8) tile[q][tx] = wide[(long)(i0 + q) * ld + r0 + tx];
    __syncthreads();
    for (int q = ty; q < 32; q += 8) tall[(r0 + q) * KP + i0 + tx] = tile[tx][q];
}

template <typename T>
__global__ __launch_bounds__(256) void k_tall_to_wide(const double *__restrict__ tall, int KP,
                                                      int ld, double *__restrict__ wide,
                                                      T *__restrict__ wideT)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long r0 = (long)blockIdx.x * 32;
    const int i0 = blockIdx.y * 32;
    for (int q = ty; q < 32; q += 8) tile[q][tx] = tall[(r0 + q) * KP + i0 + tx];
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const double v = tile[tx][q];
        if (wide) wide[(long)(i0 + q) * ld + r0 + tx] = v;
        if (wideT) wideT[(long)(i0 + q) * ld + r0 + tx] = (T)v;
    }
}

// ---------------------------------------------------------------- scalar stages
__device__ double dev_line_search_step(double lam, double delta, double f_old, double f_new,
                                       double s1, double s2)
{   // spg.py:19-33
    const double tmp = -0.5 * lam * lam * delta / (f_new - f_old - lam * delta);
    if (s1 <= tmp && tmp <= s2 * lam) return tmp;
    return 0.5 * lam;
}

// the same for a block of more than 256 threads: only the first 256 accumulate, all join the
// barriers
__device__ double block_trace_MG_n(const double *__restrict__ M, const double *__restrict__ G, int k,
                                   int KP, bool transposed, double *sm);

// tr(M * G) (or tr(M * G') when transposed), all 256 threads, fixed-order tree.
__device__ double block_trace_MG(const double *__restrict__ M, const double *__restrict__ G, int k,
                                 int KP, bool transposed, double *sm)
{
    double s = 0.0;
    for (int e = threadIdx.x; e < k * k; e += 256) {
        const int i = e / k, j = e % k;
        s += M[i * KP + j] * (transposed ? G[i * KP + j] : G[j * KP + i]);
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// spg.py:196-229 on one thread: f along x + lambda d in closed form,
//   f(x + lam d) = 0.5 (tr - 2 (s1 + lam s1d) + a0 + lam a1 + lam^2 a2) / fnorm,
// with a1 = tr(M (PQ' + QP')), a2 = tr(M QQ'): the Armijo loop with safeguarded quadratic
// interpolation needs no further pass over the data.
__device__ void linesearch_thread0(double *__restrict__ sc, const aa_spg_params &sp, double tr1, double tr2)
{
    const double a1 = tr1, a2 = tr2;
    const double tr = sc[SC_TRACE], s1 = sc[SC_S1], s1d = sc[SC_S1D], a0 = sc[SC_A0];
    const double fn = sc[SC_FNORM], f_old = sc[SC_F_OLD], delta = sc[SC_DELTA];
    int mem = sp.memory < 1 ? 1 : (sp.memory > 16 ? 16 : sp.memory);
    for (int i = mem - 1; i > 0; --i) sc[SC_FMEM0 + i] = sc[SC_FMEM0 + i - 1];
    sc[SC_FMEM0] = f_old;
    double f_max = sc[SC_FMEM0];
    for (int i = 1; i < mem; ++i) f_max = (sc[SC_FMEM0 + i] >= f_max) ? sc[SC_FMEM0 + i] : f_max;
    double lam = 1.0;
    double nfe = sc[SC_NFEVAL];
    int flags = (int)sc[SC_FLAGS];
    double f_new = 0.5 * (tr - 2.0 * (s1 + lam * s1d) + a0 + lam * a1 + lam * lam * a2) / fn;
    nfe += 1.0;
    int guard = 0;
    while (f_new > f_max + sp.gamma * lam * delta && guard < 200) {
        lam = dev_line_search_step(lam, delta, f_old, f_new, sp.sigma_one, sp.sigma_two);
        f_new = 0.5 * (tr - 2.0 * (s1 + lam * s1d) + a0 + lam * a1 + lam * lam * a2) / fn;
        nfe += 1.0;
        ++guard;
        if (fabs(lam) < sp.lambda_min) {
            flags |= AA_SPG_FLAG_LAMBDA_MIN;
            break;
        }
    }
    sc[SC_LAMBDA] = lam;
    sc[SC_F_NEW] = f_new;
    sc[SC_S1] = s1 + lam * s1d;
    sc[SC_A0] = a0 + lam * a1 + lam * lam * a2;
    sc[SC_A1] = a1;
    sc[SC_A2] = a2;
    sc[SC_NFEVAL] = nfe;
    sc[SC_FLAGS] = (double)flags;
}

__device__ double block_trace_MG_n(const double *__restrict__ M, const double *__restrict__ G, int k,
                                   int KP, bool transposed, double *sm)
{
    const int t = threadIdx.x;
    double s = 0.0;
    if (t < 256)
        for (int e = t; e < k * k; e += 256) {
            const int i = e / k, j = e % k;
            s += M[i * KP + j] * (transposed ? G[i * KP + j] : G[j * KP + i]);
        }
    if (t < 256) sm[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) sm[t] += sm[t + o];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// gram: [0] = C K C' (or (CX)(CX)'), [1] = cross1, [2] = D K D', [3] = cross2
__global__ __launch_bounds__(256) void k_scalar_stage(int stage, double *__restrict__ sc,
                                                      const double *__restrict__ gram,
                                                      const double *__restrict__ M, int k, int KP,
                                                      aa_spg_params sp, int cross2_is_transpose)
{
    __shared__ double sm[256];
    const int GS = KP * KP;
    if (gridDim.x > 1) {                 // restarts side by side: block = slot, its diagonal blocks and scalars
        const size_t off = (size_t)(blockIdx.x * k) * KP + blockIdx.x * k;
        sc += (size_t)blockIdx.x * AA_SC_STRIDE;
        gram += off;
        M += off;
    }
    double tr0 = 0.0, tr1 = 0.0, tr2 = 0.0;
    if (stage == ST_INIT_F) {
        tr0 = block_trace_MG(M, gram, k, KP, false, sm);
    } else if (stage == ST_LINESEARCH) {
        tr1 = block_trace_MG(M, gram + GS, k, KP, false, sm);
        if (cross2_is_transpose) tr1 += block_trace_MG(M, gram + GS, k, KP, true, sm);
        else tr1 += block_trace_MG(M, gram + 3 * GS, k, KP, false, sm);
        tr2 = block_trace_MG(M, gram + 2 * GS, k, KP, false, sm);
    }
    if (threadIdx.x != 0) return;
    if (stage == ST_INIT_F) {            // spg.py:153-157
        const double a0 = tr0;
        sc[SC_A0] = a0;
        sc[SC_F_OLD] = 0.5 * (sc[SC_TRACE] - 2.0 * sc[SC_S1] + a0) / sc[SC_FNORM];
        sc[SC_NFEVAL] = 1.0;
        sc[SC_FLAGS] = 0.0;
        for (int i = 0; i < 16; ++i) sc[SC_FMEM0 + i] = 0.0;   // f_mem = zeros (spg.py:153)
        sc[SC_ALPHA_SET] = (sp.alpha0 >= 0.0) ? 1.0 : 0.0;
        sc[SC_ALPHA] = sp.alpha0;
    } else if (stage == ST_ALPHA || stage == ST_BB || stage == ST_CONV) {
        scalar_stage_simple(stage, sc, sp);
    } else if (stage == ST_LINESEARCH) {
        linesearch_thread0(sc, sp, tr1, tr2);
    }
}

// the scalar stages that need no Gram trace, one thread (also called from the last block of
// the kernel that produces their inputs)
__device__ void scalar_stage_simple(int stage, double *__restrict__ sc, const aa_spg_params &sp)
{
    if (stage == ST_ALPHA) {             // spg.py:178-189
        if (sc[SC_ALPHA_SET] == 0.0) {
            const double ainv = sc[SC_AINV];
            sc[SC_ALPHA] = (fabs(ainv) > 1e-12) ? 1.0 / ainv : 1.0;
            sc[SC_ALPHA_SET] = 1.0;
        }
    } else if (stage == ST_BB) {         // spg.py:232-244
        const double lam = sc[SC_LAMBDA];
        const double sksk = lam * lam * sc[SC_DD];
        const double beta = lam * (sc[SC_DGN] - sc[SC_DELTA]);
        double al;
        if (beta <= 0.0) al = sp.alpha_max;
        else al = fmin(sp.alpha_max, fmax(sp.alpha_min, sksk / beta));
        sc[SC_ALPHA] = al;
        sc[SC_F_OLD] = sc[SC_F_NEW];
        sc[SC_NFEVAL] += 1.0;
    } else if (stage == ST_CONV) {       // spg.py:246-276
        const double rn = sqrt(sc[SC_RES2]);
        sc[SC_RES2] = rn * rn;
        bool conv = rn < sp.epsilon_two;
        if (sp.use_infinity_norm) conv = conv || (sc[SC_RESINF] < sp.epsilon_one);
        int flags = (int)sc[SC_FLAGS];
        if (conv) flags |= AA_SPG_FLAG_CONVERGED;
        else if (sc[SC_NFEVAL] > (double)sp.max_feval) flags |= AA_SPG_FLAG_MAX_FEVAL;
        sc[SC_FLAGS] = (double)flags;
    }
}

// ---------------------------------------------------------------- fused small stages
// (P Q') and (Q Q') of the line search in ONE pass over the two wide arrays (per-block
// partials), and ONE block behind it for everything that used to be five more launches: the
// fixed-order sum of the partials, the line search itself (linesearch_thread0), the Gram of
// the accepted point
//     (P + lam Q)(P + lam Q)' = PP' + lam (PQ' + QP') + lam^2 QQ'
// written to the Gram state (archetypal_analysis.py:620: no further pass over P), and the
// cost after the dictionary update from the line search's own scalars
//     0.5 (tr - 2 tr(C H D) + tr(M CXX'C')) / n        (archetypal_analysis.py:623-627).
// gram: [0] = P P' (in), [1] = P Q', [2] = Q Q' (out).
template <int KP>
__global__ __launch_bounds__(256) void k_gram_wide_pq(const double *__restrict__ P,
                                                      const double *__restrict__ Q, int ld,
                                                      double *__restrict__ partial, int cols_per_block,
                                                      int k)
{
    constexpr int JT = KP * KP / 256;
    constexpr int TPI = KP / JT;
    constexpr int CW = 32;
    constexpr int GS = KP * KP;
    __shared__ double As[KP][CW + 1], Bs[KP][CW + 1];
    const int t = threadIdx.x, i = t / TPI, j0 = (t % TPI) * JT;
    // a block walks cols_per_block columns (a multiple of 128; at most 64 blocks, so the one-block
    // finalize behind it reads at most 64 partial Grams however wide the data are)
    const int c0 = blockIdx.x * cols_per_block;
    int span = ld - c0;
    if (span > cols_per_block) span = cols_per_block;
    double acc1[JT], acc2[JT];
#pragma unroll
    for (int q = 0; q < JT; ++q) acc1[q] = acc2[q] = 0.0;
    for (int cc = 0; cc < span; cc += CW) {
        for (int e = t; e < KP * CW; e += 256) {
            As[e / CW][e % CW] = P[(long)(e / CW) * ld + c0 + cc + e % CW];
            Bs[e / CW][e % CW] = Q[(long)(e / CW) * ld + c0 + cc + e % CW];
        }
        __syncthreads();
        // rows >= k of P and Q are zero padding: a wave whose rows are all padding has nothing
        // to add (k = 5 of 32: three of the four waves, and the block is LDS-bound)
        if ((t >> 6) * (64 / TPI) < k) {
            for (int c = 0; c < CW; ++c) {
                const double a = As[i][c], b = Bs[i][c];
#pragma unroll
                for (int q = 0; q < JT; ++q) {
                    const double bq = Bs[j0 + q][c];
                    acc1[q] = fma(a, bq, acc1[q]);
                    acc2[q] = fma(b, bq, acc2[q]);
                }
            }
        }
        __syncthreads();
    }
    double *dst = partial + (size_t)blockIdx.x * 2 * GS;
#pragma unroll
    for (int q = 0; q < JT; ++q) {
        dst[i * KP + j0 + q] = acc1[q];
        dst[GS + i * KP + j0 + q] = acc2[q];
    }
}

// The same two Grams on the f64 matrix cores (round 4, pq_mfma): a wave owns 16 x 16 tiles of both
// (one tile at KP = 32), v_mfma_f64_16x16x4 with A = rows of P (or Q), B = rows of Q, four columns per
// instruction, thirty-two columns of loads in flight.  An entry's chain over the columns does not depend on
// where its tile sits, so restarts side by side (diagonal blocks of the stacked Gram) keep the bits of a
// single fit.  18 -> 7 us on the critical path of every outer iteration.
template <int KP>
__global__ __launch_bounds__(256) void k_gram_wide_pq_mfma(const double *__restrict__ P,
                                                           const double *__restrict__ Q, int ld,
                                                           double *__restrict__ partial, int cols_per_block)
{
    constexpr int T = KP / 16, GS = KP * KP;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane >> 4, lc = lane & 15;
    const int c0 = blockIdx.x * cols_per_block;
    int span = ld - c0;
    if (span > cols_per_block) span = cols_per_block;          // a multiple of 128 either way
    double *dst = partial + (size_t)blockIdx.x * 2 * GS;
    for (int tile = wave; tile < T * T; tile += 4) {
        const int mi = tile / T, nj = tile % T;
        // the contraction index is free to be permuted as long as both operands agree: lane (lc, lr) takes the
        // four CONSECUTIVE columns 16 it + 4 lr .. + 3 of its row (one 32-byte load; a row's 128 bytes come from
        // four lanes) and instruction u of the group multiplies everybody's u-th column
        const double *pa = P + (long)(16 * mi + lc) * ld + c0 + 4 * lr;
        const double *qa = Q + (long)(16 * mi + lc) * ld + c0 + 4 * lr;
        const double *qb = Q + (long)(16 * nj + lc) * ld + c0 + 4 * lr;
        f64x4 a1 = (f64x4){0.0, 0.0, 0.0, 0.0}, a2 = (f64x4){0.0, 0.0, 0.0, 0.0};
        for (int s = 0; s < span; s += 32) {                   // two groups of sixteen columns in flight
            f64x4 pv[2], qv[2], bv[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                pv[g] = *reinterpret_cast<const f64x4 *>(pa + s + 16 * g);
                qv[g] = *reinterpret_cast<const f64x4 *>(qa + s + 16 * g);
                bv[g] = *reinterpret_cast<const f64x4 *>(qb + s + 16 * g);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pv[g][u], bv[g][u], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(qv[g][u], bv[g][u], a2, 0, 0, 0);
                }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int e = (16 * mi + lr + 4 * reg) * KP + 16 * nj + lc;
            dst[e] = a1[reg];
            dst[GS + e] = a2[reg];
        }
    }
}

// partial [nb][2][GS] -> gram[1], gram[2] by the 1024 threads of a block: every element in four interleaved
// chains over the blocks, then ((0+1)+2)+3 (the order of k_gram_finalize).  Two elements per thread and trip
// with all their loads up front: 32 in flight instead of 16, half the dependent memory round trips.
__device__ __forceinline__ void sum_pq_partials(const double *__restrict__ partial, int nb, int GS,
                                                double *__restrict__ gram)
{
    const int t = threadIdx.x;
    for (int e0 = t; e0 < 2 * GS; e0 += 2048) {
        const int e1 = e0 + 1024;
        const bool two = e1 < 2 * GS;
        double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
        int b = 0;
        for (; b + 15 < nb; b += 16) {
            double va[16], vb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                va[u] = partial[(size_t)(b + u) * 2 * GS + e0];
                vb[u] = two ? partial[(size_t)(b + u) * 2 * GS + e1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                sa[u & 3] += va[u];
                sb[u & 3] += vb[u];
            }
        }
        for (; b < nb; ++b) {
            sa[b & 3] += partial[(size_t)b * 2 * GS + e0];
            if (two) sb[b & 3] += partial[(size_t)b * 2 * GS + e1];
        }
        gram[GS + e0] = ((sa[0] + sa[1]) + sa[2]) + sa[3];
        if (two) gram[GS + e1] = ((sb[0] + sb[1]) + sb[2]) + sb[3];
    }
}

// one block of 1024 threads: partials -> gram[1], gram[2], line search, Gram of the accepted
// point, cost
__global__ __launch_bounds__(1024) void k_linesearch_fin(const double *__restrict__ partial, int nb,
                                                         int KP, double *__restrict__ gram,
                                                         const double *__restrict__ M,
                                                         double *__restrict__ sc, aa_spg_params sp, int k,
                                                         double *__restrict__ ckct_state,
                                                         double n_global, double *__restrict__ cost_out,
                                                         int *__restrict__ cost_slot)
{
    __shared__ double smt[256];
    const int GS = KP * KP, t = threadIdx.x;
    // the order of k_gram_finalize: four interleaved chains over the blocks, then ((0+1)+2)+3
    sum_pq_partials(partial, nb, GS, gram);
    __syncthreads();
    // tr(M (PQ' + QP')) and tr(M QQ') in one sweep and one tree (fixed order)
    __shared__ double smt2[256];
    double tr1 = 0.0, tr2 = 0.0;
    {
        double a1 = 0.0, a2 = 0.0;
        if (t < 256)
            for (int e = t; e < k * k; e += 256) {
                const int i = e / k, j = e % k;
                const double m = M[i * KP + j];
                a1 += m * (gram[GS + j * KP + i] + gram[GS + i * KP + j]);
                a2 += m * gram[2 * GS + j * KP + i];
            }
        if (t < 256) {
            smt[t] = a1;
            smt2[t] = a2;
        }
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) {
                smt[t] += smt[t + o];
                smt2[t] += smt2[t + o];
            }
            __syncthreads();
        }
        tr1 = smt[0];
        tr2 = smt2[0];
    }
    if (t == 0) {
        linesearch_thread0(sc, sp, tr1, tr2);
        if (cost_out) {
            const int idx = cost_slot ? (*cost_slot)++ : 0;
            cost_out[idx] = 0.5 * (sc[SC_TRACE] - 2.0 * sc[SC_S1] + sc[SC_A0]) / n_global;
        }
    }
    __syncthreads();
    const double lam = sc[SC_LAMBDA];
    for (int e = t; e < GS; e += 1024) {
        const int r = e / KP, q = e % KP;
        const double v = gram[e] + lam * (gram[GS + e] + gram[GS + q * KP + r]) + lam * lam * gram[2 * GS + e];
        gram[e] = v;
        if (ckct_state) ckct_state[e] = v;
    }
}

// restarts side by side: k_linesearch_fin with the traces, the line search, the cost and the Gram of
// the accepted point once per slot, on the slot's diagonal blocks and scalar block
struct SlotRecords {        // per-slot cost records (aa_slots_*, aa_gpnh_slots_*)
    double *costs;          // [R][stride]
    int stride;
    int *counters;          // [R]
};
__global__ __launch_bounds__(1024) void k_linesearch_fin_slots(const double *__restrict__ partial, int nb,
                                                               int KP, double *__restrict__ gram,
                                                               const double *__restrict__ M,
                                                               double *__restrict__ scal, aa_spg_params sp,
                                                               int k, int R, double *__restrict__ ckct_state,
                                                               double n_global, SlotRecords rec)
{
    __shared__ double smt[256];
    __shared__ double smt2[256];
    const int GS = KP * KP, t = threadIdx.x;
    sum_pq_partials(partial, nb, GS, gram);
    __syncthreads();
    for (int r = 0; r < R; ++r) {
        const size_t off = (size_t)(r * k) * KP + r * k;
        const double *Mr = M + off, *G1 = gram + GS + off, *G2 = gram + 2 * GS + off;
        double *sc = scal + (size_t)r * AA_SC_STRIDE;
        double a1 = 0.0, a2 = 0.0;
        if (t < 256)
            for (int e = t; e < k * k; e += 256) {
                const int i = e / k, j = e % k;
                const double m = Mr[i * KP + j];
                a1 += m * (G1[j * KP + i] + G1[i * KP + j]);
                a2 += m * G2[j * KP + i];
            }
        if (t < 256) {
            smt[t] = a1;
            smt2[t] = a2;
        }
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) {
                smt[t] += smt[t + o];
                smt2[t] += smt2[t + o];
            }
            __syncthreads();
        }
        if (t == 0) {
            linesearch_thread0(sc, sp, smt[0], smt2[0]);
            double *cr = rec.costs + (size_t)r * rec.stride;
            int idx = rec.counters[r];
            if (idx >= rec.stride) idx = rec.stride - 1;          // a finished slot waiting to be replaced
            cr[idx] = 0.5 * (sc[SC_TRACE] - 2.0 * sc[SC_S1] + sc[SC_A0]) / n_global;
            rec.counters[r] = idx + 1;
        }
        __syncthreads();
    }
    for (int e = t; e < GS; e += 1024) {
        const int row = e / KP, q = e % KP;
        const int r = row / k;
        if (r >= R || q / k != r) continue;                       // diagonal blocks only
        const double lam = scal[(size_t)r * AA_SC_STRIDE + SC_LAMBDA];
        const double v = gram[e] + lam * (gram[GS + e] + gram[GS + q * KP + row]) + lam * lam * gram[2 * GS + e];
        gram[e] = v;
        if (ckct_state) ckct_state[e] = v;
    }
}

// Start of a dictionary update whose inputs are already on the device (the state a weights
// update leaves: Z'Z, C K C', C K Z in the Gram state): M = D Z'Z D
// (archetypal_analysis.py:310,330), gram[0] = C K C', the scalars, tr(C H D) = sum_i
// alpha_i (C K Z)_ii, and f(x) (spg.py:153-157) -- one block instead of seven launches.
__global__ __launch_bounds__(256) void k_dict_setup(const double *__restrict__ state /*ZtZ|CKCt|CKZ*/,
                                                    const double *__restrict__ alpha, int k, int KP,
                                                    double trace, double fnorm, double *__restrict__ Mout,
                                                    double *__restrict__ gram, double *__restrict__ sc,
                                                    aa_spg_params sp)
{
    __shared__ double sm[256];
    dict_setup_body(state, alpha, k, KP, trace, fnorm, Mout, gram, sc, sp, sm);
}

// restarts side by side: block r sets up slot r from the diagonal blocks of the Gram state -- the
// arithmetic of k_dict_setup on its k x k block (the off-diagonal blocks of M stay zero since
// aa_slots_begin: the gradient kernel then sees a block-diagonal M and the slots do not mix)
__global__ __launch_bounds__(256) void k_dict_setup_slots(const double *__restrict__ state /*ZtZ|CKCt|CKZ*/,
                                                          const double *__restrict__ alpha, int k, int KP,
                                                          double trace, double fnorm, double *__restrict__ Mout,
                                                          double *__restrict__ gram, double *__restrict__ scal,
                                                          aa_spg_params sp, unsigned slotmask)
{
    __shared__ double sm[256];
    const int GS = KP * KP, t = threadIdx.x, r = blockIdx.x, o = r * k;
    if (!((slotmask >> r) & 1u)) return;
    const size_t off = (size_t)o * KP + o;
    const double *ZtZ = state + off, *CKCt = state + GS + off, *CKZ = state + 2 * GS + off;
    const double *al = alpha + o;
    double *M = Mout + off, *G = gram + off, *sc = scal + (size_t)r * AA_SC_STRIDE;
    for (int e = t; e < k * k; e += 256) {
        const int i = e / k, j = e % k;
        M[i * KP + j] = al[i] * ZtZ[i * KP + j] * al[j];
        G[i * KP + j] = CKCt[i * KP + j];
    }
    __syncthreads();
    const double a0 = block_trace_MG(M, G, k, KP, false, sm);
    sm[t] = t < k ? al[t] * CKZ[t * KP + t] : 0.0;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (t < q) sm[t] += sm[t + q];
        __syncthreads();
    }
    if (t == 0) {
        sc[SC_TRACE] = trace;
        sc[SC_FNORM] = fnorm;
        sc[SC_S1] = sm[0];
        sc[SC_A0] = a0;
        sc[SC_F_OLD] = 0.5 * (trace - 2.0 * sm[0] + a0) / fnorm;
        sc[SC_NFEVAL] = 1.0;
        sc[SC_FLAGS] = 0.0;
        for (int i = 0; i < 16; ++i) sc[SC_FMEM0 + i] = 0.0;   // f_mem = zeros (spg.py:153)
        sc[SC_ALPHA_SET] = (sp.alpha0 >= 0.0) ? 1.0 : 0.0;
        sc[SC_ALPHA] = sp.alpha0;
    }
}

// ---------------------------------------------------------------- data-matrix reductions
template <typename T>
__global__ __launch_bounds__(256) void k_sqnorm(const T *__restrict__ X, long ldx, long n, int p,
                                                double *__restrict__ partial)
{
    __shared__ double sm[256];
    double s = 0.0;
    for (long r = blockIdx.x; r < n; r += gridDim.x)
        for (int c = threadIdx.x; c < p; c += 256) {
            const double v = (double)X[r * ldx + c];
            s = fma(v, v, s);
        }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

template <typename T>
__global__ __launch_bounds__(256) void k_diag_sum(const T *__restrict__ K, long ldx, long n,
                                                  long col_offset, double *__restrict__ partial)
{
    __shared__ double sm[256];
    double s = 0.0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long)gridDim.x * 256)
        s += (double)K[r * ldx + col_offset + r];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// d[i] = sqrt(max(0, <xi,xi> - 2 <xi,xj> + <xj,xj>)), one wave per row; the three dot
// products use the same lane partition and butterfly so that i == j gives exactly 0
// (archetypal_analysis.py:95-100 on K = XX' without forming K).
template <typename T>
__global__ __launch_bounds__(256) void k_distance_data(const T *__restrict__ X, long ldx, long n,
                                                       int p_pad, const T *__restrict__ xj,
                                                       double *__restrict__ d)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + wave;
    if (r >= n) return;
    double sii = 0.0, sij = 0.0, sjj = 0.0;
    for (int c = lane; c < p_pad; c += 64) {
        const double a = (double)X[r * ldx + c], b = (double)xj[c];
        sii = fma(a, a, sii);
        sij = fma(a, b, sij);
        sjj = fma(b, b, sjj);
    }
    sii = wave_sum(sii);
    sij = wave_sum(sij);
    sjj = wave_sum(sjj);
    if (lane == 0) d[r] = sqrt(fmax(sii - 2.0 * sij + sjj, 0.0));
}

// implicit RBF kernel: d[i] = sqrt(K_ii - 2 K_ij + K_jj) = sqrt(2 - 2 exp(-gamma ||x_i - x_j||^2)), one wave per
// row, the squared distance summed directly (i == j gives exactly 0)
__global__ __launch_bounds__(256) void k_distance_rbf(const double *__restrict__ F, long ldf, int pf, long n, long j,
                                                      double gamma, double *__restrict__ d)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + wave;
    if (r >= n) return;
    double s = 0.0;
    for (int q = lane; q < pf; q += 64) {
        const double df = F[r * ldf + q] - F[j * ldf + q];
        s = fma(df, df, s);
    }
    s = wave_sum(s);
    if (lane == 0) d[r] = sqrt(fmax(2.0 - 2.0 * exp(-gamma * s), 0.0));
}

template <typename T>
__global__ __launch_bounds__(256) void k_distance_kernel(const T *__restrict__ K, long ldx, long n,
                                                         long j, double *__restrict__ d)
{
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const double kd = (double)K[r * ldx + r], kj = (double)K[r * ldx + j],
                 jj = (double)K[j * ldx + j];
    d[r] = sqrt(kd - 2.0 * kj + jj);
}

// sum_r || x_r - sum_i z[r][i] * alpha[i] * W[i][:] ||^2, one wave per row.
template <typename T, int KP>
__global__ __launch_bounds__(256) void k_residual(const T *__restrict__ X, long ldx, long n,
                                                  int p_pad, const double *__restrict__ Z,
                                                  const double *__restrict__ W,
                                                  const double *__restrict__ alpha, int k,
                                                  double *__restrict__ partial)
{
    __shared__ double zs[4][KP];
    __shared__ double ws[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + wave;
    double s = 0.0;
    if (r < n) {
        if (lane < KP) zs[wave][lane] = (lane < k) ? Z[r * KP + lane] * (alpha ? alpha[lane] : 1.0) : 0.0;
    }
    __syncthreads();
    if (r < n) {
        for (int c = lane; c < p_pad; c += 64) {
            double rec = 0.0;
            for (int i = 0; i < k; ++i) rec = fma(zs[wave][i], W[(long)i * p_pad + c], rec);
            const double df = (double)X[r * ldx + c] - rec;
            s = fma(df, df, s);
        }
    }
    s = wave_sum(s);
    if (lane == 0) ws[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_sum_partials(const double *__restrict__ partial, long nb,
                                                      double *__restrict__ out)
{
    __shared__ double sm[256];
    double s = 0.0;
    for (long b = threadIdx.x; b < nb; b += 256) s += partial[b];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// ===========================================================================
// host launchers
// ===========================================================================
static inline long tall_rows_pb(const Ctx *c)
{
    const int RS = 256 / c->KP;
    long rpb = (c->n + c->tallBlocks - 1) / c->tallBlocks;
    return round_up(rpb < RS ? RS : rpb, RS);
}

int tall_setup(Ctx *c)
{
    long nb = (c->n + 255) / 256;
    if (nb > 256) nb = 256;           // one block per CU; finalize sums <= 256 partials
    if (nb < 1) nb = 1;
    c->tallBlocks = (int)nb;
    // gram partials: tall grams use tallBlocks blocks, wide grams p_pad/128 blocks
    long gb = nb > c->p_pad / 128 ? nb : c->p_pad / 128;
    if (gb < 256) gb = 256;          // tall Grams use up to 256 blocks whatever n is
    if (gb < 2 * (c->p_pad / 128)) gb = 2 * (c->p_pad / 128);   // k_gram_wide_pq: two Grams per block
    size_t need = (size_t)gb * c->KP * c->KP * sizeof(double);
    size_t need2 = (size_t)nb * 4 * c->KP * sizeof(double) + 4 * c->KP * sizeof(double);
    // residual / distance partials: one per 4 rows
    size_t need3 = (size_t)((c->n + 3) / 4 + 1024) * sizeof(double);
    if (need2 > need) need = need2;
    if (need3 > need) need = need3;
    AA_CHECK(c->redPartial.alloc(need + 4096));
    AA_CHECK(c->gramOut.alloc((size_t)4 * c->KP * c->KP * sizeof(double)));
    AA_CHECK(c->redOut.alloc((size_t)8 * c->KP * sizeof(double)));
    AA_CHECK(c->gramState.alloc((size_t)3 * c->KP * c->KP * sizeof(double)));
    AA_CHECK(c->scalars.alloc(SC_COUNT * sizeof(double)));
    AA_CHECK(c->proj.alloc(sizeof(ProjState)));
    {   // candidate lists of the projection: [KP][tallBlocks * RS][rows_pb / RS]
        const long rpb = tall_rows_pb(c);
        AA_CHECK(c->projList.alloc((size_t)c->KP * c->tallBlocks * rpb * sizeof(double)));
        AA_CHECK(c->projSegCnt.alloc((size_t)c->KP * c->tallBlocks * (256 / c->KP) * sizeof(int)));
    }
    AA_CHECK(c->Mdev.alloc((size_t)c->KP * c->KP * sizeof(double)));
    AA_CHECK(c->alphaDev.alloc((size_t)c->KP * sizeof(double)));
    if (g_proj_res_side && c->stream2 && c->world <= 1 && !c->force_comm) {
        // scratch set of the side-stream projection (launch_proj_side)
        AA_CHECK(c->redPartial2.alloc(need + 4096));
        AA_CHECK(c->redOut2.alloc((size_t)8 * c->KP * sizeof(double)));
        AA_CHECK(c->proj2.alloc(sizeof(ProjState)));
        const long rpb = tall_rows_pb(c);
        AA_CHECK(c->projList2.alloc((size_t)c->KP * c->tallBlocks * rpb * sizeof(double)));
        AA_CHECK(c->projSegCnt2.alloc((size_t)c->KP * c->tallBlocks * (256 / c->KP) * sizeof(int)));
        AA_CHECK(c->tmpTall2.alloc((size_t)(c->n_pad + AA_SLACK_ROWS) * c->KP * sizeof(double)));
        if (!c->evFork2) AA_CHECK_HIP(hipEventCreateWithFlags(&c->evFork2, hipEventDisableTiming));
        if (!c->evJoin2) AA_CHECK_HIP(hipEventCreateWithFlags(&c->evJoin2, hipEventDisableTiming));
    }
    return AA_OK;
}

static inline double *red_buf(Ctx *c) { return c->redOut.as<double>(); }

static int finalize_and_post(Ctx *c, int NV, unsigned max_mask, int kind, int mode, int slot,
                             bool gated, int nb = 0, const aa_spg_params *sp = nullptr,
                             int stage_after = -1)
{
    aa_spg_params spv;
    memset(&spv, 0, sizeof(spv));
    if (sp) spv = *sp;
    else stage_after = -1;
    if (nb <= 0) nb = c->tallBlocks;
    double *part = c->redPartial.as<double>();
    double *red = red_buf(c);
    ProjState *ps = c->proj.as<ProjState>();
    const ProjState *gate = gated ? ps : (const ProjState *)nullptr;
    if ((c->world <= 1 && !c->force_comm)) {
        hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(FIN_NT), 0, c->stream, part, nb, NV,
                           c->KP, max_mask, red, gate, kind, mode, c->k, ps, c->scalars.as<double>(),
                           slot, (double *)nullptr, 0, 1, stage_after, spv, c->slots_aa ? c->slots_k : 0,
                           c->slots_aa ? c->slots_R : 0);
    } else {
        // one sum all-reduce of a [world][NV][KP] buffer in which every rank fills its own
        // slot, whatever mix of sums and maxima the NV values are; the ranks' values are
        // combined in rank order by k_post.  (When a gated pass has already converged the
        // reduced values are stale but unused: k_post exits.)
        AA_CHECK(c->redGather.alloc((size_t)c->world * 5 * c->KP * sizeof(double)));
        double *gather = c->redGather.as<double>();
        const long region = (long)c->world * 4 * c->KP;
        // riders (pack_comm): a projection's closing reduction goes with the wide all-reduce that follows it
        // (ride_dst set by the caller), the gradient's <d, g_new> with the first reduction of the residual
        // projection behind it (second region of the gather buffer)
        const bool rides_wide = c->ride_gather_next && kind == POST_FIN && !gated;
        const bool rides_first = c->ride_grad_next && kind == POST_SCALAR_SUM && !gated;
        if (rides_wide) gather = c->ride_gather_next;
        if (rides_first) gather += region;
        hipLaunchKernelGGL(k_finalize_sum, dim3(1), dim3(FIN_NT), 0, c->stream, part, nb, NV,
                           c->KP, max_mask, red, gate, (int)POST_NONE, mode, c->k, ps,
                           c->scalars.as<double>(), slot, gather, c->rank, c->world, -1, spv);
        if (rides_wide || rides_first) {
            Ctx::RidePost &rp = rides_wide ? c->ride : c->ride_grad;
            rp.on = true;
            rp.kind = kind; rp.mode = mode; rp.NV = NV; rp.slot = slot; rp.gated = 0; rp.stage_after = stage_after;
            rp.max_mask = max_mask; rp.sp = spv; rp.red = red; rp.ps = ps;
            rp.gather = gather; rp.count = (long)c->world * NV * c->KP;
            if (rides_wide) c->ride_gather_next = nullptr;
            else c->ride_grad_next = false;
            AA_CHECK_HIP(hipGetLastError());
            return AA_OK;
        }
        long count = (long)c->world * NV * c->KP;
        if (c->ride_grad.on) count = region + (long)c->world * c->ride_grad.NV * c->KP;   // (the gap is summed along, unused)
        AA_CHECK(comm_allreduce(c, gather, count, 0));
        if (c->ride_grad.on) AA_CHECK(launch_ride_post(c, &c->ride_grad, gather + region));
        hipLaunchKernelGGL(k_post, dim3(1), dim3(256), 0, c->stream, kind, mode, red, c->KP, c->k, ps,
                           c->scalars.as<double>(), slot, gated ? 1 : 0, (const double *)gather,
                           c->world, NV, max_mask, stage_after, spv);
    }
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_ride_post(Ctx *c, Ctx::RidePost *rp, const double *gather)
{
    rp->on = false;
    hipLaunchKernelGGL(k_post, dim3(1), dim3(256), 0, c->stream, rp->kind, rp->mode, rp->red, c->KP, c->k, rp->ps,
                       c->scalars.as<double>(), rp->slot, rp->gated, gather, c->world, rp->NV, rp->max_mask,
                       rp->stage_after, rp->sp);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

#define TALL_DISPATCH_NT(NTHREADS, KERNEL, ...)                                             \
    do {                                                                                    \
        if (c->KP == 32)                                                                    \
            hipLaunchKernelGGL(KERNEL<32>, dim3(c->tallBlocks), dim3(NTHREADS), 0,          \
                               c->stream, __VA_ARGS__);                                     \
        else                                                                                \
            hipLaunchKernelGGL(KERNEL<64>, dim3(c->tallBlocks), dim3(NTHREADS), 0,          \
                               c->stream, __VA_ARGS__);                                     \
    } while (0)

#define TALL_DISPATCH(KERNEL, ...)                                                          \
    do {                                                                                    \
        if (c->KP == 32)                                                                    \
            hipLaunchKernelGGL(KERNEL<32>, dim3(c->tallBlocks), dim3(256), 0, c->stream,    \
                               __VA_ARGS__);                                                \
        else                                                                                \
            hipLaunchKernelGGL(KERNEL<64>, dim3(c->tallBlocks), dim3(256), 0, c->stream,    \
                               __VA_ARGS__);                                                \
    } while (0)

// Michelot passes are enqueued in batches sized from the previous projection's pass
// count; after each batch the host reads the device-side `done` flag (one small
// synchronisation, cheaper than enqueuing a dozen passes that exit immediately).
static int g_proj_hard_cap = 200;

int g_proj_mode = 0;        // 0: candidate lists, 1: iterative full passes
int g_proj_check_always = 0; // multi-rank: 1 = every list projection is checked for overflow at once (one host
                             // synchronisation; an overflowing column falls back to the iterative passes in the same
                             // projection), 0 = the check is deferred to the next poll while the lists of that kind of
                             // projection have been short -- faster, but a list that outgrows its slot between two
                             // polls (seen: the residual projection a few iterations after a FurthestSum start) then
                             // ends the fit with an error instead of being handled
int g_pq_mfma = 1;              // the line search's two wide Grams on the f64 matrix cores (k_gram_wide_pq_mfma)
int g_pq_blocks = 128;          // most blocks of k_gram_wide_pq (their partial Grams are summed by ONE block; 64 -> 128:
                                // C2, p = 25 000, 0.553 -> 0.537 ms per iteration; 256: 0.548)
int g_pack_comm = 1;            // multi-rank: small reductions ride in the tail of the next all-reduce (Ctx::ride*)
int g_setup_in_grad = 1;        // the dictionary update's set-up block inside its first gradient launch (DictSetup)
int g_gram_side = 0;            // Z'Z of the refresh after a weights update on the side stream, beside the Z'X pass
int g_grad_side = 1;            // one SPG iteration per dictionary update: g_new, x += lambda d and the BB stage on the side stream too
int g_proj_res_side = 1;        // the SPG's residual projection (flags only) on the side stream, beside the weights QP
int g_proj_small = 1;           // short columns: threshold search of a projection in one kernel (k_proj_small)
int g_fuse_finalize = 1;    // 1: the scalar stage that consumes a reduction rides in the kernel that finalizes it
int g_fin_in_last = 1;      // 1: the projections' reductions are finalized by the last block of the pass (FinTail, single rank)
int g_proj_list_cap = 2048; // multi-rank: most candidates per rank and column in the list all-reduce
                            // (the union must fit the solver's LDS: effective cap = min(this, 2048 / world))

static int proj_iterative_passes(Ctx *c, const double *wsrc, int mode, long rpb, int first_batch)
{
    double *part = c->redPartial.as<double>();
    const double *scal = c->scalars.as<double>();
    ProjState *ps = c->proj.as<ProjState>();
    int batch = first_batch;
    int total = 0;
    int hdr[2] = {0, 0};
    while (true) {
        for (int it = 0; it < batch; ++it) {
            TALL_DISPATCH(k_proj_pass, wsrc, (const double *)nullptr, 0.0, scal, -1, c->n, rpb,
                          c->k, (const ProjState *)ps, part);
            AA_CHECK(finalize_and_post(c, 2, 0u, POST_MICHELOT, 0, 0, true));
        }
        total += batch;
        AA_CHECK_HIP(hipMemcpyAsync(hdr, &ps->done, sizeof(hdr), hipMemcpyDeviceToHost, c->stream));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (hdr[0] || total >= g_proj_hard_cap) break;
        batch = 3;
    }
    c->projPassHint[mode] = hdr[1];
    return AA_OK;
}

// multi-rank: deferred overflow check of the list projections and refresh of the "lists are
// short" hints.  Called wherever the host synchronises anyway.
int proj_poll_multirank(Ctx *c)
{
    if (!(c->world > 1 || c->force_comm) || !c->proj.p) return AA_OK;
    AA_CHECK(p2p_check(c));
    ProjState *ps = c->proj.as<ProjState>();
    int h[5] = {0, 0, 0, 0, 0};
    AA_CHECK_HIP(hipMemcpyAsync(h, &ps->overflow_sticky, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    AA_REQUIRE(h[0] == 0, AA_ERR_STATE,
               "multi-rank projection: a candidate list outgrew the gather buffer between two polls "
               "(set option proj_check=1 or proj_mode=1) [flag %d, longest lists %d %d %d %d]", h[0], h[1],
               h[2], h[3], h[4]);
    // "short": even if every candidate of a column sat on ONE rank, twice as many would still fit
    // that rank's slot of the gather buffer
    int cap = g_proj_list_cap;
    if (cap * c->world > PROJ_LDS_CAP) cap = PROJ_LDS_CAP / c->world;
    for (int m = 0; m < 4; ++m) c->projListShort[m] = c->projWarm[m] && h[1 + m] > 0 && h[1 + m] <= cap / 2;
    return AA_OK;
}

// stage_after >= 0: the scalar stage (ST_ALPHA / ST_CONV) that consumes this projection's
// reductions runs inside the kernel that finalizes them.
int launch_proj(Ctx *c, const double *x, const double *g, double a_const, int a_slot, int mode,
                const aa_spg_params *sp, int stage_after)
{
    const long rpb = tall_rows_pb(c);
    double *part = c->redPartial.as<double>();
    double *scal = c->scalars.as<double>();
    ProjState *ps = c->proj.as<ProjState>();
    double *wbuf = g ? c->tmpTall.as<double>() : (double *)nullptr;
    const double *wsrc = g ? (const double *)wbuf : x;
    const bool multi = c->world > 1 || c->force_comm;
    // the reductions' second stages in the last block of their pass (FinTail)
    const bool fin_in_last = g_fin_in_last && !multi && g_fuse_finalize && (int)c->tallBlocks > 1;
    // (the 64-entries-per-thread instantiation would cover the 12 500-row shards; measured there: 0.576 ms per
    // outer iteration with it against 0.533 with the four list launches -- 32 blocks walking 12 500 strided
    // entries each are slower than 256 blocks streaming them -- so the default stays at 8192 rows; option proj_small=2)
    if (g_proj_mode == 0 && !multi && g_proj_small && c->n <= (g_proj_small >= 2 ? 256L * 64 : 256L * 32)) {
        if (c->n <= 256L * 32)
            hipLaunchKernelGGL(k_proj_small<32>, dim3(c->k), dim3(256), 0, c->stream, x, g, a_const, (const double *)scal,
                               a_slot, c->n, c->KP, c->projWarm[mode] ? mode : 0, ps, c->slots_aa ? c->slots_k : 0);
        else
            hipLaunchKernelGGL(k_proj_small<64>, dim3(c->k), dim3(256), 0, c->stream, x, g, a_const, (const double *)scal,
                               a_slot, c->n, c->KP, c->projWarm[mode] ? mode : 0, ps, c->slots_aa ? c->slots_k : 0);
    } else if (g_proj_mode == 0) {
        const int RS = 256 / c->KP;
        const long nseg = (long)c->tallBlocks * RS, segcap = rpb / RS;
        FinTail ft1;
        memset(&ft1, 0, sizeof(ft1));
        if (fin_in_last) {
            ft1.on = 1; ft1.NV = 3; ft1.max_mask = 1u; ft1.red = red_buf(c); ft1.kind = POST_FIRST; ft1.mode = 0;
            ft1.k = c->k; ft1.ps = ps; ft1.scal = scal; ft1.slot = 0; ft1.stage_after = -1;
            ft1.kslot = c->slots_aa ? c->slots_k : 0; ft1.R = c->slots_aa ? c->slots_R : 0;
        }
        TALL_DISPATCH_NT(PROJ_NT, k_proj_first, x, g, a_const, (const double *)scal, a_slot, c->n, rpb, c->k,
                      c->projWarm[mode] ? mode : 0, (const ProjState *)ps, wbuf, part, c->slots_aa ? c->slots_k : 0, ft1);
        if (!fin_in_last) AA_CHECK(finalize_and_post(c, 3, 1u, POST_FIRST, 0, 0, false));
        TALL_DISPATCH(k_proj_collect, wsrc, c->n, rpb, c->k, (const ProjState *)ps,
                      c->projList.as<double>(), c->projSegCnt.as<int>());
        if (!multi) {
            hipLaunchKernelGGL(k_proj_solve<false>, dim3(c->k), dim3(256), 0, c->stream,
                               (const double *)c->projList.as<double>(),
                               (const int *)c->projSegCnt.as<int>(), nseg, segcap, ps,
                               (double *)nullptr, 0, 0);
        } else {
            // every rank packs its candidates into its slot of [world][KP][cap + 1]; one sum
            // all-reduce puts all lists on all ranks; each rank solves the union (same bits)
            // (slot layout [KP][cap + 2]: candidates, true count, largest candidate left out)
            int cap = g_proj_list_cap;
            if (cap * c->world > PROJ_LDS_CAP) cap = PROJ_LDS_CAP / c->world;
            if (cap < 1) cap = 1;
            const size_t stride = (size_t)cap + 2, slot = (size_t)c->KP * stride;
            AA_CHECK(c->listGather.alloc((size_t)c->world * slot * sizeof(double)));
            double *gl = c->listGather.as<double>();
            AA_CHECK_HIP(hipMemsetAsync(gl, 0, (size_t)c->world * slot * sizeof(double), c->stream));
            hipLaunchKernelGGL(k_proj_solve<true>, dim3(c->k), dim3(256), 0, c->stream,
                               (const double *)c->projList.as<double>(),
                               (const int *)c->projSegCnt.as<int>(), nseg, segcap, ps,
                               gl + (size_t)c->rank * slot, cap, (int)stride);
            AA_CHECK(comm_allreduce(c, gl, (long)c->world * (long)slot, 0));
            AA_CHECK_HIP(hipMemsetAsync(&ps->list_max[mode & 3], 0, sizeof(int), c->stream));
            hipLaunchKernelGGL(k_proj_solve_gathered, dim3(c->k), dim3(256), 0, c->stream,
                               (const double *)gl, c->world, cap, c->KP, ps, mode);
            // did every column fit and converge?  (identical on all ranks)  Checked at once --
            // a host synchronisation -- only while this kind of projection is not known to have
            // short lists; after that the sticky device flag is read at the next poll
            // (proj_poll_multirank), and a missed overflow is an error there, never a wrong result
            if (!c->projListShort[mode & 3] || g_proj_check_always) {
                int conv[AA_MAX_K];
                AA_CHECK_HIP(hipMemcpyAsync(conv, ps->shrunk, (size_t)c->k * sizeof(int),
                                            hipMemcpyDeviceToHost, c->stream));
                AA_CHECK_HIP(hipStreamSynchronize(c->stream));
                bool all = true;
                for (int i = 0; i < c->k; ++i) all = all && conv[i] != 0;
                if (!all) {
                    hipLaunchKernelGGL(k_proj_fallback_init, dim3(1), dim3(64), 0, c->stream, ps, c->k);
                    AA_CHECK_HIP(hipMemsetAsync(&ps->overflow_sticky, 0, sizeof(int), c->stream));   // handled
                    AA_CHECK(proj_iterative_passes(c, wsrc, mode, rpb, 6));
                }
                AA_CHECK(proj_poll_multirank(c));
            }
        }
        AA_CHECK_HIP(hipGetLastError());
    } else {
        TALL_DISPATCH(k_proj_colmax, x, g, a_const, (const double *)scal, a_slot, c->n, rpb, c->k, wbuf, part);
        AA_CHECK(finalize_and_post(c, 1, 1u, POST_COLMAX, 0, c->projWarm[mode] ? mode : 0, false));
        AA_CHECK(proj_iterative_passes(c, wsrc, mode, rpb,
                                       c->projPassHint[mode] > 0 ? c->projPassHint[mode] + 1 : 12));
    }
    double *out = nullptr;
    if (mode == PROJ_FEAS) out = const_cast<double *>(x);
    if (mode == PROJ_DIR) out = c->Dt.as<double>();
    FinTail ft2;
    memset(&ft2, 0, sizeof(ft2));
    if (fin_in_last) {
        ft2.on = 1; ft2.NV = 4; ft2.max_mask = 8u; ft2.red = red_buf(c); ft2.kind = POST_FIN; ft2.mode = mode;
        ft2.k = c->k; ft2.ps = ps; ft2.scal = scal; ft2.slot = 0; ft2.stage_after = sp ? stage_after : -1;
        if (sp) ft2.sp = *sp;
        ft2.kslot = c->slots_aa ? c->slots_k : 0; ft2.R = c->slots_aa ? c->slots_R : 0;
    }
    TALL_DISPATCH_NT(PROJ_NT, k_proj_finish, mode, x, g, a_const, (const double *)scal, a_slot,
                  (const double *)c->H.as<double>(), (const double *)c->alphaDev.as<double>(), c->n, rpb,
                  c->k, (const ProjState *)ps, out, part, c->slots_aa ? c->slots_k : 0,
                  (c->slots_aa && mode == PROJ_FEAS) ? c->slots_cold_cols : 0xffffffffu, ft2);
    if (fin_in_last) {
    } else if (sp && stage_after >= 0 && !g_fuse_finalize) {
        AA_CHECK(finalize_and_post(c, 4, 8u, POST_FIN, mode, 0, false));
        AA_CHECK(launch_scalar_stage(c, stage_after, sp, 0));
    } else {
        AA_CHECK(finalize_and_post(c, 4, 8u, POST_FIN, mode, 0, false, 0, sp, stage_after));
    }
    AA_CHECK_HIP(hipGetLastError());
    if (mode > 0) c->projWarm[mode] = true;
    return AA_OK;
}

// The same projection on the side stream with the second scratch set: the members launch_proj
// works on are swapped for the call (host code is single-threaded per context).  Used for the
// residual projection at the end of a dictionary SPG iteration, whose only consumers are the
// convergence flags: it then runs beside the weights QP (VALU-bound, 0.3 ms) instead of in front of it.
int join_side(Ctx *c)
{
    if (!c->side_pending) return AA_OK;
    c->side_pending = false;
    AA_CHECK_HIP(hipStreamWaitEvent(c->stream, c->evJoin2, 0));
    return AA_OK;
}

// The side stream with its own scratch set (projection state, partials, reduction outputs, candidate
// lists, the w buffer): between side_begin and side_end every launch_* call of this file goes to
// stream2.  side_begin forks from the main stream (everything enqueued so far is visible), side_end
// records the event join_side waits for.
static void side_swap(Ctx *c)
{
    std::swap(c->stream, c->stream2);
    std::swap(c->tmpTall, c->tmpTall2);
    std::swap(c->redPartial, c->redPartial2);
    std::swap(c->redOut, c->redOut2);
    std::swap(c->proj, c->proj2);
    std::swap(c->projList, c->projList2);
    std::swap(c->projSegCnt, c->projSegCnt2);
    for (int m = 0; m < 4; ++m) {
        std::swap(c->projWarm[m], c->projWarm2[m]);
        std::swap(c->projPassHint[m], c->projPassHint2[m]);
    }
}

bool side_available(const Ctx *c)
{
    return g_proj_res_side && c->stream2 && c->proj2.p && c->evFork2 && c->world <= 1 && !c->force_comm &&
           g_proj_mode == 0 && g_fuse_finalize;
}

int side_begin(Ctx *c)
{
    AA_CHECK(join_side(c));
    AA_CHECK_HIP(hipEventRecord(c->evFork2, c->stream));
    AA_CHECK_HIP(hipStreamWaitEvent(c->stream2, c->evFork2, 0));
    side_swap(c);
    return AA_OK;
}

// more side work behind whatever is pending there, after everything the main stream holds so far
// (no join in between: the main stream keeps going)
int side_begin_behind(Ctx *c)
{
    AA_CHECK_HIP(hipEventRecord(c->evFork2, c->stream));
    AA_CHECK_HIP(hipStreamWaitEvent(c->stream2, c->evFork2, 0));
    side_swap(c);
    return AA_OK;
}

int side_end(Ctx *c)
{
    side_swap(c);
    AA_CHECK_HIP(hipEventRecord(c->evJoin2, c->stream2));
    c->side_pending = true;
    return AA_OK;
}

int launch_proj_side(Ctx *c, const double *x, const double *g, double a_const, int a_slot, int mode,
                     const aa_spg_params *sp, int stage_after)
{
    if (!side_available(c)) return launch_proj(c, x, g, a_const, a_slot, mode, sp, stage_after);
    AA_CHECK(side_begin(c));
    const int rc = launch_proj(c, x, g, a_const, a_slot, mode, sp, stage_after);
    AA_CHECK(side_end(c));
    return rc;
}

// xupd (nullable, needs d_for_dot): x += lambda d in the same pass.  stage_after >= 0 (with
// sp): the scalar stage that consumes <d, g> (ST_BB) runs inside the finalize kernel.
int launch_grad(Ctx *c, const double *Graw, const double *H, double *gout, double scale,
                const double *d_for_dot, int dot_slot, double *xupd, const aa_spg_params *sp,
                int stage_after, const aa_spg_params *setup_sp, double setup_fnorm)
{
    // setup_sp: the dictionary update's set-up rides along as block 0 (DictSetup)
    DictSetup ds;
    memset(&ds, 0, sizeof(ds));
    if (setup_sp) {
        ds.on = 1;
        ds.state = c->gramState.as<double>();
        ds.trace = c->trace;
        ds.fnorm = setup_fnorm;
        ds.Mout = c->Mdev.as<double>();
        ds.gram = c->gramOut.as<double>();
        ds.sc = c->scalars.as<double>();
        ds.sp = *setup_sp;
    }
    // 16 rows per wave and step with nothing in flight across steps: ~4 blocks per CU
    long gb = (c->n + 63) / 64;
    if (gb > 1024) gb = 1024;
    if (gb < 1) gb = 1;
    const long rpb = round_up((c->n + gb - 1) / gb, 64);
    const int nb = (int)((c->n + rpb - 1) / rpb);
    double *part = c->redPartial.as<double>();
    double *pdot = d_for_dot ? part : (double *)nullptr;
    const int kslot = c->slots_aa ? c->slots_k : 0;
    if (kslot && d_for_dot) pdot = c->tmpTall.as<double>();      // k_grad's mixed dot is not used: per slot below
    if (c->KP == 32)
        hipLaunchKernelGGL(k_grad<32>, dim3(nb + ds.on), dim3(256), 0, c->stream, Graw, H,
                           (const double *)c->Mdev.as<double>(), (const double *)c->alphaDev.as<double>(),
                           scale, c->n, rpb, c->k, gout, d_for_dot, pdot, xupd,
                           (const double *)c->scalars.as<double>(), kslot, c->slots_R, ds);
    else
        hipLaunchKernelGGL(k_grad<64>, dim3(nb + ds.on), dim3(256), 0, c->stream, Graw, H,
                           (const double *)c->Mdev.as<double>(), (const double *)c->alphaDev.as<double>(),
                           scale, c->n, rpb, c->k, gout, d_for_dot, pdot, xupd,
                           (const double *)c->scalars.as<double>(), kslot, c->slots_R, ds);
    if (kslot && d_for_dot) {
        const dim3 gd((unsigned)nb, (unsigned)c->slots_R);
        if (c->KP == 32)
            hipLaunchKernelGGL(k_grad_dot_slots<32>, gd, dim3(256), 0, c->stream, (const double *)gout, d_for_dot,
                               c->n, rpb, kslot, c->slots_R, part);
        else
            hipLaunchKernelGGL(k_grad_dot_slots<64>, gd, dim3(256), 0, c->stream, (const double *)gout, d_for_dot,
                               c->n, rpb, kslot, c->slots_R, part);
    }
    AA_CHECK_HIP(hipGetLastError());
    if (d_for_dot) {
        if (sp && stage_after >= 0 && !g_fuse_finalize) {
            AA_CHECK(finalize_and_post(c, 1, 0u, POST_SCALAR_SUM, 0, dot_slot, false, nb));
            AA_CHECK(launch_scalar_stage(c, stage_after, sp, 0));
        } else {
            AA_CHECK(finalize_and_post(c, 1, 0u, POST_SCALAR_SUM, 0, dot_slot, false, nb, sp, stage_after));
        }
    }
    return AA_OK;
}

int launch_tall_axpy_lambda(Ctx *c, double *x, const double *d)
{
    const long elems = c->n * c->KP;
    hipLaunchKernelGGL(k_tall_axpy, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream,
                       x, d, c->scalars.as<double>(), elems);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_tall_dot_scaled(Ctx *c, const double *x, const double *H, const double *alpha_dev, int slot)
{
    const long rpb = tall_rows_pb(c);
    double *part = c->redPartial.as<double>();
    TALL_DISPATCH(k_tall_dot_scaled, x, H, alpha_dev, c->n, rpb, c->k, part);
    AA_CHECK(finalize_and_post(c, 1, 0u, POST_SCALAR_SUM, 0, slot, false));
    return AA_OK;
}

int launch_gram_tall(Ctx *c, const double *A, const double *B, double *out_dev, bool local_only)
{
    const int want = 256;                                      // one block per CU
    long rpb = round_up((c->n_pad + want - 1) / want, 64);     // 4 row groups per step
    const int nb = (int)((c->n_pad + rpb - 1) / rpb);
    double *part = c->redPartial.as<double>();
    const int elems = c->KP * c->KP;
    if (c->KP == 32)
        hipLaunchKernelGGL(k_gram_tall<32>, dim3(nb), dim3(256), 0, c->stream, A, B, c->n_pad, rpb, part);
    else
        hipLaunchKernelGGL(k_gram_tall<64>, dim3(nb), dim3(256), 0, c->stream, A, B, c->n_pad, rpb, part);
    hipLaunchKernelGGL(k_gram_finalize, dim3((elems + 63) / 64), dim3(256), 0, c->stream, part,
                       nb, elems, out_dev);
    AA_CHECK_HIP(hipGetLastError());
    if ((c->world > 1 || c->force_comm) && !local_only) AA_CHECK(comm_allreduce(c, out_dev, elems, 0));
    return AA_OK;
}

// dst[i][j] = alpha[i] * src[i][j] * alpha[j]  (KP x KP; M = D Z'Z D of archetypal_analysis.py:
// 310,330 and the QP Hessian D C K C' D of :387), zero outside k x k
__global__ __launch_bounds__(256) void k_scale_gram(double *__restrict__ dst,
                                                    const double *__restrict__ src,
                                                    const double *__restrict__ alpha, int k, int KP, int kslot = 0)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= KP * KP) return;
    const int i = e / KP, j = e % KP;
    // kslot > 0 (restarts side by side): the diagonal blocks only -- M is block diagonal
    dst[e] = (i < k && j < k && (kslot == 0 || i / kslot == j / kslot)) ? alpha[i] * src[e] * alpha[j] : 0.0;
}

// 0.5 (tr K - 2 tr(D C K Z) + tr(D Z'Z D C K C')) / n   (archetypal_analysis.py:553-556),
// one block, fixed summation tree; state = [Z'Z | C K C' | C K Z]
__device__ __forceinline__ void iter_judge_thread0(int it, double cost0, const double *__restrict__ costs,
                                                   IterState *__restrict__ st, double tol, double mono_tol,
                                                   int criterion, int require, int upd_dict, int upd_w,
                                                   const double *__restrict__ scal, int track_spg);

__global__ __launch_bounds__(256) void k_aa_cost(const double *__restrict__ state,
                                                 const double *__restrict__ alpha, int k, int KP,
                                                 double trace, double n_global,
                                                 double *__restrict__ out,
                                                 int *__restrict__ slot_counter,
                                                 const double *__restrict__ scal, GpnhJudge jd)
{
    __shared__ double sm[256];
    const double *ZtZ = state, *CKCt = state + KP * KP, *CKZ = state + 2 * KP * KP;
    const int t = threadIdx.x;
    double acc = 0.0;
    for (int e = t; e < k * k; e += 256) {
        const int i = e / k, j = e % k;
        acc += alpha[i] * ZtZ[i * KP + j] * alpha[j] * CKCt[j * KP + i];
    }
    if (t < k) acc -= 2.0 * (alpha[t] * CKZ[t * KP + t]);
    sm[t] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) sm[t] += sm[t + o];
        __syncthreads();
    }
    // slot_counter: consecutive calls fill consecutive slots (the launch arguments stay
    // constant, so the call can sit in a captured graph)
    if (t == 0) {
        const int idx = slot_counter ? (*slot_counter)++ : 0;
        out[idx] = 0.5 * (trace + sm[0]) / n_global;
        // the outer iteration's judge rides along with the iteration's last cost (aa_iterate)
        if (jd.on)
            iter_judge_thread0(jd.it, jd.cost0, out, jd.st, jd.tol, jd.mono_tol, jd.criterion, jd.require,
                               jd.upd_dict, jd.upd_w, scal, jd.track_spg);
    }
}

__global__ void k_set_scalars(double *__restrict__ sc, double trace, double fnorm, int R = 1)
{
    if ((int)threadIdx.x < R && blockIdx.x == 0) {      // R > 1: one scalar block per restart slot
        sc[(size_t)threadIdx.x * AA_SC_STRIDE + SC_TRACE] = trace;
        sc[(size_t)threadIdx.x * AA_SC_STRIDE + SC_FNORM] = fnorm;
    }
}

// ---------------------------------------------------------------- driver preprocessing
// flags[c] = 1 when column c of the WEIGHTED field holds a NaN in any row (run_hadisst_aa.py:133,201:
// the mask is taken on weights * da, so a NaN weight or 0 * inf drops the column too).  blockIdx.y
// strides over the rows; a thread that sees a NaN stores 1 (flags are zero-initialised, the stores
// are idempotent: no atomics).
template <typename T>
__global__ __launch_bounds__(256) void k_col_has_nan(const T *__restrict__ raw, long ld, long n_total,
                                                     long p_full, const double *__restrict__ w,
                                                     unsigned char *__restrict__ flags)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= p_full) return;
    const double wc = w ? w[c] : 1.0;
    int bad = 0;
    for (long r = blockIdx.y; r < n_total; r += gridDim.y) {
        const double v = (double)raw[r * ld + c] * wc;
        bad |= (v != v) ? 1 : 0;
    }
    if (bad) flags[c] = (unsigned char)1;
}

// X[r][j] = raw[row0 + r][idx[j]] * w[idx[j]]   (weights * da, then valid_data[:n_training], :133-146,:202-208)
// rows: grid-stride over blockIdx.y (gridDim.y <= 65535 whatever n is)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_gather_weight(const TI *__restrict__ raw, long ld, long row0,
                                                       long n, const int *__restrict__ idx, long p_valid,
                                                       const double *__restrict__ w, TO *__restrict__ X,
                                                       long ldx)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= p_valid) return;
    const int src = idx[j];
    const double wj = w ? w[src] : 1.0;
    for (long r = blockIdx.y; r < n; r += gridDim.y) {
        const double v = (double)raw[(row0 + r) * ld + src] * wj;
        X[r * ldx + j] = (TO)v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_data_to_double(const T *__restrict__ X, long ldx, long n, long p,
                                                        double *__restrict__ out)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    for (long r = blockIdx.y; r < n; r += gridDim.y) out[r * p + j] = (double)X[r * ldx + j];
}

static inline unsigned row_grid(long rows) { return (unsigned)(rows < 4096 ? (rows < 1 ? 1 : rows) : 4096); }

int launch_col_has_nan(Ctx *c, const void *raw_dev, int host_dtype, long ld, long n_total, long p_full,
                       const double *w_dev, unsigned char *flags_dev)
{
    const long xb = (p_full + 255) / 256;
    // enough blocks to fill the chip when the field has few columns, never more than the rows
    long yb = (2048 + xb - 1) / xb;
    if (yb > n_total) yb = n_total;
    const dim3 grid((unsigned)xb, row_grid(yb));
    AA_CHECK_HIP(hipMemsetAsync(flags_dev, 0, (size_t)p_full, c->stream));
    if (host_dtype == AA_F32)
        hipLaunchKernelGGL(k_col_has_nan<float>, grid, dim3(256), 0, c->stream,
                           reinterpret_cast<const float *>(raw_dev), ld, n_total, p_full, w_dev, flags_dev);
    else
        hipLaunchKernelGGL(k_col_has_nan<double>, grid, dim3(256), 0, c->stream,
                           reinterpret_cast<const double *>(raw_dev), ld, n_total, p_full, w_dev, flags_dev);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_gather_weight(Ctx *c, const void *raw_dev, int host_dtype, long ld, long row0, long n,
                         const int *idx_dev, long p_valid, const double *w_dev)
{
    const dim3 grid((unsigned)((p_valid + 255) / 256), row_grid(n));
#define GW(TI, TO)                                                                                  \
    hipLaunchKernelGGL((k_gather_weight<TI, TO>), grid, dim3(256), 0, c->stream,                     \
                       reinterpret_cast<const TI *>(raw_dev), ld, row0, n, idx_dev, p_valid, w_dev,  \
                       c->X.as<TO>(), c->p_pad)
    if (host_dtype == AA_F32) {
        if (c->dtype == AA_F32) GW(float, float); else GW(float, double);
    } else {
        if (c->dtype == AA_F32) GW(double, float); else GW(double, double);
    }
#undef GW
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_data_to_double(Ctx *c, double *out_dev)
{
    const dim3 grid((unsigned)((c->p + 255) / 256), row_grid(c->n));
    if (c->dtype == AA_F32)
        hipLaunchKernelGGL(k_data_to_double<float>, grid, dim3(256), 0, c->stream, c->X.as<float>(), c->p_pad,
                           c->n, c->p, out_dev);
    else
        hipLaunchKernelGGL(k_data_to_double<double>, grid, dim3(256), 0, c->stream, c->X.as<double>(), c->p_pad,
                           c->n, c->p, out_dev);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// ---------------------------------------------------------------- scale factors (delta != 0)
// _update_kernel_aa_scale_factors (archetypal_analysis.py:243-258): spg() (spg.py:46-283) on the
// k-vector alpha with the box projection clip(alpha, 1 - delta, 1 + delta), objective and gradient
// (:220-240) from the k x k Gram state:
//   f(a)  = 0.5 (tr - 2 a . diag(CKZ) + sum_ij a_i a_j ZtZ_ij CKCt_ij) / k     (n_samples = CKZ.shape[1] = k)
//   df(a) = diag(ZtZ diag(a) CKCt - CKZ) / k
// One wave, lane = component; the two k x k coefficient matrices sit in LDS.  Every quirk of the
// generic spg() is kept: f_mem starts as zeros (spg.py:153), sigma_one is an absolute bound
// (:28), alpha0 = None derives the first step from the projected gradient (:178-189).
// Afterwards (archetypal_analysis.py:596-609) the cost with the new scale factors is compared
// with the cost the iteration started from; a violation stops the device loop (stage 3).
__device__ __forceinline__ double sf_wsum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double sf_wmax(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(64) void k_scale_factors_spg(const double *__restrict__ state /*ZtZ|CKCt|CKZ*/,
                                                          double *__restrict__ alpha, int k, int KP,
                                                          double trace, double n_global, double delta_box,
                                                          aa_spg_params sp, int it, double cost0,
                                                          const double *__restrict__ costs,
                                                          const int *__restrict__ slot,
                                                          IterState *__restrict__ st, double mono_tol,
                                                          int require, const double *slot_costs = nullptr,
                                                          int slot_stride = 0, const double *slot_cost0 = nullptr)
{
    extern __shared__ double lds[];               // B1 [k][k+1] | B2 [k][k+1] | xs [64]
    if (slot_costs) {
        // restarts side by side (aa_slots_*): block = slot -- its diagonal blocks of the Gram state, its
        // k scale factors, its cost record, counter and status; the iteration index is the record's
        const int r = blockIdx.x;
        state += (size_t)(r * k) * KP + r * k;
        alpha += r * k;
        costs = slot_costs + (size_t)r * slot_stride;
        slot += r;
        st += r;
        cost0 = slot_cost0[r];
        it = *slot / 2;
    }
    const int i = threadIdx.x, ldb = k + 1;
    double *B1 = lds, *B2 = lds + (size_t)k * ldb, *xs = lds + (size_t)2 * k * ldb;
    const int GS = KP * KP;
    const double *ZtZ = state, *CKCt = state + GS, *CKZ = state + 2 * GS;
    for (int e = i; e < k * k; e += 64) {
        const int r = e / k, q = e % k;
        B1[r * ldb + q] = ZtZ[r * KP + q] * CKCt[r * KP + q];
        B2[r * ldb + q] = ZtZ[r * KP + q] * CKCt[q * KP + r];
    }
    const bool live = i < k;
    const double ci = live ? CKZ[i * KP + i] : 0.0;
    const double lo = 1.0 - delta_box, hi = 1.0 + delta_box, kd = (double)k;
    __syncthreads();
    auto matvec = [&](const double *B, double xv) -> double {     // (B x)_i, x spread over the lanes
        __syncthreads();
        xs[i] = live ? xv : 0.0;
        __syncthreads();
        double s = 0.0;
        if (live)
            for (int q = 0; q < k; ++q) s = fma(B[i * ldb + q], xs[q], s);
        return s;
    };
    auto fval = [&](double xv) -> double {
        const double m = matvec(B1, xv);
        return 0.5 * (trace - 2.0 * sf_wsum(live ? xv * ci : 0.0) + sf_wsum(live ? xv * m : 0.0)) / kd;
    };
    auto grad = [&](double xv) -> double {
        const double m = matvec(B2, xv);                            // every lane joins the barriers
        return live ? (m - ci) / kd : 0.0;
    };
    auto proj = [&](double v) -> double { return fmin(fmax(lo, v), hi); };

    double x = live ? proj(alpha[i]) : 0.0;                         // spg.py:147-148
    double a_step = sp.alpha0;                                      // < 0: None
    bool a_set = sp.alpha0 >= 0.0;
    const int mem = sp.memory < 1 ? 1 : (sp.memory > 16 ? 16 : sp.memory);
    double fmem[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) fmem[q] = 0.0;
    double f_old = fval(x);
    int n_feval = 1, flags = 0;
    bool converged = false;
    double g = grad(x);
    int n_iter = 0;
    for (n_iter = 0; n_iter < sp.max_iterations; ++n_iter) {
        if (!a_set) {                                               // spg.py:178-189
            const double ainv = sf_wmax(live ? fabs(proj(x - g) - x) : 0.0);
            a_step = fabs(ainv) > 1e-12 ? 1.0 / ainv : 1.0;
            a_set = true;
        }
        const double d = live ? proj(x - a_step * g) - x : 0.0;     // :191-194
#pragma unroll
        for (int q = 15; q > 0; --q)
            if (q < mem) fmem[q] = fmem[q - 1];
        fmem[0] = f_old;
        double f_max = fmem[0];
#pragma unroll
        for (int q = 1; q < 16; ++q)
            if (q < mem && fmem[q] >= f_max) f_max = fmem[q];
        const double dlt = sf_wsum(d * g);
        double lam = 1.0;
        double xn = x + d;
        double f_new = fval(xn);
        n_feval += 1;
        int guard = 0;
        while (f_new > f_max + sp.gamma * lam * dlt && guard < 200) {           // :214-229
            const double tmp = -0.5 * lam * lam * dlt / (f_new - f_old - lam * dlt);
            lam = (sp.sigma_one <= tmp && tmp <= sp.sigma_two * lam) ? tmp : 0.5 * lam;
            xn = x + lam * d;
            f_new = fval(xn);
            n_feval += 1;
            ++guard;
            if (fabs(lam) < sp.lambda_min) {
                flags |= AA_SPG_FLAG_LAMBDA_MIN;
                break;
            }
        }
        const double gn = grad(xn);                                 // :232-244
        const double sksk = lam * lam * sf_wsum(d * d);
        const double beta = lam * sf_wsum(d * (gn - g));
        a_step = beta <= 0.0 ? sp.alpha_max : fmin(sp.alpha_max, fmax(sp.alpha_min, sksk / beta));
        x = xn;
        g = gn;
        f_old = fval(x);
        n_feval += 1;
        const double res = live ? proj(x - g) - x : 0.0;            // :246-266
        const double rn = sqrt(sf_wsum(res * res));
        converged = rn < sp.epsilon_two;
        if (sp.use_infinity_norm) converged = converged || sf_wmax(fabs(res)) < sp.epsilon_one;
        if (converged) break;
        if (n_feval > sp.max_feval) {
            flags |= AA_SPG_FLAG_MAX_FEVAL;
            break;
        }
    }
    if (!converged && !(flags & AA_SPG_FLAG_MAX_FEVAL)) flags |= AA_SPG_FLAG_MAX_ITER;   // :278-281
    if (slot_costs ? live : i < KP) alpha[i] = live ? x : 0.0;
    // cost with the new scale factors (archetypal_analysis.py:601-609); padding lanes carry 0
    const double m2 = matvec(B2, x);
    const double cost = 0.5 * (trace - 2.0 * sf_wsum(live ? x * ci : 0.0) + sf_wsum(live ? x * m2 : 0.0)) / n_global;
    if (i == 0 && st) {
        st->spg_flags |= flags;
        if (!st->stop) {
            const int s0 = slot ? *slot : 0;
            const double old = s0 == 0 ? cost0 : costs[s0 - 1];
            if (require && cost > old && fabs(cost - old) > mono_tol) {
                st->stop = 1;
                st->error_stage = 3;
                st->stop_iter = it;
            }
        }
    }
}

int launch_scale_factors(Ctx *c, const aa_spg_params *sp, double delta_box, int it, double cost0,
                         const double *costs, const int *slot, IterState *st, double mono_tol, int require)
{
    if (c->slots_aa) {
        const int k = c->slots_k;
        const size_t lds_s = ((size_t)2 * k * (k + 1) + 64) * sizeof(double);
        hipLaunchKernelGGL(k_scale_factors_spg, dim3((unsigned)c->slots_R), dim3(64), lds_s, c->stream,
                           (const double *)c->gramState.as<double>(), c->alphaDev.as<double>(), k, c->KP,
                           c->trace, (double)c->n_global, delta_box, *sp, 0, 0.0, (const double *)nullptr,
                           (const int *)c->slotCounters.as<int>(), c->slotStates.as<IterState>(), mono_tol, require,
                           (const double *)c->slotCosts.as<double>(), c->slots_stride,
                           (const double *)c->slotCost0.as<double>());
        AA_CHECK_HIP(hipGetLastError());
        return AA_OK;
    }
    const size_t lds = ((size_t)2 * c->k * (c->k + 1) + 64) * sizeof(double);
    hipLaunchKernelGGL(k_scale_factors_spg, dim3(1), dim3(64), lds, c->stream,
                       (const double *)c->gramState.as<double>(), c->alphaDev.as<double>(), c->k, c->KP,
                       c->trace, (double)c->n_global, delta_box, *sp, it, cost0, costs, slot, st, mono_tol,
                       require);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// ---------------------------------------------------------------- device-side loop control
// Status record of aa_iterate (device memory): written by one thread after every outer
// iteration, read by the host once per batch.
// costs[2*it], costs[2*it + 1]: cost after the dictionary / weights update of iteration `it`.
__device__ __forceinline__ void iter_judge_thread0(int it, double cost0, const double *__restrict__ costs,
                                                   IterState *__restrict__ st, double tol, double mono_tol,
                                                   int criterion, int require, int upd_dict, int upd_w,
                                                   const double *__restrict__ scal, int track_spg)
{
    if (st->stop) return;
    const double old = it == 0 ? cost0 : costs[2 * it - 1];
    const double c1 = costs[2 * it], c2 = costs[2 * it + 1];
    if (upd_dict && track_spg) {
        // the SPG loop ends converged, at the function-evaluation cap, or at the iteration cap
        // (the warning of spg.py:278-281)
        int fl = (int)scal[SC_FLAGS];
        if (!(fl & (AA_SPG_FLAG_CONVERGED | AA_SPG_FLAG_MAX_FEVAL))) fl |= AA_SPG_FLAG_MAX_ITER;
        st->spg_flags |= fl & ~AA_SPG_FLAG_CONVERGED;
    }
    // archetypal_analysis.py:167-174, evaluated after each update against the cost the
    // iteration started from
    if (upd_dict && require && c1 > old && fabs(c1 - old) > mono_tol) {
        st->stop = 1;
        st->error_stage = 1;
        st->stop_iter = it;
        return;
    }
    if (upd_w && require && c2 > old && fabs(c2 - old) > mono_tol) {
        st->stop = 1;
        st->error_stage = 2;
        st->stop_iter = it;
        return;
    }
    // :177-197
    bool conv;
    if (criterion == 0) conv = fabs(c2 - old) < tol;
    else conv = fabs((c2 - old) / fmax(fabs(c2), fabs(old))) < tol;
    st->last_iter = it;
    if (conv) {
        st->stop = 1;
        st->converged = 1;
        st->stop_iter = it;
    }
}

__global__ void k_iter_judge(int it, double cost0, const double *__restrict__ costs,
                             IterState *__restrict__ st, double tol, double mono_tol, int criterion,
                             int require, int upd_dict, int upd_w, const double *__restrict__ scal,
                             int track_spg)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    iter_judge_thread0(it, cost0, costs, st, tol, mono_tol, criterion, require, upd_dict, upd_w, scal, track_spg);
}

// keeps the factors of the iteration at which the loop stopped (later iterations of the same
// batch overwrite Ct / Zt); every block reads one word and leaves unless this is that iteration
__global__ __launch_bounds__(256) void k_iter_snapshot(int it, const IterState *__restrict__ st,
                                                       const double *__restrict__ Ct,
                                                       const double *__restrict__ Zt,
                                                       double *__restrict__ snapC,
                                                       double *__restrict__ snapZ, long elems,
                                                       const double *__restrict__ alpha,
                                                       double *__restrict__ snapAlpha, int KP)
{
    if (!st->stop || st->stop_iter != it) return;
    if (blockIdx.x == 0 && (int)threadIdx.x < KP) snapAlpha[threadIdx.x] = alpha[threadIdx.x];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < elems; i += (long)gridDim.x * 256) {
        snapC[i] = Ct[i];
        snapZ[i] = Zt[i];
    }
}

// costs[slot] = the previous entry (an update that is switched off leaves the cost unchanged)
__global__ void k_cost_carry(double *__restrict__ costs, int *__restrict__ slot, double cost0)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int idx = (*slot)++;
    costs[idx] = idx == 0 ? cost0 : costs[idx - 1];
}

int launch_iter_judge(Ctx *c, int it, double cost0, const double *costs, IterState *st,
                      const aa_iter_params *ip, bool judged)
{
    if (!judged)                                   // else: done by the cost kernel before it
        hipLaunchKernelGGL(k_iter_judge, dim3(1), dim3(64), 0, c->stream, it, cost0, costs, st, ip->tolerance,
                           ip->mono_tolerance, ip->criterion, ip->require_monotonic, ip->update_dictionary,
                           ip->update_weights, (const double *)c->scalars.as<double>(), 1);
    hipLaunchKernelGGL(k_iter_snapshot, dim3(512), dim3(256), 0, c->stream, it, (const IterState *)st,
                       (const double *)c->Ct.as<double>(), (const double *)c->Zt.as<double>(),
                       c->snapC.as<double>(), c->snapZ.as<double>(), (long)c->n_pad * c->KP,
                       (const double *)c->alphaDev.as<double>(), c->snapAlpha.as<double>(), c->KP);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_cost_carry(Ctx *c, double *costs, int *slot, double cost0)
{
    hipLaunchKernelGGL(k_cost_carry, dim3(1), dim3(64), 0, c->stream, costs, slot, cost0);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// ---------------------------------------------------------------- GPNH dictionary update
// (Z'Z/n + lambda GW) W' = Z'X/n  (gpnh_convex_coding.py:213-226, GW of :296-300): every block
// factorises the k x k system (Cholesky, in LDS) and solves it for its own 256 columns of the
// right-hand side, one column per thread.  ok[0] = 1 when a pivot is not safely positive.
// KM >= k: the substitution loops are unrolled over KM with `i < k` predicates, so the column's
// solution vector stays in registers (indexed by a run-time k it lived in scratch: 25 us a launch)
template <int KM>
__global__ __launch_bounds__(256) void k_gpnh_solve(const double *__restrict__ ZtZ /*[KP][KP]*/,
                                                    const double *__restrict__ ZtX /*[KP][ld]*/,
                                                    int ld, int p, int k, int KP, double n_samples,
                                                    double lambda, double *__restrict__ Wt,
                                                    float *__restrict__ WtF, int *__restrict__ ok)
{
    extern __shared__ double L[];                 // k x k, row-major, lower triangle
    __shared__ double dmax_s;
    __shared__ int bad;
    const int t = threadIdx.x;
    const double pref = k > 1 ? 4.0 / ((double)p * k * (k - 1)) : 0.0;
    for (int e = t; e < k * k; e += 256) {
        const int i = e / k, j = e % k;
        L[e] = ZtZ[i * KP + j] / n_samples + lambda * pref * ((i == j ? (double)k : 0.0) - 1.0);
    }
    if (t == 0) bad = 0;
    __syncthreads();
    if (t == 0) {
        double m = 0.0;
        for (int i = 0; i < k; ++i) m = fmax(m, fabs(L[i * k + i]));
        dmax_s = m;
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {                 // right-looking Cholesky
        if (t == 0) {
            const double d = L[j * k + j];
            if (!(d > 1e-13 * dmax_s)) bad = 1;
            L[j * k + j] = sqrt(d > 0.0 ? d : 1.0);
        }
        __syncthreads();
        const double piv = L[j * k + j];
        for (int i = j + 1 + t; i < k; i += 256) L[i * k + j] /= piv;
        __syncthreads();
        for (int e = t; e < (k - j - 1) * (k - j - 1); e += 256) {
            const int i = j + 1 + e / (k - j - 1), q = j + 1 + e % (k - j - 1);
            if (q <= i) L[i * k + q] -= L[i * k + j] * L[q * k + j];
        }
        __syncthreads();
    }
    if (bad) {
        if (t == 0 && blockIdx.x == 0) ok[0] = 1;     // flag: not positive definite
        return;
    }
    const int c = blockIdx.x * 256 + t;
    if (c >= ld) return;
    double y[KM];
#pragma unroll
    for (int i = 0; i < KM; ++i) {                // L y = b
        if (i < k) {
            double v = c < p ? ZtX[(long)i * ld + c] / n_samples : 0.0;
#pragma unroll
            for (int q = 0; q < i; ++q) v -= L[i * k + q] * y[q];
            y[i] = v / L[i * k + i];
        } else {
            y[i] = 0.0;
        }
    }
#pragma unroll
    for (int i = KM - 1; i >= 0; --i) {           // L' w = y
        if (i < k) {
            double v = y[i];
#pragma unroll
            for (int q = i + 1; q < KM; ++q)
                if (q < k) v -= L[q * k + i] * y[q];
            y[i] = v / L[i * k + i];
        }
    }
#pragma unroll
    for (int i = 0; i < KM; ++i) {
        if (i < KP) {
            Wt[(long)i * ld + c] = y[i];
            if (WtF) WtF[(long)i * ld + c] = (float)y[i];
        }
    }
    for (int i = KM; i < KP; ++i) {               // padding rows of the k x p factor stay zero
        Wt[(long)i * ld + c] = 0.0;
        if (WtF) WtF[(long)i * ld + c] = 0.f;
    }
}

// cost of gpnh_convex_coding.py:317-330 from the device-side pieces: tr(W'X'Z) in scal[slot],
// Z'Z and W'W (KP x KP), the GPNH penalty (:179-196) from the Gram of the dictionary
//   phi(W) = 2 / (k p (k - 1)) sum_{i<j} ||w_i - w_j||^2,  ||w_i - w_j||^2 = G_ii + G_jj - 2 G_ij
// ZtX / Wt given (the device loop): tr(W'X'Z) = <Z'X, W'> is summed here from the two k x p
// arrays -- 1670 terms on the C3 shape -- instead of <XW, Z> over n rows in two launches of their own.
__global__ __launch_bounds__(256) void k_gpnh_cost(const double *__restrict__ ZtZ,
                                                   const double *WtW /* = WtW_out when that is set */,
                                                   const double *__restrict__ scal, int slot, int k,
                                                   int KP, int p, double trace, double n_samples,
                                                   double lambda, double *__restrict__ out,
                                                   int *__restrict__ slot_counter,
                                                   const double *__restrict__ ZtX,
                                                   const double *__restrict__ Wt, int ld,
                                                   double *WtW_out, GpnhJudge jd)
{
    __shared__ double sm[256], sm2[256], sm3[256];
    const int t = threadIdx.x;
    if (WtW_out) {
        // W'W (k x k <= 256 outputs, k * ld <= 4096: the launcher checks) right here instead of two
        // launches of the wide Gram kernel: W' goes through LDS (coalesced in, row stride ld + 1
        // out), one output per thread; zero padding as k_gram_finalize leaves it
        __shared__ double wl[4096 + 64];
        for (int e = t; e < k * ld; e += 256) wl[(e / ld) * (ld + 1) + e % ld] = Wt[e];
        __syncthreads();
        // `parts` threads per output, each over its share of the columns, combined in a fixed order
        const int kk = k * k, parts = 256 / kk >= 4 ? 4 : (256 / kk >= 2 ? 2 : 1);
        const int span = ld / parts;                   // ld is a multiple of 128
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
        if (t < kk * parts) {
            const int e = t % kk, part = t / kk;
            const double *wi = wl + (e / k) * (ld + 1) + part * span, *wj = wl + (e % k) * (ld + 1) + part * span;
            for (int c = 0; c < span; c += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a4[u] = fma(wi[c + u], wj[c + u], a4[u]);
            }
        }
        sm[t] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        __syncthreads();
        for (int e = t; e < KP * KP; e += 256) {
            const int i = e / KP, j = e % KP;
            double v = 0.0;
            if (i < k && j < k)
                for (int q = 0; q < parts; ++q) v += sm[q * kk + i * k + j];
            WtW_out[e] = v;
        }
        __syncthreads();
    }
    double cross = 0.0;
    if (ZtX) {
        double c4[4] = {0.0, 0.0, 0.0, 0.0};            // four chains, fixed order
        const long total = (long)k * ld;                // padding columns are zero in both
        long e = t;
        for (; e + 3 * 256 < total; e += 4 * 256) {
#pragma unroll
            for (int u = 0; u < 4; ++u) c4[u] = fma(ZtX[e + u * 256], Wt[e + u * 256], c4[u]);
        }
        for (; e < total; e += 256) c4[0] = fma(ZtX[e], Wt[e], c4[0]);
        cross = (c4[0] + c4[1]) + (c4[2] + c4[3]);
    }
    sm3[t] = cross;
    double quad = 0.0, pen = 0.0;
    for (int e = t; e < k * k; e += 256) {
        const int i = e / k, j = e % k;
        quad += ZtZ[i * KP + j] * WtW[j * KP + i];
        if (j > i) pen += WtW[i * KP + i] + WtW[j * KP + j] - 2.0 * WtW[i * KP + j];
    }
    sm[t] = quad;
    sm2[t] = pen;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            sm[t] += sm[t + o];
            sm2[t] += sm2[t + o];
            sm3[t] += sm3[t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        double penalty = 0.0;
        if (lambda != 0.0 && k > 1) penalty = lambda * (2.0 / ((double)k * p * (k - 1.0))) * sm2[0];
        const int idx = slot_counter ? (*slot_counter)++ : 0;
        const double s1 = ZtX ? sm3[0] : scal[slot];
        out[idx] = 0.5 * (trace - 2.0 * s1 + sm[0]) / n_samples + penalty;
        // the outer iteration's judge (monotonicity, stopping rule) rides along with its last cost
        if (jd.on)
            iter_judge_thread0(jd.it, jd.cost0, out, jd.st, jd.tol, jd.mono_tol, jd.criterion, jd.require,
                               jd.upd_dict, jd.upd_w, scal, 0);
    }
}

__global__ __launch_bounds__(256) void k_copy2(const IterState *__restrict__ st, int it,
                                               const double *__restrict__ a, double *__restrict__ sa,
                                               long na, const double *__restrict__ b,
                                               double *__restrict__ sb, long nb)
{
    if (st && (!st->stop || st->stop_iter != it)) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na; i += (long)gridDim.x * 256) sa[i] = a[i];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (long)gridDim.x * 256) sb[i] = b[i];
}

// ---------------------------------------------------------------- GPNH, R restarts side by side
// (SURVEY 8(f1)).  The drivers fit the same matrix n_init times (bin/run_jra55_pca_gpnh.py:112-138);
// one GPNH iteration on the C3 shape is 11 dependent launches of a few microseconds, which several
// streams do not overlap.  The tall and wide arrays have KP = 32 component
// slots of which a k = 10 fit uses ten, so R = floor(KP / k) restarts sit SIDE BY SIDE in one set
// of arrays: Z (n x KP) holds restart r in columns [r k, (r + 1) k), W' (KP x p) in the same rows.
// The passes over X (X W, Z'X), Z'Z and W'W are the kernels of the single fit, unchanged -- an
// output element of theirs depends on its own column / row only, so every restart gets the bits
// it gets alone -- and the steps that couple components (the k x k solve, the cost, the QP, the
// judge, the snapshot) read the diagonal blocks only, restart by restart, with the arithmetic of
// the single-fit kernels.  Every slot has its own cost record, slot counter and status record, and
// takes its iteration index from its slot counter: slots start and stop independently (a stopped
// slot keeps iterating until the host replaces it; its factors were saved at the stopping
// iteration).
struct GpnhSlots {
    int R, k;
    double *costs;        // [R][stride]
    int stride;
    int *counters;        // [R]
    IterState *st;        // [R]
    double *cost0;        // [R]
    int max_outer;
};

template <int KM>
__global__ __launch_bounds__(256) void k_gpnh_solve_slots(const double *__restrict__ ZtZ /*[KP][KP]*/,
                                                          const double *__restrict__ ZtX /*[KP][ld]*/,
                                                          int ld, int p, int KP, double n_samples,
                                                          double lambda, double *__restrict__ Wt,
                                                          float *__restrict__ WtF, GpnhSlots sl)
{
    extern __shared__ double L[];                 // k x k, row-major, lower triangle
    __shared__ double dmax_s;
    __shared__ int bad;
    const int t = threadIdx.x, k = sl.k;
    const double pref = k > 1 ? 4.0 / ((double)p * k * (k - 1)) : 0.0;
    const int c = blockIdx.x * 256 + t;
    {
        const int r = blockIdx.y;                     // one slot per block row
        const int o = r * k;
        for (int e = t; e < k * k; e += 256) {
            const int i = e / k, j = e % k;
            L[e] = ZtZ[(o + i) * KP + o + j] / n_samples + lambda * pref * ((i == j ? (double)k : 0.0) - 1.0);
        }
        if (t == 0) bad = 0;
        __syncthreads();
        if (t == 0) {
            double m = 0.0;
            for (int i = 0; i < k; ++i) m = fmax(m, fabs(L[i * k + i]));
            dmax_s = m;
        }
        __syncthreads();
        for (int j = 0; j < k; ++j) {                 // right-looking Cholesky
            if (t == 0) {
                const double d = L[j * k + j];
                if (!(d > 1e-13 * dmax_s)) bad = 1;
                L[j * k + j] = sqrt(d > 0.0 ? d : 1.0);
            }
            __syncthreads();
            const double piv = L[j * k + j];
            for (int i = j + 1 + t; i < k; i += 256) L[i * k + j] /= piv;
            __syncthreads();
            for (int e = t; e < (k - j - 1) * (k - j - 1); e += 256) {
                const int i = j + 1 + e / (k - j - 1), q = j + 1 + e % (k - j - 1);
                if (q <= i) L[i * k + q] -= L[i * k + j] * L[q * k + j];
            }
            __syncthreads();
        }
        if (bad) {                                    // this slot only: its dictionary stays as it is
            if (t == 0 && blockIdx.x == 0) sl.st[r].pad0 = 1;
            return;
        }
        if (c >= ld) return;
        double y[KM];
#pragma unroll
        for (int i = 0; i < KM; ++i) {                // L y = b
            if (i < k) {
                double v = c < p ? ZtX[(long)(o + i) * ld + c] / n_samples : 0.0;
#pragma unroll
                for (int q = 0; q < i; ++q) v -= L[i * k + q] * y[q];
                y[i] = v / L[i * k + i];
            } else {
                y[i] = 0.0;
            }
        }
#pragma unroll
        for (int i = KM - 1; i >= 0; --i) {           // L' w = y
            if (i < k) {
                double v = y[i];
#pragma unroll
                for (int q = i + 1; q < KM; ++q)
                    if (q < k) v -= L[q * k + i] * y[q];
                y[i] = v / L[i * k + i];
            }
        }
#pragma unroll
        for (int i = 0; i < KM; ++i) {
            if (i < k) {
                Wt[(long)(o + i) * ld + c] = y[i];
                if (WtF) WtF[(long)(o + i) * ld + c] = (float)y[i];
            }
        }
    }
    // (component slots beyond R k are zero since aa_gpnh_slots_begin and are never written)
}

// cost of every slot in `mask` (k_gpnh_cost's arithmetic on the slot's diagonal blocks and rows):
//   what = 0: initial cost of a freshly loaded slot -> cost0[r]
//   what = 1: cost after the dictionary update -> the slot's record
//   what = 2: cost after the weights update -> the record, then the slot's judge
// form_gram: W'W of the slot is formed here first (what the single fit does after a dictionary update
// when the factor is small enough, gpnh_cost_can_gram; otherwise, and for the initial cost, the wide
// Gram kernel has left it in WtW)
__global__ __launch_bounds__(256) void k_gpnh_cost_slots(const double *__restrict__ ZtZ, double *WtW,
                                                         int KP, int p, double trace, double n_samples,
                                                         double lambda, const double *__restrict__ ZtX,
                                                         const double *__restrict__ Wt, int ld, GpnhSlots sl,
                                                         unsigned mask, int what, double tol, double mono_tol,
                                                         int criterion, int require, int form_gram)
{
    __shared__ double sm[256], sm2[256], sm3[256];
    __shared__ double wl[4096 + 64];
    const int t = threadIdx.x, k = sl.k;
    {
        const int r = blockIdx.x;                     // one slot per block
        if (!((mask >> r) & 1u)) return;
        const int o = r * k;
        const double *Wr = Wt + (long)o * ld, *Xr = ZtX + (long)o * ld;
        if (form_gram) {
            // W'W of this slot (k x k), k_gpnh_cost's in-kernel Gram: W' through LDS, `parts` threads
            // per output, combined in a fixed order
            for (int e = t; e < k * ld; e += 256) wl[(e / ld) * (ld + 1) + e % ld] = Wr[e];
            __syncthreads();
            const int kk = k * k, parts = 256 / kk >= 4 ? 4 : (256 / kk >= 2 ? 2 : 1);
            const int span = ld / parts;
            double a4[4] = {0.0, 0.0, 0.0, 0.0};
            if (t < kk * parts) {
                const int e = t % kk, part = t / kk;
                const double *wi = wl + (e / k) * (ld + 1) + part * span, *wj = wl + (e % k) * (ld + 1) + part * span;
                for (int cc = 0; cc < span; cc += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) a4[u] = fma(wi[cc + u], wj[cc + u], a4[u]);
                }
            }
            sm[t] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
            __syncthreads();
            for (int e = t; e < k * k; e += 256) {
                const int i = e / k, j = e % k;
                double v = 0.0;
                for (int q = 0; q < parts; ++q) v += sm[q * kk + i * k + j];
                WtW[(o + i) * KP + o + j] = v;
            }
            __syncthreads();
        }
        double cross;
        {
            double c4[4] = {0.0, 0.0, 0.0, 0.0};            // four chains, fixed order
            const long total = (long)k * ld;                // padding columns are zero in both
            long e = t;
            for (; e + 3 * 256 < total; e += 4 * 256) {
#pragma unroll
                for (int u = 0; u < 4; ++u) c4[u] = fma(Xr[e + u * 256], Wr[e + u * 256], c4[u]);
            }
            for (; e < total; e += 256) c4[0] = fma(Xr[e], Wr[e], c4[0]);
            cross = (c4[0] + c4[1]) + (c4[2] + c4[3]);
        }
        sm3[t] = cross;
        double quad = 0.0, pen = 0.0;
        for (int e = t; e < k * k; e += 256) {
            const int i = e / k, j = e % k;
            quad += ZtZ[(o + i) * KP + o + j] * WtW[(o + j) * KP + o + i];
            if (j > i) pen += WtW[(o + i) * KP + o + i] + WtW[(o + j) * KP + o + j] - 2.0 * WtW[(o + i) * KP + o + j];
        }
        sm[t] = quad;
        sm2[t] = pen;
        __syncthreads();
        for (int q = 128; q > 0; q >>= 1) {
            if (t < q) {
                sm[t] += sm[t + q];
                sm2[t] += sm2[t + q];
                sm3[t] += sm3[t + q];
            }
            __syncthreads();
        }
        if (t == 0) {
            double penalty = 0.0;
            if (lambda != 0.0 && k > 1) penalty = lambda * (2.0 / ((double)k * p * (k - 1.0))) * sm2[0];
            const double cost = 0.5 * (trace - 2.0 * sm3[0] + sm[0]) / n_samples + penalty;
            if (what == 0) {
                sl.cost0[r] = cost;
            } else {
                double *rec = sl.costs + (size_t)r * sl.stride;
                int idx = sl.counters[r];
                if (idx >= sl.stride) idx = sl.stride - 1;        // a finished slot waiting to be replaced
                rec[idx] = cost;
                sl.counters[r] = idx + 1;
                if (what == 2) {
                    const int it = idx / 2;
                    iter_judge_thread0(it, sl.cost0[r], rec, &sl.st[r], tol, mono_tol, criterion, require, 1, 1,
                                       nullptr, 0);
                    // the iteration cap ends a slot like the stopping rule does (not converged)
                    if (!sl.st[r].stop && it + 1 >= sl.max_outer) {
                        sl.st[r].stop = 1;
                        sl.st[r].stop_iter = it;
                    }
                }
            }
        }
    }
}

// the factors of every slot that has just stopped (status written by the judge above, in this
// iteration): its columns of Z and its rows of W' into the snapshot arrays
__global__ __launch_bounds__(256) void k_gpnh_snap_slots(const double *__restrict__ Zt, double *__restrict__ snapZ,
                                                         long n_pad, int KP, const double *__restrict__ Wt,
                                                         double *__restrict__ snapW, int ld, GpnhSlots sl)
{
    for (int r = 0; r < sl.R; ++r) {
        const IterState st = sl.st[r];
        const int it = sl.counters[r] / 2 - 1;
        if (!st.stop || st.stop_iter != it) continue;    // (a full record stops counting at an index no stop has)
        const int o = r * sl.k, k = sl.k;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n_pad * k; e += (long)gridDim.x * 256) {
            const long row = e / k;
            const int i = o + (int)(e % k);
            snapZ[row * KP + i] = Zt[row * KP + i];
        }
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)k * ld; e += (long)gridDim.x * 256)
            snapW[(long)o * ld + e] = Wt[(long)o * ld + e];
    }
}
int launch_gpnh_solve(Ctx *c, double lambda, int *ok_dev)
{
    const size_t lds = (size_t)c->k * c->k * sizeof(double);
#define GPS(KMV)                                                                                        \
    hipLaunchKernelGGL(k_gpnh_solve<KMV>, dim3((unsigned)((c->p_pad + 255) / 256)), dim3(256), lds, c->stream, \
                       (const double *)c->gramState.as<double>(), (const double *)c->ZtX.as<double>(),      \
                       (int)c->p_pad, (int)c->p, c->k, c->KP, (double)c->n_global, lambda, c->P.as<double>(), \
                       c->dtype == AA_F32 ? c->Pw.as<float>() : (float *)nullptr, ok_dev)
    if (c->k <= 16) GPS(16);
    else if (c->k <= 32) GPS(32);
    else GPS(64);
#undef GPS
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// k x k Gram of the dictionary inside the cost kernel: one output per thread, short rows
bool gpnh_cost_can_gram(const Ctx *c) { return c->k * c->k <= 256 && (long)c->k * c->p_pad <= 4096; }

int launch_gpnh_cost(Ctx *c, double lambda, double *out_dev, int *slot_counter, bool from_wide, bool gram_w,
                     const GpnhJudge *judge)
{
    double *gs = c->gramState.as<double>();
    GpnhJudge jd;
    memset(&jd, 0, sizeof(jd));
    if (judge) jd = *judge;
    hipLaunchKernelGGL(k_gpnh_cost, dim3(1), dim3(256), 0, c->stream, (const double *)gs,
                       (const double *)(gs + (size_t)c->KP * c->KP),
                       (const double *)c->scalars.as<double>(), (int)SC_S1, c->k, c->KP, (int)c->p, c->trace,
                       (double)c->n_global, lambda, out_dev, slot_counter,
                       from_wide ? (const double *)c->ZtX.as<double>() : (const double *)nullptr,
                       (const double *)c->P.as<double>(), (int)c->p_pad,
                       gram_w ? gs + (size_t)c->KP * c->KP : (double *)nullptr, jd);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// GPNH flavour of launch_iter_judge: the snapshot keeps Z and the dictionary (W', wide)
int launch_gpnh_judge(Ctx *c, int it, double cost0, const double *costs, IterState *st,
                      const aa_iter_params *ip, bool judged)
{
    if (!judged)                                   // else: done by the cost kernel before it
        hipLaunchKernelGGL(k_iter_judge, dim3(1), dim3(64), 0, c->stream, it, cost0, costs, st, ip->tolerance,
                           ip->mono_tolerance, ip->criterion, ip->require_monotonic, ip->update_dictionary,
                           ip->update_weights, (const double *)c->scalars.as<double>(), 0 /* no SPG behind it */);
    hipLaunchKernelGGL(k_copy2, dim3(256), dim3(256), 0, c->stream, (const IterState *)st, it,
                       (const double *)c->Zt.as<double>(), c->snapZ.as<double>(), (long)c->n_pad * c->KP,
                       (const double *)c->P.as<double>(), c->snapC.as<double>(), (long)c->KP * c->p_pad);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// AA restarts side by side: cost of slot blockIdx.x from the diagonal blocks of the Gram state
// (k_aa_cost's arithmetic); what = 0: initial cost -> cost0[r]; what = 2: the cost after the weights
// update -> the slot's record, then its judge (the iteration index is the record's)
__global__ __launch_bounds__(256) void k_aa_cost_slots(const double *__restrict__ state,
                                                       const double *__restrict__ alpha, int KP, double trace,
                                                       double n_global, const double *__restrict__ scal,
                                                       GpnhSlots sl, int what, double tol, double mono_tol,
                                                       int criterion, int require)
{
    __shared__ double sm[256];
    const int t = threadIdx.x, r = blockIdx.x, k = sl.k, o = r * k;
    const size_t off = (size_t)o * KP + o;
    const double *ZtZ = state + off, *CKCt = state + KP * KP + off, *CKZ = state + 2 * KP * KP + off;
    const double *al = alpha + o;
    double acc = 0.0;
    for (int e = t; e < k * k; e += 256) {
        const int i = e / k, j = e % k;
        acc += al[i] * ZtZ[i * KP + j] * al[j] * CKCt[j * KP + i];
    }
    if (t < k) acc -= 2.0 * (al[t] * CKZ[t * KP + t]);
    sm[t] = acc;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (t < q) sm[t] += sm[t + q];
        __syncthreads();
    }
    if (t == 0) {
        const double cost = 0.5 * (trace + sm[0]) / n_global;
        if (what == 0) {
            sl.cost0[r] = cost;
        } else {
            double *rec = sl.costs + (size_t)r * sl.stride;
            int idx = sl.counters[r];
            if (idx >= sl.stride) idx = sl.stride - 1;
            rec[idx] = cost;
            sl.counters[r] = idx + 1;
            const int it = idx / 2;
            iter_judge_thread0(it, sl.cost0[r], rec, &sl.st[r], tol, mono_tol, criterion, require, 1, 1,
                               scal + (size_t)r * AA_SC_STRIDE, 1);
            if (!sl.st[r].stop && it + 1 >= sl.max_outer) {
                sl.st[r].stop = 1;
                sl.st[r].stop_iter = it;
            }
        }
    }
}

// the factors of every slot that has just stopped: its columns of C' and Z (both n x KP)
__global__ __launch_bounds__(256) void k_aa_snap_slots(const double *__restrict__ Ct, double *__restrict__ snapC,
                                                       const double *__restrict__ Zt, double *__restrict__ snapZ,
                                                       long n_pad, int KP, const double *__restrict__ P,
                                                       double *__restrict__ snapP, int ld, GpnhSlots sl,
                                                       const double *__restrict__ alpha,
                                                       double *__restrict__ snapAlpha)
{
    for (int r = 0; r < sl.R; ++r) {
        const IterState st = sl.st[r];
        const int it = sl.counters[r] / 2 - 1;
        if (!st.stop || st.stop_iter != it) continue;
        const int o = r * sl.k, k = sl.k;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n_pad * k; e += (long)gridDim.x * 256) {
            const long row = e / k;
            const int i = o + (int)(e % k);
            snapC[row * KP + i] = Ct[row * KP + i];
            snapZ[row * KP + i] = Zt[row * KP + i];
        }
        if (blockIdx.x == 0 && (int)threadIdx.x < k) snapAlpha[o + threadIdx.x] = alpha[o + threadIdx.x];
        // C X as the loop carries it (P + lambda Q, update after update): what aa_get_archetypes returns
        // when the loop stops on the last iteration of a batch (no restore, no recomputation)
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)k * ld; e += (long)gridDim.x * 256)
            snapP[(long)o * ld + e] = P[(long)o * ld + e];
    }
}

// ---- launchers of the slot kernels (solver.hip: aa_gpnh_slots_*)
static GpnhSlots slots_of(Ctx *c)
{
    GpnhSlots sl;
    sl.R = c->slots_R;
    sl.k = c->slots_k;
    sl.costs = c->slotCosts.as<double>();
    sl.stride = c->slots_stride;
    sl.counters = c->slotCounters.as<int>();
    sl.st = c->slotStates.as<IterState>();
    sl.cost0 = c->slotCost0.as<double>();
    sl.max_outer = c->slots_max_outer;
    return sl;
}

int launch_aa_cost_slots(Ctx *c, int what, const aa_iter_params *ip)
{
    hipLaunchKernelGGL(k_aa_cost_slots, dim3((unsigned)c->slots_R), dim3(256), 0, c->stream,
                       (const double *)c->gramState.as<double>(), (const double *)c->alphaDev.as<double>(), c->KP,
                       c->trace, (double)c->n_global, (const double *)c->scalars.as<double>(), slots_of(c), what,
                       ip ? ip->tolerance : 0.0, ip ? ip->mono_tolerance : 0.0, ip ? ip->criterion : 0,
                       ip ? ip->require_monotonic : 0);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_aa_snap_slots(Ctx *c)
{
    hipLaunchKernelGGL(k_aa_snap_slots, dim3(256), dim3(256), 0, c->stream, (const double *)c->Ct.as<double>(),
                       c->snapC.as<double>(), (const double *)c->Zt.as<double>(), c->snapZ.as<double>(), c->n_pad,
                       c->KP, (const double *)c->P.as<double>(), c->slotSnapP.as<double>(), (int)c->p_pad, slots_of(c),
                       (const double *)c->alphaDev.as<double>(), c->snapAlpha.as<double>());
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_gpnh_solve_slots(Ctx *c, double lambda)
{
    const size_t lds = (size_t)c->slots_k * c->slots_k * sizeof(double);
    const GpnhSlots sl = slots_of(c);
#define GPS(KMV)                                                                                        \
    hipLaunchKernelGGL(k_gpnh_solve_slots<KMV>, dim3((unsigned)((c->p_pad + 255) / 256), (unsigned)sl.R), dim3(256), lds, c->stream, \
                       (const double *)c->gramState.as<double>(), (const double *)c->ZtX.as<double>(),      \
                       (int)c->p_pad, (int)c->p, c->KP, (double)c->n_global, lambda, c->P.as<double>(),      \
                       c->dtype == AA_F32 ? c->Pw.as<float>() : (float *)nullptr, sl)
    if (c->slots_k <= 16) GPS(16);
    else GPS(32);
#undef GPS
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_gpnh_cost_slots(Ctx *c, double lambda, unsigned mask, int what, const aa_iter_params *ip, bool form_gram)
{
    double *gs = c->gramState.as<double>();
    hipLaunchKernelGGL(k_gpnh_cost_slots, dim3((unsigned)c->slots_R), dim3(256), 0, c->stream, (const double *)gs,
                       gs + (size_t)c->KP * c->KP, c->KP, (int)c->p, c->trace, (double)c->n_global, lambda,
                       (const double *)c->ZtX.as<double>(), (const double *)c->P.as<double>(), (int)c->p_pad,
                       slots_of(c), mask, what, ip ? ip->tolerance : 0.0, ip ? ip->mono_tolerance : 0.0,
                       ip ? ip->criterion : 0, ip ? ip->require_monotonic : 0, form_gram ? 1 : 0);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_gpnh_snap_slots(Ctx *c)
{
    hipLaunchKernelGGL(k_gpnh_snap_slots, dim3(256), dim3(256), 0, c->stream, (const double *)c->Zt.as<double>(),
                       c->snapZ.as<double>(), c->n_pad, c->KP, (const double *)c->P.as<double>(),
                       c->snapC.as<double>(), (int)c->p_pad, slots_of(c));
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_scale_gram(Ctx *c, double *dst, const double *src)
{
    const int elems = c->KP * c->KP;
    hipLaunchKernelGGL(k_scale_gram, dim3((elems + 255) / 256), dim3(256), 0, c->stream, dst, src,
                       (const double *)c->alphaDev.as<double>(), c->k, c->KP, c->slots_aa ? c->slots_k : 0);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_aa_cost(Ctx *c, double *out_dev, int *slot_counter_dev, const GpnhJudge *judge)
{
    GpnhJudge jd;
    memset(&jd, 0, sizeof(jd));
    if (judge) jd = *judge;
    hipLaunchKernelGGL(k_aa_cost, dim3(1), dim3(256), 0, c->stream,
                       (const double *)c->gramState.as<double>(),
                       (const double *)c->alphaDev.as<double>(), c->k, c->KP, c->trace,
                       (double)c->n_global, out_dev, slot_counter_dev,
                       (const double *)c->scalars.as<double>(), jd);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_set_scalars(Ctx *c, double trace, double fnorm)
{
    hipLaunchKernelGGL(k_set_scalars, dim3(1), dim3(64), 0, c->stream, c->scalars.as<double>(), trace,
                       c->slots_aa ? (double)c->slots_k : fnorm, c->slots_aa ? c->slots_R : 1);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_gram_wide(Ctx *c, const double *A, const double *B, double *out_dev)
{
    const int nb = (int)(c->p_pad / 128);
    double *part = c->redPartial.as<double>();
    const int elems = c->KP * c->KP;
    if (c->KP == 32)
        hipLaunchKernelGGL(k_gram_wide<32>, dim3(nb), dim3(256), 0, c->stream, A, B, (int)c->p_pad, part);
    else
        hipLaunchKernelGGL(k_gram_wide<64>, dim3(nb), dim3(256), 0, c->stream, A, B, (int)c->p_pad, part);
    hipLaunchKernelGGL(k_gram_finalize, dim3((elems + 63) / 64), dim3(256), 0, c->stream, part,
                       nb, elems, out_dev);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;   // wide operands are already replicated across ranks
}

// line search of the data form in two launches (k_gram_wide_pq, k_linesearch_fin); cost_out /
// cost_slot: optional recording of the cost after the dictionary update
int launch_linesearch_fused(Ctx *c, const aa_spg_params *sp, double *cost_out, int *cost_slot)
{
    const long chunks = c->p_pad / 128;
    const int mb = g_pq_blocks < 1 ? 1 : g_pq_blocks;
    const int cpb = (int)((chunks + mb - 1) / mb) * 128;      // columns per block: <= g_pq_blocks blocks
    const int nb = (int)((c->p_pad + cpb - 1) / cpb);
    double *part = c->redPartial.as<double>();
    double *ckct = c->gramState.as<double>() + (size_t)c->KP * c->KP;
    if (g_pq_mfma && c->KP == 32)
        hipLaunchKernelGGL(k_gram_wide_pq_mfma<32>, dim3(nb), dim3(256), 0, c->stream,
                           (const double *)c->P.as<double>(), (const double *)c->Q.as<double>(),
                           (int)c->p_pad, part, cpb);
    else if (g_pq_mfma)
        hipLaunchKernelGGL(k_gram_wide_pq_mfma<64>, dim3(nb), dim3(256), 0, c->stream,
                           (const double *)c->P.as<double>(), (const double *)c->Q.as<double>(),
                           (int)c->p_pad, part, cpb);
    else if (c->KP == 32)
        hipLaunchKernelGGL(k_gram_wide_pq<32>, dim3(nb), dim3(256), 0, c->stream,
                           (const double *)c->P.as<double>(), (const double *)c->Q.as<double>(),
                           (int)c->p_pad, part, cpb, c->k);
    else
        hipLaunchKernelGGL(k_gram_wide_pq<64>, dim3(nb), dim3(256), 0, c->stream,
                           (const double *)c->P.as<double>(), (const double *)c->Q.as<double>(),
                           (int)c->p_pad, part, cpb, c->k);
    if (c->slots_aa) {
        SlotRecords rec;
        rec.costs = c->slotCosts.as<double>();
        rec.stride = c->slots_stride;
        rec.counters = c->slotCounters.as<int>();
        hipLaunchKernelGGL(k_linesearch_fin_slots, dim3(1), dim3(1024), 0, c->stream, (const double *)part, nb, c->KP,
                           c->gramOut.as<double>(), (const double *)c->Mdev.as<double>(), c->scalars.as<double>(),
                           *sp, c->slots_k, c->slots_R, ckct, (double)c->n_global, rec);
    } else
    hipLaunchKernelGGL(k_linesearch_fin, dim3(1), dim3(1024), 0, c->stream, (const double *)part, nb, c->KP,
                       c->gramOut.as<double>(), (const double *)c->Mdev.as<double>(),
                       c->scalars.as<double>(), *sp, c->k, ckct, (double)c->n_global, cost_out, cost_slot);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_dict_setup(Ctx *c, const aa_spg_params *sp, double fnorm, unsigned slotmask)
{
    if (c->slots_aa) {
        hipLaunchKernelGGL(k_dict_setup_slots, dim3((unsigned)c->slots_R), dim3(256), 0, c->stream,
                           (const double *)c->gramState.as<double>(), (const double *)c->alphaDev.as<double>(),
                           c->slots_k, c->KP, c->trace, (double)c->slots_k, c->Mdev.as<double>(),
                           c->gramOut.as<double>(), c->scalars.as<double>(), *sp, slotmask);
        AA_CHECK_HIP(hipGetLastError());
        return AA_OK;
    }
    hipLaunchKernelGGL(k_dict_setup, dim3(1), dim3(256), 0, c->stream,
                       (const double *)c->gramState.as<double>(), (const double *)c->alphaDev.as<double>(),
                       c->k, c->KP, c->trace, fnorm, c->Mdev.as<double>(), c->gramOut.as<double>(),
                       c->scalars.as<double>(), *sp);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_wide_axpy_lambda(Ctx *c, double *P, const double *Q, void *PT)
{
    const long elems = (long)c->KP * c->p_pad;
    dim3 grid((unsigned)((elems + 255) / 256));
    if (c->dtype == AA_F32)
        hipLaunchKernelGGL(k_wide_axpy<float>, grid, dim3(256), 0, c->stream, P, Q,
                           c->scalars.as<double>(), elems, reinterpret_cast<float *>(PT), (long)c->p_pad,
                           c->slots_aa ? c->slots_k : 0, c->slots_R);
    else
        hipLaunchKernelGGL(k_wide_axpy<double>, grid, dim3(256), 0, c->stream, P, Q,
                           c->scalars.as<double>(), elems,
                           (PT == (void *)P) ? (double *)nullptr : reinterpret_cast<double *>(PT), (long)c->p_pad,
                           c->slots_aa ? c->slots_k : 0, c->slots_R);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_wide_to_T(Ctx *c, const double *src, void *dstT)
{
    const long elems = (long)c->KP * c->p_pad;
    dim3 grid((unsigned)((elems + 255) / 256));
    if (c->dtype == AA_F32) {
        hipLaunchKernelGGL(k_wide_to_T<float>, grid, dim3(256), 0, c->stream, src, elems,
                           reinterpret_cast<float *>(dstT));
    } else if (dstT != (const void *)src) {
        hipLaunchKernelGGL(k_wide_to_T<double>, grid, dim3(256), 0, c->stream, src, elems,
                           reinterpret_cast<double *>(dstT));
    }
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_transpose_wide_to_tall(Ctx *c, const double *wide, double *tall)
{
    dim3 grid((unsigned)(c->n_pad / 32), (unsigned)(c->KP / 32));
    hipLaunchKernelGGL(k_wide_to_tall, grid, dim3(256), 0, c->stream, wide, (int)c->p_pad, c->KP,
                       c->n_pad, tall);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_transpose_tall_to_wide(Ctx *c, const double *tall, double *wide, void *wideT)
{
    dim3 grid((unsigned)(c->n_pad / 32), (unsigned)(c->KP / 32));
    if (c->dtype == AA_F32)
        hipLaunchKernelGGL(k_tall_to_wide<float>, grid, dim3(256), 0, c->stream, tall, c->KP,
                           (int)c->p_pad, wide, reinterpret_cast<float *>(wideT));
    else
        hipLaunchKernelGGL(k_tall_to_wide<double>, grid, dim3(256), 0, c->stream, tall, c->KP,
                           (int)c->p_pad, wide,
                           (wideT == (void *)wide) ? (double *)nullptr
                                                   : reinterpret_cast<double *>(wideT));
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_scalar_stage(Ctx *c, int stage, const aa_spg_params *sp, int cross2_is_transpose)
{
    if (c->slots_aa)
        hipLaunchKernelGGL(k_scalar_stage, dim3((unsigned)c->slots_R), dim3(256), 0, c->stream, stage,
                           c->scalars.as<double>(), c->gramOut.as<double>(), c->Mdev.as<double>(),
                           c->slots_k, c->KP, *sp, cross2_is_transpose);
    else
    hipLaunchKernelGGL(k_scalar_stage, dim3(1), dim3(256), 0, c->stream, stage,
                       c->scalars.as<double>(), c->gramOut.as<double>(), c->Mdev.as<double>(),
                       c->k, c->KP, *sp, cross2_is_transpose);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_row_sqnorm_sum(Ctx *c, double *trace_out_host)
{
    double *part = c->redPartial.as<double>();
    const int nb = 512;
    if (c->form == AA_FORM_DATA) {
        if (c->dtype == AA_F32)
            hipLaunchKernelGGL(k_sqnorm<float>, dim3(nb), dim3(256), 0, c->stream, c->X.as<float>(),
                               c->p_pad, c->n, (int)c->p, part);
        else
            hipLaunchKernelGGL(k_sqnorm<double>, dim3(nb), dim3(256), 0, c->stream,
                               c->X.as<double>(), c->p_pad, c->n, (int)c->p, part);
    } else {
        if (c->dtype == AA_F32)
            hipLaunchKernelGGL(k_diag_sum<float>, dim3(nb), dim3(256), 0, c->stream,
                               c->X.as<float>(), c->p_pad, c->n, c->row_offset, part);
        else
            hipLaunchKernelGGL(k_diag_sum<double>, dim3(nb), dim3(256), 0, c->stream,
                               c->X.as<double>(), c->p_pad, c->n, c->row_offset, part);
    }
    double *out = red_buf(c);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, part, (long)nb, out);
    AA_CHECK_HIP(hipGetLastError());
    if ((c->world > 1 || c->force_comm)) AA_CHECK(comm_allreduce(c, out, 1, 0));
    AA_CHECK_HIP(hipMemcpyAsync(trace_out_host, out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    return AA_OK;
}

// multi-rank FurthestSum: the owner of global row j publishes it to every rank WITHOUT leaving
// the device: every rank fills a float64 buffer (the owner with its row, the others with zeros),
// one sum all-reduce (x + 0 + ... + 0 is exact), and the result is converted back to the data
// type in wideScratch, where k_distance_data expects x_j.
template <typename T>
__global__ __launch_bounds__(256) void k_row_publish(const T *__restrict__ row, int own, long p_pad,
                                                     double *__restrict__ out)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q < p_pad) out[q] = own ? (double)row[q] : 0.0;
}
template <typename T>
__global__ __launch_bounds__(256) void k_row_collect(const double *__restrict__ in, long p_pad,
                                                     T *__restrict__ out)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q < p_pad) out[q] = (T)in[q];
}

int launch_row_broadcast(Ctx *c, long j_local, bool own)
{
    // staging: the float64 buffer lives in the second half of wideScratch (>= 32 * p_pad doubles)
    double *stage = c->wideScratch.as<double>() + (size_t)16 * c->p_pad;
    const dim3 grid((unsigned)((c->p_pad + 255) / 256));
    const long off = own ? j_local * c->p_pad : 0;
    if (c->dtype == AA_F32)
        hipLaunchKernelGGL(k_row_publish<float>, grid, dim3(256), 0, c->stream, c->X.as<float>() + off,
                           own ? 1 : 0, c->p_pad, stage);
    else
        hipLaunchKernelGGL(k_row_publish<double>, grid, dim3(256), 0, c->stream, c->X.as<double>() + off,
                           own ? 1 : 0, c->p_pad, stage);
    AA_CHECK(comm_allreduce(c, stage, c->p_pad, 0));
    if (c->dtype == AA_F32)
        hipLaunchKernelGGL(k_row_collect<float>, grid, dim3(256), 0, c->stream, (const double *)stage, c->p_pad,
                           c->wideScratch.as<float>());
    else
        hipLaunchKernelGGL(k_row_collect<double>, grid, dim3(256), 0, c->stream, (const double *)stage,
                           c->p_pad, c->wideScratch.as<double>());
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// xj_dev: T-typed copy of global row j (p_pad entries) already resident in wideScratch.
int launch_distance_column(Ctx *c, long j_local, int owner_has_row, const double *xj_host,
                           double *d_host)
{
    (void)owner_has_row;
    (void)xj_host;
    double *dd = c->tmpTall.as<double>();
    if (c->implicit_kernel) {
        hipLaunchKernelGGL(k_distance_rbf, dim3((unsigned)((c->n + 3) / 4)), dim3(256), 0, c->stream,
                           (const double *)c->feat.as<double>(), c->feat_ld, (int)c->feat_p, c->n, j_local, c->rbf_gamma, dd);
    } else if (c->form == AA_FORM_DATA) {
        dim3 grid((unsigned)((c->n + 3) / 4));
        if (c->dtype == AA_F32)
            hipLaunchKernelGGL(k_distance_data<float>, grid, dim3(256), 0, c->stream,
                               c->X.as<float>(), c->p_pad, c->n, (int)c->p_pad,
                               c->wideScratch.as<float>(), dd);
        else
            hipLaunchKernelGGL(k_distance_data<double>, grid, dim3(256), 0, c->stream,
                               c->X.as<double>(), c->p_pad, c->n, (int)c->p_pad,
                               c->wideScratch.as<double>(), dd);
    } else {
        dim3 grid((unsigned)((c->n + 255) / 256));
        if (c->dtype == AA_F32)
            hipLaunchKernelGGL(k_distance_kernel<float>, grid, dim3(256), 0, c->stream,
                               c->X.as<float>(), c->p_pad, c->n, j_local, dd);
        else
            hipLaunchKernelGGL(k_distance_kernel<double>, grid, dim3(256), 0, c->stream,
                               c->X.as<double>(), c->p_pad, c->n, j_local, dd);
    }
    AA_CHECK_HIP(hipGetLastError());
    AA_CHECK_HIP(hipMemcpyAsync(d_host, dd, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost,
                                c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    return AA_OK;
}

// ---------------------------------------------------------------- FurthestSum on the device
// furthest_sum.py:23-127 of the reference with the candidate list as two device arrays (running sums,
// alive flags): a pick is the live candidate with the largest running sum (the reference sorts its
// list by sum, stably, and pops the last element: the same candidate unless the largest sum is
// shared -- then the winner depends on the history of earlier sorts, the host's _Pool knows that
// rule, and this chain only raises a flag so that the caller repeats the selection there).  One
// distance column (k_distance_data / k_distance_kernel arithmetic: i == j gives exactly 0) and one
// single-block step per pick, the index of the picked row passed on in device memory: no host
// round trip per pick (the host-driven form pulled n doubles per pick and spent 5 ms per
// initialisation at 22 280 samples in NumPy: SURVEY 8(f1), the drivers' FurthestSum restarts).
struct FsState {
    int cur, tie, next_slot, pad;
    int selected[AA_MAX_K];
};

template <typename T>
__global__ __launch_bounds__(256) void k_fs_distance(const T *__restrict__ X, long ldx, long n, int p_pad,
                                                     int data_form, const FsState *__restrict__ st,
                                                     double *__restrict__ d)
{
    const long j = st->cur;
    if (data_form) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const long r = (long)blockIdx.x * 4 + wave;
        if (r >= n) return;
        const T *xj = X + j * ldx;
        double sii = 0.0, sij = 0.0, sjj = 0.0;
        for (int c = lane; c < p_pad; c += 64) {
            const double a = (double)X[r * ldx + c], b = (double)xj[c];
            sii = fma(a, a, sii);
            sij = fma(a, b, sij);
            sjj = fma(b, b, sjj);
        }
        sii = wave_sum(sii);
        sij = wave_sum(sij);
        sjj = wave_sum(sjj);
        if (lane == 0) d[r] = sqrt(fmax(sii - 2.0 * sij + sjj, 0.0));
    } else {
        const long r = (long)blockIdx.x * 256 + threadIdx.x;
        if (r >= n) return;
        const double kd = (double)X[r * ldx + r], kj = (double)X[r * ldx + j], jj = (double)X[j * ldx + j];
        d[r] = sqrt(kd - 2.0 * kj + jj);
    }
}

__global__ __launch_bounds__(256) void k_fs_init(unsigned char *__restrict__ alive, long n, int start,
                                                 const int *__restrict__ exclude, int n_ex, int k,
                                                 FsState *__restrict__ st)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        bool ok = i != start;
        for (int e = 0; e < n_ex; ++e) ok = ok && exclude[e] != i;
        alive[i] = ok ? 1 : 0;
    }
    if (i == 0) {
        st->cur = start;
        st->tie = 0;
        st->next_slot = 1;
        for (int q = 0; q < AA_MAX_K; ++q) st->selected[q] = start;
    }
}

// op 0: sums = col (the list as first built, furthest_sum.py:103-108); op 1: sums += col on the live
// candidates (:16-20 after a pick); op 2: sums -= col, the candidate `leaving` = selected[slot] comes
// back with the sum of its distances to the other selected points (:113-123).  pick != 0: afterwards
// the live candidate with the largest sum goes to selected[pick_slot] and becomes st->cur.
// set_cur_slot >= 0 (no pick): st->cur = selected[set_cur_slot] (the next replacement's `leaving`).
template <typename T>
__global__ __launch_bounds__(1024) void k_fs_step(double *__restrict__ sums, unsigned char *__restrict__ alive,
                                                  const double *__restrict__ col, long n, int op, int slot,
                                                  int k, int pick, int pick_slot, int set_cur_slot,
                                                  FsState *__restrict__ st, const T *__restrict__ X, long ldx,
                                                  int p_pad, int data_form)
{
    __shared__ double sv[1024];
    __shared__ long si[1024];
    __shared__ int sc[1024];
    __shared__ double dback[AA_MAX_K];
    const int t = threadIdx.x;
    const int leaving = op == 2 ? st->selected[slot] : -1;
    for (long i = t; i < n; i += 1024)
        if (alive[i]) sums[i] = op == 0 ? col[i] : (op == 1 ? sums[i] + col[i] : sums[i] - col[i]);
    if (op == 2) {
        // D[leaving, other] in the orientation the host path reads it (row = leaving, column = other: the
        // dissimilarity is not symmetric to the last bit), one wave per selected point, the arithmetic
        // of k_fs_distance
        const int wave = t >> 6, lane = t & 63;
        for (int q = wave; q < k; q += 16) {
            const long other = st->selected[q];
            double dv = 0.0;
            if (other != leaving) {
                if (data_form) {
                    const T *xi = X + (long)leaving * ldx, *xj = X + other * ldx;
                    double sii = 0.0, sij = 0.0, sjj = 0.0;
                    for (int c = lane; c < p_pad; c += 64) {
                        const double a = (double)xi[c], b = (double)xj[c];
                        sii = fma(a, a, sii);
                        sij = fma(a, b, sij);
                        sjj = fma(b, b, sjj);
                    }
                    sii = wave_sum(sii);
                    sij = wave_sum(sij);
                    sjj = wave_sum(sjj);
                    dv = sqrt(fmax(sii - 2.0 * sij + sjj, 0.0));
                } else {
                    const double kd = (double)X[(long)leaving * ldx + leaving], kj = (double)X[(long)leaving * ldx + other],
                                 jj = (double)X[other * ldx + other];
                    dv = sqrt(kd - 2.0 * kj + jj);
                }
            }
            if (lane == 0) dback[q] = dv;
        }
    }
    __syncthreads();
    if (op == 2 && t == 0) {
        double back = 0.0;
        for (int q = 0; q < k; ++q)                       // in the order of `selected`, as the reference sums
            if (st->selected[q] != leaving) back += dback[q];
        alive[leaving] = 1;
        sums[leaving] = back;
    }
    __syncthreads();
    if (pick) {
        double best = -INFINITY;
        long arg = -1;
        int cnt = 0;
        for (long i = t; i < n; i += 1024)
            if (alive[i]) {
                const double v = sums[i];
                if (v > best) { best = v; arg = i; cnt = 1; }
                else if (v == best) { cnt += 1; }
            }
        sv[t] = best; si[t] = arg; sc[t] = cnt;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (t < o) {
                const double a = sv[t], b = sv[t + o];
                if (b > a) { sv[t] = b; si[t] = si[t + o]; sc[t] = sc[t + o]; }
                else if (b == a) { sc[t] += sc[t + o]; if (si[t] < 0) si[t] = si[t + o]; }
            }
            __syncthreads();
        }
        if (t == 0) {
            if (sc[0] != 1 || si[0] < 0) st->tie = 1;     // shared maximum (or nothing left): the host decides
            const long a = si[0] < 0 ? 0 : si[0];
            st->selected[pick_slot] = (int)a;
            alive[a] = 0;
            st->cur = (int)a;
        }
    } else if (set_cur_slot >= 0 && t == 0) {
        st->cur = st->selected[set_cur_slot];
    }
}

int launch_furthest_sum(Ctx *c, int k, int start, const int *exclude_host, int n_ex, int extra_steps,
                        int *selected_host, int *tie_host)
{
    const long n = c->n;
    // scratch: sums[n] | col[n] | FsState | exclude[n_ex] | alive[n]
    const size_t off_col = round_up((long)n * sizeof(double), 256);
    const size_t off_st = 2 * off_col;
    const size_t off_ex = off_st + round_up((long)sizeof(FsState), 256);
    const size_t off_alive = off_ex + round_up((long)(n_ex > 0 ? n_ex : 1) * sizeof(int), 256);
    AA_CHECK(c->fsScratch.alloc(off_alive + (size_t)n));
    unsigned char *base = reinterpret_cast<unsigned char *>(c->fsScratch.p);
    double *sums = reinterpret_cast<double *>(base), *col = reinterpret_cast<double *>(base + off_col);
    FsState *st = reinterpret_cast<FsState *>(base + off_st);
    int *ex = reinterpret_cast<int *>(base + off_ex);
    unsigned char *alive = base + off_alive;
    if (n_ex > 0)
        AA_CHECK_HIP(hipMemcpyAsync(ex, exclude_host, (size_t)n_ex * sizeof(int), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_fs_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, alive, n, start,
                       (const int *)ex, n_ex, k, st);
    const int data = c->form == AA_FORM_DATA;
    auto distance = [&]() {
        const dim3 grid((unsigned)(data ? (n + 3) / 4 : (n + 255) / 256));
        if (c->dtype == AA_F32)
            hipLaunchKernelGGL(k_fs_distance<float>, grid, dim3(256), 0, c->stream, c->X.as<float>(), c->p_pad, n,
                               (int)c->p_pad, data, (const FsState *)st, col);
        else
            hipLaunchKernelGGL(k_fs_distance<double>, grid, dim3(256), 0, c->stream, c->X.as<double>(), c->p_pad, n,
                               (int)c->p_pad, data, (const FsState *)st, col);
    };
    auto step = [&](int op, int slot, int pick, int pick_slot, int set_cur_slot) {
        if (c->dtype == AA_F32)
            hipLaunchKernelGGL(k_fs_step<float>, dim3(1), dim3(1024), 0, c->stream, sums, alive, (const double *)col, n,
                               op, slot, k, pick, pick_slot, set_cur_slot, st, (const float *)c->X.as<float>(),
                               c->p_pad, (int)c->p_pad, data);
        else
            hipLaunchKernelGGL(k_fs_step<double>, dim3(1), dim3(1024), 0, c->stream, sums, alive, (const double *)col, n,
                               op, slot, k, pick, pick_slot, set_cur_slot, st, (const double *)c->X.as<double>(),
                               c->p_pad, (int)c->p_pad, data);
    };
    // the list as first built: distances to the start point; first pick
    distance();
    const int first_leaving = extra_steps > 0 ? 0 : -1;
    if (k > 1) step(0, 0, 1, 1, -1);
    else step(0, 0, 0, 0, first_leaving);
    for (int slot = 1; slot < k; ++slot) {
        distance();                                       // column of selected[slot]
        if (slot + 1 < k) step(1, 0, 1, slot + 1, -1);
        else step(1, 0, 0, 0, first_leaving);
    }
    for (int s = 0; s < extra_steps; ++s) {
        const int slot = s % k;
        distance();                                       // column of the point that leaves (st->cur)
        step(2, slot, 1, slot, -1);                       // it comes back as a candidate; the furthest takes its place
        distance();
        step(1, 0, 0, 0, s + 1 < extra_steps ? (s + 1) % k : -1);
    }
    AA_CHECK_HIP(hipGetLastError());
    FsState hst;
    AA_CHECK_HIP(hipMemcpyAsync(&hst, st, sizeof(hst), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    for (int q = 0; q < k; ++q) selected_host[q] = hst.selected[q];
    *tie_host = hst.tie;
    return AA_OK;
}

int launch_residual_cost(Ctx *c, const double *Ztall, const double *Wwide, const double *alpha_dev,
                         double *out_host)
{
    double *part = c->redPartial.as<double>();
    const long nb = (c->n + 3) / 4;
    dim3 grid((unsigned)nb);
#define RESID(T, KPV)                                                                        \
    hipLaunchKernelGGL((k_residual<T, KPV>), grid, dim3(256), 0, c->stream, c->X.as<T>(),    \
                       c->p_pad, c->n, (int)c->p_pad, Ztall, Wwide, alpha_dev, c->k, part)
    if (c->dtype == AA_F32) {
        if (c->KP == 32) RESID(float, 32); else RESID(float, 64);
    } else {
        if (c->KP == 32) RESID(double, 32); else RESID(double, 64);
    }
#undef RESID
    double *out = c->gramOut.as<double>();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, part, nb, out);
    AA_CHECK_HIP(hipGetLastError());
    if ((c->world > 1 || c->force_comm)) AA_CHECK(comm_allreduce(c, out, 1, 0));
    AA_CHECK_HIP(hipMemcpyAsync(out_host, out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));
    return AA_OK;
}

}  // namespace aa
