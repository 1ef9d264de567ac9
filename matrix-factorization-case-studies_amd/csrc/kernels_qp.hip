// kernels_qp.hip -- batched per-sample simplex QP (the weights update) and the
// stateless row-wise simplex projection.
//
// Reference: for every sample t independently, quad_simplex_spg(A, b_t, z_t)
// (spg.py:286-398) called from the serial loops at archetypal_analysis.py:359-366 and
// gpnh_convex_coding.py:244-251.  A (k x k) is shared by all samples.
//
// Four mappings of the same per-sample loop (launch_qp picks; aa_set_option("qp_mode")):
//   k_qp_quad  four lanes per sample in the f64 matrix cores' operand layout, 16 samples per
//              wave, mat-vec = 16 MFMAs on registers as they are       (default, k <= 32)
//   k_qp_wave  one wave per sample, DPP reductions, row_newbcast mat-vec: the low-latency
//              kernel for the samples the others park at their pass cap; all of k > 32
//   k_qp       one lane per sample, 64 samples per wave, mat-vec through LDS + MFMA
//              (default when max_iterations <= 4: nothing diverges)
//   k_qp_row   one DPP row of 16 lanes per sample, run to completion   (qp_mode 3)
// Samples finish after very different numbers of passes (heavy tail: mean 14, max 100-450),
// which is what the batching, the longest-first order and the parking are about.
//
// The lane kernel first.  The whole SPG state of a sample (x, g = Ax + b, d, Ad: 4*KQ
// doubles) lives in that lane's registers, so there is no cross-lane traffic outside the
// mat-vec; a lane that finishes waits for its batch (qp_refill_min).
//
// Restatement notes (each keeps the reference's decisions; differences are rounding):
//   * f along the search direction is the exact quadratic
//       f(x + lam d) = f + lam <d,g> + lam^2/2 d'Ad,
//     so the Armijo back-tracking loop (spg.py:356-368) needs no further mat-vec, and
//     x, g are advanced by lam*d, lam*Ad once the step is accepted;
//   * y = g_new - g = lam*A d, hence <d,y> = lam d'Ad (spg.py:371-376);
//   * the projection is sort-free (Michelot from t = max - 1), same support and the
//     same closed-form threshold as simplex_projection.py:13-27.
#include "aa_internal.h"

namespace aa {

#define QP_MAXMEM 8        // f_mem entries the lane / quad kernels keep in registers (spg.py:310)
#define QW_MAXMEM 32       // ... and the wave-per-sample kernel, which takes over for memory > 8
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Threshold t of the projection of w = x - a*g (components >= k excluded).
//
// Fixed-point form of Michelot's algorithm on a SUPPORT MASK: from a support S,
//   t_S = (sum_S w - 1)/|S|,   S' = {i : w_i > t_S},
// until S' == S.  The fixed point is unique (it is the KKT condition of the projection),
// so any start gives the support -- and, summing it in the same fixed order, the same
// threshold -- the reference's sorted scan finds (simplex_projection.py:13-27).  `mask`
// carries the support of the sample's previous projection: SPG identifies the active set
// after a few passes, after which one sum and one comparison sweep confirm it (the cold
// start from {w > max - 1} costs a maximum sweep and typically 3-5 rounds).  After the
// first round every S is threshold-induced with t below the root (a Newton step on a
// convex decreasing function), so supports only shrink: a support that fails to shrink
// is the fixed point up to rounding.  The comparison is w*|S| > sum_S - 1: no fp64
// division inside the loop.  Sums run in NP interleaved chains combined in a fixed order.
template <int KQ> struct QpMask { typedef unsigned int type; };
template <> struct QpMask<64> { typedef unsigned long long type; };

// FULL: k == KQ, so the `component < k` tests -- otherwise KQ wave-uniform predicates that
// hipcc keeps in SGPR pairs and spills -- vanish at compile time.
template <int KQ, bool FULL = false>
__device__ __forceinline__ double qp_project_threshold(const double (&x)[KQ], const double (&g)[KQ],
                                                       double a, int k,
                                                       typename QpMask<KQ>::type &mask,
                                                       int *rounds = nullptr)
{
    typedef typename QpMask<KQ>::type M;
    constexpr int NP = KQ >= 4 ? 4 : 1;
    M m = mask;
    double s = 0.0;
    int c = 1;
    for (int pass = 0; pass < 2 * KQ + 8; ++pass) {
        if (m == (M)0) {                       // cold start, or the warm guess emptied
            double mxp[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) mxp[q] = -INFINITY;
#pragma unroll
            for (int i = 0; i < KQ; ++i)
                if ((FULL || i < k)) mxp[i % NP] = fmax(mxp[i % NP], x[i] - a * g[i]);
            double mx = mxp[0];
#pragma unroll
            for (int q = 1; q < NP; ++q) mx = fmax(mx, mxp[q]);
            const double t0 = mx - 1.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i)
                if ((FULL || i < k) && x[i] - a * g[i] > t0) m |= (M)1 << i;
        }
        double sp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) sp[q] = 0.0;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const double w = x[i] - a * g[i];
            sp[i % NP] += ((m >> i) & (M)1) ? w : 0.0;
        }
        s = sp[0];
#pragma unroll
        for (int q = 1; q < NP; ++q) s += sp[q];
        c = __popcll((unsigned long long)m);
        const double sm1 = s - 1.0, cd = (double)c;
        M nm = 0;
#pragma unroll
        for (int i = 0; i < KQ; ++i)
            if ((FULL || i < k) && (x[i] - a * g[i]) * cd > sm1) nm |= (M)1 << i;
        if (rounds) *rounds += 1;
        if (nm == m || (pass >= 2 && (int)__popcll((unsigned long long)nm) >= c)) break;
        m = nm;
    }
    mask = m;
    return (s - 1.0) / (double)c;
}

// Continuation record of a sample whose SPG loop hit the phase-1 pass cap.
struct QpCarry {
    double alpha, f;
    int n_iter, n_feval;
};

// Device-side header of the QP scratch buffer (zeroed by the host before each update).
struct QpHeader {
    unsigned long long total_passes, max_passes;
    unsigned int next_row;        // phase-1 work counter
    unsigned int n_overflow;      // samples handed to phase 2
    unsigned int next_overflow;   // phase-2 work counter
    unsigned int n_long;          // hybrid row/wave update: head of the sorted list that the
                                  // wave-per-sample kernel takes (predicted-long samples)
    unsigned int dbg_rounds, dbg_trips, dbg_waves, pad;   // qp_profile counters of the row kernel
    unsigned int waves_done;      // live hand-over: producer waves (k_qp_quad) that have exited
    unsigned int pad1, pad2, pad3;
};

// Live hand-over of parked samples (qp_live): k_qp_quad publishes a parked sample at once (its
// iterate, its continuation record and its row go out with agent-scope stores, then ready[slot] =
// epoch) and a consumer launch of k_qp_wave that runs BESIDE k_qp_quad on the side stream takes the
// slots by ticket -- the dependent chain of the longest sample (100-450 passes at ~0.9 us) then starts
// when that sample reaches the pass cap, not when the last batch of k_qp_quad has finished.
// epoch: number of this update (flags of earlier updates never match: nothing to clear).
// mode 0: off; 1: consumer (tickets, waits for producers); 2: clean-up after both kernels (static
// slots, skips the ones a consumer finished: done[slot] == epoch).
struct QpLive {
    int mode, epoch;
    unsigned int producer_waves;
    unsigned int n;               // samples of this update: no slot beyond it can ever exist
    int *ready, *done;
};

__device__ __forceinline__ void qp_store_agent(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double qp_load_agent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(
        reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void qp_store_agent(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int qp_load_agent(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int qp_debug_wave_max(int v)   // maximum of v over the active lanes
{
    int m = 0;
    for (int r = 1; r < 200; ++r) {
        if (!__any(v >= r)) break;
        m = r;
    }
    return m;
}

// Optional cycle accounting of the lane-per-sample kernel (aa_set_option("qp_profile", 1));
// lives at byte 64 of the scratch buffer, printed by the host after the update.
struct QpDebug {
    unsigned long long trips, refills, cyc_total, cyc_proj, cyc_matvec, waves, cyc_step, cyc_fin;
    unsigned long long proj_calls, proj_rounds_wavemax, proj_rounds_lanesum, proj_lanes;
};

// ---------------------------------------------------------------------------
// phase 1: one lane per sample, A broadcast from LDS.
// ---------------------------------------------------------------------------
// out = A v for the sample of this lane.  v sits in LDS as vl[j*64] (lane-private
// column, conflict-free), A transposed as AsT[j][i] so one step reads a contiguous,
// wave-uniform (broadcast) 8*KQ-byte row.  The j loop is a real loop: its loads depend
// on j, so the compiler cannot hoist the whole matrix into registers.
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned lds_byte_address(const void *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p;
}

// Batched LDS reads with ONE wait.  Under the register pressure of this kernel hipcc
// otherwise re-uses a single VGPR quad for every ds_read_b128 of the mat-vec and waits
// lgkmcnt(0) after each one (61 % of the wave's cycles parked in s_waitcnt: SQ_WAIT_ANY).
// Loads and their s_waitcnt sit in one asm statement with early-clobber outputs.
__device__ __forceinline__ void lds_read_x8(unsigned addr, f64x2 (&q)[8])
{
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %8 offset:16\n\t"
                 "ds_read_b128 %2, %8 offset:32\n\t"
                 "ds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:64\n\t"
                 "ds_read_b128 %5, %8 offset:80\n\t"
                 "ds_read_b128 %6, %8 offset:96\n\t"
                 "ds_read_b128 %7, %8 offset:112\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]),
                   "=&v"(q[6]), "=&v"(q[7])
                 : "v"(addr)
                 : "memory");
}
__device__ __forceinline__ void lds_read_x4(unsigned addr, f64x2 (&q)[4])
{
    asm volatile("ds_read_b128 %0, %4\n\t"
                 "ds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
                 : "v"(addr)
                 : "memory");
}
__device__ __forceinline__ void lds_read_x2(unsigned addr, f64x2 (&q)[2])
{
    asm volatile("ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %2 offset:16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1])
                 : "v"(addr)
                 : "memory");
}

// out = A v for the sample of this lane.  v sits in LDS as vl[j*64] (lane-private
// column, conflict-free), A transposed as AsT[j][i] so one step reads a contiguous,
// wave-uniform (broadcast) 8*KQ-byte row.  The j loop is a real loop: its loads depend
// on j, so the compiler cannot hoist the whole matrix into registers.
template <int KQ>
__device__ __forceinline__ void qp_matvec(const double *__restrict__ AsT,
                                          const double *__restrict__ vl, int k, double (&out)[KQ])
{
#pragma unroll
    for (int i = 0; i < KQ; ++i) out[i] = 0.0;
    const unsigned a0 = lds_byte_address(AsT);
    for (int j = 0; j < k; ++j) {
        const double vj = vl[j * 64];
        const unsigned row = a0 + (unsigned)(j * KQ * 8);
        if constexpr (KQ >= 16) {
#pragma unroll
            for (int h = 0; h < KQ / 16; ++h) {
                f64x2 q[8];
                lds_read_x8(row + h * 128, q);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    out[h * 16 + 2 * i] = fma(q[i][0], vj, out[h * 16 + 2 * i]);
                    out[h * 16 + 2 * i + 1] = fma(q[i][1], vj, out[h * 16 + 2 * i + 1]);
                }
            }
        } else if constexpr (KQ == 8) {
            f64x2 q[4];
            lds_read_x4(row, q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out[2 * i] = fma(q[i][0], vj, out[2 * i]);
                out[2 * i + 1] = fma(q[i][1], vj, out[2 * i + 1]);
            }
        } else {
            f64x2 q[2];
            lds_read_x2(row, q);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                out[2 * i] = fma(q[i][0], vj, out[2 * i]);
                out[2 * i + 1] = fma(q[i][1], vj, out[2 * i + 1]);
            }
        }
    }
}

// Mat-vec of the whole wave on the f64 matrix cores (KQ >= 16): the 64 directions of the
// wave form a KQ x 64 matrix D that already sits in LDS (vbuf[j][sample], written one
// column per lane), so  (A D)' = D' A'  is a 64 x KQ x KQ GEMM in 16x16x4 tiles:
//   A-operand lane l = D[4s + (l>>4)][16mt + (l&15)]   (ds_read_b64 from vbuf)
//   B-operand lane l = A[16nt + (l&15)][4s + (l>>4)]   (constant: KQ/2 registers per lane)
//   D-result  lane l, reg r = (A d)[component 16nt + (l&15)] of sample 16mt + (l>>4) + 4r,
// which goes back through LDS (abuf[component][sample]) so that every lane ends up with the
// KQ components of ITS sample.  Compared with the per-lane FMA loop this needs no broadcast
// reads of A at all (512 ds_read_b128 and their waits per mat-vec) and 64 instead of 1024
// arithmetic instructions.  Row strides: 80 doubles for vbuf and 66 for abuf make the
// fragment reads and the result writes bank-conflict free (the lane-private accesses stay
// unit stride).  Every lane must take part (a lane's fragments belong to other samples):
// call with the full wave, outside divergent control flow.
#define QP_VS 80
#define QP_AS 66
template <int KQ>
__device__ __forceinline__ void qp_matvec_mfma(const double (&Breg)[KQ / 16][KQ / 4],
                                               const double *__restrict__ vbuf,
                                               double *__restrict__ abuf, int lane,
                                               double (&out)[KQ])
{
    constexpr int NT = KQ / 16, NS = KQ / 4;
    const int lc = lane & 15, lr = lane >> 4;
    // the columns were written by other lanes of this wave (see the note below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // two sample tiles x NT component tiles = 2*NT independent accumulators in flight: a
    // single accumulator would serialise the MFMAs on their result latency
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
        double a0[NS], a1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            a0[s] = vbuf[(4 * s + lr) * QP_VS + 32 * mp + lc];
            a1[s] = vbuf[(4 * s + lr) * QP_VS + 32 * mp + 16 + lc];
        }
        f64x4 acc0[NT], acc1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc0[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
            acc1[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc0[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], Breg[nt][s], acc0[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], Breg[nt][s], acc1[nt], 0, 0, 0);
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                abuf[(16 * nt + lc) * QP_AS + 32 * mp + lr + 4 * r] = acc0[nt][r];
                abuf[(16 * nt + lc) * QP_AS + 32 * mp + 16 + lr + 4 * r] = acc1[nt][r];
            }
    }
    // one wave per block: LDS executes a wave's operations in order; the barrier only
    // keeps the compiler from moving the column reads above the tile writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int i = 0; i < KQ; ++i) out[i] = abuf[i * QP_AS + lane];
}

// Mat-vec of the lane kernel through the SCALAR unit (KQ >= 16, default): A is the same for all
// 64 samples of the wave, so its entries travel as SGPR operands of v_fma_f64 -- s_load_dwordx16
// from the constant address space, 16 entries per chunk, two chunks in flight ahead of the one
// being consumed -- and the vector v of this lane's sample never leaves its registers: no LDS
// round trip, no cross-lane traffic, 1024 FMAs.  7 100 cycles per mat-vec against 17 800 for the
// MFMA + LDS form above (tools/probes/sgpr_matvec_probe.hip; the VALU floor is 4 096).
//   * the pointer passes through an empty asm with an SGPR constraint: without it LICM hoists all
//     1024 scalar loads out of the trip loop and spills 2 000 SGPRs into VGPR lanes;
//   * the sched_barriers keep the machine scheduler from clustering the loads of all chunks.
// v_j = x_j for a lane whose sample is starting (g = A x + b), else the search direction
// d_j = max(x_j - alpha_d g_j - td, 0) - x_j recomputed from the registers it is defined by.
typedef const __attribute__((address_space(4))) double *qp_cptr_t;
template <int KQ>
__device__ __forceinline__ void qp_matvec_sgpr(const double *__restrict__ A, const double (&x)[KQ],
                                               const double (&g)[KQ], double alpha_d, double td,
                                               bool use_x, double (&out)[KQ])
{
    constexpr int CH = 16, D = 2, NCH = KQ * KQ / CH, CPR = KQ / CH;
    qp_cptr_t Ap = (qp_cptr_t)(unsigned long long)A;
    asm volatile("" : "+s"(Ap));
#pragma unroll
    for (int i = 0; i < KQ; ++i) out[i] = 0.0;
    double buf[D + 1][CH];
#pragma unroll
    for (int q = 0; q < D; ++q)
#pragma unroll
        for (int e = 0; e < CH; ++e) buf[q][e] = Ap[q * CH + e];
    double vj = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + D < NCH) {
#pragma unroll
            for (int e = 0; e < CH; ++e) buf[(c + D) % (D + 1)][e] = Ap[(c + D) * CH + e];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int j = c / CPR, i0 = (c % CPR) * CH;
        if (c % CPR == 0) {
            const double dj = fmax(x[j] - alpha_d * g[j] - td, 0.0) - x[j];
            vj = use_x ? x[j] : dj;
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) out[i0 + e] = fma(buf[c % (D + 1)][e], vj, out[i0 + e]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// PROF: cycle accounting compiled in (aa_set_option("qp_profile", 1)); the counters cost
// registers, so the production instantiation has none of it.
// Components k..KQ-1 are padding.  Their x is 0 and their gradient is held at QP_PAD = 1e300
// (b_i = QP_PAD, the padded rows of A are zero), so x - a g is hugely negative for every step
// a > 0, they never enter a support, their direction and residual are exactly 0, and the pass
// loop needs no `component < k` predicate at all (32 wave-uniform predicates that hipcc kept
// in SGPR pairs, spilled, and branched on).  Only the start-up and the final store test i < k.
#define QP_PAD 1e300
template <int KQ, bool FULL, bool PROF, bool SGMV>
__global__ __launch_bounds__(64) void k_qp(const double *__restrict__ A /*[KQ][KQ]*/,
                                           const double *__restrict__ B, long stride_j,
                                           long stride_t, const double *__restrict__ bscale,
                                           double *__restrict__ Z, int ldz, long n, int k,
                                           aa_qp_params p, int pass_cap, int *__restrict__ iters,
                                           QpHeader *__restrict__ hdr,
                                           int *__restrict__ ovf_rows, QpCarry *__restrict__ ovf,
                                           int g_refill, QpDebug *__restrict__ dbg,
                                           const int *__restrict__ perm, int rst_a = 0, int rst_b = 0,
                                           int rst_z = 0)
{
    // restarts side by side (launch_qp_slots): blockIdx.y = slot; its Hessian, its columns of B and
    // Z, its own header (sample queue, statistics) and pass counts
    if (gridDim.y > 1) {
        const int rst = blockIdx.y;
        A += (long)rst * rst_a;
        B += (long)rst * rst_b;
        Z += (long)rst * rst_z;
        hdr += rst;
        if (iters) iters += (long)rst * n;
    }
    constexpr bool SG = SGMV && KQ >= 16;              // mat-vec through the scalar unit
    constexpr bool MFMA = KQ >= 16 && !SG;
    constexpr int VS = MFMA ? QP_VS : 64;              // row stride of the direction buffer
    __shared__ __attribute__((aligned(16))) double AsT[(MFMA || SG) ? 2 : KQ * KQ];
    __shared__ double vbuf[SG ? 64 : KQ * VS];
    __shared__ double abuf[MFMA ? KQ * QP_AS : 1];
    const int lane = threadIdx.x;
    double Breg[MFMA ? KQ / 16 : 1][MFMA ? KQ / 4 : 1];
    if constexpr (MFMA) {
#pragma unroll
        for (int nt = 0; nt < KQ / 16; ++nt)
#pragma unroll
            for (int s = 0; s < KQ / 4; ++s)
                Breg[nt][s] = A[(16 * nt + (lane & 15)) * KQ + 4 * s + (lane >> 4)];
    } else if constexpr (!SG) {
        for (int e = threadIdx.x; e < KQ * KQ; e += 64) AsT[(e % KQ) * KQ + e / KQ] = A[e];
        Breg[0][0] = 0.0;
    } else {
        Breg[0][0] = 0.0;
    }
    __syncthreads();
    double *vl = vbuf + threadIdx.x;

    double x[KQ], g[KQ], Ad[KQ];
    const int refill_min = g_refill;
    double f = 0.0, alpha = 1.0, fmem[QP_MAXMEM];
    double delta = 0.0, dd = 0.0, td = 0.0, alpha_d = 0.0;   // direction: d = P(x - alpha_d g) - x
    int n_iter = 0, n_feval = 0;
    long row = -1;
    bool active = false, exhausted = false;
    // supports of this sample's latest direction / residual projection (the two differ by
    // the step: x - alpha g against x - g), each the warm start of the next one of its kind
    typename QpMask<KQ>::type support = 0, support_r = 0;
    const int mem = p.memory < 1 ? 1 : (p.memory > QP_MAXMEM ? QP_MAXMEM : p.memory);
    if constexpr (!SG) {
#pragma unroll
        for (int i = 0; i < KQ; ++i) vl[i * VS] = 0.0;
    }
    bool starting_mv = false;                          // this lane's mat-vec input is x (start-up)

    auto matvec = [&](double (&out)[KQ]) {            // out = A v, v = this lane's LDS column
        if constexpr (SG) qp_matvec_sgpr<KQ>(A, x, g, alpha_d, td, starting_mv, out);
        else if constexpr (MFMA) qp_matvec_mfma<KQ>(Breg, vbuf, abuf, lane, out);
        else qp_matvec<KQ>(AsT, vl, k, out);
    };

    unsigned long long st_total = 0ull;
    int st_max = 0;
    long long pc_proj = 0, pc_mv = 0, pc_refills = 0, pc_trips = 0, pc_step = 0, pc_fin = 0;
    long long pc_lane_rounds = 0, pc_lane_calls = 0, pc_wavemax = 0;
    const long long pc_start = PROF ? clock64() : 0;
#define QP_TIC(var) const long long var = PROF ? clock64() : 0
#define QP_TOC(acc, var) if constexpr (PROF) acc += clock64() - var
    // the trip bound is a watchdog only (each sample needs <= max_iterations trips)
    for (long trip = 0; trip < (1L << 24); ++trip) {
        if constexpr (PROF) pc_trips = trip;
        // Refill idle lanes in batches: the start-up of a sample (strided loads, a
        // projection and a mat-vec) is executed by the whole wave, so it is only entered
        // when enough lanes are waiting (or nothing else is left to do).
        const int n_idle = __popcll(__ballot(!active && !exhausted));
        const bool refill = n_idle >= refill_min || !__any(active);
        bool starting = false;
        if (refill && !active && !exhausted) {
            const unsigned int nxt = atomicAdd(&hdr->next_row, 1u);
            if ((long)nxt < n) {
                row = perm ? (long)perm[nxt] : (long)nxt;   // longest-first order (k_qp_order_rows)
                starting = true;
                // ---- start-up: x = P(z0); g = A x + b; f = x'(g + b)/2      (spg.py:298-315)
#pragma unroll
                for (int i = 0; i < KQ; ++i) {
                    x[i] = (FULL || i < k) ? Z[row * ldz + i] : -QP_PAD;   // padding: never in a support
                    g[i] = 0.0;
                }
                support = 0;
                const double t0 = qp_project_threshold<KQ, true>(x, g, 0.0, k, support);
                support_r = support;
#pragma unroll
                for (int i = 0; i < KQ; ++i) x[i] = (FULL || i < k) ? fmax(x[i] - t0, 0.0) : 0.0;
                if constexpr (!SG) {
#pragma unroll
                    for (int i = 0; i < KQ; ++i) vl[i * VS] = x[i];
                }
            } else {
                exhausted = true;   // queue drained: this lane idles
            }
        }
        // A starting lane has put x into its LDS column, an active lane will put its search
        // direction there: ONE collective mat-vec per trip serves both (g = A x + b for the
        // former, A d for the latter), so pulling in new samples costs no extra mat-vec.
        if (!__any(active || starting)) break;
        QP_TIC(tp0);
        if (active) {
            // ---- one pass of the loop at spg.py:318-396, up to the search direction
            if (n_iter == 0) {
                if (p.alpha_min <= p.alpha0 && p.alpha0 <= p.alpha_max) {
                    alpha = p.alpha0;
                } else {
                    const double t1 = qp_project_threshold<KQ, true>(x, g, 1.0, k, support_r);
                    double ainv = 0.0;
#pragma unroll
                    for (int i = 0; i < KQ; ++i)
                        ainv = fmax(ainv, fabs(fmax(x[i] - g[i] - t1, 0.0) - x[i]));
                    if (fabs(ainv) < 1e-12) ainv = 1.0;
                    alpha = fmin(fmax(p.alpha_min, 1.0 / ainv), p.alpha_max);
                }
            }
            int rd_d = 0;
            td = qp_project_threshold<KQ, true>(x, g, alpha, k, support, PROF ? &rd_d : nullptr);
            if constexpr (PROF) {
                pc_lane_rounds += rd_d;
                pc_lane_calls += 1;
            }
            alpha_d = alpha;
            delta = 0.0;
            dd = 0.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const double di = fmax(x[i] - alpha_d * g[i] - td, 0.0) - x[i];
                if constexpr (!SG) vl[i * VS] = di;   // to LDS for the mat-vec; recomputed below
                delta = fma(di, g[i], delta);
                dd = fma(di, di, dd);
            }
        }
        QP_TOC(pc_proj, tp0);
        QP_TIC(tm1);
        starting_mv = starting;
        matvec(Ad);                                    // collective (idle lanes: stale columns)
        QP_TOC(pc_mv, tm1);
        QP_TIC(ts0);
        if (starting) {
            // ---- rest of the start-up: g = A x + b; f = x'(g + b)/2      (spg.py:298-315)
            if constexpr (PROF) pc_refills += 1;
            double xg = 0.0, xb = 0.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const double bi =
                    (FULL || i < k) ? -B[i * stride_j + row * stride_t] * (bscale ? bscale[i] : 1.0) : QP_PAD;
                g[i] = Ad[i] + bi;
                xg = fma(x[i], g[i], xg);
                xb = fma(x[i], bi, xb);
            }
            f = 0.5 * (xg + xb);
            n_feval = 1;
            n_iter = 0;
#pragma unroll
            for (int i = 0; i < QP_MAXMEM; ++i) fmem[i] = NAN;
            active = true;                             // its first pass runs in the next trip
        } else if (active) {
            // d is recomputed from (x, g, alpha_d, td) -- the same three operations, so the same
            // bits -- instead of read back from LDS (32 dependent-latency reads per use)
            double dAd = 0.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const double di = fmax(x[i] - alpha_d * g[i] - td, 0.0) - x[i];
                dAd = fma(di, Ad[i], dAd);
            }
            // non-monotone reference value (spg.py:341-344): roll, store, nanmax
#pragma unroll
            for (int i = QP_MAXMEM - 1; i > 0; --i)
                if (i < mem) fmem[i] = fmem[i - 1];
            fmem[0] = f;
            double f_max = f;
#pragma unroll
            for (int i = 1; i < QP_MAXMEM; ++i)
                if (i < mem && fmem[i] > f_max) f_max = fmem[i];

            double lam = 1.0;
            double f_new = f + lam * delta + 0.5 * lam * lam * dAd;
            n_feval += 1;
            int guard = 0;
            while (f_new > f_max + p.gamma * lam * delta && guard < 200) {
                const double tmp = -0.5 * lam * lam * delta / (f_new - f - lam * delta);
                lam = (p.sigma_one <= tmp && tmp <= p.sigma_two * lam) ? tmp : 0.5 * lam;
                f_new = f + lam * delta + 0.5 * lam * lam * dAd;
                n_feval += 1;
                ++guard;
                if (fabs(lam) < p.lambda_min) break;
            }
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const double di = fmax(x[i] - alpha_d * g[i] - td, 0.0) - x[i];
                x[i] = fma(lam, di, x[i]);
                g[i] = fma(lam, Ad[i], g[i]);
            }
            const double sksk = lam * lam * dd;
            const double beta = lam * (lam * dAd);
            alpha = (beta <= 0.0) ? p.alpha_max : fmin(p.alpha_max, fmax(p.alpha_min, sksk / beta));
            f = f_new;
            n_feval += 1;

            QP_TOC(pc_step, ts0);
            QP_TIC(tp1);
            int rd_r = 0;
            const double tr = qp_project_threshold<KQ, true>(x, g, 1.0, k, support_r, PROF ? &rd_r : nullptr);
            if constexpr (PROF) {
                pc_lane_rounds += rd_r;
                pc_lane_calls += 1;
                pc_wavemax += qp_debug_wave_max(rd_r);
            }
            QP_TOC(pc_proj, tp1);
            QP_TIC(tf0);
            double r2 = 0.0, rinf = 0.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i)
                {
                    const double r = fmax(x[i] - g[i] - tr, 0.0) - x[i];
                    r2 = fma(r, r, r2);
                    rinf = fmax(rinf, fabs(r));
                }
            n_iter += 1;
            const bool conv = (sqrt(r2) < p.epsilon_two) || (rinf < p.epsilon_one);
            const bool finished = conv || n_feval > p.max_feval || n_iter >= p.max_iterations;
            if (finished || n_iter >= pass_cap) {
#pragma unroll
                for (int i = 0; i < KQ; ++i)
                    if (FULL || i < k) Z[row * ldz + i] = x[i];
                if (finished) {
                    if (iters) iters[row] = n_iter;
                    st_total += (unsigned long long)n_iter;      // one atomic pair per wave at exit:
                    st_max = n_iter > st_max ? n_iter : st_max;   // 2 x 100 000 atomics on one cache
                                                                  // line cost more than the QPs
                } else {
                    // hand the sample to the low-latency wave-per-sample kernel
                    const unsigned int slot = atomicAdd(&hdr->n_overflow, 1u);
                    ovf_rows[slot] = (int)row;
                    QpCarry cr;
                    cr.alpha = alpha;
                    cr.f = f;
                    cr.n_iter = n_iter;
                    cr.n_feval = n_feval;
                    ovf[slot] = cr;
                }
                active = false;
            }
            QP_TOC(pc_fin, tf0);
        }
    }
    {   // wave totals (fixed-order butterfly), one atomic pair per wave
        unsigned long long tot = st_total;
        int mx = st_max;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            tot += __shfl_xor(tot, o, 64);
            const int om = __shfl_xor(mx, o, 64);
            mx = om > mx ? om : mx;
        }
        if (threadIdx.x == 0 && tot) {
            atomicAdd(&hdr->total_passes, tot);
            atomicMax(&hdr->max_passes, (unsigned long long)mx);
        }
    }
    if constexpr (PROF) {
        atomicAdd(&dbg->proj_rounds_lanesum, (unsigned long long)pc_lane_rounds);
        atomicAdd(&dbg->proj_lanes, (unsigned long long)pc_lane_calls);
    }
    if (PROF && threadIdx.x == 0) {
        atomicAdd(&dbg->proj_calls, (unsigned long long)pc_trips + 1);
        atomicAdd(&dbg->proj_rounds_wavemax, (unsigned long long)pc_wavemax);
        atomicAdd(&dbg->cyc_step, (unsigned long long)pc_step);
        atomicAdd(&dbg->cyc_fin, (unsigned long long)pc_fin);
        atomicAdd(&dbg->trips, (unsigned long long)(pc_trips + 1));
        atomicAdd(&dbg->refills, (unsigned long long)pc_refills);
        atomicAdd(&dbg->cyc_total, (unsigned long long)(clock64() - pc_start));
        atomicAdd(&dbg->cyc_proj, (unsigned long long)pc_proj);
        atomicAdd(&dbg->cyc_matvec, (unsigned long long)pc_mv);
        atomicAdd(&dbg->waves, 1ull);
    }
#undef QP_TIC
#undef QP_TOC
}

// ---------------------------------------------------------------------------
// phase 2 (and the whole update when k > 32): ONE WAVE PER SAMPLE, lane = component.
// Short critical path per SPG pass (a handful of wave reductions and a k-step
// broadcast mat-vec), so the few samples that need hundreds of passes do not hold a
// 64-sample wave hostage.  memory == 1 only on the continuation path (the host keeps
// samples in phase 1 otherwise); fresh samples support any memory.
// ---------------------------------------------------------------------------
// sqrt(v) < e (spg.py:388-390: ||P(x - g) - x||_2 < epsilon_two) without the square root in the loop:
// the correctly rounded square root is monotone, so with  lim = the smallest double whose square root
// is >= e  the test is exactly  v < lim  (same decision for every v, NaN included).  lim sits within an
// ulp or two of e * e and is found once per wave with the device's own sqrt; *ok = false (e * e
// denormal or not finite: never in practice) keeps the sqrt form.  ~20 instructions off every pass of the
// latency-bound wave-per-sample kernel.
__device__ __forceinline__ double qp_sq_limit(double e, bool *ok)
{
    double y = e * e;
    *ok = (e > 0.0) && (y > 1e-290) && (y < 1e290);
    if (!*ok) return 0.0;
    for (int i = 0; i < 4 && sqrt(y) >= e; ++i) y = __longlong_as_double(__double_as_longlong(y) - 1);
    for (int i = 0; i < 8 && sqrt(y) < e; ++i) y = __longlong_as_double(__double_as_longlong(y) + 1);
    *ok = sqrt(y) >= e && sqrt(__longlong_as_double(__double_as_longlong(y) - 1)) < e;
    return y;
}

// lane exchange across rows of 16 / halves of 32 (gfx950 v_permlane16_swap / v_permlane32_swap)
typedef unsigned int qq_u2 __attribute__((ext_vector_type(2)));

template <bool R32>
__device__ __forceinline__ void qq_xchg(unsigned int v, unsigned int &a, unsigned int &b)
{
    // both operands = v: afterwards `a` holds the lower partner's word and `b` the upper partner's
    // in BOTH lanes of a pair (rows r / r^1 for permlane16, halves for permlane32)
    qq_u2 r;
    if constexpr (R32) r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    else r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    a = r[0];
    b = r[1];
}
template <bool R32>
__device__ __forceinline__ void qq_xchg_d(double v, double &a, double &b)
{
    unsigned int la, lb, ha, hb;
    qq_xchg<R32>((unsigned int)__double2loint(v), la, lb);
    qq_xchg<R32>((unsigned int)__double2hiint(v), ha, hb);
    a = __hiloint2double((int)ha, (int)la);
    b = __hiloint2double((int)hb, (int)lb);
}
// Wave-wide reductions on the DPP crossbar (no LDS traffic): inclusive row scan with
// row_shr 1/2/4/8, then row_bcast15 / row_bcast31 carry the row totals upward; the total
// ends in lane 63 and is broadcast through an SGPR.  Fixed order => deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double qw_dpp(double old, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// row shift with zero fill (bound_ctrl): no `old` operand to materialise
template <int CTRL>
__device__ __forceinline__ double qw_dpp_z(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qw_readlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// HALF (k <= 32): lanes 32..63 MIRROR lanes 0..31 (lane l+32 holds component l too), the
// total of a half needs one step less (lane 31 / lane 63), and two different quantities
// -- one per half -- are reduced by the same five instructions.
__device__ __forceinline__ double qw_scan_rows(double v)      // inclusive scan inside rows of 16
{
    v += qw_dpp_z<0x111>(v);   // row_shr:1
    v += qw_dpp_z<0x112>(v);   // row_shr:2
    v += qw_dpp_z<0x114>(v);   // row_shr:4
    v += qw_dpp_z<0x118>(v);   // row_shr:8
    return v;
}
template <bool HALF>
__device__ __forceinline__ double qw_sum(double v)
{
    v = qw_scan_rows(v);
    v += qw_dpp<0x142, 0xa>(0.0, v);   // row_bcast:15 -> rows 1, 3
    if constexpr (HALF) return qw_readlane(v, 31);
    v += qw_dpp<0x143, 0xc>(0.0, v);   // row_bcast:31 -> rows 2, 3
    return qw_readlane(v, 63);
}
// sums of a and b over the components (both arguments valid in every lane)
template <bool HALF>
__device__ __forceinline__ void qw_sum2(double a, double b, int lane, double &sa, double &sb)
{
    if constexpr (HALF) {
        double v = qw_scan_rows(lane < 32 ? a : b);
        v += qw_dpp<0x142, 0xa>(0.0, v);
        sa = qw_readlane(v, 31);
        sb = qw_readlane(v, 63);
    } else {
        sa = qw_sum<false>(a);
        sb = qw_sum<false>(b);
    }
}
// maxima: lanes without a source keep their own value (old = v), which is neutral for max
template <bool HALF>
__device__ __forceinline__ double qw_max(double v)
{
    v = fmax(v, qw_dpp<0x111, 0xf>(v, v));
    v = fmax(v, qw_dpp<0x112, 0xf>(v, v));
    v = fmax(v, qw_dpp<0x114, 0xf>(v, v));
    v = fmax(v, qw_dpp<0x118, 0xf>(v, v));
    v = fmax(v, qw_dpp<0x142, 0xa>(v, v));
    if constexpr (HALF) return qw_readlane(v, 31);
    v = fmax(v, qw_dpp<0x143, 0xc>(v, v));
    return qw_readlane(v, 63);
}
// Threshold of the projection of the wave-distributed vector w (w = -inf on lanes without
// a component): the support-mask fixed point described at qp_project_threshold, the mask
// being a wave-uniform bit set (one reduction per round; a confirmed warm support costs
// exactly one).  Division-free comparison w*|S| > sum_S - 1.
// a wave-uniform 64-bit value, told to the compiler (it then lives in an SGPR pair and the
// branches on it are scalar; hipcc otherwise keeps the support masks in VGPRs and predicates)
__device__ __forceinline__ unsigned long long qw_uniform64(unsigned long long v)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// w on the lanes of the mask, 0 elsewhere: the mask goes straight into v_cndmask as its lane mask
__device__ __forceinline__ double qw_select(unsigned long long lanes, double w)
{
    int lo, hi;
    asm("v_cndmask_b32_e64 %0, 0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4"
        : "=&v"(lo), "=&v"(hi)
        : "v"(__double2loint(w)), "v"(__double2hiint(w)), "s"(lanes));
    return __hiloint2double(hi, lo);
}
// cold start of the support ({w > max - 1}); out of line: inlined, hipcc runs its max-scan
// speculatively in EVERY Michelot round (45 instructions of a latency-bound loop)
template <bool HALF>
__device__ __attribute__((noinline)) unsigned long long qw_cold_support(double w)
{
    const double t0 = qw_max<HALF>(w) - 1.0;
    return __ballot(w > t0);
}
// Threshold of the projection of the wave-distributed vector w (w = -inf on lanes without a
// component).  The support is kept as a LANE mask (k <= 32: both mirror halves set), which is
// what __ballot returns and what v_cndmask consumes.
template <bool HALF>
__device__ __forceinline__ double qw_threshold(double w, int comp, unsigned long long &mask)
{
    unsigned long long m = qw_uniform64(mask);
    double s = 0.0;
    int c = 1;
    for (int pass = 0; pass < 136; ++pass) {
        if (__builtin_expect(m == 0ull, 0)) m = qw_uniform64(qw_cold_support<HALF>(w));   // first projection, or the warm guess emptied
        c = HALF ? __popc((unsigned int)m) : __popcll(m);
        s = qw_sum<HALF>(qw_select(m, w));
        const unsigned long long nm = __ballot(w * (double)c > s - 1.0);
        if (nm == m || (pass >= 2 && (HALF ? __popc((unsigned int)nm) : __popcll(nm)) >= c)) break;
        m = nm;
    }
    mask = m;
    return (s - 1.0) / (double)c;
}

// out_i = sum_j Arow[j] * v_j (lane = component i; v distributed one component per lane).
// Blocks of four columns whose v entries are all zero are skipped (their terms are exact
// zeros): the SPG direction is non-zero only on the union of two small supports.
template <int KQ>
__device__ __forceinline__ double qw_matvec(const double (&Arow)[KQ], double v)
{
    const unsigned long long nz = __ballot(v != 0.0);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < KQ; j += 4) {
        if ((nz >> j) & 0xfull) {
            a0 = fma(Arow[j], qw_readlane(v, j), a0);
            a1 = fma(Arow[j + 1], qw_readlane(v, j + 1), a1);
            a0 = fma(Arow[j + 2], qw_readlane(v, j + 2), a0);
            a1 = fma(Arow[j + 3], qw_readlane(v, j + 3), a1);
        }
    }
    return a0 + a1;
}
// Mirrored form (k <= 32): lane (i, h = lane >> 5) holds only columns 16h .. 16h+15 of row i
// (half the registers) and sums its half of the product; the halves are exchanged and added,
// so both mirror lanes end with the full (A v)_i.
__device__ __forceinline__ double qw_matvec_half(const double (&Ahalf)[16], double v, int lane)
{
    const unsigned long long nz = __ballot(v != 0.0);
    const bool hi = lane >= 32;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        if (((nz >> j) & 0xfull) | ((nz >> (16 + j)) & 0xfull)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double vlo = qw_readlane(v, j + u), vhi = qw_readlane(v, 16 + j + u);
                const double vj = hi ? vhi : vlo;
                if (u & 1) a1 = fma(Ahalf[j + u], vj, a1);
                else a0 = fma(Ahalf[j + u], vj, a0);
            }
        }
    }
    const double part = a0 + a1;
    const double other = __hiloint2double(__shfl_xor(__double2hiint(part), 32, 64),
                                          __shfl_xor(__double2loint(part), 32, 64));
    return hi ? other + part : part + other;       // columns 0..15 first in both lanes
}

// The same product on the DPP crossbar (k <= 32, default): v_fmac_f64 with row_newbcast:N
// multiplies lane N of the caller's ROW into the whole row -- one instruction per column instead
// of four v_readlane, their moves and two selects.  A row of 16 lanes only sees its own 16
// components of v, so rows 2 and 3 (the mirror half) first swap theirs (v_permlane16_swap):
//   row 0: components  0..15 x columns  0..15      row 2: components  0..15 x columns 16..31
//   row 1: components 16..31 x columns 16..31      row 3: components 16..31 x columns  0..15
// and the two halves of (A v)_i meet through v_permlane32_swap; both mirror lanes add the same
// two numbers.  Acol: the 16 entries of row `component` in this lane's column block.
__device__ __forceinline__ double qw_matvec_bcast(const double (&Acol)[16], double v, int lane)
{
    const bool odd_block = ((lane >> 4) == 1) || ((lane >> 4) == 2);      // this row sees columns 16..31
    qq_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned int)__double2loint(v), (unsigned int)__double2loint(v), false, false);
    qq_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned int)__double2hiint(v), (unsigned int)__double2hiint(v), false, false);
    // [0]: rows {0,0,2,2} of v, [1]: rows {1,1,3,3}; the mirror rows equal rows 0 and 1
    const double vs = odd_block ? __hiloint2double((int)hi[1], (int)lo[1]) : __hiloint2double((int)hi[0], (int)lo[0]);
    double a0 = 0.0, a1 = 0.0;
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %15 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %16 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %18 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(a0), "+v"(a1)
                 : "v"(vs), "v"(Acol[0]), "v"(Acol[1]), "v"(Acol[2]), "v"(Acol[3]), "v"(Acol[4]), "v"(Acol[5]),
                   "v"(Acol[6]), "v"(Acol[7]), "v"(Acol[8]), "v"(Acol[9]), "v"(Acol[10]), "v"(Acol[11]),
                   "v"(Acol[12]), "v"(Acol[13]), "v"(Acol[14]), "v"(Acol[15]));
    const double part = a0 + a1;
    double pa, pb;
    qq_xchg_d<true>(part, pa, pb);                 // lanes l and l ^ 32
    return pa + pb;
}

// MEM1: the non-monotone memory is 1 (spg.py:310 default, and always on the continuation path): the
// f_mem array -- 32 wave-uniform doubles, i.e. 64 SGPRs that hipcc spills to VGPR lanes around every
// reduction -- does not exist.
template <int KQ, bool MEM1 = false, bool LAZY = true>
__device__ __forceinline__ void qp_wave_body(const double *__restrict__ A /*[KQ][KQ]*/,
                                                 const double *__restrict__ B, long stride_j,
                                                 long stride_t, const double *__restrict__ bscale,
                                                 double *__restrict__ Z, int ldz, long n_fresh,
                                                 int k, aa_qp_params p, int *__restrict__ iters,
                                                 QpHeader *__restrict__ hdr,
                                                 const int *__restrict__ ovf_rows,
                                                 const QpCarry *__restrict__ ovf,
                                                 double *__restrict__ zslot /*[slot][KQ] or null*/,
                                                 const int *__restrict__ fresh_list = nullptr,
                                                 const unsigned int *__restrict__ count_ptr = nullptr,
                                                 int park_at = 1 << 30, unsigned int *__restrict__ n_parked = nullptr,
                                                 int *__restrict__ park_rows = nullptr,
                                                 QpCarry *__restrict__ park = nullptr,
                                                 QpLive lv = QpLive{0, 0, 0u, 0u, nullptr, nullptr},
                                                 int rst_b = 0, long rst_n = 0, unsigned int blk0 = 0u)
{
    // blk0: the first blk0 blocks of the grid do something else (k_qp_wave_ord)
    if (gridDim.y > 1) {                               // restarts side by side: blockIdx.y = slot
        const int rst = blockIdx.y;
        A += (long)rst * KQ * KQ;
        if (bscale) bscale += rst * 64;
        B += (long)rst * rst_b;
        Z += (long)rst * rst_b;
        hdr += rst;
        ovf_rows += rst * rst_n;
        ovf += rst * rst_n;
        if (iters) iters += rst * rst_n;
    }
    // park_at (continuation path): a sample still running after that many passes is written back
    // and appended to a second overflow list (n_parked / park_rows / park) -- the handful of
    // samples with hundreds of passes, which a later launch finishes beside the next pass over X
    constexpr bool HALF = KQ == 32;
    // latency-bound waves: take issue priority over the bandwidth-bound GEMM waves they may
    // share a SIMD with (the stragglers run concurrently with the Z'X pass)
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int comp = HALF ? (lane & 31) : lane;     // HALF: the upper half mirrors the lower
    const bool live = comp < k;
    // row `comp` of A in registers (mirrored form: each of the two lanes of a component keeps
    // one half of the row)
    double Arow[HALF ? 16 : KQ];
    // HALF: rows 1 and 2 of the wave take columns 16..31 (see qw_matvec_bcast)
    const int col0 = HALF ? ((((lane >> 4) == 1) || ((lane >> 4) == 2)) ? 16 : 0) : 0;
#pragma unroll
    for (int j = 0; j < (HALF ? 16 : KQ); ++j) Arow[j] = A[comp * KQ + col0 + j];
    auto matvec = [&](double v) -> double {
        if constexpr (HALF) return qw_matvec_bcast(Arow, v, lane);
        else return qw_matvec<KQ>(Arow, v);
    };
    constexpr int NMEM = MEM1 ? 1 : QW_MAXMEM;
    bool sq_ok;
    const double sq_lim = qp_sq_limit(p.epsilon_two, &sq_ok);
    const int mem = MEM1 ? 1 : (p.memory < 1 ? 1 : (p.memory > QW_MAXMEM ? QW_MAXMEM : p.memory));
    // n_fresh >= 0: process rows [0, n_fresh) from scratch; otherwise the overflow list
    // fresh_list: the first hdr->n_long entries of the sorted sample list, from scratch
    const bool fresh = n_fresh >= 0 || fresh_list != nullptr;
    const unsigned int count = count_ptr ? *count_ptr
                                         : (fresh_list ? hdr->n_long : (fresh ? (unsigned int)n_fresh : hdr->n_overflow));

    const unsigned int wave_id =
        (unsigned int)__builtin_amdgcn_readfirstlane((int)((blockIdx.x - blk0) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const unsigned int n_waves = (gridDim.x - blk0) * (blockDim.x >> 6);
    unsigned long long wv_total = 0ull;      // pass statistics of this wave: one atomic pair at exit
    int wv_max = 0;
    unsigned int static_slot = wave_id;
    // next slot of this wave, or 0xffffffff: static stride (mode 0 / 2), tickets (mode 1)
    auto next_slot = [&]() -> unsigned int {
        if (lv.mode == 3) {                            // plain tickets (qp_wave_queue): a wave takes the next slot when it is free
            unsigned int tk = 0u;
            if (lane == 0) tk = atomicAdd(&hdr->next_overflow, 1u);
            tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
            return tk < count ? tk : 0xffffffffu;
        }
        if (lv.mode != 1) {
            while (static_slot < count) {
                const unsigned int sl = static_slot;
                static_slot += n_waves;
                if (lv.mode == 2 && lv.done[sl] == lv.epoch) continue;      // a consumer finished it
                return sl;
            }
            return 0xffffffffu;
        }
        unsigned int tk = 0u;
        if (lane == 0) tk = atomicAdd(&hdr->next_overflow, 1u);
        tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
        if (tk >= lv.n) return 0xffffffffu;          // (ready[] has lv.n entries)
        // waiting waves stay out of the way: lowest issue priority, one flag load per nap (0.4 us at
        // first, 3 us after the first 32), the producers' counters only every eighth nap
        __builtin_amdgcn_s_setprio(0);
        for (int spin = 0; spin < 400000; ++spin) {                    // bounded: the clean-up launch covers a give-up
            if (qp_load_agent(&lv.ready[tk]) == lv.epoch) {
                __builtin_amdgcn_s_setprio(3);
                return tk;
            }
            if ((spin & 7) == 7 &&
                __hip_atomic_load(&hdr->waves_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= lv.producer_waves) {
                // every producer has exited, and published before it did: the slot exists or never will
                if (tk >= __hip_atomic_load(&hdr->n_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    return 0xffffffffu;
            }
            if (spin < 32) __builtin_amdgcn_s_sleep(16);
            else __builtin_amdgcn_s_sleep(127);
        }
        return 0xffffffffu;
    };
    for (unsigned int slot = next_slot(); slot != 0xffffffffu; slot = next_slot()) {
        const bool shared = lv.mode == 1;           // the producer kernel is still running: coherent loads
        const long row = fresh_list ? (long)fresh_list[slot]
                                    : (fresh ? (long)slot : (long)(shared ? qp_load_agent(&ovf_rows[slot]) : ovf_rows[slot]));

        double x = live ? (shared ? qp_load_agent(&Z[row * ldz + comp]) : Z[row * ldz + comp]) : 0.0;
        const double b = live ? -B[comp * stride_j + row * stride_t] * (bscale ? bscale[comp] : 1.0) : 0.0;
        double f, alpha = 1.0, fmem[NMEM];
        int n_iter, n_feval;
        unsigned long long support = 0ull, support_r = 0ull;   // latest direction / residual supports
#pragma unroll
        for (int i = 0; i < NMEM; ++i) fmem[i] = NAN;
        if (fresh) {
            const double t0 = qw_threshold<HALF>(live ? x : -INFINITY, comp, support);
            x = live ? fmax(x - t0, 0.0) : 0.0;
            support_r = support;
        }
        double g = matvec(x) + b;
        if (fresh) {
            double xg, xb;
            qw_sum2<HALF>(x * g, x * b, lane, xg, xb);
            f = 0.5 * (xg + xb);
            n_iter = 0;
            n_feval = 1;
        } else if (shared) {
            const double *cd = reinterpret_cast<const double *>(&ovf[slot]);
            alpha = qp_load_agent(cd);
            f = qp_load_agent(cd + 1);
            const long long packed = __double_as_longlong(qp_load_agent(cd + 2));
            n_iter = (int)(packed & 0xffffffffll);
            n_feval = (int)(packed >> 32);
        } else {
            const QpCarry cr = ovf[slot];
            f = cr.f;
            alpha = cr.alpha;
            n_iter = cr.n_iter;
            n_feval = cr.n_feval;
        }

        bool parked = false;
        // The stopping test of a pass (spg.py:378-396: r = P(x - g) - x at the new point, ||r||_2 < eps_two
        // or ||r||_inf < eps_one) costs a projection, a reduction and a compare -- a quarter of the pass --
        // and on all but the last one or two passes of a sample its answer is "no".  Round 4: the test of
        // pass i is decided at the top of pass i + 1, where d = P(x - alpha g) - x at the same point has
        // just been formed: for a feasible x the projected-gradient map satisfies
        //     ||P(x - g) - x||_2  >=  ||P(x - alpha g) - x||_2 / max(1, alpha)
        // (||d(alpha)|| is nondecreasing and ||d(alpha)|| / alpha nonincreasing in alpha), and
        // ||r||_inf >= ||r||_2 / sqrt(k); so  <d, d>  >  4 max(1, alpha)^2 max(eps_two^2, k eps_one^2)
        // PROVES that neither stopping test fires (a factor 2 in the norm against rounding), and the
        // residual projection is skipped.  Whenever the certificate fails the reference's test runs as
        // before: same decisions, same pass counts, same iterates -- 25 % fewer instructions on the
        // dependent chain of the longest sample.
        const double cert = 4.0 * fmax(p.epsilon_two * p.epsilon_two, (double)k * p.epsilon_one * p.epsilon_one);
        auto stop_test = [&]() -> bool {
            const double tr = qw_threshold<HALF>(live ? x - g : -INFINITY, comp, support_r);
            const double r = live ? fmax(x - g - tr, 0.0) - x : 0.0;
            const double r2 = qw_sum<HALF>(r * r);
            // max |r| < epsilon_one  <=>  no lane has |r| >= epsilon_one (no reduction)
            const bool rinf_small = __ballot(!(fabs(r) < p.epsilon_one)) == 0ull;
            return (sq_ok ? r2 < sq_lim : sqrt(r2) < p.epsilon_two) || rinf_small;
        };
        bool pending = false;                      // the stopping test of the previous pass is still open
        // `guard_w` bounds the loop even if the arithmetic goes non-finite
        for (int guard_w = 0; guard_w < p.max_iterations + 2; ++guard_w) {
            if (n_iter == 0) {
                if (p.alpha_min <= p.alpha0 && p.alpha0 <= p.alpha_max) {
                    alpha = p.alpha0;
                } else {
                    const double t1 = qw_threshold<HALF>(live ? x - g : -INFINITY, comp, support_r);
                    double ainv = qw_max<HALF>(live ? fabs(fmax(x - g - t1, 0.0) - x) : 0.0);
                    if (fabs(ainv) < 1e-12) ainv = 1.0;
                    alpha = fmin(fmax(p.alpha_min, 1.0 / ainv), p.alpha_max);
                }
            }
            const double td = qw_threshold<HALF>(live ? x - alpha * g : -INFINITY, comp, support);
            const double d = live ? fmax(x - alpha * g - td, 0.0) - x : 0.0;
            double delta, dd;
            qw_sum2<HALF>(d * g, d * d, lane, delta, dd);
            if (pending) {
                const double am = fmax(1.0, alpha);
                if (!(dd > cert * am * am) && stop_test()) break;      // converged at the point the last pass left
                pending = false;
            }
            const double Ad = matvec(d);
            const double dAd = qw_sum<HALF>(d * Ad);

            double f_max = f;
            if constexpr (!MEM1) {
                if (mem > 1) {                     // memory == 1 (the default): f_max = f
#pragma unroll
                    for (int i = NMEM - 1; i > 0; --i)
                        if (i < mem) fmem[i] = fmem[i - 1];
                    fmem[0] = f;
#pragma unroll
                    for (int i = 1; i < NMEM; ++i)
                        if (i < mem && fmem[i] > f_max) f_max = fmem[i];
                }
            }

            double lam = 1.0;
            double f_new = f + lam * delta + 0.5 * lam * lam * dAd;
            n_feval += 1;
            int guard = 0;
            while (f_new > f_max + p.gamma * lam * delta && guard < 200) {
                const double tmp = -0.5 * lam * lam * delta / (f_new - f - lam * delta);
                lam = (p.sigma_one <= tmp && tmp <= p.sigma_two * lam) ? tmp : 0.5 * lam;
                f_new = f + lam * delta + 0.5 * lam * lam * dAd;
                n_feval += 1;
                ++guard;
                if (fabs(lam) < p.lambda_min) break;
            }
            x = fma(lam, d, x);
            g = fma(lam, Ad, g);
            const double sksk = lam * lam * dd;
            const double beta = lam * (lam * dAd);
            alpha = (beta <= 0.0) ? p.alpha_max : fmin(p.alpha_max, fmax(p.alpha_min, sksk / beta));
            f = f_new;
            n_feval += 1;
            n_iter += 1;
            // the caps end the sample whatever the stopping test says (spg.py:391-396: same iterate)
            if (n_feval > p.max_feval || n_iter >= p.max_iterations) break;
            if constexpr (!LAZY) {                 // (A/B switch qp_wave_lazy = 0: the test after every pass, as before round 4)
                if (stop_test()) break;
            } else if (n_iter >= park_at) {
                if (stop_test()) break;            // decided here: a parked sample starts its next launch clean
            } else {
                pending = true;
            }
            if (n_iter >= park_at) {
                parked = true;
                break;
            }
        }
        if (live && lane == comp) {
            if (zslot) zslot[(size_t)slot * KQ + comp] = x;     // deferred commit (launch_qp_tail_fixup)
            else Z[row * ldz + comp] = x;
        }
        if (parked) {
            if (lane == 0) {
                const unsigned int s2 = atomicAdd(n_parked, 1u);
                park_rows[s2] = (int)row;
                QpCarry cr;
                cr.alpha = alpha;
                cr.f = f;
                cr.n_iter = n_iter;
                cr.n_feval = n_feval;
                park[s2] = cr;
            }
            continue;
        }
        if (lane == 0 && iters) iters[row] = n_iter;
        if (lane == 0 && lv.mode == 1) lv.done[slot] = lv.epoch;    // read by the clean-up launch (next kernel)
        wv_total += (unsigned long long)n_iter;
        wv_max = n_iter > wv_max ? n_iter : wv_max;
    }
    if (lane == 0 && wv_total) {
        atomicAdd(&hdr->total_passes, wv_total);
        atomicMax(&hdr->max_passes, (unsigned long long)wv_max);
    }
}

#define QW_ARGS const double *__restrict__ A, const double *__restrict__ B, long stride_j, long stride_t,            \
                const double *__restrict__ bscale, double *__restrict__ Z, int ldz, long n_fresh, int k,            \
                aa_qp_params p, int *__restrict__ iters, QpHeader *__restrict__ hdr,                                \
                const int *__restrict__ ovf_rows, const QpCarry *__restrict__ ovf, double *__restrict__ zslot,      \
                const int *__restrict__ fresh_list = nullptr, const unsigned int *__restrict__ count_ptr = nullptr, \
                int park_at = 1 << 30, unsigned int *__restrict__ n_parked = nullptr,                               \
                int *__restrict__ park_rows = nullptr, QpCarry *__restrict__ park = nullptr,                        \
                QpLive lv = QpLive{0, 0, 0u, 0u, nullptr, nullptr}, int rst_b = 0, long rst_n = 0
#define QW_PASS A, B, stride_j, stride_t, bscale, Z, ldz, n_fresh, k, p, iters, hdr, ovf_rows, ovf, zslot,           \
                fresh_list, count_ptr, park_at, n_parked, park_rows, park, lv, rst_b, rst_n
template <int KQ, bool MEM1 = false, bool LAZY = true>
__global__ __launch_bounds__(256) void k_qp_wave(QW_ARGS) { qp_wave_body<KQ, MEM1, LAZY>(QW_PASS); }
// the live consumers (QpLive mode 1): blocks of 16 waves that are launched with a whole CU's LDS
// as (unused) dynamic shared memory while k_qp_quad asks for 1/12 of it per wave -- LDS becomes the
// resource that keeps the two kernels on DIFFERENT CUs: the latency-bound chains of the consumers
// (four per SIMD) do not share issue slots with the MFMA-heavy waves of k_qp_quad, which stretched
// a consumer's pass from 0.9 to ~1.8 us when the two kernels were mixed on every SIMD
__global__ __launch_bounds__(1024) void k_qp_wave_live(QW_ARGS) { qp_wave_body<32, true>(QW_PASS); }
#undef QW_ARGS
#undef QW_PASS
// continuation launches (memory == 1 by construction of the callers)
#define QW32_LAUNCH(...)                                                              \
    do {                                                                              \
        if (g_qp_wave_mem1 && g_qp_wave_lazy) hipLaunchKernelGGL((k_qp_wave<32, true, true>), __VA_ARGS__);  \
        else if (g_qp_wave_mem1) hipLaunchKernelGGL((k_qp_wave<32, true, false>), __VA_ARGS__);   \
        else hipLaunchKernelGGL((k_qp_wave<32, false>), __VA_ARGS__);                 \
    } while (0)

// max_iterations <= 0: the reference's loop body never runs and x = P(x0) is returned.
template <int KQ>
__global__ __launch_bounds__(256) void k_qp_project_only(const double *__restrict__ Z0,
                                                         double *__restrict__ Z, int ldz, long n,
                                                         int k)
{
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    double x[KQ], g[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
        x[i] = (i < k) ? Z0[row * ldz + i] : 0.0;
        g[i] = 0.0;
    }
    typename QpMask<KQ>::type support = 0;
    const double t0 = qp_project_threshold<KQ>(x, g, 0.0, k, support);
#pragma unroll
    for (int i = 0; i < KQ; ++i)
        if (i < k) Z[row * ldz + i] = fmax(x[i] - t0, 0.0);
}

// ---------------------------------------------------------------------------
// ROW kernel (default for k <= 32): ONE DPP ROW (16 lanes) PER SAMPLE, CPL = 1 or 2
// components per lane, four independent samples per wave, several waves per SIMD.
//
// Why a third mapping: the lane-per-sample kernel needs the whole sample state in one
// lane's registers (one wave per SIMD, ~18 us per SPG pass), so it cannot finish the
// heavy tail (a pass cap hands ~4 % of the samples, ~20 % of the passes, to the
// wave-per-sample kernel) and its duration is bounded below by cap x 18 us; the
// wave-per-sample kernel has the short pass (1.4 us) but runs one sample per wave.  A
// 16-lane row keeps the short critical path -- every reduction is a four-step xor
// butterfly on the DPP crossbar (quad_perm, quad_perm, row_half_mirror, row_mirror; both
// partners add the same two numbers, so all 16 lanes end with identical bits and every
// row-uniform decision is taken identically by every lane of the row) -- while a wave
// carries four samples and a SIMD several waves, so the issue slots one latency-bound
// chain leaves empty are filled by others.  Rows pull samples from the global queue
// independently (longest first, k_qp_order_*), run them to completion (no pass cap, no
// second kernel), and waves that hold a long-running sample raise their issue priority, so
// the longest chain of the update runs at its own latency from t = 0 while the short
// samples fill the machine around it.
//
// Per SPG pass of a row: direction projection (support-mask Michelot, one sum + one count
// reduction per round), d, <d,g>, <d,d>; the direction goes through LDS (one 16-byte write
// per lane, KQ/2 broadcast 16-byte reads) and A d is CPL*KQ FMAs against this lane's rows
// of A (registers); d'Ad, the scalar Armijo loop, x, g, the BB step, the residual
// projection and its norm: seven DPP reductions per pass.  A sample that starts puts P(z0)
// through the same mat-vec slot (g = A x + b), like the lane kernel.  f_mem (spg.py:310,
// 341-344) is distributed over the row (lane r holds f_mem[r]): memory up to 16.
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double qr_xchg(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int qr_xchg_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// 0xB1 = quad_perm [1,0,3,2], 0x4E = quad_perm [2,3,0,1], 0x141 = row_half_mirror (the
// values are quad-uniform by then, so i <-> 7 - i exchanges the two quads), 0x140 = row_mirror
__device__ __forceinline__ double qr_sum(double v)
{
    v += qr_xchg<0xB1>(v);
    v += qr_xchg<0x4E>(v);
    v += qr_xchg<0x141>(v);
    v += qr_xchg<0x140>(v);
    return v;
}
__device__ __forceinline__ double qr_max(double v)
{
    v = fmax(v, qr_xchg<0xB1>(v));
    v = fmax(v, qr_xchg<0x4E>(v));
    v = fmax(v, qr_xchg<0x141>(v));
    v = fmax(v, qr_xchg<0x140>(v));
    return v;
}
__device__ __forceinline__ int qr_sum_i(int v)
{
    v += qr_xchg_i<0xB1>(v);
    v += qr_xchg_i<0x4E>(v);
    v += qr_xchg_i<0x141>(v);
    v += qr_xchg_i<0x140>(v);
    return v;
}
// does any lane of this row (lanes rowshift .. rowshift + 15) satisfy p?
__device__ __forceinline__ bool qr_any(bool p, int rowshift)
{
    return ((__ballot(p) >> rowshift) & 0xffffull) != 0ull;
}

// Threshold of the projection of the row-distributed vector w (CPL components per lane;
// padding components carry a hugely negative w).  Same support-mask fixed point, same break
// rule and the same closed form as qp_project_threshold / qw_threshold; `mask` holds the
// support bits of this lane's own components.  Rows leave the loop independently.
template <int CPL>
__device__ __forceinline__ double qr_threshold(const double (&w)[CPL], unsigned &mask, int rowshift,
                                               unsigned int &rounds)
{
    unsigned m = mask;
    int c = qr_sum_i(__popc(m));
    double s = 0.0;
    for (int pass = 0; pass < 32 * CPL + 8; ++pass) {
        rounds += 1u;
        if (c == 0) {                              // cold start, or the warm guess emptied
            double mx = w[0];
#pragma unroll
            for (int q = 1; q < CPL; ++q) mx = fmax(mx, w[q]);
            const double t0 = qr_max(mx) - 1.0;
            m = 0u;
#pragma unroll
            for (int q = 0; q < CPL; ++q)
                if (w[q] > t0) m |= 1u << q;
            c = qr_sum_i(__popc(m));
        }
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < CPL; ++q) part += ((m >> q) & 1u) ? w[q] : 0.0;
        s = qr_sum(part);
        const double sm1 = s - 1.0, cd = (double)c;
        unsigned nm = 0u;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (w[q] * cd > sm1) nm |= 1u << q;
        const int cn = qr_sum_i(__popc(nm));
        const bool same = !qr_any(nm != m, rowshift);
        if (same || (pass >= 2 && cn >= c)) break;
        m = nm;
        c = cn;
    }
    mask = m;
    return (s - 1.0) / (double)c;
}

#define QR_MAXMEM 16
template <int CPL>
__global__ __launch_bounds__(64) void k_qp_row(const double *__restrict__ A /*[lda][lda], zero padded*/,
                                               int lda, const double *__restrict__ B, long stride_j,
                                               long stride_t, const double *__restrict__ bscale,
                                               double *__restrict__ Z, int ldz, long n, int k,
                                               aa_qp_params p, int *__restrict__ iters,
                                               QpHeader *__restrict__ hdr,
                                               const int *__restrict__ perm, int hot_passes, int prof, int QR_CHUNK,
                                               int pass_cap, int *__restrict__ ovf_rows,
                                               QpCarry *__restrict__ ovf)
{
    constexpr int KQ = 16 * CPL;
    // hybrid update: the first n_long samples of the sorted list (predicted long) are being
    // solved by the wave-per-sample kernel on the side stream; this kernel takes the rest
    const long list0 = perm ? (long)hdr->n_long : 0;
    const long nq = n - list0;                     // entries of the list this kernel works off
    __shared__ __attribute__((aligned(16))) double vb[4][KQ];
    const int lane = threadIdx.x, r = lane & 15, rowid = lane >> 4, rowshift = lane & 48;
    const int comp0 = r * CPL;                     // this lane owns components comp0 .. comp0 + CPL - 1
    // rows comp0.. of A in registers (A is symmetric: row = column)
    double Ar[CPL][KQ];
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
        for (int j = 0; j < KQ; ++j) Ar[q][j] = A[(comp0 + q) * lda + j];
    bool live[CPL];
    double bs[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        live[q] = comp0 + q < k;
        bs[q] = (bscale && live[q]) ? bscale[comp0 + q] : 1.0;
    }
    const int mem = p.memory < 1 ? 1 : (p.memory > QR_MAXMEM ? QR_MAXMEM : p.memory);

    double x[CPL], g[CPL], d[CPL], v[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) x[q] = g[q] = d[q] = v[q] = 0.0;
    double f = 0.0, alpha = 1.0, delta = 0.0, dd = 0.0;
    double fm = NAN;                               // lane r: f_mem[r]
    int n_iter = 0, n_feval = 0, predicted = 0;
    long row = -1;
    bool active = false;
    unsigned sup = 0u, sup_r = 0u;                 // supports of the latest direction / residual projection
    int prio = 0;
    unsigned int dbg_rounds = 0u, dbg_trips = 0u;  // qp_profile: Michelot rounds / trips of this lane
    unsigned long long st_total = 0ull;            // pass statistics of this row, one atomic per wave at exit
    int st_max = 0;

    // Work distribution.  The queue is the sample list in longest-first order; a wave takes
    // QR_CHUNK tickets per atomic and always has the NEXT chunk's atomic in flight, and every
    // row holds, next to the sample it works on, the one it will work on next with its z0 and
    // b already loaded -- so neither the atomic's round trip nor the two global loads of a
    // start-up sit on the critical path of the three other rows of the wave.
    long nxt = -1;                                 // this row's next sample (prefetched), -1: none
    double zn[CPL], bn[CPL];
    int npred = 0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) zn[q] = bn[q] = 0.0;
    long q_base = 0;                               // wave-uniform: local chunk [q_base, q_base + q_left)
    int q_left = 0;
    unsigned int q_ahead = 0u;                     // ticket of the chunk whose atomic is in flight
    bool have_ahead = false, drained = false;      // wave-uniform
    int s_next = 0;                                // static assignment: next position of this wave's list
    const long long t_start = prof ? clock64() : 0;

    for (long trip = 0; trip < (1L << 26); ++trip) {       // watchdog bound only
        // a wave that carries a long-running sample stops pulling work (its remaining rows idle
        // once their local entries are used up) and takes issue priority: the longest chains of
        // the update then advance at one pass per ~1500 cycles while short samples fill the
        // rest of the machine
        const bool hot_wave = __any(active && (n_iter >= hot_passes || predicted >= 2 * hot_passes));
        if ((hot_wave ? 1 : 0) != prio) {
            if (hot_wave) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
            prio = hot_wave ? 1 : 0;
        }
        // ---- rows without a sample take their prefetched one
        bool starting = false;
        if (!active && nxt >= 0) {                 // zn / bn stay valid until the pop at the end of the trip
            starting = true;
            row = nxt;
            predicted = npred;
            nxt = -1;
        }
        const bool busy = __any(active || starting);
        if (busy) {
        dbg_trips += 1u;

        if (starting) {
            // ---- start-up: x = P(z0)                                   (spg.py:298-300)
            double w0[CPL];
            unsigned m0 = 0u;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                w0[q] = live[q] ? zn[q] : -QP_PAD;
                if (live[q] && zn[q] > 0.0) m0 |= 1u << q;     // z0 is (nearly) feasible: warm support
            }
            const double t0 = qr_threshold<CPL>(w0, m0, rowshift, dbg_rounds);
            sup = m0;
            sup_r = m0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                x[q] = live[q] ? fmax(w0[q] - t0, 0.0) : 0.0;
                v[q] = x[q];
            }
        } else if (active) {
            // ---- one pass of the loop at spg.py:318-396, up to the search direction
            if (n_iter == 0) {
                if (p.alpha_min <= p.alpha0 && p.alpha0 <= p.alpha_max) {
                    alpha = p.alpha0;
                } else {
                    double w1[CPL];
#pragma unroll
                    for (int q = 0; q < CPL; ++q) w1[q] = x[q] - g[q];
                    const double t1 = qr_threshold<CPL>(w1, sup_r, rowshift, dbg_rounds);
                    double am = 0.0;
#pragma unroll
                    for (int q = 0; q < CPL; ++q) am = fmax(am, fabs(fmax(w1[q] - t1, 0.0) - x[q]));
                    double ainv = qr_max(am);
                    if (fabs(ainv) < 1e-12) ainv = 1.0;
                    alpha = fmin(fmax(p.alpha_min, 1.0 / ainv), p.alpha_max);
                }
            }
            double w[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) w[q] = x[q] - alpha * g[q];
            const double td = qr_threshold<CPL>(w, sup, rowshift, dbg_rounds);
            double pg = 0.0, pd = 0.0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                d[q] = fmax(w[q] - td, 0.0) - x[q];
                v[q] = d[q];
                pg = fma(d[q], g[q], pg);
                pd = fma(d[q], d[q], pd);
            }
            delta = qr_sum(pg);
            dd = qr_sum(pd);
        } else {
#pragma unroll
            for (int q = 0; q < CPL; ++q) v[q] = 0.0;
        }

        // ---- A v for the four rows of the wave: v through LDS, rows of A from registers
        double Av[CPL];
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < CPL; ++q) vb[rowid][comp0 + q] = v[q];
            // one wave per block: LDS executes a wave's operations in order; the fences only
            // keep the compiler from moving the reads above the writes of the other lanes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double acc[CPL][4];
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[q][u] = 0.0;
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const double vj = vb[rowid][j];
#pragma unroll
                for (int q = 0; q < CPL; ++q) acc[q][j & 3] = fma(Ar[q][j], vj, acc[q][j & 3]);
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) Av[q] = (acc[q][0] + acc[q][1]) + (acc[q][2] + acc[q][3]);
        }

        if (starting) {
            // ---- g = A x + b; f = x'(g + b)/2                              (spg.py:302-315)
            double pxg = 0.0, pxb = 0.0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                g[q] = Av[q] + bn[q];
                pxg = fma(x[q], g[q], pxg);
                pxb = fma(x[q], bn[q], pxb);
            }
            f = 0.5 * (qr_sum(pxg) + qr_sum(pxb));
            n_feval = 1;
            n_iter = 0;
            fm = NAN;
            active = true;                         // its first pass runs in the next trip
        } else if (active) {
            double pq = 0.0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) pq = fma(d[q], Av[q], pq);
            const double dAd = qr_sum(pq);
            // non-monotone reference value (spg.py:341-344): roll, store, nanmax
            double f_max = f;
            if (mem > 1) {
                const double up = qr_xchg<0x111>(fm);          // row_shr:1 -> f_mem[r - 1]
                fm = r == 0 ? f : (r < mem ? up : NAN);
                f_max = qr_max(fm == fm ? fm : -INFINITY);
            }
            double lam = 1.0;
            double f_new = f + lam * delta + 0.5 * lam * lam * dAd;
            n_feval += 1;
            int guard = 0;
            while (f_new > f_max + p.gamma * lam * delta && guard < 200) {
                const double tmp = -0.5 * lam * lam * delta / (f_new - f - lam * delta);
                lam = (p.sigma_one <= tmp && tmp <= p.sigma_two * lam) ? tmp : 0.5 * lam;
                f_new = f + lam * delta + 0.5 * lam * lam * dAd;
                n_feval += 1;
                ++guard;
                if (fabs(lam) < p.lambda_min) break;
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                x[q] = fma(lam, d[q], x[q]);
                g[q] = fma(lam, Av[q], g[q]);
            }
            const double sksk = lam * lam * dd;
            const double beta = lam * (lam * dAd);
            alpha = (beta <= 0.0) ? p.alpha_max : fmin(p.alpha_max, fmax(p.alpha_min, sksk / beta));
            f = f_new;
            n_feval += 1;

            double wr[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) wr[q] = x[q] - g[q];
            const double tr = qr_threshold<CPL>(wr, sup_r, rowshift, dbg_rounds);
            double pr2 = 0.0;
            bool big = false;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const double rr = fmax(wr[q] - tr, 0.0) - x[q];
                pr2 = fma(rr, rr, pr2);
                big = big || !(fabs(rr) < p.epsilon_one);
            }
            const double r2 = qr_sum(pr2);
            const bool rinf_small = !qr_any(big, rowshift);
            n_iter += 1;
            const bool conv = (sqrt(r2) < p.epsilon_two) || rinf_small;
            const bool finished = conv || n_feval > p.max_feval || n_iter >= p.max_iterations;
            if (finished || n_iter >= pass_cap) {
#pragma unroll
                for (int q = 0; q < CPL; ++q)
                    if (live[q]) Z[row * ldz + comp0 + q] = x[q];
                if (finished) {
                    if (r == 0 && iters) iters[row] = n_iter;
                    st_total += (unsigned long long)n_iter;
                    st_max = n_iter > st_max ? n_iter : st_max;
                } else if (r == 0) {
                    // an unexpectedly long sample: hand it, with its SPG state, to the
                    // wave-per-sample kernel (1 us per pass instead of ~3)
                    const unsigned int slot = atomicAdd(&hdr->n_overflow, 1u);
                    ovf_rows[slot] = (int)row;
                    QpCarry cr;
                    cr.alpha = alpha;
                    cr.f = f;
                    cr.n_iter = n_iter;
                    cr.n_feval = n_feval;
                    ovf[slot] = cr;
                }
                active = false;
            }
        }
        }   // busy
        // ---- rows without a prefetched sample pop the local chunk
        {
            const bool need = nxt < 0 && !drained;
            const unsigned long long nbm = __ballot(need && r == 0);
            if (nbm != 0ull) {
                if (QR_CHUNK == 0) {
                    // static assignment, no atomics at all: wave w owns the list positions
                    // w, w + W, w + 2W, ... (W waves).  The list is in longest-first order, so
                    // every wave gets the same mix of long and short samples (its first one from
                    // the W longest, its second from the next W, ...), and the four rows of the
                    // wave share that list dynamically.
                    q_base = 0;
                    q_left = 4;
                } else
                if (q_left == 0) {
                    if (have_ahead) {                       // the chunk requested a while ago
                        const long t0 = (long)(unsigned int)__builtin_amdgcn_readfirstlane((int)q_ahead);
                        have_ahead = false;
                        q_base = t0;
                        q_left = t0 >= nq ? 0 : (nq - t0 < QR_CHUNK ? (int)(nq - t0) : QR_CHUNK);
                        if (q_left == 0) drained = true;
                    }
                    if (q_left == 0 && !drained && !hot_wave) {   // first trip (or after a hot phase)
                        unsigned int t1 = 0u;
                        if (lane == 0) t1 = atomicAdd(&hdr->next_row, (unsigned int)QR_CHUNK);
                        const long t0 = (long)(unsigned int)__builtin_amdgcn_readfirstlane((int)t1);
                        q_base = t0;
                        q_left = t0 >= nq ? 0 : (nq - t0 < QR_CHUNK ? (int)(nq - t0) : QR_CHUNK);
                        if (q_left == 0) drained = true;
                    }
                    if (q_left > 0 && !hot_wave) {          // request the chunk after this one
                        if (lane == 0) q_ahead = atomicAdd(&hdr->next_row, (unsigned int)QR_CHUNK);
                        have_ahead = true;
                    }
                }
                const int rank = __popcll(nbm & ((1ull << rowshift) - 1ull));
                long idx = list0 + q_base + rank;
                bool take = need && rank < q_left;
                if (QR_CHUNK == 0) {
                    idx = list0 + (long)blockIdx.x + (long)(s_next + rank) * (long)gridDim.x;
                    take = need && idx < n;
                }
                if (take) {
                    nxt = perm ? (long)perm[idx] : idx;
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        zn[q] = live[q] ? Z[nxt * ldz + comp0 + q] : 0.0;
                        bn[q] = live[q] ? -B[(comp0 + q) * stride_j + nxt * stride_t] * bs[q] : QP_PAD;
                    }
                    npred = (perm && iters) ? iters[nxt] : 0;      // pass count of the previous update
                }
                const int served = __popcll(nbm) < q_left ? __popcll(nbm) : q_left;
                q_base += served;
                q_left -= served;
                if (QR_CHUNK == 0) {
                    s_next += __popcll(nbm);
                    q_left = 0;
                    if (list0 + (long)blockIdx.x + (long)s_next * (long)gridDim.x >= n) drained = true;
                }
            }
        }
        if (!__any(active || nxt >= 0) && (drained || (q_left == 0 && !have_ahead && hot_wave))) break;
    }
    {   // statistics: the four rows' totals combined in lane 0, one atomic pair per wave
        unsigned long long tot = 0ull;
        int mx = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)st_total, 16 * q);
            const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(st_total >> 32), 16 * q);
            tot += ((unsigned long long)hi << 32) | lo;
            const int m = __builtin_amdgcn_readlane(st_max, 16 * q);
            mx = m > mx ? m : mx;
        }
        if (lane == 0 && tot) {
            atomicAdd(&hdr->total_passes, tot);
            atomicMax(&hdr->max_passes, (unsigned long long)mx);
        }
    }
    if (prof && r == 0) {                          // per row: Michelot rounds, trips of its wave
        atomicAdd(&hdr->dbg_rounds, dbg_rounds);
        if (lane == 0) atomicAdd(&hdr->dbg_trips, dbg_trips);
        if (lane == 0) atomicAdd(&hdr->dbg_waves, 1u);
        if (lane == 0) {
            // QpDebug area (byte 64 of the scratch): [0] max cycles of a wave, [1] its trips packed,
            // [2] max trips of a wave, [3] sum of cycles, [4] cycles of the wave with most trips
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(hdr) + 64);
            const unsigned long long cyc = (unsigned long long)(clock64() - t_start);
            atomicMax(&dbg[0], cyc);
            atomicMax(&dbg[2], (unsigned long long)dbg_trips);
            atomicAdd(&dbg[3], cyc);
            atomicMax(&dbg[4], ((unsigned long long)dbg_trips << 40) | (cyc & ((1ull << 40) - 1ull)));
            atomicMax(&dbg[1], (cyc << 20) | (unsigned long long)(dbg_trips & 0xfffffu));
        }
    }
}

// ---------------------------------------------------------------------------
// FOUR LANES PER SAMPLE, laid out as the f64 matrix cores want their operands (round 2).
//
// v_mfma_f64_16x16x4 takes its K x N operand with lane l holding (k = l>>4, n = l&15) and leaves
// the M x N result with lane l, register r holding (m = 4r + (l>>4), n = l&15).  Let the N index
// be the SAMPLE (16 per wave) and K / M the component: lane (s = l&15, q = l>>4) of sample slot s
// then supplies, in step t, component 4t + q of the direction, and receives components 4r + q
// (+16 per further M tile) of A d -- the SAME residue class q mod 4 on both sides.  So with
// component 4j + q of every vector (x, g, d, A d) in register j of lane (s, q), the k x k
// mat-vec of 16 samples is KQ/4 * KQ/16 matrix instructions on registers as they are: no LDS
// round trip, no transposition, no broadcast loads (the lane-per-sample kernel spends 17 800 of
// its 43 500 cycles per trip on exactly that), and A's operand tiles stay in KQ/2 registers.
// Element-wise work costs the same instructions per sample as one lane per sample (a VALU
// instruction covers 64 sample-components either way); reductions are KQ/4 in-lane terms and a
// two-step butterfly over the four lanes of a sample (v_permlane16_swap / v_permlane32_swap,
// gfx950: lanes l, l^16, l^32, l^48).  Both partners of a step add the same two numbers, so the
// four lanes hold identical bits and take every sample-uniform decision identically.
// A wave batches 16 samples instead of 64, so the batching waste of the divergent pass counts
// shrinks as well.  All cross-lane operations sit in wave-uniform control flow (idle slots run
// along with their results discarded).  Samples that reach the pass cap are parked for the
// wave-per-sample kernel exactly as in k_qp.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double qq_sum(double v)      // over the four lanes of a sample
{
    double a, b;
    qq_xchg_d<false>(v, a, b);
    v = a + b;
    qq_xchg_d<true>(v, a, b);
    return a + b;
}
__device__ __forceinline__ double qq_max(double v)
{
    double a, b;
    qq_xchg_d<false>(v, a, b);
    v = fmax(a, b);
    qq_xchg_d<true>(v, a, b);
    return fmax(a, b);
}
__device__ __forceinline__ unsigned int qq_sum_u(unsigned int v)
{
    unsigned int a, b;
    qq_xchg<false>(v, a, b);
    v = a + b;
    qq_xchg<true>(v, a, b);
    return a + b;
}

// Threshold of the projection of w (this lane's J components of it) for the samples with `live`
// set; same fixed point, same break rules as qp_project_threshold.  One four-word reduction per
// Michelot round: the sum and size of the NEXT support travel with the "support changed" flag.
template <int J>
__device__ __forceinline__ double qq_threshold(const double (&w)[J], unsigned int &mask, bool live)
{
    unsigned int m = mask;
    bool done = !live;
    double sl = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) sl += ((m >> j) & 1u) ? w[j] : 0.0;
    double s = qq_sum(sl);
    unsigned int c = qq_sum_u((unsigned int)__popc(m));
    if (__any(!done && c == 0u)) {                 // cold start (a sample's first projection)
        double mx = w[0];
#pragma unroll
        for (int j = 1; j < J; ++j) mx = fmax(mx, w[j]);
        mx = qq_max(mx);
        if (!done && c == 0u) {
            const double t0 = mx - 1.0;
#pragma unroll
            for (int j = 0; j < J; ++j) m |= (w[j] > t0) ? (1u << j) : 0u;
        }
        sl = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) sl += ((m >> j) & 1u) ? w[j] : 0.0;
        s = qq_sum(sl);
        c = qq_sum_u((unsigned int)__popc(m));
    }
    for (int pass = 0; pass < 8 * J + 8; ++pass) {
        const double sm1 = s - 1.0, cd = (double)c;
        unsigned int nm = 0u;
        double s2 = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool in = w[j] * cd > sm1;
            nm |= in ? (1u << j) : 0u;
            s2 += in ? w[j] : 0.0;
        }
        const unsigned int cf = qq_sum_u((unsigned int)__popc(nm) | (nm != m ? 0x10000u : 0u));
        s2 = qq_sum(s2);
        const unsigned int c2 = cf & 0xffffu;
        if (!done) {
            if ((cf >> 16) == 0u || (pass >= 2 && c2 >= c)) {
                done = true;
            } else {
                m = nm;
                s = s2;
                c = c2;
            }
        }
        if (!__any(!done)) break;
    }
    if (live) mask = m;
    return (s - 1.0) / (double)c;
}

template <int MT, bool MEM1, bool LAZYQ = false>   // M tiles of 16 components: KQ = 16 MT, J = 4 MT components per lane;
                               // MEM1: memory == 1 (the default), no history of f in registers
                               // LAZYQ: the stopping test of a pass is decided in the next trip, and skipped for
                               // the whole wave when <d, d> proves it negative for every sample (see qp_wave_body)
__device__ __forceinline__ void qp_quad_body(const double *__restrict__ A /*[lda][lda], zero padded*/,
                                             int lda, const double *__restrict__ B, long stride_j,
                                             long stride_t, const double *__restrict__ bscale,
                                             double *__restrict__ Z, int ldz, long n, int k,
                                             const aa_qp_params &p, int pass_cap, int *__restrict__ iters,
                                             QpHeader *__restrict__ hdr, int *__restrict__ ovf_rows,
                                             QpCarry *__restrict__ ovf, int refill_min,
                                             const int *__restrict__ perm, long max_trips,
                                             int live_epoch, int *__restrict__ ovf_ready, int rst_b, long rst_n)
{
    // restarts side by side (launch_qp_slots_aa): blockIdx.y = slot -- its Hessian (lda x lda), scale
    // vector, columns of B and Z, header and overflow lists
    if (gridDim.y > 1) {
        const int rst = blockIdx.y;
        A += (long)rst * lda * lda;
        if (bscale) bscale += rst * 64;
        B += (long)rst * rst_b;
        Z += (long)rst * rst_b;
        hdr += rst;
        ovf_rows += rst * rst_n;
        ovf += rst * rst_n;
        if (iters) iters += rst * rst_n;
    }
    if (perm && hdr->pad1 == 0u) perm = nullptr;    // the order of k_qp_wave_ord did not complete (k_qp_setup)
    constexpr int J = 4 * MT;
    const int lane = threadIdx.x, sl = lane & 15, q = lane >> 4;
    bool sq_ok;
    const double sq_lim = qp_sq_limit(p.epsilon_two, &sq_ok);
    // LAZYQ: ||P(x - g) - x||_2 >= ||P(x - alpha g) - x||_2 / max(1, alpha); a residual whose square exceeds
    // 4 max(eps2^2, k eps1^2) passes neither stopping test
    const double cert = 4.0 * fmax(p.epsilon_two * p.epsilon_two, (double)k * p.epsilon_one * p.epsilon_one);
    bool pending = false;                           // the test behind this sample's latest pass is still owed
    unsigned int lz_trips = 0u, lz_skipped = 0u;    // LAZYQ: trips with an owed test, and those that skipped it (qp_profile)
    double H[MT][J];                                // A's operand tiles (constant)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < J; ++t) H[mt][t] = A[(long)(16 * mt + sl) * lda + 4 * t + q];
    double x[J], g[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        x[j] = 0.0;
        g[j] = 0.0;
    }
    double f = 0.0, alpha = 1.0, fmem[MEM1 ? 1 : QP_MAXMEM];
#pragma unroll
    for (int i = 0; i < (MEM1 ? 1 : QP_MAXMEM); ++i) fmem[i] = NAN;
    int n_iter = 0, n_feval = 0;
    long row = 0;
    bool active = false, exhausted = false;
    unsigned int support = 0u, support_r = 0u;
    const int mem = p.memory < 1 ? 1 : (p.memory > QP_MAXMEM ? QP_MAXMEM : p.memory);
    const bool alpha0_ok = p.alpha_min <= p.alpha0 && p.alpha0 <= p.alpha_max;
    // static interleaved slices of the (longest-first) sample list: no queue, no atomics
    long next_idx = (long)blockIdx.x * 16 + sl;
    const long idx_stride = (long)gridDim.x * 16;
    unsigned long long st_total = 0ull;
    int st_max = 0;

    for (long trip = 0; trip < max_trips; ++trip) {
        const bool idle = !active && !exhausted;
        const int n_idle = (int)__popcll(__ballot(idle));
        const bool refill = n_idle >= 4 * refill_min || !__any(active);
        bool starting = false;
        if (refill && idle) {
            if (next_idx < n) {
                row = perm ? (long)perm[next_idx] : next_idx;
                next_idx += idx_stride;
                starting = true;
            } else {
                exhausted = true;
            }
        }
        if (!__any(active || starting)) break;
        if (__any(starting)) {
            // ---- start-up: x = P(z0)                                          (spg.py:298-306)
            double xs[J];
#pragma unroll
            for (int j = 0; j < J; ++j) xs[j] = (starting && 4 * j + q < k) ? Z[row * ldz + 4 * j + q] : -QP_PAD;
            unsigned int m0 = 0u;
            const double t0 = qq_threshold<J>(xs, m0, starting);
            if (starting) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    x[j] = (4 * j + q < k) ? fmax(xs[j] - t0, 0.0) : 0.0;
                    g[j] = 0.0;
                }
                support = m0;
                support_r = m0;
            }
        }
        const bool was_active = active;
        if (__any(was_active && n_iter == 0)) {
            if (alpha0_ok) {
                if (was_active && n_iter == 0) alpha = p.alpha0;
            } else {
                double w1[J];
#pragma unroll
                for (int j = 0; j < J; ++j) w1[j] = x[j] - g[j];
                const bool first = was_active && n_iter == 0;
                const double t1 = qq_threshold<J>(w1, support_r, first);
                double ainv = 0.0;
#pragma unroll
                for (int j = 0; j < J; ++j) ainv = fmax(ainv, fabs(fmax(w1[j] - t1, 0.0) - x[j]));
                ainv = qq_max(ainv);
                if (fabs(ainv) < 1e-12) ainv = 1.0;
                if (first) alpha = fmin(fmax(p.alpha_min, 1.0 / ainv), p.alpha_max);
            }
        }
        // ---- search direction d = P(x - alpha g) - x                          (spg.py:318-338)
        double d[J];
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
        {
            double w[J];
#pragma unroll
            for (int j = 0; j < J; ++j) w[j] = x[j] - alpha * g[j];
            const double td = qq_threshold<J>(w, support, was_active);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                d[j] = fmax(w[j] - td, 0.0) - x[j];
                r0 = fma(d[j], g[j], r0);
                r1 = fma(d[j], d[j], r1);
            }
        }
        // ---- A v for the whole wave: v = x for a starting sample, d for an active one
        double Ad[J];
        {
            f64x4 acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][0] = (f64x4){0.0, 0.0, 0.0, 0.0};
                acc[mt][1] = (f64x4){0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
            for (int t = 0; t < J; ++t) {
                const double v = starting ? x[t] : (was_active ? d[t] : 0.0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][t & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(H[mt][t], v, acc[mt][t & 1], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ad[4 * mt + r] = acc[mt][0][r] + acc[mt][1][r];
        }
        if (starting) {
            // ---- rest of the start-up: g = A x + b; f = x'(g + b)/2           (spg.py:298-315)
            r0 = 0.0;
            r1 = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int comp = 4 * j + q;
                const double bi =
                    comp < k ? -B[comp * stride_j + row * stride_t] * (bscale ? bscale[comp] : 1.0) : QP_PAD;
                g[j] = Ad[j] + bi;
                r0 = fma(x[j], g[j], r0);
                r1 = fma(x[j], bi, r1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < J; ++j) r2 = fma(d[j], Ad[j], r2);
        }
        r0 = qq_sum(r0);
        r1 = qq_sum(r1);
        r2 = qq_sum(r2);
        if (starting) {
            f = 0.5 * (r0 + r1);
            n_feval = 1;
            n_iter = 0;
#pragma unroll
            for (int i = 0; i < (MEM1 ? 1 : QP_MAXMEM); ++i) fmem[i] = NAN;
            active = true;                             // its first pass runs in the next trip
            pending = false;
        }
        if (!__any(was_active)) continue;
        // ---- residual of the projected gradient at (x, g), for the samples in `who`   (spg.py:378-396)
        double q2 = 0.0, qinf = 0.0;
        auto residual = [&](bool who) {
            double w[J];
#pragma unroll
            for (int j = 0; j < J; ++j) w[j] = x[j] - g[j];
            const double tr = qq_threshold<J>(w, support_r, who);
            double a2 = 0.0, ai = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const double r = fmax(w[j] - tr, 0.0) - x[j];
                a2 = fma(r, r, a2);
                ai = fmax(ai, fabs(r));
            }
            q2 = qq_sum(a2);
            qinf = qq_max(ai);
        };
        // the sample leaves this kernel: finished (its pass count is final) or parked for the wave kernel
        auto retire = [&](bool finished) {
            if (live_epoch && !finished) {
                // live hand-over: a consumer wave of k_qp_wave may pick this sample up while this
                // kernel is still running, possibly on another XCD: everything it reads goes out
                // with agent-scope stores, drained before the ready flag
#pragma unroll
                for (int j = 0; j < J; ++j)
                    if (4 * j + q < k) qp_store_agent(&Z[row * ldz + 4 * j + q], x[j]);
                unsigned int slot = 0u;
                if (q == 0) {
                    slot = atomicAdd(&hdr->n_overflow, 1u);
                    qp_store_agent(&ovf_rows[slot], (int)row);
                    double *cd = reinterpret_cast<double *>(&ovf[slot]);
                    qp_store_agent(cd, alpha);
                    qp_store_agent(cd + 1, f);
                    qp_store_agent(cd + 2, __longlong_as_double(((long long)n_feval << 32) | (unsigned int)n_iter));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (q == 0) qp_store_agent(&ovf_ready[slot], live_epoch);
            } else {
#pragma unroll
                for (int j = 0; j < J; ++j)
                    if (4 * j + q < k) Z[row * ldz + 4 * j + q] = x[j];
                if (q == 0) {
                    if (finished) {
                        if (iters) iters[row] = n_iter;
                        st_total += (unsigned long long)n_iter;
                        st_max = n_iter > st_max ? n_iter : st_max;
                    } else {
                        const unsigned int slot = atomicAdd(&hdr->n_overflow, 1u);
                        ovf_rows[slot] = (int)row;
                        QpCarry cr;
                        cr.alpha = alpha;
                        cr.f = f;
                        cr.n_iter = n_iter;
                        cr.n_feval = n_feval;
                        ovf[slot] = cr;
                        // "at least the cap": what the ordering blocks of k_qp_wave_ord may read
                        // before the wave that finishes this sample has written the final count
                        if (iters) iters[row] = n_iter;
                    }
                }
            }
            active = false;
        };
        bool upd = was_active;
        if constexpr (LAZYQ) {
            // the test owed from the previous pass, on the (x, g) that pass left: run for the samples whose
            // direction does not rule it out -- for none of them in most trips of a batch of similar samples
            const double am = fmax(1.0, alpha);
            const bool need = was_active && pending && !(r1 > cert * am * am);
            if (__any(was_active && pending)) {
                lz_trips += 1u;
                if (!__any(need)) lz_skipped += 1u;
            }
            if (__any(need)) {
                residual(need);
                if (need && ((sq_ok ? q2 < sq_lim : sqrt(q2) < p.epsilon_two) || (qinf < p.epsilon_one))) {
                    retire(true);
                    upd = false;
                }
            }
            pending = false;
        }
        if (upd) {
            const double delta = r0, dd = r1, dAd = r2;
            // non-monotone reference value (spg.py:341-344): roll, store, nanmax
            double f_max = f;
            if constexpr (!MEM1) {
#pragma unroll
                for (int i = QP_MAXMEM - 1; i > 0; --i)
                    if (i < mem) fmem[i] = fmem[i - 1];
                fmem[0] = f;
#pragma unroll
                for (int i = 1; i < QP_MAXMEM; ++i)
                    if (i < mem && fmem[i] > f_max) f_max = fmem[i];
            }
            double lam = 1.0;
            double f_new = f + lam * delta + 0.5 * lam * lam * dAd;
            n_feval += 1;
            int guard = 0;
            while (f_new > f_max + p.gamma * lam * delta && guard < 200) {
                const double tmp = -0.5 * lam * lam * delta / (f_new - f - lam * delta);
                lam = (p.sigma_one <= tmp && tmp <= p.sigma_two * lam) ? tmp : 0.5 * lam;
                f_new = f + lam * delta + 0.5 * lam * lam * dAd;
                n_feval += 1;
                ++guard;
                if (fabs(lam) < p.lambda_min) break;
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                x[j] = fma(lam, d[j], x[j]);
                g[j] = fma(lam, Ad[j], g[j]);
            }
            const double sksk = lam * lam * dd;
            const double beta = lam * (lam * dAd);
            alpha = (beta <= 0.0) ? p.alpha_max : fmin(p.alpha_max, fmax(p.alpha_min, sksk / beta));
            f = f_new;
            n_feval += 1;
        }
        bool test_now = false;
        if (upd) {
            n_iter += 1;
            if constexpr (LAZYQ) {
                if (n_feval > p.max_feval || n_iter >= p.max_iterations) retire(true);
                else if (n_iter >= pass_cap) test_now = true;       // finished or parked: decided here
                else pending = true;
            } else {
                test_now = true;
            }
        }
        if (__any(test_now)) {
            residual(test_now);
            if (test_now) {
                const bool conv = (sq_ok ? q2 < sq_lim : sqrt(q2) < p.epsilon_two) || (qinf < p.epsilon_one);
                const bool finished = conv || n_feval > p.max_feval || n_iter >= p.max_iterations;
                if (finished || n_iter >= pass_cap) retire(finished);
            }
        }
    }
    if constexpr (LAZYQ) {
        if (threadIdx.x == 0 && lz_trips) {
            atomicAdd(&hdr->dbg_trips, lz_trips);
            atomicAdd(&hdr->dbg_rounds, lz_skipped);
        }
    }
    {   // wave totals (fixed-order butterfly), one atomic pair per wave
        unsigned long long tot = st_total;
        int mx = st_max;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            tot += __shfl_xor(tot, o, 64);
            const int om = __shfl_xor(mx, o, 64);
            mx = om > mx ? om : mx;
        }
        if (threadIdx.x == 0 && tot) {
            atomicAdd(&hdr->total_passes, tot);
            atomicMax(&hdr->max_passes, (unsigned long long)mx);
        }
    }
    if (live_epoch) {                                  // every flag of this wave is out before it counts as done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) atomicAdd(&hdr->waves_done, 1u);
    }
}

// The same body at three register budgets: OCC waves per SIMD (2: whatever the compiler likes,
// 3: <= 168 registers, 4: <= 128 with a little scratch).
#define QQ_ARGS const double *__restrict__ A, int lda, const double *__restrict__ B, long stride_j, long stride_t,   \
                const double *__restrict__ bscale, double *__restrict__ Z, int ldz, long n, int k, aa_qp_params p, \
                int pass_cap, int *__restrict__ iters, QpHeader *__restrict__ hdr, int *__restrict__ ovf_rows,      \
                QpCarry *__restrict__ ovf, int refill_min, const int *__restrict__ perm, long max_trips,           \
                int live_epoch, int *__restrict__ ovf_ready, int rst_b, long rst_n
#define QQ_PASS A, lda, B, stride_j, stride_t, bscale, Z, ldz, n, k, p, pass_cap, iters, hdr, ovf_rows, ovf,        \
                refill_min, perm, max_trips, live_epoch, ovf_ready, rst_b, rst_n
template <int MT, bool MEM1, bool LAZYQ = false>
__global__ __launch_bounds__(64) void k_qp_quad(QQ_ARGS) { qp_quad_body<MT, MEM1, LAZYQ>(QQ_PASS); }
template <int MT, bool MEM1, bool LAZYQ = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_qp_quad_w3(QQ_ARGS)
{
    qp_quad_body<MT, MEM1, LAZYQ>(QQ_PASS);
}
template <int MT, bool MEM1, bool LAZYQ = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_qp_quad_w4(QQ_ARGS)
{
    qp_quad_body<MT, MEM1, LAZYQ>(QQ_PASS);
}
#undef QQ_ARGS
#undef QQ_PASS

// Passes a sample may spend in the lane-per-sample kernel before it is handed to the
// wave-per-sample kernel.
int g_qp_pass_cap = 24;        // settable with aa_set_option("qp_pass_cap", v)
int g_qp_matvec = 0;           // lane kernel mat-vec: 0 f64 MFMA + LDS (default), 1 scalar-unit broadcast (SGPR
                               // operands; 7 100 against 17 800 cycles in isolation, but 2 % slower inside the
                               // kernel: 256 VGPRs + 97 AGPRs + spilled SGPRs)
int g_qp_row_waves = 2048;     // most waves of the row kernel (k_qp_row): 2 per SIMD, all resident
int g_qp_row_hot = 24;         // passes after which a sample's wave takes issue priority
int g_qp_row_chunk = 0;        // queue tickets a wave takes per atomic; 0: static strided assignment, no queue
int g_qp_row_long = 0;         // hybrid: samples with >= this many passes in the previous update go to the
                               // wave-per-sample kernel on the side stream (0: no side stream)
int g_qp_row_cap = 1 << 30;        // passes after which the row kernel hands a sample to the wave-per-sample kernel
int g_qp_refill_min = 64;      // idle lanes of a wave that trigger a refill (1..64); 64 = only
                               // when the whole wave is idle: a sample's start-up (strided row
                               // loads, a cold projection) is executed by the whole wave, and
                               // mid-flight refills cost more than the idle lanes they fill
                               // (2.64 ms per outer iteration against 2.83 at 24)
int g_qp_mode = 0;             // 0: by size (see launch_qp), 1: wave-per-sample only,
                               // 2: lane-per-sample then wave-per-sample, 3: row kernel,
                               // 4: four lanes per sample (matrix-core layout) then wave-per-sample
int g_qp_quad_waves = 8192;    // most waves of k_qp_quad: up to 131 072 samples every wave takes ONE batch of 16
                               // and the hardware hands the batches (longest first) to SIMDs as they free up
int g_qp_quad_refill = 16;     // idle sample slots (of 16) of a wave that trigger a refill
int g_qp_quad_cap = 0;         // passes after which k_qp_quad parks a sample for the wave kernel; 0: by size --
                               // 32 from 65 536 samples per GPU (100 000: 1.985 against 2.010 ms per outer
                               // iteration), 24 below (12 500: 0.572 against 0.582)
int g_qp_wave_queue = 0;        // continuation launch, experiment: waves take the parked samples by ticket instead of fixed strides (slower: 492-494 against 496-499 it/s; the chains are latency-bound and four waves share a SIMD at almost no cost)
int g_qp_wave_blocks = 1024;   // blocks (4 waves each) of the wave-per-sample kernel when it finishes parked samples
int g_qp_live = 0;             // k_qp_quad hands parked samples to a concurrent k_qp_wave launch (QpLive)
int g_qp_live_occ = 3;         // register budget of k_qp_quad beside the consumers (waves per SIMD)
int g_qp_live_blocks = 48;     // CUs given to the consumers (one block of 16 waves each)
#define QP_LIVE_LDS 163840     // a CU's LDS
int g_qp_quad_occ = 3;         // register budget of k_qp_quad: waves per SIMD (2, 3 or 4)
int g_qp_quad_lazy = 0;        // four-lane QP, opt-in: the stopping test of a pass decided in the next trip (LAZYQ) -- same results; 497/497/494 it/s with, 503/498 without at the driver's flags, 524/523 against 522 at the default flags: a third of the trips skip the test for the whole wave (qp_profile: 32 % in iterations 8-14, 31 % around 30), which saves less than the extra partial trip per batch and the 22 VGPR spills (7 without) cost
int g_qp_fused_order = 1;   // four-lane QP: the sample order of the NEXT update is formed by extra blocks of this update's continuation launch (k_qp_wave_ord)
int g_qp_wave_lazy = 1;        // wave-per-sample kernel: stopping test of a pass decided at the top of the next one, skipped when <d, d> proves it negative (0: after every pass)
int g_qp_wave_mem1 = 1;        // continuation launches of the wave-per-sample kernel: 1 = the memory-1 instantiation (no f_mem array: 311 fewer SGPR spills), 0 = the generic one (A/B)
int g_qp_overlap_tail = 0;     // 1: stragglers on a side stream, overlapped with the Z'X pass
int g_qp_tail_cap = 96;        // with qp_overlap_tail: only samples beyond this many passes go to the side stream (0: all parked ones)
int g_qp_profile = 0;          // cycle accounting of k_qp (printed when stats are requested)
int g_qp_sort = 1;             // order the samples by the previous update's pass counts
int g_qp_waves = 1024;         // most waves the lane-per-sample kernel is launched with
static int qp_pass_cap() { return g_qp_pass_cap < 1 ? 1 : g_qp_pass_cap; }

// Order of the samples for the lane kernel: by the pass count of the PREVIOUS weights update,
// longest first (counting sort: block histograms, then block-wise scatter into per-bucket
// ranges; the order inside a bucket depends on block scheduling, which
// cannot change any result: samples are independent).  A wave then works on 64 samples of
// similar length -- a batch ends when its slowest sample does (mean 18.6 instead of 23.8
// trips at a 0.83 correlation between consecutive updates) -- and the long batches start
// first, the short ones fill the second round.
#define QP_SORT_BUCKETS 64
#define QP_SORT_ROWS_PER_BLOCK 1024

__device__ __forceinline__ int qp_sort_bucket(int it)
{
    return it < 0 ? 0 : (it >= QP_SORT_BUCKETS ? QP_SORT_BUCKETS - 1 : it);
}

__global__ __launch_bounds__(256) void k_qp_order_hist(const int *__restrict__ prev_iters, long n,
                                                       int *__restrict__ ghist)
{
    __shared__ int hist[QP_SORT_BUCKETS];
    if (threadIdx.x < QP_SORT_BUCKETS) hist[threadIdx.x] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * QP_SORT_ROWS_PER_BLOCK;
#pragma unroll
    for (int j = 0; j < QP_SORT_ROWS_PER_BLOCK / 256; ++j) {
        const long r = r0 + threadIdx.x + 256 * j;
        if (r < n) atomicAdd(&hist[qp_sort_bucket(prev_iters[r])], 1);
    }
    __syncthreads();
    if (threadIdx.x < QP_SORT_BUCKETS && hist[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], hist[threadIdx.x]);
}

// ghist: complete histogram; gcursor: per-bucket fill counters (zeroed).  Buckets are laid
// out in DESCENDING order of the pass count; a block reserves one range per bucket.
__global__ __launch_bounds__(256) void k_qp_order_scatter(const int *__restrict__ prev_iters, long n,
                                                          const int *__restrict__ ghist,
                                                          int *__restrict__ gcursor,
                                                          int *__restrict__ perm,
                                                          QpHeader *__restrict__ hdr, int long_from)
{
    __shared__ int hist[QP_SORT_BUCKETS], cursor[QP_SORT_BUCKETS];
    const int t = threadIdx.x;
    // samples whose previous update needed >= long_from passes: the head of the sorted list
    if (hdr && blockIdx.x == 0 && t == 0) {
        int m = 0;
        if (long_from > 0)
            for (int b = QP_SORT_BUCKETS - 1; b >= long_from && b >= 0; --b) m += ghist[b];
        hdr->n_long = (unsigned int)m;
    }
    if (t < QP_SORT_BUCKETS) hist[t] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * QP_SORT_ROWS_PER_BLOCK;
    int bk[QP_SORT_ROWS_PER_BLOCK / 256];
#pragma unroll
    for (int j = 0; j < QP_SORT_ROWS_PER_BLOCK / 256; ++j) {
        const long r = r0 + t + 256 * j;
        bk[j] = r < n ? qp_sort_bucket(prev_iters[r]) : -1;
        if (bk[j] >= 0) atomicAdd(&hist[bk[j]], 1);
    }
    __syncthreads();
    if (t < QP_SORT_BUCKETS) {
        int base = 0;
        for (int b = QP_SORT_BUCKETS - 1; b > t; --b) base += ghist[b];
        cursor[t] = base + (hist[t] ? atomicAdd(&gcursor[t], hist[t]) : 0);   // this block's range
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < QP_SORT_ROWS_PER_BLOCK / 256; ++j)
        if (bk[j] >= 0) perm[atomicAdd(&cursor[bk[j]], 1)] = (int)(r0 + t + 256 * j);
}

// The same counting sort as extra blocks of the continuation launch (k_qp_wave_ord): the order the
// NEXT update's k_qp_quad takes its samples in is formed while this update's stragglers run, off the
// critical path (the two ordering launches above cost 33 us per update in front of k_qp_quad, on a
// chip that idles).  A sample's bucket is read ONCE, into a register, and serves both the histogram
// and the scatter, so the order is a permutation whatever the concurrent waves of this launch write
// to the pass counts of the parked samples meanwhile (k_qp_quad leaves "the cap" there, the wave that
// finishes the sample the final count: the head of the list either way).  area: [64] histogram,
// [64] cursors, [0] blocks past the histogram, [1] blocks past the scatter -- zeroed by k_qp_setup,
// which also turns [1] == blocks of the PREVIOUS update into QpHeader::pad1 (order complete: use it).
// The blocks are the first of the grid, hence resident together; a wait that does not end (it cannot,
// short of a dead chip) gives up after 2 ms and the next update runs unordered.
struct QpOrder {
    const int *iters;
    int *perm, *area;
    long n;
    int blocks;
};
#define QP_ORDER_AREA (2 * QP_SORT_BUCKETS + 4)
#define QP_ORDER_MAX_BLOCKS 256

__device__ __forceinline__ void qp_order_block(const QpOrder &od)
{
    __shared__ int hist[QP_SORT_BUCKETS], cursor[QP_SORT_BUCKETS], ok;
    int *ghist = od.area, *gcursor = od.area + QP_SORT_BUCKETS, *flags = od.area + 2 * QP_SORT_BUCKETS;
    const int t = threadIdx.x;
    if (t < QP_SORT_BUCKETS) hist[t] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * QP_SORT_ROWS_PER_BLOCK;
    int bk[QP_SORT_ROWS_PER_BLOCK / 256];
#pragma unroll
    for (int j = 0; j < QP_SORT_ROWS_PER_BLOCK / 256; ++j) {
        const long r = r0 + t + 256 * j;
        bk[j] = r < od.n ? qp_sort_bucket(od.iters[r]) : -1;
        if (bk[j] >= 0) atomicAdd(&hist[bk[j]], 1);
    }
    __syncthreads();
    if (t < QP_SORT_BUCKETS && hist[t]) atomicAdd(&ghist[t], hist[t]);
    __threadfence();
    __syncthreads();
    if (t == 0) {
        __hip_atomic_fetch_add(&flags[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = (long long)wall_clock64();         // 100 MHz
        int good = 1;
        while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < od.blocks) {
            if ((long long)wall_clock64() - t0 > 200000ll) { good = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        ok = good;
    }
    __syncthreads();
    if (!ok) return;
    if (t < QP_SORT_BUCKETS) {
        int base = 0;
        for (int b = QP_SORT_BUCKETS - 1; b > t; --b)
            base += __hip_atomic_load(&ghist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cursor[t] = base + (hist[t] ? atomicAdd(&gcursor[t], hist[t]) : 0);   // this block's range
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < QP_SORT_ROWS_PER_BLOCK / 256; ++j)
        if (bk[j] >= 0) od.perm[atomicAdd(&cursor[bk[j]], 1)] = (int)(r0 + t + 256 * j);
    __threadfence();
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(&flags[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// continuation launch of k_qp_wave<32, true, LAZY> (the overflow list of k_qp_quad) with od.blocks
// ordering blocks in front
template <bool LAZY>
__global__ __launch_bounds__(256) void k_qp_wave_ord(QpOrder od, const double *__restrict__ A,
                                                     const double *__restrict__ B, long stride_j, long stride_t,
                                                     const double *__restrict__ bscale, double *__restrict__ Z,
                                                     int ldz, int k, aa_qp_params p, int *__restrict__ iters,
                                                     QpHeader *__restrict__ hdr, const int *__restrict__ ovf_rows,
                                                     const QpCarry *__restrict__ ovf, int park_at = 1 << 30,
                                                     unsigned int *__restrict__ n_parked = nullptr,
                                                     int *__restrict__ park_rows = nullptr,
                                                     QpCarry *__restrict__ park = nullptr, int queue = 0)
{
    if ((int)blockIdx.x < od.blocks) {
        qp_order_block(od);
        return;
    }
    qp_wave_body<32, true, LAZY>(A, B, stride_j, stride_t, bscale, Z, ldz, (long)-1, k, p, iters, hdr, ovf_rows, ovf,
                                 (double *)nullptr, (const int *)nullptr, (const unsigned int *)nullptr, park_at,
                                 n_parked, park_rows, park,
                                 QpLive{queue ? 3 : 0, 0, 0u, 0u, nullptr, nullptr}, 0, 0L, (unsigned int)od.blocks);
}

// device-side set-up of the QP scratch: header zeroed, A = D G D padded to KQ and KW,
// b-scale = D.  One block.
__global__ __launch_bounds__(256) void k_qp_setup(QpHeader *__restrict__ hdr, double *__restrict__ Ad,
                                                  double *__restrict__ A2d, double *__restrict__ bsd,
                                                  const double *__restrict__ gram,
                                                  const double *__restrict__ alpha, int k, int KQ,
                                                  int KW, int KP, int *__restrict__ sort_hist,
                                                  int order_blocks = 0)
{
    const int t = threadIdx.x;
    // order_blocks: the ordering blocks the previous update's k_qp_wave_ord ran with -- all of them through?
    __shared__ int order_ok;
    if (t == 0)
        order_ok = order_blocks < 0 ? 1     // (an order, if any, comes from the ordering launches in stream order)
                                    : ((sort_hist && order_blocks > 0 && sort_hist[2 * QP_SORT_BUCKETS + 1] == order_blocks) ? 1 : 0);
    __syncthreads();
    if (sort_hist && t < QP_ORDER_AREA) sort_hist[t] = 0;   // histogram + cursors (+ flags) of the ordering
    if (t == 0) {
        hdr->pad1 = (unsigned int)order_ok;
        hdr->total_passes = 0ull;
        hdr->max_passes = 0ull;
        hdr->next_row = 0u;
        hdr->n_overflow = 0u;
        hdr->next_overflow = 0u;
        hdr->n_long = 0u;
        hdr->dbg_rounds = hdr->dbg_trips = hdr->dbg_waves = 0u;
        hdr->pad = 0u;
        hdr->waves_done = 0u;
    }
    for (int e = t; e < KQ * KQ; e += 256) {
        const int i = e / KQ, j = e % KQ;
        Ad[e] = (i < k && j < k) ? alpha[i] * gram[i * KP + j] * alpha[j] : 0.0;
    }
    for (int e = t; e < KW * KW; e += 256) {
        const int i = e / KW, j = e % KW;
        A2d[e] = (i < k && j < k) ? alpha[i] * gram[i * KP + j] * alpha[j] : 0.0;
    }
    if (t < 64) bsd[t] = t < k ? alpha[t] : 1.0;
}

// counting sort of the samples by their previous pass counts, longest first -> *perm_out
static int qp_order_rows(Ctx *c, const int *iters_dev, long n, const int **perm_out,
                         QpHeader *hdr = nullptr, int long_from = 0, bool zeroed = false)
{
    AA_CHECK(c->qpPerm.alloc((size_t)n * sizeof(int) + QP_ORDER_AREA * sizeof(int)));
    int *pm = c->qpPerm.as<int>();
    int *ghist = pm + n, *gcursor = ghist + QP_SORT_BUCKETS;
    if (!zeroed) AA_CHECK_HIP(hipMemsetAsync(ghist, 0, 2 * QP_SORT_BUCKETS * sizeof(int), c->stream));
    const unsigned nblk = (unsigned)((n + QP_SORT_ROWS_PER_BLOCK - 1) / QP_SORT_ROWS_PER_BLOCK);
    hipLaunchKernelGGL(k_qp_order_hist, dim3(nblk), dim3(256), 0, c->stream, iters_dev, n, ghist);
    hipLaunchKernelGGL(k_qp_order_scatter, dim3(nblk), dim3(256), 0, c->stream, iters_dev, n,
                       (const int *)ghist, gcursor, pm, hdr, long_from);
    *perm_out = pm;
    return AA_OK;
}

// R QPs per sample, one per restart slot (GPNH restarts side by side, kernels_tall.hip: GpnhSlots):
// Hessian of slot r = diagonal block r of the KP x KP Gram, b and z = its k columns of the tall
// arrays.  One set-up launch and ONE launch of the lane-per-sample kernel for all slots
// (grid.y = R), every sample run to completion there (no pass cap: the drivers' setting is a single
// SPG iteration per update, where this kernel is what the single fit uses too).
__global__ __launch_bounds__(256) void k_qp_setup_slots(QpHeader *__restrict__ hdr, double *__restrict__ Ad,
                                                        const double *__restrict__ gram, int k, int KQ, int KP)
{
    const int t = threadIdx.x, r = blockIdx.x, o = r * k;
    if (t == 0) {
        QpHeader *h = hdr + r;
        h->total_passes = 0ull;
        h->max_passes = 0ull;
        h->next_row = 0u;
        h->n_overflow = 0u;
        h->next_overflow = 0u;
        h->n_long = 0u;
        h->dbg_rounds = h->dbg_trips = h->dbg_waves = 0u;
        h->pad = 0u;
        h->waves_done = 0u;
    }
    double *A = Ad + (size_t)r * KQ * KQ;
    for (int e = t; e < KQ * KQ; e += 256) {
        const int i = e / KQ, j = e % KQ;
        A[e] = (i < k && j < k) ? 1.0 * gram[(o + i) * KP + o + j] * 1.0 : 0.0;
    }
}

int launch_qp_slots(Ctx *c, int R, int k, const double *gram_dev, const aa_qp_params *p)
{
    int KQ = 4;
    while (KQ < k) KQ *= 2;
    AA_REQUIRE(KQ <= 32 && R >= 1 && R * k <= c->KP, AA_ERR_ARG, "QP slots: k = %d, R = %d unsupported", k, R);
    AA_REQUIRE(p->memory <= QP_MAXMEM, AA_ERR_ARG, "QP slots: memory = %d > %d", p->memory, QP_MAXMEM);
    const long n = c->n;
    const size_t off_A = 64 * (size_t)R;                       // R headers of 64 bytes
    const size_t bytes = off_A + (size_t)R * KQ * KQ * sizeof(double);
    AA_CHECK(c->qpStats.alloc(bytes < 4096 ? 4096 : bytes));
    unsigned char *base = reinterpret_cast<unsigned char *>(c->qpStats.p);
    QpHeader *hdr = reinterpret_cast<QpHeader *>(base);
    double *Ad = reinterpret_cast<double *>(base + off_A);
    static_assert(sizeof(QpHeader) == 64, "one header per 64 bytes");
    hipLaunchKernelGGL(k_qp_setup_slots, dim3((unsigned)R), dim3(256), 0, c->stream, hdr, Ad, gram_dev, k, KQ, c->KP);
    long waves = (n + 63) / 64;
    if (waves > g_qp_waves) waves = g_qp_waves;
    const dim3 grid((unsigned)waves, (unsigned)R);
    const int cap = p->max_iterations;
    double *Zt = c->Zt.as<double>();
    const double *Bt = c->Gr.as<double>();
#define QPS(KQV, FULLV)                                                                                  \
    hipLaunchKernelGGL((k_qp<KQV, FULLV, false, false>), grid, dim3(64), 0, c->stream, (const double *)Ad, Bt, \
                       (long)1, (long)c->KP, (const double *)nullptr, Zt, c->KP, n, k, *p, cap, (int *)nullptr,  \
                       hdr, (int *)nullptr, (QpCarry *)nullptr, g_qp_refill_min, (QpDebug *)nullptr,              \
                       (const int *)nullptr, KQV * KQV, k, k)
#define QPSK(KQV) do { if (k == KQV) QPS(KQV, true); else QPS(KQV, false); } while (0)
    switch (KQ) { case 4: QPSK(4); break; case 8: QPSK(8); break; case 16: QPSK(16); break; default: QPSK(32); break; }
#undef QPSK
#undef QPS
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// AA restarts side by side: the weights QPs of R slots in ONE launch of the four-lanes-per-sample kernel
// and ONE of the wave-per-sample kernel (grid.y = slot), each slot with its own Hessian
// D C K C' D (diagonal block of the Gram state), scale vector, header and overflow list -- per
// sample the arithmetic of the single fit (launch_qp, quad mode, no sample ordering).
__global__ __launch_bounds__(256) void k_qp_setup_slots_aa(QpHeader *__restrict__ hdr, double *__restrict__ A2d,
                                                           double *__restrict__ bsd, const double *__restrict__ gram,
                                                           const double *__restrict__ alpha, int k, int KP)
{
    const int t = threadIdx.x, r = blockIdx.x, o = r * k;
    if (t == 0) {
        QpHeader *h = hdr + r;
        h->total_passes = 0ull;
        h->max_passes = 0ull;
        h->next_row = 0u;
        h->n_overflow = 0u;
        h->next_overflow = 0u;
        h->n_long = 0u;
        h->dbg_rounds = h->dbg_trips = h->dbg_waves = 0u;
        h->pad = 0u;
        h->waves_done = 0u;
    }
    double *A2 = A2d + (size_t)r * 32 * 32;
    for (int e = t; e < 32 * 32; e += 256) {
        const int i = e / 32, j = e % 32;
        A2[e] = (i < k && j < k) ? alpha[o + i] * gram[(o + i) * KP + o + j] * alpha[o + j] : 0.0;
    }
    if (t < 64) bsd[r * 64 + t] = t < k ? alpha[o + t] : 1.0;
}

int launch_qp_slots_aa(Ctx *c, const aa_qp_params *p)
{
    const int R = c->slots_R, k = c->slots_k;
    AA_REQUIRE(k <= 16 && p->memory <= 1 && p->max_iterations >= 1, AA_ERR_ARG,
               "AA slots: k <= 16, QP memory 1");
    const long n = c->n;
    AA_REQUIRE(n < 65536, AA_ERR_ARG, "AA slots: fewer than 65 536 samples");
    const long n_al = round_up(n, 16);
    const size_t off_A = 64 * (size_t)R;
    const size_t off_bs = off_A + (size_t)R * 32 * 32 * sizeof(double);
    const size_t off_rows = off_bs + (size_t)R * 64 * sizeof(double);
    const size_t off_ovf = off_rows + (size_t)R * n_al * sizeof(int);
    const size_t bytes = off_ovf + (size_t)R * n_al * sizeof(QpCarry);
    AA_CHECK(c->qpStats.alloc(bytes));
    unsigned char *base = reinterpret_cast<unsigned char *>(c->qpStats.p);
    QpHeader *hdr = reinterpret_cast<QpHeader *>(base);
    double *A2d = reinterpret_cast<double *>(base + off_A);
    double *bsd = reinterpret_cast<double *>(base + off_bs);
    int *ovf_rows = reinterpret_cast<int *>(base + off_rows);
    QpCarry *ovf = reinterpret_cast<QpCarry *>(base + off_ovf);
    const double *gram = c->gramState.as<double>() + (size_t)c->KP * c->KP;          // C K C'
    hipLaunchKernelGGL(k_qp_setup_slots_aa, dim3((unsigned)R), dim3(256), 0, c->stream, hdr, A2d, bsd, gram,
                       (const double *)c->alphaDev.as<double>(), k, c->KP);
    int cap = g_qp_quad_cap > 0 ? g_qp_quad_cap : 24;
    if (p->max_iterations <= cap) cap = p->max_iterations;
    // (a single fit orders its samples by their previous pass counts from 4097 samples on: which wave
    // takes a sample does not enter its arithmetic, so the slots go without)
    long waves = (n + 15) / 16;
    if (waves > g_qp_quad_waves) waves = g_qp_quad_waves;
    const long rounds = (n + 16 * waves - 1) / (16 * waves);
    const long max_trips = 16 * rounds * ((long)cap + 2) + 16;
    const int refill = g_qp_quad_refill < 1 ? 1 : (g_qp_quad_refill > 16 ? 16 : g_qp_quad_refill);
    const double *Bt = c->Gr.as<double>();
    double *Zt = c->Zt.as<double>();
    const dim3 grid((unsigned)waves, (unsigned)R);
#define QQS(KERN)                                                                                         \
    hipLaunchKernelGGL((KERN<1, true>), grid, dim3(64), 0, c->stream, (const double *)A2d, 32, Bt, (long)1,      \
                       (long)c->KP, (const double *)bsd, Zt, c->KP, n, k, *p, cap, (int *)nullptr, hdr, ovf_rows, \
                       ovf, refill, (const int *)nullptr, max_trips, 0, (int *)nullptr, k, n_al)
    if (g_qp_quad_occ >= 4) QQS(k_qp_quad_w4);
    else if (g_qp_quad_occ == 3) QQS(k_qp_quad_w3);
    else QQS(k_qp_quad);
#undef QQS
    // (a wave per parked sample: no more blocks per slot than a quarter of its samples -- which block
    // continues a sample does not enter its arithmetic, and R x 1024 mostly idle blocks cost 49 us)
    long wblocks = (n + 3) / 4;
    if (wblocks < 64) wblocks = 64;
    if (wblocks > g_qp_wave_blocks) wblocks = g_qp_wave_blocks;
    if (cap < p->max_iterations)
        QW32_LAUNCH(dim3((unsigned)wblocks, (unsigned)R), dim3(256), 0, c->stream,
                           (const double *)A2d, Bt, (long)1, (long)c->KP, (const double *)bsd, Zt, c->KP, (long)-1, k,
                           *p, (int *)nullptr, hdr, (const int *)ovf_rows, (const QpCarry *)ovf, (double *)nullptr,
                           (const int *)nullptr, (const unsigned int *)nullptr, 1 << 30, (unsigned int *)nullptr,
                           (int *)nullptr, (QpCarry *)nullptr, QpLive{0, 0, 0u, 0u, nullptr, nullptr}, k, n_al);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_qp(Ctx *c, const double *A_host, const double *Btall, long stride_j, long stride_t,
              const double *bscale_host, double *Ztall, int ldz, long n, int k,
              const aa_qp_params *p, int *iters_dev, aa_qp_stats *stats, const double *gram_dev,
              bool defer_tail)
{
    int KQ = 4;
    while (KQ < k) KQ *= 2;
    AA_REQUIRE(KQ <= 64, AA_ERR_ARG, "QP: k = %d > 64 unsupported", k);
    AA_REQUIRE(n < (1L << 31), AA_ERR_ARG, "QP: too many samples");
    // default (qp_mode 0), k <= 32: four lanes per sample in the matrix-core operand layout
    // (k_qp_quad) followed by the wave-per-sample kernel for the samples that reach its pass cap --
    // ahead of the other mappings at every size measured (1 600 .. 100 000 samples per GPU:
    // 0.63 ms per outer iteration at 12 500 rows against 0.69 for the row kernel and 0.81 for
    // lane + wave; 1.99 against 2.21 ms at 100 000) -- except when every sample gets the same
    // one or two passes (GPNH's weights QP with max_iterations = 1): nothing diverges then and
    // the lane-per-sample kernel, 64 samples per wave, is cheaper (0.175 against 0.188 ms on the
    // C3 stand-in).  k > 32: one wave per sample.
    // spg.py:310 allocates f_mem of any length; the kernels keep it in registers: 8 entries in the
    // throughput kernels, 16 in the row kernel, 32 in the wave-per-sample kernel, which therefore
    // takes the whole update when memory > 8 (a setting the reference's callers never use)
    const bool big_mem = p->memory > QP_MAXMEM;
    const bool row_mode = KQ <= 32 && g_qp_mode == 3 && p->memory <= QR_MAXMEM;
    const bool quad_mode = !big_mem && KQ <= 32 && (g_qp_mode == 4 || (g_qp_mode == 0 && p->max_iterations > 4));
    const bool wave_only = !row_mode && !quad_mode && (big_mem || KQ > 32 || g_qp_mode == 1 || g_qp_mode == 3);   // else: lane + wave
    const int KW = KQ > 32 ? 64 : 32;              // A padding of the wave and row kernels
    AA_REQUIRE(p->memory <= QW_MAXMEM, AA_ERR_ARG,
               "QP: memory = %d exceeds the HIP backend limit of %d", p->memory, QW_MAXMEM);
    // scratch layout: QpHeader | A[KQ*KQ] | A2[KW*KW] | bscale[64] | ovf_rows[n] | ovf[n]
    const size_t off_A = 192;             // QpHeader at 0, QpDebug at 64 (96 bytes)
    const size_t off_A2 = off_A + (size_t)KQ * KQ * sizeof(double);
    const size_t off_bs = off_A2 + (size_t)KW * KW * sizeof(double);
    const size_t off_rows = off_bs + 64 * sizeof(double);
    const size_t off_ovf = off_rows + round_up((long)n * sizeof(int), 16);
    const size_t off_rows2 = off_ovf + (size_t)n * sizeof(QpCarry);          // second overflow list (qp_tail_cap)
    const size_t off_ovf2 = off_rows2 + round_up((long)n * sizeof(int), 16);
    const size_t bytes = off_ovf2 + (size_t)n * sizeof(QpCarry);
    AA_CHECK(c->qpStats.alloc(bytes));
    // samples in the order of their pass counts in the previous update of this context (iters_dev
    // still holds them; the kernels below overwrite them); pointless when everybody gets the same
    // one or two passes, and for the kernel that takes a wave per sample anyway
    // and for a few thousand samples (measured: 1610 and 3000 rows 2 % faster unordered, 12 500 rows
    // 3 % slower)
    const bool will_sort = g_qp_sort && p->max_iterations > 2 && n > 4096 && !wave_only && iters_dev &&
                           iters_dev == c->qpIters.as<int>() && c->qp_iters_valid;
    int *sort_hist = nullptr;
    // four-lane kernel: the order is formed by extra blocks of the PREVIOUS update's continuation launch
    // (k_qp_wave_ord) and this update's launch forms the next one; an update without a predecessor runs
    // unordered
    int quad_cap = g_qp_quad_cap > 0 ? g_qp_quad_cap : (n >= 65536 ? 32 : 24);
    if (p->memory > 1 || p->max_iterations <= quad_cap) quad_cap = p->max_iterations;
    // (deferred stragglers: only the two-stage form, whose first stage is an ordinary launch on the main stream)
    const bool defer_ok = defer_tail && !stats && c->stream2 && ldz == c->KP && KW == 32 && c->dtype == AA_F32 &&
                          c->tmpTall.bytes >= (size_t)n * 32 * sizeof(double);
    const bool two_stage = defer_ok && g_qp_tail_cap > quad_cap && g_qp_tail_cap < p->max_iterations;
    const bool fused_order = g_qp_fused_order && g_qp_sort && quad_mode && !A_host && KW == 32 && p->memory <= 1 &&
                             g_qp_wave_mem1 && !g_qp_live && (!defer_ok || two_stage) && quad_cap < p->max_iterations &&
                             p->max_iterations > 2 && n > 4096 &&
                             n <= (long)QP_ORDER_MAX_BLOCKS * QP_SORT_ROWS_PER_BLOCK && iters_dev &&
                             iters_dev == c->qpIters.as<int>();
    const int order_blocks = (int)((n + QP_SORT_ROWS_PER_BLOCK - 1) / QP_SORT_ROWS_PER_BLOCK);
    const bool use_prefetched = fused_order && c->qp_perm_ready && c->qp_perm_n == n;
    c->qp_perm_ready = false;
    if (A_host) {
        std::vector<unsigned char> host(off_rows, 0);
        double *Ah = reinterpret_cast<double *>(host.data() + off_A);
        double *A2h = reinterpret_cast<double *>(host.data() + off_A2);
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                Ah[i * KQ + j] = A_host[i * k + j];
                A2h[i * KW + j] = A_host[i * k + j];
            }
        double *bs = reinterpret_cast<double *>(host.data() + off_bs);
        for (int i = 0; i < 64; ++i) bs[i] = (bscale_host && i < k) ? bscale_host[i] : 1.0;
        reinterpret_cast<QpHeader *>(host.data())->pad1 = 1u;      // an order, if any, is formed in stream order
        AA_CHECK_HIP(hipMemcpyAsync(c->qpStats.p, host.data(), off_rows, hipMemcpyHostToDevice, c->stream));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));   // host vector goes out of scope
    } else {
        AA_REQUIRE(gram_dev != nullptr, AA_ERR_ARG, "QP: no Hessian");
        if (will_sort || fused_order) {                // its histograms are zeroed by the set-up kernel
            AA_CHECK(c->qpPerm.alloc((size_t)n * sizeof(int) + QP_ORDER_AREA * sizeof(int)));
            sort_hist = c->qpPerm.as<int>() + n;
        }
        unsigned char *b0 = reinterpret_cast<unsigned char *>(c->qpStats.p);
        hipLaunchKernelGGL(k_qp_setup, dim3(1), dim3(256), 0, c->stream, reinterpret_cast<QpHeader *>(b0),
                           reinterpret_cast<double *>(b0 + off_A), reinterpret_cast<double *>(b0 + off_A2),
                           reinterpret_cast<double *>(b0 + off_bs), gram_dev,
                           (const double *)c->alphaDev.as<double>(), k, KQ, KW, c->KP, sort_hist,
                           fused_order ? (use_prefetched ? order_blocks : 0) : -1);
    }
    unsigned char *base = reinterpret_cast<unsigned char *>(c->qpStats.p);
    QpHeader *hdr = reinterpret_cast<QpHeader *>(base);
    const double *Ad = reinterpret_cast<const double *>(base + off_A);
    const double *A2d = reinterpret_cast<const double *>(base + off_A2);
    const double *bsd = (bscale_host || !A_host) ? reinterpret_cast<const double *>(base + off_bs) : nullptr;
    int *ovf_rows = reinterpret_cast<int *>(base + off_rows);
    QpCarry *ovf = reinterpret_cast<QpCarry *>(base + off_ovf);
    int *ovf2_rows = reinterpret_cast<int *>(base + off_rows2);
    QpCarry *ovf2 = reinterpret_cast<QpCarry *>(base + off_ovf2);

    if (p->max_iterations <= 0) {
        dim3 grid((unsigned)((n + 255) / 256));
#define QPP(KQV) hipLaunchKernelGGL(k_qp_project_only<KQV>, grid, dim3(256), 0, c->stream, Ztall, Ztall, ldz, n, k)
        switch (KQ) { case 4: QPP(4); break; case 8: QPP(8); break; case 16: QPP(16); break;
                      case 32: QPP(32); break; default: QPP(64); break; }
#undef QPP
    } else if (row_mode) {
        // samples ordered by their previous pass count, longest first (iters_dev still holds the
        // counts of the previous update of this context).  Hybrid: the head of the list -- the
        // samples that needed >= g_qp_row_long passes last time, ~2 % of them but the whole
        // critical path -- goes to the wave-per-sample kernel (1 us per pass) on the side stream,
        // started at once; the row kernel (four samples per wave, ~3 us per pass but four times
        // the throughput) works off the rest concurrently and hands samples that unexpectedly
        // reach the pass cap to a second wave-per-sample launch.
        const int *perm = nullptr;
        if (g_qp_profile) AA_CHECK_HIP(hipMemsetAsync(base + 64, 0, 64, c->stream));
        const bool sorted = will_sort;
        const bool hybrid = sorted && g_qp_row_long > 0 && c->stream2 && KW == 32 && p->memory <= 1;
        if (sorted) AA_CHECK(qp_order_rows(c, iters_dev, n, &perm, hdr, hybrid ? g_qp_row_long : 0, sort_hist != nullptr));
        int cap = g_qp_row_cap < 1 ? 1 : g_qp_row_cap;
        if (p->memory > 1 || p->max_iterations <= cap || KW != 32) cap = p->max_iterations;
        if (hybrid) {
            AA_CHECK_HIP(hipEventRecord(c->evFork, c->stream));
            AA_CHECK_HIP(hipStreamWaitEvent(c->stream2, c->evFork, 0));
            QW32_LAUNCH(dim3(512), dim3(256), 0, c->stream2, A2d, Btall, stride_j,
                               stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                               (const int *)ovf_rows, (const QpCarry *)ovf, (double *)nullptr, perm);
            AA_CHECK_HIP(hipEventRecord(c->evJoin, c->stream2));
        }
        long waves = (n + 3) / 4;
        if (waves > g_qp_row_waves) waves = g_qp_row_waves;
        if (k <= 16)
            hipLaunchKernelGGL(k_qp_row<1>, dim3((unsigned)waves), dim3(64), 0, c->stream, A2d, KW, Btall,
                               stride_j, stride_t, bsd, Ztall, ldz, n, k, *p, iters_dev, hdr, perm,
                               g_qp_row_hot, g_qp_profile, g_qp_row_chunk, cap, ovf_rows, ovf);
        else
            hipLaunchKernelGGL(k_qp_row<2>, dim3((unsigned)waves), dim3(64), 0, c->stream, A2d, KW, Btall,
                               stride_j, stride_t, bsd, Ztall, ldz, n, k, *p, iters_dev, hdr, perm,
                               g_qp_row_hot, g_qp_profile, g_qp_row_chunk, cap, ovf_rows, ovf);
        if (hybrid) AA_CHECK_HIP(hipStreamWaitEvent(c->stream, c->evJoin, 0));
        if (cap < p->max_iterations)
            QW32_LAUNCH(dim3((unsigned)g_qp_wave_blocks), dim3(256), 0, c->stream, A2d, Btall, stride_j,
                               stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                               (const int *)ovf_rows, (const QpCarry *)ovf, (double *)nullptr);
    } else if (quad_mode) {
        int cap = g_qp_quad_cap > 0 ? g_qp_quad_cap : (n >= 65536 ? 32 : 24);
        if (p->memory > 1 || p->max_iterations <= cap) cap = p->max_iterations;
        long waves = (n + 15) / 16;
        if (waves > g_qp_quad_waves) waves = g_qp_quad_waves;
        const int *perm = nullptr;
        hipStream_t s_main = c->stream;
        // watchdog only: a wave's slots take their samples one after the other, each at most
        // cap passes and a start-up trip
        const long rounds = (n + 16 * waves - 1) / (16 * waves);
        const long max_trips = 16 * rounds * ((long)cap + 2) + 16;
        const int refill = g_qp_quad_refill < 1 ? 1 : (g_qp_quad_refill > 16 ? 16 : g_qp_quad_refill);
        // live hand-over (qp_live): the consumer launch of k_qp_wave goes out FIRST, on the side stream,
        // so that its waves (80 VGPRs) are resident when k_qp_quad fills the rest of every SIMD --
        // the quad kernel then runs at the four-waves-per-SIMD budget (128 VGPRs: 80 + 3 x 128 <= 512)
        const bool live = g_qp_live && cap < p->max_iterations && c->stream3 && KW == 32 && !defer_tail;
        QpLive lv{0, 0, 0u, 0u, nullptr, nullptr};
        if (live) {
            if ((long)c->qp_live_cap < n) {
                AA_CHECK_HIP(hipStreamSynchronize(c->stream));
                AA_CHECK_HIP(hipStreamSynchronize(c->stream3));
                AA_CHECK(c->qpLive.alloc((size_t)2 * n * sizeof(int)));       // zeroed: matches no epoch
                c->qp_live_cap = n;
                c->qp_live_epoch = 0;
            }
            if (c->qp_live_epoch == 0x7fffffff) {
                AA_CHECK_HIP(hipMemsetAsync(c->qpLive.p, 0, c->qpLive.bytes, c->stream));
                c->qp_live_epoch = 0;
            }
            lv.mode = 1;
            lv.epoch = ++c->qp_live_epoch;
            lv.producer_waves = (unsigned int)waves;
            lv.n = (unsigned int)n;
            lv.ready = c->qpLive.as<int>();
            lv.done = lv.ready + c->qp_live_cap;
            AA_CHECK_HIP(hipEventRecord(c->evFork, c->stream));
            AA_CHECK_HIP(hipStreamWaitEvent(c->stream3, c->evFork, 0));
            static bool lds_attr_set = false;
            if (!lds_attr_set) {
                AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_qp_wave_live),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, QP_LIVE_LDS));
                lds_attr_set = true;
            }
            hipLaunchKernelGGL(k_qp_wave_live, dim3((unsigned)g_qp_live_blocks), dim3(1024), QP_LIVE_LDS, c->stream3, A2d, Btall,
                               stride_j, stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                               (const int *)ovf_rows, (const QpCarry *)ovf, (double *)nullptr,
                               (const int *)nullptr, (const unsigned int *)nullptr, 1 << 30,
                               (unsigned int *)nullptr, (int *)nullptr, (QpCarry *)nullptr, lv);
            AA_CHECK_HIP(hipEventRecord(c->evJoin, c->stream3));
        }
        // (the ordering kernels run while the cross-stream dependency of the consumers resolves: the
        // consumers are resident on their CUs before k_qp_quad fills the chip)
        if (fused_order) perm = use_prefetched ? c->qpPerm.as<int>() : nullptr;   // (the kernel checks QpHeader::pad1)
        else if (will_sort) AA_CHECK(qp_order_rows(c, iters_dev, n, &perm, nullptr, 0, sort_hist != nullptr));
        const int quad_occ = live ? g_qp_live_occ : g_qp_quad_occ;
        // beside live consumers every wave of k_qp_quad asks for its share of a CU's LDS (see k_qp_wave_live)
        const unsigned quad_lds = live ? (unsigned)((QP_LIVE_LDS / (4 * quad_occ)) & ~511) : 0u;
        const int live_epoch = lv.epoch;
        int *ovf_ready = lv.ready;
#define QQK(KERN, MTV, M1V)                                                                         \
    do {                                                                                            \
        if (g_qp_quad_lazy && !live)                                                                \
            hipLaunchKernelGGL((KERN<MTV, M1V, true>), dim3((unsigned)waves), dim3(64), quad_lds, s_main, A2d, KW, Btall, \
                               stride_j, stride_t, bsd, Ztall, ldz, n, k, *p, cap, iters_dev, hdr, ovf_rows, ovf,   \
                               refill, perm, max_trips, live_epoch, ovf_ready, 0, 0L);                              \
        else                                                                                        \
            hipLaunchKernelGGL((KERN<MTV, M1V, false>), dim3((unsigned)waves), dim3(64), quad_lds, s_main, A2d, KW, Btall, \
                               stride_j, stride_t, bsd, Ztall, ldz, n, k, *p, cap, iters_dev, hdr, ovf_rows, ovf,   \
                               refill, perm, max_trips, live_epoch, ovf_ready, 0, 0L);                              \
    } while (0)
#define QQL(MTV, M1V)                                                                               \
    do {                                                                                            \
        if (quad_occ >= 4) QQK(k_qp_quad_w4, MTV, M1V);                                             \
        else if (quad_occ == 3) QQK(k_qp_quad_w3, MTV, M1V);                                        \
        else QQK(k_qp_quad, MTV, M1V);                                                              \
    } while (0)
        if (k <= 16) { if (p->memory <= 1) QQL(1, true); else QQL(1, false); }
        else         { if (p->memory <= 1) QQL(2, true); else QQL(2, false); }
#undef QQL
#undef QQK
        if (live) {
            // clean-up: whatever the consumers did not take (they give up after a bounded wait)
            AA_CHECK_HIP(hipStreamWaitEvent(c->stream, c->evJoin, 0));
            lv.mode = 2;
            QW32_LAUNCH(dim3(64), dim3(256), 0, c->stream, A2d, Btall,
                               stride_j, stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                               (const int *)ovf_rows, (const QpCarry *)ovf, (double *)nullptr,
                               (const int *)nullptr, (const unsigned int *)nullptr, 1 << 30,
                               (unsigned int *)nullptr, (int *)nullptr, (QpCarry *)nullptr, lv);
        } else if (cap < p->max_iterations) {
            // the parked samples, one wave each.  defer_tail: on the side stream, results to
            // tmpTall by slot; the caller's Z'X pass runs beside them on the weights as k_qp_quad
            // left them and launch_qp_tail_fixup adds the rank-m correction afterwards -- the
            // dependent chain of the longest sample (100-450 passes at ~0.9 us) hides behind an
            // HBM-bound pass instead of idling the chip
            const bool defer = defer_ok;
            const int tail_cap = g_qp_tail_cap;
            if (defer && tail_cap > cap && tail_cap < p->max_iterations) {
                // two stages: everything up to tail_cap passes here, at full speed on the whole chip;
                // the few samples beyond it on the side stream beside the caller's Z'X pass
                if (fused_order) {
                    const QpOrder od{iters_dev, c->qpPerm.as<int>(), c->qpPerm.as<int>() + n, n, order_blocks};
                    const dim3 og((unsigned)(g_qp_wave_blocks + order_blocks));
                    if (g_qp_wave_lazy)
                        hipLaunchKernelGGL(k_qp_wave_ord<true>, og, dim3(256), 0, c->stream, od, A2d, Btall, stride_j,
                                           stride_t, bsd, Ztall, ldz, k, *p, iters_dev, hdr, (const int *)ovf_rows,
                                           (const QpCarry *)ovf, tail_cap, &hdr->pad, ovf2_rows, ovf2);
                    else
                        hipLaunchKernelGGL(k_qp_wave_ord<false>, og, dim3(256), 0, c->stream, od, A2d, Btall, stride_j,
                                           stride_t, bsd, Ztall, ldz, k, *p, iters_dev, hdr, (const int *)ovf_rows,
                                           (const QpCarry *)ovf, tail_cap, &hdr->pad, ovf2_rows, ovf2);
                    c->qp_perm_ready = true;
                    c->qp_perm_n = n;
                } else
                QW32_LAUNCH(dim3((unsigned)g_qp_wave_blocks), dim3(256), 0, c->stream, A2d, Btall,
                                   stride_j, stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                                   (const int *)ovf_rows, (const QpCarry *)ovf, (double *)nullptr,
                                   (const int *)nullptr, (const unsigned int *)nullptr, tail_cap, &hdr->pad, ovf2_rows, ovf2);
                // the handful beyond tail_cap on the side stream.  (Tried: a CU-masked stream pair,
                // hipExtStreamCreateWithCUMask -- 8 or 16 CUs for these chains alone, the rest of the chip for
                // the caller's pass, so that the two share no SIMD.  With masked queues alive EVERY launch of the
                // process, on any stream, took ~45 us longer (k_qp_setup 6 -> 50 us) and the iteration went
                // from 2.0 to 3.0-3.2 ms: profiles/round4_ab.txt.)
                hipStream_t ss = c->stream2;
                AA_CHECK_HIP(hipEventRecord(c->evFork, c->stream));
                AA_CHECK_HIP(hipStreamWaitEvent(ss, c->evFork, 0));
                QW32_LAUNCH(dim3(64), dim3(256), 0, ss, A2d, Btall,
                                   stride_j, stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                                   (const int *)ovf2_rows, (const QpCarry *)ovf2, c->tmpTall.as<double>(),
                                   (const int *)nullptr, (const unsigned int *)&hdr->pad);
                AA_CHECK_HIP(hipEventRecord(c->evJoin, ss));
                c->qp_tail_pending = true;
                c->qp_tail_rows = ovf2_rows;
                c->qp_tail_count = &hdr->pad;
            } else {
                hipStream_t s2 = c->stream;
                double *zslot = nullptr;
                if (defer) {
                    AA_CHECK_HIP(hipEventRecord(c->evFork, c->stream));
                    AA_CHECK_HIP(hipStreamWaitEvent(c->stream2, c->evFork, 0));
                    s2 = c->stream2;
                    zslot = c->tmpTall.as<double>();
                }
                if (fused_order && !defer) {
                    const QpOrder od{iters_dev, c->qpPerm.as<int>(), c->qpPerm.as<int>() + n, n, order_blocks};
                    const dim3 og((unsigned)(g_qp_wave_blocks + order_blocks));
                    if (g_qp_wave_lazy)
                        hipLaunchKernelGGL(k_qp_wave_ord<true>, og, dim3(256), 0, s2, od, A2d, Btall, stride_j, stride_t,
                                           bsd, Ztall, ldz, k, *p, iters_dev, hdr, (const int *)ovf_rows,
                                           (const QpCarry *)ovf, 1 << 30, (unsigned int *)nullptr, (int *)nullptr,
                                           (QpCarry *)nullptr, g_qp_wave_queue);
                    else
                        hipLaunchKernelGGL(k_qp_wave_ord<false>, og, dim3(256), 0, s2, od, A2d, Btall, stride_j, stride_t,
                                           bsd, Ztall, ldz, k, *p, iters_dev, hdr, (const int *)ovf_rows,
                                           (const QpCarry *)ovf, 1 << 30, (unsigned int *)nullptr, (int *)nullptr,
                                           (QpCarry *)nullptr, g_qp_wave_queue);
                    c->qp_perm_ready = true;
                    c->qp_perm_n = n;
                } else
                QW32_LAUNCH(dim3((unsigned)g_qp_wave_blocks), dim3(256), 0, s2, A2d, Btall, stride_j,
                                   stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                                   (const int *)ovf_rows, (const QpCarry *)ovf, zslot);
                if (defer) {
                    AA_CHECK_HIP(hipEventRecord(c->evJoin, c->stream2));
                    c->qp_tail_pending = true;
                    c->qp_tail_rows = ovf_rows;
                    c->qp_tail_count = &hdr->n_overflow;
                }
            }
        }
    } else if (wave_only) {
        long blocks = (n + 3) / 4;
        if (blocks > 2048) blocks = 2048;
#define QWF(KWV, M1V)                                                                            \
    hipLaunchKernelGGL((k_qp_wave<KWV, M1V>), dim3((unsigned)blocks), dim3(256), 0, c->stream, A2d, Btall, \
                       stride_j, stride_t, bsd, Ztall, ldz, n, k, *p, iters_dev, hdr,               \
                       (const int *)ovf_rows, (const QpCarry *)ovf, (double *)nullptr)
        if (KW == 64) { if (p->memory <= 1) QWF(64, true); else QWF(64, false); }
        else          { if (p->memory <= 1) QWF(32, true); else QWF(32, false); }
#undef QWF
    } else {
        // phase 1: every sample gets up to pass_cap passes in a lane
        int cap = qp_pass_cap();
        if (p->memory > 1 || p->max_iterations <= cap) cap = p->max_iterations;
        // one wave per SIMD: lanes that finish early pull further samples, so a wave's
        // trip count is the sum over ~n/65536 samples per lane instead of two full rounds
        long waves = (n + 63) / 64;
        if (waves > g_qp_waves) waves = g_qp_waves;
        dim3 grid((unsigned)waves);
        // samples ordered by their previous pass count (iters_dev still holds the counts of
        // the previous update of this context; the kernels below overwrite them)
        const int *perm = nullptr;
        if (will_sort) AA_CHECK(qp_order_rows(c, iters_dev, n, &perm, nullptr, 0, sort_hist != nullptr));
        QpDebug *dbgp = g_qp_profile ? reinterpret_cast<QpDebug *>(base + 64) : (QpDebug *)nullptr;
        if (dbgp) AA_CHECK_HIP(hipMemsetAsync(dbgp, 0, sizeof(QpDebug), c->stream));
#define QPL4(KQV, FULLV, PROFV, SGV)                                                          \
    hipLaunchKernelGGL((k_qp<KQV, FULLV, PROFV, SGV>), grid, dim3(64), 0, c->stream, Ad, Btall, stride_j,  \
                       stride_t, bsd, Ztall, ldz, n, k, *p, cap, iters_dev, hdr, ovf_rows, ovf,  \
                       g_qp_refill_min, dbgp, perm)
#define QPL3(KQV, FULLV, PROFV)                                                               \
    do {                                                                                      \
        if (g_qp_matvec == 1 && KQV >= 16) QPL4(KQV, FULLV, PROFV, true);                     \
        else QPL4(KQV, FULLV, PROFV, false);                                                  \
    } while (0)
#define QPL(KQV)                                                                              \
    do {                                                                                      \
        if (dbgp) { if (k == KQV) QPL3(KQV, true, true); else QPL3(KQV, false, true); }       \
        else      { if (k == KQV) QPL3(KQV, true, false); else QPL3(KQV, false, false); }     \
    } while (0)
        switch (KQ) { case 4: QPL(4); break; case 8: QPL(8); break; case 16: QPL(16); break;
                      default: QPL(32); break; }
#undef QPL4
#undef QPL3
#undef QPL
        if (cap < p->max_iterations) {
            // phase 2: the stragglers, one wave each (grid is fixed; the count is read on
            // the device, so no host synchronisation between the phases)
            const bool defer = defer_tail && !stats && c->stream2 && ldz == c->KP && KW == 32 &&
                               c->dtype == AA_F32 &&
                               c->tmpTall.bytes >= (size_t)n * 32 * sizeof(double);
            hipStream_t s2 = c->stream;
            double *zslot = nullptr;
            if (defer) {
                AA_CHECK_HIP(hipEventRecord(c->evFork, c->stream));
                AA_CHECK_HIP(hipStreamWaitEvent(c->stream2, c->evFork, 0));
                s2 = c->stream2;
                zslot = c->tmpTall.as<double>();
            }
            QW32_LAUNCH(dim3((unsigned)g_qp_wave_blocks), dim3(256), 0, s2, A2d, Btall, stride_j,
                               stride_t, bsd, Ztall, ldz, (long)-1, k, *p, iters_dev, hdr,
                               (const int *)ovf_rows, (const QpCarry *)ovf, zslot);
            if (defer) {
                AA_CHECK_HIP(hipEventRecord(c->evJoin, c->stream2));
                c->qp_tail_pending = true;
                c->qp_tail_rows = ovf_rows;
                c->qp_tail_count = &hdr->n_overflow;
            }
        }
    }
    AA_CHECK_HIP(hipGetLastError());
    if (stats) {
        QpHeader h;
        AA_CHECK_HIP(hipMemcpyAsync(&h, hdr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
        stats->total_passes = (long)h.total_passes;
        stats->max_passes = (int)h.max_passes;
        stats->reserved = (int)h.n_overflow;
        if (g_qp_profile && row_mode && h.dbg_waves) {
            unsigned long long d5[5];
            AA_CHECK_HIP(ctx_memcpy(c, d5, base + 64, sizeof(d5), hipMemcpyDeviceToHost));
            fprintf(stderr, "[qp_profile] row kernel waves: slowest %.0f kcycles (its trips %llu), mean %.0f kcycles; "
                    "most trips %llu (that wave: %.0f kcycles = %.0f cycles per trip)\n",
                    (double)(d5[1] >> 20) / 1e3, d5[1] & 0xfffffull, (double)d5[3] / h.dbg_waves / 1e3,
                    d5[4] >> 40, (double)(d5[4] & ((1ull << 40) - 1ull)) / 1e3,
                    (double)(d5[4] & ((1ull << 40) - 1ull)) / (double)(d5[4] >> 40));
        }
        if (g_qp_profile && row_mode && h.dbg_waves)
            fprintf(stderr, "[qp_profile] row kernel: %u waves, %.1f trips per wave, %.2f Michelot rounds per "
                    "row and trip, %.2f passes per sample, %u predicted-long samples on the side stream, "
                    "%u handed over at the pass cap\n", h.dbg_waves, (double)h.dbg_trips / h.dbg_waves,
                    (double)h.dbg_rounds / (4.0 * h.dbg_trips), (double)h.total_passes / (double)n, h.n_long,
                    h.n_overflow);
        if (g_qp_profile && quad_mode && h.dbg_trips)
            fprintf(stderr, "[qp_profile] four-lane kernel, lazy stopping test: %u trips owed a test, %u of them (%.1f %%) "
                    "skipped it for the whole wave\n", h.dbg_trips, h.dbg_rounds, 100.0 * h.dbg_rounds / h.dbg_trips);
        if (g_qp_profile && !row_mode) {
            QpDebug d;
            AA_CHECK_HIP(ctx_memcpy(c, &d, base + 64, sizeof(d), hipMemcpyDeviceToHost));
            if (d.waves)
                fprintf(stderr, "[qp_profile] waves %llu trips/wave %.1f refills/wave %.1f cycles/wave %.0f "
                        "(projection part %.1f%%, mat-vec %.1f%%, step %.1f%%, residual+finish %.1f%%) cycles/trip %.0f\n",
                        d.waves, (double)d.trips / d.waves, (double)d.refills / d.waves,
                        (double)d.cyc_total / d.waves, 100.0 * d.cyc_proj / d.cyc_total,
                        100.0 * d.cyc_matvec / d.cyc_total, 100.0 * d.cyc_step / d.cyc_total,
                        100.0 * d.cyc_fin / d.cyc_total, (double)d.cyc_total / d.trips);
            if (d.waves && d.proj_lanes)
                fprintf(stderr, "[qp_profile] Michelot rounds per projection: lane mean %.2f, wave max (residual "
                        "projection) %.2f\n", (double)d.proj_rounds_lanesum / d.proj_lanes,
                        (double)d.proj_rounds_wavemax / d.proj_calls);
        }
    }
    return AA_OK;
}

// commit of the deferred stragglers: Z[row_s] = z_new[s]
__global__ __launch_bounds__(256) void k_qp_commit_tail(const unsigned int *__restrict__ count_dev,
                                                        const int *__restrict__ rows,
                                                        const double *__restrict__ zslot,
                                                        double *__restrict__ Z, int KP, int k)
{
    const unsigned int count = *count_dev;
    const int comp = threadIdx.x % KP;
    for (unsigned int s = blockIdx.x * (256 / KP) + threadIdx.x / KP; s < count; s += gridDim.x * (256 / KP))
        if (comp < k) Z[(size_t)rows[s] * KP + comp] = zslot[(size_t)s * KP + comp];
}

int launch_qp_tail_fixup(Ctx *c, double *Ztall)
{
    if (!c->qp_tail_pending) return AA_OK;
    c->qp_tail_pending = false;
    AA_CHECK_HIP(hipStreamWaitEvent(c->stream, c->evJoin, 0));
    AA_CHECK(launch_reduce_rows_fixup(c, c->qp_tail_count, c->qp_tail_rows, c->tmpTall.as<double>(), Ztall));
    hipLaunchKernelGGL(k_qp_commit_tail, dim3(64), dim3(256), 0, c->stream, c->qp_tail_count, c->qp_tail_rows,
                       (const double *)c->tmpTall.as<double>(), Ztall, c->KP, c->k);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// ---------------------------------------------------------------------------
// stateless row-wise projection of an arbitrary rows x cols matrix (row-major).
// One wave per row (cols <= 2048) or one block per row; Michelot passes over the row.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

template <int NW>   // waves per row (1 or 4)
__global__ __launch_bounds__(64 * NW) void k_simplex_rows(const double *__restrict__ in,
                                                          double *__restrict__ out, long rows,
                                                          long cols)
{
    __shared__ double sm[2 * NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row = blockIdx.x;
    if (row >= rows) return;
    const double *x = in + row * cols;
    constexpr int NT = 64 * NW;

    double mx = -INFINITY;
    for (long c = tid; c < cols; c += NT) mx = fmax(mx, x[c]);
    mx = wave_max_d(mx);
    if (NW > 1) {
        if (lane == 0) sm[wave] = mx;
        __syncthreads();
        mx = sm[0];
        for (int w = 1; w < NW; ++w) mx = fmax(mx, sm[w]);
        __syncthreads();
    }
    double t = mx - 1.0;
    double prev = 0.0;
    for (long pass = 0; pass < cols + 2; ++pass) {
        double s = 0.0, cnt = 0.0;
        for (long c = tid; c < cols; c += NT) {
            const double w = x[c];
            if (w > t) {
                s += w;
                cnt += 1.0;
            }
        }
        s = wave_sum_d(s);
        cnt = wave_sum_d(cnt);
        if (NW > 1) {
            if (lane == 0) {
                sm[wave] = s;
                sm[NW + wave] = cnt;
            }
            __syncthreads();
            s = sm[0];
            cnt = sm[NW];
            for (int w = 1; w < NW; ++w) {
                s += sm[w];
                cnt += sm[NW + w];
            }
            __syncthreads();
        }
        const bool conv = (prev > 0.0) && (cnt >= prev);
        t = (s - 1.0) / cnt;
        prev = cnt;
        if (conv) break;     // uniform across the block: s, cnt are identical in every thread
    }
    for (long c = tid; c < cols; c += NT) out[row * cols + c] = fmax(x[c] - t, 0.0);
}

int launch_simplex_rows_generic(hipStream_t s, const double *in, double *out, long rows, long cols)
{
    if (rows <= 0 || cols <= 0) return AA_OK;
    if (cols <= 2048)
        hipLaunchKernelGGL(k_simplex_rows<1>, dim3((unsigned)rows), dim3(64), 0, s, in, out, rows, cols);
    else
        hipLaunchKernelGGL(k_simplex_rows<4>, dim3((unsigned)rows), dim3(256), 0, s, in, out, rows, cols);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

}  // namespace aa
