// kernels_gemm.hip -- the two skinny-GEMM shapes of the AA/GPNH solver against the
// resident data matrix X [n_pad][ldx] (gfx950 / CDNA4 only).
//
//   reduce-over-rows  out[i][c] = sum_r A[r][i] * X[r][c]      (k x p result)
//       C X   archetypal_analysis.py:262,297,544,618     (A = C', tall)
//       D X   (search direction; gives (C + lambda D) X by linearity)
//       Z'X   archetypal_analysis.py:548,641; gpnh_convex_coding.py:219
//       C K   archetypal_analysis.py:208,288,409,484     (kernel form)
//   row-local         out[r][i] = sum_c X[r][c] * B[i][c]      (n x k result)
//       (CX) X'   archetypal_analysis.py:299,545,619
//       X (X'Z)   archetypal_analysis.py:549,642
//       X W       gpnh_convex_coding.py:270,292,352
//       K Z       archetypal_analysis.py:411,502             (kernel form)
//
// float32 (v_mfma_f32_32x32x2_f32): every X element is read exactly once per pass.
// Algorithmic traffic per pass = n*p*4 bytes; 2*k*n*p flop; at k = 32 the intensity is
// 16 flop/B (< the 19.7 flop/B ridge) => HBM-bound.  reduce-over-rows loads X straight
// from HBM into the MFMA operand layout; row-local contracts along the contiguous axis and
// stages X through wave-private LDS tiles ("wave-streaming", k_row_local_f32_ws; older
// variants stay selectable).
// float64, the reference dtype (v_mfma_f64_16x16x4_f64): the same two structures,
// k_reduce_rows_f64_mfma and k_row_local_f64_ws / _mfma; the f64 VALU kernels they replaced
// remain behind aa_set_option("f64_mfma", 0).
#include "aa_internal.h"

namespace aa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// XCD-aware (column group, row slab) of a block.  Workgroups are dealt round-robin over the 8
// XCDs (blocks b and b + 8 share one; MI355X_MICROARCH.md, Workgroup dispatch), and every XCD
// has its own L2.  With the natural map (column group = blockIdx.x) all blocks of a column
// group sit on one XCD, so each of the 8 XCDs pulls the WHOLE tall operand through its L2
// (8 x 25.6 MB at the headline size: the 1.19x traffic of round 1).  Here the blocks of one row
// slab -- which read the same rows of the tall operand -- are given linear ids that agree
// modulo 8, so every slab of the tall operand is fetched by one XCD only.  Placement is a speed
// matter only: any map that is a bijection is correct.
__device__ __forceinline__ void xcd_aware_block(int &colgroup, int &slab)
{
    const int ncg = (int)gridDim.x, nsl = (int)gridDim.y;
    colgroup = (int)blockIdx.x;
    slab = (int)blockIdx.y;
    if (nsl % 8 == 0) {
        const int b = (int)blockIdx.x + ncg * (int)blockIdx.y;   // dispatch order: x fastest
        const int q = b >> 3, r = b & 7;
        slab = r + 8 * (q / ncg);
        colgroup = q % ncg;
    }
}

// ---------------------------------------------------------------------------
// reduce over rows, float32 MFMA.
// grid = (ceil(p_pad/512), nslab); block = 4 waves; wave w owns the 128-column
// strip c0 = (4*blockIdx.x + w)*128 and rows [slab*rows_per_slab, +rows_per_slab).
// MFMA 32x32x2: A-operand lane l = A[row r0+2u+(l>>5)][component l&31],
// B-operand lane l = X[row r0+2u+(l>>5)][c0 + 4*(l&31) + m]  (one dwordx4 load feeds
// the four MFMAs m = 0..3, whose output columns are c0 + 4*j + m).
// ---------------------------------------------------------------------------
template <int NCT, int U>
__global__ __launch_bounds__(256) void k_reduce_rows_f32(const float *__restrict__ X, long ldx,
                                                         const double *__restrict__ A,
                                                         long rows_per_slab, long n_pad, int p_pad,
                                                         float *__restrict__ partial)
{
    constexpr int KP = 32 * NCT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int colgroup, slab;
    xcd_aware_block(colgroup, slab);
    const int c0 = (colgroup * 4 + wave) * 128;
    if (c0 >= p_pad) return;   // wave-uniform; the kernel has no barriers
    const int h = lane >> 5, j = lane & 31;
    const long r_begin = (long)slab * rows_per_slab;
    long r_end = r_begin + rows_per_slab;
    if (r_end > n_pad) r_end = n_pad;

    f32x16 acc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][m][e] = 0.f;

    // Two register sets, software-pipelined one half-step (2*U rows) ahead: while the
    // MFMAs of one set run, the U dwordx4 loads of the other are in flight.  The
    // sched_barriers keep hipcc from sinking the loads down to their first use.
    const float *xp = X + (r_begin + h) * ldx + c0 + 4 * j;
    const double *ap = A + (r_begin + h) * KP + j;
    f32x4 xa[U], xb[U];
    double aa[U][NCT], ab[U][NCT];
#define RR_LOAD(XV, AV, ROWOFF)                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
        XV[u] = *reinterpret_cast<const f32x4 *>(xp + (long)((ROWOFF) + 2 * u) * ldx);        \
        _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct)                                    \
            AV[u][ct] = ap[((ROWOFF) + 2 * u) * KP + ct * 32];                                \
    }
#define RR_COMPUTE(XV, AV)                                                                   \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                             \
        _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) {                                  \
            const float af = (float)AV[u][ct];                                                \
            _Pragma("unroll") for (int m = 0; m < 4; ++m)                                     \
                acc[ct][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, XV[u][m], acc[ct][m],  \
                                                                  0, 0, 0);                   \
        }
    RR_LOAD(xa, aa, 0)
    for (long r = r_begin; r < r_end; r += 4 * U) {
        RR_LOAD(xb, ab, 2 * U)      // may run past r_end: X and A carry AA_SLACK_ROWS zero rows
        __builtin_amdgcn_sched_barrier(0);
        RR_COMPUTE(xa, aa)
        __builtin_amdgcn_sched_barrier(0);
        RR_LOAD(xa, aa, 4 * U)
        __builtin_amdgcn_sched_barrier(0);
        RR_COMPUTE(xb, ab)          // the row range is a multiple of 4*U
        __builtin_amdgcn_sched_barrier(0);
        xp += (long)(4 * U) * ldx;
        ap += (4 * U) * KP;
    }
#undef RR_LOAD
#undef RR_COMPUTE

    // D layout (32x32): column = lane&31 (-> 4 data columns c0+4j+m), row (component)
    // = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    float *out = partial + (size_t)slab * KP * p_pad;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int comp = ct * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            f32x4 v = {acc[ct][0][reg], acc[ct][1][reg], acc[ct][2][reg], acc[ct][3][reg]};
            *reinterpret_cast<f32x4 *>(out + (size_t)comp * p_pad + c0 + 4 * j) = v;
        }
}

// reduce over rows, float64 VALU: thread = data column, KP accumulators in registers,
// A[r][*] is wave-uniform (scalar loads).
template <int KP>
__global__ __launch_bounds__(256) void k_reduce_rows_f64(const double *__restrict__ X, long ldx,
                                                         const double *__restrict__ A,
                                                         long rows_per_slab, long n_pad, int p_pad,
                                                         double *__restrict__ partial)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p_pad) return;
    const long r_begin = (long)blockIdx.y * rows_per_slab;
    long r_end = r_begin + rows_per_slab;
    if (r_end > n_pad) r_end = n_pad;
    double acc[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) acc[i] = 0.0;
    for (long r = r_begin; r < r_end; ++r) {
        const double x = X[r * ldx + c];
        const double *a = A + r * KP;
#pragma unroll
        for (int i = 0; i < KP; ++i) acc[i] = fma(a[i], x, acc[i]);
    }
    double *out = partial + (size_t)blockIdx.y * KP * p_pad;
#pragma unroll
    for (int i = 0; i < KP; ++i) out[(size_t)i * p_pad + c] = acc[i];
}

// ---------------------------------------------------------------------------
// float64 data (the reference dtype) on the f64 matrix cores, v_mfma_f64_16x16x4_f64:
// A-operand lane l = Aop[m = l&15][k = l>>4], B-operand lane l = Bop[n = l&15][k = l>>4],
// D lane l, reg r = D[m = (l>>4) + 4r][n = l&15]   (D = Aop * Bop').
// ---------------------------------------------------------------------------
typedef double f64x4g __attribute__((ext_vector_type(4)));
typedef double f64x2g __attribute__((ext_vector_type(2)));

// reduce over rows: D[m = component][n = column] += sum over 4 rows.  A wave owns a
// 64-column strip and a row slab; per 4 rows a lane loads NT doubles of the tall operand and
// two 16-byte pieces of X (columns c0 + 32u + 2*(l&15) + e: the two doubles of a piece feed
// the B operands of two different column tiles, so the loads are 256 contiguous bytes per
// row and quarter-wave).  Two register sets, software-pipelined as in k_reduce_rows_f32.
template <int NT>
__global__ __launch_bounds__(256) void k_reduce_rows_f64_mfma(const double *__restrict__ X, long ldx,
                                                              const double *__restrict__ A,
                                                              long rows_per_slab, long n_pad,
                                                              int p_pad, double *__restrict__ partial)
{
    constexpr int KP = 16 * NT, U = 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int colgroup, slab;
    xcd_aware_block(colgroup, slab);
    const int c0 = (colgroup * 4 + wave) * 64;
    if (c0 >= p_pad) return;   // wave-uniform; the kernel has no barriers
    const int lc = lane & 15, lr = lane >> 4;
    const long r_begin = (long)slab * rows_per_slab;
    long r_end = r_begin + rows_per_slab;
    if (r_end > n_pad) r_end = n_pad;

    f64x4g acc[NT][2][2];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[ti][u][e] = (f64x4g){0.0, 0.0, 0.0, 0.0};

    const double *xp = X + (r_begin + lr) * ldx + c0 + 2 * lc;
    const double *ap = A + (r_begin + lr) * KP + lc;
    f64x2g xa[U][2], xb[U][2];
    double aa[U][NT], ab[U][NT];
#define RD_LOAD(XV, AV, ROWOFF)                                                              \
    _Pragma("unroll") for (int q = 0; q < U; ++q) {                                           \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                         \
            XV[q][u] = *reinterpret_cast<const f64x2g *>(xp + (long)((ROWOFF) + 4 * q) * ldx + 32 * u); \
        _Pragma("unroll") for (int ti = 0; ti < NT; ++ti)                                     \
            AV[q][ti] = ap[((ROWOFF) + 4 * q) * KP + 16 * ti];                                \
    }
#define RD_COMPUTE(XV, AV)                                                                   \
    _Pragma("unroll") for (int q = 0; q < U; ++q)                                             \
        _Pragma("unroll") for (int ti = 0; ti < NT; ++ti)                                     \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                     \
                _Pragma("unroll") for (int e = 0; e < 2; ++e)                                 \
                    acc[ti][u][e] = __builtin_amdgcn_mfma_f64_16x16x4f64(AV[q][ti], XV[q][u][e],   \
                                                                         acc[ti][u][e], 0, 0, 0);
    RD_LOAD(xa, aa, 0)
    for (long r = r_begin; r < r_end; r += 8 * U) {
        RD_LOAD(xb, ab, 4 * U)      // may run past r_end: X and A carry AA_SLACK_ROWS zero rows
        __builtin_amdgcn_sched_barrier(0);
        RD_COMPUTE(xa, aa)
        __builtin_amdgcn_sched_barrier(0);
        RD_LOAD(xa, aa, 8 * U)
        __builtin_amdgcn_sched_barrier(0);
        RD_COMPUTE(xb, ab)          // the row range is a multiple of 8*U
        __builtin_amdgcn_sched_barrier(0);
        xp += (long)(8 * U) * ldx;
        ap += (8 * U) * KP;
    }
#undef RD_LOAD
#undef RD_COMPUTE
    double *out = partial + (size_t)slab * KP * p_pad;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int comp = 16 * ti + lr + 4 * reg;
                f64x2g v = {acc[ti][u][0][reg], acc[ti][u][1][reg]};
                *reinterpret_cast<f64x2g *>(out + (size_t)comp * p_pad + c0 + 32 * u + 2 * lc) = v;
            }
}

// row-local: D[m = row][n = component].  Block = 64 rows (a wave per 16), both operands
// staged through LDS in 32-column tiles with coalesced 16-byte loads, register-prefetched one
// tile ahead; LDS row stride 34 doubles => conflict-free fragment reads.
// Short, wide matrices (n = 1610, p = 25 000: the HadISST shape) have too few 64-row blocks to
// fill the chip (26 of them: 0.7 ms per pass for 322 MB), so the contraction is also split over
// blockIdx.y: chunk `col_chunk` columns per block, partial results to out + blockIdx.y * n_pad *
// KP, summed in a fixed order by k_sum_chunks.  col_chunk = p_pad and gridDim.y = 1: unsplit.
template <int NT>
__global__ __launch_bounds__(256) void k_row_local_f64_mfma(const double *__restrict__ X, long ldx,
                                                            const double *__restrict__ B, int p_pad,
                                                            double *__restrict__ out, long n_pad,
                                                            int col_chunk)
{
    constexpr int KP = 16 * NT, TC = 32, LS = 34;
    const int c_begin = (int)blockIdx.y * col_chunk;
    int c_end = c_begin + col_chunk;
    if (c_end > p_pad) c_end = p_pad;
    out += (size_t)blockIdx.y * n_pad * KP;
    constexpr int NB = KP * 16 / 256;                 // B chunks per thread and tile
    __shared__ __attribute__((aligned(16))) double xs[64 * LS];
    __shared__ __attribute__((aligned(16))) double bs[KP * LS];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int lc = lane & 15, lr = lane >> 4;
    const long r0 = (long)blockIdx.x * 64;            // grid = n_pad / 64 exactly
    f64x4g acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f64x4g){0.0, 0.0, 0.0, 0.0};

    f64x2g sx[4], sb[NB];
    auto load_tile = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cid = t + 256 * e;
            sx[e] = *reinterpret_cast<const f64x2g *>(X + (r0 + (cid >> 4)) * ldx + c0 + 2 * (cid & 15));
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int cid = t + 256 * e;
            sb[e] = *reinterpret_cast<const f64x2g *>(B + (long)(cid >> 4) * p_pad + c0 + 2 * (cid & 15));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cid = t + 256 * e;
            *reinterpret_cast<f64x2g *>(xs + (cid >> 4) * LS + 2 * (cid & 15)) = sx[e];
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int cid = t + 256 * e;
            *reinterpret_cast<f64x2g *>(bs + (cid >> 4) * LS + 2 * (cid & 15)) = sb[e];
        }
    };
    if (c_begin < c_end) load_tile(c_begin);
    for (int c0 = c_begin; c0 < c_end; c0 += TC) {
        store_tile();
        __syncthreads();
        if (c0 + TC < c_end) load_tile(c0 + TC);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < TC / 4; ++s) {
            const double a = xs[(wave * 16 + lc) * LS + 4 * s + lr];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bs[(16 * nt + lc) * LS + 4 * s + lr],
                                                               acc[nt], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            out[(r0 + wave * 16 + lr + 4 * reg) * KP + 16 * nt + lc] = acc[nt][reg];
}

// out[e] = sum over the column chunks of partial[s][e], s in increasing order (deterministic)
__global__ __launch_bounds__(256) void k_sum_chunks(const double *__restrict__ partial, long elems,
                                                    int nsplit, double *__restrict__ out)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    double s = partial[e];
    for (int q = 1; q < nsplit; ++q) s += partial[(size_t)q * elems + e];
    out[e] = s;
}

// row-local, float64, "wave-streaming" form (the structure of k_row_local_f32_ws): a block is
// W waves on one CU, every wave owns 32 rows (two 16-row MFMA tiles) for the whole kernel
// and streams them through a wave-private LDS tile of 32 columns (256-byte row segments in,
// fragments out, no barrier), one tile register-prefetched ahead; the small operand is shared
// as double-buffered 64-column slabs (one barrier per slab).  Row strides of 34 / 66 doubles
// make the fragment reads conflict free.  Dynamic LDS = 2 * KP * 66 * 8 + W * 32 * 34 * 8 B.
template <int NT>
__global__ __launch_bounds__(NT == 2 ? 1024 : 768) void k_row_local_f64_ws(const double *__restrict__ X, long ldx,
                                                                             const double *__restrict__ B,
                                                                             int p_pad, double *__restrict__ out,
                                                                             long n_pad, int W)
{
    constexpr int KP = 16 * NT, SB = 64, TC = 32, XS = 34, BS = 66;
    constexpr int NBI = NT;                          // B chunks per thread and slab at >= 512 threads
    extern __shared__ __attribute__((aligned(16))) double wsd_smem[];
    const int t = threadIdx.x, lane = t & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double *bs = wsd_smem;
    double *xs = wsd_smem + 2 * KP * BS + wave * (32 * XS);
    const long r0 = ((long)blockIdx.x * W + wave) * 32;
    const bool active = r0 < n_pad;
    const long r0c = active ? r0 : n_pad - 32;       // surplus waves recompute the last tile
    const int lc = lane & 15, lr = lane >> 4;

    f64x4g acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f64x4g){0.0, 0.0, 0.0, 0.0};

    f64x2g sb[NBI], sx[8];
    auto load_b = [&](int c0) {
#pragma unroll
        for (int e = 0; e < NBI; ++e) {
            const int cid = (t + e * nthreads) & (KP * 32 - 1);     // 32 chunks of 16 B per component
            sb[e] = *reinterpret_cast<const f64x2g *>(B + (long)(cid >> 5) * p_pad + c0 + 2 * (cid & 31));
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NBI; ++e) {
            const int cid = (t + e * nthreads) & (KP * 32 - 1);
            *reinterpret_cast<f64x2g *>(bs + buf * (KP * BS) + (cid >> 5) * BS + 2 * (cid & 31)) = sb[e];
        }
    };
    const int xr = lane >> 4, xc = lane & 15;
    const double *gx = X + (r0c + xr) * ldx + 2 * xc;
    auto load_x = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            sx[e] = *reinterpret_cast<const f64x2g *>(gx + (long)(4 * e) * ldx + c0);
    };
    auto store_x = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *reinterpret_cast<f64x2g *>(xs + (4 * e + xr) * XS + 2 * xc) = sx[e];
    };
    auto compute = [&](int buf, int half) {
        const double *bb = bs + buf * (KP * BS) + half * TC;
#pragma unroll
        for (int s = 0; s < TC / 4; ++s) {
            double bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bb[(16 * nt + lc) * BS + 4 * s + lr];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const double a = xs[(16 * mt + lc) * XS + 4 * s + lr];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    };

    const int nslab = p_pad / SB;
    load_b(0);
    load_x(0);
    store_b(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int c0 = s * SB;
        const int snext = s + 1 < nslab ? s + 1 : s;       // last slab: harmless reload
        store_x();                                // wave-private: LDS is in order per wave
        load_x(c0 + TC);
        load_b(snext * SB);
        __builtin_amdgcn_sched_barrier(0);
        compute(s & 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        store_x();
        store_b((s + 1) & 1);                     // that buffer was last read before the previous barrier
        load_x(snext * SB);
        __builtin_amdgcn_sched_barrier(0);
        compute(s & 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    out[(r0 + 16 * mt + lr + 4 * reg) * KP + 16 * nt + lc] = acc[mt][nt][reg];
    }
}

// second stage of the split-row reduction: fixed summation order => deterministic.
template <typename TP, typename TO>
__global__ __launch_bounds__(256) void k_reduce_partials(const TP *__restrict__ partial, long nslab,
                                                         long elems, double *__restrict__ out,
                                                         TO *__restrict__ outT)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= elems) return;
    double s = 0.0;
#pragma unroll 16
    for (long sl = 0; sl < nslab; ++sl) s += (double)partial[sl * elems + idx];
    out[idx] = s;
    if (outT) outT[idx] = (TO)s;
}

// ---------------------------------------------------------------------------
// row-local, float32 MFMA.  wave owns 32*RT rows; contraction over the columns.
// A-operand lane l = X[r0 + 32*rt + (l&31)][c + 8u + 4*(l>>5) + m]
// B-operand lane l = B[32*ct + (l&31)][c + 8u + 4*(l>>5) + m]      (m = 0..3 from one
// dwordx4 load each); D[row][component]: component = lane&31, row = (reg&3)+8*(reg>>2)+4*(l>>5).
// ---------------------------------------------------------------------------
template <int NCT, int RT>
__global__ __launch_bounds__(256) void k_row_local_f32(const float *__restrict__ X, long ldx,
                                                       const float *__restrict__ B, int p_pad,
                                                       double *__restrict__ out, long n_pad)
{
    constexpr int KP = 32 * NCT;
    constexpr int U = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 4 + wave) * (32 * RT);
    if (r0 >= n_pad) return;
    const int h = lane >> 5, j = lane & 31;

    f32x16 acc[RT][NCT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.f;

    const float *xp = X + (r0 + j) * ldx + 4 * h;
    const float *bp = B + (long)j * p_pad + 4 * h;
    for (int c = 0; c < p_pad; c += 8 * U) {
        f32x4 xv[U][RT], bv[U][NCT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                xv[u][rt] = *reinterpret_cast<const f32x4 *>(xp + (long)(rt * 32) * ldx + c + 8 * u);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bv[u][ct] = *reinterpret_cast<const f32x4 *>(bp + (long)(ct * 32) * p_pad + c + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            xv[u][rt][m], bv[u][ct][m], acc[rt][ct], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long row = r0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                out[row * KP + ct * 32 + j] = (double)acc[rt][ct][reg];
            }
}

// row-local, float32 MFMA, X staged through wave-private LDS.
// Each wave owns 32 rows and streams them in [32 rows][128 columns] tiles (16 KB): the
// HBM side is 16 fully coalesced loads (two 512-byte row segments per wave-instruction,
// register-prefetched one tile ahead), the MFMA side reads its A fragments from LDS with
// ds_read_b128.  16-byte chunk c of row r is stored at chunk c ^ (r & 15), so the 16
// lanes of a ds_read_b128 group (16 different rows, same logical chunk) hit 16 different
// bank groups.  The tile is private to the wave: no barriers.
template <int NCT>
__global__ __launch_bounds__(256) void k_row_local_f32_lds(const float *__restrict__ X, long ldx,
                                                           const float *__restrict__ B, int p_pad,
                                                           double *__restrict__ out, long n_pad)
{
    constexpr int KP = 32 * NCT;
    __shared__ __attribute__((aligned(16))) float tiles[4][32 * 128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (r0 >= n_pad) return;
    const int h = lane >> 5, j = lane & 31;
    float *my = tiles[wave];

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;

    const float *gbase = X + (r0 + h) * ldx + 4 * j;       // lane: row 2e + h, chunk j
    const float *bp = B + (long)j * p_pad + 4 * h;
    f32x4 stage[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) stage[e] = *reinterpret_cast<const f32x4 *>(gbase + (long)(2 * e) * ldx);

    for (int c0 = 0; c0 < p_pad; c0 += 128) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 2 * e + h;
            *reinterpret_cast<f32x4 *>(my + row * 128 + ((j ^ (row & 15)) << 2)) = stage[e];
        }
        if (c0 + 128 < p_pad) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[e] = *reinterpret_cast<const f32x4 *>(gbase + (long)(2 * e) * ldx + c0 + 128);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMA block
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(my + j * 128 + (((2 * q + h) ^ (j & 15)) << 2));
            f32x4 bv[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bv[ct] = *reinterpret_cast<const f32x4 *>(bp + (long)(ct * 32) * p_pad + c0 + 8 * q);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[ct][m], acc[ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            out[row * KP + ct * 32 + j] = (double)acc[ct][reg];
        }
}

// row-local, float32 MFMA, classic block-tiled form: the 4 waves of a block own 32 rows
// each (X tiles wave-private in LDS) and SHARE the [KP][TC] tile of the small operand B
// through LDS, so B crosses L2->L1 once per block instead of once per wave (in the
// wave-private kernels B traffic equals the X traffic).  Both tiles are register-
// prefetched one step ahead.  TC = tile columns (32/64/128); DB = double-buffered LDS
// (one barrier per tile instead of two).  16-byte chunk c of row r sits at a swizzled
// chunk so that the 16 lanes of a ds_read_b128 group (16 rows, same logical chunk) hit 16
// different bank groups: c ^ (r & 15) for rows >= 256 B, c ^ ((r >> 1) & 7) for 128-B rows.
template <int TC>
__device__ __forceinline__ int rl_phys_chunk(int row, int chunk)
{
    if constexpr (TC >= 64) return chunk ^ (row & 15);
    else return chunk ^ ((row >> 1) & 7);
}

// blockIdx.y: column chunk of `col_chunk` columns (a multiple of TC), partial result to
// out + blockIdx.y * n_pad * KP, summed in a fixed order by k_sum_chunks -- short shards (the
// 12 500 rows of an 8-GPU run are 98 row blocks) otherwise leave most CUs idle.
template <int NCT, int TC, bool DB, bool ACC64 = false>
__global__ __launch_bounds__(256) void k_row_local_f32_blk(const float *__restrict__ X, long ldx,
                                                           const float *__restrict__ B, int p_pad,
                                                           double *__restrict__ out, long n_pad,
                                                           int col_chunk)
{
    constexpr int KP = 32 * NCT;
    constexpr int CPR = TC / 4;                  // 16-byte chunks per tile row
    constexpr int RPI = 64 / CPR;                // tile rows covered by one wave-instruction
    constexpr int NXI = 32 / RPI;                // X load instructions per wave per tile
    constexpr int NBI = KP * CPR / 256;          // B load instructions per thread per tile
    constexpr int NBUF = DB ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float xs[NBUF][4][32 * TC];
    __shared__ __attribute__((aligned(16))) float bs[NBUF][KP * TC];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 32;      // grid = n_pad / 128 exactly
    const int h = lane >> 5, j = lane & 31;

    f32x16 acc[NCT];
    double accd[ACC64 ? NCT : 1][16];            // ACC64: see k_row_local_f32_ws
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
#pragma unroll
    for (int ct = 0; ct < (ACC64 ? NCT : 1); ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) accd[ct][e] = 0.0;

    const int xr = lane / CPR, xc = lane % CPR;
    const float *gx = X + (r0 + xr) * ldx + 4 * xc;
    f32x4 sx[NXI], sb[NBI];
    auto load_tile = [&](int c0) {
#pragma unroll
        for (int e = 0; e < NXI; ++e)
            sx[e] = *reinterpret_cast<const f32x4 *>(gx + (long)(RPI * e) * ldx + c0);
#pragma unroll
        for (int e = 0; e < NBI; ++e) {
            const int cid = t + 256 * e;
            sb[e] = *reinterpret_cast<const f32x4 *>(B + (long)(cid / CPR) * p_pad + c0 + 4 * (cid % CPR));
        }
    };
    auto store_tile = [&](int buf) {
        float *myx = xs[buf][wave];
#pragma unroll
        for (int e = 0; e < NXI; ++e) {
            const int row = RPI * e + xr;
            *reinterpret_cast<f32x4 *>(myx + ((row * CPR + rl_phys_chunk<TC>(row, xc)) << 2)) = sx[e];
        }
#pragma unroll
        for (int e = 0; e < NBI; ++e) {
            const int cid = t + 256 * e, comp = cid / CPR, ch = cid % CPR;
            *reinterpret_cast<f32x4 *>(bs[buf] + ((comp * CPR + rl_phys_chunk<TC>(comp, ch)) << 2)) = sb[e];
        }
    };
    auto compute_tile = [&](int buf) {
        const float *myx = xs[buf][wave];
#pragma unroll
        for (int q = 0; q < TC / 8; ++q) {
            const int pc = rl_phys_chunk<TC>(j, 2 * q + h) << 2;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(myx + j * TC + pc);
            f32x4 bv[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bv[ct] = *reinterpret_cast<const f32x4 *>(bs[buf] + (ct * 32 + j) * TC + pc);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[ct][m], acc[ct], 0, 0, 0);
            if constexpr (ACC64) {
                if ((q & 3) == 3) {                 // 32 columns done: the fp32 chain ends here
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            accd[ct][e] += (double)acc[ct][e];
                            acc[ct][e] = 0.f;
                        }
                }
            }
        }
    };

    const int c_begin = (int)blockIdx.y * col_chunk;
    const int c_end = c_begin + col_chunk < p_pad ? c_begin + col_chunk : p_pad;
    out += (size_t)blockIdx.y * (size_t)n_pad * KP;
    load_tile(c_begin);
    if constexpr (DB) {
        store_tile(0);
        __syncthreads();
        int buf = 0;
        for (int c0 = c_begin; c0 < c_end; c0 += TC) {
            const bool more = c0 + TC < c_end;
            if (more) load_tile(c0 + TC);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(buf);
            if (more) store_tile(buf ^ 1);     // last read two barriers ago
            __syncthreads();
            buf ^= 1;
        }
    } else {
        for (int c0 = c_begin; c0 < c_end; c0 += TC) {
            store_tile(0);
            __syncthreads();
            if (c0 + TC < c_end) load_tile(c0 + TC);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if constexpr (ACC64) out[row * KP + ct * 32 + j] = accd[ct][reg];
            else out[row * KP + ct * 32 + j] = (double)acc[ct][reg];
        }
}

// row-local, float32 MFMA, "wave-streaming" form.  The block-tiled kernel above spends
// its time at block barriers (two per 64..128 columns), so HBM time and MFMA time add up
// instead of overlapping.  Here a block is W waves on one CU; every wave owns one 32-row
// tile for the whole kernel and streams its X through a WAVE-PRIVATE LDS tile (coalesced
// 256-byte row segments in, MFMA fragment layout out; only s_waitcnt, no barrier), one
// tile register-prefetched ahead.  The small operand B is shared by the whole block as
// double-buffered 128-column slabs in LDS: one barrier per 128 columns, B crosses
// L2->LDS once per W*32 rows.  Dynamic LDS = 2 slabs (KP*512 B each) + W * 8 KB.
// ACC64: the fp32 accumulation chain is cut after every 32 columns (16 matrix instructions) and
// the partial sums are added up in float64 (VALU, 32 more registers => 3 waves per SIMD, W <= 12).
// x_r . p_i is a COHERENT sum (a sample against an archetype built from samples like it), so the
// fp32 rounding of a long running sum is systematic: relative error ~ eps32 * L / 400 for chains of
// L columns -- 6.9e-7 unchained (measured, p = 4096), 8.8e-8 at L = 512, 5e-9 at L = 32.  On
// bench.py's parity_converged problem that difference decides whether a float32 run stays on
// the reference's trajectory.
template <int NCT, bool ACC64>
__global__ __launch_bounds__(!ACC64 ? (NCT == 1 ? 1024 : 768) : (NCT == 1 ? 768 : 512)) void k_row_local_f32_ws(const float *__restrict__ X, long ldx,
                                                           const float *__restrict__ B, int p_pad,
                                                           double *__restrict__ out, long n_pad,
                                                           int W, int stagger)
{
    constexpr int KP = 32 * NCT;
    constexpr int SB = 128, TC = 64;
    constexpr int NBI = 2 * NCT;                 // B chunks per thread per slab at >= 512 threads
    extern __shared__ __attribute__((aligned(16))) float ws_smem[];
    const int t = threadIdx.x, lane = t & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // scalar: uniform row base
    float *bs = ws_smem;
    float *xs = ws_smem + 2 * KP * SB + wave * (32 * TC);
    const long r0 = ((long)blockIdx.x * W + wave) * 32;
    const bool active = r0 < n_pad;              // wave-uniform
    // surplus waves of the last block recompute the last tile (no branches in the loop)
    const long r0c = active ? r0 : n_pad - 32;
    const int h = lane >> 5, j = lane & 31;

    f32x16 acc[NCT];
    double accd[ACC64 ? NCT : 1][16];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
#pragma unroll
    for (int ct = 0; ct < (ACC64 ? NCT : 1); ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) accd[ct][e] = 0.0;

    f32x4 sb[NBI], sx[8];
    auto load_b = [&](int c0) {
#pragma unroll
        for (int e = 0; e < NBI; ++e) {
            const int cid = (t + e * nthreads) & (KP * 32 - 1);   // >= 512 threads: wraps only
            sb[e] = *reinterpret_cast<const f32x4 *>(B + (long)(cid >> 5) * p_pad + c0 + 4 * (cid & 31));
        }                                                          // onto identical data
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NBI; ++e) {
            const int cid = (t + e * nthreads) & (KP * 32 - 1);
            const int comp = cid >> 5, ch = cid & 31;
            *reinterpret_cast<f32x4 *>(bs + buf * (KP * SB) + ((comp * 32 + (ch ^ (comp & 15))) << 2)) = sb[e];
        }
    };
    const int xr = lane >> 4, xc = lane & 15;
    const float *gx = X + (r0c + xr) * ldx + 4 * xc;
    auto load_x = [&](int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            sx[e] = *reinterpret_cast<const f32x4 *>(gx + (long)(4 * e) * ldx + c0);
    };
    auto store_x = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = 4 * e + xr;
            *reinterpret_cast<f32x4 *>(xs + ((row * 16 + (xc ^ (row & 15))) << 2)) = sx[e];
        }
    };
    auto compute = [&](int buf, int half) {
        const float *bb = bs + buf * (KP * SB);
        f32x4 a, bv[NCT], an, bn[NCT];
        auto frag = [&](int q, f32x4 &fa, f32x4 (&fb)[NCT]) {
            const int pcx = ((2 * q + h) ^ (j & 15)) << 2;
            const int pcb = ((half * 16 + 2 * q + h) ^ (j & 15)) << 2;
            fa = *reinterpret_cast<const f32x4 *>(xs + j * TC + pcx);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                fb[ct] = *reinterpret_cast<const f32x4 *>(bb + (ct * 32 + j) * SB + pcb);
        };
        if constexpr (!ACC64 || NCT == 1) frag(0, a, bv);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if constexpr (!ACC64 || NCT == 1) {
                if (q + 1 < 8) frag(q + 1, an, bn);     // fragments of the next step in flight
            } else {
                frag(q, a, bv);                         // k > 32 with float64 sums: no second fragment set
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[ct][m], acc[ct], 0, 0, 0);
            if constexpr (!ACC64 || NCT == 1) {
                a = an;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) bv[ct] = bn[ct];
            }
            if constexpr (ACC64) {
                if ((q & 3) == 3) {                 // 32 columns done: the chain ends here
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            accd[ct][e] += (double)acc[ct][e];
                            acc[ct][e] = 0.f;
                        }
                }
            }
        }
    };

    // blocks start at different column slabs (cyclic order) so that the chip does not
    // sweep one narrow column band of X -- one set of HBM channels -- at a time
    const int nslab = p_pad / SB;
    const int s_first = (int)(((long)blockIdx.x * stagger) % nslab);
    load_b(s_first * SB);
    load_x(s_first * SB);
    store_b(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        int scur = s_first + s;
        if (scur >= nslab) scur -= nslab;
        const int c0 = scur * SB;
        int snext = s + 1 < nslab ? scur + 1 : scur;       // last slab: harmless reload
        if (snext >= nslab) snext -= nslab;
        store_x();                                // wave-private: LDS is in order per wave
        load_x(c0 + TC);
        load_b(snext * SB);
        __builtin_amdgcn_sched_barrier(0);
        compute(s & 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        store_x();
        store_b((s + 1) & 1);                     // that buffer was last read before the previous barrier
        load_x(snext * SB);
        __builtin_amdgcn_sched_barrier(0);
        compute(s & 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if constexpr (ACC64) out[row * KP + ct * 32 + j] = accd[ct][reg];
                else out[row * KP + ct * 32 + j] = (double)acc[ct][reg];
            }
    }
}

// row-local, float32 MFMA, wave-streaming with LDS-DMA staging (k <= 32).  The structure of
// k_row_local_f32_ws -- a block is W waves on one CU, every wave owns one 32-row tile and sweeps
// the columns -- with both operands brought in by `global_load_lds_dwordx4`: no staging registers
// (32 fewer than the register-staged kernel, which is what makes room for the float64 sums at four
// waves per SIMD), no ds_write instructions.  A wave-instruction writes 64 x 16 B = 1 KiB of LDS
// in lane order (a "piece"), so the XOR swizzle that keeps the fragment reads conflict free sits
// on the SOURCE address: lane l of a piece fetches the chunk whose swizzled position is l.
//   X: a tile is 32 rows x 32 columns = 4 pieces (8 rows x 128 B each); every wave owns a RING of
//      R pieces (R = 9..12, whatever 160 KB of LDS allow for W waves): tile t's pieces sit in ring
//      slots (4t + i) mod R, and R - 4 pieces of the tiles behind it are in flight while it is
//      multiplied (the register-staged kernel keeps 8 KB per wave in flight).
//   B: 64-column slabs (8 pieces: 4 components x 256 B each), two buffers shared by the block;
//      every wave issues one piece per slab (waves past the eighth repeat earlier pieces:
//      identical bytes).
// Every wave issues the same number of DMA instructions per step, so the waits are counted
// (`s_waitcnt vmcnt(N)`: hipcc does not track what lands in LDS; LDS-DMA data is ordered for a
// ds_read only by the issuing wave's covering vmcnt, plus a barrier for other waves' reads) and the
// block barrier is the raw s_barrier (a __syncthreads would drain the DMAs in flight).  The fp32
// accumulation chain ends after every tile (32 columns); the pieces are summed in float64 (see
// k_row_local_f32_ws).
template <bool NT>
__device__ __forceinline__ void dma16(const float *src, float *lds_uniform)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_uniform, 16, 0, NT ? 2 : 0);
}

template <int R, bool NT, bool EARLY = false>
__global__ __launch_bounds__(1024) void k_row_local_f32_dma(const float *__restrict__ X, long ldx,
                                                            const float *__restrict__ B, int p_pad,
                                                            double *__restrict__ out, long n_pad, int W, int prio)
{
    constexpr int KP = 32, SB = 64, TC = 32;
    static_assert(R >= 8 && R <= 12, "ring of 8..12 pieces");
    extern __shared__ __attribute__((aligned(16))) float dma_smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float *bs = dma_smem;                                   // [2][KP * SB]
    float *xs = dma_smem + 2 * KP * SB + wave * (R * 256);  // R pieces of 256 floats, wave-private
    // prio bit 1 (row_local_reverse): the row tiles from the last to the first -- the pass before this one
    // (reduce-over-rows) walks the rows upwards and leaves the LAST rows in the memory-side cache, the pass
    // after it starts at the first rows again
    const long blk = (prio & 2) ? (long)(gridDim.x - 1 - blockIdx.x) : (long)blockIdx.x;
    const long r0 = (blk * W + wave) * 32;
    const bool active = r0 < n_pad;
    const long r0c = active ? r0 : n_pad - 32;
    const int h = lane >> 5, j = lane & 31;

    // X piece i of a tile (rows 8i .. 8i+7): lane -> row 8i + (lane >> 3), swizzled position lane & 7
    int xoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        xoff[i] = (int)(row * ldx) + 4 * chunk;
    }
    const float *gx = X + r0c * ldx;
    // B piece of this wave: components 4 pb .. 4 pb + 3 of the slab, lane -> component 4 pb + (lane >> 4)
    const int pb = wave & 7;
    const int bcomp = 4 * pb + (lane >> 4);
    const int boff = bcomp * p_pad + 4 * ((lane & 15) ^ (bcomp & 15));
    const int ntile = p_pad / TC, nslab = p_pad / SB;
    int issued = 0;                                          // X pieces issued so far (uniform)
    auto issue_piece = [&]() {                               // next piece of the sweep into its ring slot
        int tl = issued >> 2;
        if (tl >= ntile) tl = ntile - 1;                     // past the end: harmless reloads
        const int i = issued & 3;
        const int slot = issued % R;
        const int off = i == 0 ? xoff[0] : (i == 1 ? xoff[1] : (i == 2 ? xoff[2] : xoff[3]));
        dma16<NT>(gx + off + tl * TC, xs + slot * 256);
        ++issued;
    };
    auto issue_b = [&](int sl, int buf) { dma16<false>(B + boff + sl * SB, bs + buf * (KP * SB) + pb * 256); };

    // two fp32 accumulator sets, alternating tile by tile: the float64 flush of one set runs in the
    // shadow of the matrix instructions that fill the other (the flush needs the LAST instruction of
    // its chain to have retired: 16 passes)
    f32x16 acc0, acc1;
    double accd[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc0[e] = 0.f;
        acc1[e] = 0.f;
        accd[e] = 0.0;
    }
    auto tile_mfma = [&](int tl, f32x16 &acc) {
        const int s = tl >> 1, ts = tl & 1;
        if (ts == 0) {
            // pieces 4 tl .. 4 tl + 3 and (older) this wave's B(s) piece have landed when at most the
            // R - 4 younger X pieces are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 4) : "memory");
            __builtin_amdgcn_s_barrier();                    // all waves: B(s) complete, slab s-1 read
            issue_b(s + 1 < nslab ? s + 1 : s, (s + 1) & 1);
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 3) : "memory");   // + the B piece issued one tile ago
        }
        const int base = (4 * tl) % R;                       // ring slot of this tile's first piece
        int slot = base + (j >> 3);
        slot = slot >= R ? slot - R : slot;
        const float *xb = xs + slot * 256 + (j & 7) * TC;
        const float *bb = bs + (s & 1) * (KP * SB) + j * SB;
        if constexpr (EARLY) {
            // all fragments of the tile into registers first (32 VGPRs), the LDS reads waited for, and
            // the tile's four ring slots handed back to the DMA BEFORE the 16 matrix instructions run:
            // 8 KB per wave in flight during the arithmetic instead of 4 (experiment, row_local_early)
            f32x4 av[4], bvv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pcx = ((2 * q + h) ^ ((j >> 1) & 7)) << 2;
                const int pcb = ((ts * 8 + 2 * q + h) ^ (j & 15)) << 2;
                av[q] = *reinterpret_cast<const f32x4 *>(xb + pcx);
                bvv[q] = *reinterpret_cast<const f32x4 *>(bb + pcb);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece();
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][m], bvv[q][m], acc, 0, 0, 0);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pcx = ((2 * q + h) ^ ((j >> 1) & 7)) << 2;
            const int pcb = ((ts * 8 + 2 * q + h) ^ (j & 15)) << 2;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(xb + pcx);
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bb + pcb);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[m], acc, 0, 0, 0);
        }
        // the four slots of this tile are free again (their reads were waited for before the MFMAs)
        // prio: the wave that is about to put 4 KB in flight goes ahead of the waves that have
        // matrix instructions and float64 sums to issue (experiment, aa_set_option row_local_prio)
        if (prio & 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece();
        if (prio & 1) __builtin_amdgcn_s_setprio(0);
    };
    auto flush = [&](f32x16 &acc) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            accd[e] += (double)acc[e];
            acc[e] = 0.f;
        }
    };

    issue_b(0, 0);
#pragma unroll
    for (int i = 0; i < R; ++i) issue_piece();
    for (int tl = 0; tl < ntile; tl += 2) {                  // p_pad is a multiple of 128: ntile is even
        tile_mfma(tl, acc0);
        flush(acc1);                                         // the previous tile's sums (zeros at tl = 0)
        tile_mfma(tl + 1, acc1);
        flush(acc0);
    }
    flush(acc1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the trailing reloads
    if (active) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            out[row * KP + j] = accd[reg];
        }
    }
}

// row-local, float64 VALU: block = 64 rows, X tile staged through LDS with coalesced
// loads; thread (row = t&63, q = t>>6) accumulates the components [q*KP/4, (q+1)*KP/4).
template <int KP>
__global__ __launch_bounds__(256) void k_row_local_f64(const double *__restrict__ X, long ldx,
                                                       const double *__restrict__ B, int p_pad,
                                                       double *__restrict__ out, long n_pad)
{
    constexpr int KQ = KP / 4;
    __shared__ double tile[64][65];
    const int t = threadIdx.x, row = t & 63, q = t >> 6;
    const long r0 = (long)blockIdx.x * 64;
    if (r0 >= n_pad) return;
    double acc[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) acc[i] = 0.0;
    for (int c0 = 0; c0 < p_pad; c0 += 64) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int idx = e * 256 + t;
            tile[idx >> 6][idx & 63] = X[(r0 + (idx >> 6)) * ldx + c0 + (idx & 63)];
        }
        __syncthreads();
        const double *bq = B + (long)(q * KQ) * p_pad + c0;
        for (int cc = 0; cc < 64; ++cc) {
            const double x = tile[row][cc];
#pragma unroll
            for (int i = 0; i < KQ; ++i) acc[i] = fma(x, bq[(long)i * p_pad + cc], acc[i]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < KQ; ++i) out[(r0 + row) * KP + q * KQ + i] = acc[i];
}

// rank-m correction of a reduce-over-rows result whose tall operand changed in m rows after
// the pass had read it (the QP's stragglers finish while Z'X is being accumulated):
//   extra[g][i][c] = sum over the slots s of slab g of (znew[s][i] - Z[rows[s]][i]) * X[rows[s]][c]
// written as G = gridDim.y further slabs of the split-row partials.  The slot count lives on
// the device; slab g takes the contiguous slot range [g*chunk, (g+1)*chunk).  Same MFMA
// mapping as k_reduce_rows_f32 (a wave owns a 128-column strip), with the rows gathered: a
// wave reads 64 row indices at a time (one coalesced load) and hands them out with
// ds_bpermute, so a step of 2*U slots costs one memory latency (X rows and z rows together).
__global__ __launch_bounds__(256) void k_reduce_rows_fixup_f32(const float *__restrict__ X, long ldx,
                                                               const unsigned int *__restrict__ count_dev,
                                                               const int *__restrict__ rows,
                                                               const double *__restrict__ zslot,
                                                               const double *__restrict__ Ztall,
                                                               int p_pad, float *__restrict__ extra)
{
    constexpr int KP = 32, U = 8;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = (blockIdx.x * 4 + wave) * 128;
    if (c0 >= p_pad) return;   // wave-uniform; the kernel has no barriers
    const int h = lane >> 5, j = lane & 31;
    const long count = (long)*count_dev, G = gridDim.y;
    long chunk = (count + G - 1) / G;
    chunk = (chunk + 2 * U - 1) / (2 * U) * (2 * U);
    const long s_begin = (long)blockIdx.y * chunk;
    long s_end = s_begin + chunk;
    if (s_end > count) s_end = count;

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    for (long b = s_begin; b < s_end; b += 64) {
        const int mine = (b + lane < s_end) ? rows[b + lane] : 0;
        for (int q = 0; q < 64 && b + q < s_end; q += 2 * U) {
            f32x4 xv[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sl = q + 2 * u + h;                 // slot of this lane's half
                const long slot = b + sl;
                const bool valid = slot < s_end;
                const long row = __shfl(mine, sl & 63, 64);   // 0 for slots past the end
                xv[u] = *reinterpret_cast<const f32x4 *>(X + row * ldx + c0 + 4 * j);
                av[u] = valid ? (float)(zslot[slot * KP + j] - Ztall[row * KP + j]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], xv[u][m], acc[m], 0, 0, 0);
        }
    }
    float *out = extra + (size_t)blockIdx.y * KP * p_pad;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int comp = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        f32x4 v = {acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]};
        *reinterpret_cast<f32x4 *>(out + (size_t)comp * p_pad + c0 + 4 * j) = v;
    }
}

// measurement only: stream X once with 16-byte loads, no arithmetic to speak of -- the
// read bandwidth the memory system delivers to a kernel of this shape (aa_time_kernel 2)
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream_probe(const f32x4 *__restrict__ x, long n16,
                                                      float *__restrict__ sink)
{
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += x[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.f;   // keeps the loads alive
}

// the load pattern of k_row_local_f32_ws without its LDS / MFMA work: a wave owns 32 rows and
// walks the columns in 64-column tiles (TILED = false: 32 pieces of 256 B, 16 KB apart, per
// tile, as in the row-major matrix; TILED = true: the same bytes as one contiguous 8 KB chunk,
// i.e. what a [32 x 64]-tiled storage of X would give)
template <bool TILED>
__global__ __launch_bounds__(832) void k_stream_probe_rows(const float *__restrict__ X, long ldx,
                                                           long n_pad, int p_pad,
                                                           float *__restrict__ sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 13 + wave) * 32;
    if (r0 >= n_pad) return;
    const int xr = lane >> 4, xc = lane & 15;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < p_pad; c0 += 128) {
        f32x4 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tile = e >> 3, ee = e & 7;
            const long off = TILED ? (r0 * ldx + (long)(c0 / 64 + tile) * 2048 + (4 * ee + xr) * 64 + 4 * xc)
                                   : ((r0 + 4 * ee + xr) * ldx + c0 + 64 * tile + 4 * xc);
            v[e] = *reinterpret_cast<const f32x4 *>(X + off);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += v[e];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.f;
}

// variant: 0 = 4 loads in flight per thread x 4096 blocks; 1 = 8 x 2048; 2 = 16 x 1024; 3 = 8 x 8192
int launch_stream_probe(Ctx *c, int variant)
{
    const long bytes = (long)c->n_pad * c->p_pad * (c->dtype == AA_F32 ? 4 : 8);
    const f32x4 *x = reinterpret_cast<const f32x4 *>(c->X.p);
    float *sink = c->partial.as<float>();
    const long tiles = c->n_pad / 32;
    switch (variant) {
        case 4: hipLaunchKernelGGL(k_stream_probe_rows<false>, dim3((unsigned)((tiles + 12) / 13)), dim3(832), 0, c->stream, c->X.as<float>(), c->p_pad, c->n_pad, (int)c->p_pad, sink); break;
        case 5: hipLaunchKernelGGL(k_stream_probe_rows<true>, dim3((unsigned)((tiles + 12) / 13)), dim3(832), 0, c->stream, c->X.as<float>(), c->p_pad, c->n_pad, (int)c->p_pad, sink); break;
        case 1: hipLaunchKernelGGL(k_stream_probe<8>, dim3(2048), dim3(256), 0, c->stream, x, bytes / 16, sink); break;
        case 2: hipLaunchKernelGGL(k_stream_probe<16>, dim3(1024), dim3(256), 0, c->stream, x, bytes / 16, sink); break;
        case 3: hipLaunchKernelGGL(k_stream_probe<8>, dim3(8192), dim3(256), 0, c->stream, x, bytes / 16, sink); break;
        default: hipLaunchKernelGGL(k_stream_probe<4>, dim3(4096), dim3(256), 0, c->stream, x, bytes / 16, sink); break;
    }
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
#define PASS_NAME(WHICH, ...) snprintf(c->pass_names[WHICH], sizeof(c->pass_names[WHICH]), __VA_ARGS__)

static void gemm_event(Ctx *c, int which)
{
    if (!c->time_gemm) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess) {
        (void)hipEventRecord(e, c->stream);
        c->gemmEvents[which].push_back(e);
    }
}

int launch_reduce_rows(Ctx *c, const double *A_tall, double *out_wide, void *outT, bool main_only)
{
    dim3 block(256);
    gemm_event(c, 0);
    if (c->dtype == AA_F32) {
        dim3 grid((unsigned)((c->p_pad + 511) / 512), (unsigned)c->nslab);
        float *part = c->partial.as<float>();
#define RRL(NCTV, UV)                                                                         \
    hipLaunchKernelGGL((k_reduce_rows_f32<NCTV, UV>), grid, block, 0, c->stream, c->X.as<float>(), \
                       c->p_pad, A_tall, c->rows_per_slab, c->n_pad, (int)c->p_pad, part)
        if (c->KP == 32) {
            if (g_reduce_rows_unroll == 8) RRL(1, 8);
            else RRL(1, 4);
        } else {
            RRL(2, 4);
        }
        PASS_NAME(0, "k_reduce_rows_f32<%d,%d>", c->KP / 32, (c->KP == 32 && g_reduce_rows_unroll == 8) ? 8 : 4);
#undef RRL
    } else {
        dim3 grid((unsigned)((c->p_pad + 255) / 256), (unsigned)c->nslab);
        double *part = c->partial.as<double>();
        if (g_f64_mfma) PASS_NAME(0, "k_reduce_rows_f64_mfma<%d>", c->KP == 32 ? 2 : 4);
        else PASS_NAME(0, "k_reduce_rows_f64<%d>", c->KP);
        if (g_f64_mfma) {
            if (c->KP == 32)
                hipLaunchKernelGGL(k_reduce_rows_f64_mfma<2>, grid, block, 0, c->stream, c->X.as<double>(),
                                   c->p_pad, A_tall, c->rows_per_slab, c->n_pad, (int)c->p_pad, part);
            else
                hipLaunchKernelGGL(k_reduce_rows_f64_mfma<4>, grid, block, 0, c->stream, c->X.as<double>(),
                                   c->p_pad, A_tall, c->rows_per_slab, c->n_pad, (int)c->p_pad, part);
        } else if (c->KP == 32)
            hipLaunchKernelGGL(k_reduce_rows_f64<32>, grid, block, 0, c->stream, c->X.as<double>(),
                               c->p_pad, A_tall, c->rows_per_slab, c->n_pad, (int)c->p_pad, part);
        else
            hipLaunchKernelGGL(k_reduce_rows_f64<64>, grid, block, 0, c->stream, c->X.as<double>(),
                               c->p_pad, A_tall, c->rows_per_slab, c->n_pad, (int)c->p_pad, part);
    }
    AA_CHECK_HIP(hipGetLastError());
    gemm_event(c, 0);
    if (main_only) return AA_OK;
    return launch_reduce_rows_finish(c, out_wide, outT, 0);
}

int launch_reduce_rows_finish(Ctx *c, double *out_wide, void *outT, int extra_slabs)
{
    const long elems = (long)c->KP * c->p_pad;
    const long slabs = c->nslab + extra_slabs;
    dim3 block(256), grid((unsigned)((elems + 255) / 256));
    if (c->dtype == AA_F32) {
        float *oT = ((c->world > 1 || c->force_comm)) ? nullptr : reinterpret_cast<float *>(outT);
        hipLaunchKernelGGL((k_reduce_partials<float, float>), grid, block, 0, c->stream,
                           (const float *)c->partial.as<float>(), slabs, elems, out_wide, oT);
    } else {
        double *oT = ((c->world > 1 || c->force_comm) || outT == (void *)out_wide) ? nullptr
                                                                : reinterpret_cast<double *>(outT);
        hipLaunchKernelGGL((k_reduce_partials<double, double>), grid, block, 0, c->stream,
                           (const double *)c->partial.as<double>(), slabs, elems, out_wide, oT);
    }
    AA_CHECK_HIP(hipGetLastError());
    if ((c->world > 1 || c->force_comm)) {
        // riders in the buffer's tail (Ctx::ride*): one all-reduce for the lot
        double *tail = out_wide + elems;
        long extra = 0;
        if (c->ride_dst == tail) extra = c->ride_count;
        if (c->ride.on) {
            AA_REQUIRE(c->ride.gather >= tail && c->ride.gather + c->ride.count <= tail + AA_WIDE_TAIL(c->KP), AA_ERR_STATE,
                       "a reduction was left waiting for an all-reduce that did not come");
            const long end = (long)(c->ride.gather - tail) + c->ride.count;
            if (end > extra) extra = end;
        }
        AA_CHECK(comm_allreduce(c, out_wide, elems + extra, 0));
        if (c->ride.on) AA_CHECK(launch_ride_post(c, &c->ride, c->ride.gather));
        c->ride_dst = nullptr;
        c->ride_count = 0;
        if (outT && outT != (void *)out_wide) AA_CHECK(launch_wide_to_T(c, out_wide, outT));
    }
    return AA_OK;
}

int launch_reduce_rows_fixup(Ctx *c, const unsigned int *count_dev, const int *rows_dev,
                             const double *zslot, const double *Ztall)
{
    AA_REQUIRE(c->KP == 32 && c->dtype == AA_F32, AA_ERR_STATE, "tail fix-up: float32 data, k <= 32");
    const size_t slab_elems = (size_t)c->KP * c->p_pad;
    dim3 grid((unsigned)((c->p_pad + 511) / 512), QP_FIX_SLABS), block(256);
    hipLaunchKernelGGL(k_reduce_rows_fixup_f32, grid, block, 0, c->stream, c->X.as<float>(), c->p_pad,
                       count_dev, rows_dev, zslot, Ztall, (int)c->p_pad,
                       c->partial.as<float>() + (size_t)c->nslab * slab_elems);
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// ---------------------------------------------------------------------------
// Implicit RBF kernel (archetypal_analysis.py:673-910 with K never formed):
//   out[r][i] = sum_c exp(-gamma ||x_r - x_c||^2) V[c][i]        (V, out tall [n_pad][KP], float64)
// Both pass shapes of the kernel form reduce to this product (K is symmetric: C K = (K C')').  A block
// owns 64 rows and walks over 64-column tiles of K: S = X_R X_C' through LDS in chunks of 16 features
// (a thread holds a 4 x 4 patch), E = exp(-gamma max(|x_r|^2 + |x_c|^2 - 2 S, 0)) goes to LDS, then
// out_R += E V_C (a thread holds KP / 4 components of one row).  float64 VALU throughout: the
// per-sample QPs behind it need a float64 gradient.  n^2 (2 p + 2 KP) flop per
// product: for n up to a few ten thousand; grid.y splits the columns so that the chip is full.
// ---------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_rbf_kv(const double *__restrict__ F, long ldf, int pf,
                                                const double *__restrict__ nrm, double gamma,
                                                const double *__restrict__ V, long n, long n_pad,
                                                long tiles_per_split, double *__restrict__ out)
{
    constexpr int PC = 16, CPT = KP / 4;
    extern __shared__ __attribute__((aligned(16))) double rbf_smem[];     // 67 KB (KP = 32) / 83 KB (KP = 64)
    double (*es)[65] = reinterpret_cast<double (*)[65]>(rbf_smem);
    double (*vs)[KP] = reinterpret_cast<double (*)[KP]>(rbf_smem + 64 * 65);
    double (*xr)[PC + 1] = reinterpret_cast<double (*)[PC + 1]>(rbf_smem + 64 * 65 + 64 * KP);
    double (*xc)[PC + 1] = reinterpret_cast<double (*)[PC + 1]>(rbf_smem + 64 * 65 + 64 * KP + 64 * (PC + 1));
    const int t = threadIdx.x;
    const long r0 = (long)blockIdx.x * 64;
    const int tr = t >> 4, tc = t & 15;                  // 4 x 4 patch of S
    const int orow = t >> 2, oc0 = (t & 3) * CPT;        // one row, CPT components of the output
    double acc[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) acc[i] = 0.0;
    const long ntile = n_pad / 64;
    long ct0 = (long)blockIdx.y * tiles_per_split, ct1 = ct0 + tiles_per_split;
    if (ct1 > ntile) ct1 = ntile;
    for (long ct = ct0; ct < ct1; ++ct) {
        const long c0 = ct * 64;
        if (c0 >= n) break;                              // columns past n carry zero rows of V
        double sv[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) sv[a][b] = 0.0;
        for (int q0 = 0; q0 < pf; q0 += PC) {
            __syncthreads();
            for (int e = t; e < 64 * PC; e += 256) {
                const int rr = e / PC, qq = e % PC;
                const bool ok = q0 + qq < pf;
                xr[rr][qq] = ok ? F[(r0 + rr) * ldf + q0 + qq] : 0.0;      // rows < n_pad exist (zero padded)
                xc[rr][qq] = ok ? F[(c0 + rr) * ldf + q0 + qq] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < PC; ++qq) {
                double ra[4], cb[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) ra[a] = xr[4 * tr + a][qq];
#pragma unroll
                for (int b = 0; b < 4; ++b) cb[b] = xc[4 * tc + b][qq];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) sv[a][b] = fma(ra[a], cb[b], sv[a][b]);
            }
        }
        __syncthreads();                                 // the previous tile's E and V have been consumed
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long r = r0 + 4 * tr + a, cc = c0 + 4 * tc + b;
                const double d2 = fmax(nrm[r] + nrm[cc] - 2.0 * sv[a][b], 0.0);
                es[4 * tr + a][4 * tc + b] = (r == cc) ? 1.0 : exp(-gamma * d2);
            }
        for (int e = t; e < 64 * KP; e += 256) vs[e / KP][e % KP] = V[(c0 + e / KP) * KP + e % KP];
        __syncthreads();
        for (int cc = 0; cc < 64; ++cc) {
            const double ev = es[orow][cc];
#pragma unroll
            for (int i = 0; i < CPT; ++i) acc[i] = fma(ev, vs[cc][oc0 + i], acc[i]);
        }
    }
    double *dst = out + (size_t)blockIdx.y * n_pad * KP;
    const long r = r0 + orow;
#pragma unroll
    for (int i = 0; i < CPT; ++i) dst[r * KP + oc0 + i] = r < n ? acc[i] : 0.0;     // padding rows stay zero
}

__global__ __launch_bounds__(256) void k_rbf_norms(const double *__restrict__ F, long ldf, int pf, long n_pad,
                                                   double *__restrict__ nrm)
{
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_pad) return;
    double s = 0.0;
    for (int q = 0; q < pf; ++q) s = fma(F[r * ldf + q], F[r * ldf + q], s);
    nrm[r] = s;
}

int launch_rbf_norms(Ctx *c)
{
    hipLaunchKernelGGL(k_rbf_norms, dim3((unsigned)((c->n_pad + 255) / 256)), dim3(256), 0, c->stream,
                       (const double *)c->feat.as<double>(), c->feat_ld, (int)c->feat_p, c->n_pad, c->featNorm.as<double>());
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

int launch_implicit_kv(Ctx *c, const double *V_tall, double *out_tall)
{
    AA_REQUIRE(c->implicit_kernel == 1 && c->feat.p, AA_ERR_STATE, "no implicit kernel set");
    const long rtiles = c->n_pad / 64, ntile = c->n_pad / 64;
    long nsplit = (1024 + rtiles - 1) / rtiles;          // ~4 blocks per CU
    if (nsplit > ntile) nsplit = ntile;
    if (nsplit < 1) nsplit = 1;
    const long tps = (ntile + nsplit - 1) / nsplit;
    nsplit = (ntile + tps - 1) / tps;
    double *dst = out_tall;
    if (nsplit > 1) {
        AA_CHECK(c->rlPartial.alloc((size_t)nsplit * c->n_pad * c->KP * sizeof(double)));
        dst = c->rlPartial.as<double>();
    }
    const dim3 grid((unsigned)rtiles, (unsigned)nsplit);
    const size_t lds = (size_t)(64 * 65 + 64 * c->KP + 2 * 64 * 17) * sizeof(double);
    static bool attr_rbf[64] = {false};
    if (!attr_rbf[c->device & 63]) {
        AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rbf_kv<32>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rbf_kv<64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_rbf[c->device & 63] = true;
    }
    if (c->KP == 32)
        hipLaunchKernelGGL(k_rbf_kv<32>, grid, dim3(256), lds, c->stream, (const double *)c->feat.as<double>(), c->feat_ld,
                           (int)c->feat_p, (const double *)c->featNorm.as<double>(), c->rbf_gamma, V_tall, c->n, c->n_pad,
                           tps, dst);
    else
        hipLaunchKernelGGL(k_rbf_kv<64>, grid, dim3(256), lds, c->stream, (const double *)c->feat.as<double>(), c->feat_ld,
                           (int)c->feat_p, (const double *)c->featNorm.as<double>(), c->rbf_gamma, V_tall, c->n, c->n_pad,
                           tps, dst);
    if (nsplit > 1) {
        const long elems = c->n_pad * c->KP;
        hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream,
                           (const double *)dst, elems, (int)nsplit, out_tall);
    }
    AA_CHECK_HIP(hipGetLastError());
    return AA_OK;
}

// 0: operands straight from global memory; 1: X staged in wave-private LDS;
// >= 2: block-tiled, B shared through LDS: 2 = 64-column tiles, 3 = 64 double-buffered,
// 4 (default) = 128, 5 = 32 double-buffered, 6 = 128 double-buffered, 7 = 32;
// 8: wave-streaming (wave-private X tiles, B slabs shared per block); 9: the same with LDS-DMA staging
// and float64 sums of 32-column pieces (k <= 32).  aa_set_option.
int g_row_local_variant = -1;   // -1: by size (wave-streaming from 32768 rows per GPU -- 9 for k <= 32, 8 above -- else 4)
int g_f64_mfma = 1;            // float64 data: pass kernels on the f64 matrix cores (0: f64 VALU;
                               // row-local: 1 = wave-streaming from 32768 rows, else block-tiled;
                               // 2 / 3 = always wave-streaming / always block-tiled)
int g_reduce_rows_unroll = 4;  // row pairs per software-pipeline half step (4 or 8; k <= 32)
int g_reduce_rows_blocks = 512; // target block count of the reduce-over-rows kernel
int g_row_local_stagger = 0;   // variant 8: column-slab offset between consecutive blocks
int g_row_local_waves = 0;     // variant 8: waves per block (0 = one block per CU)
int g_row_local_acc64 = 1;     // float32 row-local kernels: cut the fp32 accumulation chain every 32 columns and sum the
                               // pieces in float64 -- 0: never (row_local_variant 8 for large shards), 1: in the
                               // block-tiled kernel and, for large shards with k <= 32, through the LDS-DMA kernel
                               // (variant 9), 2: in the register-staged wave-streaming kernel too (k > 32; costs
                               // it a wave per SIMD: 0.31 -> 0.6 ms)
int g_row_local_ring = 8;      // LDS-DMA kernel: pieces in a wave's ring (8: one tile in flight per wave -- measured
                               // fastest, 0.376 ms back to back against 0.388 at 11; 0: what LDS allows)
int g_row_local_nt = 0;        // LDS-DMA kernel: non-temporal hint on the X stream
int g_row_local_early = 0;     // LDS-DMA kernel: fragments to registers first, the next pieces issued before the MFMAs
int g_row_local_reverse = 0;   // LDS-DMA kernel, experiment: row tiles from the last to the first, so that the rows the previous pass read last come first (memory-side cache reuse between consecutive passes): no effect, 499 / 488 / 493 against 494 / 496 it/s, pass times unchanged
int g_row_local_prio = 0;      // LDS-DMA kernel: raised wave priority while a wave issues its DMA pieces
int g_row_local_chunk = 0;     // experiment: force the column chunk of the block-tiled float32 kernel (0: by size)
int g_row_local_split = 1;     // block-tiled kernels: split the contraction over column chunks when there are few row blocks
static int row_local_variant(const Ctx *c)
{
    if (g_row_local_variant >= 0) return g_row_local_variant;
    if (c->n_pad / 32 < 8 * 128) return 4;
    // large shards: wave-streaming; k <= 32 with float64 sums of 32-column pieces through the LDS-DMA
    // kernel (1 % slower than the register-staged kernel with one fp32 chain per row, 86 x closer to
    // the exact product: rms error 3.5e-9 against 3.1e-7 of sum |x||b| at p = 4096)
    return (c->KP == 32 && g_row_local_acc64 >= 1) ? 9 : 8;
}

int launch_row_local(Ctx *c, const void *B_wideT, double *out_tall)
{
    dim3 block(256);
    gemm_event(c, 1);
    if (c->dtype == AA_F32 && row_local_variant(c) == 9 && c->KP == 32) {
        // LDS-DMA wave-streaming kernel with float64 sums
        const float *B = reinterpret_cast<const float *>(B_wideT);
        const long tiles = c->n_pad / 32;
        int W = g_row_local_waves > 0 ? g_row_local_waves : (int)((tiles + 255) / 256);
        if (W < 8) W = 8;
        if (W > 16) W = 16;
        int R = (160 * 1024 - 2 * 32 * 64 * 4) / (W * 1024);      // ring pieces per wave that 160 KB allow
        if (R > 12) R = 12;
        if (g_row_local_ring > 0 && g_row_local_ring < R) R = g_row_local_ring;
        if (R < 8) R = 8;
        const size_t lds = ((size_t)2 * 32 * 64 + (size_t)W * R * 256) * sizeof(float);
        dim3 grid((unsigned)((tiles + W - 1) / W)), blk((unsigned)(64 * W));
#define RLDE()                                                                                     \
    do {                                                                                           \
        static bool attr_done_e[64] = {false};                                                     \
        if (!attr_done_e[c->device & 63]) {                                                        \
            AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_row_local_f32_dma<8, false, true>), \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
            attr_done_e[c->device & 63] = true;                                                    \
        }                                                                                          \
        hipLaunchKernelGGL((k_row_local_f32_dma<8, false, true>), grid, blk, lds, c->stream, c->X.as<float>(), \
                           c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad, W, (g_row_local_prio ? 1 : 0) | (g_row_local_reverse ? 2 : 0));   \
    } while (0)
#define RLD(RV, NTV)                                                                               \
    do {                                                                                           \
        static bool attr_done[64] = {false};                                                       \
        if (!attr_done[c->device & 63]) {                                                          \
            AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_row_local_f32_dma<RV, NTV>), \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
            attr_done[c->device & 63] = true;                                                      \
        }                                                                                          \
        hipLaunchKernelGGL((k_row_local_f32_dma<RV, NTV>), grid, blk, lds, c->stream, c->X.as<float>(),   \
                           c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad, W, (g_row_local_prio ? 1 : 0) | (g_row_local_reverse ? 2 : 0));   \
    } while (0)
#define RLD2(RV) do { if (g_row_local_early && RV == 8) RLDE(); else if (g_row_local_nt) RLD(RV, true); else RLD(RV, false); } while (0)
        PASS_NAME(1, "k_row_local_f32_dma<%d>", R);
        switch (R) {
            case 8: RLD2(8); break;
            case 9: RLD2(9); break;
            case 10: RLD2(10); break;
            case 11: RLD2(11); break;
            default: RLD2(12); break;
        }
#undef RLD2
#undef RLD
#undef RLDE
    } else if (c->dtype == AA_F32 && row_local_variant(c) == 8) {
        // wave-streaming kernel: W waves per block, one block per CU where possible
        const float *B = reinterpret_cast<const float *>(B_wideT);
        const int nct = c->KP / 32;
        const long tiles = c->n_pad / 32;
        const bool a64 = g_row_local_acc64 >= 2;
        const int wmax = !a64 ? (nct == 1 ? 16 : 12) : (nct == 1 ? 12 : 8);   // 160 KB of LDS; float64 sums: 3 (2) waves per SIMD
        int W = g_row_local_waves > 0 ? g_row_local_waves : (int)((tiles + 255) / 256);
        if (W < 8) W = 8;                                 // the B slab loader assumes >= 512 threads
        if (W > wmax) W = wmax;
        const size_t lds = ((size_t)2 * c->KP * 128 + (size_t)W * 32 * 64) * sizeof(float);
        static bool attr_set[64] = {false};                 // the attribute is per device
        if (!attr_set[c->device & 63]) {
            const void *fns[4] = {reinterpret_cast<const void *>(&k_row_local_f32_ws<1, false>),
                                  reinterpret_cast<const void *>(&k_row_local_f32_ws<2, false>),
                                  reinterpret_cast<const void *>(&k_row_local_f32_ws<1, true>),
                                  reinterpret_cast<const void *>(&k_row_local_f32_ws<2, true>)};
            for (const void *fn : fns)
                AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[c->device & 63] = true;
        }
        dim3 grid((unsigned)((tiles + W - 1) / W)), blk((unsigned)(64 * W));
#define RLW(NCTV, A64V)                                                                              \
    hipLaunchKernelGGL((k_row_local_f32_ws<NCTV, A64V>), grid, blk, lds, c->stream, c->X.as<float>(), \
                       c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad, W, g_row_local_stagger)
        PASS_NAME(1, "k_row_local_f32_ws<%d,%d>", nct, (int)a64);
        if (nct == 1) { if (a64) RLW(1, true); else RLW(1, false); }
        else          { if (a64) RLW(2, true); else RLW(2, false); }
#undef RLW
    } else if (c->dtype == AA_F32 && row_local_variant(c) >= 2) {
        const float *B = reinterpret_cast<const float *>(B_wideT);
        // few row blocks (short shards): split the contraction over column chunks as well, about
        // three blocks per CU, chunks of >= 512 columns (a multiple of every tile width)
        const long rblocks = c->n_pad / 128;
        int nsplit = 1, chunk = (int)c->p_pad;
        if (rblocks < 384 && g_row_local_split) {
            nsplit = (int)((768 + rblocks - 1) / rblocks);
            const int max_split = (int)(c->p_pad / 512);
            if (nsplit > max_split) nsplit = max_split;
            if (nsplit < 1) nsplit = 1;
        }
        double *dst = out_tall;
        if (nsplit > 1) {
            chunk = (int)round_up((c->p_pad + nsplit - 1) / nsplit, 128);
            nsplit = (int)((c->p_pad + chunk - 1) / chunk);
        }
        if (g_row_local_chunk > 0) {
            chunk = (int)round_up(g_row_local_chunk, 128);
            if (row_local_variant(c) == 7 || row_local_variant(c) == 5) chunk = (int)round_up(g_row_local_chunk, 32);
            nsplit = (int)((c->p_pad + chunk - 1) / chunk);
        }
        if (nsplit > 1) {
            AA_CHECK(c->rlPartial.alloc((size_t)nsplit * c->n_pad * c->KP * sizeof(double)));
            dst = c->rlPartial.as<double>();
        }
        dim3 grid((unsigned)rblocks, (unsigned)nsplit);
#define RLB(NCTV, TCV, DBV)                                                                   \
    do {                                                                                      \
        if (g_row_local_acc64)                                                                \
            hipLaunchKernelGGL((k_row_local_f32_blk<NCTV, TCV, DBV, true>), grid, block, 0, c->stream, \
                               c->X.as<float>(), c->p_pad, B, (int)c->p_pad, dst, c->n_pad, chunk);    \
        else                                                                                  \
            hipLaunchKernelGGL((k_row_local_f32_blk<NCTV, TCV, DBV, false>), grid, block, 0, c->stream, \
                               c->X.as<float>(), c->p_pad, B, (int)c->p_pad, dst, c->n_pad, chunk);    \
    } while (0)
        const int v = row_local_variant(c);
        PASS_NAME(1, "k_row_local_f32_blk[v%d,acc64=%d,split=%d]", v, g_row_local_acc64 != 0, nsplit);
        if (c->KP == 32) {
            switch (v) {
                case 3: RLB(1, 64, true); break;
                case 4: RLB(1, 128, false); break;
                case 5: RLB(1, 32, true); break;
                case 6: RLB(1, 128, true); break;
                case 7: RLB(1, 32, false); break;
                default: RLB(1, 64, false); break;
            }
        } else {
            switch (v) {
                case 3: case 5: case 6: RLB(2, 64, true); break;
                default: RLB(2, 64, false); break;
            }
        }
#undef RLB
        if (nsplit > 1) {
            const long elems = c->n_pad * c->KP;
            hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream,
                               (const double *)dst, elems, nsplit, out_tall);
        }
    } else if (c->dtype == AA_F32 && row_local_variant(c) == 1) {
        const float *B = reinterpret_cast<const float *>(B_wideT);
        dim3 grid((unsigned)(c->n_pad / 128));
        PASS_NAME(1, "k_row_local_f32_lds<%d>", c->KP / 32);
        if (c->KP == 32)
            hipLaunchKernelGGL(k_row_local_f32_lds<1>, grid, block, 0, c->stream, c->X.as<float>(),
                               c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad);
        else
            hipLaunchKernelGGL(k_row_local_f32_lds<2>, grid, block, 0, c->stream, c->X.as<float>(),
                               c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad);
    } else if (c->dtype == AA_F32) {
        const float *B = reinterpret_cast<const float *>(B_wideT);
        PASS_NAME(1, "k_row_local_f32<%d>", c->KP / 32);
        if (c->KP == 32) {
            constexpr int RT = 2;
            dim3 grid((unsigned)((c->n_pad + 128 * RT - 1) / (128 * RT)));
            hipLaunchKernelGGL((k_row_local_f32<1, RT>), grid, block, 0, c->stream,
                               c->X.as<float>(), c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad);
        } else {
            constexpr int RT = 1;
            dim3 grid((unsigned)((c->n_pad + 128 * RT - 1) / (128 * RT)));
            hipLaunchKernelGGL((k_row_local_f32<2, RT>), grid, block, 0, c->stream,
                               c->X.as<float>(), c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad);
        }
    } else {
        const double *B = reinterpret_cast<const double *>(B_wideT);
        dim3 grid((unsigned)(c->n_pad / 64));
        if (g_f64_mfma == 2 || (g_f64_mfma == 1 && c->n_pad / 32 >= 8 * 128)) {
            // wave-streaming kernel, W waves per block (see the float32 sibling)
            const int nt = c->KP / 16;
            const long tiles = c->n_pad / 32;
            const int wmax = nt == 2 ? 14 : 10;                // 160 KB of LDS
            int W = (int)((tiles + 255) / 256);
            if (W < 8) W = 8;
            if (W > wmax) W = wmax;
            const size_t lds = ((size_t)2 * c->KP * 66 + (size_t)W * 32 * 34) * sizeof(double);
            static bool attr_set64[64] = {false};               // the attribute is per device
            if (!attr_set64[c->device & 63]) {
                AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_row_local_f64_ws<2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_row_local_f64_ws<4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set64[c->device & 63] = true;
            }
            dim3 gw((unsigned)((tiles + W - 1) / W)), bw((unsigned)(64 * W));
            PASS_NAME(1, "k_row_local_f64_ws<%d>", nt);
            if (nt == 2)
                hipLaunchKernelGGL(k_row_local_f64_ws<2>, gw, bw, lds, c->stream, c->X.as<double>(), c->p_pad,
                                   B, (int)c->p_pad, out_tall, c->n_pad, W);
            else
                hipLaunchKernelGGL(k_row_local_f64_ws<4>, gw, bw, lds, c->stream, c->X.as<double>(), c->p_pad,
                                   B, (int)c->p_pad, out_tall, c->n_pad, W);
        } else if (g_f64_mfma) {
            // few row blocks (short, wide data): split the contraction over column chunks too
            const long rblocks = c->n_pad / 64;
            int nsplit = 1;
            if (rblocks < 192 && g_row_local_split) {
                nsplit = (int)((512 + rblocks - 1) / rblocks);
                const int max_split = (int)(c->p_pad / 256);       // >= 256 columns per chunk
                if (nsplit > max_split) nsplit = max_split;
                if (nsplit < 1) nsplit = 1;
            }
            int chunk = (int)c->p_pad;
            double *dst = out_tall;
            if (nsplit > 1) {
                chunk = (int)round_up((c->p_pad + nsplit - 1) / nsplit, 32);
                nsplit = (int)((c->p_pad + chunk - 1) / chunk);
                AA_CHECK(c->rlPartial.alloc((size_t)nsplit * c->n_pad * c->KP * sizeof(double)));
                dst = c->rlPartial.as<double>();
            }
            dim3 g2((unsigned)rblocks, (unsigned)nsplit);
            PASS_NAME(1, "k_row_local_f64_mfma<%d>[split=%d]", c->KP == 32 ? 2 : 4, nsplit);
            if (c->KP == 32)
                hipLaunchKernelGGL(k_row_local_f64_mfma<2>, g2, block, 0, c->stream, c->X.as<double>(),
                                   c->p_pad, B, (int)c->p_pad, dst, c->n_pad, chunk);
            else
                hipLaunchKernelGGL(k_row_local_f64_mfma<4>, g2, block, 0, c->stream, c->X.as<double>(),
                                   c->p_pad, B, (int)c->p_pad, dst, c->n_pad, chunk);
            if (nsplit > 1) {
                const long elems = c->n_pad * c->KP;
                hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream,
                                   (const double *)dst, elems, nsplit, out_tall);
            }
        } else if (PASS_NAME(1, "k_row_local_f64<%d>", c->KP), c->KP == 32)
            hipLaunchKernelGGL(k_row_local_f64<32>, grid, block, 0, c->stream, c->X.as<double>(),
                               c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad);
        else
            hipLaunchKernelGGL(k_row_local_f64<64>, grid, block, 0, c->stream, c->X.as<double>(),
                               c->p_pad, B, (int)c->p_pad, out_tall, c->n_pad);
    }
    AA_CHECK_HIP(hipGetLastError());
    gemm_event(c, 1);
    return AA_OK;
}

}  // namespace aa
