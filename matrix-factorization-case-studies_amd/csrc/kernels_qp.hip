// kernels_qp.hip -- batched per-sample simplex QP (the weights update) and the
// stateless row-wise simplex projection.
//
// Reference: for every sample t independently, quad_simplex_spg(A, b_t, z_t)
// (spg.py:286-398) called from the serial loops at archetypal_analysis.py:359-366 and
// gpnh_convex_coding.py:244-251.  A (k x k) is shared by all samples.
//
// Mapping: ONE LANE PER SAMPLE.  The whole SPG state of a sample (x, g = Ax + b, d,
// Ad: 4*KQ doubles) lives in that lane's registers; A is read with wave-uniform
// addresses (scalar loads / broadcast), so the k x k mat-vec costs KQ^2 v_fma_f64 per
// 64 samples and there is no cross-lane traffic at all.  Samples finish after very
// different numbers of iterations (heavy tail), so a lane that finishes pulls the next
// sample from a global counter instead of idling until its wave is done.
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

#define QP_MAXMEM 8

template <int KQ>
__device__ __forceinline__ double qp_project_threshold(const double (&x)[KQ], const double (&g)[KQ],
                                                       double a, int k)
{
    // threshold t of the projection of w = x - a*g (components >= k excluded)
    double mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < KQ; ++i)
        if (i < k) mx = fmax(mx, x[i] - a * g[i]);
    double t = mx - 1.0;
    int prev = 0;
    for (int pass = 0; pass < KQ + 2; ++pass) {
        double s = 0.0;
        int c = 0;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const double w = x[i] - a * g[i];
            if (i < k && w > t) {
                s += w;
                c += 1;
            }
        }
        const double tn = (s - 1.0) / (double)c;
        const bool conv = (prev > 0) && (c >= prev);
        t = tn;
        prev = c;
        if (conv) break;
    }
    return t;
}

template <int KQ>
__global__ __launch_bounds__(64) void k_qp(const double *__restrict__ A /*[KQ][KQ]*/,
                                           const double *__restrict__ B, long stride_j,
                                           long stride_t, const double *__restrict__ bscale,
                                           const double *__restrict__ Z0, double *__restrict__ Z,
                                           int ldz, long n, int k, aa_qp_params p,
                                           int *__restrict__ iters,
                                           unsigned long long *__restrict__ stats,
                                           unsigned int *__restrict__ counter)
{
    double x[KQ], g[KQ], d[KQ], Ad[KQ];
    double f = 0.0, alpha = 1.0, fmem[QP_MAXMEM];
    int n_iter = 0, n_feval = 0;
    long row = -1;
    bool active = false, exhausted = false;
    const int mem = p.memory < 1 ? 1 : (p.memory > QP_MAXMEM ? QP_MAXMEM : p.memory);

    while (true) {
        if (!active && !exhausted) {
            const unsigned int nxt = atomicAdd(counter, 1u);
            if ((long)nxt < n) {
                row = (long)nxt;
                active = true;
                // ---- start-up: x = P(z0); g = A x + b; f = x'(g + b)/2      (spg.py:298-315)
                double b[KQ];
#pragma unroll
                for (int i = 0; i < KQ; ++i) {
                    x[i] = (i < k) ? Z0[row * ldz + i] : 0.0;
                    g[i] = 0.0;
                    b[i] = (i < k) ? -B[i * stride_j + row * stride_t] * (bscale ? bscale[i] : 1.0) : 0.0;
                }
                const double t0 = qp_project_threshold<KQ>(x, g, 0.0, k);
#pragma unroll
                for (int i = 0; i < KQ; ++i) x[i] = (i < k) ? fmax(x[i] - t0, 0.0) : 0.0;
                double xg = 0.0, xb = 0.0;
#pragma unroll
                for (int i = 0; i < KQ; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < KQ; ++j) s = fma(A[i * KQ + j], x[j], s);
                    g[i] = s + b[i];
                    xg = fma(x[i], g[i], xg);
                    xb = fma(x[i], b[i], xb);
                }
                f = 0.5 * (xg + xb);
                n_feval = 1;
                n_iter = 0;
#pragma unroll
                for (int i = 0; i < QP_MAXMEM; ++i) fmem[i] = NAN;
            } else {
                exhausted = true;   // queue drained: this lane idles
            }
        }
        if (!__any(active)) break;
        if (active) {
            // ---- one pass of the loop at spg.py:318-396
            if (n_iter == 0) {
                if (p.alpha_min <= p.alpha0 && p.alpha0 <= p.alpha_max) {
                    alpha = p.alpha0;
                } else {
                    const double t1 = qp_project_threshold<KQ>(x, g, 1.0, k);
                    double ainv = 0.0;
#pragma unroll
                    for (int i = 0; i < KQ; ++i)
                        if (i < k) ainv = fmax(ainv, fabs(fmax(x[i] - g[i] - t1, 0.0) - x[i]));
                    if (fabs(ainv) < 1e-12) ainv = 1.0;
                    alpha = fmin(fmax(p.alpha_min, 1.0 / ainv), p.alpha_max);
                }
            }
            const double td = qp_project_threshold<KQ>(x, g, alpha, k);
            double delta = 0.0, dd = 0.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                d[i] = (i < k) ? fmax(x[i] - alpha * g[i] - td, 0.0) - x[i] : 0.0;
                delta = fma(d[i], g[i], delta);
                dd = fma(d[i], d[i], dd);
            }
            double dAd = 0.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < KQ; ++j) s = fma(A[i * KQ + j], d[j], s);
                Ad[i] = s;
                dAd = fma(d[i], s, dAd);
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
                x[i] = fma(lam, d[i], x[i]);
                g[i] = fma(lam, Ad[i], g[i]);
            }
            const double sksk = lam * lam * dd;
            const double beta = lam * (lam * dAd);
            alpha = (beta <= 0.0) ? p.alpha_max : fmin(p.alpha_max, fmax(p.alpha_min, sksk / beta));
            f = f_new;
            n_feval += 1;

            const double tr = qp_project_threshold<KQ>(x, g, 1.0, k);
            double r2 = 0.0, rinf = 0.0;
#pragma unroll
            for (int i = 0; i < KQ; ++i)
                if (i < k) {
                    const double r = fmax(x[i] - g[i] - tr, 0.0) - x[i];
                    r2 = fma(r, r, r2);
                    rinf = fmax(rinf, fabs(r));
                }
            n_iter += 1;
            const bool conv = (sqrt(r2) < p.epsilon_two) || (rinf < p.epsilon_one);
            if (conv || n_feval > p.max_feval || n_iter >= p.max_iterations) {
#pragma unroll
                for (int i = 0; i < KQ; ++i)
                    if (i < k) Z[row * ldz + i] = x[i];
                if (iters) iters[row] = n_iter;
                atomicAdd(&stats[0], (unsigned long long)n_iter);
                atomicMax(&stats[1], (unsigned long long)n_iter);
                active = false;
            }
        }
    }
}

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
    const double t0 = qp_project_threshold<KQ>(x, g, 0.0, k);
#pragma unroll
    for (int i = 0; i < KQ; ++i)
        if (i < k) Z[row * ldz + i] = fmax(x[i] - t0, 0.0);
}

// Apad: device buffer [KQ][KQ] (zero padded copy of the host k x k matrix)
int launch_qp(Ctx *c, const double *A_host, const double *Btall, long stride_j, long stride_t,
              const double *bscale_host, double *Ztall, int ldz, long n, int k,
              const aa_qp_params *p, int *iters_dev, aa_qp_stats *stats)
{
    int KQ = 4;
    while (KQ < k) KQ *= 2;
    AA_REQUIRE(KQ <= 64, AA_ERR_ARG, "QP: k = %d > 64 unsupported", k);
    // stage A (padded) + bscale + counters in qpStats buffer:
    //   [0..1] stats, [2] counter, then doubles: A[KQ*KQ], bscale[KQ]
    const size_t hdr = 64;
    const size_t bytes = hdr + ((size_t)KQ * KQ + KQ) * sizeof(double);
    AA_CHECK(c->qpStats.alloc(bytes));
    std::vector<unsigned char> host(bytes, 0);
    double *Ah = reinterpret_cast<double *>(host.data() + hdr);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) Ah[i * KQ + j] = A_host[i * k + j];
    double *bs = Ah + (size_t)KQ * KQ;
    for (int i = 0; i < KQ; ++i) bs[i] = (bscale_host && i < k) ? bscale_host[i] : 1.0;
    AA_CHECK_HIP(hipMemcpyAsync(c->qpStats.p, host.data(), bytes, hipMemcpyHostToDevice, c->stream));
    AA_CHECK_HIP(hipStreamSynchronize(c->stream));   // host vector goes out of scope
    unsigned long long *st = c->qpStats.as<unsigned long long>();
    unsigned int *counter = reinterpret_cast<unsigned int *>(st + 2);
    const double *Ad = reinterpret_cast<const double *>(reinterpret_cast<unsigned char *>(c->qpStats.p) + hdr);
    const double *bsd = bscale_host ? Ad + (size_t)KQ * KQ : nullptr;

    if (p->max_iterations <= 0) {
        dim3 grid((unsigned)((n + 255) / 256));
#define QPP(KQV) hipLaunchKernelGGL(k_qp_project_only<KQV>, grid, dim3(256), 0, c->stream, Ztall, Ztall, ldz, n, k)
        switch (KQ) { case 4: QPP(4); break; case 8: QPP(8); break; case 16: QPP(16); break;
                      case 32: QPP(32); break; default: QPP(64); break; }
#undef QPP
    } else {
        long waves = (n + 63) / 64;
        const long max_waves = 256L * 4 * 2;     // 2 waves per SIMD on every CU
        if (waves > max_waves) waves = max_waves;
        dim3 grid((unsigned)waves);
#define QPL(KQV) hipLaunchKernelGGL(k_qp<KQV>, grid, dim3(64), 0, c->stream, Ad, Btall, stride_j, stride_t, bsd, (const double *)Ztall, Ztall, ldz, n, k, *p, iters_dev, st, counter)
        switch (KQ) { case 4: QPL(4); break; case 8: QPL(8); break; case 16: QPL(16); break;
                      case 32: QPL(32); break; default: QPL(64); break; }
#undef QPL
    }
    AA_CHECK_HIP(hipGetLastError());
    if (stats) {
        unsigned long long h[2] = {0, 0};
        AA_CHECK_HIP(hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        AA_CHECK_HIP(hipStreamSynchronize(c->stream));
        stats->total_passes = (long)h[0];
        stats->max_passes = (int)h[1];
        stats->reserved = 0;
    }
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
