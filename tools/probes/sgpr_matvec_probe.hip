// scratch: cycles of a 32x32 float64 mat-vec per lane (one sample per lane) with the matrix fed
// through SCALAR loads (s_load_dwordx16 -> v_fma_f64 with an SGPR-pair operand), software
// pipelined D chunks ahead, one wave per SIMD.  Compare with the lane QP kernel's MFMA + LDS
// mat-vec (~17 800 cycles) and the VALU floor (1024 v_fma_f64 = 4096 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(4))) double *cptr_t;

template <int KQ, int CH, int D>
__device__ __forceinline__ void matvec_sgpr(const double *__restrict__ A, const double (&v)[KQ], double (&out)[KQ])
{
    constexpr int NCH = KQ * KQ / CH, CPR = KQ / CH;
    cptr_t Ap = (cptr_t)(unsigned long long)A;
    asm volatile("" : "+s"(Ap));
#pragma unroll
    for (int i = 0; i < KQ; ++i) out[i] = 0.0;
    double buf[D + 1][CH];
#pragma unroll
    for (int q = 0; q < D; ++q)
#pragma unroll
        for (int e = 0; e < CH; ++e) buf[q][e] = Ap[q * CH + e];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + D < NCH) {
#pragma unroll
            for (int e = 0; e < CH; ++e) buf[(c + D) % (D + 1)][e] = Ap[(c + D) * CH + e];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int j = c / CPR, i0 = (c % CPR) * CH;
#pragma unroll
        for (int e = 0; e < CH; ++e) out[i0 + e] = fma(buf[c % (D + 1)][e], v[j], out[i0 + e]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CH, int D>
__global__ __launch_bounds__(64) void probe(const double *__restrict__ A, const double *__restrict__ X, double *__restrict__ O,
                                            long long *cyc, int reps)
{
    double v[32], out[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = X[threadIdx.x * 32 + i];
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        matvec_sgpr<32, CH, D>(A, v, out);
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = out[i] * 1e-3 + v[i];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
#pragma unroll
    for (int i = 0; i < 32; ++i) O[(blockIdx.x * 64 + threadIdx.x) * 32 + i] = v[i];
}

template <int CH, int D> void run(const char *name, double *A, double *X, double *O, long long *cyc)
{
    for (int grid : {1, 1024}) {
        hipLaunchKernelGGL((probe<CH, D>), dim3(grid), dim3(64), 0, 0, A, X, O, cyc, 50);
        hipDeviceSynchronize();
        long long h[1024];
        hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < grid; ++i) s += h[i];
        printf("%-28s grid %4d: %.0f cycles per mat-vec\n", name, grid, s / grid / 50);
    }
}
int main()
{
    double *A, *X, *O; long long *cyc;
    hipMalloc(&A, 32 * 32 * 8); hipMalloc(&X, 64 * 32 * 8); hipMalloc(&O, 1024 * 64 * 32 * 8); hipMalloc(&cyc, 1024 * 8);
    hipMemset(A, 0, 32 * 32 * 8); hipMemset(X, 0, 64 * 32 * 8);
    run<16, 1>("chunk 16, 1 ahead", A, X, O, cyc);
    run<8, 2>("chunk 8, 2 ahead", A, X, O, cyc);
    run<8, 3>("chunk 8, 3 ahead", A, X, O, cyc);
    run<16, 2>("chunk 16, 2 ahead", A, X, O, cyc);
    return 0;
}
