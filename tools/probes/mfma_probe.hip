// scratch: issue rate / latency of v_mfma_f64_16x16x4_f64 and v_fma_f64 on one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(int mode, int n, double *out, long long *cyc)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    const long long t0 = clock64();
    if (mode == 0) {          // dependent MFMA chain
        for (int i = 0; i < n; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else if (mode == 1) {   // 4 independent MFMA chains
        for (int i = 0; i < n; i += 4) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    } else if (mode == 2) {   // dependent v_fma_f64 chain
        for (int i = 0; i < n; ++i) v0 = fma(a, b, v0);
    } else {                  // 4 independent v_fma_f64 chains
        for (int i = 0; i < n; i += 4) {
            v0 = fma(a, b, v0); v1 = fma(a, v0 * 0 + b, v1); v2 = fma(b, a, v2); v3 = fma(b, b, v3);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3;
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&cyc, 4096 * 8);
    const char *names[] = {"mfma_f64 16x16x4 dependent", "mfma_f64 16x16x4 4 chains", "v_fma_f64 dependent", "v_fma_f64 4 chains"};
    for (int grid : {1, 1024, 4096})
        for (int mode = 0; mode < 4; ++mode) {
            const int n = 4096;
            hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, mode, n, out, cyc);
            hipDeviceSynchronize();
            long long h[4096]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
            printf("grid %4d  %-30s %.1f cycles per instruction\n", grid, names[mode], s / grid / n);
        }
    return 0;
}
