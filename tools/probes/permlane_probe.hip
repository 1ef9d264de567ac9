// scratch: semantics of v_permlane16_swap / v_permlane32_swap with both operands equal, as the
// four-lane butterfly of k_qp_quad uses them: out[l] must be in[l&15] + in[16+(l&15)] + in[32+(l&15)] + in[48+(l&15)]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(double *o, const double *in)
{
    double v = in[threadIdx.x];
    for (int step = 0; step < 2; ++step) {
        unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        u2 a = step ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        u2 b = step ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    o[threadIdx.x] = v;
}
int main()
{
    double h[64], r[64], *d, *o;
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 0.001 + (i >> 4) * 100.0;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int s = i & 15;
        const double want = (h[s] + h[16 + s]) + (h[32 + s] + h[48 + s]);
        if (r[i] != want) { ++bad; if (bad < 5) printf("lane %d: got %.6f want %.6f\n", i, r[i], want); }
    }
    printf("permlane butterfly: %s (%d bad)\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
