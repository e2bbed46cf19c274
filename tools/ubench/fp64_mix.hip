// fp64_mix.hip -- do the FP64 matrix cores and the FP64 vector ALU of a gfx950 SIMD run side by side?
// Workgroups of 8 waves (two per SIMD); mode 0: all waves issue independent v_mfma_f64_16x16x4 chains, mode 1: all
// waves issue independent v_fma_f64 chains, mode 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mix(double *out, int iters_m, int iters_v, int mode, double a, double b)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    double s = 0;
    if (do_mfma) {
        d4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (d4){0, 0, 0, 0};
        const double av = a + threadIdx.x * 1e-9;
        for (int i = 0; i < iters_m; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, acc[r], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    } else {
        double acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = threadIdx.x * 1e-9 + r;
        for (int i = 0; i < iters_v; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fma(acc[r], a, b);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double *out;
    if (hipMalloc(&out, sizeof(double) * 512 * 1024) != hipSuccess) return 1;
    const int blocks = 256;           // one workgroup per CU: 2 waves per SIMD
    const int im = 20000, iv = 110000;   // chosen so that an MFMA wave and a VALU wave take about the same time alone
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(mix, dim3(blocks), dim3(512), 0, 0, out, im, iv, mode, 1.0000001, 1e-9);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double nm = mode == 0 ? 8 : (mode == 2 ? 4 : 0), nv = mode == 1 ? 8 : (mode == 2 ? 4 : 0);
            const double fl = blocks * (nm * 4.0 * im * 2048.0 + nv * 16.0 * iv * 128.0);
            if (rep) printf("mode %d  %.3f ms  %.2f TFLOP/s  (mfma waves %g, valu waves %g per workgroup)\n", mode, ms, fl / ms / 1e9, nm, nv);
        }
    }
    return 0;
}
