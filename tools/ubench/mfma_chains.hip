// mfma_chains.hip -- how many independent v_mfma_f64_16x16x4 accumulator chains does a SIMD need in flight?
// R chains per wave (VGPR accumulators), W waves per SIMD; prints TFLOP/s and cycles per MFMA per SIMD at 2.1 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(512) void chains(double *out, int iters, double a, double b)
{
    d4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = (d4){0, 0, 0, 0};
    const double av = a + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, acc[r], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R>
static void run(double *out, int wps)
{
    const int iters = 40000 / R;
    const int threads = wps >= 2 ? 512 : 256, blocks = 256 * wps * 256 / threads;   // wps waves per SIMD over 256 CUs
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(chains<R>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)blocks * (threads / 64) * R * iters;
        if (rep) printf("chains/wave %d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s  %.0f cycles per MFMA per SIMD (2.1 GHz)\n", R, wps, ms,
                        n_mfma * 2048 / ms / 1e9, ms * 1e-3 * 2.1e9 / (n_mfma / 1024));
    }
}

int main()
{
    double *out;
    if (hipMalloc(&out, sizeof(double) * 512 * 4096) != hipSuccess) return 1;
    for (int wps : {1, 2, 4}) {
        run<1>(out, wps); run<2>(out, wps); run<4>(out, wps); run<8>(out, wps);
    }
    return 0;
}
