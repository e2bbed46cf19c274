// mfma4_rate.hip -- does v_mfma_f64_4x4x4_4b keep its 17-cycle issue rate when every instruction
// reads different A/B registers (as in a real product), and how long is the accumulator drain?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int R, int NA>
__global__ __launch_bounds__(256) void vary(double *out, int iters, double s)
{
    double a[NA], b[NA], acc[R];
#pragma unroll
    for (int i = 0; i < NA; ++i) { a[i] = s + threadIdx.x * 1e-9 + i; b[i] = s * 0.5 + i; }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r)
                acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(k + r) % NA], b[k], acc[r], 0, 0, 0);
    }
    double t = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) t += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// product-shaped: NCH chains, each K deep, then results are read by VALU (drain), repeated
template <int NCH, int KD>
__global__ __launch_bounds__(256) void drain(double *out, int iters, double s)
{
    double a[KD], b[KD];
#pragma unroll
    for (int i = 0; i < KD; ++i) { a[i] = s + threadIdx.x * 1e-9 + i; b[i] = s * 0.5 + i; }
    double t = 0;
    for (int it = 0; it < iters; ++it) {
        double acc[NCH];
#pragma unroll
        for (int r = 0; r < NCH; ++r) acc[r] = 0;
#pragma unroll
        for (int k = 0; k < KD; ++k)
#pragma unroll
            for (int r = 0; r < NCH; ++r)
                acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[k], b[(k + r) % KD], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NCH; ++r) t += acc[r];
        a[0] = t * 1e-30 + a[0];          // make the next round depend on the drain
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <typename F>
static float time_it(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *out; (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
    const int iters = 2000, blocks = 256;   // 1 wave per SIMD
#define V(R, NA) { float ms = time_it([&] { hipLaunchKernelGGL((vary<R, NA>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }); \
    printf("vary R=%d NA=%d: %.1f cycles/mfma (2.4GHz nominal)\n", R, NA, ms * 1e-3 * 2.4e9 / ((double)R * NA * iters)); }
    V(4, 4) V(8, 4) V(16, 4) V(16, 8) V(8, 16)
#define D(NCH, KD) { float ms = time_it([&] { hipLaunchKernelGGL((drain<NCH, KD>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }); \
    printf("drain NCH=%d KD=%d: %.1f cycles/mfma, %.0f cycles/round\n", NCH, KD, ms * 1e-3 * 2.4e9 / ((double)NCH * KD * iters), ms * 1e-3 * 2.4e9 / iters); }
    D(4, 4) D(16, 4) D(16, 8) D(32, 4)
    return 0;
}
