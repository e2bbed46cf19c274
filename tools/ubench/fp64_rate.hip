// fp64_rate.hip -- microbenchmarks behind DESIGN.md's FP64 numbers: how fast can ONE wave,
// or several waves per SIMD, issue v_fma_f64 / v_mfma_f64 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R>
__global__ __launch_bounds__(256) void fma_chain(double *out, int iters, double a, double b)
{
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = threadIdx.x * 1e-9 + r;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma(acc[r], a, b);
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));

// R independent 16x16x4 f64 MFMA accumulators
template <int R>
__global__ __launch_bounds__(256) void mfma16_chain(double *out, int iters, double a, double b)
{
    d4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = (d4){0, 0, 0, 0};
    const double av = a + threadIdx.x * 1e-9, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[r], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// R independent 4x4x4 (4 blocks) f64 MFMA accumulators: 1 double per lane
template <int R>
__global__ __launch_bounds__(256) void mfma4_chain(double *out, int iters, double a, double b)
{
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0;
    const double av = a + threadIdx.x * 1e-9, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[r], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// mixed: each iteration RV VALU fma + RM MFMA 4x4x4 (independent), same wave
template <int RV, int RM>
__global__ __launch_bounds__(256) void mixed_chain(double *out, int iters, double a, double b)
{
    double acc[RV], m[RM];
#pragma unroll
    for (int r = 0; r < RV; ++r) acc[r] = threadIdx.x * 1e-9 + r;
#pragma unroll
    for (int r = 0; r < RM; ++r) m[r] = 0;
    const double av = a + threadIdx.x * 1e-9, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < RM; ++r) m[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, m[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RV; ++r) acc[r] = fma(acc[r], a, b);
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < RV; ++r) s += acc[r];
#pragma unroll
    for (int r = 0; r < RM; ++r) s += m[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_it(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();                       // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main()
{
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 65536));
    const int iters = 20000;
    printf("# kind R waves_per_simd ms Tflops cycles_per_wave_instr(at 2.4GHz nominal)\n");
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;   // 256 threads = 4 waves = one per SIMD
#define RUN_FMA(R) { float ms = time_it([&] { hipLaunchKernelGGL(fma_chain<R>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }); \
        double fl = 2.0 * R * iters * 256.0 * blocks; \
        printf("fma %d %d %.3f %.2f %.2f\n", R, wps, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)R * iters * wps)); }
        RUN_FMA(2) RUN_FMA(4) RUN_FMA(8) RUN_FMA(16) RUN_FMA(32)
#define RUN_M16(R) { float ms = time_it([&] { hipLaunchKernelGGL(mfma16_chain<R>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }); \
        double fl = 2.0 * 1024 * R * iters * 4.0 * blocks; \
        printf("mfma16x16x4 %d %d %.3f %.2f %.2f\n", R, wps, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)R * iters * wps)); }
        RUN_M16(1) RUN_M16(2) RUN_M16(4)
#define RUN_M4(R) { float ms = time_it([&] { hipLaunchKernelGGL(mfma4_chain<R>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }); \
        double fl = 2.0 * 256 * R * iters * 4.0 * blocks; \
        printf("mfma4x4x4 %d %d %.3f %.2f %.2f\n", R, wps, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)R * iters * wps)); }
        RUN_M4(1) RUN_M4(2) RUN_M4(4) RUN_M4(8)
#define RUN_MIX(RV, RM) { float ms = time_it([&] { hipLaunchKernelGGL((mixed_chain<RV, RM>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }); \
        double fl = (2.0 * 64 * RV + 2.0 * 256 * RM) * iters * 4.0 * blocks; \
        printf("mixed v%d m%d %d %.3f %.2f\n", RV, RM, wps, ms, fl / ms / 1e9); }
        RUN_MIX(16, 4) RUN_MIX(16, 8) RUN_MIX(8, 8)
    }
    hipFree(out);
    return 0;
}
