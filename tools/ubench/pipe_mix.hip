// pipe_mix.hip -- what does a gfx950 SIMD overlap with v_mfma_f64_16x16x4_f64?
//
// Experiment B (same wave): a loop body of 4 DEPENDENT MFMAs followed by NF filler instructions of one kind
// (independent of the MFMAs), at 1, 2 and 4 waves per SIMD.  Output: shader cycles per loop body, per SIMD
// (= wave cycles / waves per SIMD), so that "256" means the matrix pipe is saturated and anything above it
// is time the filler took away from it.
// Experiment A (two waves of one SIMD, different roles): waves 0-3 run the MFMA loop, waves 4-7 a filler-only
// loop; each role's cycles alone and together.
//
// build: hipcc -O3 --offload-arch=gfx950 -o pipe_mix pipe_mix.hip ; run: ./pipe_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

enum { F_NONE, F_FMA64, F_ADD64, F_MOV32, F_XOR32, F_DPP32, F_FMA32, F_MOV64, F_LDSR128, F_LDSW128, F_SWAP32, F_LDSW2_64, F_MUL64, F_NKINDS };
static const char *kNames[] = {"none", "v_fma_f64", "v_add_f64", "v_mov_b32", "v_xor_b32", "v_mov_b32_dpp", "v_fma_f32", "v_mov_b64",
                               "ds_read_b128", "ds_write_b128", "v_permlane32_swap", "ds_write2_b64", "v_mul_f64"};

template <int KIND>
__device__ __forceinline__ void filler(double &a, double &b, int &i0, int &i1, float &f, d2 &q, int ldsaddr)
{
    if (KIND == F_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (KIND == F_ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    if (KIND == F_MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    if (KIND == F_MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(i0) : "v"(i1));
    if (KIND == F_XOR32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i0) : "v"(i1));
    if (KIND == F_DPP32) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(i0) : "v"(i1));
    if (KIND == F_FMA32) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));
    if (KIND == F_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(a) : "v"(b));
    if (KIND == F_LDSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(ldsaddr));
    if (KIND == F_LDSW128) asm volatile("ds_write_b128 %0, %1" : : "v"(ldsaddr), "v"(q));
    if (KIND == F_LDSW2_64) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" : : "v"(ldsaddr), "v"(a), "v"(b));
    if (KIND == F_SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(i0), "+v"(i1));
}

// role 0: MFMA body with NF fillers; role 1 (waves >= 4 when split != 0): fillers only, 16 per body
template <int KIND, int NF>
__global__ __launch_bounds__(1024) void body(long long *cyc, double *sink, int iters, int split, double av, double bv)
{
    __shared__ d4 lds[4096];
    const int wave = threadIdx.x >> 6;
    const int ldsaddr = (threadIdx.x & 1023) * 32;
    lds[threadIdx.x] = (d4){av, bv, av, bv};
    __syncthreads();
    d4 acc = {0, 0, 0, 0};
    double a[8], b = bv;
    int i0[8], i1 = threadIdx.x;
    float f[8];
    d2 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = av + j;
        i0[j] = j;
        f[j] = (float)j;
        q[j] = (d2){av, bv};
    }
    const bool filler_only = split && ((wave >= 4) == (split == 1));
    const bool mfma_only = split && !filler_only;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (filler_only) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) filler<KIND>(a[j & 7], b, i0[j & 7], i1, f[j & 7], q[j & 7], ldsaddr);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            if (!mfma_only) {
#pragma unroll
                for (int j = 0; j < NF; ++j) filler<KIND>(a[j & 7], b, i0[j & 7], i1, f[j & 7], q[j & 7], ldsaddr);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] + i0[j] + f[j] + q[j][0] + q[j][1];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s + i1;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * 32 + wave] = t0;
        cyc[blockIdx.x * 32 + 16 + wave] = t1;
    }
}

template <int KIND, int NF>
static void run(long long *d_cyc, double *d_sink, int wps, int split)
{
    const int iters = 4000, blocks = 256;
    std::vector<long long> h(blocks * 32);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((body<KIND, NF>), dim3(blocks), dim3(256 * wps), 0, 0, d_cyc, d_sink, iters, split, 1.0000001, 1e-9);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * blocks * 32, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0;
    int n0 = 0, n1 = 0;
    for (int b = 0; b < blocks; ++b) {
        long long lo = h[b * 32], hi = h[b * 32 + 16];
        for (int w = 0; w < 4 * wps; ++w) {
            const long long d = h[b * 32 + 16 + w] - h[b * 32 + w];
            if (split && ((w >= 4) == (split == 1))) { m1 += d; ++n1; } else if (split) { m0 += d; ++n0; }
            lo = std::min(lo, h[b * 32 + w]);
            hi = std::max(hi, h[b * 32 + 16 + w]);
        }
        if (!split) { m0 += (double)(hi - lo) * wps; n0 += 1; }     // block wall time: all waves together
    }
    m0 /= n0 * (double)iters;
    if (!split)
        printf("B  %-18s NF=%2d waves/SIMD=%d  wave cycles/body %8.1f  per SIMD %8.1f\n", kNames[KIND], NF, wps, m0, m0 / wps);
    else
        printf("A%d %-18s (16 per body) beside an MFMA wave: mfma wave %8.1f cycles/body, filler wave %8.1f cycles/body\n", split, kNames[KIND], m0,
               m1 / (n1 * (double)iters));
}


// Experiment C: a "product": 12 MFMAs on two accumulator chains (8 interleaved + 4) and NF v_add_f64, either in a burst
// behind the MFMAs (INTER = 0; DEP = 1: the burst's first instruction reads the accumulator, as the combinations of a
// three-product complex multiplication do) or spread NF/12 behind every MFMA (INTER = 1).
template <int NF, int INTER, int DEP, int PRIO = 0>
__global__ __launch_bounds__(1024) void product(long long *cyc, double *sink, int iters, double av, double bv)
{
    const int wave = threadIdx.x >> 6;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double a[8], b = bv;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = av + j;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (m < 8 && (m & 1))
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc1, 0, 0, 0);
            else
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc0, 0, 0, 0);
            if (INTER) {
#pragma unroll
                for (int j = 0; j < NF / 12; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j & 7]) : "v"(b));
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (!INTER) {
            if (DEP && NF > 0) {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[0]) : "v"(acc0[0]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[1]) : "v"(acc1[0]));
            }
#pragma unroll
            for (int j = DEP ? 2 : 0; j < NF; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j & 7]) : "v"(b));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * 32 + wave] = t0;
        cyc[blockIdx.x * 32 + 16 + wave] = t1;
    }
}

template <int NF, int INTER, int DEP, int PRIO = 0>
static void run_product(long long *d_cyc, double *d_sink)
{
    const int iters = 2000, blocks = 256;
    std::vector<long long> h(blocks * 32);
    for (int wps : {1, 2, 4}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((product<NF, INTER, DEP, PRIO>), dim3(blocks), dim3(256 * wps), 0, 0, d_cyc, d_sink, iters, 1.0000001, 1e-9);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * blocks * 32, hipMemcpyDeviceToHost);
        double wall = 0;
        for (int b = 0; b < blocks; ++b) {
            long long lo = h[b * 32], hi = h[b * 32 + 16];
            for (int w = 0; w < 4 * wps; ++w) {
                lo = std::min(lo, h[b * 32 + w]);
                hi = std::max(hi, h[b * 32 + 16 + w]);
            }
            wall += (double)(hi - lo);
        }
        wall /= blocks * (double)iters * wps;      // SIMD cycles per product
        printf("C%s NF=%3d %s%s waves/SIMD=%d  SIMD cycles per 12-MFMA product %8.1f  matrix-pipe share %.3f\n", PRIO ? " setprio" : "", NF,
               INTER ? "interleaved" : "burst", DEP ? " (dependent)" : "", wps, wall, 768.0 / wall);
    }
}

template <int KIND>
static void kind(long long *d_cyc, double *d_sink)
{
    for (int wps : {1, 2, 4}) {
        run<KIND, 8>(d_cyc, d_sink, wps, 0);
        run<KIND, 16>(d_cyc, d_sink, wps, 0);
        run<KIND, 32>(d_cyc, d_sink, wps, 0);
    }
    run<KIND, 0>(d_cyc, d_sink, 2, 1);
    run<KIND, 0>(d_cyc, d_sink, 2, 2);
}

int main()
{
    long long *d_cyc;
    double *d_sink;
    if (hipMalloc(&d_cyc, sizeof(long long) * 256 * 32) != hipSuccess) return 1;
    if (hipMalloc(&d_sink, sizeof(double) * 256 * 1024) != hipSuccess) return 1;
    run_product<0, 0, 0>(d_cyc, d_sink);
    run_product<24, 0, 1>(d_cyc, d_sink);
    run_product<48, 0, 1>(d_cyc, d_sink);
    run_product<96, 0, 1>(d_cyc, d_sink);
    run_product<144, 0, 1>(d_cyc, d_sink);
    run_product<48, 0, 0>(d_cyc, d_sink);
    run_product<96, 0, 0>(d_cyc, d_sink);
    run_product<24, 1, 0>(d_cyc, d_sink);
    run_product<48, 1, 0>(d_cyc, d_sink);
    run_product<96, 1, 0>(d_cyc, d_sink);
    run_product<144, 1, 0>(d_cyc, d_sink);
    run_product<48, 0, 1, 1>(d_cyc, d_sink);
    run_product<96, 0, 1, 1>(d_cyc, d_sink);
    run_product<144, 0, 1, 1>(d_cyc, d_sink);
    run_product<96, 0, 0, 1>(d_cyc, d_sink);
    for (int wps : {1, 2, 4}) run<F_NONE, 0>(d_cyc, d_sink, wps, 0);
    kind<F_FMA64>(d_cyc, d_sink);
    kind<F_ADD64>(d_cyc, d_sink);
    kind<F_MUL64>(d_cyc, d_sink);
    kind<F_MOV32>(d_cyc, d_sink);
    kind<F_XOR32>(d_cyc, d_sink);
    kind<F_DPP32>(d_cyc, d_sink);
    kind<F_FMA32>(d_cyc, d_sink);
    kind<F_MOV64>(d_cyc, d_sink);
    kind<F_LDSR128>(d_cyc, d_sink);
    kind<F_LDSW128>(d_cyc, d_sink);
    kind<F_LDSW2_64>(d_cyc, d_sink);
    kind<F_SWAP32>(d_cyc, d_sink);
    return 0;
}
