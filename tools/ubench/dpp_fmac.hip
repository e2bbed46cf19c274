// dpp_fmac.hip -- gfx950: v_fmac_f64 with a DPP row_newbcast source (the only DPP control the FP64 ALU takes).
//
//   (1) semantics: src0 of lane l is read from lane 16 (l >> 4) + n of the same 16-lane row;
//   (2) issue rate: cycles per instruction for (a) one dependent chain, (b) 2 / 4 independent accumulators,
//       (c) the complex matrix-vector pattern of action_thin_kernel (32 FMACs into 2 + 2 accumulators),
//       each at 1, 2 and 4 waves per SIMD, beside plain v_fma_f64;
//
//   (3) the same with only lanes 0..31 active: idle passes are NOT skipped (same time).
// The tick of s_memtime is not the shader clock: read the numbers relative to one another (peak = the 4-waves row).
//
// build: hipcc -O3 --offload-arch=gfx950 -o dpp_fmac dpp_fmac.hip ; run: ./dpp_fmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMAC_BC(ACC, X, M, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(X), "v"(M))
#define FMAC_BCN(ACC, X, M, N) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(X), "v"(M))

__global__ void semantics(double *out)
{
    const int l = threadIdx.x;
    double x = 100.0 + l, m = 1.0, acc = 0.0;
    FMAC_BC(acc, x, m, 5);
    out[l] = acc;                       // expect 100 + 16 (l >> 4) + 5
    double acc2 = 0.0;
    FMAC_BCN(acc2, x, m, 15);
    out[64 + l] = acc2;                 // expect -(100 + 16 (l >> 4) + 15)
}

// MODE 0: dependent chain of DPP FMACs; 1: two accumulators; 2: four; 3: plain v_fma_f64 dependent; 4: matvec pattern
template <int MODE>
__global__ __launch_bounds__(1024) void rate(long long *cyc, double *sink, int iters, double mv, int half)
{
    if (half && (threadIdx.x & 32))          // only lanes 0..31 of every wave stay active: are the idle passes skipped?
        return;
    double x = 1.0 + threadIdx.x * 1e-3, xi = 0.5;
    double m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        m[j] = mv * (j + 1);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define R4(N) FMAC_BC(a0, x, m[N], N); FMAC_BC(a0, x, m[N + 1], N); FMAC_BC(a0, x, m[N + 2], N); FMAC_BC(a0, x, m[N + 3], N);
            R4(0) R4(4) R4(8) R4(12) R4(0) R4(4) R4(8) R4(12)
#undef R4
        } else if (MODE == 1) {
#define R4(N) FMAC_BC(a0, x, m[N], N); FMAC_BC(a1, x, m[N + 1], N); FMAC_BC(a0, x, m[N + 2], N); FMAC_BC(a1, x, m[N + 3], N);
            R4(0) R4(4) R4(8) R4(12) R4(0) R4(4) R4(8) R4(12)
#undef R4
        } else if (MODE == 2) {
#define R4(N) FMAC_BC(a0, x, m[N], N); FMAC_BC(a1, x, m[N + 1], N); FMAC_BC(a2, x, m[N + 2], N); FMAC_BC(a3, x, m[N + 3], N);
            R4(0) R4(4) R4(8) R4(12) R4(0) R4(4) R4(8) R4(12)
#undef R4
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(m[j & 15]));
        } else {
            // y += M x, 8 complex entries per lane: re += mr xr - mi xi ; im += mr xi + mi xr   (m[2j], m[2j+1])
#define C1(J) FMAC_BC(a0, x, m[2 * J], J); FMAC_BC(a1, xi, m[2 * J], J); FMAC_BCN(a0, xi, m[2 * J + 1], J); FMAC_BC(a1, x, m[2 * J + 1], J);
            C1(0) C1(1) C1(2) C1(3) C1(4) C1(5) C1(6) C1(7)
#undef C1
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int MODE>
static void run(const char *name, long long *d_cyc, double *d_sink, int half = 0)
{
    const int iters = 2000;
    for (int waves : {4, 8, 16}) {       // 1, 2 or 4 waves per SIMD
        hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(64 * waves), 0, 0, d_cyc, d_sink, iters, 1e-9, half);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(64 * waves), 0, 0, d_cyc, d_sink, iters, 1e-9, half);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 16);
        hipMemcpy(h.data(), d_cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        double sum = 0;
        int cnt = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < waves; ++w) {
                sum += (double)h[b * 16 + w];
                ++cnt;
            }
        // s_memtime counts at 100 MHz; the shader clock is read back from the device properties
        printf("%-28s waves/SIMD %d: %8.1f memtime ticks per iteration of 32 instructions\n", name, waves / 4, sum / cnt / iters);
    }
}

int main()
{
    double *d_out;
    hipMalloc(&d_out, sizeof(double) * 128);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d_out);
    double h[128];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (h[l] != 100.0 + 16 * (l >> 4) + 5) ++bad;
        if (h[64 + l] != -(100.0 + 16 * (l >> 4) + 15)) ++bad;
    }
    printf("semantics: %d mismatches (lane 0: %g, lane 17: %g, lane 63: %g; negated lane 20: %g)\n", bad, h[0], h[17], h[63], h[64 + 20]);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("shader clock %d kHz; s_memtime ticks are 10 ns\n", clk);
    long long *d_cyc;
    double *d_sink;
    hipMalloc(&d_cyc, sizeof(long long) * 256 * 16);
    hipMalloc(&d_sink, sizeof(double) * 256 * 1024);
    run<0>("dpp fmac, 1 chain", d_cyc, d_sink);
    run<1>("dpp fmac, 2 accumulators", d_cyc, d_sink);
    run<2>("dpp fmac, 4 accumulators", d_cyc, d_sink);
    run<3>("v_fma_f64, 1 chain", d_cyc, d_sink);
    run<4>("complex matvec pattern", d_cyc, d_sink);
    run<4>("matvec, lanes 0..31 only", d_cyc, d_sink, 1);
    run<3>("v_fma_f64 chain, lanes 0..31", d_cyc, d_sink, 1);
    return 0;
}
