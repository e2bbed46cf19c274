// horner_step.hip -- gfx950: what ONE Horner step u <- v + (G u) / k of the vector flow (action_thin.hip) costs, register
// resident, for three ways of laying a 16 x 16 complex matrix-vector product over a wavefront, at 1 / 2 / 4 (/ 8) waves per
// SIMD.  Asked for by VERDICT r3 (#1): "a single wave issues FP64 at half rate" predicts 2 x from a second wave per SIMD and
// the kernel shows 1.12 x.
//
//   V0  round-3 layout: two chains per wave, DPP row = (direction, column half): 32 FMACs, then add, swap16, add, fma,
//       swap16, four DPP rotates (18 dependent instructions between two products)
//   V1  two chains per wave, DPP row = (direction, component of the RESULT): the lane holds a whole row of G (re and im),
//       32 FMACs, 3 adds, fma (twice), ONE swap16
//   V2  one chain per wave, DPP row = (component of the source, component of the result): the lane holds 16 reals,
//       16 FMACs, 3 adds (the last twice), swap16, swap32, add, fma
// Reported: ns per step and wave, and member-steps per microsecond and SIMD (a member = two chains).
//
// build: hipcc -O3 --offload-arch=gfx950:xnack- -o horner_step horner_step.hip ; run: ./horner_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define DEV __device__ __forceinline__

__constant__ double kInv[25] = {0.0,      1.0,      1.0 / 2,  1.0 / 3,  1.0 / 4,  1.0 / 5,  1.0 / 6,  1.0 / 7,  1.0 / 8,
                                1.0 / 9,  1.0 / 10, 1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17,
                                1.0 / 18, 1.0 / 19, 1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23, 1.0 / 24};

DEV void swap16(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
DEV void swap32(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
DEV double rot8_odd_rows(double u)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(u), __double2loint(u), 0x128, 0xA, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(u), __double2hiint(u), 0x128, 0xA, 0xF, false);
    return __hiloint2double(hi, lo);
}

// ---- V0: the round-3 product (copied from action_thin.hip) ----
#define MAC0(J, MR, MI)                                                               \
    "v_fmac_f64_dpp %0, %4, %" #MR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, %5, %" #MR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%5, %" #MI " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, %4, %" #MI " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
DEV void matvec0(double &a0, double &a1, double &b0, double &b1, double xr, double xi, const double (&mr)[8], const double (&mi)[8])
{
    asm("s_nop 1\n" MAC0(0, 6, 14) MAC0(1, 7, 15) MAC0(2, 8, 16) MAC0(3, 9, 17) MAC0(4, 10, 18) MAC0(5, 11, 19) MAC0(6, 12, 20)
            MAC0(7, 13, 21)
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
        : "v"(xr), "v"(xi), "v"(mr[0]), "v"(mr[1]), "v"(mr[2]), "v"(mr[3]), "v"(mr[4]), "v"(mr[5]), "v"(mr[6]), "v"(mr[7]),
          "v"(mi[0]), "v"(mi[1]), "v"(mi[2]), "v"(mi[3]), "v"(mi[4]), "v"(mi[5]), "v"(mi[6]), "v"(mi[7]));
}

// ---- V1: acc += P[j] xr[j] + Q[j] xi[j] over eight columns J0 .. J0 + 7 ----
#define MAC1(J, A, B, P, Q)                                                           \
    "v_fmac_f64_dpp %" #A ", %4, %" #P " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %" #B ", %5, %" #Q " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
DEV void matvec1_lo(double &a0, double &a1, double &a2, double &a3, double xr, double xi, const double *p, const double *q)
{
    asm("s_nop 1\n" MAC1(0, 0, 1, 6, 14) MAC1(1, 2, 3, 7, 15) MAC1(2, 0, 1, 8, 16) MAC1(3, 2, 3, 9, 17) MAC1(4, 0, 1, 10, 18)
            MAC1(5, 2, 3, 11, 19) MAC1(6, 0, 1, 12, 20) MAC1(7, 2, 3, 13, 21)
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(xr), "v"(xi), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(q[0]),
          "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
}
DEV void matvec1_hi(double &a0, double &a1, double &a2, double &a3, double xr, double xi, const double *p, const double *q)
{
    asm(MAC1(8, 0, 1, 6, 14) MAC1(9, 2, 3, 7, 15) MAC1(10, 0, 1, 8, 16) MAC1(11, 2, 3, 9, 17) MAC1(12, 0, 1, 10, 18)
            MAC1(13, 2, 3, 11, 19) MAC1(14, 0, 1, 12, 20) MAC1(15, 2, 3, 13, 21)
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(xr), "v"(xi), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(q[0]),
          "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
}

// ---- V2: acc += P[j] x[j] over sixteen columns, four accumulators ----
#define MAC2(J, A, P) "v_fmac_f64_dpp %" #A ", %4, %" #P " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
DEV void matvec2(double &a0, double &a1, double &a2, double &a3, double x, const double *p)
{
    asm("s_nop 1\n" MAC2(0, 0, 5) MAC2(1, 1, 6) MAC2(2, 2, 7) MAC2(3, 3, 8) MAC2(4, 0, 9) MAC2(5, 1, 10) MAC2(6, 2, 11) MAC2(7, 3, 12)
            MAC2(8, 0, 13) MAC2(9, 1, 14) MAC2(10, 2, 15) MAC2(11, 3, 16) MAC2(12, 0, 17) MAC2(13, 1, 18) MAC2(14, 2, 19)
                MAC2(15, 3, 20)
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(x), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]),
          "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]));
}
// two accumulators (for >= 2 waves per SIMD the issue interval covers the FMA latency)
#define MAC2B(J, A, P) "v_fmac_f64_dpp %" #A ", %2, %" #P " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
DEV void matvec2b(double &a0, double &a1, double x, const double *p)
{
    asm("s_nop 1\n" MAC2B(0, 0, 3) MAC2B(1, 1, 4) MAC2B(2, 0, 5) MAC2B(3, 1, 6) MAC2B(4, 0, 7) MAC2B(5, 1, 8) MAC2B(6, 0, 9)
            MAC2B(7, 1, 10) MAC2B(8, 0, 11) MAC2B(9, 1, 12) MAC2B(10, 0, 13) MAC2B(11, 1, 14) MAC2B(12, 0, 15) MAC2B(13, 1, 16)
                MAC2B(14, 0, 17) MAC2B(15, 1, 18)
        : "+v"(a0), "+v"(a1)
        : "v"(x), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]),
          "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]));
}

template <int V, int WPS>
__global__ __launch_bounds__(256 * WPS) void steps(double *sink, const double *src, int iters, int m)
{
    const int lane = threadIdx.x & 63;
    double P[16], Q[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        P[j] = src[(lane * 16 + j) & 1023] * 1e-2;
        Q[j] = src[(lane * 16 + j + 517) & 1023] * 1e-2;
    }
    const double v0 = src[lane], v1 = src[64 + lane];
    double xr = v0, xi = v1, sel = (lane & 16) ? v1 : v0, x = v0;
    for (int it = 0; it < iters; ++it) {
        for (int kk = m; kk >= 1; --kk) {
            const double inv = kInv[kk];
            if (V == 0) {
                double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
                const double(&mr)[8] = reinterpret_cast<const double(&)[8]>(P[0]);
                const double(&mi)[8] = reinterpret_cast<const double(&)[8]>(P[8]);
                matvec0(a0, a1, b0, b1, xr, xi, mr, mi);
                double yr = a0 + a1, yi = b0 + b1;
                swap16(yr, yi);
                const double mine = fma(yr + yi, inv, sel);
                double part = mine, other = mine;
                swap16(part, other);
                xr = rot8_odd_rows(part);
                xi = rot8_odd_rows(other);
            } else if (V == 1) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                matvec1_lo(a0, a1, a2, a3, xr, xi, P, Q);
                matvec1_hi(a0, a1, a2, a3, xr, xi, P + 8, Q + 8);
                const double y = (a0 + a1) + (a2 + a3);
                double part = fma(y, inv, sel), other = fma(y, inv, sel);
                asm volatile("" : "+v"(part), "+v"(other));  // two registers for the swap
                swap16(part, other);
                xr = part;
                xi = other;
            } else if (V == 2) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                matvec2(a0, a1, a2, a3, x, P);
                const double t0 = a0 + a1, t1 = a2 + a3;
                double a = t0 + t1, b = t0 + t1;
                asm volatile("" : "+v"(a), "+v"(b));
                swap16(a, b);
                swap32(a, b);
                x = fma(a + b, inv, sel);
            } else {
                double a0 = 0.0, a1 = 0.0;
                matvec2b(a0, a1, x, P);
                double a = a0 + a1, b = a0 + a1;
                asm volatile("" : "+v"(a), "+v"(b));
                swap16(a, b);
                swap32(a, b);
                x = fma(a + b, inv, sel);
            }
        }
        sel = (V >= 2) ? x : ((lane & 16) ? xi : xr);
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = xr + xi + x + sel;
}

// dependent chains of the cross-lane primitives, one wave per SIMD: latency per instruction pair
template <int WHAT>
__global__ __launch_bounds__(256) void prim(double *sink, const double *src, int iters)
{
    double a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (WHAT == 0)
                swap16(a, b);
            else if (WHAT == 1)
                swap32(a, b);
            else if (WHAT == 2)
                a = rot8_odd_rows(a);
            else if (WHAT == 3)
                a = fma(a, b, b);
            else {
                double t = a;
                asm volatile("v_mov_b64 %0, %1" : "=v"(a) : "v"(t));
            }
        }
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}

static double g_clock_ghz = 2.4;

template <int V, int WPS>
static void run(const char *name, double *d_sink, const double *d_src)
{
    const int iters = 400, m = 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((steps<V, WPS>), dim3(256), dim3(256 * WPS), 0, 0, d_sink, d_src, iters, m);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    const double ns_step = best * 1e6 / (iters * m);
    const double member_steps_per_wave = V >= 2 ? 0.5 : 1.0;
    printf("%-34s waves/SIMD %d: %7.1f ns per step and wave (~%4.0f cycles at %.2f GHz) | %6.2f member-steps / us / SIMD\n", name, WPS, ns_step,
           ns_step * g_clock_ghz, g_clock_ghz, WPS * member_steps_per_wave / (ns_step * 1e-3));
}

template <int WHAT>
static void run_prim(const char *name, double *d_sink, const double *d_src)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((prim<WHAT>), dim3(256), dim3(256), 0, 0, d_sink, d_src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    printf("%-34s dependent: %6.2f ns each (~%4.1f cycles at %.2f GHz)\n", name, best * 1e6 / (iters * 8.0), best * 1e6 / (iters * 8.0) * g_clock_ghz,
           g_clock_ghz);
}


// throughput of one kind of instruction: eight independent registers per wave, WPS waves per SIMD
template <int WHAT, int WPS>
__global__ __launch_bounds__(256 * WPS) void thr(double *sink, const double *src, int iters)
{
    double r[8], q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        r[u] = src[(threadIdx.x + u) & 1023];
        q[u] = src[(threadIdx.x + u + 77) & 1023];
    }
    const double z = src[5] * 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (WHAT == 0)
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[u]) : "v"(q[u]));
            else if (WHAT == 1)
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[u]) : "v"(q[u]));
            else if (WHAT == 2)
                asm volatile("v_mov_b64 %0, %1" : "=v"(r[u]) : "v"(q[u]));
            else if (WHAT == 3)
                asm volatile("v_mov_b64 %0, 0" : "=v"(r[u]));
            else if (WHAT == 4)
                asm volatile("v_mul_f64 %0, %1, %1" : "=v"(r[u]) : "v"(z));
            else if (WHAT == 5)
            {
                int lo = __double2loint(r[u]), lo2 = __double2loint(q[u]);
                asm volatile("s_nop 0\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
                r[u] = __hiloint2double(__double2hiint(r[u]), lo);
                q[u] = __hiloint2double(__double2hiint(q[u]), lo2);
            }
            else if (WHAT == 6)
                asm volatile("v_fmac_f64_dpp %0, %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(r[u]) : "v"(q[u]));
            else if (WHAT == 7)
            {
                int lo;
                asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(__double2loint(q[u])));
                r[u] = __hiloint2double(__double2hiint(r[u]), lo);
            }
            else if (WHAT == 8)
            {
                int lo = __double2loint(r[u]);
                asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(lo) : "v"(__double2loint(q[u])));
                r[u] = __hiloint2double(__double2hiint(r[u]), lo);
            }
            else if (WHAT == 9)
                asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(r[u]) : "v"(q[u]));
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        acc += r[u] + q[u];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WHAT, int WPS>
static void run_thr(const char *name, double *d_sink, const double *d_src)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((thr<WHAT, WPS>), dim3(256), dim3(256 * WPS), 0, 0, d_sink, d_src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    const double ns = best * 1e6 / (iters * 8.0 * WPS);      // per instruction and SIMD
    printf("%-34s waves/SIMD %d: %6.2f ns per instruction and SIMD (~%4.1f cycles at %.2f GHz)\n", name, WPS, ns, ns * g_clock_ghz, g_clock_ghz);
}

// clock under this kind of load: s_memtime (100 MHz) against a loop of known length (dependent v_fma_f64 = measured below)
__global__ void spin(long long *out, int iters)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    const long long c0 = __builtin_readcyclecounter();
    double a = 1.0;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a));
    const long long c1 = __builtin_readcyclecounter();
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = c1 - c0;
    }
    if (a == 123.0)
        out[2] = 1;
}

int main()
{
    double *d_sink, *d_src;
    hipMalloc(&d_sink, sizeof(double) * 256 * 2048);
    hipMalloc(&d_src, sizeof(double) * 1024);
    std::vector<double> h(1024);
    for (int i = 0; i < 1024; ++i)
        h[i] = 0.3 + 0.001 * ((i * 7919) % 613);
    hipMemcpy(d_src, h.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
    long long *d_t;
    hipMalloc(&d_t, 64);
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, d_t, 200000);
    long long ht[2];
    hipMemcpy(ht, d_t, 16, hipMemcpyDeviceToHost);
    printf("s_memrealtime ticks %lld (10 ns each), s_memtime/readcyclecounter ticks %lld -> ratio %.3f\n", ht[0], ht[1], (double)ht[1] / ht[0]);
    run_prim<3>("v_fma_f64", d_sink, d_src);
    run_prim<0>("v_permlane16_swap x2 (a double)", d_sink, d_src);
    run_prim<1>("v_permlane32_swap x2 (a double)", d_sink, d_src);
    run_prim<2>("v_mov_b32_dpp row_ror x2 (double)", d_sink, d_src);
    run_prim<4>("v_mov_b64", d_sink, d_src);
#define THR(W, NAME) run_thr<W, 1>(NAME, d_sink, d_src); run_thr<W, 2>(NAME, d_sink, d_src); run_thr<W, 4>(NAME, d_sink, d_src);
    THR(0, "thr v_fma_f64") THR(1, "thr v_add_f64") THR(6, "thr v_fmac_f64_dpp") THR(2, "thr v_mov_b64 v,v") THR(3, "thr v_mov_b64 v,0")
    THR(4, "thr v_mul_f64 (zeroing)") THR(5, "thr v_permlane16_swap_b32") THR(7, "thr v_mov_b32") THR(8, "thr v_mov_b32_dpp") THR(9, "thr v_pk_mov_b32")
    run<0, 1>("V0 round-3 (2 chains, col halves)", d_sink, d_src);
    run<0, 2>("V0 round-3 (2 chains, col halves)", d_sink, d_src);
    run<0, 4>("V0 round-3 (2 chains, col halves)", d_sink, d_src);
    run<1, 1>("V1 2 chains, rows = result part", d_sink, d_src);
    run<1, 2>("V1 2 chains, rows = result part", d_sink, d_src);
    run<1, 4>("V1 2 chains, rows = result part", d_sink, d_src);
    run<2, 1>("V2 1 chain/wave, 4 accumulators", d_sink, d_src);
    run<2, 2>("V2 1 chain/wave, 4 accumulators", d_sink, d_src);
    run<2, 4>("V2 1 chain/wave, 4 accumulators", d_sink, d_src);
    run<3, 1>("V3 1 chain/wave, 2 accumulators", d_sink, d_src);
    run<3, 2>("V3 1 chain/wave, 2 accumulators", d_sink, d_src);
    run<3, 4>("V3 1 chain/wave, 2 accumulators", d_sink, d_src);
    return 0;
}
