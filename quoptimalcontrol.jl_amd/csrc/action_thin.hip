// action_thin.hip -- rank-one states, 9 <= n <= 16, member-invariant control operators: the whole evaluation on
// VECTORS.  gfx950, one wavefront per ensemble member.
//
// The chain of sweep_thin.hip needs P_t only applied to a vector:
//     v_{t+1} = exp(G_t) v_t          (src/GRAPE.jl:226 / :245-246 on the factor of X_t = v_t v_t', resp. X_t = v_t)
//     w_t     = exp(G_t)' w_{t+1}     (src/GRAPE.jl:228 / :248-249)
// with G_t = (-i dt)(A_k + sum_c x[c,t] B_c) (src/timeevolution.jl:101-108).  exp(G) v is the truncated Taylor series
// in Horner form, u_m+1 = v, u_j = v + G u_j+1 / j, u_1 = exp(G) v: m matrix-VECTOR products (m = 8 while
// |G| <= 0.08, the bound the matrix Taylor-8 of the expm kernels works under; m from the table below otherwise, the
// generator split into p equal pieces beyond |G| = 1.4) instead of a matrix exponential of 3+ dense products, and no
// propagator is ever formed, stored or read back: C4 (16 x 16 Liouvillian, 1000 slices, 1024 members) moves 33 MB of
// vector records instead of 3 x 4.2 GB of propagators, and does a third of the flops.  Same mathematics as the
// reference's exp(..) * X to rounding (the parity tests hold it to the 1e-10 bar against the oracle's dense evaluation).
//
// Kernels.  The two chains of a member are independent until the gradient and run side by side (forward chain at slice i,
// backward chain at slice N-1-i, one instruction stream).  A product is `v_fmac_f64 ... row_newbcast:j` -- the one DPP control
// the FP64 ALU takes: src0 read from lane j of the 16-lane DPP row, at the plain FMA rate (tools/ubench/dpp_fmac.hip) -- so
// the vector's entries never leave their lanes and there is no reduction tree.  How a 16 x 16 complex matrix is laid over the
// wave decides what a product costs besides its FMACs:
//   action_parts_kernel<false>  shared controls, n <= 16, one member per wave: DPP row = (direction, component of the result),
//                               the lane holds a whole row; 32 FMACs + 7, one v_permlane16_swap (the default of C4)
//   action_parts_kernel<true>   the same, two members per wave: DPP row = one chain, the lane computes both components;
//                               64 FMACs + 8 per two members, no swap (ensembles beyond one member per SIMD)
//   action_thin_kernel          per-member control operators (at most six): DPP row = (direction, column half), 32 FMACs + 18
//   action_thin2_kernel         n = 17..32: a wave per member and direction, DPP row = 16 rows x one 16-column half, 64 FMACs + 18
//   chain_prop_kernel           the propagators of the expm kernel instead of the series (ensembles of 80..239 members, and
//                               on a chunked time axis down to one problem): one product per slice
// One wave per SIMD issues one instruction of ANY kind every ~2.6 ns (tools/ubench/horner_step.hip): at C4's size these
// kernels are counted in instructions, and their slice loops are written for that count.
//
// Inputs: a pre-pass (action_rows_kernel) forms, per slice and control array, the images of Gc_t = (-i dt) sum_c x[c,t] B_c and
// of Gc_t' -- laid out so that a wave-level load reads whole 256-byte runs -- and max(|Gc_t|_1, |Gc_t|_inf); the member's
// [A'_k | A'_k'] stays in registers.  The records v_0..v_N and w_0..w_N go to HBM element-major, and the forms kernels --
// one LANE per slice, fully parallel -- evaluate the bilinear forms of sweep_thin.hip,
//     a = w_t' B_c v_t,  b = v_t' B_c w_t,  s = w_N' v_N,
//     sandwich  g[c,t] = -dt Im(conj(s) a - s b),  F = 1 - (|s|^2 / n)^2;   left mult.  g[c,t] = -/+ 2 dt Im(a conj(s)), F = Re(conj(s)^2)
// with the (member-invariant) B_c read through scalar loads, (value, column) lists, or on the matrix cores.
#include "grape_kernels.hpp"
#include "cmat.hpp"
#include "done_signal.hpp"
#include "tile.hpp"
#include <cstdlib>

namespace grape {

namespace {

// theta_m = (tol (m+1)!)^(1/(m+1)), tol = 3.7e-16, m = 1..24: the largest |G| the degree-m series serves
// (theta_8 = 0.08 = kTheta8: the same truncation the matrix Taylor-8 of the expm kernels accepts)
__constant__ double kActTheta[24] = {2.72029e-08, 1.30452e-05, 0.000306975, 0.00213539, 0.0080215, 0.0211045, 0.0443318, 0.08,
                                     0.129657,    0.194144,    0.273765,    0.368421,   0.477734,  0.601146,  0.737991,  0.887549,
                                     1.04908,     1.22184,     1.40513,     1.59825,    1.80056,   2.01145,   2.23034,   2.45671};
__constant__ double kActInv[25] = {0.0,      1.0,      1.0 / 2,  1.0 / 3,  1.0 / 4,  1.0 / 5,  1.0 / 6,  1.0 / 7,  1.0 / 8,
                                   1.0 / 9,  1.0 / 10, 1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17,
                                   1.0 / 18, 1.0 / 19, 1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23, 1.0 / 24};
constexpr double kActPiece = 1.40513;     // beyond it the generator is split: terms up to e^theta lose digits to cancellation
constexpr int kActMaxPieces = 2047;          // (11 bits of the 16-bit plan entry; |G| > 2800 per slice is not a pulse)

// degree for |G| <= th: the number of table entries below th, plus one (25: beyond the table)
GRAPE_DEV int act_degree(double th)
{
    int lo = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 1)
        if (lo + step <= 24 && th > kActTheta[lo + step - 1])
            lo += step;
    return lo + 1;
}

// y += M x for the lane's 8 complex entries: 4 independent chains of 8 FMACs, x broadcast from lane j of the DPP row.
// The leading s_nop covers the two wait states a DPP read needs behind the VALU write of xr / xi (the compiler's hazard
// recogniser does not look inside inline asm).
#define GRAPE_ACT_MAC(J, MR, MI)                                                      \
    "v_fmac_f64_dpp %0, %4, %" #MR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, %5, %" #MR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%5, %" #MI " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, %4, %" #MI " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
GRAPE_DEV void act_matvec(double &a0, double &a1, double &b0, double &b1, double xr, double xi, const double (&mr)[8],
                          const double (&mi)[8])
{
    asm("s_nop 1\n" GRAPE_ACT_MAC(0, 6, 14) GRAPE_ACT_MAC(1, 7, 15) GRAPE_ACT_MAC(2, 8, 16) GRAPE_ACT_MAC(3, 9, 17)
            GRAPE_ACT_MAC(4, 10, 18) GRAPE_ACT_MAC(5, 11, 19) GRAPE_ACT_MAC(6, 12, 20) GRAPE_ACT_MAC(7, 13, 21)
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
        : "v"(xr), "v"(xi), "v"(mr[0]), "v"(mr[1]), "v"(mr[2]), "v"(mr[3]), "v"(mr[4]), "v"(mr[5]), "v"(mr[6]), "v"(mr[7]),
          "v"(mi[0]), "v"(mi[1]), "v"(mi[2]), "v"(mi[3]), "v"(mi[4]), "v"(mi[5]), "v"(mi[6]), "v"(mi[7]));
}
// the same for the second eight columns of a 16-column half (n = 17..32): lanes 8..15 of the row
GRAPE_DEV void act_matvec_hi(double &a0, double &a1, double &b0, double &b1, double xr, double xi, const double (&mr)[8],
                             const double (&mi)[8])
{
    asm("s_nop 1\n" GRAPE_ACT_MAC(8, 6, 14) GRAPE_ACT_MAC(9, 7, 15) GRAPE_ACT_MAC(10, 8, 16) GRAPE_ACT_MAC(11, 9, 17)
            GRAPE_ACT_MAC(12, 10, 18) GRAPE_ACT_MAC(13, 11, 19) GRAPE_ACT_MAC(14, 12, 20) GRAPE_ACT_MAC(15, 13, 21)
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
        : "v"(xr), "v"(xi), "v"(mr[0]), "v"(mr[1]), "v"(mr[2]), "v"(mr[3]), "v"(mr[4]), "v"(mr[5]), "v"(mr[6]), "v"(mr[7]),
          "v"(mi[0]), "v"(mi[1]), "v"(mi[2]), "v"(mi[3]), "v"(mi[4]), "v"(mi[5]), "v"(mi[6]), "v"(mi[7]));
}
#undef GRAPE_ACT_MAC

// v_permlane16_swap on a double: the odd rows of a trade places with the even rows of b
GRAPE_DEV void swap16(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}

// v_permlane32_swap on a double: lanes 32..63 of a trade places with lanes 0..31 of b
GRAPE_DEV void swap32(double &a, double &b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}

// the plan of every step i of a chain pair (forward slice i, backward slice N-1-i): Taylor degree (low 5 bits) and number
// of pieces from the larger of the two slices' bounds.  Worked out by all lanes at once into the wave's own LDS strip --
// inside the chain the table search is five dependent scalar-memory round trips per slice (measured: 45 % of the kernel
// in s_waitcnt)
// gscale: |s_k| of a member whose control operators are member 0's times s_k (1: the plain sum, bit for bit)
GRAPE_DEV void act_make_plan(unsigned short *s_plan, const double *__restrict__ gn, double an, int N, int forced, int lane,
                             double gscale = 1.0)
{
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = min(i0 + lane, N - 1);
        double th = fma(gscale, fmax(gn[i], gn[N - 1 - i]), an);
        int pieces = 1;
        if (forced >= 0)
            pieces = 1 << min(forced, 10);
        else if (th > kActPiece)
            pieces = (int)fmin(ceil(th / kActPiece), (double)kActMaxPieces);
        if (pieces != 1)
            th /= (double)pieces;
        s_plan[i] = (unsigned short)(min(act_degree(th), 24) | (pieces << 5));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the same for per-member control operators: |Gc|  <=  sum_c |x[c,t]| bn[c]
GRAPE_DEV void act_make_plan_own(unsigned short *s_plan, const double *__restrict__ x, const double *__restrict__ bn, double an,
                                 int K, int N, int forced, int lane)
{
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = min(i0 + lane, N - 1);
        double f = 0.0, b = 0.0;
        for (int c = 0; c < K; ++c) {
            f = fma(fabs(x[c + (size_t)i * K]), bn[c], f);
            b = fma(fabs(x[c + (size_t)(N - 1 - i) * K]), bn[c], b);
        }
        double th = an + (f > b || f != f ? f : b);            // NaN-propagating maximum
        int pieces = 1;
        if (forced >= 0)
            pieces = 1 << min(forced, 10);
        else if (th > kActPiece)
            pieces = (int)fmin(ceil(th / kActPiece), (double)kActMaxPieces);
        if (pieces != 1)
            th /= (double)pieces;
        s_plan[i] = (unsigned short)(min(act_degree(th), 24) | (pieces << 5));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// rows 1 and 3 (h = 1): rotate by 8 lanes inside the row; rows 0 and 2 keep their value
GRAPE_DEV double rot8_odd_rows(double u)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(u), __double2loint(u), 0x128, 0xA, 0xF, false);   // row_ror:8
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(u), __double2hiint(u), 0x128, 0xA, 0xF, false);
    return __hiloint2double(hi, lo);
}

}  // namespace

// grid (N, n_x), 256 threads: element e = NB r + j of Gc_t and of Gc_t'; act_b = per control [B'_c | B'_c'], row-major NB x NB.
// PLANAR (the 16 x 16 kernel below): the images are written as six planes of doubles per slice,
//     [re Gc | im Gc | -im Gc | re Gc' | im Gc' | -im Gc'],
// so that a lane of action_parts_kernel reads its 16 + 16 operands with the signs it multiplies with already in place (negation is exact: the sums below are the reference's, bit for bit).
template <int NB, bool PLANAR>
__global__ __launch_bounds__(256) void action_rows_kernel(const TileParams p)
{
    constexpr int NN = NB * NB;
    __shared__ double s_abs[NN];
    __shared__ double s_sum[2 * NB];
    const int t = blockIdx.x, y = blockIdx.y, K = p.K, N = p.N;
    const double *__restrict__ x = p.x + ((size_t)y * N + t) * K;
    const double2 *__restrict__ Bb = p.act_b_ref ? p.act_b_ref : p.act_b;      // member 0's set of control operators
    double2 *__restrict__ dst = p.act_g + ((size_t)y * N + t) * (PLANAR ? 3 : 2) * NN;
    for (int e = threadIdx.x; e < NN; e += 256) {
        double2 g0, g1;
        {
            const double x0 = x[0];
            const double2 b0 = Bb[e], b1 = Bb[NN + e];
            g0 = make_double2(b0.x * x0, b0.y * x0);              // (0 + B_1 x_1) first, timeevolution.jl:101-108
            g1 = make_double2(b1.x * x0, b1.y * x0);
        }
        for (int c = 1; c < K; ++c) {
            const double xc = x[c];
            const double2 b0 = Bb[(size_t)c * 2 * NN + e], b1 = Bb[(size_t)c * 2 * NN + NN + e];
            g0.x = fma(b0.x, xc, g0.x);
            g0.y = fma(b0.y, xc, g0.y);
            g1.x = fma(b1.x, xc, g1.x);
            g1.y = fma(b1.y, xc, g1.y);
        }
        if (PLANAR) {
            // inside a plane entry (r, j) sits at [j >> 1][r][j & 1]: the 16 lanes of a DPP row read one pair of columns
            // as 256 contiguous bytes (a wave-level load touches 8 cache lines; row-major rows, 128 bytes apart per lane,
            // were 64 lines per instruction and left the kernel waiting for the texture addresser: 2.6 ms instead of 1.4)
            double *__restrict__ pl = reinterpret_cast<double *>(dst);
            const int rr = e / NB, jj = e % NB, q = ((jj >> 1) * NB + rr) * 2 + (jj & 1);
            pl[q] = g0.x;
            pl[NN + q] = g0.y;
            pl[2 * NN + q] = -g0.y;
            pl[3 * NN + q] = g1.x;
            pl[4 * NN + q] = g1.y;
            pl[5 * NN + q] = -g1.y;
        } else {
            // column-major: the 16 lanes of a DPP row of action_thin2_kernel (16 consecutive matrix rows, one column) read 256
            // contiguous bytes (row-major rows, 512 bytes apart per lane, were 64 cache lines per load instruction: the
            // kernel ran at the texture addresser's rate)
            const int rr = e / NB, jj = e % NB;
            dst[jj * NB + rr] = g0;
            dst[NN + jj * NB + rr] = g1;
        }
        s_abs[e] = fabs(g0.x) + fabs(g0.y);
    }
    __syncthreads();
    const int e = threadIdx.x;
    if (e < 2 * NB) {                                             // NB column sums, NB row sums
        double sum = 0.0;
        for (int q = 0; q < NB; ++q)
            sum += e < NB ? s_abs[q * NB + e] : s_abs[(e - NB) * NB + q];
        s_sum[e] = sum;
    }
    __syncthreads();
    if (e == 0) {
        double best = 0.0;
        for (int q = 0; q < 2 * NB; ++q)
            if (!(s_sum[q] <= best))                              // NaN-propagating
                best = s_sum[q];
        p.act_gn[(size_t)y * N + t] = best;
    }
}

// Workgroups of kActWaves members (one wave each, no communication): the launcher sizes the dynamic LDS so that exactly as
// many workgroups fit a compute unit as an even spread needs.  With one-wave workgroups the dispatcher doubled up waves
// on some SIMDs while others idled (1024 members: 1.99 ms, 768 and fewer: 1.33 ms).
// This kernel serves members with their OWN control operators (amplitude-scaled controls of a robustness ensemble, ...), at
// most kActOwnK of them, on the round-3 layout (DPP row = (direction, column half): the lane keeps its half rows of every
// B'_kc in registers, 16 per control, which whole rows would double) and forms the control sum itself, in the reference's
// order; no pre-pass, the norm bound is sum_c |x_c| |B'_kc|.  Shared controls: action_parts_kernel below.
constexpr int kActWaves = 4;
constexpr int kActOwnK = 6;
__global__ __launch_bounds__(64 * kActWaves) void action_thin_kernel(const TileParams p)
{
    const int lane = threadIdx.x & 63, r = lane & 15, d = lane >> 5, h = (lane >> 4) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.x * kActWaves + wave, y = blockIdx.y, N = p.N;
    if (k >= p.E)
        return;
    const size_t kw = (size_t)y * p.E + k;
    const int off = d * 256 + r * 16 + 8 * h;
    const double2 *__restrict__ Ak = p.act_a + (size_t)k * 512 + off;
    const double an = p.act_an[k];
    // this chain's records, element-major: element r of slice t at [r][t] -- action_forms_kernel (lane = slice) reads them coalesced
    double2 *__restrict__ rec = (d ? p.wrec : p.states) + kw * (size_t)(N + 1) * 16 + (size_t)r * (N + 1);
    double ar[8], ai[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double2 a = Ak[j];
        ar[j] = a.x;
        ai[j] = a.y;
    }
    double vr, vi;                                                // the chain's vector, element r (both halves)
    {
        const double2 t2 = p.vecs[(size_t)k * 32 + d * 16 + r];
        vr = t2.x;
        vi = t2.y;
    }
    if (h == 0)
        rec[d ? N : 0] = make_double2(vr, vi);
    const int K = p.K;
    const double *__restrict__ xy = p.x + (size_t)y * N * K;      // this control array, x[c + t K]
    double br[kActOwnK][8], bi[kActOwnK][8], xq[kActOwnK];       // own half rows of B'_kc; the next slice's controls
    {
        const double2 *__restrict__ Bk = p.act_b + (size_t)k * K * 512 + off;
        const double *__restrict__ xs = xy + (size_t)(d ? N - 1 : 0) * K;
#pragma unroll
        for (int c = 0; c < kActOwnK; ++c) {
            xq[c] = c < K ? xs[c] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double2 b = c < K ? Bk[(size_t)c * 512 + j] : make_double2(0.0, 0.0);
                br[c][j] = b.x;
                bi[c][j] = b.y;
            }
        }
    }
    extern __shared__ unsigned short s_plan_all[];
    unsigned short *s_plan = s_plan_all + (size_t)wave * N;       // this wave's own plan: no workgroup barrier anywhere
    act_make_plan_own(s_plan, xy, p.act_bn + (size_t)k * K, an, K, N, p.s_forced, lane);
    // loop state: the vector as the products read it, x[(r + 8h) mod 16] (in the h = 0 rows that IS element r), and the
    // component this row updates (h = 0: real part, h = 1: imaginary part of element r)
    double xr = rot8_odd_rows(vr), xi = rot8_odd_rows(vi);
    double sel = h ? vi : vr;
    unsigned plan = s_plan[0];
    for (int i = 0; i < N; ++i) {
        double mr[8], mi[8];
        const int tn = d ? max(N - 2 - i, 0) : min(i + 1, N - 1);
        {
#pragma unroll
            for (int j = 0; j < 8; ++j) {                         // (0 + B_1 x_1) first, A last
                mr[j] = br[0][j] * xq[0];
                mi[j] = bi[0][j] * xq[0];
            }
#pragma unroll
            for (int c = 1; c < kActOwnK; ++c)
                if (c < K) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        mr[j] = fma(br[c][j], xq[c], mr[j]);
                        mi[j] = fma(bi[c][j], xq[c], mi[j]);
                    }
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mr[j] += ar[j];
                mi[j] += ai[j];
            }
            const double *__restrict__ xs = xy + (size_t)tn * K;
#pragma unroll
            for (int c = 0; c < kActOwnK; ++c)
                if (c < K)
                    xq[c] = xs[c];
        }
        const int m = __builtin_amdgcn_readfirstlane(plan & 31), pieces = __builtin_amdgcn_readfirstlane(plan >> 5);
        plan = s_plan[min(i + 1, N - 1)];
        if (pieces != 1) {                                        // exp(G) = exp(G / p)^p: the pieces see G / p
            const double inv_p = 1.0 / (double)pieces;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mr[j] *= inv_p;
                mi[j] *= inv_p;
            }
        }
        // one Horner step u <- v + (G u) / kk; returns this row's component of the new u before it is shared
        auto step = [&](int kk) -> double {
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
            act_matvec(a0, a1, b0, b1, xr, xi, mr, mi);
            double yr = a0 + a1, yi = b0 + b1;
            swap16(yr, yi);                                       // h = 0 rows: both real halves; h = 1 rows: both imaginary halves
            const double mine = fma(yr + yi, kActInv[kk], sel);
            double part = mine, other = mine;
            swap16(part, other);                                  // part: real part in every row, other: imaginary part
            xr = rot8_odd_rows(part);
            xi = rot8_odd_rows(other);
            return mine;
        };
        for (int piece = 0; piece < pieces; ++piece) {
            for (int kk = m; kk >= 2; --kk)
                (void)step(kk);
            sel = step(1);
        }
        if (h == 0)
            rec[d ? N - 1 - i : i + 1] = make_double2(xr, xi);
    }
}

// ---- round 4: the shared-controls kernel on a layout whose products end in ONE cross-row exchange -------------------
// DPP row rho = 2 d + c (d: direction, c: component of the RESULT: 0 = re, 1 = im) holds in lane r the WHOLE row r of
// M = G_t (d = 0) or G_t' (d = 1) as the two real vectors it multiplies with,
//     c = 0:  P = re M[r][.],  Q = -im M[r][.]      y_re[r] = sum_j P[j] x_re[j] + Q[j] x_im[j]
//     c = 1:  P = im M[r][.],  Q =  re M[r][.]      y_im[r] = sum_j P[j] x_re[j] + Q[j] x_im[j]
// (32 doubles, from the pre-pass's signed planes) and BOTH components of element r of the vector, x_re[r], x_im[r]: lane j
// of a row is the broadcast source of column j for either component, no rotation.  A product is 32 FMACs into two
// accumulators (what a lane issues alone is paced by the SIMD's issue logic, ~7 cycles per FP64 instruction, so two
// chains cover the FMA latency), one add, the Horner update (twice: the swap needs two registers) and ONE
// v_permlane16_swap that hands the re row its new x_im and the im row its new x_re: 39 vector instructions per product
// pair instead of 50 and 4 dependent instructions between two products instead of 18 (tools/ubench/horner_step.hip,
// profiles/r04_horner_step.txt: 117 against 146 ns per step with one wave per SIMD, 91 against 108 with four).
#define GRAPE_ACT_MAC2(J, P, Q)                                                        \
    "v_fmac_f64_dpp %0, %2, %" #P " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, %3, %" #Q " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
GRAPE_DEV void act_matvec_parts(double &a0, double &a1, double xr, double xi, const double (&P)[16], const double (&Q)[16])
{
    // (the two moves are also the two wait states a DPP read needs behind the VALU write of xr / xi: no s_nop)
    asm("v_mov_b64 %0, 0\nv_mov_b64 %1, 0\n" GRAPE_ACT_MAC2(0, 4, 12) GRAPE_ACT_MAC2(1, 5, 13) GRAPE_ACT_MAC2(2, 6, 14) GRAPE_ACT_MAC2(3, 7, 15)
            GRAPE_ACT_MAC2(4, 8, 16) GRAPE_ACT_MAC2(5, 9, 17) GRAPE_ACT_MAC2(6, 10, 18) GRAPE_ACT_MAC2(7, 11, 19)
        : "=&v"(a0), "=&v"(a1)
        : "v"(xr), "v"(xi), "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]), "v"(P[4]), "v"(P[5]), "v"(P[6]), "v"(P[7]), "v"(Q[0]),
          "v"(Q[1]), "v"(Q[2]), "v"(Q[3]), "v"(Q[4]), "v"(Q[5]), "v"(Q[6]), "v"(Q[7]));
    asm(GRAPE_ACT_MAC2(8, 4, 12) GRAPE_ACT_MAC2(9, 5, 13) GRAPE_ACT_MAC2(10, 6, 14) GRAPE_ACT_MAC2(11, 7, 15)
            GRAPE_ACT_MAC2(12, 8, 16) GRAPE_ACT_MAC2(13, 9, 17) GRAPE_ACT_MAC2(14, 10, 18) GRAPE_ACT_MAC2(15, 11, 19)
        : "+v"(a0), "+v"(a1)
        : "v"(xr), "v"(xi), "v"(P[8]), "v"(P[9]), "v"(P[10]), "v"(P[11]), "v"(P[12]), "v"(P[13]), "v"(P[14]), "v"(P[15]),
          "v"(Q[8]), "v"(Q[9]), "v"(Q[10]), "v"(Q[11]), "v"(Q[12]), "v"(Q[13]), "v"(Q[14]), "v"(Q[15]));
}
#undef GRAPE_ACT_MAC2

// WHOLE rows: the lane computes BOTH components of its row's result -- y_re += Mr x_re - Mi x_im, y_im += Mi x_re + Mr x_im
// over eight columns J0 .. J0 + 7, four accumulators.  No cross-row traffic at all: a DPP row is one chain, a wave carries
// four chains (two members).
#define GRAPE_ACT_MAC4(J, R, I)                                                        \
    "v_fmac_f64_dpp %0, %4, %" #R " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"  \
    "v_fmac_f64_dpp %2, %4, %" #I " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"  \
    "v_fmac_f64_dpp %1, -%5, %" #I " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, %5, %" #R " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
GRAPE_DEV void act_matvec_whole(double &a0, double &a1, double &b0, double &b1, double xr, double xi, const double (&Mr)[16],
                                const double (&Mi)[16])
{
    asm("v_mov_b64 %0, 0\nv_mov_b64 %1, 0\nv_mov_b64 %2, 0\nv_mov_b64 %3, 0\n" GRAPE_ACT_MAC4(0, 6, 14) GRAPE_ACT_MAC4(1, 7, 15)
            GRAPE_ACT_MAC4(2, 8, 16) GRAPE_ACT_MAC4(3, 9, 17) GRAPE_ACT_MAC4(4, 10, 18) GRAPE_ACT_MAC4(5, 11, 19)
                GRAPE_ACT_MAC4(6, 12, 20) GRAPE_ACT_MAC4(7, 13, 21)
        : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
        : "v"(xr), "v"(xi), "v"(Mr[0]), "v"(Mr[1]), "v"(Mr[2]), "v"(Mr[3]), "v"(Mr[4]), "v"(Mr[5]), "v"(Mr[6]), "v"(Mr[7]),
          "v"(Mi[0]), "v"(Mi[1]), "v"(Mi[2]), "v"(Mi[3]), "v"(Mi[4]), "v"(Mi[5]), "v"(Mi[6]), "v"(Mi[7]));
    asm(GRAPE_ACT_MAC4(8, 6, 14) GRAPE_ACT_MAC4(9, 7, 15) GRAPE_ACT_MAC4(10, 8, 16) GRAPE_ACT_MAC4(11, 9, 17)
            GRAPE_ACT_MAC4(12, 10, 18) GRAPE_ACT_MAC4(13, 11, 19) GRAPE_ACT_MAC4(14, 12, 20) GRAPE_ACT_MAC4(15, 13, 21)
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
        : "v"(xr), "v"(xi), "v"(Mr[8]), "v"(Mr[9]), "v"(Mr[10]), "v"(Mr[11]), "v"(Mr[12]), "v"(Mr[13]), "v"(Mr[14]), "v"(Mr[15]),
          "v"(Mi[8]), "v"(Mi[9]), "v"(Mi[10]), "v"(Mi[11]), "v"(Mi[12]), "v"(Mi[13]), "v"(Mi[14]), "v"(Mi[15]));
}
#undef GRAPE_ACT_MAC4

// WHOLE = false: one member per wave, DPP row = (direction, component of the result) as described above -- every SIMD has a
//                wave up to four members per compute unit.
// WHOLE = true:  TWO members per wave, DPP row q = one chain (member q >> 1, direction q & 1), the lane computes both
//                components of its row: 64 FMACs + 8 other vector instructions per product of two members (36 per member
//                against 39), half the additions and loads of the G build per member, no swap.  Chosen from
//                8 members per compute unit on (launch_action_nb): below that half the SIMDs would idle.
//                The two members of a wave share the plan (the larger norm bound: never fewer Taylor terms).
template <bool WHOLE>
__global__ __launch_bounds__(64 * kActWaves, 2) void action_parts_kernel(const TileParams p)
{
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int d = WHOLE ? q & 1 : q >> 1, c = WHOLE ? 0 : q & 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * kActWaves + wave, y = blockIdx.y, N = p.N;
    if ((WHOLE ? 2 * unit : unit) >= p.E)
        return;
    // (an odd ensemble's last wave: its second pair of rows repeats the last member -- the same values to the same addresses)
    const int k = WHOLE ? min(2 * unit + (q >> 1), p.E - 1) : unit;
    const size_t kw = (size_t)y * p.E + k;
    // the pre-pass's planes of slice t: [re | im | -im] of Gc_t (d = 0) or Gc_t' (d = 1), 256 doubles each; entry (r, j) of a
    // plane sits at [j >> 1][r][j & 1]
    const double2 *__restrict__ Gy = p.act_g + (size_t)y * N * 768 + d * 384 + r;
    const int offP = (!WHOLE && c) ? 128 : 0, offQ = WHOLE ? 128 : c ? 0 : 256;
    const double *__restrict__ gn = p.act_gn + (size_t)y * N;
    double an = p.act_an[k];
    const double sk = p.ctrl_scale ? p.ctrl_scale[k] : 1.0;       // B_k = s_k B_0: G = A'_k + s_k Gc_t
    double sk_abs = fabs(sk);
    if (WHOLE) {                                                  // the wave's two members share the plan: the larger bound
        const double other = __shfl_xor(an, 32, 64), others = __shfl_xor(sk_abs, 32, 64);
        an = (an != an || other != other) ? an + other : fmax(an, other);      // (a NaN bound stays a NaN)
        sk_abs = (sk_abs != sk_abs || others != others) ? sk_abs + others : fmax(sk_abs, others);
    }
    double2 *__restrict__ rec = (d ? p.wrec : p.states) + kw * (size_t)(N + 1) * 16 + (size_t)r * (N + 1);
    double aP[16], aQ[16];                                        // the member's A'_k (A'_k'), arranged and signed like P, Q
    {
        const double2 *__restrict__ Ak = p.act_a + (size_t)k * 512 + d * 256 + r * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double2 a = Ak[j];
            aP[j] = (!WHOLE && c) ? a.y : a.x;
            aQ[j] = WHOLE ? a.y : c ? a.x : -a.y;
        }
    }
    double xr, xi;                                                // element r of the chain's vector, both components
    {
        const double2 t2 = p.vecs[(size_t)k * 32 + d * 16 + r];
        xr = t2.x;
        xi = t2.y;
    }
    rec[d ? N : 0] = make_double2(xr, xi);                        // (both rows of a chain: the same value to the same address --
                                                                  //  an unconditional store keeps the compiler's vmcnt exact)
    extern __shared__ unsigned short s_plan_all[];
    unsigned short *s_plan = s_plan_all + (size_t)wave * N;       // this wave's own plan: no workgroup barrier anywhere
    act_make_plan(s_plan, gn, an, N, p.s_forced, lane, sk_abs);
    double sel = c ? xi : xr, sel2 = xi;                          // the component(s) of v this row updates
    // Two register sets take turns: one holds G_t = Gc_t + A' (the products' operands), the other receives the planes of
    // the next slice at the top of slice t and has A' added in place (no third set).  One wave per SIMD issues
    // ONE instruction of any kind every ~2.6 ns, scalar ones included (tools/ubench/horner_step.hip), so the hot path is
    // counted in instructions: the products of a slice are unrolled (degrees up to 8: a compare and a branch per step,
    // no loop counter, no table load, 1/kk a literal), 39 vector instructions per product of which 32 FMACs.
    // Per-lane pointers walk the pulse (forward chain up, backward chain down): the planes of the next slice and the
    // record slot cost an addition each per slice, no index arithmetic, no clamps -- the chain prefetches ONE slice past
    // its end of the pulse, which the host layer pads (the values are never used).
    const double2 *gpP = Gy + (size_t)(d ? N - 1 : 0) * 768 + offP, *gpQ = Gy + (size_t)(d ? N - 1 : 0) * 768 + offQ;
    const long gstep = d ? -768 : 768;
    double2 *recp = rec + (d ? N - 1 : 1);
    const long rstep = d ? -1 : 1;
    auto fetch = [&](double (&nP)[16], double (&nQ)[16]) {
#if defined(GRAPE_ACT_ABL) && (GRAPE_ACT_ABL & 1)                  // timing ablation (wrong results): no plane loads
        if (N > 0 && gstep != 0 && lane >= 0)
            return;
#endif
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double2 a = gpP[16 * j], b = gpQ[16 * j];
            nP[2 * j] = a.x;
            nP[2 * j + 1] = a.y;
            nQ[2 * j] = b.x;
            nQ[2 * j + 1] = b.y;
        }
        gpP += gstep;
        gpQ += gstep;
    };
    auto build = [&](double (&nP)[16], double (&nQ)[16]) {        // G = Gc + A' (A last, timeevolution.jl:108)
        // every plane load was issued a whole slice ago and only the slice's record store is younger: ONE wait (the
        // compiler, which sees the builtin, would otherwise place one in front of every second addition)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0f71);                       // vmcnt(1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            nP[j] = fma(sk, nP[j], aP[j]);                        // (s_k = 1: the plain sum, bit for bit)
            nQ[j] = fma(sk, nQ[j], aQ[j]);
        }
    };
    unsigned plan;
    // one Horner step u <- v + (G u) inv, inv = 1 / kk; `last`: the result is the chain's next vector v
    auto step = [&](const double (&P)[16], const double (&Q)[16], double inv, bool last) {
        if (WHOLE) {
            double a0, a1, b0, b1;
            act_matvec_whole(a0, a1, b0, b1, xr, xi, P, Q);
            xr = fma(a0 + a1, inv, sel);
            xi = fma(b0 + b1, inv, sel2);
            if (last) {
                sel = xr;
                sel2 = xi;
            }
        } else {
            double a0, a1;
            act_matvec_parts(a0, a1, xr, xi, P, Q);
            const double ysum = a0 + a1;
            double mine = fma(ysum, inv, sel), other = fma(ysum, inv, sel);
            asm volatile("" : "+v"(mine), "+v"(other));           // two registers: the swap overwrites both
            if (last)
                sel = mine;
            swap16(mine, other);                                  // mine: re of element r in both rows, other: im
            xr = mine;
            xi = other;
        }
    };
    auto slice = [&](double (&P)[16], double (&Q)[16], double (&nP)[16], double (&nQ)[16], const unsigned short *next_plan) {
        fetch(nP, nQ);
        const int m = __builtin_amdgcn_readfirstlane(plan & 31), pieces = __builtin_amdgcn_readfirstlane(plan >> 5);
        plan = *next_plan;
        if (pieces == 1 && m <= 8) {                              // (every slice of a pulse whose |G_t| stays below 0.08)
            if (m >= 8) step(P, Q, 1.0 / 8, false);
            if (m >= 7) step(P, Q, 1.0 / 7, false);
            if (m >= 6) step(P, Q, 1.0 / 6, false);
            if (m >= 5) step(P, Q, 1.0 / 5, false);
            if (m >= 4) step(P, Q, 1.0 / 4, false);
            if (m >= 3) step(P, Q, 1.0 / 3, false);
            if (m >= 2) step(P, Q, 1.0 / 2, false);
            step(P, Q, 1.0, true);
        } else {                                                  // degrees beyond 8, or the generator in pieces:
            const double inv_p = 1.0 / (double)pieces;            // exp(G) = exp(G / p)^p, the 1 / p rides on the Horner factor
            for (int piece = 0; piece < pieces; ++piece) {
                for (int kk = m; kk >= 2; --kk)
                    step(P, Q, kActInv[kk] * inv_p, false);
                step(P, Q, inv_p, true);
            }
        }
#if !(defined(GRAPE_ACT_ABL) && (GRAPE_ACT_ABL & 2))              // timing ablation: no record stores
        *recp = make_double2(xr, xi);
#endif
        recp += rstep;
    };
    double P0[16], Q0[16], P1[16], Q1[16];
    fetch(P0, Q0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    build(P0, Q0);
    // (the plan of slice t + 1 is read at the top of slice t: read where it is used, every slice waited ~50 ns for LDS)
    plan = s_plan[0];
    int i = 0;
    for (; i + 2 <= N; i += 2) {
        slice(P0, Q0, P1, Q1, s_plan + i + 1);
        build(P1, Q1);
        slice(P1, Q1, P0, Q0, s_plan + min(i + 2, N - 1));
        build(P0, Q0);
    }
    if (i < N)
        slice(P0, Q0, P1, Q1, s_plan + i);
}

// n = 17..32: one wavefront per member AND direction (d = wave & 1; a workgroup = two members).  DPP row rho = 2 Rb + H holds in
// lane l the 16 complex entries M[16 Rb + l][16 H .. 16 H + 15] (32 VGPRs) and x[16 H + l]: 64 FMACs per product.  The
// column halves meet through one v_permlane16_swap of (re, im) as above -- row rho then holds the row's component
// R_rho = (re b0, im b0, re b1, im b1) of the 16-blocks b0, b1 of the new vector -- and TWO more swaps hand every row the
// block its columns need: v_permlane32_swap of (R, R) gives (R0 R1 R0 R1), (R2 R3 R2 R3); v_permlane16_swap of those
// gives (R0 R2 R0 R2) = re of block H and (R1 R3 R1 R3) = im of block H.  82 vector instructions per product.
// (Measured alternatives: both chains in one wave with whole rows per lane -- 148 instructions per product for the pair,
// no swaps, but 256 + 158 registers and one wave per SIMD: 12.6 ms against 9.0 ms for 1024 members of 2000 slices;
// blocks handed round by four swaps and selects after an all-gather: 100 instructions, 9.0 ms.)
// Round 4: the slice loop of action_parts_kernel -- images read as whole 256-byte runs (column-major, see action_rows_kernel),
// two register sets taking turns with A' added in place, per-lane pointers walking the pulse, every row storing the record of
// the element it holds (two rows the same value to the same address: an unconditional store), one vmcnt wait per slice,
// the products of a slice unrolled for degrees up to 8.
__global__ __launch_bounds__(64 * kActWaves, 2) void action_thin2_kernel(const TileParams p)
{
    const int lane = threadIdx.x & 63, l = lane & 15, Rb = lane >> 5, H = (lane >> 4) & 1, el = 16 * Rb + l;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), d = wave & 1;
    const int k = blockIdx.x * (kActWaves / 2) + (wave >> 1), y = blockIdx.y, N = p.N;
    if (k >= p.E)
        return;
    const size_t kw = (size_t)y * p.E + k;
    // entry (row, col) of an image at [col][row]: the lane's 16 entries are 32 double2 apart, the row's 16 lanes contiguous
    const int off = d * 1024 + (16 * H) * 32 + el;
    const double *__restrict__ gn = p.act_gn + (size_t)y * N;
    const double an = p.act_an[k];
    const double sk = p.ctrl_scale ? p.ctrl_scale[k] : 1.0;       // B_k = s_k B_0: G = A'_k + s_k Gc_t
    double2 a[16];
    {
        const double2 *__restrict__ Ak = p.act_a + (size_t)k * 2048 + d * 1024 + el * 32 + 16 * H;      // (row-major upload)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            a[j] = Ak[j];
    }
    // every row holds, as x, element 16 H + l of the vector (rows 0 and 2 the same ones, rows 1 and 3): it records that element
    double2 *recp = (d ? p.wrec : p.states) + kw * (size_t)(N + 1) * 32 + (size_t)(16 * H + l) * (N + 1);
    double xr, xi, sel;                                           // x[16 H + l]; this row's component of element el
    {
        const double2 *__restrict__ v0 = p.vecs + (size_t)k * 64 + d * 32;
        const double2 own = v0[el], col = v0[16 * H + l];
        xr = col.x;
        xi = col.y;
        sel = H ? own.y : own.x;
        recp[d ? N : 0] = col;
    }
    recp += d ? N - 1 : 1;
    const long rstep = d ? -1 : 1, gstep = d ? -2048 : 2048;
    const double2 *gp = p.act_g + (size_t)y * N * 2048 + (size_t)(d ? N - 1 : 0) * 2048 + off;
    extern __shared__ unsigned short s_plan_all[];
    unsigned short *s_plan = s_plan_all + (size_t)wave * N;
    act_make_plan(s_plan, gn, an, N, p.s_forced, lane, fabs(sk)); // (both waves of a member: the same plan)
    auto fetch = [&](double2 (&g)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            g[j] = gp[32 * j];
        gp += gstep;                                              // (one slice past the end of the pulse at the last step: padded)
    };
    auto build = [&](double2 (&g)[16]) {                          // G = Gc + A' (A last, timeevolution.jl:108)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0f71);                       // vmcnt(1): every load is a slice old, only the record store is younger
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            g[j].x = fma(sk, g[j].x, a[j].x);                     // (s_k = 1: the plain sum, bit for bit)
            g[j].y = fma(sk, g[j].y, a[j].y);
        }
    };
    auto step = [&](const double2 (&g)[16], double inv, bool last) {
        double mr[8], mi[8], nr[8], ni[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mr[j] = g[j].x;
            mi[j] = g[j].y;
            nr[j] = g[8 + j].x;
            ni[j] = g[8 + j].y;
        }
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        act_matvec(a0, a1, b0, b1, xr, xi, mr, mi);
        act_matvec_hi(a0, a1, b0, b1, xr, xi, nr, ni);
        double yr = a0 + a1, yi = b0 + b1;
        swap16(yr, yi);                                           // H = 0 rows: both real halves; H = 1 rows: both imaginary halves
        const double mine = fma(yr + yi, inv, sel);               // row rho: R_rho = (re b0, im b0, re b1, im b1)
        if (last)
            sel = mine;
        xr = mine;
        xi = mine;
        swap32(xr, xi);                                           // (R0 R1 R0 R1), (R2 R3 R2 R3)
        swap16(xr, xi);                                           // (R0 R2 R0 R2) = re of block H, (R1 R3 R1 R3) = im of block H
    };
    unsigned plan;
    auto slice = [&](const double2 (&g)[16], double2 (&gnext)[16], const unsigned short *next_plan) {
        fetch(gnext);
        const int m = __builtin_amdgcn_readfirstlane(plan & 31), pieces = __builtin_amdgcn_readfirstlane(plan >> 5);
        plan = *next_plan;
        if (pieces == 1 && m <= 8) {
            if (m >= 8) step(g, 1.0 / 8, false);
            if (m >= 7) step(g, 1.0 / 7, false);
            if (m >= 6) step(g, 1.0 / 6, false);
            if (m >= 5) step(g, 1.0 / 5, false);
            if (m >= 4) step(g, 1.0 / 4, false);
            if (m >= 3) step(g, 1.0 / 3, false);
            if (m >= 2) step(g, 1.0 / 2, false);
            step(g, 1.0, true);
        } else {                                                  // degrees beyond 8, or the generator in pieces:
            const double inv_p = 1.0 / (double)pieces;            // exp(G) = exp(G / p)^p, the 1 / p rides on the Horner factor
            for (int piece = 0; piece < pieces; ++piece) {
                for (int kk = m; kk >= 2; --kk)
                    step(g, kActInv[kk] * inv_p, false);
                step(g, inv_p, true);
            }
        }
        *recp = make_double2(xr, xi);
        recp += rstep;
    };
    double2 g0[16], g1[16];
    fetch(g0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    build(g0);
    plan = s_plan[0];                                            // (the next slice's plan is read a slice ahead)
    int i = 0;
    for (; i + 2 <= N; i += 2) {
        slice(g0, g1, s_plan + i + 1);
        build(g1);
        slice(g1, g0, s_plan + min(i + 2, N - 1));
        build(g0);
    }
    if (i < N)
        slice(g0, g1, s_plan + i);
}

// The same two chains with the PROPAGATORS of the expm kernel (p.thin == 2: ensembles too small for the Taylor flow above --
// a member's chains take N x 8 dependent products there whatever the ensemble size -- or with many per-member controls):
// one matrix-vector product per slice.  The expm kernel stores P_t AND P_t^T (D-layout dumps, untransposed for every t): a
// lane's operands are then entries (8h + j, r) of a dump for either chain -- 16 entries apart per lane, consecutive across
// the 16 lanes of a row, so every load instruction reads whole 256-byte runs.  (Rows read per lane from the P_t dump alone
// -- 128 contiguous bytes per lane, 32 cache lines per instruction -- ran at the texture addresser's rate: 1.1 us per slice
// with four members per compute unit.)  sweep_thin.hip's chain spends ~100 vector
// instructions per product on cross-lane sums; this one 50, with both chains in one wave.
__global__ __launch_bounds__(64) void chain_prop_kernel(const TileParams p)
{
    const int lane = threadIdx.x & 63, r = lane & 15, d = lane >> 5, h = (lane >> 4) & 1;
    const int wave = 0;
    const int k = blockIdx.x, y = blockIdx.y, N = p.N;
    if (k >= p.E)
        return;
    const size_t kw = (size_t)y * p.E + k;
    // Small ensembles (grid.z = chunk, p.tp_chunks > 1): this workgroup walks the slices [lo, hi) only -- forward from v_lo,
    // backward from w_hi, both handed over by the scan launch (this kernel again, with the chunk products in the place of
    // the propagators: launch_chain_prop) in p.tp_vec: [direction][control array][unit][element][chunk boundary 0 .. C]
    const int C = p.tp_chunks > 1 ? p.tp_chunks : 1, ch = blockIdx.z;
    const int lo = C > 1 ? ch * p.tp_S : 0, hi = C > 1 ? min(N, lo + p.tp_S) : N, len = hi - lo;
    // entry (row, col) of a dump sits at 64 (row >> 2) + 16 (row & 3) + col.  Forward: rows of P_t = columns of the P_t^T dump,
    // backward: rows of P_t' = conjugated columns of the P_t dump: entry (8h + j, r) of either, consecutive across the lanes r
    const double2 *__restrict__ Pk = (d ? p.props : p.props_t) + kw * (size_t)N * 256 + 128 * h + r;
    const double sgn = d ? -1.0 : 1.0;                            // backward: the conjugate
    double2 *__restrict__ rec = (d ? p.wrec : p.states) + kw * (size_t)(N + 1) * 16 + (size_t)r * (N + 1);
    double2 *rec0 = rec + (d ? N : 0);                            // where the chain's first vector is recorded
    double vr, vi;
    if (C > 1) {
        double2 *start = p.tp_vec + (((size_t)d * gridDim.y * p.E + kw) * 16 + r) * (size_t)(C + 1) + ch + d;
        const double2 t2 = *start;
        vr = t2.x;
        vi = t2.y;
        if (d ? ch != C - 1 : ch != 0)                            // v_lo / w_hi are recorded by the neighbouring chunk's last step:
            rec0 = start;                                         // the prologue's stores rewrite the hand-over entry with itself
    } else {
        const double2 t2 = p.vecs[(size_t)k * 32 + d * 16 + r];
        vr = t2.x;
        vi = t2.y;
    }
    *rec0 = make_double2(vr, vi);
    // The dumps come from HBM (E N x 4 KB each, nothing of it cached) at ~1 us latency and the chain consumes a slice every
    // ~0.3 us.  A register ring (32 registers per slice) holds four slices at most: 0.875 us per slice measured, 0.325 with
    // the operands cached.  So the ring lives in LDS and is filled by LDS-DMA (global_load_lds_dwordx4: one instruction =
    // 64 lanes x 16 B, lane-linear -- exactly one operand index j of a slice): kRing slices of 8 KB in flight per wave, no
    // registers, and the wave reads its own bytes back with ds_read_b128.  The DMA loads are inline asm, i.e. outside the
    // compiler's s_waitcnt bookkeeping: per step this wave issues 8 of them and ONE record store, in that order, and waits
    // for "all but the 9 (kRing - 1) youngest" before it reads a slot.
    typedef double d2x __attribute__((ext_vector_type(2)));
    constexpr int kRing = 7;
    extern __shared__ double2 s_ring_all[];
    const unsigned ring0 = (unsigned)(size_t)s_ring_all + (unsigned)wave * (kRing * 8192);
    auto dma = [&](int slot, int i) {                             // slice of step i -> ring slot (clamped: always eight loads)
        const int ic = min(i, len - 1);
        // entry j of the lane: 256 j bytes further on.  The instruction offset moves the LDS address too: M0 advances by 1024 - 256
        const double2 *src = Pk + (size_t)(d ? hi - 1 - ic : lo + ic) * 256;
        const unsigned dst = ring0 + (unsigned)slot * 8192u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, 0x300\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:256\n\t"
                     "s_add_u32 m0, m0, 0x300\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:512\n\t"
                     "s_add_u32 m0, m0, 0x300\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:768\n\t"
                     "s_add_u32 m0, m0, 0x300\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "s_add_u32 m0, m0, 0x300\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1280\n\t"
                     "s_add_u32 m0, m0, 0x300\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1536\n\t"
                     "s_add_u32 m0, m0, 0x300\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1792\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(dst)
                     : "memory", "scc");
    };
    // prologue: the record store of v_0 / w_N above is this wave's only older vector-memory operation
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
        dma(u, u);
        if (u + 1 < kRing)                                        // keep the "8 loads, 1 store" rhythm of the steady state
            *rec0 = make_double2(vr, vi);
    }
    double xr = rot8_odd_rows(vr), xi = rot8_odd_rows(vi);
    auto step = [&](int slot, int i) {
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(9 * (kRing - 1)) : "memory");
        const d2x *mine = reinterpret_cast<const d2x *>(s_ring_all) + ((size_t)wave * kRing + slot) * 512 + lane;
        double mr[8], mi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const d2x v = mine[j * 64];
            mr[j] = v[0];
            mi[j] = sgn * v[1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot is read: it may be refilled
        dma(slot, i + kRing);
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        act_matvec(a0, a1, b0, b1, xr, xi, mr, mi);
        double yr = a0 + a1, yi = b0 + b1;
        swap16(yr, yi);                                           // h = 0 rows: both real halves; h = 1 rows: both imaginary halves
        double part = yr + yi, other = part;
        swap16(part, other);                                      // part: real part in every row, other: imaginary part
        rec[d ? hi - 1 - i : lo + i + 1] = make_double2(part, other);   // (element r, from both rows: the same value)
        xr = rot8_odd_rows(part);
        xi = rot8_odd_rows(other);
    };
    int i = 0;
    for (; i + kRing <= len; i += kRing) {
        step(0, i);
        step(1, i + 1);
        step(2, i + 2);
        step(3, i + 3);
        step(4, i + 4);
        step(5, i + 5);
        step(6, i + 6);
    }
    static_assert(kRing == 7, "unrolled by hand");
    if (i + 0 < len) step(0, i);
    if (i + 1 < len) step(1, i + 1);
    if (i + 2 < len) step(2, i + 2);
    if (i + 3 < len) step(3, i + 3);
    if (i + 4 < len) step(4, i + 4);
    if (i + 5 < len) step(5, i + 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // DMA loads still in flight target this wave's LDS
}

// grid (ceil(N / 64), E, n_x): lane = slice.  act_bf = the K control operators B_c, row-major, zero padded to NB x NB
template <int SAND, bool HERMB, int NB>
__global__ __launch_bounds__(64) void action_forms_kernel(const TileParams p)
{
    const int lane = threadIdx.x, k = blockIdx.y, y = blockIdx.z, K = p.K, N = p.N;
    const int t = blockIdx.x * 64 + lane, tc = min(t, N - 1);
    const size_t kw = (size_t)y * p.E + k;
    const double2 *__restrict__ V = p.states + kw * (size_t)(N + 1) * NB;
    const double2 *__restrict__ W = p.wrec + kw * (size_t)(N + 1) * NB;
    double vr[NB], vi[NB], wr[NB], wi[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const double2 a = V[(size_t)i * (N + 1) + tc], b = W[(size_t)i * (N + 1) + tc];
        vr[i] = a.x;
        vi[i] = a.y;
        wr[i] = b.x;
        wi[i] = b.y;
    }
    double s_re = 0.0, s_im = 0.0;                                // s = w_N' v_N (uniform)
    {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const double2 a = V[(size_t)i * (N + 1) + N], b = W[(size_t)i * (N + 1) + N];
            s_re = fma(b.x, a.x, fma(b.y, a.y, s_re));            // conj(w) v
            s_im = fma(b.x, a.y, fma(-b.y, a.x, s_im));
        }
    }
    const double gs = SAND ? -p.dt * (HERMB ? 2.0 : 1.0) : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    double *__restrict__ out_member = p.member_out + ((size_t)y * p.E_members + k) * ((size_t)K * N + 1);
    for (int c = 0; c < K; ++c) {
        // (constant address space: the operators are read through the scalar cache, a row per wait)
        const __attribute__((address_space(4))) unsigned long long *Bc =
            (const __attribute__((address_space(4))) unsigned long long *)(uintptr_t)(p.act_bf + ((size_t)(p.act_shared ? 0 : k) * K + c) * NB * NB);
        double a_r = 0.0, a_i = 0.0, b_r = 0.0, b_i = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double ur = 0.0, ui = 0.0, xr = 0.0, xi = 0.0;
            unsigned long long q[32];                             // 16 entries of row i: four s_load_dwordx16, one wait
#pragma unroll
            for (int jb = 0; jb < NB; jb += 16) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                q[j] = Bc[2 * (NB * i + jb) + j];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned long long qx = q[2 * j], qy = q[2 * j + 1];
                const double2 b = make_double2(__longlong_as_double((long long)qx), __longlong_as_double((long long)qy));
                ur = fma(b.x, vr[jb + j], ur);                    // (B v)[i]
                ur = fma(-b.y, vi[jb + j], ur);
                ui = fma(b.x, vi[jb + j], ui);
                ui = fma(b.y, vr[jb + j], ui);
                if (SAND && !HERMB) {
                    xr = fma(b.x, wr[jb + j], xr);                // (B w)[i]
                    xr = fma(-b.y, wi[jb + j], xr);
                    xi = fma(b.x, wi[jb + j], xi);
                    xi = fma(b.y, wr[jb + j], xi);
                }
            }
            }
            a_r = fma(wr[i], ur, fma(wi[i], ui, a_r));            // conj(w[i]) (B v)[i]
            a_i = fma(wr[i], ui, fma(-wi[i], ur, a_i));
            if (SAND && !HERMB) {
                b_r = fma(vr[i], xr, fma(vi[i], xi, b_r));        // conj(v[i]) (B w)[i]
                b_i = fma(vr[i], xi, fma(-vi[i], xr, b_i));
            }
        }
        double val = s_re * a_i - s_im * a_r;                     // Im(conj(s) a)
        if (SAND && !HERMB)
            val -= s_re * b_i + s_im * b_r;                       // - Im(s b)
        if (t < N) {
            out_member[c + (size_t)t * K] = gs * val;
            fold_store(p, c + (size_t)t * K, gs * val);
        }
    }
    if (blockIdx.x == 0 && lane == 0) {
        double Fk;
        if (SAND) {
            const double z = (s_re * s_re + s_im * s_im) / (double)p.n;
            Fk = 1.0 - z * z;
        } else {
            Fk = s_re * s_re - s_im * s_im;
        }
        out_member[(size_t)K * N] = Fk;
        fold_store(p, (size_t)K * N, Fk);
    }
    if (threadIdx.x == 0)
        fold_publish(p);
}

// DENSE control operators, n <= 16, on the matrix cores: the records of 16 slices ARE a 16 x 16 matrix V (element-major:
// row = element, column = slice), so (B_c v_t) for 16 slices is one complex 16 x 16 product -- 16 v_mfma_f64_16x16x4 with
// B_c as the A operand (lane l: B_c[l & 15][4 kb + (l >> 4)]) and the records as the B operand (lane l: element
// 4 kb + (l >> 4) of slice l & 15, loaded as they lie) -- instead of 1024 FMAs per lane; the result's D layout puts
// rows 4 r + (l >> 4) of column l & 15 in the lane, which is where the lane's w entries are, so conj(w_t)' (B_c v_t) is four
// complex multiply-adds per lane and two cross-lane additions.  A wavefront serves 64 slices as four such tiles.
template <int SAND, bool HERMB>
__global__ __launch_bounds__(64) void action_forms_mfma_kernel(const TileParams p)
{
    constexpr bool NEEDB = SAND && !HERMB;
    const int lane = threadIdx.x, k = blockIdx.y, y = blockIdx.z, K = p.K, N = p.N;
    const int col = lane & 15, g = lane >> 4, t0 = blockIdx.x * 64;
    const size_t kw = (size_t)y * p.E + k;
    const double2 *__restrict__ V = p.states + kw * (size_t)(N + 1) * 16;
    const double2 *__restrict__ W = p.wrec + kw * (size_t)(N + 1) * 16;
    double vr[4][4], vi[4][4], wr[4][4], wi[4][4];                // [tile][kb]: element 4 kb + g at slice t0 + 16 tile + col
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int tc = min(t0 + 16 * tile + col, N - 1);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double2 a = V[(size_t)(4 * kb + g) * (N + 1) + tc], b = W[(size_t)(4 * kb + g) * (N + 1) + tc];
            vr[tile][kb] = a.x;
            vi[tile][kb] = a.y;
            wr[tile][kb] = b.x;
            wi[tile][kb] = b.y;
        }
    }
    double s_re = 0.0, s_im = 0.0;                                // s = w_N' v_N (uniform)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double2 a = V[(size_t)i * (N + 1) + N], b = W[(size_t)i * (N + 1) + N];
        s_re = fma(b.x, a.x, fma(b.y, a.y, s_re));
        s_im = fma(b.x, a.y, fma(-b.y, a.x, s_im));
    }
    const double gs = SAND ? -p.dt * (HERMB ? 2.0 : 1.0) : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    double *__restrict__ out_member = p.member_out + ((size_t)y * p.E_members + k) * ((size_t)K * N + 1);
    for (int c = 0; c < K; ++c) {
        const double2 *__restrict__ Bc = p.act_bf + ((size_t)(p.act_shared ? 0 : k) * K + c) * 256;
        double br[4], bi[4], nbi[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double2 b = Bc[col * 16 + 4 * kb + g];
            br[kb] = b.x;
            bi[kb] = b.y;
            nbi[kb] = -b.y;
        }
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            d4 ur = {0.0, 0.0, 0.0, 0.0}, ui = ur, zr = ur, zi = ur;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {                      // U = B_c V (and Z = B_c W)
                ur = __builtin_amdgcn_mfma_f64_16x16x4f64(br[kb], vr[tile][kb], ur, 0, 0, 0);
                ui = __builtin_amdgcn_mfma_f64_16x16x4f64(br[kb], vi[tile][kb], ui, 0, 0, 0);
                if (NEEDB) {
                    zr = __builtin_amdgcn_mfma_f64_16x16x4f64(br[kb], wr[tile][kb], zr, 0, 0, 0);
                    zi = __builtin_amdgcn_mfma_f64_16x16x4f64(br[kb], wi[tile][kb], zi, 0, 0, 0);
                }
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                ur = __builtin_amdgcn_mfma_f64_16x16x4f64(nbi[kb], vi[tile][kb], ur, 0, 0, 0);
                ui = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[kb], vr[tile][kb], ui, 0, 0, 0);
                if (NEEDB) {
                    zr = __builtin_amdgcn_mfma_f64_16x16x4f64(nbi[kb], wi[tile][kb], zr, 0, 0, 0);
                    zi = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[kb], wr[tile][kb], zi, 0, 0, 0);
                }
            }
            double a_r = 0.0, a_i = 0.0, b_r = 0.0, b_i = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                         // rows 4 r + g of the lane's column
                a_r = fma(wr[tile][r], ur[r], fma(wi[tile][r], ui[r], a_r));      // conj(w[i]) (B v)[i]
                a_i = fma(wr[tile][r], ui[r], fma(-wi[tile][r], ur[r], a_i));
                if (NEEDB) {
                    b_r = fma(vr[tile][r], zr[r], fma(vi[tile][r], zi[r], b_r));  // conj(v[i]) (B w)[i]
                    b_i = fma(vr[tile][r], zi[r], fma(-vi[tile][r], zr[r], b_i));
                }
            }
            a_r += __shfl_xor(a_r, 16, 64);
            a_i += __shfl_xor(a_i, 16, 64);
            a_r += __shfl_xor(a_r, 32, 64);
            a_i += __shfl_xor(a_i, 32, 64);
            double val = s_re * a_i - s_im * a_r;                 // Im(conj(s) a)
            if (NEEDB) {
                b_r += __shfl_xor(b_r, 16, 64);
                b_i += __shfl_xor(b_i, 16, 64);
                b_r += __shfl_xor(b_r, 32, 64);
                b_i += __shfl_xor(b_i, 32, 64);
                val -= s_re * b_i + s_im * b_r;                   // - Im(s b)
            }
            const int t = t0 + 16 * tile + col;
            if (g == 0 && t < N) {
                out_member[c + (size_t)t * K] = gs * val;
                fold_store(p, c + (size_t)t * K, gs * val);
            }
        }
    }
    if (blockIdx.x == 0 && lane == 0) {
        double Fk;
        if (SAND) {
            const double z = (s_re * s_re + s_im * s_im) / (double)p.n;
            Fk = 1.0 - z * z;
        } else {
            Fk = s_re * s_re - s_im * s_im;
        }
        out_member[(size_t)K * N] = Fk;
        fold_store(p, (size_t)K * N, Fk);
    }
    if (threadIdx.x == 0)
        fold_publish(p);
}

// The same for SPARSE control operators (at most R non-zeros per row: Pauli-type controls and their Liouville-space
// commutators have 1..4): per control and row R (value, column) pairs, zero padded (act_bs / act_bo, staged in LDS and read
// back as broadcasts); v_t (and w_t where b is needed) of the wave's 64 slices sits in LDS element-major, so a column is
// one ds_read_b128 at  column offset + 16 lane.  16 (5 R + 4) vector instructions per control instead of 1100: the kernel
// is left with reading the records.  (A per-entry zero test in the dense kernel -- scalar OR / compare / branch per entry --
// measured slower than the dense kernel itself: 326 vs 299 us at C4.)
template <int SAND, bool HERMB, int R, int NB>
__global__ __launch_bounds__(64) void action_forms_sparse_kernel(const TileParams p)
{
    constexpr bool NEEDB = SAND && !HERMB;
    extern __shared__ double2 s_forms[];
    const int lane = threadIdx.x, k = blockIdx.y, y = blockIdx.z, K = p.K, N = p.N;
    const int t = blockIdx.x * 64 + lane, tc = min(t, N - 1);
    const size_t kw = (size_t)y * p.E + k;
    const double2 *__restrict__ V = p.states + kw * (size_t)(N + 1) * NB;
    const double2 *__restrict__ W = p.wrec + kw * (size_t)(N + 1) * NB;
    double2 *s_v = s_forms, *s_w = s_forms + 64 * NB;             // [NB][64] each (s_w only when b is needed)
    double2 *s_tab = s_forms + (NEEDB ? 128 : 64) * NB;           // [K][NB][R] values
    int *s_off = reinterpret_cast<int *>(s_tab + (size_t)K * NB * R);   // [K][NB][R] byte offsets of the column inside s_v
    double *s_fold = reinterpret_cast<double *>(s_off + (size_t)K * NB * R);   // [64][K] outputs (single problems, fold_fg)
    double vr[NB], vi[NB], wr[NB], wi[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const double2 a = V[(size_t)i * (N + 1) + tc], b = W[(size_t)i * (N + 1) + tc];
        vr[i] = a.x;
        vi[i] = a.y;
        wr[i] = b.x;
        wi[i] = b.y;
        s_v[i * 64 + lane] = a;
        if (NEEDB)
            s_w[i * 64 + lane] = b;
    }
    {
        const size_t sel = (size_t)(p.act_shared ? 0 : k) * K * NB * R;      // per-member lists when the controls are the member's own
        for (int q = lane; q < K * NB * R; q += 64) {
            s_tab[q] = p.act_bs[sel + q];
            s_off[q] = p.act_bo[sel + q];
        }
    }
    double s_re = 0.0, s_im = 0.0;                                // s = w_N' v_N (uniform)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const double2 a = V[(size_t)i * (N + 1) + N], b = W[(size_t)i * (N + 1) + N];
        s_re = fma(b.x, a.x, fma(b.y, a.y, s_re));
        s_im = fma(b.x, a.y, fma(-b.y, a.x, s_im));
    }
    __syncthreads();
    const double gs = SAND ? -p.dt * (HERMB ? 2.0 : 1.0) : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    double *__restrict__ out_member = p.member_out + ((size_t)y * p.E_members + k) * ((size_t)K * N + 1);
    const char *vcol = reinterpret_cast<const char *>(s_v + lane), *wcol = reinterpret_cast<const char *>(s_w + lane);
    for (int c = 0; c < K; ++c) {
        const double2 *tab = s_tab + (size_t)c * NB * R;
        const int *off = s_off + (size_t)c * NB * R;
        double a_r = 0.0, a_i = 0.0, b_r = 0.0, b_i = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double ur = 0.0, ui = 0.0, xr = 0.0, xi = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const double2 b = tab[i * R + q];
                const int o = off[i * R + q];
                const double2 vj = *reinterpret_cast<const double2 *>(vcol + o);
                ur = fma(b.x, vj.x, ur);                          // (B v)[i]
                ur = fma(-b.y, vj.y, ur);
                ui = fma(b.x, vj.y, ui);
                ui = fma(b.y, vj.x, ui);
                if (NEEDB) {
                    const double2 wj = *reinterpret_cast<const double2 *>(wcol + o);
                    xr = fma(b.x, wj.x, xr);                      // (B w)[i]
                    xr = fma(-b.y, wj.y, xr);
                    xi = fma(b.x, wj.y, xi);
                    xi = fma(b.y, wj.x, xi);
                }
            }
            a_r = fma(wr[i], ur, fma(wi[i], ui, a_r));            // conj(w[i]) (B v)[i]
            a_i = fma(wr[i], ui, fma(-wi[i], ur, a_i));
            if (NEEDB) {
                b_r = fma(vr[i], xr, fma(vi[i], xi, b_r));        // conj(v[i]) (B w)[i]
                b_i = fma(vr[i], xi, fma(-vi[i], xr, b_i));
            }
        }
        double val = s_re * a_i - s_im * a_r;                     // Im(conj(s) a)
        if (NEEDB)
            val -= s_re * b_i + s_im * b_r;                       // - Im(s b)
        if (t < N)
            out_member[c + (size_t)t * K] = gs * val;
        if (p.fold_fg)
            s_fold[lane * K + c] = gs * val;
    }
    if (p.fold_fg) {                                              // the wave's 64 K outputs are contiguous: whole-line stores
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int base = blockIdx.x * 64 * K, total = K * N;
        for (int i = lane; i < 64 * K; i += 64)
            if (base + i < total)
                fold_store(p, (size_t)(base + i), s_fold[i]);
    }
    if (blockIdx.x == 0 && lane == 0) {
        double Fk;
        if (SAND) {
            const double z = (s_re * s_re + s_im * s_im) / (double)p.n;
            Fk = 1.0 - z * z;
        } else {
            Fk = s_re * s_re - s_im * s_im;
        }
        out_member[(size_t)K * N] = Fk;
        fold_store(p, (size_t)K * N, Fk);
    }
    if (threadIdx.x == 0)
        fold_publish(p);
}

template <int R, int NB>
static void launch_forms_sparse(int sandwich, const TileParams &p, dim3 grid, size_t lds, hipStream_t stream)
{
    if (!sandwich)
        GRAPE_LAUNCH((action_forms_sparse_kernel<0, true, R, NB>), grid, dim3(64), lds, stream, p);
    else if (p.herm_ctrl)
        GRAPE_LAUNCH((action_forms_sparse_kernel<1, true, R, NB>), grid, dim3(64), lds, stream, p);
    else
        GRAPE_LAUNCH((action_forms_sparse_kernel<1, false, R, NB>), grid, dim3(64), lds, stream, p);
}

template <int NB>
static hipError_t launch_forms_nb(int sandwich, const TileParams &p, hipStream_t stream);

template <int NB>
static hipError_t launch_action_nb(int sandwich, const TileParams &p, hipStream_t stream)
{
    const bool hoisted = p.act_shared || p.ctrl_scale;            // one control sum per slice serves every member (x s_k)
    const bool parts = NB == 16 && hoisted;
    if (parts)
        GRAPE_LAUNCH((action_rows_kernel<NB, true>), dim3(p.N, p.n_x), dim3(256), 0, stream, p);
    else if (hoisted)
        GRAPE_LAUNCH((action_rows_kernel<NB, false>), dim3(p.N, p.n_x), dim3(256), 0, stream, p);
    else if (NB != 16 || p.K > kActOwnK)
        return hipErrorInvalidConfiguration;                      // (the host layer keeps such ensembles on the expm flow)
    const size_t plan_bytes = sizeof(unsigned short) * (size_t)p.N * kActWaves;
    if (plan_bytes > 64 * 1024)                                   // (the host layer keeps such pulses on the expm flow)
        return hipErrorInvalidConfiguration;
    // Two members per wave (whole rows per lane) where that does not leave SIMDs idle.  With S = 4 x compute units SIMDs and at
    // most two waves of either kernel per SIMD, measured at C4's shape (profiles/r04_C4_whole.txt, chains only): up to S members
    // one member per wave (1024: 0.95 against 1.36 ms); S .. 2 S two per wave, one wave per SIMD (2048: 1.47 against 1.83 ms);
    // 2 S .. 3 S one per wave again (3072: 2.73 against 2.82 -- its second round is a half-empty one-wave round); beyond that two
    // per wave (4096: 2.96 against 3.70 ms).  GRAPE_ACT_WHOLE=0 / 1 forces either.
    const char *whole_var = std::getenv("GRAPE_ACT_WHOLE");
    const int whole_env = whole_var ? std::atoi(whole_var) : -1;
    const long simds = 4 * (long)(p.cus > 0 ? p.cus : 256), units = (long)(p.E_plan ? p.E_plan : p.E) * p.n_x;
    const bool whole = parts && (whole_env >= 0 ? whole_env != 0 : (units > simds && units <= 2 * simds) || units > 3 * simds);
    // as many workgroups per compute unit as an even spread of the launch needs, and no more: the LDS request is the limiter
    const int per_group = NB == 16 ? (whole ? 2 * kActWaves : kActWaves) : kActWaves / 2;          // members of a workgroup
    const long groups = (long)((p.E + per_group - 1) / per_group) * p.n_x, cus = p.cus > 0 ? p.cus : 256;
    const long per_cu = (groups + cus - 1) / cus;
    size_t lds = (size_t)(160 * 1024) / (size_t)per_cu;
    lds = lds > 1024 ? (lds - 512) & ~(size_t)255 : lds;
    if (lds < plan_bytes)
        lds = plan_bytes;
    auto kern = NB == 16 ? (parts ? (whole ? action_parts_kernel<true> : action_parts_kernel<false>)
                                  : action_thin_kernel)
                         : action_thin2_kernel;
    if (lds > 64 * 1024) {
        hipError_t e = ensure_dynamic_lds((const void *)kern, lds);
        if (e != hipSuccess)
            return e;
    }
    GRAPE_LAUNCH_AS(NB == 16 ? (parts ? "action_parts_kernel" : "action_thin_kernel") : "action_thin2_kernel", kern,
                    dim3((p.E + per_group - 1) / per_group, p.n_x), dim3(64 * kActWaves), lds, stream, p);
    if (p.ev_mid) {
        hipError_t e = hipEventRecord(p.ev_mid, stream);
        if (e != hipSuccess)
            return e;
    }
    return launch_forms_nb<NB>(sandwich, p, stream);
}

template <int NB>
static hipError_t launch_forms_nb(int sandwich, const TileParams &p, hipStream_t stream)
{
    const dim3 grid((p.N + 63) / 64, p.E, p.n_x);
    if (p.act_R > 0) {                                            // sparse control operators: (value, column) lists
        const bool needb = sandwich && !p.herm_ctrl;
        const size_t lds_f = sizeof(double2) * (needb ? 128 : 64) * NB + (sizeof(double2) + sizeof(int)) * (size_t)p.K * NB * p.act_R +
                             (p.fold_fg ? sizeof(double) * 64 * (size_t)p.K : 0);
        switch (lds_f <= 64 * 1024 ? p.act_R : 0) {
        case 1: launch_forms_sparse<1, NB>(sandwich, p, grid, lds_f, stream); return hipGetLastError();
        case 2: launch_forms_sparse<2, NB>(sandwich, p, grid, lds_f, stream); return hipGetLastError();
        case 3: launch_forms_sparse<3, NB>(sandwich, p, grid, lds_f, stream); return hipGetLastError();
        case 4: launch_forms_sparse<4, NB>(sandwich, p, grid, lds_f, stream); return hipGetLastError();
        case 6: launch_forms_sparse<6, NB>(sandwich, p, grid, lds_f, stream); return hipGetLastError();
        default: break;
        }
    }
    if constexpr (NB == 16) {                                    // dense operators, n <= 16: the forms on the matrix cores
        const bool valu = std::getenv("GRAPE_FORMS_VALU") != nullptr;     // (the vector-ALU kernel: tests, A/B timing)
        if (!valu) {
            if (!sandwich)
                GRAPE_LAUNCH((action_forms_mfma_kernel<0, true>), grid, dim3(64), 0, stream, p);
            else if (p.herm_ctrl)
                GRAPE_LAUNCH((action_forms_mfma_kernel<1, true>), grid, dim3(64), 0, stream, p);
            else
                GRAPE_LAUNCH((action_forms_mfma_kernel<1, false>), grid, dim3(64), 0, stream, p);
            return hipGetLastError();
        }
    }
    if (!sandwich)
        GRAPE_LAUNCH((action_forms_kernel<0, true, NB>), grid, dim3(64), 0, stream, p);
    else if (p.herm_ctrl)
        GRAPE_LAUNCH((action_forms_kernel<1, true, NB>), grid, dim3(64), 0, stream, p);
    else
        GRAPE_LAUNCH((action_forms_kernel<1, false, NB>), grid, dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_chain_prop(int sandwich, const TileParams &p, hipStream_t stream)
{
    if (!p.props_t || !p.wrec)
        return hipErrorInvalidValue;
    // one member per workgroup: its LDS ring (7 slices x 8 KB) lets two of them share a compute unit
    const size_t lds = 7 * 8192;
    if (p.tp_chunks > 1) {
        // small ensembles, the time axis in chunks (the chunk products Q_c and Q_c^T are there: chunk_product_deep_kernel,
        // launched by launch_nt): the vectors at the chunk boundaries by this kernel on the chunk products -- C dependent
        // products -- then a workgroup per (member, chunk) on the propagators
        if (!p.tp_q || !p.tp_qt || !p.tp_vec)
            return hipErrorInvalidValue;
        TileParams s = p;
        s.props = p.tp_q;
        s.props_t = p.tp_qt;
        s.N = p.tp_chunks;
        s.tp_chunks = 0;
        s.states = p.tp_vec;
        s.wrec = p.tp_vec + (size_t)p.n_x * p.E * 16 * (p.tp_chunks + 1);
        GRAPE_LAUNCH(chain_prop_kernel, dim3(p.E, p.n_x), dim3(64), lds, stream, s);
        GRAPE_LAUNCH(chain_prop_kernel, dim3(p.E, p.n_x, p.tp_chunks), dim3(64), lds, stream, p);
    } else {
        GRAPE_LAUNCH(chain_prop_kernel, dim3(p.E, p.n_x), dim3(64), lds, stream, p);
    }
    return launch_forms_nb<16>(sandwich, p, stream);
}

// p.E = members (the host layer hands over member counts, not tile units); n = 9..16 -> 16 x 16 images, 17..32 -> 32 x 32
hipError_t launch_action_thin(int sandwich, const TileParams &p, hipStream_t stream)
{
    return p.n <= 16 ? launch_action_nb<16>(sandwich, p, stream) : launch_action_nb<32>(sandwich, p, stream);
}

}  // namespace grape
