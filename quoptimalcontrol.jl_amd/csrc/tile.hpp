// tile.hpp -- wave-level complex matrices on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// A wave owns whole (16*NT) x (16*NT) ComplexF64 matrices, NT = 1 (n <= 16) or 2 (n <= 32),
// zero-padded; this is the dense-contraction path the north star reserves for MFMA (the
// d^2 x d^2 Liouvillian products of config C4 and the 32 x 32 products of C5).
//
// "D layout" (the MFMA accumulator layout, MI355X guide section 3): for tile (I, J), register
// r in 0..3, lane l:   element (row 16I + 4r + (l>>4),  col 16J + (l&15)).
// Facts this file is built on (verified on gfx950 by tests/test_gpu_tile.py):
//   * D layout == the B-operand layout: register r of a D tile is the b-operand of k-block r.
//   * used as the A operand, the D registers of tile (Kt, I) present the TRANSPOSE: so a product
//     Z^T * B needs no data movement at all, and
//   * the A-operand layout of Z itself (lane l, k-block kb: Z[16I + (l&15)][16Kt + 4kb + (l>>4)])
//     is obtained from the D registers through a small LDS image (to_a_layout), padded so the
//     transposing 16-byte reads are at most 2-way bank-conflicted.
// A dump of the D registers to memory ((tile*4 + r)*64 + lane, double2) is therefore lane
// contiguous (1 KiB per wave access) and is the workspace format of this kernel family.
#pragma once
#include <hip/hip_runtime.h>

namespace grape {

typedef double d4 __attribute__((ext_vector_type(4)));

#define GRAPE_DEV __device__ __forceinline__

template <int NT>
struct TMat {                     // D layout
    d4 re[NT][NT];
    d4 im[NT][NT];
};

template <int NT>
struct TOp {                      // an MFMA operand set: one f64 per lane per (tile, tile, k-block)
    double re[NT][NT][4];
    double im[NT][NT][4];
};

constexpr int kTileImage = 68 * 3 + 17 * 3 + 16;      // double2 slots of one padded LDS tile image

template <int NT>
GRAPE_DEV void tzero(TMat<NT> &m)
{
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            m.re[i][j] = (d4){0, 0, 0, 0};
            m.im[i][j] = (d4){0, 0, 0, 0};
        }
}

// out = op(A) * op(B), operands given per (tile, tile, k-block).  CONJ_A / CONJ_B conjugate.
// a(I, Kt, kb) -> {re, im} of the A operand, b(Kt, J, kb) likewise.
template <int NT, bool CONJ_A, bool CONJ_B, typename FA, typename FB>
GRAPE_DEV void tprod(TMat<NT> &out, FA a, FB b)
{
    // k-blocks outermost, all NT x NT output tiles innermost, accumulating in `out` itself.  (The order does not matter
    // for the issue rate: tools/ubench/mfma_chains.hip shows ONE dependent v_mfma_f64_16x16x4 chain of one wave already
    // running at one instruction per ~62 cycles -- back-to-back accumulation has no penalty on gfx950.)
    tzero(out);
    // Three real products per complex product: T1 = Ar Br, T2 = Ai Bi accumulate in out.re / out.im, then
    // re = T1 - T2 and im = -(T1 + T2) + (Ar + Ai)(Br + Bi) with the third chain accumulating onto -(T1 + T2).
    // 12 NT^3 matrix-core instructions instead of 16 NT^3 for 16 NT^2 extra FP64 vector additions; the rounding error
    // stays O(eps n |A| |B|) normwise.
#pragma unroll
    for (int Kt = 0; Kt < NT; ++Kt)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            double ar[NT], ai[NT], br[NT], bi[NT];
#pragma unroll
            for (int I = 0; I < NT; ++I) a(I, Kt, kb, ar[I], ai[I]);
#pragma unroll
            for (int J = 0; J < NT; ++J) b(Kt, J, kb, br[J], bi[J]);
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    out.re[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[I], br[J], out.re[I][J], 0, 0, 0);
                    out.im[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[I], bi[J], out.im[I][J], 0, 0, 0);
                }
        }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            const d4 t1 = out.re[I][J], t2 = out.im[I][J];
            if (CONJ_A != CONJ_B) {          // Ai Bi enters with the opposite sign
                out.re[I][J] = t1 + t2;
                out.im[I][J] = t2 - t1;
            } else {
                out.re[I][J] = t1 - t2;
                out.im[I][J] = -(t1 + t2);
            }
        }
#pragma unroll
    for (int Kt = 0; Kt < NT; ++Kt)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            double as[NT], bs[NT];
#pragma unroll
            for (int I = 0; I < NT; ++I) {
                double r, i;
                a(I, Kt, kb, r, i);
                as[I] = CONJ_A ? r - i : r + i;
            }
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                double r, i;
                b(Kt, J, kb, r, i);
                bs[J] = CONJ_B ? r - i : r + i;
            }
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = 0; J < NT; ++J)
                    out.im[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[I], bs[J], out.im[I][J], 0, 0, 0);
            if (NT == 2)
                __builtin_amdgcn_sched_barrier(0);        // keep the operand sums of one k-block local (registers)
        }
}

// out = op(Z)^T * op(W)   (both in D layout, no data movement)
template <int NT, bool CONJ_Z, bool CONJ_W>
GRAPE_DEV void tmul_tn(TMat<NT> &out, const TMat<NT> &z, const TMat<NT> &w)
{
    tprod<NT, CONJ_Z, CONJ_W>(
        out, [&](int I, int Kt, int kb, double &r, double &i) { r = z.re[Kt][I][kb]; i = z.im[Kt][I][kb]; },
        [&](int Kt, int J, int kb, double &r, double &i) { r = w.re[Kt][J][kb]; i = w.im[Kt][J][kb]; });
}

// out = op(A) * op(W),  A given in A-operand layout (TOp), W in D layout
template <int NT, bool CONJ_A, bool CONJ_W>
GRAPE_DEV void tmul_an(TMat<NT> &out, const TOp<NT> &a, const TMat<NT> &w)
{
    tprod<NT, CONJ_A, CONJ_W>(
        out, [&](int I, int Kt, int kb, double &r, double &i) { r = a.re[I][Kt][kb]; i = a.im[I][Kt][kb]; },
        [&](int Kt, int J, int kb, double &r, double &i) { r = w.re[Kt][J][kb]; i = w.im[Kt][J][kb]; });
}

// out = op(Z)^T * op(B),  Z in D layout, B an operand set whose registers are the B layout
// of the wanted right factor (e.g. the A layout of P is the B layout of P^T)
template <int NT, bool CONJ_Z, bool CONJ_B>
GRAPE_DEV void tmul_tb(TMat<NT> &out, const TMat<NT> &z, const TOp<NT> &b)
{
    tprod<NT, CONJ_Z, CONJ_B>(
        out, [&](int I, int Kt, int kb, double &r, double &i) { r = z.re[Kt][I][kb]; i = z.im[Kt][I][kb]; },
        [&](int Kt, int J, int kb, double &r, double &i) { r = b.re[J][Kt][kb]; i = b.im[J][Kt][kb]; });
}

// A-operand layout of Z from its D registers, through the wave's private LDS image.
// a.re[I][Kt][kb] (lane l) = Z[16I + (l&15)][16Kt + 4kb + (l>>4)].
template <int NT>
GRAPE_DEV void to_a_layout(TOp<NT> &a, const TMat<NT> &z, double2 *__restrict__ img, int lane)
{
    const int rho = lane & 15, q = lane >> 4;
    const int wr = 17 * (lane >> 4) + (lane & 15);               // write slot of (r, lane) minus 68 r
    const int rd = 68 * (rho >> 2) + 17 * (rho & 3) + q;         // read slot of (rho, 4kb + q) minus 4 kb
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int Kt = 0; Kt < NT; ++Kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                img[68 * r + wr] = make_double2(z.re[I][Kt][r], z.im[I][Kt][r]);
            __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the wave's own writes
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const double2 v = img[rd + 4 * kb];
                a.re[I][Kt][kb] = v.x;
                a.im[I][Kt][kb] = v.y;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
}

// the same conversion, a ROW of tiles at a time (img: NT images): NT tiles are written, one wait, NT are read back --
// 2 NT synchronisation points instead of 2 NT^2 (a wave that has its SIMD to itself pays every one of them)
template <int NT>
GRAPE_DEV void to_a_layout_rows(TOp<NT> &a, const TMat<NT> &z, double2 *__restrict__ img, int lane)
{
    const int rho = lane & 15, q = lane >> 4;
    const int wr = 17 * (lane >> 4) + (lane & 15);
    const int rd = 68 * (rho >> 2) + 17 * (rho & 3) + q;
#pragma unroll
    for (int I = 0; I < NT; ++I) {
#pragma unroll
        for (int Kt = 0; Kt < NT; ++Kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                img[Kt * kTileImage + 68 * r + wr] = make_double2(z.re[I][Kt][r], z.im[I][Kt][r]);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int Kt = 0; Kt < NT; ++Kt)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const double2 v = img[Kt * kTileImage + rd + 4 * kb];
                a.re[I][Kt][kb] = v.x;
                a.im[I][Kt][kb] = v.y;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// memory dump <-> D registers: element (tile, r, lane) at ((I*NT + J)*4 + r)*64 + lane
template <int NT>
GRAPE_DEV void tload(TMat<NT> &m, const double2 *__restrict__ src, int lane)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 v = src[((I * NT + J) * 4 + r) * 64 + lane];
                m.re[I][J][r] = v.x;
                m.im[I][J][r] = v.y;
            }
}

template <int NT>
GRAPE_DEV void tstore(double2 *__restrict__ dst, const TMat<NT> &m, int lane)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[((I * NT + J) * 4 + r) * 64 + lane] = make_double2(m.re[I][J][r], m.im[I][J][r]);
}

GRAPE_DEV double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    return v;
}

// butterfly sum of M independent values at once: the M shuffle/add chains interleave, so the
// ~100-cycle latency of each cross-lane step is paid once per step instead of once per value
// `halves`: leave out the xor-8 step, so lanes with bit 3 clear / set end up with the sums over
// columns 0..7 / 8..15 of a tile -- the two members of a block-diagonally packed pair.
template <int M>
GRAPE_DEV void wave_sum_n(double (&v)[M], bool halves = false)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        if (halves && d == 8)
            continue;
        double o[M];
#pragma unroll
        for (int m = 0; m < M; ++m)
            o[m] = __shfl_xor(v[m], d, 64);
#pragma unroll
        for (int m = 0; m < M; ++m)
            v[m] += o[m];
    }
}

// per-lane partial of sum over all elements of conj?(A) .* B (no cross-lane step)
template <int NT, bool CONJ_A>
GRAPE_DEV void tdot_partial(double &sr, double &si, const TMat<NT> &a, const TMat<NT> &b)
{
    sr = 0.0;
    si = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ar = a.re[I][J][r], ai = CONJ_A ? -a.im[I][J][r] : a.im[I][J][r];
                const double br = b.re[I][J][r], bi = b.im[I][J][r];
                sr = fma(ar, br, sr);
                sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si);
                si = fma(ai, br, si);
            }
}

GRAPE_DEV double wave_max(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v = fmax(v, __shfl_xor(v, d, 64));
    return v;
}

// sum over all elements of conj?(A) .* B  (elementwise, both D layout) -> wave-uniform complex
template <int NT, bool CONJ_A>
GRAPE_DEV void tdot(double &zr, double &zi, const TMat<NT> &a, const TMat<NT> &b)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ar = a.re[I][J][r], ai = CONJ_A ? -a.im[I][J][r] : a.im[I][J][r];
                const double br = b.re[I][J][r], bi = b.im[I][J][r];
                sr = fma(ar, br, sr);
                sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si);
                si = fma(ai, br, si);
            }
    zr = wave_sum(sr);
    zi = wave_sum(si);
}

// ---- cross-lane sums without LDS round trips (gfx950: DPP row rotations, v_permlane16_swap / v_permlane32_swap)

// sum over the 16 lanes of a DPP row (lanes sharing l >> 4), result in every lane: rotations by 8, 4, 2, 1
template <int M>
GRAPE_DEV void row_sum_n(double (&v)[M])
{
#define GRAPE_ROR_STEP(CTRL)                                                                                   \
    {                                                                                                          \
        double o[M];                                                                                           \
        _Pragma("unroll") for (int m = 0; m < M; ++m)                                                          \
        {                                                                                                      \
            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[m]), CTRL, 0xF, 0xF, true);         \
            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[m]), CTRL, 0xF, 0xF, true);         \
            o[m] = __hiloint2double(hi, lo);                                                                   \
        }                                                                                                      \
        _Pragma("unroll") for (int m = 0; m < M; ++m) v[m] += o[m];                                            \
    }
    GRAPE_ROR_STEP(0x128)          // row_ror:8
    GRAPE_ROR_STEP(0x124)          // row_ror:4
    GRAPE_ROR_STEP(0x122)          // row_ror:2
    GRAPE_ROR_STEP(0x121)          // row_ror:1
#undef GRAPE_ROR_STEP
}

// v = a + b after exchanging halves: v_permlane32_swap / v_permlane16_swap (gfx950) trade the upper half (odd
// rows) of the first register for the lower half (even rows) of the second, so with (a, b) = (lower-index value,
// upper-index value) the sum of the two results is, in every lane, own + partner of the value that lane KEEPS
// (lanes 0..31 / even rows keep a, the others b): one step of a reduce-scatter without selects.  With a == b it
// is a plain all-reduce step.
GRAPE_DEV double swap32_add(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
GRAPE_DEV double swap16_add(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// sum over the 4 rows (lanes sharing l & 15), result in every lane
template <int M>
GRAPE_DEV void col_sum_n(double (&v)[M])
{
#pragma unroll
    for (int m = 0; m < M; ++m)
        v[m] = swap16_add(v[m], v[m]);
#pragma unroll
    for (int m = 0; m < M; ++m)
        v[m] = swap32_add(v[m], v[m]);
}

// one DPP-masked step of the reduce-scatter inside a row of 16 lanes: lanes whose bank (group of 4 lanes) is in
// UPPER keep b, the others a; partners by the DPP control CTRL (row_ror:8 pairs l with l ^ 8, row_half_mirror
// pairs l with 7 - l inside each group of 8: one lane on either side of bit 2)
template <int CTRL, int UPPER>
GRAPE_DEV double dpp_pair_add(double a, double b)
{
    constexpr int LOWER = 0xF & ~UPPER;
    int klo = __double2loint(a), khi = __double2hiint(a);                 // kept value: a, or b in the upper banks
    klo = __builtin_amdgcn_update_dpp(klo, __double2loint(b), 0xE4, 0xF, UPPER, false);      // quad_perm [0,1,2,3]
    khi = __builtin_amdgcn_update_dpp(khi, __double2hiint(b), 0xE4, 0xF, UPPER, false);
    int rlo = 0, rhi = 0;                                                 // the partner's value of the same index
    rlo = __builtin_amdgcn_update_dpp(rlo, __double2loint(a), CTRL, 0xF, LOWER, false);
    rhi = __builtin_amdgcn_update_dpp(rhi, __double2hiint(a), CTRL, 0xF, LOWER, false);
    rlo = __builtin_amdgcn_update_dpp(rlo, __double2loint(b), CTRL, 0xF, UPPER, false);
    rhi = __builtin_amdgcn_update_dpp(rhi, __double2hiint(b), CTRL, 0xF, UPPER, false);
    return __hiloint2double(khi, klo) + __hiloint2double(rhi, rlo);
}

template <int CTRL>
GRAPE_DEV double dpp_quad_add(double a)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), CTRL, 0xF, 0xF, true);
    return a + __hiloint2double(hi, lo);
}

// 16 values per lane in, the wave-wide sum of value number (lane >> 2) out (in all four lanes of that quad):
// a reduce-scatter -- 8 + 4 + 2 + 1 pair steps and two quad steps, 17 additions instead of 16 x 6.
GRAPE_DEV double reduce_scatter16(const double (&v)[16])
{
    double a[8], b[4], c[2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a[i] = swap32_add(v[i], v[i + 8]);                // lanes >= 32 keep values 8..15
#pragma unroll
    for (int i = 0; i < 4; ++i)
        b[i] = swap16_add(a[i], a[i + 4]);                // odd rows keep the upper four of their eight
#pragma unroll
    for (int i = 0; i < 2; ++i)
        c[i] = dpp_pair_add<0x128, 0xC>(b[i], b[i + 2]);  // row_ror:8; banks 2, 3 (lane bit 3) keep the upper two
    double d = dpp_pair_add<0x141, 0xA>(c[0], c[1]);      // row_half_mirror; banks 1, 3 (lane bit 2) keep the upper one
    d = dpp_quad_add<0x4E>(d);                            // quad_perm [2,3,0,1]
    d = dpp_quad_add<0xB1>(d);                            // quad_perm [1,0,3,2]
    return d;
}

// wave-wide maximum of a non-negative double, result in every lane: row rotations + permlane swaps (no LDS round trips)
GRAPE_DEV double wave_max_fast(double v)
{
#define GRAPE_ROR_MAX(CTRL)                                                                                    \
    {                                                                                                          \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);                \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);                \
        v = fmax(v, __hiloint2double(hi, lo));                                                                 \
    }
    GRAPE_ROR_MAX(0x128)
    GRAPE_ROR_MAX(0x124)
    GRAPE_ROR_MAX(0x122)
    GRAPE_ROR_MAX(0x121)
#undef GRAPE_ROR_MAX
    {
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = fmax(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
    }
    {
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = fmax(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
    }
    return v;
}

// squarings from a norm bound already divided by theta8: 0 when v <= 1 (and for NaN: the polynomial propagates it),
// else 1 + the binary exponent of v (one more than needed when v is an exact power of two) -- scalar integer
// arithmetic on the bits of v, no FP64 vector instruction
GRAPE_DEV int squarings_from_ratio(double v)
{
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    const int e = (hi >> 20) & 0x7ff;
    if (hi < 0 || e < 1023 || e == 0x7ff)
        return 0;
    if (e == 1023 && (hi & 0xfffff) == 0 && __builtin_amdgcn_readfirstlane(__double2loint(v)) == 0)
        return 0;                                                  // v == 1 exactly
    return min(e - 1022, 60);
}

}  // namespace grape
