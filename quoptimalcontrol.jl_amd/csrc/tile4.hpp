// tile4.hpp -- wave-level complex matrix products on v_mfma_f64_4x4x4_4b_f64.
//
// Why not the 16x16x4 FP64 MFMA (tile.hpp)?  Measured on gfx950 (tools/ubench/fp64_rate.hip): from
// ONE wave per SIMD -- the occupancy a one-wave-per-member chain gets -- v_mfma_f64_16x16x4
// issues every ~150 cycles (33 TFLOP/s), v_mfma_f64_4x4x4_4b every ~17 (72 TFLOP/s).
//
// Lane layouts of the 4-block instruction, found by brute force on the hardware
// (tools/ubench/mfma4_probe.hip):  D_b = A_b * B_b + C_b  for blocks b = 0..3 with
//     A_b[i][k] at lane 16k + 4b + i      B_b[k][j] at lane 16k + 4b + j      D_b[i][j] at lane 16i + 4b + j
// Reading c = 4b + j as a column index 0..15: D and B are 4 x 16 "row strips" (row = lane>>4,
// col = lane&15) and, if all four A_b are the SAME 4x4 block, one instruction computes
//     D(4 x 16) += A(4 x 4) * B(4 x 16).
// A row strip r of a 16-column tile is exactly register r of tile.hpp's "D layout", so matrices
// keep the same register/dump format (TMat); a product out = op(Z) * W takes
//   * the RIGHT operand W and the result in registers (TMat), and
//   * the LEFT operand from an LDS image of Z: lane (i = l&3, k = l>>4) reads element
//     (4I + i, 4K + k) -- or its conjugate transpose -- with one ds_read_b128 (4 lanes share an
//     address = broadcast; the image's row stride of 20 double2 makes the 16 distinct addresses
//     of a wave instruction hit 16 disjoint bank quads: conflict-free).
// Right-multiplication and transposition of a register matrix go through the same image.
#pragma once
#include <hip/hip_runtime.h>

#include "tile.hpp"

namespace grape {

constexpr int kImgRow = 20;                       // double2 per image row (16 + 4 pad)
constexpr int kImgTile = 16 * kImgRow;            // double2 per 16x16 tile image

// image slot of element (row, col) of a (16 NT)^2 matrix
template <int NT>
GRAPE_DEV int img_slot(int row, int col)
{
    return ((row >> 4) * NT + (col >> 4)) * kImgTile + (row & 15) * kImgRow + (col & 15);
}

// registers (D layout) -> LDS image
template <int NT>
GRAPE_DEV void img_store(double2 *__restrict__ img, const TMat<NT> &z, int lane)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                img[(I * NT + J) * kImgTile + (4 * r + (lane >> 4)) * kImgRow + (lane & 15)] =
                    make_double2(z.re[I][J][r], z.im[I][J][r]);
}

// LDS image -> registers, optionally conjugate-transposed: out = Z or Z^H
template <int NT, bool HERM>
GRAPE_DEV void img_load(TMat<NT> &out, const double2 *__restrict__ img, int lane)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + 4 * r + (lane >> 4), col = 16 * J + (lane & 15);
                const double2 v = HERM ? img[img_slot<NT>(col, row)] : img[img_slot<NT>(row, col)];
                out.re[I][J][r] = v.x;
                out.im[I][J][r] = HERM ? -v.y : v.y;
            }
}

// Orders this wave's LDS image writes against its later image reads.  The LDS executes one
// wave's instructions in issue order, so no counter wait is needed for that -- only that the
// compiler keeps the order (it inserts the lgkmcnt waits for the values themselves).
GRAPE_DEV void lds_fence()
{
    __builtin_amdgcn_wave_barrier();
}

// out = op(Z) * W,  Z in the LDS image (op = identity or conjugate transpose), W and out in registers.
// ACC: accumulate into out instead of overwriting.  NEG: subtract the product (out -= ...) when ACC.
template <int NT, bool HERM, bool ACC = false>
GRAPE_DEV void tmul4(TMat<NT> &out, const double2 *__restrict__ img, const TMat<NT> &w, int lane)
{
    const int i = lane & 3, k = lane >> 4;
#pragma unroll
    for (int TI = 0; TI < NT; ++TI) {             // output row tile = 4 row strips
        // all left-operand blocks of the row tile first (one LDS round trip), then 4 strips x NT
        // column tiles x 4 chains = 16 NT independent accumulation chains walked K-outermost:
        // consecutive MFMAs never depend on each other and the pipeline drains once per row tile.
        double2 a[4][4 * NT];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int K = 0; K < 4 * NT; ++K) {
                const int I = 4 * TI + s4;
                a[s4][K] = HERM ? img[img_slot<NT>(4 * K + k, 4 * I + i)] : img[img_slot<NT>(4 * I + i, 4 * K + k)];
            }
        double crr[4][NT], cii[4][NT], cri[4][NT], cir[4][NT];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                crr[s4][J] = ACC ? out.re[TI][J][s4] : 0.0;
                cri[s4][J] = ACC ? out.im[TI][J][s4] : 0.0;
                cii[s4][J] = 0.0;
                cir[s4][J] = 0.0;
            }
#pragma unroll
        for (int K = 0; K < 4 * NT; ++K)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double ar = a[s4][K].x, ai = HERM ? -a[s4][K].y : a[s4][K].y;
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    const double br = w.re[K >> 2][J][K & 3], bi = w.im[K >> 2][J][K & 3];
                    crr[s4][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, br, crr[s4][J], 0, 0, 0);
                    cii[s4][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(ai, bi, cii[s4][J], 0, 0, 0);
                    cri[s4][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, bi, cri[s4][J], 0, 0, 0);
                    cir[s4][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(ai, br, cir[s4][J], 0, 0, 0);
                }
            }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                out.re[TI][J][s4] = crr[s4][J] - cii[s4][J];
                out.im[TI][J][s4] = cri[s4][J] + cir[s4][J];
            }
    }
}

// per-lane partial of the trace of a register matrix (no cross-lane step)
template <int NT>
GRAPE_DEV void ttrace_partial(double &sr, double &si, const TMat<NT> &m, int lane)
{
    sr = 0.0;
    si = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15)) {
                sr += m.re[I][I][r];
                si += m.im[I][I][r];
            }
}

// wave-uniform trace of a register matrix
template <int NT>
GRAPE_DEV void ttrace(double &tr, double &ti, const TMat<NT> &m, int lane)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15)) {
                sr += m.re[I][I][r];
                si += m.im[I][I][r];
            }
    tr = wave_sum(sr);
    ti = wave_sum(si);
}

}  // namespace grape
