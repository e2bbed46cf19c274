// exact_tile.hip -- exact control gradient / ADGRAPE functional for the MFMA tile family (n = 5..32).
//
// Same mathematics as exact_grad.hip (read its header): from the debug flow's stored P_t, X_t, L_{t+1},
//     dPhi/dx[c,t] = tr(dP_t[c] W1)  [+ conj(tr(dP_t[c] W2))] ,
//     UnitaryGate: W1 = X_t L_{t+1}'          sandwich: W1 = X_t (L_{t+1} P_t)' ,  W2 = (P_t X_t)' L_{t+1}
// with dP_t[c] the forward-mode derivative of the Taylor-8 + scaling/squaring evaluation of prop_tile_kernel,
// on the FP64 matrix cores.  One wave per (member, slice).  Layout facts used (tile.hpp): D registers as the A
// operand are the transpose; to_a_layout(Z) is the A-operand layout of Z AND, read as a D-layout matrix, Z^T;
// so tr(dP W) = sum dP .* (W^T in D layout) needs one conversion of W and no product per control.
// This is the physically meaningful open-system case: 16 x 16 Liouvillians with vectorised states (n x 1).
#include "cmat.hpp"
#include "grape_kernels.hpp"
#include "tile.hpp"

namespace grape {

template <int NT>
GRAPE_DEV void tlin2(TMat<NT> &o, double a, const TMat<NT> &x, double b, const TMat<NT> &y)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            o.re[I][J] = a * x.re[I][J] + b * y.re[I][J];
            o.im[I][J] = a * x.im[I][J] + b * y.im[I][J];
        }
}

template <int NT>
GRAPE_DEV void tacc(TMat<NT> &o, double a, const TMat<NT> &x)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            o.re[I][J] += a * x.re[I][J];
            o.im[I][J] += a * x.im[I][J];
        }
}

template <int NT>
GRAPE_DEV void tadd_identity(TMat<NT> &m, double v, int lane)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15))
                m.re[I][I][r] += v;
}

// out = M * W   (M through its A-operand layout)
template <int NT>
GRAPE_DEV void tmm(TMat<NT> &out, const TMat<NT> &m, const TMat<NT> &w, double2 *img, int lane)
{
    TOp<NT> a;
    to_a_layout(a, m, img, lane);
    tmul_an<NT, false, false>(out, a, w);
}

// out = M * W'  (both through their A-operand layouts; the B operand of tile (Kt, J) is W's A layout tile (J, Kt), conjugated)
template <int NT>
GRAPE_DEV void tmm_abh(TMat<NT> &out, const TOp<NT> &ma, const TOp<NT> &wa)
{
    tprod<NT, false, true>(
        out, [&](int I, int Kt, int kb, double &r, double &i) { r = ma.re[I][Kt][kb]; i = ma.im[I][Kt][kb]; },
        [&](int Kt, int J, int kb, double &r, double &i) { r = wa.re[J][Kt][kb]; i = wa.im[J][Kt][kb]; });
}

// sum over all elements of A .* B^T, B^T supplied as to_a_layout(B): tr(A B).
// (to_a_layout register set [I][Kt] is tile (Kt, I) of B^T in D layout: the tile indices swap)
template <int NT>
GRAPE_DEV void ttrace_ab(double &zr, double &zi, const TMat<NT> &a, const TOp<NT> &bt)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ar = a.re[I][J][r], ai = a.im[I][J][r];
                const double br = bt.re[J][I][r], bi = bt.im[J][I][r];
                sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si); si = fma(ai, br, si);
            }
    zr = wave_sum(sr);
    zi = wave_sum(si);
}

// sum over all elements of A .* BT, both in D layout: tr(A B) with BT the dump of B^T
template <int NT>
GRAPE_DEV void ttrace_elem(double &zr, double &zi, const TMat<NT> &a, const TMat<NT> &bt)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ar = a.re[I][J][r], ai = a.im[I][J][r];
                const double br = bt.re[I][J][r], bi = bt.im[I][J][r];
                sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si); si = fma(ai, br, si);
            }
    zr = wave_sum(sr);
    zi = wave_sum(si);
}

template <int NT, int SAND>
__global__ __launch_bounds__(256) void exact_tile_kernel(const TileParams p, int objective)
{
    constexpr int TSZ = NT * NT * 256;
    extern __shared__ double2 s_ex[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double2 *img = s_ex + (size_t)wave * kTileImage;
    const int k = blockIdx.y;
    const int K = p.K, N = p.N;
    const int t = blockIdx.x * 4 + wave;
    if (t >= N)
        return;                                               // whole wave (no barriers below)
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;   // [A | B_c | B_c^T | Xi | Xt]
    TMat<NT> P, X, L;
    tload(P, p.props + ((size_t)k * N + t) * TSZ, lane);
    tload(X, p.states + ((size_t)k * N + t) * TSZ, lane);
    if (t + 1 < N)
        tload(L, p.costates + ((size_t)k * N + t + 1) * TSZ, lane);
    else
        tload(L, ops + (size_t)(2 + 2 * K) * TSZ, lane);       // Xt
    // W1, W2 and Phi
    TMat<NT> W1, W2;
    double phr, phi;
    {
        TMat<NT> V;
        TOp<NT> XA;
        tmm(V, P, X, img, lane);                              // V = P X
        to_a_layout(XA, X, img, lane);
        if (SAND) {
            TMat<NT> Z;
            TOp<NT> ZA;
            tmm(Z, L, P, img, lane);                          // Z = L P
            tdot<NT, true>(phr, phi, Z, V);                   // Phi = tr(L' P X P') = tr((L P)' (P X))
            to_a_layout(ZA, Z, img, lane);
            tmm_abh(W1, XA, ZA);                              // W1 = X (L P)'
            tmul_tn<NT, true, false>(W2, V, L);               // W2 = (P X)' L
        } else {
            TOp<NT> LA;
            tdot<NT, true>(phr, phi, L, V);                   // Phi = tr(L' P X)
            to_a_layout(LA, L, img, lane);
            tmm_abh(W1, XA, LA);                              // W1 = X L'
        }
    }
    // generator (as in prop_tile_kernel) and the shared part of the Taylor evaluation
    TMat<NT> G;
    if (p.variant == 0) tzero(G); else tload(G, ops, lane);
    for (int c = 0; c < K; ++c) {
        const double xv = p.x[c + (size_t)t * K];
        TMat<NT> B;
        tload(B, ops + (size_t)(1 + c) * TSZ, lane);
        tacc(G, xv, B);
    }
    if (p.variant == 0) {
        TMat<NT> A;
        tload(A, ops, lane);
        tacc(G, 1.0, A);
    }
    const double dt = p.dt;
    double colmax = 0.0;
#pragma unroll
    for (int J = 0; J < NT; ++J) {
        double cs = 0.0;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double hr = G.re[I][J][r], hi = G.im[I][J][r];
                G.re[I][J][r] = dt * hi;
                G.im[I][J][r] = -dt * hr;
                cs += fabs(G.re[I][J][r]) + fabs(G.im[I][J][r]);
            }
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        colmax = fmax(colmax, cs);
    }
    colmax = wave_max(colmax);
    const int s = p.s_forced >= 0 ? p.s_forced : squarings_for(colmax);
    const double sc = s > 0 ? ldexp(1.0, -s) : 1.0;
    if (s > 0) tlin2(G, sc, G, 0.0, G);
    TMat<NT> A2, T1, A4, U, T2;
    tmm(A2, G, G, img, lane);
    tlin2(T1, kX1, G, kX2, A2);
    tmm(A4, A2, T1, img, lane);
    tlin2(U, kX3, A2, 1.0, A4);
    tlin2(T2, kX5, G, kX6, A2);
    tacc(T2, kX7, A4);
    tadd_identity(T2, kX4, lane);
    TMat<NT> Ps;
    if (s > 0) {                                              // value at the scaled point, for the squaring chain rule
        tmm(Ps, U, T2, img, lane);
        tacc(Ps, 1.0, G);
        tacc(Ps, kY2, A2);
        tadd_identity(Ps, 1.0, lane);
    }
    double *__restrict__ out = p.member_out + (size_t)k * ((size_t)K * N + 1);
    const double D2 = 1.0 / ((double)p.n * (double)p.n);
    const bool c1 = SAND || objective == 1;
    // one derivative per slice in the direction W1 (and W2 where the sandwich needs it) instead of one per control:
    // tr(DF_G[B] W) = tr(DF_G[W] B) for the polynomial F the evaluation above is (exact_grad.hip); a control costs a trace
    auto frechet = [&](TMat<NT> &dP, const TMat<NT> &W) {
        TMat<NT> E, dA2, dT1, dA4, dU, dT2, tmp;
        tlin2(E, sc, W, 0.0, W);
        tmm(dA2, E, G, img, lane);
        tmm(tmp, G, E, img, lane);
        tacc(dA2, 1.0, tmp);
        tlin2(dT1, kX1, E, kX2, dA2);
        tmm(dA4, dA2, T1, img, lane);
        tmm(tmp, A2, dT1, img, lane);
        tacc(dA4, 1.0, tmp);
        tlin2(dU, kX3, dA2, 1.0, dA4);
        tlin2(dT2, kX5, E, kX6, dA2);
        tacc(dT2, kX7, dA4);
        tmm(dP, dU, T2, img, lane);
        tmm(tmp, U, dT2, img, lane);
        tacc(dP, 1.0, tmp);
        tacc(dP, 1.0, E);
        tacc(dP, kY2, dA2);
        if (s > 0) {
            TMat<NT> Pq = Ps, t2;
            for (int i = 0; i < s; ++i) {
                tmm(tmp, dP, Pq, img, lane);
                tmm(t2, Pq, dP, img, lane);
                tacc(tmp, 1.0, t2);
                dP = tmp;
                tmm(t2, Pq, Pq, img, lane);
                Pq = t2;
            }
        }
    };
    TMat<NT> D1, D2m;
    frechet(D1, W1);
    const bool second = SAND && !p.herm_states;               // Hermitian X, L: W2 == W1
    if (second)
        frechet(D2m, W2);
    for (int c = 0; c < K; ++c) {
        TMat<NT> BT;
        tload(BT, ops + (size_t)(1 + K + c) * TSZ, lane);     // B_c^T: tr(D B_c) = sum D .* B_c^T
        double ar, ai, dr, di;
        ttrace_elem(ar, ai, D1, BT);
        dr = dt * ai;                                         // B'_c = (-i dt) B_c
        di = -dt * ar;
        if (SAND) {
            if (second) {
                ttrace_elem(ar, ai, D2m, BT);
                dr += dt * ai;
                di -= -dt * ar;
            } else {
                dr += dr;                                     // + conj of the same number
                di = 0.0;
            }
        }
        const double g = c1 ? -2.0 * D2 * (phr * dr + phi * di) : 2.0 * (phr * dr - phi * di);
        if (lane == 0)
            out[c + (size_t)t * K] = g;
    }
    if (t == N - 1 && lane == 0)
        out[(size_t)K * N] = c1 ? 1.0 - D2 * (phr * phr + phi * phi) : phr * phr - phi * phi;
}

template <int NT>
static hipError_t launch_exact_nt(int sandwich, const TileParams &p, int objective, hipStream_t stream)
{
    const dim3 grid((p.N + 3) / 4, p.E), block(256);
    const size_t lds = sizeof(double2) * 4 * (size_t)kTileImage;
    if (sandwich) GRAPE_LAUNCH((exact_tile_kernel<NT, 1>), grid, block, lds, stream, p, objective);
    else          GRAPE_LAUNCH((exact_tile_kernel<NT, 0>), grid, block, lds, stream, p, objective);
    return hipGetLastError();
}

hipError_t launch_exact_tile(int n, int sandwich, const TileParams &p, int objective, hipStream_t stream)
{
    switch (tile_count(n)) {
    case 1: return launch_exact_nt<1>(sandwich, p, objective, stream);
    case 2: return launch_exact_nt<2>(sandwich, p, objective, stream);
    case 3: return launch_grid_exact(3, sandwich, p, objective, stream);       // n = 33..64: sweep_grid.hip
    case 4: return launch_grid_exact(4, sandwich, p, objective, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace grape
