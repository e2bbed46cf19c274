// cmat.hpp -- register-resident small complex matrices for the lane-per-time-chunk kernels.
//
// One lane owns whole n x n ComplexF64 matrices (n <= 4) as split re/im arrays that the
// compiler keeps in VGPRs (every index below is a compile-time constant after unrolling).
// Column-major element index e = i + j*n, as in the reference's Julia arrays.
#pragma once
#include <hip/hip_runtime.h>

#include "grape_kernels.hpp"   // kTheta8

namespace grape {

template <int N>
struct CMat {
    double re[N * N];
    double im[N * N];
};

#define GRAPE_DEV __device__ __forceinline__

template <int N>
GRAPE_DEV void set_identity(CMat<N> &a)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        a.re[e] = ((e % N) == (e / N)) ? 1.0 : 0.0;
        a.im[e] = 0.0;
    }
}

// C = A * B
template <int N>
GRAPE_DEV void mul(CMat<N> &c, const CMat<N> &a, const CMat<N> &b)
{
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double ar = a.re[i + k * N], ai = a.im[i + k * N];
                const double br = b.re[k + j * N], bi = b.im[k + j * N];
                sr = fma(ar, br, sr);
                sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si);
                si = fma(ai, br, si);
            }
            c.re[i + j * N] = sr;
            c.im[i + j * N] = si;
        }
}

// C = A^H * B
template <int N>
GRAPE_DEV void mul_ah_b(CMat<N> &c, const CMat<N> &a, const CMat<N> &b)
{
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double ar = a.re[k + i * N], ai = -a.im[k + i * N];
                const double br = b.re[k + j * N], bi = b.im[k + j * N];
                sr = fma(ar, br, sr);
                sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si);
                si = fma(ai, br, si);
            }
            c.re[i + j * N] = sr;
            c.im[i + j * N] = si;
        }
}

// C = A * B^H
template <int N>
GRAPE_DEV void mul_a_bh(CMat<N> &c, const CMat<N> &a, const CMat<N> &b)
{
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double ar = a.re[i + k * N], ai = a.im[i + k * N];
                const double br = b.re[j + k * N], bi = -b.im[j + k * N];
                sr = fma(ar, br, sr);
                sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si);
                si = fma(ai, br, si);
            }
            c.re[i + j * N] = sr;
            c.im[i + j * N] = si;
        }
}

// tr(A^H B) = sum_e conj(A_e) B_e
template <int N>
GRAPE_DEV void trace_ah_b(double &zr, double &zi, const CMat<N> &a, const CMat<N> &b)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        sr = fma(a.re[e], b.re[e], sr);
        sr = fma(a.im[e], b.im[e], sr);
        si = fma(a.re[e], b.im[e], si);
        si = fma(-a.im[e], b.re[e], si);
    }
    zr = sr;
    zi = si;
}

// wave shuffles of a whole matrix
template <int N>
GRAPE_DEV void shfl_up(CMat<N> &dst, const CMat<N> &src, int delta)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        dst.re[e] = __shfl_up(src.re[e], delta, 64);
        dst.im[e] = __shfl_up(src.im[e], delta, 64);
    }
}

template <int N>
GRAPE_DEV void shfl_down(CMat<N> &dst, const CMat<N> &src, int delta)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        dst.re[e] = __shfl_down(src.re[e], delta, 64);
        dst.im[e] = __shfl_down(src.im[e], delta, 64);
    }
}

// ---------------------------------------------------------------------------------------
// Matrix exponential: degree-8 Taylor polynomial evaluated with 3 matrix products
// (Bader, Blanes, Casas, "Computing the matrix exponential with an optimized Taylor
// polynomial approximation", Mathematics 7(12), 2019, eq. for T8), preceded by scaling with
// 2^-s and followed by s squarings.  Division-free and pivot-free, so every lane of a wave
// runs the same instruction stream; with |G| <= theta8 the truncation error is below
// 2^-53 |G|.  The reference calls Julia's LinearAlgebra.exp! (Higham-2005 Pade, restated in
// oracle/grape_oracle.c); both are backward stable and agree to ~1e-16 per propagator.
// ---------------------------------------------------------------------------------------
// kTheta8 (grape_kernels.hpp): the norm up to which the degree-8 polynomial is used unscaled

constexpr double kSqrt177 = 13.304134695650071;
constexpr double kX3 = 2.0 / 3.0;
constexpr double kX1 = kX3 * (1.0 + kSqrt177) / 88.0;
constexpr double kX2 = kX3 * (1.0 + kSqrt177) / 352.0;
constexpr double kX4 = (-271.0 + 29.0 * kSqrt177) / (315.0 * kX3);
constexpr double kX5 = 11.0 * (-1.0 + kSqrt177) / (1260.0 * kX3);
constexpr double kX6 = 11.0 * (-9.0 + kSqrt177) / (5040.0 * kX3);
constexpr double kX7 = (89.0 - kSqrt177) / (5040.0 * kX3 * kX3);
constexpr double kY2 = (857.0 - 58.0 * kSqrt177) / 630.0;

// upper bound of the 1-norm: max column sum of |re|+|im|  (>= |G|_1, <= sqrt(2)|G|_1)
template <int N>
GRAPE_DEV double norm1_bound(const CMat<N> &g)
{
    double best = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i)
            s += fabs(g.re[i + j * N]) + fabs(g.im[i + j * N]);
        best = fmax(best, s);
    }
    return best;
}

// number of squarings for a generator of norm bound theta (0 when theta <= theta8)
GRAPE_DEV int squarings_for(double theta)
{
    if (!(theta > kTheta8))
        return 0;          // also the NaN case: propagate NaN through the polynomial
    int ex;
    const double m = frexp(theta / kTheta8, &ex);   // theta/theta8 = m * 2^ex, m in [0.5,1)
    int s = (m == 0.5) ? ex - 1 : ex;
    return s > 60 ? 60 : s;
}

// C = G * G for an anti-Hermitian G (G = -i dt H, H Hermitian): the square is Hermitian, so only
// the upper triangle is computed (diagonal: -sum_k |g_ik|^2, real) and mirrored -- half the FMAs.
template <int N>
GRAPE_DEV void square_antihermitian(CMat<N> &c, const CMat<N> &g)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            d = fma(g.re[i + k * N], g.re[i + k * N], d);
            d = fma(g.im[i + k * N], g.im[i + k * N], d);
        }
        c.re[i + i * N] = -d;
        c.im[i + i * N] = 0.0;
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double ar = g.re[i + k * N], ai = g.im[i + k * N];
                const double br = g.re[k + j * N], bi = g.im[k + j * N];
                sr = fma(ar, br, sr);
                sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si);
                si = fma(ai, br, si);
            }
            c.re[i + j * N] = sr;
            c.im[i + j * N] = si;
            c.re[j + i * N] = sr;
            c.im[j + i * N] = -si;
        }
    }
}

// p = exp(g); g is destroyed.  s_forced < 0: choose s from the norm of g.
// ANTIHERM: g is known to be anti-Hermitian (unitary data flow).
template <int N, bool ANTIHERM = false>
GRAPE_DEV void expm_t8(CMat<N> &p, CMat<N> &g, int s_forced)
{
    const int s = s_forced >= 0 ? s_forced : squarings_for(norm1_bound(g));
    if (s > 0) {
        const double sc = ldexp(1.0, -s);
#pragma unroll
        for (int e = 0; e < N * N; ++e) {
            g.re[e] *= sc;
            g.im[e] *= sc;
        }
    }
    CMat<N> a2, a4, t;
    if (ANTIHERM)
        square_antihermitian(a2, g);
    else
        mul(a2, g, g);
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        t.re[e] = fma(kX1, g.re[e], kX2 * a2.re[e]);
        t.im[e] = fma(kX1, g.im[e], kX2 * a2.im[e]);
    }
    mul(a4, a2, t);
    CMat<N> u;
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        u.re[e] = fma(kX3, a2.re[e], a4.re[e]);
        u.im[e] = fma(kX3, a2.im[e], a4.im[e]);
        t.re[e] = fma(kX5, g.re[e], fma(kX6, a2.re[e], kX7 * a4.re[e]));
        t.im[e] = fma(kX5, g.im[e], fma(kX6, a2.im[e], kX7 * a4.im[e]));
        if ((e % N) == (e / N))
            t.re[e] += kX4;
    }
    mul(p, u, t);                                  // A8
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        p.re[e] += fma(kY2, a2.re[e], g.re[e]);
        p.im[e] += fma(kY2, a2.im[e], g.im[e]);
        if ((e % N) == (e / N))
            p.re[e] += 1.0;
    }
    for (int i = 0; i < s; ++i) {                  // undo the scaling
        mul(t, p, p);
        p = t;
    }
}

}  // namespace grape
