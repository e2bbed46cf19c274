// cmatp.hpp -- small complex matrices split over a PAIR of adjacent lanes (n even: 2, 4).
//
// Why: with one lane owning whole 4 x 4 ComplexF64 matrices (cmat.hpp, 64 VGPRs each) the sweep needs
// ~400 registers, i.e. ONE wave per SIMD, where a wave alone reaches only ~60 % of the FP64 issue rate
// and nothing hides LDS / HBM latency.  Here lanes 2c and 2c+1 share time chunk c: lane parity p owns
// the NC = n/2 columns {p NC .. p NC + NC - 1} of every matrix (32 VGPRs for n = 4), the whole kernel
// fits in < 256 registers, two waves share every SIMD, and the scan's products are split over the
// pair as well.
//
// Index convention ("own block first"): lane p stores X_loc[r, jl] = X[sigma_p(r), p NC + jl] with
// the ROW order permuted so that its own index block comes first:
//     r = b NC + rl   <->   global row  i = (b xor p) NC + rl .
// With this convention every register index below is a compile-time constant, although the two
// lanes of a pair hold different parts (see the derivations at each product).  What depends on p at
// run time are only memory addresses (global workspace, LDS operator images).
// The partner's registers come through DPP quad_perm [1,0,3,2] (v_mov_b32_dpp, no LDS).
#pragma once
#include <hip/hip_runtime.h>

#include "cmat.hpp"

namespace grape {

template <int N>
struct PMat {                          // own columns, local row order: element (r, jl) at r + jl*N
    static constexpr int NC = N / 2;
    double re[N * (N / 2)];
    double im[N * (N / 2)];
};

// value of the other lane of the pair (lane ^ 1)
GRAPE_DEV double pair_swap(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int N>
GRAPE_DEV void fetch_partner(PMat<N> &out, const PMat<N> &in)
{
#pragma unroll
    for (int e = 0; e < N * PMat<N>::NC; ++e) {
        out.re[e] = pair_swap(in.re[e]);
        out.im[e] = pair_swap(in.im[e]);
    }
}

template <int N>
GRAPE_DEV void pset_identity(PMat<N> &a)
{
#pragma unroll
    for (int jl = 0; jl < PMat<N>::NC; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            a.re[r + jl * N] = (r == jl) ? 1.0 : 0.0;     // global (p NC + jl, p NC + jl) is local row (0, jl)
            a.im[r + jl * N] = 0.0;
        }
}

// local row with the block bit flipped: the same global row as seen by the partner
template <int N>
constexpr int flip_block(int r) { return (r + N / 2) % N; }

// C = A * B.   C_loc[r, jl] = sum_kl A_loc[r, kl] B_loc[kl, jl] + Apar[rbar, kl] B_loc[NC + kl, jl]
// (k = own block: A's column is mine; k = partner block: A's column is the partner's, whose local
// row order has the blocks swapped).  CONJ_A / CONJ_B conjugate the operand elementwise.
template <int N>
GRAPE_DEV void pmul(PMat<N> &c, const PMat<N> &a, const PMat<N> &apar, const PMat<N> &b)
{
    constexpr int NC = N / 2;
#pragma unroll
    for (int jl = 0; jl < NC; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int kl = 0; kl < NC; ++kl) {
                {
                    const double ar = a.re[r + kl * N], ai = a.im[r + kl * N];
                    const double br = b.re[kl + jl * N], bi = b.im[kl + jl * N];
                    sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                    si = fma(ar, bi, si); si = fma(ai, br, si);
                }
                {
                    const double ar = apar.re[flip_block<N>(r) + kl * N], ai = apar.im[flip_block<N>(r) + kl * N];
                    const double br = b.re[NC + kl + jl * N], bi = b.im[NC + kl + jl * N];
                    sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                    si = fma(ar, bi, si); si = fma(ai, br, si);
                }
            }
            c.re[r + jl * N] = sr;
            c.im[r + jl * N] = si;
        }
}

// C = A^H * B.   C_loc[(0,rl), jl] = sum_s conj(A_loc[s, rl]) B_loc[s, jl]
//                C_loc[(1,rl), jl] = sum_s conj(Apar[sbar, rl]) B_loc[s, jl]
template <int N>
GRAPE_DEV void pmul_ah_b(PMat<N> &c, const PMat<N> &a, const PMat<N> &apar, const PMat<N> &b)
{
    constexpr int NC = N / 2;
#pragma unroll
    for (int jl = 0; jl < NC; ++jl)
#pragma unroll
        for (int rl = 0; rl < NC; ++rl) {
            double sr = 0.0, si = 0.0, tr = 0.0, ti = 0.0;
#pragma unroll
            for (int s = 0; s < N; ++s) {
                const double br = b.re[s + jl * N], bi = b.im[s + jl * N];
                {
                    const double ar = a.re[s + rl * N], ai = -a.im[s + rl * N];
                    sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                    si = fma(ar, bi, si); si = fma(ai, br, si);
                }
                {
                    const double ar = apar.re[flip_block<N>(s) + rl * N], ai = -apar.im[flip_block<N>(s) + rl * N];
                    tr = fma(ar, br, tr); tr = fma(-ai, bi, tr);
                    ti = fma(ar, bi, ti); ti = fma(ai, br, ti);
                }
            }
            c.re[rl + jl * N] = sr;
            c.im[rl + jl * N] = si;
            c.re[NC + rl + jl * N] = tr;
            c.im[NC + rl + jl * N] = ti;
        }
}

// C = A * B^H.  C_loc[r, jl] = sum_kl A_loc[r, kl] conj(B_loc[(0,jl), kl]) + Apar[rbar, kl] conj(Bpar[(1,jl), kl])
// (row p NC + jl of B: my own block-0 local row in my columns, the partner's block-1 local row in its columns)
template <int N>
GRAPE_DEV void pmul_a_bh(PMat<N> &c, const PMat<N> &a, const PMat<N> &apar, const PMat<N> &b, const PMat<N> &bpar)
{
    constexpr int NC = N / 2;
#pragma unroll
    for (int jl = 0; jl < NC; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int kl = 0; kl < NC; ++kl) {
                {
                    const double ar = a.re[r + kl * N], ai = a.im[r + kl * N];
                    const double br = b.re[jl + kl * N], bi = -b.im[jl + kl * N];
                    sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                    si = fma(ar, bi, si); si = fma(ai, br, si);
                }
                {
                    const double ar = apar.re[flip_block<N>(r) + kl * N], ai = apar.im[flip_block<N>(r) + kl * N];
                    const double br = bpar.re[NC + jl + kl * N], bi = -bpar.im[NC + jl + kl * N];
                    sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                    si = fma(ar, bi, si); si = fma(ai, br, si);
                }
            }
            c.re[r + jl * N] = sr;
            c.im[r + jl * N] = si;
        }
}

// tr(A^H B) = sum over all elements conj(A_e) B_e: own half, then the pair sum (both lanes get it)
template <int N>
GRAPE_DEV void ptrace_ah_b(double &zr, double &zi, const PMat<N> &a, const PMat<N> &b)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int e = 0; e < N * PMat<N>::NC; ++e) {
        sr = fma(a.re[e], b.re[e], sr);
        sr = fma(a.im[e], b.im[e], sr);
        si = fma(a.re[e], b.im[e], si);
        si = fma(-a.im[e], b.re[e], si);
    }
    zr = sr + pair_swap(sr);
    zi = si + pair_swap(si);
}

// tr(M): own diagonal entries are the local rows (0, jl) of column jl
template <int N>
GRAPE_DEV void ptrace(double &zr, double &zi, const PMat<N> &m)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int jl = 0; jl < PMat<N>::NC; ++jl) {
        sr += m.re[jl + jl * N];
        si += m.im[jl + jl * N];
    }
    zr = sr + pair_swap(sr);
    zi = si + pair_swap(si);
}

template <int N>
GRAPE_DEV void pshfl_up(PMat<N> &dst, const PMat<N> &src, int delta_lanes)
{
#pragma unroll
    for (int e = 0; e < N * PMat<N>::NC; ++e) {
        dst.re[e] = __shfl_up(src.re[e], delta_lanes, 64);
        dst.im[e] = __shfl_up(src.im[e], delta_lanes, 64);
    }
}

template <int N>
GRAPE_DEV void pshfl_down(PMat<N> &dst, const PMat<N> &src, int delta_lanes)
{
#pragma unroll
    for (int e = 0; e < N * PMat<N>::NC; ++e) {
        dst.re[e] = __shfl_down(src.re[e], delta_lanes, 64);
        dst.im[e] = __shfl_down(src.im[e], delta_lanes, 64);
    }
}

// max column sum of |re| + |im| over ALL columns (>= |G|_1): own columns, then the pair maximum
template <int N>
GRAPE_DEV double pnorm1_bound(const PMat<N> &g)
{
    double best = 0.0;
#pragma unroll
    for (int jl = 0; jl < PMat<N>::NC; ++jl) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < N; ++r)
            s += fabs(g.re[r + jl * N]) + fabs(g.im[r + jl * N]);
        best = fmax(best, s);
    }
    return fmax(best, pair_swap(best));
}

// Partner half of a Hermitian (SIGN = +1) or anti-Hermitian (SIGN = -1) matrix with HALF the DPP traffic:
// the partner's entries whose global row lies in MY index block are conjugates of entries I own,
//     par[(1,rl), jl] = SIGN conj(g[(1,jl), rl]) ,
// only the partner's diagonal block (both indices in its block) has to be fetched.
template <int N, int SIGN>
GRAPE_DEV void partner_of_hermitian(PMat<N> &par, const PMat<N> &g)
{
    constexpr int NC = N / 2;
#pragma unroll
    for (int jl = 0; jl < NC; ++jl)
#pragma unroll
        for (int rl = 0; rl < NC; ++rl) {
            par.re[rl + jl * N] = pair_swap(g.re[rl + jl * N]);
            par.im[rl + jl * N] = pair_swap(g.im[rl + jl * N]);
            par.re[NC + rl + jl * N] = (SIGN > 0) ? g.re[NC + jl + rl * N] : -g.re[NC + jl + rl * N];
            par.im[NC + rl + jl * N] = (SIGN > 0) ? -g.im[NC + jl + rl * N] : g.im[NC + jl + rl * N];
        }
}

// C = G * G for an anti-Hermitian G (unitary flow): the square is Hermitian, so the own diagonal block
// needs its upper triangle only (real diagonal: -sum_s |G[s, jl]|^2 from my own column); 3/4 of the FMAs.
template <int N>
GRAPE_DEV void psquare_antihermitian(PMat<N> &c, const PMat<N> &g, const PMat<N> &gpar)
{
    constexpr int NC = N / 2;
#pragma unroll
    for (int jl = 0; jl < NC; ++jl) {
        // block-1 rows (global rows of the partner's block): the generic product
#pragma unroll
        for (int r = NC; r < N; ++r) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int kl = 0; kl < NC; ++kl) {
                {
                    const double ar = g.re[r + kl * N], ai = g.im[r + kl * N];
                    const double br = g.re[kl + jl * N], bi = g.im[kl + jl * N];
                    sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                    si = fma(ar, bi, si); si = fma(ai, br, si);
                }
                {
                    const double ar = gpar.re[flip_block<N>(r) + kl * N], ai = gpar.im[flip_block<N>(r) + kl * N];
                    const double br = g.re[NC + kl + jl * N], bi = g.im[NC + kl + jl * N];
                    sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                    si = fma(ar, bi, si); si = fma(ai, br, si);
                }
            }
            c.re[r + jl * N] = sr;
            c.im[r + jl * N] = si;
        }
        // own diagonal block: (G G)[i, j] = -(G' G)[i, j] = -sum_s conj(G[s, i]) G[s, j], both columns mine
        {
            double d = 0.0;
#pragma unroll
            for (int s = 0; s < N; ++s) {
                d = fma(g.re[s + jl * N], g.re[s + jl * N], d);
                d = fma(g.im[s + jl * N], g.im[s + jl * N], d);
            }
            c.re[jl + jl * N] = -d;
            c.im[jl + jl * N] = 0.0;
        }
#pragma unroll
        for (int rl = 0; rl < jl; ++rl) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int s = 0; s < N; ++s) {
                const double ar = g.re[s + rl * N], ai = -g.im[s + rl * N];
                const double br = g.re[s + jl * N], bi = g.im[s + jl * N];
                sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
                si = fma(ar, bi, si); si = fma(ai, br, si);
            }
            c.re[rl + jl * N] = -sr;
            c.im[rl + jl * N] = -si;
            c.re[jl + rl * N] = -sr;
            c.im[jl + rl * N] = si;
        }
    }
}

// p = exp(g), degree-8 Taylor polynomial in 3 products + scaling/squaring: cmat.hpp's expm_t8 on pair matrices
// (same coefficients, same operation order per element, so both layouts give the same propagators up to
// the summation order inside the products).  g is destroyed.
// ANTIHERM: g is known to be anti-Hermitian (unitary flow): half the partner traffic for g and g^2, and the
// Hermitian half product for g^2.  norm_bound >= 0: an upper bound of |g|_1 supplied by the caller.
template <int N, bool ANTIHERM = false>
GRAPE_DEV void pexpm_t8(PMat<N> &p, PMat<N> &g, int s_forced, double norm_bound = -1.0)
{
    constexpr int NE = N * PMat<N>::NC;
    const int s = s_forced >= 0 ? s_forced : squarings_for(norm_bound >= 0.0 ? norm_bound : pnorm1_bound(g));
    if (s > 0) {
        const double sc = ldexp(1.0, -s);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            g.re[e] *= sc;
            g.im[e] *= sc;
        }
    }
    PMat<N> par, a2, a4, t;
    if (ANTIHERM) {
        partner_of_hermitian<N, -1>(par, g);
        psquare_antihermitian(a2, g, par);
    } else {
        fetch_partner(par, g);
        pmul(a2, g, par, g);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        t.re[e] = fma(kX1, g.re[e], kX2 * a2.re[e]);
        t.im[e] = fma(kX1, g.im[e], kX2 * a2.im[e]);
    }
    if (ANTIHERM)
        partner_of_hermitian<N, 1>(par, a2);
    else
        fetch_partner(par, a2);
    pmul(a4, a2, par, t);
    PMat<N> u;
#pragma unroll
    for (int jl = 0; jl < PMat<N>::NC; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const int e = r + jl * N;
            u.re[e] = fma(kX3, a2.re[e], a4.re[e]);
            u.im[e] = fma(kX3, a2.im[e], a4.im[e]);
            t.re[e] = fma(kX5, g.re[e], fma(kX6, a2.re[e], kX7 * a4.re[e]));
            t.im[e] = fma(kX5, g.im[e], fma(kX6, a2.im[e], kX7 * a4.im[e]));
            if (r == jl)
                t.re[e] += kX4;
        }
    fetch_partner(par, u);
    pmul(p, u, par, t);                            // A8
#pragma unroll
    for (int jl = 0; jl < PMat<N>::NC; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const int e = r + jl * N;
            p.re[e] += fma(kY2, a2.re[e], g.re[e]);
            p.im[e] += fma(kY2, a2.im[e], g.im[e]);
            if (r == jl)
                p.re[e] += 1.0;
        }
    for (int i = 0; i < s; ++i) {                  // undo the scaling
        fetch_partner(par, p);
        pmul(t, p, par, p);
        p = t;
    }
}

}  // namespace grape
