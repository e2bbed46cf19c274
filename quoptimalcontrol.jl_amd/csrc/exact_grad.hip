// exact_grad.hip -- exact control gradient for small operators (n <= 4): the functional path of the reference.
//
// The reference's GRAPE gradient (grad_func!, src/GRAPE.jl:261-287) is first order in dt.  Its ADGRAPE path
// (src/solve.jl:268-361 with pw_evolve, src/timeevolution.jl:28-39) differentiates the functional
//     F(x) = sum_k w_k C1(Xt_k, U_k Xi_k [U_k'])         U_k = prod_t exp(-i dt H_t)
// exactly with Zygote, and src/grape_tools.jl:26-57 (eig_factors / expm_exact_gradient, unused) sketches the
// exact derivative of the propagator.  This kernel gives that exact gradient without an AD tape:
//     dPhi/dx[c,t] = tr( L_{t+1}' dP_t[c] X_t )                                  (UnitaryGate)
//                  = tr( L_{t+1}' (dP_t[c] X_t P_t' + P_t X_t dP_t[c]') )        (State/CoherenceTransfer)
// with X_t the state before slice t, L_{t+1} the costate after it (both left in HBM by the sweep's debug flow,
// GRAPE_FLAG_KEEP_COSTATES), and dP_t[c] the Frechet derivative of exp at G_t = -i dt H_t in direction
// B'_c = -i dt B_c, obtained by differentiating the sweep's own Taylor-8 + scaling/squaring evaluation
// (cmat.hpp: expm_t8) operation by operation -- exact to the rounding of P_t itself.
//     C1-type objectives (1 - |Phi/D|^2):   dF = -(2/D^2) Re( conj(Phi) dPhi )
//     GRAPE UnitaryGate objective Re(z^2), z = conj(Phi):   dF = 2 Re( z conj(dPhi) )
// One lane per (member, slice); a wave covers 64 slices of one member, so the member's operators are wave
// uniform.  The register footprint (a dozen 4 x 4 complex matrices) spills to scratch: this is the optional
// accuracy path, not the throughput path.
#include <cstdlib>

#include "cmat.hpp"
#include "cmatp.hpp"
#include "grape_kernels.hpp"

namespace grape {

template <int N>
GRAPE_DEV void lin2(CMat<N> &o, double a, const CMat<N> &x, double b, const CMat<N> &y)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        o.re[e] = fma(a, x.re[e], b * y.re[e]);
        o.im[e] = fma(a, x.im[e], b * y.im[e]);
    }
}

template <int N>
GRAPE_DEV void acc(CMat<N> &o, double a, const CMat<N> &x)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        o.re[e] = fma(a, x.re[e], o.re[e]);
        o.im[e] = fma(a, x.im[e], o.im[e]);
    }
}

// o += a * b
template <int N>
GRAPE_DEV void mul_acc(CMat<N> &o, const CMat<N> &a, const CMat<N> &b)
{
    CMat<N> t;
    mul(t, a, b);
    acc(o, 1.0, t);
}

// tr(A * B) = sum_ij A[i,j] B[j,i]
template <int N>
GRAPE_DEV void trace_ab(double &zr, double &zi, const CMat<N> &a, const CMat<N> &b)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double ar = a.re[i + j * N], ai = a.im[i + j * N];
            const double br = b.re[j + i * N], bi = b.im[j + i * N];
            sr = fma(ar, br, sr); sr = fma(-ai, bi, sr);
            si = fma(ar, bi, si); si = fma(ai, br, si);
        }
    zr = sr;
    zi = si;
}

template <int N, int SAND>
__global__ __launch_bounds__(64) void exact_grad_kernel(const double2 *__restrict__ ops_all, const double *__restrict__ x_all,
                                                        const ExactParams p)
{
    constexpr int NN = N * N;
    const int k = blockIdx.y;
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int K = p.K, Nsl = p.N;
    const bool live = t < Nsl;
    const int tt = live ? t : Nsl - 1;                   // surplus lanes repeat the last slice (never stored)
    const double2 *__restrict__ ops = ops_all + (size_t)k * (K + 3) * NN;
    const double2 *__restrict__ opB = ops + NN;
    const double2 *__restrict__ opXt = ops + (size_t)(2 + K) * NN;
    auto ws_load = [&](CMat<N> &m, const double2 *__restrict__ ws, int slice) {
        const int L = slice / p.S, j = slice - L * p.S;
        const double2 *__restrict__ base = ws + ((size_t)k * p.S + j) * NN * p.CH + L;
#pragma unroll
        for (int e = 0; e < NN; ++e) {
            const double2 v = base[(size_t)e * p.CH];
            m.re[e] = v.x;
            m.im[e] = v.y;
        }
    };
    CMat<N> P, X, Ln;
    ws_load(P, p.props, tt);
    ws_load(X, p.states, tt);
    if (tt + 1 < Nsl) {
        ws_load(Ln, p.costates, tt + 1);                 // costate after slice t
    } else {
#pragma unroll
        for (int e = 0; e < NN; ++e) {
            const double2 v = opXt[e];
            Ln.re[e] = v.x;
            Ln.im[e] = v.y;
        }
    }
    // W: dPhi = tr(dP W1) [+ conj(tr(dP W2))] ;  Phi = tr(L_{t+1}' X_{t+1})
    CMat<N> W1, W2, tmp;
    double phr, phi;
    if (SAND) {
        CMat<N> Y;
        mul_a_bh(Y, X, P);                               // X P'
        mul_a_bh(W1, Y, Ln);                             // X P' L'
        mul(tmp, P, Y);                                  // X_{t+1} = P X P'
        trace_ah_b(phr, phi, Ln, tmp);
        mul_ah_b(Y, P, Ln);                              // P' L
        mul_ah_b(W2, X, Y);                              // X' P' L
    } else {
        mul_a_bh(W1, X, Ln);                             // X L'
        mul(tmp, P, X);                                  // X_{t+1} = P X
        trace_ah_b(phr, phi, Ln, tmp);
    }
    // generator of this slice and the shared part of the Taylor-8 evaluation
    CMat<N> G;
    if (p.variant == 0) {
#pragma unroll
        for (int e = 0; e < NN; ++e) { G.re[e] = 0.0; G.im[e] = 0.0; }
    } else {
#pragma unroll
        for (int e = 0; e < NN; ++e) { const double2 a = ops[e]; G.re[e] = a.x; G.im[e] = a.y; }
    }
    for (int c = 0; c < K; ++c) {
        const double xv = x_all[c + (size_t)tt * K];
#pragma unroll
        for (int e = 0; e < NN; ++e) {
            const double2 b = opB[c * NN + e];
            G.re[e] = fma(b.x, xv, G.re[e]);
            G.im[e] = fma(b.y, xv, G.im[e]);
        }
    }
    if (p.variant == 0) {
#pragma unroll
        for (int e = 0; e < NN; ++e) { const double2 a = ops[e]; G.re[e] += a.x; G.im[e] += a.y; }
    }
    const int s = p.s_forced >= 0 ? p.s_forced : squarings_for(norm1_bound(G));
    const double sc = s > 0 ? ldexp(1.0, -s) : 1.0;
#pragma unroll
    for (int e = 0; e < NN; ++e) { G.re[e] *= sc; G.im[e] *= sc; }
    CMat<N> A2, T1, A4, U, T2;
    mul(A2, G, G);
    lin2(T1, kX1, G, kX2, A2);
    mul(A4, A2, T1);
    lin2(U, kX3, A2, 1.0, A4);
    lin2(T2, kX5, G, kX6, A2);
    acc(T2, kX7, A4);
#pragma unroll
    for (int i = 0; i < N; ++i) T2.re[i + i * N] += kX4;
    // value at the scaled point (needed by the squaring chain rule): Ps = U T2 + G + y2 A2 + I
    CMat<N> Ps;
    mul(Ps, U, T2);
    acc(Ps, 1.0, G);
    acc(Ps, kY2, A2);
#pragma unroll
    for (int i = 0; i < N; ++i) Ps.re[i + i * N] += 1.0;

    double *__restrict__ out = p.member_out + (size_t)k * ((size_t)K * Nsl + 1);
    const double D2 = 1.0 / ((double)N * (double)N);
    // The map G -> P evaluated above is a POLYNOMIAL F in G (Taylor-8 at G / 2^s, squared s times), and for a polynomial
    //     tr( DF_G[B] W ) = tr( DF_G[W] B )        (every term  tr(G^j B G^(k-1-j) W) = tr(G^(k-1-j) W G^j B), summed over j)
    // -- so ONE derivative in the direction W1 (and one in W2 where the sandwich needs it) serves all K controls, each of
    // which is left with a trace.  (Round 2 differentiated in the K directions B'_c: 6 + 2 s products per control.)
    auto frechet = [&](CMat<N> &dP, const CMat<N> &W) {
        CMat<N> E;                                       // direction, scaled like G
#pragma unroll
        for (int e = 0; e < NN; ++e) { E.re[e] = sc * W.re[e]; E.im[e] = sc * W.im[e]; }
        CMat<N> dA2, dT1, dA4, dU, dT2;
        mul(dA2, E, G);
        mul_acc(dA2, G, E);
        lin2(dT1, kX1, E, kX2, dA2);
        mul(dA4, dA2, T1);
        mul_acc(dA4, A2, dT1);
        lin2(dU, kX3, dA2, 1.0, dA4);
        lin2(dT2, kX5, E, kX6, dA2);
        acc(dT2, kX7, dA4);
        mul(dP, dU, T2);
        mul_acc(dP, U, dT2);
        acc(dP, 1.0, E);
        acc(dP, kY2, dA2);
        CMat<N> Pq = Ps;                                 // undo the scaling: P <- P^2, dP <- dP P + P dP
        for (int i = 0; i < s; ++i) {
            mul(tmp, dP, Pq);
            mul_acc(tmp, Pq, dP);
            dP = tmp;
            mul(tmp, Pq, Pq);
            Pq = tmp;
        }
    };
    CMat<N> D1, D2m;
    frechet(D1, W1);
    const bool second = SAND && !p.herm_states;          // Hermitian X, L: W2 == W1
    if (second)
        frechet(D2m, W2);
    for (int c = 0; c < K; ++c) {
        CMat<N> Bc;
#pragma unroll
        for (int e = 0; e < NN; ++e) { const double2 b = opB[c * NN + e]; Bc.re[e] = b.x; Bc.im[e] = b.y; }
        double ar, ai, dr, di;
        trace_ab(ar, ai, D1, Bc);                        // tr(L' dP_c X [P']) = tr(DF[W1] B'_c)
        dr = ar;
        di = ai;
        if (SAND) {
            if (second)
                trace_ab(ar, ai, D2m, Bc);               // + conj(tr(dP_c X' P' L))
            dr += ar;
            di -= ai;
        }
        double g;
        if (SAND || p.objective == 1)
            g = -2.0 * D2 * (phr * dr + phi * di);       // -(2/D^2) Re(conj(Phi) dPhi)
        else
            g = 2.0 * (phr * dr - phi * di);             // F = Re(z^2), z = conj(Phi): dF = 2 Re(z dz) = 2 Re(Phi dPhi)
        if (live)
            out[c + (size_t)t * K] = g;
    }
    if (live && t == Nsl - 1) {
        double F;
        if (SAND || p.objective == 1)
            F = 1.0 - D2 * (phr * phr + phi * phi);      // C1, src/cost_functions.jl:13-17
        else
            F = phr * phr - phi * phi;                   // Re(z^2), z = conj(Phi): src/cost_functions.jl:99-101
        out[(size_t)K * Nsl] = F;
    }
}

// ---- n = 2, 4: the same evaluation on LANE PAIRS (cmatp.hpp: lane parity p owns n/2 columns of every matrix, the partner's
// half comes through DPP).  With whole 4 x 4 matrices per lane the kernel above holds a dozen 64-register matrices: 512
// registers and 1.1 - 2.3 KB of scratch per lane (420 us at C3).  A pair-split matrix is 32 registers; the intermediates are
// sequenced so that at most eleven are alive.
template <int N>
GRAPE_DEV void plin2(PMat<N> &o, double a, const PMat<N> &x, double b, const PMat<N> &y)
{
#pragma unroll
    for (int e = 0; e < N * PMat<N>::NC; ++e) {
        o.re[e] = fma(a, x.re[e], b * y.re[e]);
        o.im[e] = fma(a, x.im[e], b * y.im[e]);
    }
}

template <int N>
GRAPE_DEV void pacc(PMat<N> &o, double a, const PMat<N> &x)
{
#pragma unroll
    for (int e = 0; e < N * PMat<N>::NC; ++e) {
        o.re[e] = fma(a, x.re[e], o.re[e]);
        o.im[e] = fma(a, x.im[e], o.im[e]);
    }
}

// o += a * b   (a's partner half fetched here)
template <int N>
GRAPE_DEV void pmul_acc(PMat<N> &o, const PMat<N> &a, const PMat<N> &b)
{
    PMat<N> par, t;
    fetch_partner(par, a);
    pmul(t, a, par, b);
    pacc(o, 1.0, t);
}

template <int N>
GRAPE_DEV void pmul_f(PMat<N> &o, const PMat<N> &a, const PMat<N> &b)
{
    PMat<N> par;
    fetch_partner(par, a);
    pmul(o, a, par, b);
}

// tr(A B) = sum_ij A[i,j] B[j,i]: the pair layout of B^T is not at hand, so through  tr(A B) = tr((A')' B): conj-transpose
// products are what cmatp.hpp has -- here B comes from memory, loaded directly TRANSPOSED (see pload_t below)
template <int N>
GRAPE_DEV void ptrace_elem(double &zr, double &zi, const PMat<N> &a, const PMat<N> &bt)
{
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int e = 0; e < N * PMat<N>::NC; ++e) {
        sr = fma(a.re[e], bt.re[e], sr);
        sr = fma(-a.im[e], bt.im[e], sr);
        si = fma(a.re[e], bt.im[e], si);
        si = fma(a.im[e], bt.re[e], si);
    }
    zr = sr + pair_swap(sr);
    zi = si + pair_swap(si);
}

// W1IN (UnitaryGate, Hermitian generators -- round 4): the backward sweep of the unitary flow has left W_t = X_t L_{t+1}' in
// `states` and Phi = tr(L' X) per member in `zphi` (SweepParams::dump_w1): no P_t / X_t / L_{t+1} loads, two products less,
// and a sweep of 0.09 instead of 0.16 ms in front of this kernel.
template <int N, int SAND, bool W1IN = false>
__global__ __launch_bounds__(64) void exact_pair_kernel(const double2 *__restrict__ ops_all, const double *__restrict__ x_all,
                                                        const ExactParams p)
{
    constexpr int NN = N * N, NC = N / 2, NE = N * NC;
    const int k = blockIdx.y, lane = threadIdx.x, par = lane & 1;
    const int t = blockIdx.x * 32 + (lane >> 1);
    const int K = p.K, Nsl = p.N;
    const bool live = t < Nsl;
    const int tt = live ? t : Nsl - 1;                   // surplus pairs repeat the last slice (never stored)
    const double2 *__restrict__ ops = ops_all + (size_t)k * (K + 3) * NN;
    const double2 *__restrict__ opB = ops + NN;
    const double2 *__restrict__ opXt = ops + (size_t)(2 + K) * NN;
    // local element (r, jl) of the pair layout <-> global (i, j), column-major i + j n
    int gidx[NE], gidx_t[NE];
#pragma unroll
    for (int jl = 0; jl < NC; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const int i = (((r / NC) ^ par) * NC) + (r % NC), j = par * NC + jl;
            gidx[r + jl * N] = i + j * N;
            gidx_t[r + jl * N] = j + i * N;              // the transposed matrix' element
        }
    auto ws_load = [&](PMat<N> &m, const double2 *__restrict__ ws, int slice) {
        const int L = slice / p.S, j = slice - L * p.S;
        const double2 *__restrict__ base = ws + ((size_t)k * p.S + j) * NN * p.CH + L;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const double2 v = base[(size_t)gidx[e] * p.CH];
            m.re[e] = v.x;
            m.im[e] = v.y;
        }
    };
    auto op_load = [&](PMat<N> &m, const double2 *__restrict__ src, const int (&idx)[NE]) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const double2 v = src[idx[e]];
            m.re[e] = v.x;
            m.im[e] = v.y;
        }
    };
    PMat<N> W1, W2;
    double phr, phi;
    if (W1IN) {
        ws_load(W1, p.states, tt);
        phr = p.zphi[2 * (size_t)k];
        phi = p.zphi[2 * (size_t)k + 1];
    } else {
        PMat<N> P, X, Ln, Lp, tmp, tp;
        ws_load(P, p.props, tt);
        ws_load(X, p.states, tt);
        if (tt + 1 < Nsl)
            ws_load(Ln, p.costates, tt + 1);             // costate after slice t
        else
            op_load(Ln, opXt, gidx);
        fetch_partner(Lp, Ln);
        if (SAND) {
            PMat<N> Y, Pp, Xp;
            fetch_partner(Pp, P);
            fetch_partner(Xp, X);
            pmul_a_bh(Y, X, Xp, P, Pp);                  // X P'
            fetch_partner(tp, Y);
            pmul_a_bh(W1, Y, tp, Ln, Lp);                // X P' L'
            pmul(tmp, P, Pp, Y);                         // X_{t+1} = P X P'
            ptrace_ah_b(phr, phi, Ln, tmp);
            if (!p.herm_states) {
                pmul_ah_b(Y, P, Pp, Ln);                 // P' L
                pmul_ah_b(W2, X, Xp, Y);                 // X' P' L
            }
        } else {
            PMat<N> Xp;
            fetch_partner(Xp, X);
            pmul_a_bh(W1, X, Xp, Ln, Lp);                // X L'
            pmul_f(tmp, P, X);                           // X_{t+1} = P X
            ptrace_ah_b(phr, phi, Ln, tmp);
        }
    }
    // generator of this slice (the sweep's order of the sum) and the shared part of the Taylor-8 evaluation
    PMat<N> G;
    if (p.variant == 0) {
#pragma unroll
        for (int e = 0; e < NE; ++e) { G.re[e] = 0.0; G.im[e] = 0.0; }
    } else {
        op_load(G, ops, gidx);
    }
    for (int c = 0; c < K; ++c) {
        const double xv = x_all[c + (size_t)tt * K];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const double2 b = opB[c * NN + gidx[e]];
            G.re[e] = fma(b.x, xv, G.re[e]);
            G.im[e] = fma(b.y, xv, G.im[e]);
        }
    }
    if (p.variant == 0) {
#pragma unroll
        for (int e = 0; e < NE; ++e) { const double2 a = ops[gidx[e]]; G.re[e] += a.x; G.im[e] += a.y; }
    }
    const int s = p.s_forced >= 0 ? p.s_forced : squarings_for(pnorm1_bound(G));
    const double sc = s > 0 ? ldexp(1.0, -s) : 1.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) { G.re[e] *= sc; G.im[e] *= sc; }
    PMat<N> A2, A4, T1;
    pmul_f(A2, G, G);
    plin2(T1, kX1, G, kX2, A2);
    pmul_f(A4, A2, T1);
    // U = x3 A2 + A4 and T2 = x5 G + x6 A2 + x7 A4 + x4 I are rebuilt where they are used (two registers sets less alive)
    auto make_U = [&](PMat<N> &U) { plin2(U, kX3, A2, 1.0, A4); };
    auto make_T2 = [&](PMat<N> &T2) {
        plin2(T2, kX5, G, kX6, A2);
        pacc(T2, kX7, A4);
#pragma unroll
        for (int jl = 0; jl < NC; ++jl) T2.re[jl + jl * N] += kX4;      // own diagonal: local row (0, jl) of column jl
    };
    auto frechet = [&](PMat<N> &dP, const PMat<N> &W) {
        PMat<N> E, dA2, dT, dA4, Tm;
#pragma unroll
        for (int e = 0; e < NE; ++e) { E.re[e] = sc * W.re[e]; E.im[e] = sc * W.im[e]; }
        pmul_f(dA2, E, G);
        pmul_acc(dA2, G, E);
        plin2(dT, kX1, E, kX2, dA2);                     // dT1
        pmul_f(dA4, dA2, T1);
        pmul_acc(dA4, A2, dT);
        plin2(dT, kX3, dA2, 1.0, dA4);                   // dU
        make_T2(Tm);
        pmul_f(dP, dT, Tm);                              // dU T2
        plin2(dT, kX5, E, kX6, dA2);                     // dT2
        pacc(dT, kX7, dA4);
        make_U(Tm);
        pmul_acc(dP, Tm, dT);                            // + U dT2
        pacc(dP, 1.0, E);
        pacc(dP, kY2, dA2);
        if (s > 0) {                                     // undo the scaling: P <- P^2, dP <- dP P + P dP
            PMat<N> Pq, T2;
            make_T2(T2);
            pmul_f(Pq, Tm, T2);                          // value at the scaled point: U T2 + G + y2 A2 + I
            pacc(Pq, 1.0, G);
            pacc(Pq, kY2, A2);
#pragma unroll
            for (int jl = 0; jl < NC; ++jl) Pq.re[jl + jl * N] += 1.0;
            for (int i = 0; i < s; ++i) {
                pmul_f(dT, dP, Pq);
                pmul_acc(dT, Pq, dP);
                dP = dT;
                pmul_f(dT, Pq, Pq);
                Pq = dT;
            }
        }
    };
    double *__restrict__ out = p.member_out + (size_t)k * ((size_t)K * Nsl + 1);
    const double D2 = 1.0 / ((double)N * (double)N);
    PMat<N> D1, D2m;
    frechet(D1, W1);
    const bool second = SAND && !p.herm_states;          // Hermitian X, L: W2 == W1
    if (second)
        frechet(D2m, W2);
    for (int c = 0; c < K; ++c) {
        PMat<N> BT;                                      // B'_c transposed, in the pair layout: tr(D B) = sum D .* B^T
        op_load(BT, opB + c * NN, gidx_t);
        double ar, ai, dr, di;
        ptrace_elem(ar, ai, D1, BT);
        dr = ar;
        di = ai;
        if (SAND) {
            if (second)
                ptrace_elem(ar, ai, D2m, BT);
            dr += ar;
            di -= ai;
        }
        double g;
        if (SAND || p.objective == 1)
            g = -2.0 * D2 * (phr * dr + phi * di);       // -(2/D^2) Re(conj(Phi) dPhi)
        else
            g = 2.0 * (phr * dr - phi * di);             // F = Re(z^2), z = conj(Phi): dF = 2 Re(Phi dPhi)
        if (live && par == 0)
            out[c + (size_t)t * K] = g;
    }
    if (live && par == 0 && t == Nsl - 1) {
        double F;
        if (SAND || p.objective == 1)
            F = 1.0 - D2 * (phr * phr + phi * phi);      // C1, src/cost_functions.jl:13-17
        else
            F = phr * phr - phi * phi;                   // Re(z^2), z = conj(Phi): src/cost_functions.jl:99-101
        out[(size_t)K * Nsl] = F;
    }
}

template <int N>
static hipError_t launch_exact_pair(int sandwich, const ExactParams &p, hipStream_t stream)
{
    const dim3 grid((p.N + 31) / 32, p.E), block(64);
    if (p.w1_in && sandwich) return hipErrorInvalidConfiguration;
    if (p.w1_in)       GRAPE_LAUNCH_AS("exact_pair_kernel", (exact_pair_kernel<N, 0, true>), grid, block, 0, stream, p.ops, p.x, p);
    else if (sandwich) GRAPE_LAUNCH((exact_pair_kernel<N, 1>), grid, block, 0, stream, p.ops, p.x, p);
    else               GRAPE_LAUNCH((exact_pair_kernel<N, 0>), grid, block, 0, stream, p.ops, p.x, p);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_exact_n(int sandwich, const ExactParams &p, hipStream_t stream)
{
    const dim3 grid((p.N + 63) / 64, p.E), block(64);
    if (sandwich) GRAPE_LAUNCH((exact_grad_kernel<N, 1>), grid, block, 0, stream, p.ops, p.x, p);
    else          GRAPE_LAUNCH((exact_grad_kernel<N, 0>), grid, block, 0, stream, p.ops, p.x, p);
    return hipGetLastError();
}

hipError_t launch_exact_grad(int n, int sandwich, const ExactParams &p, hipStream_t stream)
{
    static const bool lane_env = std::getenv("GRAPE_EXACT_LANE") != nullptr;          // (the round-2 mapping, for comparison)
    const bool lane_kernel = lane_env && !p.w1_in;
    if (p.w1_in && n != 2 && n != 4) return hipErrorInvalidConfiguration;
    switch (n) {
    case 2: return lane_kernel ? launch_exact_n<2>(sandwich, p, stream) : launch_exact_pair<2>(sandwich, p, stream);
    case 3: return launch_exact_n<3>(sandwich, p, stream);
    case 4: return lane_kernel ? launch_exact_n<4>(sandwich, p, stream) : launch_exact_pair<4>(sandwich, p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace grape
