// grape_host.hpp -- pure host-side analysis of the operators handed to grape_set_operators (no HIP calls), kept
// separate so that the CPU sanitizer build can drive it directly (tests/san/host_detect.cpp):
//   factor_rank_one      M = u u' ?  (pure-state density operators, vec(rho) vec(rho)')
//   build_sparse_lists   (coefficient, position) lists of control operators with few non-zeros
//   hermitian_to_rounding  M == M' to 4 ulp of its largest entry (all entries finite)
//   controls_scaled      B_{k,c} = s_k B_{0,c} for every member k and control c?  (amplitude inhomogeneity)
//   build_any_sparse     shared control operators with few non-zeros, by element and by control (size-generic family)
// All matrices are n x n complex, column-major, interleaved {re, im}.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace grape_host {

// M == u u' to 8 ulp of its largest entry?  u (2 n doubles) = M[:, j] / sqrt(M[j, j]) for the largest diagonal j.
inline bool factor_rank_one(const double *M, int n, double *u)
{
    int jb = 0;
    double scale = 0.0;
    for (int j = 0; j < n; ++j) {
        if (M[2 * (j + (size_t)j * n)] > M[2 * (jb + (size_t)jb * n)]) jb = j;
        for (int i = 0; i < n; ++i) {
            const double re = M[2 * (i + (size_t)j * n)], im = M[2 * (i + (size_t)j * n) + 1];
            if (!std::isfinite(re) || !std::isfinite(im))
                return false;                                  // (fmax drops NaN: non-finite entries are checked explicitly)
            scale = std::fmax(scale, std::fmax(std::fabs(re), std::fabs(im)));
        }
    }
    const double d = M[2 * (jb + (size_t)jb * n)];
    if (!(d > 0.0) || !(std::fabs(M[2 * (jb + (size_t)jb * n) + 1]) <= 4e-16 * scale) || !std::isfinite(scale))
        return false;
    const double inv = 1.0 / std::sqrt(d);
    for (int i = 0; i < n; ++i) {                              // column jb: u_i conj(u_jb), u_jb real
        u[2 * i] = M[2 * (i + (size_t)jb * n)] * inv;
        u[2 * i + 1] = M[2 * (i + (size_t)jb * n) + 1] * inv;
    }
    double dev = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const double pr = u[2 * i] * u[2 * j] + u[2 * i + 1] * u[2 * j + 1];      // u_i conj(u_j)
            const double pi = u[2 * i + 1] * u[2 * j] - u[2 * i] * u[2 * j + 1];
            dev = std::fmax(dev, std::fmax(std::fabs(pr - M[2 * (i + (size_t)j * n)]), std::fabs(pi - M[2 * (i + (size_t)j * n) + 1])));
        }
    return dev <= 8e-16 * scale;                               // false on NaN
}

// M == M' to 4 ulp of its largest entry and free of NaN / Inf?
// Are the members' control operators member 0's times ONE real scalar per member, B_{k,c} = s_k B_{0,c} -- what
// EnsembleProblem.B_g produces for amplitude inhomogeneity, B_g(k) = (1 + eps_k) B (src/problems.jl:33-41)?  B: E sets of K
// operators of nn entries.  s_k comes from the largest entry of member 0 and every entry is then compared AFTER scaling, to 4 ulp
// of itself (the caller's own product (1 + eps) b is rounded once, s_k b once more); zero patterns must agree exactly.
// s[k] on return (s[0] = 1).  false: not of that form, or a non-finite entry / factor.
inline bool controls_scaled(const double *B, size_t E, size_t K, size_t nn, std::vector<double> &s)
{
    const size_t len = 2 * K * nn;
    size_t best = 0;
    for (size_t e = 0; e < len; ++e) {
        if (!std::isfinite(B[e])) return false;
        if (std::fabs(B[e]) > std::fabs(B[best])) best = e;
    }
    if (B[best] == 0.0) return false;
    s.assign(E, 1.0);
    for (size_t k = 1; k < E; ++k) {
        const double *Bk = B + k * len;
        const double sk = Bk[best] / B[best];
        if (!std::isfinite(sk)) return false;
        for (size_t e = 0; e < len; ++e) {
            const double want = sk * B[e], have = Bk[e];
            if (!std::isfinite(have)) return false;
            if ((B[e] == 0.0) != (have == 0.0) && sk != 0.0) return false;
            if (!(std::fabs(have - want) <= 8.9e-16 * std::fabs(have))) return false;
        }
        s[k] = sk;
    }
    return true;
}

inline bool hermitian_to_rounding(const double *M, int n)
{
    double scale = 0.0, dev = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const double re = M[2 * (i + (size_t)j * n)], im = M[2 * (i + (size_t)j * n) + 1];
            const double tr = M[2 * (j + (size_t)i * n)], ti = M[2 * (j + (size_t)i * n) + 1];
            if (!std::isfinite(re) || !std::isfinite(im))
                return false;
            scale = std::fmax(scale, std::fmax(std::fabs(re), std::fabs(im)));
            dev = std::fmax(dev, std::fmax(std::fabs(re - tr), std::fabs(im + ti)));
        }
    return dev <= 4e-16 * scale;
}

// Every B (E*K operators, n x n) with at most `max_nz` non-zeros?  Then coef[(k*K+c)*max_nz + e] = B[i][j] and
// addr[...] = j * row_stride + i (the position of M[j][i] in a row-major image with `row_stride` columns), zero padded.
inline bool build_sparse_lists(const double *B, size_t E, size_t K, int n, int row_stride, int max_nz,
                               std::vector<double> &coef, std::vector<int32_t> &addr)
{
    coef.assign(E * K * (size_t)max_nz * 2, 0.0);
    addr.assign(E * K * (size_t)max_nz, 0);
    const size_t nn = (size_t)n * n;
    for (size_t m = 0; m < E * K; ++m) {
        const double *M = B + 2 * m * nn;
        int cnt = 0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const double re = M[2 * (i + (size_t)j * n)], im = M[2 * (i + (size_t)j * n) + 1];
                if (re == 0.0 && im == 0.0) continue;
                if (cnt == max_nz) return false;
                const size_t e = m * (size_t)max_nz + cnt++;
                coef[2 * e] = re;
                coef[2 * e + 1] = im;
                addr[e] = j * row_stride + i;                  // B[i][j] multiplies M[j][i]: row j, column i
            }
    }
    return true;
}

// Size-generic family (sweep_any.hip), shared control operators with few non-zeros (local Pauli-type drives on >= 7 qubits: n
// entries of n^2): the non-zeros of member 0's B_1..B_K in two orders, positions = i + j n (column-major, as the kernel's matrices):
//   by ELEMENT  tidx[m] = the m-th element that any control touches (ascending), tptr[m] .. tptr[m + 1] = its entries (control,
//               coefficient), controls ascending -- H[e] = A[e] + sum over them of x_c coefficient: the dense sum without its
//               zero terms, in the same order;
//   by CONTROL  cptr[c] .. cptr[c + 1] = the (position, coefficient) of B_c's non-zeros, for the gradient traces.
// Returns the number of non-zeros (0: none at all -- nothing built), or -1 beyond max_total.
inline long build_any_sparse(const double *B0, size_t K, int n, size_t max_total, std::vector<int32_t> &tidx, std::vector<int32_t> &tptr,
                             std::vector<int32_t> &ectl, std::vector<double> &ecoef, std::vector<int32_t> &cptr,
                             std::vector<int32_t> &caddr, std::vector<double> &ccoef)
{
    const size_t nn = (size_t)n * n;
    std::vector<int32_t> cnt(nn, 0);
    cptr.assign(K + 1, 0);
    size_t total = 0;
    for (size_t c = 0; c < K; ++c) {
        for (size_t e = 0; e < nn; ++e)
            if (B0[2 * (c * nn + e)] != 0.0 || B0[2 * (c * nn + e) + 1] != 0.0) {
                ++cnt[e];
                ++total;
                if (total > max_total)
                    return -1;
            }
        cptr[c + 1] = (int32_t)total;
    }
    tidx.clear();
    tptr.assign(1, 0);
    if (total == 0)
        return 0;
    std::vector<int32_t> slot(nn, -1);                        // element -> its first entry
    for (size_t e = 0; e < nn; ++e)
        if (cnt[e]) {
            slot[e] = tptr.back();
            tidx.push_back((int32_t)e);
            tptr.push_back(tptr.back() + cnt[e]);
        }
    ectl.assign(total, 0);
    ecoef.assign(2 * total, 0.0);
    caddr.assign(total, 0);
    ccoef.assign(2 * total, 0.0);
    size_t q = 0;
    for (size_t c = 0; c < K; ++c)
        for (size_t e = 0; e < nn; ++e) {
            const double re = B0[2 * (c * nn + e)], im = B0[2 * (c * nn + e) + 1];
            if (re == 0.0 && im == 0.0)
                continue;
            const size_t at = (size_t)slot[e]++;          // (controls ascending: c is the outer loop)
            ectl[at] = (int32_t)c;
            ecoef[2 * at] = re;
            ecoef[2 * at + 1] = im;
            caddr[q] = (int32_t)e;
            ccoef[2 * q] = re;
            ccoef[2 * q + 1] = im;
            ++q;
        }
    return (long)total;
}

}  // namespace grape_host
