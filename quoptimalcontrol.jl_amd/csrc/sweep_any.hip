// sweep_any.hip -- the size-generic path: any operator dimension the specialised families do not cover (n = 1, n > 64).
//
// `_fom_and_gradient_GRAPE!` (src/GRAPE.jl:25-96) works for whatever size its matrices have; :101 sends "too large" systems
// to it.  The register family (n = 2..4), the tile family (5..32) and the grid family (33..64) are built around fixed tile
// counts; this kernel is the engine's answer for everything else: correct at the same 1e-10 bar, NOT fast -- plain vector
// FP64, every matrix in HBM / L2, one workgroup of 1024 threads per (member, control array) walking the reference's own
// (general) data flow slice by slice:
//     G_t = (-i dt)(A + sum_c x[c,t] B_c) in the reference's association, P_t = exp(G_t) (degree-8 Taylor polynomial in three
//     products + squarings, theta8 = 0.08: cmat.hpp's constants), forward states stored (src/GRAPE.jl:53-63), costates
//     pulled back (:65-75), gradient traces (:261-303) and the figure of merit at t = N (:77, :94).
// A product C = op(A) op(B) is n^2 dot products spread over the threads (thread idx -> element (idx % n, idx / n): the left
// operand's column is read coalesced, the right operand's entry is a broadcast).  Reductions (norm bound, traces) go through
// LDS in a fixed order: results are bitwise reproducible.  Operators and workspace are plain column-major n x n ComplexF64
// per member: ops [A | B_1..B_K | Xi | Xt], props / states / costates N matrices each, scratch 6 matrices.
#include "cmat.hpp"
#include "grape_kernels.hpp"

namespace grape {

constexpr int kAnyThreads = 1024;

// C = op(A) op(B), all n x n column-major; HA / HB: conjugate transpose.  Every thread of the workgroup calls it; ends in a
// workgroup barrier (the workgroup's global stores are visible to all its threads behind it: one compute unit, one L1).
template <bool HA, bool HB>
__device__ void any_mm(int n, const double2 *__restrict__ A, const double2 *__restrict__ B, double2 *__restrict__ C)
{
    const int nn = n * n;
    for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads) {
        const int i = idx % n, j = idx / n;
        double sr = 0.0, si = 0.0;
        for (int k = 0; k < n; ++k) {
            const double2 a = HA ? A[k + (size_t)i * n] : A[i + (size_t)k * n];
            const double2 b = HB ? B[j + (size_t)k * n] : B[k + (size_t)j * n];
            const double ar = a.x, ai = HA ? -a.y : a.y, br = b.x, bi = HB ? -b.y : b.y;
            sr = fma(ar, br, sr);
            sr = fma(-ai, bi, sr);
            si = fma(ar, bi, si);
            si = fma(ai, br, si);
        }
        C[idx] = make_double2(sr, si);
    }
    __syncthreads();
}

// sum over the workgroup of one complex value per thread, in thread order (deterministic); result in every thread
__device__ double2 any_block_sum(double vr, double vi, double *s_red)
{
    __syncthreads();
    s_red[threadIdx.x] = vr;
    s_red[kAnyThreads + threadIdx.x] = vi;
    __syncthreads();
    for (int d = kAnyThreads / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            s_red[threadIdx.x] += s_red[threadIdx.x + d];
            s_red[kAnyThreads + threadIdx.x] += s_red[kAnyThreads + threadIdx.x + d];
        }
        __syncthreads();
    }
    return make_double2(s_red[0], s_red[kAnyThreads]);
}

__global__ __launch_bounds__(kAnyThreads) void any_sweep_kernel(const AnyParams p)
{
    __shared__ double s_red[2 * kAnyThreads];
    const int n = p.n, nn = n * n, K = p.K, N = p.N;
    const int k = blockIdx.x, z = blockIdx.y;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (K + 3) * nn;
    const double2 *__restrict__ opA = ops, *__restrict__ opB = ops + nn, *__restrict__ opXi = ops + (size_t)(1 + K) * nn,
                  *__restrict__ opXt = opXi + nn;
    const double *__restrict__ x = p.x + (size_t)z * K * N;
    const size_t kw = (size_t)z * p.E + k;
    double2 *__restrict__ Pk = p.props + kw * N * nn, *__restrict__ Xk = p.states + kw * N * nn;
    double2 *__restrict__ Lk = p.costates ? p.costates + kw * N * nn : nullptr;
    double2 *__restrict__ sc = p.scratch + kw * 6 * nn;
    double2 *G = sc, *A2 = sc + nn, *A4 = sc + 2 * (size_t)nn, *T = sc + 3 * (size_t)nn, *U = sc + 4 * (size_t)nn, *Y = sc + 5 * (size_t)nn;
    double *__restrict__ out = p.member_out + ((size_t)z * p.E_rows + k) * ((size_t)K * N + 1);
    const double dt = p.dt;

    // ------------------------------------------------------------ propagators, src/timeevolution.jl:98-110 (:45-57 static)
    for (int t = 0; t < N; ++t) {
        for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads) {
            double hr, hi;
            if (p.variant == 0) {                      // (0 + B_1 x_1 + ...) + A
                hr = 0.0;
                hi = 0.0;
            } else {                                   // A + B_1 x_1 + ...
                hr = opA[idx].x;
                hi = opA[idx].y;
            }
            for (int c = 0; c < K; ++c) {
                const double xv = x[c + (size_t)t * K];
                const double2 b = opB[(size_t)c * nn + idx];
                hr = fma(b.x, xv, hr);
                hi = fma(b.y, xv, hi);
            }
            if (p.variant == 0) {
                hr += opA[idx].x;
                hi += opA[idx].y;
            }
            G[idx] = make_double2(dt * hi, -dt * hr);  // (-i dt) H
        }
        __syncthreads();
        // |G|_1 bound: max column sum of |re| + |im|
        double cs = 0.0;
        for (int j = threadIdx.x; j < n; j += kAnyThreads) {
            double s = 0.0;
            for (int i = 0; i < n; ++i)
                s += fabs(G[i + (size_t)j * n].x) + fabs(G[i + (size_t)j * n].y);
            cs = fmax(cs, s);
        }
        __syncthreads();
        s_red[threadIdx.x] = cs;
        __syncthreads();
        for (int d = kAnyThreads / 2; d >= 1; d >>= 1) {
            if ((int)threadIdx.x < d)
                s_red[threadIdx.x] = fmax(s_red[threadIdx.x], s_red[threadIdx.x + d]);
            __syncthreads();
        }
        const double colmax = s_red[0];
        __syncthreads();
        const int s = p.s_forced >= 0 ? p.s_forced : squarings_for(colmax);
        if (s > 0) {
            const double scl = ldexp(1.0, -s);
            for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads)
                G[idx] = make_double2(G[idx].x * scl, G[idx].y * scl);
            __syncthreads();
        }
        any_mm<false, false>(n, G, G, A2);
        for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads)
            T[idx] = make_double2(fma(kX1, G[idx].x, kX2 * A2[idx].x), fma(kX1, G[idx].y, kX2 * A2[idx].y));
        __syncthreads();
        any_mm<false, false>(n, A2, T, A4);
        for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads) {
            const bool diag = (idx % n) == (idx / n);
            U[idx] = make_double2(fma(kX3, A2[idx].x, A4[idx].x), fma(kX3, A2[idx].y, A4[idx].y));
            T[idx] = make_double2(fma(kX5, G[idx].x, fma(kX6, A2[idx].x, kX7 * A4[idx].x)) + (diag ? kX4 : 0.0),
                                  fma(kX5, G[idx].y, fma(kX6, A2[idx].y, kX7 * A4[idx].y)));
        }
        __syncthreads();
        double2 *P = Pk + (size_t)t * nn;
        any_mm<false, false>(n, U, T, s > 0 ? Y : P);
        double2 *cur = s > 0 ? Y : P;
        for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads) {
            const bool diag = (idx % n) == (idx / n);
            cur[idx] = make_double2(cur[idx].x + fma(kY2, A2[idx].x, G[idx].x) + (diag ? 1.0 : 0.0),
                                    cur[idx].y + fma(kY2, A2[idx].y, G[idx].y));
        }
        __syncthreads();
        for (int i = 0; i < s; ++i) {                  // undo the scaling: ping-pong Y <-> U, the last square lands in P
            double2 *dst = (i == s - 1) ? P : (cur == Y ? U : Y);
            any_mm<false, false>(n, cur, cur, dst);
            cur = dst;
        }
    }
    // ------------------------------------------------------------ forward sweep, src/GRAPE.jl:53-63
    for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads)
        Xk[idx] = opXi[idx];
    __syncthreads();
    for (int t = 0; t + 1 < N; ++t) {
        const double2 *P = Pk + (size_t)t * nn;
        if (p.sand) {
            any_mm<false, true>(n, Xk + (size_t)t * nn, P, Y);                       // X P'       (:245)
            any_mm<false, false>(n, P, Y, Xk + (size_t)(t + 1) * nn);                // P (X P')   (:246)
        } else {
            any_mm<false, false>(n, P, Xk + (size_t)t * nn, Xk + (size_t)(t + 1) * nn);       // :226
        }
    }
    // ------------------------------------------------------------ backward sweep + gradient, :65-92
    const double gs = p.sand ? -dt : (p.variant == 0 ? -2.0 * dt : 2.0 * dt);
    double2 *Lc = A2, *Ln = A4;                        // costate at t + 1 / at t (scratch, swapped per slice)
    for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads)
        Lc[idx] = opXt[idx];
    __syncthreads();
    for (int t = N - 1; t >= 0; --t) {
        const double2 *P = Pk + (size_t)t * nn, *X = Xk + (size_t)t * nn;
        if (p.sand) {
            any_mm<false, false>(n, Lc, P, Y);                                       // L P        (:248)
            any_mm<true, false>(n, P, Y, Ln);                                        // P' (L P)   (:249)
        } else {
            any_mm<true, false>(n, P, Lc, Ln);                                       // P' L       (:228)
        }
        if (Lk) {
            for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads)
                Lk[(size_t)t * nn + idx] = Ln[idx];
        }
        // z = tr(X' L)
        double zr_p = 0.0, zi_p = 0.0;
        for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads) {
            const double2 a = X[idx], b = Ln[idx];
            zr_p = fma(a.x, b.x, zr_p);
            zr_p = fma(a.y, b.y, zr_p);
            zi_p = fma(a.x, b.y, zi_p);
            zi_p = fma(-a.y, b.x, zi_p);
        }
        const double2 zz = any_block_sum(zr_p, zi_p, s_red);
        // R = X L' [- L' X]
        any_mm<false, true>(n, X, Ln, T);
        if (p.sand) {
            any_mm<true, false>(n, Ln, X, U);
            for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads)
                T[idx] = make_double2(T[idx].x - U[idx].x, T[idx].y - U[idx].y);
            __syncthreads();
        }
        for (int c = 0; c < K; ++c) {                  // w = sum_ij B_c[i][j] R[j][i]
            const double2 *Bc = opB + (size_t)c * nn;
            double wr = 0.0, wi = 0.0;
            for (int idx = threadIdx.x; idx < nn; idx += kAnyThreads) {
                const int i = idx % n, j = idx / n;
                const double2 b = Bc[idx], r = T[j + (size_t)i * n];
                wr = fma(b.x, r.x, wr);
                wr = fma(-b.y, r.y, wr);
                wi = fma(b.x, r.y, wi);
                wi = fma(b.y, r.x, wi);
            }
            const double2 ww = any_block_sum(wr, wi, s_red);
            if (threadIdx.x == 0) {
                const double im = p.sand ? ww.y : fma(ww.x, zz.y, ww.y * zz.x);
                out[c + (size_t)t * K] = gs * im;
            }
        }
        if (t == N - 1 && threadIdx.x == 0) {          // figure of merit at t = N (:77, :94)
            if (p.sand) {
                const double inv = 1.0 / (double)n;
                const double ar = zz.x * inv, ai = zz.y * inv;
                out[(size_t)K * N] = 1.0 - (ar * ar + ai * ai);                      // src/cost_functions.jl:13-17
            } else {
                out[(size_t)K * N] = zz.x * zz.x - zz.y * zz.y;                      // Re(z^2), :99-101
            }
        }
        __syncthreads();
        double2 *tmp = Lc;
        Lc = Ln;
        Ln = tmp;
    }
}

hipError_t launch_sweep_any(const AnyParams &p, hipStream_t stream)
{
    GRAPE_LAUNCH(any_sweep_kernel, dim3(p.E, p.n_x), dim3(kAnyThreads), 0, stream, p);
    return hipGetLastError();
}

}  // namespace grape
