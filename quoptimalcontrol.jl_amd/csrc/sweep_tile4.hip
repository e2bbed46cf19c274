// sweep_tile4.hip -- the n = 5..32 hot path on v_mfma_f64_4x4x4_4b_f64 (tile4.hpp); same data
// layout, workspace format and kernel split as sweep_tile.hip (which stays as the 16x16x4
// reference implementation, GRAPE_TILE_MFMA16=1), roughly twice its speed at one wave per SIMD.
//
// Every product is  registers <- op(LDS image) * registers  (tile4.hpp), so each step is arranged
// with the propagator -- and the stored forward state -- as LEFT operands:
//   forward  UG:  X' = P X
//            ST:  Y = P X ;  X' = (P Y')'                               (one transpose through LDS)
//   backward UG:  L' = P' L ;  W = L'' ;  R = X W = X L''               (tr(X' L) = conj(tr R))
//            ST:  Y = P' L ;  W = P' Y' = (P' L P)' ;  L' = W' ;
//                 R = X W = X L'' ,  N1 = X' L'
//   gradient      tr(L' B X)            = sum R .* B^T                          (UnitaryGate)
//                 tr(L' [B, X])         = sum R .* B^T - conj( sum N1 .* conj(B) )   (sandwich)
//   figure of merit  tr(X' L) = conj(tr R)  (UG) ;  tr(L' X) = conj(tr N1)  (sandwich)
#include "cmat.hpp"
#include "grape_kernels.hpp"
#include "tile4.hpp"

namespace grape {

template <int NT>
GRAPE_DEV void add_identity(TMat<NT> &m, double v, int lane)
{
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15))
                m.re[I][I][r] += v;
}

// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void prop_tile4_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    constexpr int IMG = NT * NT * kImgTile;
    extern __shared__ double2 s_dyn4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.y;
    const int t = blockIdx.x * 4 + wave;
    if (t >= p.N)
        return;
    const int K = p.K;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    double2 *img = s_dyn4 + (size_t)wave * IMG;

    TMat<NT> G;
    if (p.variant == 0)
        tzero(G);
    else
        tload(G, ops, lane);
    for (int c = 0; c < K; ++c) {
        const double xv = p.x[c + (size_t)t * K];
        TMat<NT> B;
        tload(B, ops + (size_t)(1 + c) * TSZ, lane);
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    G.re[I][J][r] = fma(B.re[I][J][r], xv, G.re[I][J][r]);
                    G.im[I][J][r] = fma(B.im[I][J][r], xv, G.im[I][J][r]);
                }
    }
    if (p.variant == 0) {
        TMat<NT> A;
        tload(A, ops, lane);
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                G.re[I][J] += A.re[I][J];
                G.im[I][J] += A.im[I][J];
            }
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
                G.re[I][J][r] = dt * hi;                       // (-i dt) H
                G.im[I][J][r] = -dt * hr;
                cs += fabs(G.re[I][J][r]) + fabs(G.im[I][J][r]);
            }
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        colmax = fmax(colmax, cs);
    }
    colmax = wave_max(colmax);
    const int s = p.s_forced >= 0 ? p.s_forced : squarings_for(colmax);
    if (s > 0) {
        const double sc = ldexp(1.0, -s);
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                G.re[I][J] *= sc;
                G.im[I][J] *= sc;
            }
    }

    // expm_t8 (cmat.hpp).  All factors are polynomials in G and commute, so whichever factor is
    // convenient goes to the LDS image as the left operand.
    TMat<NT> A2, A4, U, T, P;
    img_store(img, G, lane);
    lds_fence();
    tmul4<NT, false>(A2, img, G, lane);                        // A2 = G G
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            T.re[I][J] = kX1 * G.re[I][J] + kX2 * A2.re[I][J];
            T.im[I][J] = kX1 * G.im[I][J] + kX2 * A2.im[I][J];
        }
    lds_fence();
    img_store(img, A2, lane);
    lds_fence();
    tmul4<NT, false>(A4, img, T, lane);                        // A4 = A2 (x1 G + x2 A2)
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            U.re[I][J] = kX3 * A2.re[I][J] + A4.re[I][J];
            U.im[I][J] = kX3 * A2.im[I][J] + A4.im[I][J];
            T.re[I][J] = kX5 * G.re[I][J] + kX6 * A2.re[I][J] + kX7 * A4.re[I][J];
            T.im[I][J] = kX5 * G.im[I][J] + kX6 * A2.im[I][J] + kX7 * A4.im[I][J];
        }
    add_identity(T, kX4, lane);
    lds_fence();
    img_store(img, U, lane);
    lds_fence();
    tmul4<NT, false>(P, img, T, lane);                         // A8
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            P.re[I][J] += G.re[I][J] + kY2 * A2.re[I][J];
            P.im[I][J] += G.im[I][J] + kY2 * A2.im[I][J];
        }
    add_identity(P, 1.0, lane);
    for (int i = 0; i < s; ++i) {
        lds_fence();
        img_store(img, P, lane);
        lds_fence();
        tmul4<NT, false>(T, img, P, lane);
        P = T;
    }
    tstore(p.props + ((size_t)k * p.N + t) * TSZ, P, lane);
}

// ---------------------------------------------------------------------------------------------
template <int NT, int SAND, bool KEEPL>
__global__ __launch_bounds__(64) void chain_tile4_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    constexpr int IMG = NT * NT * kImgTile;
    extern __shared__ double2 s_dyn4[];
    double2 *img = s_dyn4;                                      // one matrix image at a time
    double2 *s_bops = s_dyn4 + IMG;                             // optional cache: B_c and B_c^T dumps
    const int lane = threadIdx.x;
    const int k = blockIdx.x;
    const int K = p.K, N = p.N;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ opB = ops + (size_t)TSZ;         // B_c, then B_c^T
    const double2 *__restrict__ Pk = p.props + (size_t)k * N * TSZ;
    double2 *__restrict__ Xk = p.states + (size_t)k * N * TSZ;
    double *__restrict__ out = p.member_out + (size_t)k * ((size_t)K * N + 1);
    const bool b_lds = p.bt_in_lds != 0;
    if (b_lds) {
        for (int i = lane; i < 2 * K * TSZ; i += 64)
            s_bops[i] = opB[i];
        lds_fence();
    }

    // ------------------------------------------------------------ forward sweep
    {
        TMat<NT> X, Y, Pm, Pn;
        tload(X, ops + (size_t)(1 + 2 * K) * TSZ, lane);        // Xi
        tload(Pm, Pk, lane);
        for (int t = 0; t < N; ++t) {
            tstore(Xk + (size_t)t * TSZ, X, lane);
            if (t + 2 < N)
                tload(Pn, Pk + (size_t)(t + 1) * TSZ, lane);    // next slice's P in flight
            if (t + 1 < N) {                                    // X_N is never read
                img_store(img, Pm, lane);
                lds_fence();
                tmul4<NT, false>(Y, img, X, lane);              // Y = P X
                if (SAND) {
                    TMat<NT> Yh;
                    lds_fence();
                    img_store(img, Y, lane);
                    lds_fence();
                    img_load<NT, true>(Yh, img, lane);          // Y'
                    lds_fence();
                    img_store(img, Pm, lane);
                    lds_fence();
                    tmul4<NT, false>(Y, img, Yh, lane);         // P Y' = (P X P')'
                    lds_fence();
                    img_store(img, Y, lane);
                    lds_fence();
                    img_load<NT, true>(X, img, lane);           // X' = P X P'
                } else {
                    X = Y;
                }
                lds_fence();
            }
            Pm = Pn;
        }
    }

    // ------------------------------------------------------------ backward sweep + gradient
    TMat<NT> L, Y, W, R, Pm, Pn, Xm, Xn;
    tload(L, ops + (size_t)(2 + 2 * K) * TSZ, lane);            // Xt
    const double gs = SAND ? -p.dt : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    tload(Pm, Pk + (size_t)(N - 1) * TSZ, lane);
    tload(Xm, Xk + (size_t)(N - 1) * TSZ, lane);
    for (int t = N - 1; t >= 0; --t) {
        if (t > 0) {
            tload(Pn, Pk + (size_t)(t - 1) * TSZ, lane);
            tload(Xn, Xk + (size_t)(t - 1) * TSZ, lane);
        }
        img_store(img, Pm, lane);
        lds_fence();
        tmul4<NT, true>(Y, img, L, lane);                       // Y = P' L
        if (SAND) {
            TMat<NT> Yh;
            lds_fence();
            img_store(img, Y, lane);
            lds_fence();
            img_load<NT, true>(Yh, img, lane);                  // Y'
            lds_fence();
            img_store(img, Pm, lane);
            lds_fence();
            tmul4<NT, true>(W, img, Yh, lane);                  // W = P' Y' = (P' L P)'
            lds_fence();
            img_store(img, W, lane);
            lds_fence();
            img_load<NT, true>(L, img, lane);                   // L_t = W'
        } else {
            L = Y;
            lds_fence();
            img_store(img, L, lane);
            lds_fence();
            img_load<NT, true>(W, img, lane);                   // W = L_t'
        }
        if (KEEPL)
            tstore(p.costates + ((size_t)k * N + t) * TSZ, L, lane);
        lds_fence();
        img_store(img, Xm, lane);                               // X_t as left operand
        lds_fence();
        tmul4<NT, false>(R, img, W, lane);                      // R = X L'
        if (SAND)
            tmul4<NT, true>(Y, img, L, lane);                   // N1 = X' L
        // all cross-lane sums of this slice are taken together (wave_sum_n): the trace that
        // gives tr(X'L) / tr(L'X) and, per control, sum R.*B^T (and sum conj(B).*N1)
        double zr = 0.0, zi = 0.0;
        for (int c0 = 0; c0 < K; c0 += 4) {
            constexpr int PER = SAND ? 4 : 2;
            double v[2 + 4 * PER];
            ttrace_partial(v[0], v[1], SAND ? Y : R, lane);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = c0 + cc;
                TMat<NT> Bm;
#pragma unroll
                for (int q = 0; q < PER; ++q) v[2 + cc * PER + q] = 0.0;
                if (c < K) {
                    if (b_lds) tload(Bm, s_bops + (size_t)(K + c) * TSZ, lane);
                    else       tload(Bm, opB + (size_t)(K + c) * TSZ, lane);
                    tdot_partial<NT, false>(v[2 + cc * PER], v[3 + cc * PER], Bm, R);     // sum_ij B[i,j] R[j,i]
                    if (SAND) {
                        if (b_lds) tload(Bm, s_bops + (size_t)c * TSZ, lane);
                        else       tload(Bm, opB + (size_t)c * TSZ, lane);
                        tdot_partial<NT, true>(v[4 + cc * PER], v[5 + cc * PER], Bm, Y);  // conj(tr(B L' X))
                    }
                }
            }
            wave_sum_n(v);
            zr = v[0];
            zi = -v[1];                                         // tr(X'L) = conj(tr R) ; tr(L'X) = conj(tr N1)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = c0 + cc;
                double wr = v[2 + cc * PER], wi = v[3 + cc * PER];
                if (SAND) {
                    wr -= v[4 + cc * PER];
                    wi += v[5 + cc * PER];
                }
                const double im = SAND ? wi : fma(wr, zi, wi * zr);
                if (c < K && lane == 0)
                    out[c + (size_t)t * K] = gs * im;
            }
        }
        if (t == N - 1 && lane == 0) {
            if (SAND) {
                const double inv = 1.0 / (double)p.n;
                const double ar = zr * inv, ai = zi * inv;
                out[(size_t)K * N] = 1.0 - (ar * ar + ai * ai);
            } else {
                out[(size_t)K * N] = zr * zr - zi * zi;
            }
        }
        lds_fence();
        Pm = Pn;
        Xm = Xn;
    }
}

// ---------------------------------------------------------------------------------------------
template <int NT>
static hipError_t launch_nt4(int sandwich, bool keepl, const TileParams &p, hipStream_t stream)
{
    const size_t img_bytes = sizeof(double2) * NT * NT * kImgTile;
    if (4 * img_bytes > 64 * 1024) {                          // above the default dynamic-LDS cap: opt in
        hipError_t ea = hipFuncSetAttribute((const void *)prop_tile4_kernel<NT>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * img_bytes));
        if (ea != hipSuccess)
            return ea;
    }
    hipLaunchKernelGGL(prop_tile4_kernel<NT>, dim3((p.N + 3) / 4, p.E), dim3(256), 4 * img_bytes, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    TileParams q = p;
    const size_t b_bytes = sizeof(double2) * 2 * (size_t)p.K * NT * NT * 256;
    q.bt_in_lds = (img_bytes + b_bytes <= 38 * 1024) ? 1 : 0;     // 4 waves per CU must fit in 160 KB
    const size_t lds = img_bytes + (q.bt_in_lds ? b_bytes : 0);
    if (sandwich) {
        if (keepl) hipLaunchKernelGGL((chain_tile4_kernel<NT, 1, true>), dim3(p.E), dim3(64), lds, stream, q);
        else       hipLaunchKernelGGL((chain_tile4_kernel<NT, 1, false>), dim3(p.E), dim3(64), lds, stream, q);
    } else {
        if (keepl) hipLaunchKernelGGL((chain_tile4_kernel<NT, 0, true>), dim3(p.E), dim3(64), lds, stream, q);
        else       hipLaunchKernelGGL((chain_tile4_kernel<NT, 0, false>), dim3(p.E), dim3(64), lds, stream, q);
    }
    return hipGetLastError();
}

hipError_t launch_sweep_tile4(int n, int sandwich, bool keep_costates, const TileParams &p, hipStream_t stream)
{
    switch (tile_count(n)) {
    case 1: return launch_nt4<1>(sandwich, keep_costates, p, stream);
    case 2: return launch_nt4<2>(sandwich, keep_costates, p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace grape
