// sweep_tile.hip -- the GRAPE hot path for operators of dimension 5..32 on the FP64 matrix
// cores (tile.hpp): one wavefront owns a whole zero-padded (16 NT)^2 complex matrix.
//
//   prop_tile_kernel   pw_prop_save!  (src/timeevolution.jl:98-110): one wave per (member, slice):
//                      H = sum_j B_j x[j,t] + A in D layout straight from the pre-tiled operators,
//                      G = -i dt H, expm_t8 with MFMA products, P_t dumped in D layout.
//   chain_tile_kernel  evolve_func! forward and backward + grad_func! + fom_func
//                      (src/GRAPE.jl:216-287, src/cost_functions.jl:99-111): one wave per member
//                      walks the time axis serially; with E >= #SIMDs the ensemble alone fills the
//                      chip.  The reference's data flow is kept (forward states stored, costates
//                      pulled back in registers).
//   chain_tile_split_kernel / chain_tile_unitary_kernel   the same sweep with two waves per member / for Hermitian
//                      generators (M_t = P' M P, no stored states).
//   chunk_product_kernel, chunk_scan_*_kernel   SMALL ensembles (single problems): the time axis in chunks, one wave
//                      per (member, chunk); the chain kernels then run with grid.z = chunk ("Time-parallel ..." below).
//
// Products are arranged so that every operand is either a lane-contiguous dump load or the D
// registers of the running matrix (tile.hpp):
//   forward  UG:  X' = P X                 = tmul_an(P_A, X)
//            ST:  X' = P X P'  ->  Y^T = X^T P^T = tmul_tb(X, P_A);  X' = Y P' = tmul_tb<conj B>(Y^T, P_A)
//   backward UG:  L' = P' L                = tmul_tn<conj Z>(P, L)      (P^H as A operand: free)
//            ST:  Y^T = L^T conj(P) = tmul_tn<., conj W>(L, P);  L' = Y P = tmul_tn(Y^T, P)
//   gradient:     R = X L' (one A-layout conversion each of X and L), sandwich: minus L' X
//                 = tmul_tn<conj Z>(L, X);  tr(L' B_c X) = sum R .* (B_c^T in D layout).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "cmat.hpp"          // Taylor-8 coefficients, squarings_for
#include "grape_kernels.hpp"
#include "tile.hpp"
#include "done_signal.hpp"

namespace grape {

// ---------------------------------------------------------------------------------------------
// slices per workgroup: for NT = 1 the member's K+1 generator tiles are staged in LDS once per
// workgroup and every wave walks kPropSlices/4 slices with them (read from L2 once per 64 slices
// instead of once per slice -- the kernel was L2-bandwidth bound: 24 KB of operators per 4 KB of P)
constexpr int kPropSlices = 64;

// Four waves per workgroup, one slice each.  NT = 2 runs them one per SIMD with the whole register file: eight waves
// (two per SIMD, <= 256 registers each at the price of ~70 spilled VGPRs, the second wave's MFMAs covering the first
// one's H build, layout conversions, norm and Taylor combinations) was the better choice with four-product complex
// multiplication (137 vs 147 ms at C5) and is not with three (166 vs 133 ms: the third accumulator set spills).
constexpr int kPropWaves = 4;
template <int NT>
__global__ __launch_bounds__(64 * kPropWaves, NT == 1 ? 4 : 1) void prop_tile_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;                     // double2 per matrix dump
    constexpr int WPB = kPropWaves;
    constexpr int NIMG = NT;                               // LDS conversion images per wave: a row of tiles per pass
    const bool STAGE = p.stage_ops != 0;                   // set by the launcher when the generator tiles fit in LDS
    extern __shared__ double2 s_prop[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.y;
    const int K = p.K;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;   // [A | B_c | B_c^T | Xi | Xt]
    double2 *img = s_prop + (size_t)wave * NIMG * kTileImage;
    double2 *s_ops = s_prop + WPB * NIMG * kTileImage;     // STAGE: [A | B_1..B_K]
    // FUSE (rank-one chain, NT = 1): this workgroup walks every slice of the member, wave w the slices t = w mod WPB,
    // and the forward vector v_t is handed from wave to wave through LDS: [v even | v odd | flag]
    const bool FUSE = NT == 1 && p.fuse_fwd != 0;
    double2 *s_vec = s_ops + (STAGE ? (size_t)(K + 1) * TSZ : 0);
    volatile int *s_flag = reinterpret_cast<volatile int *>(s_vec + 32);
    if (STAGE) {
        for (int i = threadIdx.x; i < (K + 1) * TSZ; i += 64 * WPB)
            s_ops[i] = ops[i];
    }
    if (FUSE) {
        if (threadIdx.x < 16)
            s_vec[threadIdx.x] = p.vecs[(size_t)k * 32 + threadIdx.x];
        if (threadIdx.x == 0)
            *s_flag = 0;
    }
    if (STAGE || FUSE)
        __syncthreads();
    const int t_lo = FUSE ? 0 : blockIdx.x * p.prop_slices;
    const int t_hi = FUSE ? p.N : min(p.N, t_lo + p.prop_slices);
  for (int t = t_lo + wave; t < t_hi; t += WPB) {
    TMat<NT> G;
    if (p.variant == 0)
        tzero(G);
    else if (STAGE)
        tload(G, s_ops, lane);
    else
        tload(G, ops, lane);
    for (int c = 0; c < K; ++c) {
        const double xv = p.x[(size_t)blockIdx.z * K * p.N + c + (size_t)t * K];
        TMat<NT> B;
        if (STAGE)
            tload(B, s_ops + (size_t)(1 + c) * TSZ, lane);
        else
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
        if (STAGE)
            tload(A, s_ops, lane);
        else
            tload(A, ops, lane);
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                G.re[I][J] += A.re[I][J];
                G.im[I][J] += A.im[I][J];
            }
    }
    // G = (-i dt) H ; column sums of |re|+|im| bound the 1-norm
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
        cs = swap16_add(cs, cs);                           // rows live on lane>>4 and r
        cs = swap32_add(cs, cs);
        colmax = fmax(colmax, cs);
    }
    colmax = wave_max_fast(colmax);
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

    auto conv = [&](TOp<NT> &o, const TMat<NT> &z, double2 *im, int ln) {
        to_a_layout_rows(o, z, im, ln);
    };
    // expm_t8 (cmat.hpp) with MFMA products (three-product complex multiplication, tile.hpp); every matrix is a
    // polynomial in G
    TOp<NT> opa;
    TMat<NT> A2, A4, U, T;
    conv(opa, G, img, lane);
    tmul_an<NT, false, false>(A2, opa, G);                 // A2 = G G
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            T.re[I][J] = kX1 * G.re[I][J] + kX2 * A2.re[I][J];
            T.im[I][J] = kX1 * G.im[I][J] + kX2 * A2.im[I][J];
        }
    conv(opa, A2, img, lane);
    tmul_an<NT, false, false>(A4, opa, T);                 // A4 = A2 (x1 G + x2 A2)
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            U.re[I][J] = kX3 * A2.re[I][J] + A4.re[I][J];
            U.im[I][J] = kX3 * A2.im[I][J] + A4.im[I][J];
            T.re[I][J] = kX5 * G.re[I][J] + kX6 * A2.re[I][J] + kX7 * A4.re[I][J];
            T.im[I][J] = kX5 * G.im[I][J] + kX6 * A2.im[I][J] + kX7 * A4.im[I][J];
        }
    // identity: element (row, col) with row == col  <=>  I == J and 4r + (lane>>4) == lane&15
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15))
                T.re[I][I][r] += kX4;
    conv(opa, U, img, lane);
    TMat<NT> P;
    tmul_an<NT, false, false>(P, opa, T);                  // A8
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            P.re[I][J] += G.re[I][J] + kY2 * A2.re[I][J];
            P.im[I][J] += G.im[I][J] + kY2 * A2.im[I][J];
        }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15))
                P.re[I][I][r] += 1.0;
    for (int i = 0; i < s; ++i) {
        conv(opa, P, img, lane);
        tmul_an<NT, false, false>(T, opa, P);
        P = T;
    }
    if (NT == 1 && p.thin == 1 && (t & 1)) {
        // rank-one chain (sweep_thin.hip): odd slices are stored transposed -- the A-operand layout of P is the
        // D layout of P^T -- so that its matrix-vector products never convert between vector formats
        conv(opa, P, img, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            P.re[0][0][r] = opa.re[0][0][r];
            P.im[0][0][r] = opa.im[0][0][r];
        }
    }
    if constexpr (NT == 1) {
        if (FUSE && p.fuse_fwd == 1) {
            // v_{t+1} = P_t v_t as chain_thin_kernel's forward pass does it (same register contents: the D layout of
            // P_t for even t, of P_t^T for odd t; same summation trees), the vector taken from / handed to the
            // neighbouring waves through LDS.  All WPB waves are resident, so the wait cannot starve.  The hand-over
            // is the serial part of this kernel (N steps per member): nothing but LDS traffic may sit between the
            // flag read and the flag write -- no memory fence (it would wait for the global stores), the stores of
            // P_t and of the record come afterwards.  A wave's LDS operations complete in order.
            const int g = lane >> 4, c = lane & 15;
            const double2 *vb = s_vec + (t & 1) * 16;
            double2 *vn = s_vec + ((t + 1) & 1) * 16;
            double2 *__restrict__ V = p.states + ((size_t)blockIdx.z * p.E + k) * (size_t)(p.N + 1) * 16;
            while (*s_flag != t)
                __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const double2 rec = vb[c];                     // per-column: x[c]
            double2 out_rec = make_double2(0.0, 0.0);
            if ((t & 1) == 0) {                            // per-column in, gathered out
                double y[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[2 * r] = fma(P.re[0][0][r], rec.x, -P.im[0][0][r] * rec.y);
                    y[2 * r + 1] = fma(P.re[0][0][r], rec.y, P.im[0][0][r] * rec.x);
                }
                row_sum_n(y);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c == r)
                        vn[4 * r + g] = make_double2(y[2 * r], y[2 * r + 1]);
            } else {                                       // gathered in, per-column out
                double acc[2] = {0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 v = vb[4 * r + g];
                    acc[0] = fma(P.re[0][0][r], v.x, acc[0]);
                    acc[0] = fma(-P.im[0][0][r], v.y, acc[0]);
                    acc[1] = fma(P.re[0][0][r], v.y, acc[1]);
                    acc[1] = fma(P.im[0][0][r], v.x, acc[1]);
                }
                col_sum_n(acc);
                if (g == 0)
                    vn[c] = make_double2(acc[0], acc[1]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            asm volatile("" ::: "memory");
            if (t == p.N - 1)
                out_rec = vn[c];
            if (lane == 0)
                *s_flag = t + 1;
            if (g == 0) {
                V[(size_t)t * 16 + c] = rec;
                if (t == p.N - 1)
                    V[(size_t)p.N * 16 + c] = out_rec;
            }
        }
    }
    tstore(p.props + (((size_t)blockIdx.z * p.E + k) * p.N + t) * TSZ, P, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// Gradient traces from sparse control operators (TileParams.sparse): R (the matrix whose trace with B_c is wanted:
// M_t, or [X_t, L_t'] / X_t L_t') goes to the wave's LDS image once, every lane picks the entry of its list position,
// and one reduce-scatter per eight controls delivers  out_t[c] = gs * sum_nz (SAND ? Im : Im(. * z)) .
template <int NT, int SAND>
GRAPE_DEV void sparse_traces(const TMat<NT> &R, double2 *__restrict__ s_M, const double2 *__restrict__ s_coef,
                             const int *__restrict__ s_addr, int K, double zr, double zi, double gs,
                             double *__restrict__ out_t, int lane, bool writer_ok, int nz, double *__restrict__ fold_t = nullptr,
                             double fold_w = 0.0)
{
    constexpr int MS = 16 * NT + 1;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_M[(16 * I + 4 * r + (lane >> 4)) * MS + 16 * J + (lane & 15)] = make_double2(R.re[I][J][r], R.im[I][J][r]);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int c0 = 0; c0 < K; c0 += 8) {                            // eight controls per reduce-scatter (upper half zero)
        double q16[16];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            q16[cc] = 0.0;
            q16[8 + cc] = 0.0;
            if (c0 + cc < K) {
                const double2 cf = s_coef[(c0 + cc) * nz + lane];
                const double2 mv = s_M[s_addr[(c0 + cc) * nz + lane]];
                const double pr = cf.x * mv.x - cf.y * mv.y, pi = cf.x * mv.y + cf.y * mv.x;
                q16[cc] = SAND ? pi : fma(pr, zi, pi * zr);
                for (int e = lane + 64; e < nz; e += 64) {          // lists longer than a wavefront (nz = 128, 192, 256)
                    const double2 cf2 = s_coef[(c0 + cc) * nz + e];
                    const double2 mv2 = s_M[s_addr[(c0 + cc) * nz + e]];
                    const double pr2 = cf2.x * mv2.x - cf2.y * mv2.y, pi2 = cf2.x * mv2.y + cf2.y * mv2.x;
                    q16[cc] += SAND ? pi2 : fma(pr2, zi, pi2 * zr);
                }
            }
        }
        const double tot = reduce_scatter16(q16);
        const int c = c0 + (lane >> 2);
        if (writer_ok && (lane & 3) == 0 && lane < 32 && c < K) {
            out_t[c] = gs * tot;
            if (fold_t)                                            // one problem: this kernel closes the evaluation
                fold_t[c] = fma(gs * tot, fold_w, 0.0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                            // the image is overwritten by the next slice
    __builtin_amdgcn_wave_barrier();
}

// sparse-list staging: [coefficients K * sp_nz double2 | image of M | positions K * sp_nz int]
template <int NT>
GRAPE_DEV void stage_sparse_lists(const TileParams &p, int k, int lane, int nthreads, double2 *s_coef, int *s_addr)
{
    const double2 *__restrict__ gc = p.sp_coef + (size_t)k * p.K * p.sp_nz;
    const int32_t *__restrict__ ga = p.sp_addr + (size_t)k * p.K * p.sp_nz;
    for (int i = lane; i < p.K * p.sp_nz; i += nthreads) {
        s_coef[i] = gc[i];
        s_addr[i] = ga[i];
    }
}

// ---------------------------------------------------------------------------------------------
template <int NT, int SAND, bool KEEPL, bool PACK2, bool SPARSE = false>
__global__ __launch_bounds__(64) void chain_tile_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    // LDS: the layout-conversion image, then (when it was given room at launch) this member's
    // K transposed control operators, read K times per slice by the gradient traces
    extern __shared__ double2 s_dynt[];
    double2 *s_img = s_dynt;
    double2 *s_bt = s_dynt + kTileImage + 1;                       // SPARSE: coefficients | image | positions instead
    double2 *s_coef = s_bt;
    double2 *s_M = s_coef + (size_t)p.K * p.sp_nz;
    int *s_addr = reinterpret_cast<int *>(s_M + 16 * NT * (16 * NT + 1));
    const int lane = threadIdx.x;
    const int k = blockIdx.x;
    const int K = p.K, N = p.N;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ opBT = ops + (size_t)(1 + K) * TSZ;
    const bool bt_lds = p.bt_in_lds != 0;
    if (SPARSE) {
        stage_sparse_lists<NT>(p, k, lane, 64, s_coef, s_addr);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    } else if (bt_lds) {
        for (int i = lane; i < K * TSZ; i += 64)
            s_bt[i] = opBT[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    const size_t kw = (size_t)blockIdx.y * p.E + k;                // workspace row: (control array, unit)
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    double2 *__restrict__ Xk = p.states + kw * N * TSZ;
    // pack2: this wave carries members 2k (tile rows/cols 0..7) and 2k+1 (8..15); after the
    // cross-lane sums lane 0 holds the first member's values and lane 8 the second's
    constexpr bool pack2 = PACK2;
    const int member = pack2 ? 2 * k + ((lane >> 3) & 1) : k;
    const bool writer = (pack2 ? (lane == 0 || lane == 8) : lane == 0) && member < p.E_members;
    double *__restrict__ out = p.member_out + ((size_t)blockIdx.y * p.E_members + member) * ((size_t)K * N + 1);

    // time-parallel mode (TileParams.tp_chunks, small ensembles): this wavefront owns the slices [t_lo, t_hi); the
    // state at t_lo is U Xi [U'] with U the product of the chunks before, the costate at t_hi R' Xt [R] with R the
    // product of the chunks after (chunk_scan_general_kernel)
    const int C = p.tp_chunks;
    const int t_lo = C ? (int)blockIdx.z * p.tp_S : 0, t_hi = C ? min(N, t_lo + p.tp_S) : N;
    // ------------------------------------------------------------ forward sweep
    {
        TMat<NT> X, Pm, Y;
        TOp<NT> PA;
        TMat<NT> Pn;
        tload(X, ops + (size_t)(1 + 2 * K) * TSZ, lane);           // Xi
        if (C) {
            TMat<NT> Ut;
            auto apply = [&](const double2 *__restrict__ src) {    // X <- U X [U'],  src = the D-layout dump of U^T
                tload(Ut, src, lane);
                tmul_tn<NT, false, false>(Y, Ut, X);
                X = Y;
                if (SAND) {
                    to_a_layout(PA, X, s_img, lane);
                    tmul_an<NT, false, true>(Y, PA, Ut);
                    X = Y;
                }
            };
            if (p.tp_groups)                                       // two-level scan: U = U_local B_group
                apply(p.tp_a + ((2 * (size_t)gridDim.y * p.E + kw) * p.tp_groups + blockIdx.z / p.tp_gsize) * TSZ);
            apply(p.tp_u + (kw * C + blockIdx.z) * TSZ);
        }
        tload(Pm, Pk + (size_t)t_lo * TSZ, lane);
        for (int t = t_lo; t < t_hi; ++t) {
            tstore(Xk + (size_t)t * TSZ, X, lane);
            tload(Pn, Pk + (size_t)min(t + 1, t_hi - 1) * TSZ, lane);   // next slice's P in flight (clamped, not branched round:
                                                                        //  chain_tile_split_kernel's note)
            if (t + 1 < t_hi) {                                    // the state after the last slice is never read
                to_a_layout(PA, Pm, s_img, lane);
                if (SAND) {
                    tmul_tb<NT, false, false>(Y, X, PA);           // (P X)^T
                    tmul_tb<NT, false, true>(X, Y, PA);            // (P X) P'
                } else {
                    tmul_an<NT, false, false>(Y, PA, X);
                    X = Y;
                }
            }
            Pm = Pn;
        }
    }

    // ------------------------------------------------------------ backward sweep + gradient
    TMat<NT> L, Pm, X, Y, R, Pn, Xn;
    TOp<NT> XA, LA;
    tload(L, ops + (size_t)(2 + 2 * K) * TSZ, lane);               // Xt
    if (C) {
        auto pull = [&](const double2 *__restrict__ src) {         // L <- R' L [R]
            tload(Pm, src, lane);
            if (SAND) {
                tmul_tn<NT, false, true>(Y, L, Pm);
                tmul_tn<NT, false, false>(L, Y, Pm);
            } else {
                tmul_tn<NT, true, false>(Y, Pm, L);
                L = Y;
            }
        };
        if (p.tp_groups)                                           // two-level scan: R = A_group R_local
            pull(p.tp_a + (kw * p.tp_groups + blockIdx.z / p.tp_gsize) * TSZ);
        pull(p.tp_r + (kw * C + blockIdx.z) * TSZ);
    }
    // (Hermitian states and control operators: Im tr(B [X, L']) = 2 Im tr(B X L) -- see chain_tile_split_kernel)
    const bool herm2 = SAND && p.herm_states != 0 && p.herm_ctrl != 0;
    const double gs = SAND ? (herm2 ? -2.0 * p.dt : -p.dt) : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    double z_keep_r = 0.0, z_keep_i = 0.0;
    tload(Pm, Pk + (size_t)(t_hi - 1) * TSZ, lane);
    tload(X, Xk + (size_t)(t_hi - 1) * TSZ, lane);
    for (int t = t_hi - 1; t >= t_lo; --t) {
        {                                                          // next slice's P, X in flight (clamped at the first slice)
            const int tp = max(t - 1, t_lo);
            tload(Pn, Pk + (size_t)tp * TSZ, lane);
            tload(Xn, Xk + (size_t)tp * TSZ, lane);
        }
        if (SAND) {
            tmul_tn<NT, false, true>(Y, L, Pm);                    // (P' L)^T
            tmul_tn<NT, false, false>(L, Y, Pm);                   // P' L P
        } else {
            tmul_tn<NT, true, false>(Y, Pm, L);                    // P' L
            L = Y;
        }
        if (KEEPL)
            tstore(p.costates + (kw * N + t) * TSZ, L, lane);
        // R = X L' : A layout of X, B layout of L' = conj(A layout of L)
        to_a_layout(XA, X, s_img, lane);
        to_a_layout(LA, L, s_img, lane);
        tprod<NT, false, true>(
            R, [&](int I, int Kt, int kb, double &r, double &i) { r = XA.re[I][Kt][kb]; i = XA.im[I][Kt][kb]; },
            [&](int Kt, int J, int kb, double &r, double &i) { r = LA.re[J][Kt][kb]; i = LA.im[J][Kt][kb]; });
        if (SAND && !herm2) {
            tmul_tn<NT, true, false>(Y, L, X);                     // L' X
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    R.re[I][J] -= Y.re[I][J];
                    R.im[I][J] -= Y.im[I][J];
                }
        }
        // all cross-lane sums of this slice are taken together (wave_sum_n): tr(X' L) and, per
        // control, sum R .* B^T
        double zr = 0.0, zi = 0.0;
        if (SPARSE) {
            // tr(X_t' L_t) does not depend on t, not even for non-unitary P: taken once, at the
            // first slice processed; every lane's share of g[c, t] is then one real number
            if (t == t_hi - 1) {
                double zz[2];
                tdot_partial<NT, true>(zz[0], zz[1], X, L);
                wave_sum_n(zz);
                z_keep_r = zz[0];
                z_keep_i = zz[1];
            }
            zr = z_keep_r;
            zi = z_keep_i;
            sparse_traces<NT, SAND>(R, s_M, s_coef, s_addr, K, zr, zi, gs, out + (size_t)t * K, lane, member < p.E_members, p.sp_nz);
        } else
        for (int c0 = 0; c0 < K; c0 += 4) {
            double v[2 + 8];
            tdot_partial<NT, true>(v[0], v[1], X, L);              // tr(X' L)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = c0 + cc;
                v[2 + 2 * cc] = 0.0;
                v[3 + 2 * cc] = 0.0;
                if (c < K) {
                    TMat<NT> BT;
                    if (bt_lds)
                        tload(BT, s_bt + (size_t)c * TSZ, lane);
                    else
                        tload(BT, opBT + (size_t)c * TSZ, lane);
                    tdot_partial<NT, false>(v[2 + 2 * cc], v[3 + 2 * cc], BT, R);   // sum_ij B[i,j] R[j,i]
                }
            }
            wave_sum_n(v, pack2);
            zr = v[0];
            zi = v[1];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = c0 + cc;
                const double wr = v[2 + 2 * cc], wi = v[3 + 2 * cc];
                const double im = SAND ? wi : fma(wr, zi, wi * zr);
                if (c < K && writer)
                    out[c + (size_t)t * K] = gs * im;
            }
        }
        if (t == N - 1 && writer) {
            if (SAND) {
                const double inv = 1.0 / (double)p.n;
                const double ar = zr * inv, ai = zi * inv;
                out[(size_t)K * N] = 1.0 - (ar * ar + ai * ai);
            } else {
                out[(size_t)K * N] = zr * zr - zi * zi;
            }
        }
        Pm = Pn;
        X = Xn;
    }
}

// ---------------------------------------------------------------------------------------------
// Two wavefronts per member that MEET IN THE MIDDLE of the time axis (n <= 16, general flow, ensembles too small to put two
// members on every SIMD -- C4 with full-rank states).  The costate chain L_t = P_t' L_{t+1} [P_t] needs no forward state, so
// it can start at t = N while the forward chain starts at t = 0:
//   phase 1   wave 0: t = 0 .. Nh-1      X_t stored, X_{t+1} = P_t X_t [P_t']                       (2 products / slice, UG 1)
//             wave 1: t = N-1 .. Nh      L_t = P_t' L_{t+1} [P_t], L_t stored                       (2 products / slice, UG 1)
//   (one workgroup barrier: the stored states are visible; nothing is exchanged -- each wave keeps its own chain)
//   phase 2   wave 0: t = Nh .. N-1      L_t loaded, gradient from (X_t, L_t), X_{t+1} = P_t X_t [P_t']
//             wave 1: t = Nh-1 .. 0      X_t loaded, L_t = P_t' L_{t+1} [P_t], gradient from (X_t, L_t)
// -- the reference's general flow (src/GRAPE.jl:216-287) product for product: 5 per slice (UG 3), the same for both waves, so
// Nh = N / 2.  (Rounds 2-4 split the axis differently: the second wave stored prefix products V_j of its part and rebuilt
// X_t = V_j X_Nh V_j' -- 5.45 products per slice, an exchange of two matrices, a third wave measured slower; round 1 split
// every product's real / imaginary part over two waves at a barrier + LDS swap per product: 10.1 ms per C4dense evaluation.)
// When Xi, Xt are Hermitian (checked on the host; density operators), X_t and L_t stay Hermitian under P X P' / P' L P:
//   * the A-operand image of X is the D layout of X^T = conj(X) and L' = L is a right operand as it stands: the product
//     Y = X L takes both from the registers they are in (no LDS layout conversion);
//   * [X, L'] = Y - Y', one conversion (to_a_layout of Y IS the D layout of Y^T) instead of a second product -- and with
//     Hermitian control operators tr(B Y') = conj(tr(B Y)), i.e. Im tr(B [X, L]) = 2 Im tr(B Y): Y' is never formed.
// HERM / SPARSE are template arguments, and no vector-memory instruction of the slice loops sits behind a branch: at a join
// the compiler's s_waitcnt pass has to assume the path that issued nothing (round 4 found a wait for the prefetch just
// issued behind `if (t + 1 < N) load`; a lane-0 store behind a branch costs 0.6-1.9 k cycles per slice against a store by
// all lanes, tools/ubench/branch_wait.hip).  Every load is issued RG - 1 (RB - 1) slices ahead of its use into a ring of
// register buffers; the loops are unrolled by the ring size (a buffer is a fixed set of registers; `Pm = Pn` at the end of
// an iteration, as rounds 2-4 had it, is a wait for the load issued one product earlier).  The waves' own
// cycle counts (tools/split_stamps.py, profiles/r05_split_stamps.txt) before / after: 23-29 % of a wave's cycles waiting
// for P_t in pass 1, 3.7 k cycles per slice in the generic list traces.  What is left: phase 1 is bound by HBM (8 KB per
// slice and wave, half of it stores).
#ifndef GRAPE_SPLIT_ABL
#define GRAPE_SPLIT_ABL 0
#endif
// A Hermitian 16 x 16 state in 3 KB instead of 4: rows 0..7 as they are (D registers 0, 1), of rows 8..15 only columns 8..15
// -- register 2's upper half-rows in the lanes that hold them, register 3's rotated into the other lanes (row_ror:8) -- and the
// block (rows 8..15, columns 0..7) comes back as the conjugate transpose of (rows 0..7, columns 8..15) through the wave's LDS
// image (two writes, two reads).  The stored states are a quarter of the split chain's HBM bytes, which is what bounds it.
struct HState {
    double re[3], im[3];
};
GRAPE_DEV double ror8(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x128, 0xF, 0xF, true);      // row_ror:8
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x128, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
GRAPE_DEV void hstore(double2 *__restrict__ dst, const TMat<1> &m, int lane)
{
    const bool up = (lane & 8) != 0;                               // column >= 8
    const double qr = ror8(m.re[0][0][3]), qi = ror8(m.im[0][0][3]);
    dst[lane] = make_double2(m.re[0][0][0], m.im[0][0][0]);
    dst[64 + lane] = make_double2(m.re[0][0][1], m.im[0][0][1]);
    dst[128 + lane] = make_double2(up ? m.re[0][0][2] : qr, up ? m.im[0][0][2] : qi);
}
GRAPE_DEV void hload(HState &h, const double2 *__restrict__ src, int lane)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double2 v = src[r * 64 + lane];
        h.re[r] = v.x;
        h.im[r] = v.y;
    }
}
GRAPE_DEV void hunpack(TMat<1> &m, const HState &h, double2 *__restrict__ img, int lane)
{
    const int hi = lane >> 4, col = lane & 15;
    const bool up = (lane & 8) != 0;
    img[(hi) * 17 + col] = make_double2(h.re[0], h.im[0]);         // rows 0..3
    img[(4 + hi) * 17 + col] = make_double2(h.re[1], h.im[1]);     // rows 4..7
    const double qr = ror8(h.re[2]), qi = ror8(h.im[2]);           // (lanes with column >= 8: register 3's entry)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const double2 t2 = img[(col & 7) * 17 + 8 + hi], t3 = img[(col & 7) * 17 + 12 + hi];      // X[col][8 + hi], X[col][12 + hi]
    m.re[0][0][0] = h.re[0];
    m.im[0][0][0] = h.im[0];
    m.re[0][0][1] = h.re[1];
    m.im[0][0][1] = h.im[1];
    m.re[0][0][2] = up ? h.re[2] : t2.x;
    m.im[0][0][2] = up ? h.im[2] : -t2.y;
    m.re[0][0][3] = up ? qr : t3.x;
    m.im[0][0][3] = up ? qi : -t3.y;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}
// EXPM (member-invariant control operators, TileParams.split_expm): phase 1 FORMS the propagators it needs instead of
// reading them -- G_t = A'_k + Gc_t from the pre-pass's control sum (shared by all members: served by L2 / Infinity Cache),
// the degree-8 Taylor polynomial in three products + squarings exactly as prop_hoist1_kernel evaluates it -- and stores P_t
// for the other wave's phase 2.  No expm kernel runs: every P_t is written once and read once (18 -> 14 KB of HBM traffic
// per member and slice), and the expm's products fill the matrix pipe of the phase that waits for HBM.
template <int SAND, int SPARSE = 0, int HERM = 0, bool EXPM = false>
__global__ __launch_bounds__(128, 2) void chain_tile_split_kernel(const TileParams p)
{
    constexpr int NT = 1, TSZ = 256, PARTS = 2;
    extern __shared__ double2 s_dynt[];
#ifdef GRAPE_SPLIT_STAMP            // diagnostic build (tools/split_stamps.py): cycles between points of the slice loops, summed per wave
    long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
#define ST_BEGIN() st_last = __builtin_readcyclecounter()
#define ST_MARK(i)                                            \
    {                                                         \
        __builtin_amdgcn_sched_barrier(0);                    \
        const long long now_ = __builtin_readcyclecounter();  \
        __builtin_amdgcn_sched_barrier(0);                    \
        st_acc[i] += now_ - st_last;                          \
        st_last = now_;                                       \
    }
#define ST_DEP(x)                                              \
    {                                                          \
        int lo_ = __double2loint(x);                           \
        __builtin_amdgcn_sched_barrier(0);                     \
        asm volatile("v_mov_b32 %0, %0" : "+v"(lo_));          \
        __builtin_amdgcn_sched_barrier(0);                     \
    }
#else
#define ST_BEGIN()
#define ST_MARK(i)
#define ST_DEP(x)
#endif
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double2 *s_img = s_dynt + (size_t)part * (kTileImage + 1);
    double2 *s_bt = s_dynt + (size_t)PARTS * (kTileImage + 1);     // SPARSE: coefficients | one image of R per wave | positions
    double2 *s_coef = s_bt;
    double2 *s_M = s_coef + (size_t)p.K * p.sp_nz + (size_t)part * (16 * 17);
    int *s_addr = reinterpret_cast<int *>(s_coef + (size_t)p.K * p.sp_nz + PARTS * (16 * 17));
    const int k = blockIdx.x;
    const int K = p.K, N = p.N;
    const int Nh = p.split_at;                                     // wave 0 stores X_t for t < Nh, wave 1 stores L_t for t >= Nh
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ opBT = ops + (size_t)(1 + K) * TSZ;
    const bool bt_lds = p.bt_in_lds != 0;
    constexpr bool herm = SAND && HERM >= 1;
    constexpr bool herm2 = SAND && HERM == 2;
    if (SPARSE) {
        stage_sparse_lists<NT>(p, k, (int)threadIdx.x, 64 * PARTS, s_coef, s_addr);
    } else if (bt_lds) {
        for (int i = threadIdx.x; i < K * TSZ; i += 64 * PARTS)
            s_bt[i] = opBT[i];
    }
    const size_t kw = (size_t)blockIdx.y * p.E + k;                // workspace row: (control array, member)
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    double2 *__restrict__ Pw = p.props + kw * N * TSZ;             // (EXPM: phase 1 writes them)
    double2 *__restrict__ Sk = p.states + kw * N * TSZ;            // slot t: X_t (t < Nh) or L_t (t >= Nh)
    double *__restrict__ out = p.member_out + kw * ((size_t)K * N + 1);
#ifndef GRAPE_SPLIT_RG
#define GRAPE_SPLIT_RG 4
#endif
#ifndef GRAPE_SPLIT_RB
#define GRAPE_SPLIT_RB 3
#endif
    constexpr int RG = GRAPE_SPLIT_RG, RB = GRAPE_SPLIT_RB;        // ring sizes of phase 1 / phase 2 (which keeps more matrices live)
    const bool fwd = part == 0;

    TMat<1> C, Y;                                                  // this wave's chain: X (wave 0) / L (wave 1)
    tload(C, ops + (size_t)(fwd ? 1 + 2 * K : 2 + 2 * K) * TSZ, lane);      // Xi / Xt
    auto push = [&](const TMat<1> &Pt) {                           // X <- P X [P']
        TOp<1> PA;
        to_a_layout(PA, Pt, s_img, lane);
        if (SAND) {
            tmul_tb<NT, false, false>(Y, C, PA);                   // (P X)^T
            tmul_tb<NT, false, true>(C, Y, PA);                    // (P X) P'
        } else {
            tmul_an<NT, false, false>(Y, PA, C);
            C = Y;
        }
    };
    auto pull = [&](const TMat<1> &Pt) {                           // L <- P' L [P]
        if (SAND) {
            tmul_tn<NT, false, true>(Y, C, Pt);                    // (P' L)^T
            tmul_tn<NT, false, false>(C, Y, Pt);                   // P' L P
        } else {
            tmul_tn<NT, true, false>(Y, Pt, C);                    // P' L
            C = Y;
        }
    };
    const int d = fwd ? 1 : -1;                                    // this wave's direction on the time axis
    // stored states: Hermitian ones packed (3 KB of the 4 KB slot)
    constexpr bool hpack = herm && !(GRAPE_SPLIT_ABL & 4);
    using SState = typename std::conditional<hpack, HState, TMat<1>>::type;
    auto sstore = [&](double2 *dst, const TMat<1> &m) {
        if constexpr (hpack) hstore(dst, m, lane);
        else tstore(dst, m, lane);
    };
    auto sload = [&](SState &h, const double2 *src) {
        if constexpr (hpack) hload(h, src, lane);
        else tload(h, src, lane);
    };

    // ------------------------------------------------------------ phase 1
    {
        TMat<1> Pb[RG];                                            // ring: P_t -- or, EXPM, the control sums Gc_t
        TMat<1> Ah;                                                // EXPM: A'_k = (-i dt) A_k
        double nA = 0.0;
        const double2 *__restrict__ src = EXPM ? p.gc + (size_t)blockIdx.y * N * TSZ : Pk;
        const double *__restrict__ gcn = p.gcn + (size_t)blockIdx.y * N;
        double sk = 1.0;
        if constexpr (EXPM) {
            tload(Ah, p.ha + (size_t)k * TSZ, lane);
            nA = p.ha_norm[k];
            if (p.ctrl_scale)
                sk = p.ctrl_scale[k];
        }
        const int a = fwd ? 0 : N - 1, cnt = fwd ? Nh : N - Nh, last = a + d * (cnt - 1);
        auto clampt = [&](int t) { return fwd ? min(t, last) : max(t, last); };
        // P_t = exp(G_t), G_t = A'_k + Gc_t: the constants and their arrangement are prop_hoist1_kernel's
        //   A4' = A2 (A2 + c1 G),  A8' = (A4' + c3 A2)(x4 I + x5 G + x6 A2 + c7 A4'),  P = x2 A8' + (I + G + y2 A2)
        auto expm_t = [&](TMat<1> &Pt, const TMat<1> &Gc, int t) {
            constexpr double c1 = kX1 / kX2, c3 = kX3 / kX2, c7 = kX7 * kX2;
            TMat<1> G, A2, T, A4;
            TOp<1> OA;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                G.re[0][0][r] = fma(sk, Gc.re[0][0][r], Ah.re[0][0][r]);   // (B_k = s_k B_0; s_k = 1: the plain sum, bit for bit)
                G.im[0][0][r] = fma(sk, Gc.im[0][0][r], Ah.im[0][0][r]);
            }
            const int sq = p.s_forced >= 0 ? p.s_forced : squarings_from_ratio(fma(fabs(sk), gcn[t], nA));   // (bounds are stored / theta8)
            if (sq > 0) {
                const double sc = ldexp(1.0, -sq);
                G.re[0][0] *= sc;
                G.im[0][0] *= sc;
            }
            to_a_layout(OA, G, s_img, lane);
            tmul_an<NT, false, false>(A2, OA, G);                  // A2 = G G
            T.re[0][0] = c1 * G.re[0][0] + A2.re[0][0];
            T.im[0][0] = c1 * G.im[0][0] + A2.im[0][0];
            to_a_layout(OA, A2, s_img, lane);
            tmul_an<NT, false, false>(A4, OA, T);                  // A4'
            TMat<1> U;
            U.re[0][0] = c3 * A2.re[0][0] + A4.re[0][0];
            U.im[0][0] = c3 * A2.im[0][0] + A4.im[0][0];
            T.re[0][0] = c7 * A4.re[0][0] + (kX6 * A2.re[0][0] + kX5 * G.re[0][0]);
            T.im[0][0] = c7 * A4.im[0][0] + (kX6 * A2.im[0][0] + kX5 * G.im[0][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + (lane >> 4) == (lane & 15))
                    T.re[0][0][r] += kX4;
            to_a_layout(OA, U, s_img, lane);
            tmul_an<NT, false, false>(Pt, OA, T);                  // A8'
            Pt.re[0][0] = kX2 * Pt.re[0][0] + (kY2 * A2.re[0][0] + G.re[0][0]);
            Pt.im[0][0] = kX2 * Pt.im[0][0] + (kY2 * A2.im[0][0] + G.im[0][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + (lane >> 4) == (lane & 15))
                    Pt.re[0][0][r] += 1.0;
            for (int i = 0; i < sq; ++i) {                         // undo the scaling
                to_a_layout(OA, Pt, s_img, lane);
                tmul_an<NT, false, false>(T, OA, Pt);
                Pt = T;
            }
        };
        auto step = [&](int t, const TMat<1> &In) {
            TMat<1> Pl;
            if constexpr (EXPM) {
                expm_t(Pl, In, t);
                tstore(Pw + (size_t)t * TSZ, Pl, lane);            // for the other wave's phase 2
                ST_DEP(Pl.im[0][0][3])
                ST_MARK(0)
            }
            const TMat<1> &Pt = EXPM ? Pl : In;
            if (fwd) {
                if (!(GRAPE_SPLIT_ABL & 1)) sstore(Sk + (size_t)t * TSZ, C);            // X_t
                ST_MARK(9)
                push(Pt);
            } else {
                pull(Pt);
                ST_DEP(C.im[0][0][3])
                ST_MARK(9)
                if (!(GRAPE_SPLIT_ABL & 1)) sstore(Sk + (size_t)t * TSZ, C);            // L_t
            }
            ST_DEP(C.im[0][0][3])
            ST_MARK(1)
        };
        int t = a, left = cnt;
        ST_BEGIN();
        for (; left % RG; --left, t += d) {                        // the first (cnt mod RG) slices one at a time
            tload(Pb[0], src + (size_t)t * TSZ, lane);
            step(t, Pb[0]);
        }
        if (left > 0) {
#pragma unroll
            for (int i = 0; i < RG - 1; ++i)
                tload(Pb[i], src + (size_t)clampt(t + d * i) * TSZ, lane);
            for (; left > 0; left -= RG, t += d * RG) {
#pragma unroll
                for (int i = 0; i < RG; ++i) {
                    tload(Pb[(i + RG - 1) % RG], src + (size_t)clampt(t + d * (i + RG - 1)) * TSZ, lane);   // (clamped: re-reads the last slice)
                    step(t + d * i, Pb[i]);
                }
            }
        }
    }
    __syncthreads();                                               // the other wave's stored states are complete (and visible: one CU)
    ST_MARK(2)

    // ------------------------------------------------------------ phase 2: the rest of the chain + gradient
    const double gs = SAND ? (herm2 ? -2.0 * p.dt : -p.dt) : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    TMat<1> R;
    bool z_known = false;
    double z_keep_r = 0.0, z_keep_i = 0.0;
    // SPARSE == 2 (K = 4 lists of 64 entries: C4's shape): a lane keeps its four list entries in registers
    double2 cf4[4];
    int ad4[4];
    if constexpr (SPARSE == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cf4[c] = s_coef[c * 64 + lane];
            ad4[c] = s_addr[c * 64 + lane];
        }
    }
    double *s_g = reinterpret_cast<double *>(s_img);               // one slice's K gradient entries on their way out
    // gradient entries of one slice from X_t, L_t (costate after pulling back through slice t)
    auto emit = [&](int t, const TMat<1> &X, const TMat<1> &L) {
        if (SPARSE && !z_known) {
            // tr(X_t' L_t) is the same for every t (also for non-unitary P): taken at the first slice this wave emits
            double zz[2];
            tdot_partial<NT, true>(zz[0], zz[1], X, L);
            wave_sum_n(zz);
            z_keep_r = zz[0];
            z_keep_i = zz[1];
            z_known = true;
        }
        if constexpr (herm) {
            tprod<NT, true, false>(
                R, [&](int, int, int kb, double &r, double &i) { r = X.re[0][0][kb]; i = X.im[0][0][kb]; },
                [&](int, int, int kb, double &r, double &i) { r = L.re[0][0][kb]; i = L.im[0][0][kb]; });        // X L
        } else {
            TOp<1> XA, LA;
            to_a_layout(XA, X, s_img, lane);
            to_a_layout(LA, L, s_img, lane);
            tprod<NT, false, true>(
                R, [&](int I, int Kt, int kb, double &r, double &i) { r = XA.re[I][Kt][kb]; i = XA.im[I][Kt][kb]; },
                [&](int Kt, int J, int kb, double &r, double &i) { r = LA.re[J][Kt][kb]; i = LA.im[J][Kt][kb]; });   // X L'
        }
        ST_DEP(R.im[0][0][3])
        ST_MARK(6)
        if constexpr (SAND) {
            if constexpr (herm2) {
                // (Hermitian states and controls: 2 Im tr(B Y), see above)
            } else if constexpr (herm) {                           // [X, L'] = Y - Y',  Y' = conj(Y^T)
                TOp<1> RT;
                to_a_layout(RT, R, s_img, lane);                   // D layout of Y^T
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    R.re[0][0][r] -= RT.re[0][0][r];
                    R.im[0][0][r] += RT.im[0][0][r];
                }
            } else {
                tmul_tn<NT, true, false>(Y, L, X);                 // L' X
                R.re[0][0] -= Y.re[0][0];
                R.im[0][0] -= Y.im[0][0];
            }
        }
        ST_MARK(7)
        if constexpr (SPARSE == 2) {
            // R to the wave's image, four picks, a reduce-scatter over the four lane rows + one row sum: row c holds control
            // c's trace in all its 16 lanes, and EVERY lane stores it (16 lanes the same 8 bytes: no branch round the store)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_M[(4 * r + (lane >> 4)) * 17 + (lane & 15)] = make_double2(R.re[0][0][r], R.im[0][0][r]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            double q4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double2 mv = s_M[ad4[c]];
                const double pr = cf4[c].x * mv.x - cf4[c].y * mv.y, pi = cf4[c].x * mv.y + cf4[c].y * mv.x;
                q4[c] = SAND ? pi : fma(pr, z_keep_i, pi * z_keep_r);
            }
            double b1[1] = {swap16_add(swap32_add(q4[0], q4[2]), swap32_add(q4[1], q4[3]))};   // row r keeps control r
            row_sum_n(b1);
            out[(size_t)t * 4 + (lane >> 4)] = gs * b1[0];
            __builtin_amdgcn_s_waitcnt(0xc07f);                    // the image is overwritten by the next slice
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (SPARSE == 1) {
            // (the lane number behind an empty asm: the per-lane addresses of the entry lists are recomputed here, a dozen
            // vector instructions, instead of being hoisted out of the slice loop)
            int lane_here = lane;
            asm volatile("" : "+v"(lane_here));
            // the K entries go to LDS (the wave's conversion image is idle here) and from there to HBM by ALL lanes (lanes
            // beyond K repeat entry K - 1): the store sits behind no branch
            sparse_traces<NT, SAND>(R, s_M, s_coef, s_addr, K, z_keep_r, z_keep_i, gs, s_g, lane_here, true, p.sp_nz);
            const int idx = min(lane, K - 1);                      // (K <= 16 with lists)
            out[(size_t)t * K + idx] = s_g[idx];
        } else {
            for (int c0 = 0; c0 < K || c0 == 0; c0 += 4) {
                double v[2 + 8];
                v[0] = 0.0;
                v[1] = 0.0;
                if (!SAND || t == N - 1)
                    tdot_partial<NT, true>(v[0], v[1], X, L);      // tr(X' L)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = c0 + cc;
                    v[2 + 2 * cc] = 0.0;
                    v[3 + 2 * cc] = 0.0;
                    if (c < K) {
                        TMat<NT> BT;
                        if (bt_lds)
                            tload(BT, s_bt + (size_t)c * TSZ, lane);
                        else
                            tload(BT, opBT + (size_t)c * TSZ, lane);
                        tdot_partial<NT, false>(v[2 + 2 * cc], v[3 + 2 * cc], BT, R);
                    }
                }
                wave_sum_n(v);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = c0 + cc;
                    const double wr = v[2 + 2 * cc], wi = v[3 + 2 * cc];
                    const double im = SAND ? wi : fma(wr, v[1], wi * v[0]);
                    if (c < K && lane == 0)
                        s_g[c] = gs * im;                          // (to LDS; stored below by all lanes, no branch round it)
                }
                if (t == N - 1) {
                    z_keep_r = v[0];
                    z_keep_i = v[1];
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int idx = min(lane, K - 1);                      // (K <= 64: the launcher keeps larger K off this kernel)
            out[(size_t)t * K + idx] = s_g[idx];
        }
        ST_MARK(8)
    };
    {
        TMat<1> Pb[RB];
        SState Sb[RB];
        const int a = fwd ? Nh : Nh - 1, cnt = fwd ? N - Nh : Nh, last = a + d * (cnt - 1);
        auto clampt = [&](int t) { return fwd ? min(t, last) : max(t, last); };
        auto step = [&](int t, const TMat<1> &Pt, const SState &Ss) {
            TMat<1> St;
            if constexpr (hpack) hunpack(St, Ss, s_img, lane);
            else St = Ss;
            if (fwd) {
                ST_DEP(St.im[0][0][3])
                ST_MARK(5)
                emit(t, C, St);                                    // (X_t, L_t)
                push(Pt);                                          // X_{t+1}  (the one past N - 1 is not used)
                ST_DEP(C.im[0][0][3])
                ST_MARK(4)
            } else {
                pull(Pt);                                          // L_t
                ST_DEP(C.im[0][0][3])
                ST_MARK(4)
                ST_DEP(St.im[0][0][3])
                ST_MARK(5)
                emit(t, St, C);
            }
        };
        int t = a, left = cnt;
        for (; left % RB; --left, t += d) {
            tload(Pb[0], Pk + (size_t)t * TSZ, lane);
            sload(Sb[0], Sk + (size_t)t * TSZ);
            step(t, Pb[0], Sb[0]);
        }
        if (left > 0) {
#pragma unroll
            for (int i = 0; i < RB - 1; ++i) {
                tload(Pb[i], Pk + (size_t)clampt(t + d * i) * TSZ, lane);
                sload(Sb[i], Sk + (size_t)clampt(t + d * i) * TSZ);
            }
            for (; left > 0; left -= RB, t += d * RB) {
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    tload(Pb[(i + RB - 1) % RB], Pk + (size_t)clampt(t + d * (i + RB - 1)) * TSZ, lane);
                    if (!(GRAPE_SPLIT_ABL & 2)) sload(Sb[(i + RB - 1) % RB], Sk + (size_t)clampt(t + d * (i + RB - 1)) * TSZ);
                    step(t + d * i, Pb[i], Sb[i]);
                }
            }
        }
        if (fwd && cnt > 0 && lane == 0) {                         // the figure of merit: tr(X' L) (wave 0 ends at t = N - 1)
            const double zr = z_keep_r, zi = z_keep_i;
            if (SAND) {
                const double inv = 1.0 / (double)p.n;
                const double ar = zr * inv, ai = zi * inv;
                out[(size_t)K * N] = 1.0 - (ar * ar + ai * ai);
            } else {
                out[(size_t)K * N] = zr * zr - zi * zi;
            }
        }
#ifdef GRAPE_SPLIT_STAMP
        if (lane == 0)
            for (int i = 0; i < 10; ++i)
                out[(size_t)K * (fwd ? Nh : 0) + i] = (double)st_acc[i];     // (rows of this wave's own slices)
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// Unitary flow (every generator Hermitian, so every P_t is unitary; same idea as sweep_small.hip):
// the gradient matrix M_t = X_t L_t' (UnitaryGate) or [X_t, L_t'] (sandwich) obeys
// M_t = P_t' M_{t+1} P_t, so no forward state is stored or re-read.  The forward pass only
// accumulates X_N = P_{N-1} ... P_0 Xi (UnitaryGate) or the total product T (sandwich:
// X_N = T Xi T'), 1 product per slice; the backward pass carries M with 2 products per slice.
// tr(X_t' L_t) is conj(tr M_t) (UnitaryGate) or t-invariant (sandwich, taken at t = N).
// SPARSE (TileParams.sparse): every control operator has at most kSparseMax non-zeros (lists of TileParams.sp_nz entries).  The K dense transposed
// operators (K x 16 KB at NT = 2, re-read by every wave for every slice: 3/4 of this kernel's memory traffic at C5,
// which made it Infinity-Cache-bound) are replaced by K x 64 (coefficient, position) entries staged in LDS; M_t is
// written to an LDS image once per slice and the entries pick what they need.  One reduce-scatter per slice.
template <int NT, int SAND, bool PACK2, bool SPARSE = false>
__global__ __launch_bounds__(64) void chain_tile_unitary_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    constexpr int MS = 16 * NT + 1;                  // row stride of the LDS image of M
    extern __shared__ double2 s_dynt[];
    double2 *s_img = s_dynt;
    double2 *s_bt = s_dynt + kTileImage + 1;         // dense: transposed operators;  sparse: coefficients, M image, positions
    double2 *s_coef = s_bt;
    double2 *s_M = s_coef + (size_t)p.K * p.sp_nz;
    int *s_addr = reinterpret_cast<int *>(s_M + 16 * NT * MS);
    const int lane = threadIdx.x;
    const int k = blockIdx.x;
    const int K = p.K, N = p.N;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ opBT = ops + (size_t)(1 + K) * TSZ;
    const bool bt_lds = p.bt_in_lds != 0;
    if (SPARSE) {
        stage_sparse_lists<NT>(p, k, lane, 64, s_coef, s_addr);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    } else if (bt_lds) {
        for (int i = lane; i < K * TSZ; i += 64)
            s_bt[i] = opBT[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    const double2 *__restrict__ Pk = p.props + ((size_t)blockIdx.y * p.E + k) * N * TSZ;
    constexpr bool pack2 = PACK2;
    const int member = pack2 ? 2 * k + ((lane >> 3) & 1) : k;
    const bool writer = (pack2 ? (lane == 0 || lane == 8) : lane == 0) && member < p.E_members;
    double *__restrict__ out = p.member_out + ((size_t)blockIdx.y * p.E_members + member) * ((size_t)K * N + 1);

    TMat<NT> M, L, Y, Pm, Pn;
    double zr = 0.0, zi = 0.0;
    // time-parallel mode (TileParams.tp_chunks): this wavefront owns the slices [t_lo, t_hi) of its unit; M_N and the
    // product R of everything after the chunk come from chunk_scan_kernel, M at the chunk's end is R' M_N R
    const int C = p.tp_chunks;
    const int t_lo = C ? (int)blockIdx.z * p.tp_S : 0, t_hi = C ? min(N, t_lo + p.tp_S) : N;
    if (C) {
        const size_t kw = (size_t)blockIdx.y * p.E + k;
        tload(M, p.tp_m + kw * TSZ, lane);
        if (SAND) {
            zr = p.tp_z[(kw * 64 + lane) * 2];
            zi = p.tp_z[(kw * 64 + lane) * 2 + 1];
        }
        if (p.tp_groups) {                                         // two-level scan: R = A_group R_local
            tload(Pm, p.tp_a + (kw * p.tp_groups + blockIdx.z / p.tp_gsize) * TSZ, lane);
            tmul_tn<NT, false, true>(Y, M, Pm);
            tmul_tn<NT, false, false>(M, Y, Pm);                   // A' M_N A
        }
        tload(Pm, p.tp_r + (kw * C + blockIdx.z) * TSZ, lane);
        tmul_tn<NT, false, true>(Y, M, Pm);                        // (R' M)^T
        tmul_tn<NT, false, false>(M, Y, Pm);                       // R' M R
    } else
    // ------------------------------------------------------------ forward: X_N = T Xi [T'],  T = P_{N-1} ... P_0
    // The total product is taken from the LAST slice down, V <- P_t^T V: a D-layout matrix is its own transpose as
    // an A operand, so the N products need no layout conversion (they did: 4 LDS round trips per slice at NT = 2,
    // half of this pass's time with one wave per SIMD), and V ends as T^T, which is T as an A operand again.
    {
        TMat<NT> X, V;
        tzero(V);
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + (lane >> 4) == (lane & 15))
                    V.re[I][I][r] = 1.0;
        // two slices in flight: this pass is one product per 16 NT^2 KB streamed, close to HBM-bound
        TMat<NT> Pnn;
        tload(Pm, Pk + (size_t)(N - 1) * TSZ, lane);
        tload(Pn, Pk + (size_t)max(N - 2, 0) * TSZ, lane);
        for (int t = N - 1; t >= 0; --t) {
            tload(Pnn, Pk + (size_t)max(t - 2, 0) * TSZ, lane);
            tmul_tn<NT, false, false>(Y, Pm, V);                   // P_t^T V
            V = Y;
            Pm = Pn;
            Pn = Pnn;
        }
        {
            TMat<NT> Xi;
            tload(Xi, ops + (size_t)(1 + 2 * K) * TSZ, lane);
            tmul_tn<NT, false, false>(X, V, Xi);                   // (T^T)^T Xi = T Xi
        }
        if (SAND) {                                                // X_N = (T Xi) T' = (T Xi) conj(T^T)
            TOp<NT> YA;
            to_a_layout(YA, X, s_img, lane);
            tmul_an<NT, false, true>(Y, YA, V);
            X = Y;
        }
        tload(L, ops + (size_t)(2 + 2 * K) * TSZ, lane);           // L_N = Xt
        // M_N = X_N L_N'  [ - L_N' X_N ]
        TOp<NT> XA, LA;
        to_a_layout(XA, X, s_img, lane);
        to_a_layout(LA, L, s_img, lane);
        tprod<NT, false, true>(
            M, [&](int I, int Kt, int kb, double &r, double &i) { r = XA.re[I][Kt][kb]; i = XA.im[I][Kt][kb]; },
            [&](int Kt, int J, int kb, double &r, double &i) { r = LA.re[J][Kt][kb]; i = LA.im[J][Kt][kb]; });
        if (SAND) {
            tmul_tn<NT, true, false>(Y, L, X);                     // L' X
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    M.re[I][J] -= Y.re[I][J];
                    M.im[I][J] -= Y.im[I][J];
                }
            double zz[2];
            tdot_partial<NT, true>(zz[0], zz[1], X, L);            // tr(X' L), the same for every t
            wave_sum_n(zz, pack2);
            zr = zz[0];
            zi = zz[1];
        }
    }

    // ------------------------------------------------------------ backward: M_t = P_t' M_{t+1} P_t
    const double gs = SAND ? -p.dt : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    if (SPARSE && !SAND) {                                         // z = conj(tr M), the same for every t
        double zz[2] = {0.0, 0.0};
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + (lane >> 4) == (lane & 15)) {
                    zz[0] += M.re[I][I][r];
                    zz[1] += M.im[I][I][r];
                }
        wave_sum_n(zz);
        zr = zz[0];
        zi = -zz[1];
    }
    TMat<NT> Pnn;
    tload(Pm, Pk + (size_t)(t_hi - 1) * TSZ, lane);
    tload(Pn, Pk + (size_t)max(t_hi - 2, 0) * TSZ, lane);
    for (int t = t_hi - 1; t >= t_lo; --t) {
        tload(Pnn, Pk + (size_t)max(t - 2, 0) * TSZ, lane);        // two slices in flight
        tmul_tn<NT, false, true>(Y, M, Pm);                        // (P' M)^T
        tmul_tn<NT, false, false>(M, Y, Pm);                       // P' M P
        if (SPARSE) {
            // tr(B_c M_t) = sum over the non-zeros B_c[i][j] of B_c[i][j] M_t[j][i]; with z = conj(tr M) taken once
            // (the trace is invariant under M -> P' M P) every lane's share of g[c, t] is one real number
            sparse_traces<NT, SAND>(M, s_M, s_coef, s_addr, K, zr, zi, gs, out + (size_t)t * K, lane, true, p.sp_nz,
                                    p.fold_fg ? fold_dst(p) + (size_t)t * K : nullptr, p.fold_fg ? p.fold_wts[0] : 0.0);
        } else
        for (int c0 = 0; c0 < K; c0 += 4) {
            double v[2 + 8];
            v[0] = 0.0;
            v[1] = 0.0;
            if (!SAND) {                                           // tr(M): z_t = conj(tr M_t)
#pragma unroll
                for (int I = 0; I < NT; ++I)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * r + (lane >> 4) == (lane & 15)) {
                            v[0] += M.re[I][I][r];
                            v[1] += M.im[I][I][r];
                        }
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = c0 + cc;
                v[2 + 2 * cc] = 0.0;
                v[3 + 2 * cc] = 0.0;
                if (c < K) {
                    TMat<NT> BT;
                    if (bt_lds)
                        tload(BT, s_bt + (size_t)c * TSZ, lane);
                    else
                        tload(BT, opBT + (size_t)c * TSZ, lane);
                    tdot_partial<NT, false>(v[2 + 2 * cc], v[3 + 2 * cc], BT, M);   // sum_ij B[i,j] M[j,i]
                }
            }
            wave_sum_n(v, pack2);
            if (!SAND) {
                zr = v[0];
                zi = -v[1];
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = c0 + cc;
                const double wr = v[2 + 2 * cc], wi = v[3 + 2 * cc];
                const double im = SAND ? wi : fma(wr, zi, wi * zr);
                if (c < K && writer) {
                    out[c + (size_t)t * K] = gs * im;
                    fold_store(p, c + (size_t)t * K, gs * im);
                }
            }
        }
        if (t == N - 1 && writer) {
            double Fk;
            if (SAND) {
                const double inv = 1.0 / (double)p.n;
                const double ar = zr * inv, ai = zi * inv;
                Fk = 1.0 - (ar * ar + ai * ai);
            } else {
                Fk = zr * zr - zi * zi;
            }
            out[(size_t)K * N] = Fk;
            fold_store(p, (size_t)K * N, Fk);
        }
        Pm = Pn;
        Pn = Pnn;
    }
    if (lane == 0)                                                 // (single-wave workgroups)
        fold_publish(p);
}

// ---------------------------------------------------------------------------------------------
// Time-parallel unitary chain for SMALL ensembles (fewer units than SIMDs: the sequential chain above leaves the
// device idle and takes N x 3 dependent products -- 27 ms for one 32 x 32 problem of 2000 slices).  The same idea as
// the chunk scan of sweep_small.hip / sweep_pair.hip, on tiles:
//   chunk_product_kernel   one wavefront per (unit, chunk): Q_c = P_hi-1 ... P_lo        (S products, in parallel)
//   chunk_scan_kernel      one wavefront per unit: R_c = Q_C-1 ... Q_c+1 for every c, T = R_-1, then X_N, M_N, z
//                          exactly as the sequential kernel forms them                    (C products, serial)
//   chain_tile_unitary_kernel with grid.z = C: M at the end of chunk c = R_c' M_N R_c, then the backward sweep
//                          and the gradient traces over the chunk's own slices           (2 S products, in parallel)
// One more product per slice than the sequential chain, 3 S + C dependent products instead of 3 N.
// Transposes are free as A operands (a D-layout dump of Z is Z^T as the left factor), so every product here is
// Z^T W on D-layout registers; the chunk kernels convert once at the end to hand out Q_c and R_c untransposed.
template <int NT>
GRAPE_DEV void transpose_via_a_layout(TMat<NT> &zt, const TMat<NT> &z, double2 *__restrict__ img, int lane)
{
    TOp<NT> a;
    to_a_layout(a, z, img, lane);                                  // a[I][Kt][kb] = Z[16I + c][16Kt + 4kb + g] = Z^T in D layout at [Kt][I][kb]
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int Kt = 0; Kt < NT; ++Kt)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                zt.re[Kt][I][kb] = a.re[I][Kt][kb];
                zt.im[Kt][I][kb] = a.im[I][Kt][kb];
            }
}

template <int NT>
GRAPE_DEV void tidentity(TMat<NT> &m, int lane)
{
    tzero(m);
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15))
                m.re[I][I][r] = 1.0;
}

template <int NT>
__global__ __launch_bounds__(64) void chunk_product_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    extern __shared__ double2 s_dynt[];
    const int lane = threadIdx.x, k = blockIdx.x, c = blockIdx.z;
    const int N = p.N, C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    const int t_lo = c * p.tp_S, t_hi = min(N, t_lo + p.tp_S);
    TMat<NT> V, Y, Pm, Pn;
    tidentity(V, lane);
    tload(Pm, Pk + (size_t)(t_hi - 1) * TSZ, lane);
    for (int t = t_hi - 1; t >= t_lo; --t) {                       // V <- P_t^T V: ends as (P_hi-1 ... P_lo)^T
        tload(Pn, Pk + (size_t)max(t - 1, 0) * TSZ, lane);
        tmul_tn<NT, false, false>(Y, Pm, V);
        V = Y;
        Pm = Pn;
    }
    if (p.tp_qt)
        tstore(p.tp_qt + (kw * C + c) * TSZ, V, lane);             // Q_c^T: the general flow's prefix scan needs Q_c as a left factor
    transpose_via_a_layout(Y, V, s_dynt, lane);
    tstore(p.tp_q + (kw * C + c) * TSZ, Y, lane);
}

// The same for 16 x 16 with kDeep propagators in flight (registers: 16 per dump): a chunk's products are dependent and take
// ~0.2 us each, a dump from HBM / Infinity Cache ~1 us -- with one dump ahead the chain ran at 0.64 (one problem, 22-slice
// chunks) .. 0.96 us (64 members, 125-slice chunks) per slice.  Used by the chunked propagator chain of action_thin.hip.
constexpr int kDeep = 6;
__global__ __launch_bounds__(64) void chunk_product_deep_kernel(const TileParams p)
{
    constexpr int TSZ = 256;
    extern __shared__ double2 s_dynt[];
    const int lane = threadIdx.x, k = blockIdx.x, c = blockIdx.z;
    const int N = p.N, C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    const int t_lo = c * p.tp_S, t_hi = min(N, t_lo + p.tp_S);
    TMat<1> V, Y, ring[kDeep];
    tidentity(V, lane);
#pragma unroll
    for (int u = 0; u < kDeep; ++u) {
        tload(ring[u], Pk + (size_t)max(t_hi - 1 - u, t_lo) * TSZ, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int t = t_hi - 1; t >= t_lo; t -= kDeep) {                // V <- P_t^T V: ends as (P_hi-1 ... P_lo)^T
#pragma unroll
        for (int u = 0; u < kDeep; ++u) {
            if (t - u >= t_lo) {
                tmul_tn<1, false, false>(Y, ring[u], V);
                V = Y;
            }
            __builtin_amdgcn_sched_barrier(0);                     // (the scheduler issued the six refills youngest-first: every
            tload(ring[u], Pk + (size_t)max(t - u - kDeep, t_lo) * TSZ, lane);   //  product then waited for all of them)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.tp_qt)
        tstore(p.tp_qt + (kw * C + c) * TSZ, V, lane);
    transpose_via_a_layout(Y, V, s_dynt, lane);
    tstore(p.tp_q + (kw * C + c) * TSZ, Y, lane);
}

// The same with FOUR wavefronts per chunk: the dependent chain is what a single problem waits for (16 products of 0.6 us
// at N = 1000), so each wave multiplies a quarter of the chunk's slices, the three upper quarters go to wave 0 through LDS
// (D-layout dumps) and wave 0 multiplies them on: S / 4 + 3 dependent products instead of S.
__global__ __launch_bounds__(256) void chunk_product_quad_kernel(const TileParams p)
{
    constexpr int TSZ = 256;
    extern __shared__ double2 s_dynt[];                            // [layout-conversion image of wave 0 | 3 dumps]
    double2 *s_dump = s_dynt + kTileImage + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.x, c = blockIdx.z;
    const int N = p.N, C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    const int t_lo = c * p.tp_S, t_hi = min(N, t_lo + p.tp_S);
    const int sub = (t_hi - t_lo + 3) / 4;
    const int a = min(t_hi, t_lo + wave * sub), b = min(t_hi, a + sub);      // this wave's slices (wave 0: the earliest)
    TMat<1> V, Y, ring[kDeep];
    tidentity(V, lane);
#pragma unroll
    for (int u = 0; u < kDeep; ++u) {
        tload(ring[u], Pk + (size_t)min(max(b - 1 - u, a), N - 1) * TSZ, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int t = b - 1; t >= a; t -= kDeep) {                      // V <- P_t^T V: ends as P_a^T ... P_b-1^T
#pragma unroll
        for (int u = 0; u < kDeep; ++u) {
            if (t - u >= a) {
                tmul_tn<1, false, false>(Y, ring[u], V);
                V = Y;
            }
            __builtin_amdgcn_sched_barrier(0);
            tload(ring[u], Pk + (size_t)min(max(t - u - kDeep, a), N - 1) * TSZ, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wave > 0)
        tstore(s_dump + (size_t)(wave - 1) * TSZ, V, lane);
    __syncthreads();
    if (wave != 0)
        return;
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {                                  // V <- V V_j+1: the quarters in time order
        TOp<1> va;
        TMat<1> W;
        to_a_layout(va, V, s_dynt, lane);
        tload(W, s_dump + (size_t)j * TSZ, lane);
        tmul_an<1, false, false>(Y, va, W);
        V = Y;
    }
    if (p.tp_qt)
        tstore(p.tp_qt + (kw * C + c) * TSZ, V, lane);
    transpose_via_a_layout(Y, V, s_dynt, lane);
    tstore(p.tp_q + (kw * C + c) * TSZ, Y, lane);
}

// general (non-unitary) flow: U_c = Q_c-1 ... Q_0 (handed out transposed: a free left factor) and R_c = Q_C-1 ... Q_c+1,
// one wavefront per direction.  GROUPS: the two-level form -- blockIdx.z >> 1 is a group of tp_gsize consecutive chunks,
// the products stay inside the group, and the suffix wavefront also hands out the group's own product (plain and
// transposed: the scan over the groups is this kernel again, without GROUPS, on those).
// tp_a, in blocks of (control arrays x units x groups) tiles: [A_j | group products | B_j^T | group products transposed]
template <int NT, bool GROUPS>
__global__ __launch_bounds__(64) void chunk_scan_general_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    extern __shared__ double2 s_dynt[];
    const int lane = threadIdx.x, k = blockIdx.x;
    const int C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const int j = (int)blockIdx.z >> 1;
    const int c_lo = GROUPS ? j * p.tp_gsize : 0, c_hi = GROUPS ? min(C, c_lo + p.tp_gsize) : C;
    TMat<NT> V, Y, T, Q, Qn;
    if ((blockIdx.z & 1) == 0) {                                   // the two scans are independent: one wavefront each
        const double2 *__restrict__ Qt = p.tp_qt + kw * C * TSZ;
        double2 *__restrict__ Uk = p.tp_u + kw * C * TSZ;
        tidentity(V, lane);                                        // V = U_c
        tload(Q, Qt + (size_t)c_lo * TSZ, lane);
        for (int c = c_lo; c < c_hi; ++c) {
            tload(Qn, Qt + (size_t)min(c + 1, C - 1) * TSZ, lane);
            tmul_tn<NT, false, false>(Y, Q, V);                    // U_{c+1} = Q_c U_c
            transpose_via_a_layout(T, V, s_dynt, lane);
            tstore(Uk + (size_t)c * TSZ, T, lane);                 // U_c^T
            V = Y;
            Q = Qn;
        }
    } else {
        const double2 *__restrict__ Qk = p.tp_q + kw * C * TSZ;
        double2 *__restrict__ Rk = p.tp_r + kw * C * TSZ;
        tidentity(V, lane);                                        // V = R_c^T
        tload(Q, Qk + (size_t)(c_hi - 1) * TSZ, lane);
        for (int c = c_hi - 1; c >= c_lo; --c) {
            tload(Qn, Qk + (size_t)max(c - 1, 0) * TSZ, lane);
            tmul_tn<NT, false, false>(Y, Q, V);                    // R_{c-1}^T = Q_c^T R_c^T
            transpose_via_a_layout(T, V, s_dynt, lane);
            tstore(Rk + (size_t)c * TSZ, T, lane);                 // R_c
            V = Y;
            Q = Qn;
        }
        if (GROUPS) {                                              // V = (the group's product)^T
            const size_t blk = (size_t)gridDim.y * p.E * p.tp_groups, at = kw * p.tp_groups + j;
            tstore(p.tp_a + (3 * blk + at) * TSZ, V, lane);
            transpose_via_a_layout(T, V, s_dynt, lane);
            tstore(p.tp_a + (blk + at) * TSZ, T, lane);
        }
    }
}

// two-level scan (many chunks): one wavefront per GROUP of tp_gsize consecutive chunks forms, for every chunk, the
// product of the chunks after it inside the group, and the group's own product; chunk_scan_kernel then runs over the
// groups instead of the chunks (2 sqrt(C) dependent products instead of C)
template <int NT>
__global__ __launch_bounds__(64) void chunk_scan_group_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    extern __shared__ double2 s_dynt[];
    const int lane = threadIdx.x, k = blockIdx.x, j = blockIdx.z;
    const int C = p.tp_chunks, G = p.tp_groups;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Qk = p.tp_q + kw * C * TSZ;
    double2 *__restrict__ Rk = p.tp_r + kw * C * TSZ;
    const int c_lo = j * p.tp_gsize, c_hi = min(C, c_lo + p.tp_gsize);
    TMat<NT> V, Y, T, Q, Qn;
    tidentity(V, lane);
    tload(Q, Qk + (size_t)(c_hi - 1) * TSZ, lane);
    for (int c = c_hi - 1; c >= c_lo; --c) {
        tload(Qn, Qk + (size_t)max(c - 1, 0) * TSZ, lane);
        tmul_tn<NT, false, false>(Y, Q, V);                        // issued first: the conversion below runs while the matrix cores work
        transpose_via_a_layout(T, V, s_dynt, lane);
        tstore(Rk + (size_t)c * TSZ, T, lane);                     // product of the chunks after c inside the group
        V = Y;
        Q = Qn;
    }
    transpose_via_a_layout(Y, V, s_dynt, lane);
    tstore(p.tp_a + ((kw + (size_t)gridDim.y * p.E) * G + j) * TSZ, Y, lane);   // the group's product, behind the A_j block
}

template <int NT, int SAND, bool PACK2>
__global__ __launch_bounds__(64) void chunk_scan_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    extern __shared__ double2 s_dynt[];
    double2 *s_img = s_dynt;
    const int lane = threadIdx.x, k = blockIdx.x;
    const int K = p.K, C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ Qk = p.tp_q + kw * C * TSZ;
    double2 *__restrict__ Rk = p.tp_r + kw * C * TSZ;
    TMat<NT> V, Y, T, Q, Qn;
    tidentity(V, lane);                                            // V = R_c^T, from the last chunk down
    tload(Q, Qk + (size_t)(C - 1) * TSZ, lane);
    for (int c = C - 1; c >= 0; --c) {
        tload(Qn, Qk + (size_t)max(c - 1, 0) * TSZ, lane);
        tmul_tn<NT, false, false>(Y, Q, V);                        // R_{c-1}^T = Q_c^T R_c^T (issued first: the conversion overlaps it)
        transpose_via_a_layout(T, V, s_img, lane);
        tstore(Rk + (size_t)c * TSZ, T, lane);                     // R_c
        V = Y;
        Q = Qn;
    }
    // V = T^T: X_N, M_N and z as chain_tile_unitary_kernel forms them
    TMat<NT> X, L, M;
    {
        TMat<NT> Xi;
        tload(Xi, ops + (size_t)(1 + 2 * K) * TSZ, lane);
        tmul_tn<NT, false, false>(X, V, Xi);                       // T Xi
    }
    if (SAND) {
        TOp<NT> YA;
        to_a_layout(YA, X, s_img, lane);
        tmul_an<NT, false, true>(Y, YA, V);                        // (T Xi) T'
        X = Y;
    }
    tload(L, ops + (size_t)(2 + 2 * K) * TSZ, lane);               // L_N = Xt
    TOp<NT> XA, LA;
    to_a_layout(XA, X, s_img, lane);
    to_a_layout(LA, L, s_img, lane);
    tprod<NT, false, true>(
        M, [&](int I, int Kt, int kb, double &r, double &i) { r = XA.re[I][Kt][kb]; i = XA.im[I][Kt][kb]; },
        [&](int Kt, int J, int kb, double &r, double &i) { r = LA.re[J][Kt][kb]; i = LA.im[J][Kt][kb]; });
    if (SAND) {
        tmul_tn<NT, true, false>(Y, L, X);                         // L' X
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                M.re[I][J] -= Y.re[I][J];
                M.im[I][J] -= Y.im[I][J];
            }
        double zz[2];
        tdot_partial<NT, true>(zz[0], zz[1], X, L);                // tr(X' L), the same for every t
        wave_sum_n(zz, PACK2);
        p.tp_z[(kw * 64 + lane) * 2] = zz[0];
        p.tp_z[(kw * 64 + lane) * 2 + 1] = zz[1];
    }
    tstore(p.tp_m + kw * TSZ, M, lane);
}

// ---------------------------------------------------------------------------------------------
int tile_count(int n) { return n <= 4 ? 0 : (n <= 16 ? 1 : (n <= 32 ? 2 : (n <= 48 ? 3 : (n <= 64 ? 4 : 0)))); }

static bool tile_chain_env(const char *what)                       // GRAPE_TILE_CHAIN=split|1w: tuning / ablation
{
    static const char *chain_env = std::getenv("GRAPE_TILE_CHAIN");
    if (!std::strcmp(what, "1w") && std::getenv("GRAPE_TILE_1WAVE"))
        return true;
    return chain_env && !std::strcmp(chain_env, what);
}

// does launch_sweep_tile run the time-split two-wave chain for this problem?  (then the `states` workspace
// holds X_t only for the first part of the time axis and prefix products for the rest)
bool tile_chain_is_split(const TileParams &p, bool keepl)
{
    // (any ensemble size: past 1024 members the workgroups queue for the 2 x 1024 wave slots, and the two-wave kernel is still
    // ahead of one wave per member -- C4dense-shaped, E = 2048 / 4096: 10.3 / 19.6 ms against 11.2 / 22.3; GRAPE_SPLIT_MAX_E
    // restores a limit for comparisons)
    static const long e_max = std::getenv("GRAPE_SPLIT_MAX_E") ? std::atol(std::getenv("GRAPE_SPLIT_MAX_E")) : (1L << 40);
    return tile_count(p.n) == 1 && !p.pack2 && !keepl && !p.unitary && !p.thin && (p.E_plan ? p.E_plan : p.E) < e_max && p.N >= 4 &&
           p.tp_chunks < 2 && p.K <= 64 && !tile_chain_env("1w");
}

// the two-wave chain with member-invariant control operators forms the propagators itself (chain_tile_split_kernel<.., EXPM>):
// launch_nt then runs ctrl_sum_kernel only.  GRAPE_SPLIT_EXPM=0 keeps prop_hoist1_kernel + the chain that reads P_t.
static bool split_forms_props(const TileParams &p, bool keepl)
{
    if (p.hoist != 1 || !tile_chain_is_split(p, keepl))
        return false;
    const char *e = std::getenv("GRAPE_SPLIT_EXPM");               // (read per launch, as GRAPE_HOIST2 is: tests switch it)
    return !(e && e[0] == '0');
}

// rank-one chain: is the forward vector pass fused into the expm kernel for this launch?  (0 no, 1 yes, 2 ablation:
// the fused kernel's grid without the hand-over.)  One workgroup then walks a member's N slices, so a member takes at
// least N x (one propagator / 4 waves) -- 1.6 ms at C4's N = 1000 however small the ensemble; it pays from half a
// device of workgroups on (four workgroups of four waves per CU).  Measured at C4, separate / fused, ms per
// evaluation: E = 256 1.40 / 1.63, 448 2.00 / 2.11, 512 2.35 / 2.12, 1024 3.62 / 3.15, 1536 5.99 / 5.64, 4096 14.3 / 11.9.
int tile_fuse_forward(const TileParams &p)
{
    if (tile_count(p.n) != 1 || p.thin != 1 || std::getenv("GRAPE_NO_FUSE"))
        return 0;
    if (sizeof(double2) * (4 * (size_t)kTileImage + (size_t)(p.K + 1) * 256) > 64 * 1024)   // generators not staged
        return 0;
    // (E_plan: a member-chunked launch takes the whole ensemble's decision -- its results must not depend on the chunk size)
    const long wgs = (long)(p.E_plan ? p.E_plan : p.E) * p.n_x, slots = 4L * (p.cus > 0 ? p.cus : 256);
    if (std::getenv("GRAPE_FORCE_FUSE") || 2 * wgs >= slots)
        return std::getenv("GRAPE_FUSE_ABL") ? 2 : 1;
    return 0;
}

template <int NT>
static hipError_t launch_nt(int sandwich, bool keepl, const TileParams &p, hipStream_t stream)
{
    {
        TileParams q = p;
        const size_t ops_bytes = sizeof(double2) * (size_t)(p.K + 1) * NT * NT * 256;
        constexpr int WPB = kPropWaves;
        const size_t img_bytes = sizeof(double2) * WPB * NT * (size_t)kTileImage;
        // the member's K + 1 generator tiles in LDS, read once per kPropSlices slices: NT = 1 keeps room for four
        // workgroups per CU; NT = 2 runs one workgroup per CU anyway (registers) and may take what is left of the
        // 160 KB -- without it every wave waits for K + 1 dependent 16 KB fetches from L2 / Infinity Cache per slice
        q.stage_ops = (img_bytes + ops_bytes <= (size_t)(NT == 1 ? 64 : 160) * 1024) ? 1 : 0;
        // slices per workgroup: kPropSlices when that still fills the device, fewer for small ensembles (a single 32 x 32
        // problem of 2000 slices ran 32 workgroups: 235 us) -- as many as make ONE round of resident workgroups
        // (NT = 2 holds a CU per workgroup, NT = 1 a quarter), in multiples of the four waves
        int per_block = WPB;
        const bool hoisted = p.hoist != 0;                         // prop_hoist.hip: A'_k in registers, nothing staged
        if (q.stage_ops || hoisted) {
            const long resident = (long)(p.cus > 0 ? p.cus : 256) * (NT == 1 ? 4 : (hoisted ? 3 : 1));
            const long total = (long)p.N * p.E * p.n_x;
            const long want = ((total + resident - 1) / resident + WPB - 1) / WPB * WPB;
            per_block = (int)std::min<long>(kPropSlices, std::max<long>(WPB, want));
        }
        q.prop_slices = per_block;
        // rank-one chain: fuse the forward vector pass into this kernel when one workgroup per member fills the
        // device (four workgroups of four waves per CU): the last round of workgroups must be at least 90 % full
        q.fuse_fwd = (NT == 1 && (q.stage_ops || hoisted)) ? tile_fuse_forward(p) : 0;
        const size_t lds = img_bytes + (q.stage_ops ? ops_bytes : 0) + (q.fuse_fwd ? sizeof(double2) * 33 : 0);
        if (lds > 64 * 1024) {
            hipError_t ea = ensure_dynamic_lds((const void *)prop_tile_kernel<NT>, lds);
            if (ea != hipSuccess)
                return ea;
        }
        hipError_t e;
        // 32 x 32, member-invariant controls, an ensemble that fills the device (C5): sweep_grid.hip's expm kernel -- a workgroup
        // per propagator as prop_hoist2_kernel, but operand sums formed in registers and k-contiguous ds_read_b128 pairs instead
        // of a third LDS plane: 85.5 against 96-97 ms at C5, matrix pipe 0.72 against 0.64.  GRAPE_HOIST2=1 keeps prop_hoist2.
        const bool keep_hoist2 = std::getenv("GRAPE_HOIST2") != nullptr;
        if (NT == 1 && split_forms_props(p, keepl)) {              // chain_tile_split_kernel<.., EXPM>: only the control sums
            e = launch_ctrl_sum(1, q, stream);
        } else if (hoisted && NT == 2 && p.hoist == 1 && !keep_hoist2 &&
                   (long)(p.E_plan ? p.E_plan : p.E) * p.n_x >= (long)(p.cus > 0 ? p.cus : 256)) {
            e = launch_grid_prop(2, q, stream);
        } else if (hoisted) {                                      // member-invariant controls: prop_hoist.hip
            e = launch_prop_hoist(NT, q, stream);
        } else {
            GRAPE_LAUNCH((prop_tile_kernel<NT>), dim3(q.fuse_fwd ? 1 : (p.N + per_block - 1) / per_block, p.E, p.n_x),
                               dim3(64 * WPB), lds, stream, q);
            e = hipGetLastError();
        }
        if (e != hipSuccess)
            return e;
        if (p.ev_mid) {
            e = hipEventRecord(p.ev_mid, stream);
            if (e != hipSuccess)
                return e;
        }
        if constexpr (NT == 1) {
            if (p.thin == 2) {
                if (p.tp_chunks > 1) {                                 // small ensembles: Q_c and Q_c^T of every chunk of the time axis
                    static const bool one_wave = std::getenv("GRAPE_CHUNK_PRODUCT_1W") != nullptr;
                    if (p.tp_S >= 8 && !one_wave)                      // four waves per chunk: S / 4 + 3 dependent products
                        GRAPE_LAUNCH(chunk_product_quad_kernel, dim3(p.E, p.n_x, p.tp_chunks), dim3(256),
                                           sizeof(double2) * (kTileImage + 1 + 3 * 256), stream, q);
                    else
                        GRAPE_LAUNCH(chunk_product_deep_kernel, dim3(p.E, p.n_x, p.tp_chunks), dim3(64),
                                           sizeof(double2) * (kTileImage + 1), stream, q);
                    e = hipGetLastError();
                    if (e != hipSuccess)
                        return e;
                }
                return launch_chain_prop(sandwich, q, stream);         // action_thin.hip: one DPP matrix-vector product per slice
            }
        }
        if (NT == 1 && p.thin)
            return launch_chain_thin(sandwich, q, stream);
    }
    TileParams q = p;
    const size_t bt_bytes = sizeof(double2) * (size_t)p.K * NT * NT * 256;
    q.bt_in_lds = bt_bytes <= 36 * 1024 ? 1 : 0;                   // 4 waves per CU must still fit
    const size_t lds = sizeof(double2) * (kTileImage + 1) + (q.bt_in_lds ? bt_bytes : 0);
    const dim3 grid(p.E, p.n_x), block(64);                        // y: control array of a batched evaluation
    const bool pk = (NT == 1) && p.pack2;
    if (tile_chain_is_split(p, keepl)) {
        // one member per wave would leave every SIMD with a single wave: two waves per member that meet in the middle of the
        // time axis (see chain_tile_split_kernel; both waves do the same work per slice, so the middle is N / 2)
        int n0 = p.N / 2;
        if (const char *sp = std::getenv("GRAPE_TILE_SPLIT_PERMILLE"))
            n0 = (int)((long long)p.N * std::atoi(sp) / 1000);
        if (n0 < 0) n0 = 0;                                        // (N = 1: wave 1 owns the slice's costate, wave 0 its gradient)
        if (n0 > p.N - 1) n0 = p.N - 1;                            // (wave 0 always ends at t = N - 1: it writes the figure of merit)
        q.split_at = n0;
        q.split_expm = split_forms_props(p, keepl) ? 1 : 0;
        constexpr int parts = 2;
        const size_t bt_b = sizeof(double2) * (size_t)p.K * 256;
        const size_t img_b = sizeof(double2) * (size_t)parts * (kTileImage + 1);
        q.bt_in_lds = img_b + bt_b <= 40 * 1024 ? 1 : 0;           // 4 workgroups per CU must still fit
        const size_t lds2 = img_b + (q.bt_in_lds ? bt_b : 0);
        const dim3 blk(64 * parts);
        const int hm = !sandwich || !p.herm_states ? 0 : (p.herm_ctrl ? 2 : 1);      // (see HERM at the kernel)
        const bool lists4 = p.sparse && p.K == 4 && p.sp_nz == 64 && !std::getenv("GRAPE_SPLIT_LISTS_LDS");   // SPARSE = 2
#define GRAPE_SPLIT_GO2(SP, LDS, EX)                                                                             \
    {                                                                                                            \
        if (!sandwich)    GRAPE_LAUNCH((chain_tile_split_kernel<0, SP, 0, EX>), grid, blk, LDS, stream, q);      \
        else if (hm == 2) GRAPE_LAUNCH((chain_tile_split_kernel<1, SP, 2, EX>), grid, blk, LDS, stream, q);      \
        else if (hm == 1) GRAPE_LAUNCH((chain_tile_split_kernel<1, SP, 1, EX>), grid, blk, LDS, stream, q);      \
        else              GRAPE_LAUNCH((chain_tile_split_kernel<1, SP, 0, EX>), grid, blk, LDS, stream, q);      \
    }
#define GRAPE_SPLIT_GO(SP, LDS) \
    { if (q.split_expm) GRAPE_SPLIT_GO2(SP, LDS, true) else GRAPE_SPLIT_GO2(SP, LDS, false) }
        if (p.sparse) {
            const size_t lds_sp = img_b + sizeof(double2) * ((size_t)p.K * p.sp_nz + (size_t)parts * 16 * 17) +
                                  sizeof(int32_t) * (size_t)p.K * p.sp_nz;
            if (lists4) GRAPE_SPLIT_GO(2, lds_sp) else GRAPE_SPLIT_GO(1, lds_sp)
            return hipGetLastError();
        }
        GRAPE_SPLIT_GO(0, lds2)
#undef GRAPE_SPLIT_GO
#undef GRAPE_SPLIT_GO2
        return hipGetLastError();
    }
    // unitary flow, small ensembles: the time axis in chunks (chunk_product_kernel / chunk_scan_kernel above), grid.z = chunk
    const bool tp = p.tp_chunks > 1 && !(keepl && p.unitary);      // (stored costates always come with the general flow)
    q.tp_chunks = tp ? p.tp_chunks : 0;
    const dim3 ugrid(p.E, p.n_x, tp ? p.tp_chunks : 1);
    if (tp) {
        const size_t lds_img = sizeof(double2) * (kTileImage + 1);
        if (p.unitary)
            q.tp_qt = nullptr;                                     // only the general flow's prefix scan needs Q_c^T

        const bool coop = NT == 2 && coop_applies(q, sandwich, keepl);     // few units: four waves per product (sweep_coop.hip)
        if (coop) {
            hipError_t ec = launch_coop_chunk_product(q, stream);
            if (ec != hipSuccess)
                return ec;
        } else
        GRAPE_LAUNCH((chunk_product_kernel<NT>), ugrid, block, lds_img, stream, q);
        if (!p.unitary) {
            if (q.tp_groups) {                                     // two levels: inside the groups, then over the groups
                GRAPE_LAUNCH((chunk_scan_general_kernel<NT, true>), dim3(p.E, p.n_x, 2 * q.tp_groups), block, lds_img, stream, q);
                TileParams s2 = q;
                const size_t blk = (size_t)p.n_x * p.E * q.tp_groups * NT * NT * 256;
                s2.tp_chunks = q.tp_groups;
                s2.tp_r = q.tp_a;
                s2.tp_q = q.tp_a + blk;
                s2.tp_u = q.tp_a + 2 * blk;
                s2.tp_qt = q.tp_a + 3 * blk;
                GRAPE_LAUNCH((chunk_scan_general_kernel<NT, false>), dim3(p.E, p.n_x, 2), block, lds_img, stream, s2);
            } else {
                GRAPE_LAUNCH((chunk_scan_general_kernel<NT, false>), dim3(p.E, p.n_x, 2), block, lds_img, stream, q);
            }
        }
        else {
            TileParams s2 = q;                                     // what the serial scan runs over: chunks, or groups of chunks
            if (q.tp_groups) {
                if (coop) {
                    hipError_t ec = launch_coop_scan_group(q, stream);
                    if (ec != hipSuccess)
                        return ec;
                } else
                GRAPE_LAUNCH((chunk_scan_group_kernel<NT>), dim3(p.E, p.n_x, q.tp_groups), block, lds_img, stream, q);
                s2.tp_chunks = q.tp_groups;
                s2.tp_q = q.tp_a + (size_t)p.n_x * p.E * q.tp_groups * NT * NT * 256;
                s2.tp_r = q.tp_a;
            }
            if (coop) {
                hipError_t ec = launch_coop_scan(sandwich, s2, stream);
                if (ec != hipSuccess)
                    return ec;
            } else
            if (sandwich) { if (pk) GRAPE_LAUNCH((chunk_scan_kernel<NT, 1, NT == 1>), grid, block, lds_img, stream, s2);
                            else    GRAPE_LAUNCH((chunk_scan_kernel<NT, 1, false>), grid, block, lds_img, stream, s2); }
            else          { if (pk) GRAPE_LAUNCH((chunk_scan_kernel<NT, 0, NT == 1>), grid, block, lds_img, stream, s2);
                            else    GRAPE_LAUNCH((chunk_scan_kernel<NT, 0, false>), grid, block, lds_img, stream, s2); }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
#define GRAPE_LAUNCH_CHAIN(KERNEL) GRAPE_LAUNCH(KERNEL, ugrid, block, lds, stream, q)
#define GRAPE_LAUNCH_UNI(KERNEL) GRAPE_LAUNCH(KERNEL, ugrid, block, lds, stream, q)
    if (NT == 2 && tp && p.unitary && !keepl && p.sparse && !pk && coop_applies(q, sandwich, keepl)) {
        hipError_t ec = launch_coop_chain_unitary(sandwich, q, stream);
        if (ec != hipSuccess)
            return ec;
    } else if (p.unitary && !keepl && p.sparse && !pk) {
        // image for layout conversions | coefficients | image of M | positions
        const size_t lds_sp = sizeof(double2) * (kTileImage + 1 + (size_t)p.K * p.sp_nz + 16 * NT * (16 * NT + 1)) +
                              sizeof(int32_t) * (size_t)p.K * p.sp_nz;
        if (sandwich) GRAPE_LAUNCH((chain_tile_unitary_kernel<NT, 1, false, true>), ugrid, block, lds_sp, stream, q);
        else          GRAPE_LAUNCH((chain_tile_unitary_kernel<NT, 0, false, true>), ugrid, block, lds_sp, stream, q);
    } else if (p.unitary && !keepl) {
        if (sandwich) { if (pk) GRAPE_LAUNCH_UNI((chain_tile_unitary_kernel<NT, 1, NT == 1>)); else GRAPE_LAUNCH_UNI((chain_tile_unitary_kernel<NT, 1, false>)); }
        else          { if (pk) GRAPE_LAUNCH_UNI((chain_tile_unitary_kernel<NT, 0, NT == 1>)); else GRAPE_LAUNCH_UNI((chain_tile_unitary_kernel<NT, 0, false>)); }
    } else if (p.sparse && !pk) {
        const size_t lds_sp = sizeof(double2) * (kTileImage + 1 + (size_t)p.K * p.sp_nz + 16 * NT * (16 * NT + 1)) +
                              sizeof(int32_t) * (size_t)p.K * p.sp_nz;
#define GRAPE_LAUNCH_SP(KERNEL) GRAPE_LAUNCH(KERNEL, ugrid, block, lds_sp, stream, q)
        if (sandwich) { if (keepl) GRAPE_LAUNCH_SP((chain_tile_kernel<NT, 1, true, false, true>)); else GRAPE_LAUNCH_SP((chain_tile_kernel<NT, 1, false, false, true>)); }
        else          { if (keepl) GRAPE_LAUNCH_SP((chain_tile_kernel<NT, 0, true, false, true>)); else GRAPE_LAUNCH_SP((chain_tile_kernel<NT, 0, false, false, true>)); }
#undef GRAPE_LAUNCH_SP
    } else if (sandwich) {
        if (keepl) { if (pk) GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 1, true, NT == 1>)); else GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 1, true, false>)); }
        else       { if (pk) GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 1, false, NT == 1>)); else GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 1, false, false>)); }
    } else {
        if (keepl) { if (pk) GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 0, true, NT == 1>)); else GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 0, true, false>)); }
        else       { if (pk) GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 0, false, NT == 1>)); else GRAPE_LAUNCH_CHAIN((chain_tile_kernel<NT, 0, false, false>)); }
    }
#undef GRAPE_LAUNCH_CHAIN
#undef GRAPE_LAUNCH_UNI
    return hipGetLastError();
}

hipError_t launch_sweep_tile(int n, int sandwich, bool keep_costates, const TileParams &p, hipStream_t stream)
{
    if (p.action)
        return launch_action_thin(sandwich, p, stream);
    switch (tile_count(n)) {
    case 1: return launch_nt<1>(sandwich, keep_costates, p, stream);
    case 2: return launch_nt<2>(sandwich, keep_costates, p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace grape
