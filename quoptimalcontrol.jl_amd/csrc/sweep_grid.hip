// sweep_grid.hip -- the GRAPE hot path for operators beyond one wavefront's registers: n = 33 .. 64 (NT = 3, 4 tiles of 16
// per side; `_fom_and_gradient_GRAPE!` src/GRAPE.jl:25-96 is size-generic, and :101 sends "too large" systems to it).
//
// A 64 x 64 ComplexF64 matrix is 64 KB -- 256 registers per lane of ONE wave.  Here a WORKGROUP of NT x NT waves owns every
// matrix, wave (I, J) its 16 x 16 tile (I, J) in the FP64 matrix cores' accumulator ("D") layout of tile.hpp: 16 registers
// per matrix and wave.  Element-wise work (the H build, the Taylor combinations, the traces) never leaves the owner.  A
// product  C = op(A) op(B)  goes through two images in LDS (re plane | im plane each):
//     barrier -- owners write their tiles of A and B -- barrier -- wave (I, J) reads the fragments of row I of op(A) and of
//     column J of op(B) as MFMA operands and runs 4 NT k-blocks x 3 v_mfma_f64_16x16x4 (the three-product complex
//     multiplication of tile.hpp, operand sums formed in registers).
// The images are written K-CONTIGUOUS for their role: the left operand as [output row][k], the right operand as [output
// column][k] (a plain or a transposed write of the owner's tile; a conjugate transpose swaps which of the two it is), and a
// k-block takes the k values 16 Kt + 4 (lane >> 4) + kb -- any assignment of k to (k-block, lane group) will do as long as
// both operands use the same one -- so a lane's operands of FOUR k-blocks are 32 contiguous bytes: two ds_read_b128 per
// plane where the MFMA's native k order needs four ds_read_b64 (the first version: 3.1 LDS instructions per MFMA, matrix pipe
// 0.43 / 0.56 busy at C6).  Row pitch 16 NT + 2 doubles: (pitch / 2) odd spreads the 64 lanes' 16-byte reads evenly over
// the bank groups.  An operand that is still in its image from the previous product is not written again (P in P' (L P)).
// LDS: 4 planes x 16 NT x (16 NT + 2) doubles = 132 KB at NT = 4: one workgroup of 16 waves per compute unit, four waves
// per SIMD, whose MFMA phases cover each other's LDS phases.
//
// Data flow = the reference's own (general) flow for every system type and both formula variants: grid_prop_kernel forms
// G_t = (-i dt)(A + sum_c x[c,t] B_c) in the reference's association and P_t = exp(G_t) (degree-8 Taylor polynomial in three
// products + squarings, cmat.hpp's constants, theta8 = 0.08) for a block of slices per workgroup; grid_chain_kernel, one
// workgroup per member, stores the forward states X_t (src/GRAPE.jl:53-63), pulls the costates back (:65-75) and takes the
// gradient traces of :261-303 and the figure of merit at t = N (:77, :94) from  R_t = X_t L_t'  [- L_t' X_t].
// Workspace and operator format: tile.hpp's D-layout dumps, element (tile, r, lane) at ((I NT + J) 4 + r) 64 + lane -- the
// host packing, the reduction kernels and grape_get_trajectory are the tile family's.
#include "cmat.hpp"
#include "grape_kernels.hpp"
#include "tile.hpp"

namespace grape {

struct GT {                        // this wave's tile of a matrix, D layout
    d4 re, im;
};

template <int NT>
struct GridGeom {
    static constexpr int DIM = 16 * NT;
    static constexpr int P = DIM + 2;             // row pitch (doubles): P / 2 odd -- 16-byte reads of 16 rows x 4 groups hit every bank group 4 times
    static constexpr int PLANE = DIM * P;         // doubles per plane; an image is [re plane | im plane]
    static constexpr int WAVES = NT * NT;
    static constexpr int TSZ = NT * NT * 256;     // double2 per matrix dump
};

constexpr int kGridGroup = 4;                     // controls per trace reduction
constexpr int kGridRed = 2 + 2 * kGridGroup;      // doubles a wave contributes per reduction: z, then (re, im) per control

size_t grid_lds_bytes(int NT, bool chain)
{
    const size_t dim = 16 * (size_t)NT, plane = dim * (dim + 2);
    return sizeof(double) * (4 * plane + (chain ? 2 * (size_t)NT * NT * kGridRed : (size_t)NT * dim));
}

GRAPE_DEV GT gt_load(const double2 *__restrict__ dump, int tile, int lane)
{
    GT t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2 v = dump[(tile * 4 + r) * 64 + lane];
        t.re[r] = v.x;
        t.im[r] = v.y;
    }
    return t;
}

GRAPE_DEV void gt_store(double2 *__restrict__ dump, int tile, int lane, const GT &t)
{
#pragma unroll
    for (int r = 0; r < 4; ++r)
        dump[(tile * 4 + r) * 64 + lane] = make_double2(t.re[r], t.im[r]);
}

// own tile -> image, element (row, col) = (16 I + 4 r + (lane >> 4), 16 J + (lane & 15)) of the matrix at [row][col]
// (TRANS = false) or at [col][row] (TRANS = true)
template <int NT, bool TRANS>
GRAPE_DEV void grid_put(double *__restrict__ img, const GT &t, int I, int J, int lane)
{
    constexpr int P = GridGeom<NT>::P, PLANE = GridGeom<NT>::PLANE;
    const int row = 16 * I + (lane >> 4), col = 16 * J + (lane & 15);
    const int at = TRANS ? col * P + row : row * P + col;
    constexpr int step = TRANS ? 4 : 4 * P;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        img[at + r * step] = t.re[r];
        img[PLANE + at + r * step] = t.im[r];
    }
}

// tile (I, J) of C = L R for a left operand given as ia[i][k] = L[i][k] and a right operand given as ib[j][k] = R[k][j];
// CA / CB: that operand is to be conjugated (so  A'  as the left operand = the TRANSPOSED image of A with CA, and  B'  as the
// right operand = the PLAIN image of B with CB).  k-block (Kt, kb) of lane l covers k = 16 Kt + 4 (l >> 4) + kb.
// (x + i sa y)(u + i sb v): re = xu - sa sb yv, im = (x + sa y)(u + sb v) - xu - sa sb yv.
template <int NT, bool CA, bool CB>
GRAPE_DEV GT grid_mma(const double *__restrict__ ia, const double *__restrict__ ib, int I, int J, int lane)
{
    constexpr int P = GridGeom<NT>::P, PLANE = GridGeom<NT>::PLANE;
    const int lo = lane & 15, hi = lane >> 4;
    const double2 *__restrict__ ar2 = reinterpret_cast<const double2 *>(ia + (16 * I + lo) * P + 4 * hi);
    const double2 *__restrict__ ai2 = reinterpret_cast<const double2 *>(ia + PLANE + (16 * I + lo) * P + 4 * hi);
    const double2 *__restrict__ br2 = reinterpret_cast<const double2 *>(ib + (16 * J + lo) * P + 4 * hi);
    const double2 *__restrict__ bi2 = reinterpret_cast<const double2 *>(ib + PLANE + (16 * J + lo) * P + 4 * hi);
    d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0};
#pragma unroll
    for (int Kt = 0; Kt < NT; ++Kt) {
        const double2 a0 = ar2[8 * Kt], a1 = ar2[8 * Kt + 1], c0 = ai2[8 * Kt], c1 = ai2[8 * Kt + 1];
        const double2 b0 = br2[8 * Kt], b1 = br2[8 * Kt + 1], e0 = bi2[8 * Kt], e1 = bi2[8 * Kt + 1];
        const double ar[4] = {a0.x, a0.y, a1.x, a1.y}, ai[4] = {c0.x, c0.y, c1.x, c1.y};
        const double br[4] = {b0.x, b0.y, b1.x, b1.y}, bi[4] = {e0.x, e0.y, e1.x, e1.y};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double as = CA ? ar[kb] - ai[kb] : ar[kb] + ai[kb];
            const double bs = CB ? br[kb] - bi[kb] : br[kb] + bi[kb];
            t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[kb], br[kb], t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[kb], bi[kb], t2, 0, 0, 0);
            t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(as, bs, t3, 0, 0, 0);
        }
    }
    GT out;
    if (CA != CB) {
        out.re = t1 + t2;
        out.im = t3 - t1 + t2;
    } else {
        out.re = t1 - t2;
        out.im = t3 - t1 - t2;
    }
    return out;
}

// The workgroup barrier around the images: the wave's own LDS operations drained, then s_barrier.  (__syncthreads() also
// waits for the wave's GLOBAL loads -- vmcnt(0) -- which would pull the next slice's prefetched tiles onto the critical
// path of every product.)
GRAPE_DEV void grid_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

GRAPE_DEV void gt_add_identity(GT &t, double c, int I, int J, int lane)
{
    if (I == J) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + (lane >> 4) == (lane & 15))
                t.re[r] += c;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// P_t = exp((-i dt) H_t) for the slices [blockIdx.x * prop_slices, ...) of member blockIdx.y, control array blockIdx.z
// HOIST: member-invariant control operators -- Gc_t = (-i dt) sum_c x[c,t] B_c and its norm bound come from the pre-pass
// (ctrl_sum_kernel, prop_hoist.hip), A'_k = (-i dt) A_k stays in registers, the next slice's Gc tile is in flight under the
// products, the squarings come from the two norm bounds (|A'_k|_1 + |Gc_t|_1 >= |G_t|_1: never fewer) -- no K + 1 operator
// tile fetches and no cross-wave norm reduction per slice.
template <int NT, bool HOIST>
__global__ __launch_bounds__(64 * NT * NT) void grid_prop_kernel(const TileParams p)
{
    using G_ = GridGeom<NT>;
    constexpr int TSZ = G_::TSZ, PLANE = G_::PLANE, DIM = G_::DIM;
    extern __shared__ double s_grid[];
    double *img0 = s_grid, *img1 = s_grid + 2 * PLANE, *s_col = s_grid + 4 * PLANE;      // s_col[I][column]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave / NT, J = wave % NT, tile = I * NT + J;
    const int k = blockIdx.y, K = p.K;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;             // [A | B_c | B_c^T | Xi | Xt]
    const double *__restrict__ x = p.x + (size_t)blockIdx.z * K * p.N;
    const int t_lo = blockIdx.x * p.prop_slices, t_hi = min(p.N, t_lo + p.prop_slices);
    const double dt = p.dt;
    GT Ah, Gc;
    double nA = 0.0;
    const double sk = (HOIST && p.ctrl_scale) ? p.ctrl_scale[k] : 1.0;      // B_k = s_k B_0: G = A'_k + s_k Gc_t
    if (HOIST) {
        Ah = gt_load(p.ha + (size_t)k * TSZ, tile, lane);
        nA = p.ha_norm[k];
        Gc = gt_load(p.gc + ((size_t)blockIdx.z * p.N + t_lo) * TSZ, tile, lane);
    }
    for (int t = t_lo; t < t_hi; ++t) {
        // H in the reference's association: in-place variant sum_c B_c x_c first, then + A (src/timeevolution.jl:101-108);
        // static variant A first (:49-52)
        GT G;
        int s_h = 0;
        if (HOIST) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                G.re[r] = fma(sk, Gc.re[r], Ah.re[r]);            // (s_k = 1: the plain sum, bit for bit)
                G.im[r] = fma(sk, Gc.im[r], Ah.im[r]);
            }
            const double bound = fma(fabs(sk), p.gcn[(size_t)blockIdx.z * p.N + t], nA) * kTheta8;
            s_h = squarings_for(bound);
            Gc = gt_load(p.gc + ((size_t)blockIdx.z * p.N + min(t + 1, t_hi - 1)) * TSZ, tile, lane);
        } else if (p.variant == 0) {
            G.re = (d4){0, 0, 0, 0};
            G.im = (d4){0, 0, 0, 0};
        } else {
            G = gt_load(ops, tile, lane);
        }
        double cs = 0.0;                               // this tile's share of its columns' sums of |re| + |im|
        if (!HOIST) {
            for (int c = 0; c < K; ++c) {
                const double xv = x[c + (size_t)t * K];
                const GT B = gt_load(ops + (size_t)(1 + c) * TSZ, tile, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    G.re[r] = fma(B.re[r], xv, G.re[r]);
                    G.im[r] = fma(B.im[r], xv, G.im[r]);
                }
            }
            if (p.variant == 0) {
                const GT A = gt_load(ops, tile, lane);
                G.re += A.re;
                G.im += A.im;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double hr = G.re[r], hi = G.im[r];
                G.re[r] = dt * hi;                     // (-i dt)(hr + i hi)
                G.im[r] = -dt * hr;
                cs += fabs(G.re[r]) + fabs(G.im[r]);
            }
            cs = swap16_add(cs, cs);
            cs = swap32_add(cs, cs);
        }
        grid_barrier();                                // the previous slice's readers are done
        grid_put<NT, false>(img0, G, I, J, lane);
        grid_put<NT, true>(img1, G, I, J, lane);
        if (!HOIST && lane < 16)
            s_col[I * DIM + 16 * J + lane] = cs;
        grid_barrier();
        int s = p.s_forced >= 0 ? p.s_forced : s_h;
        if (!HOIST && p.s_forced < 0) {
            double colmax = 0.0;
            if (lane < DIM) {
#pragma unroll
                for (int ii = 0; ii < NT; ++ii)
                    colmax += s_col[ii * DIM + lane];
            }
            colmax = wave_max_fast(colmax);            // the same number in every wave: upper bound of |G|_1
            s = squarings_for(colmax);
        }
        // A2 = G G on the unscaled generator, then the power-of-two scaling (exact) on both
        GT A2 = grid_mma<NT, false, false>(img0, img1, I, J, lane);
        if (s > 0) {
            const double sc = ldexp(1.0, -s), sc2 = ldexp(1.0, -2 * s);
            G.re *= sc;
            G.im *= sc;
            A2.re *= sc2;
            A2.im *= sc2;
        }
        GT T;
        T.re = kX1 * G.re + kX2 * A2.re;
        T.im = kX1 * G.im + kX2 * A2.im;
        grid_barrier();
        grid_put<NT, false>(img0, A2, I, J, lane);
        grid_put<NT, true>(img1, T, I, J, lane);
        grid_barrier();
        const GT A4 = grid_mma<NT, false, false>(img0, img1, I, J, lane);      // A4 = A2 (x1 G + x2 A2)
        GT U;
        U.re = kX3 * A2.re + A4.re;
        U.im = kX3 * A2.im + A4.im;
        T.re = kX5 * G.re + kX6 * A2.re + kX7 * A4.re;
        T.im = kX5 * G.im + kX6 * A2.im + kX7 * A4.im;
        gt_add_identity(T, kX4, I, J, lane);
        grid_barrier();
        grid_put<NT, false>(img0, U, I, J, lane);
        grid_put<NT, true>(img1, T, I, J, lane);
        grid_barrier();
        GT Pm = grid_mma<NT, false, false>(img0, img1, I, J, lane);            // A8
        Pm.re += G.re + kY2 * A2.re;
        Pm.im += G.im + kY2 * A2.im;
        gt_add_identity(Pm, 1.0, I, J, lane);
        for (int i = 0; i < s; ++i) {                  // undo the scaling
            grid_barrier();
            grid_put<NT, false>(img0, Pm, I, J, lane);
            grid_put<NT, true>(img1, Pm, I, J, lane);
            grid_barrier();
            Pm = grid_mma<NT, false, false>(img0, img1, I, J, lane);
        }
        gt_store(p.props + (((size_t)blockIdx.z * p.E + k) * p.N + t) * TSZ, tile, lane, Pm);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// one workgroup per (member, control array): forward states, costates, gradient, figure of merit
// SPARSE (NT >= 3): control operators with few non-zeros (Pauli-type controls: TileParams.sp_coef / sp_addr hold, per
// control, sp_nz entries (B_c[i][j], position of R[j][i] in a plain image of pitch 16 NT + 2)).  R_t goes to image 0 once
// and WAVE c forms control c's trace itself from its list (held in registers for the whole sweep) and tr R from the image's
// diagonal: no K operator-tile fetches per slice, no cross-wave reduction, the gradient entry leaves from the wave that
// summed it.
// SPNE: list entries a lane holds per control -- 0 dense traces, 1 lists of 64 (single Pauli strings), 4 up to 256.
template <int NT, int SAND, bool KEEPL, int SPNE>
__global__ __launch_bounds__(64 * NT * NT) void grid_chain_kernel(const TileParams p)
{
    using G_ = GridGeom<NT>;
    constexpr int TSZ = G_::TSZ, PLANE = G_::PLANE, WAVES = G_::WAVES, DIM = G_::DIM, P = G_::P;
    constexpr bool SPARSE = SPNE > 0;
    constexpr int NCW = SPARSE ? (16 + WAVES - 1) / WAVES : 1;        // controls a wave may own (K <= 16)
    constexpr int NEM = SPARSE ? SPNE : 1;
    extern __shared__ double s_grid[];
    double *img0 = s_grid, *img1 = s_grid + 2 * PLANE, *s_red = s_grid + 4 * PLANE;      // s_red[2][WAVES][kGridRed]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave / NT, J = wave % NT, tile = I * NT + J;
    const int k = blockIdx.x, K = p.K, N = p.N;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ opBT = ops + (size_t)(1 + K) * TSZ;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    double2 *__restrict__ Xk = p.states + kw * N * TSZ;
    double *__restrict__ out = p.member_out + ((size_t)blockIdx.y * p.E_members + k) * ((size_t)K * N + 1);
    // Chunked time axis (round 6: single problems / ensembles far smaller than the device walked their N slices in ONE
    // workgroup): this workgroup owns the slices [t_lo, t_hi) of chunk blockIdx.z; the state at t_lo and the costate behind
    // t_hi - 1 come from the boundary scan -- THIS kernel run on the chunk products Q_c in the place of the propagators
    // (p.tp_scan: N = number of chunks, states -> tp_u, costates (KEEPL) -> tp_r, no output rows).
    const bool win = p.tp_window != 0;
    const int CHN = win ? p.tp_chunks : 1, chn = win ? (int)blockIdx.z : 0;
    const int t_lo = win ? chn * p.tp_S : 0, t_hi = win ? min(N, t_lo + p.tp_S) : N;
    const bool emit = p.tp_scan == 0;

    // ------------------------------------------------------------ forward sweep, src/GRAPE.jl:53-63
    {
        GT X = win ? gt_load(p.tp_u + (kw * CHN + chn) * TSZ, tile, lane)
                   : gt_load(ops + (size_t)(1 + 2 * K) * TSZ, tile, lane);      // Xi
        GT Pt = gt_load(Pk + (size_t)t_lo * TSZ, tile, lane);
        for (int t = t_lo; t < t_hi; ++t) {
            gt_store(Xk + (size_t)t * TSZ, tile, lane, X);
            if (t + 1 < t_hi) {                        // (the state behind the last slice is never read)
                grid_barrier();
                grid_put<NT, false>(img0, Pt, I, J, lane);                      // P plain: the left operand of P X, the right one of . P'
                grid_put<NT, true>(img1, X, I, J, lane);
                grid_barrier();
                Pt = gt_load(Pk + (size_t)(t + 1) * TSZ, tile, lane);           // next slice's tile in flight under the products
                if (SAND) {
                    const GT Y = grid_mma<NT, false, false>(img0, img1, I, J, lane);      // P X            (:245-246 as (P X) P')
                    grid_barrier();
                    grid_put<NT, false>(img1, Y, I, J, lane);
                    grid_barrier();
                    X = grid_mma<NT, false, true>(img1, img0, I, J, lane);      // (P X) P'
                } else {
                    X = grid_mma<NT, false, false>(img0, img1, I, J, lane);     // P X            (:226)
                }
            }
        }
    }
    __syncthreads();                                   // (the wave's own X_t stores have left before it reads them back)
    // ------------------------------------------------------------ backward sweep + gradient, :65-92
    // Hermitian states (X_t, L_t stay Hermitian under the sandwich) AND Hermitian control operators: with Y = X L,
    // [X, L'] = Y - Y' and tr(B Y') = conj(tr(B Y)), so Im tr(B [X, L']) = 2 Im tr(B Y) -- the second product is not formed
    const bool herm2 = SAND && p.herm_states != 0 && p.herm_ctrl != 0;
    const double gs = SAND ? (herm2 ? -2.0 * p.dt : -p.dt) : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    GT L = (win && chn + 1 < CHN) ? gt_load(p.tp_r + (kw * CHN + chn + 1) * TSZ, tile, lane)
                                  : gt_load(ops + (size_t)(2 + 2 * K) * TSZ, tile, lane);                // Xt
    GT Pt = gt_load(Pk + (size_t)(t_hi - 1) * TSZ, tile, lane);
    GT X = gt_load(Xk + (size_t)(t_hi - 1) * TSZ, tile, lane);
    int buf = 0;
    double2 sp_c[NCW][NEM];
    int sp_a[NCW][NEM];
    const int sp_ne = SPARSE ? p.sp_nz >> 6 : 0;                    // list entries per lane (1 .. 4)
    if (SPARSE) {
#pragma unroll
        for (int ci = 0; ci < NCW; ++ci)
#pragma unroll
            for (int e = 0; e < NEM; ++e) {
                const int c = wave + ci * WAVES;
                const bool on = c < K && e < sp_ne;
                const size_t at = ((size_t)k * K + (on ? c : 0)) * p.sp_nz + (on ? e : 0) * 64 + lane;
                sp_c[ci][e] = on ? p.sp_coef[at] : make_double2(0.0, 0.0);
                sp_a[ci][e] = on ? p.sp_addr[at] : 0;
            }
    }
    for (int t = t_hi - 1; t >= t_lo; --t) {
        const int tp = max(t - 1, t_lo);
        grid_barrier();
        grid_put<NT, true>(img0, Pt, I, J, lane);                               // P transposed: P' as a left operand, P as a right one
        grid_put<NT, SAND == 0>(img1, L, I, J, lane);                           // L: left operand of L P (plain) / right operand of P' L (transposed)
        grid_barrier();
        Pt = gt_load(Pk + (size_t)tp * TSZ, tile, lane);
        if (SAND) {
            const GT Y = grid_mma<NT, false, false>(img1, img0, I, J, lane);    // L P             (:248)
            grid_barrier();
            grid_put<NT, true>(img1, Y, I, J, lane);
            grid_barrier();
            L = grid_mma<NT, true, false>(img0, img1, I, J, lane);              // P' (L P)        (:249)
        } else {
            L = grid_mma<NT, true, false>(img0, img1, I, J, lane);              // P' L            (:228)
        }
        if (KEEPL)
            gt_store(p.costates + (kw * N + t) * TSZ, tile, lane, L);
        double zr_p = 0.0, zi_p = 0.0;                  // this lane's share of tr(X' L)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zr_p = fma(X.re[r], L.re[r], zr_p);
            zr_p = fma(X.im[r], L.im[r], zr_p);
            zi_p = fma(X.re[r], L.im[r], zi_p);
            zi_p = fma(-X.im[r], L.re[r], zi_p);
        }
        grid_barrier();
        grid_put<NT, false>(img0, X, I, J, lane);
        grid_put<NT, false>(img1, L, I, J, lane);
        grid_barrier();
        GT R = grid_mma<NT, false, true>(img0, img1, I, J, lane);               // X L'
        if (SAND && !herm2) {
            grid_barrier();
            grid_put<NT, true>(img0, X, I, J, lane);
            grid_put<NT, true>(img1, L, I, J, lane);
            grid_barrier();
            const GT R2 = grid_mma<NT, true, false>(img1, img0, I, J, lane);    // L' X           ([X, L'], src/tools.jl:17-19)
            R.re -= R2.re;
            R.im -= R2.im;
        }
        X = gt_load(Xk + (size_t)tp * TSZ, tile, lane);                         // (in flight under the traces)
        if (SPARSE) {
            grid_barrier();                             // every wave has read its operands of R
            grid_put<NT, false>(img0, R, I, J, lane);
            grid_barrier();
#pragma unroll
            for (int ci = 0; ci < NCW; ++ci) {
                const int c = wave + ci * WAVES;
                if (c < K) {                            // (wave-uniform)
                    double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int e = 0; e < NEM; ++e)
                        if (e < sp_ne) {
                            const double rr = img0[sp_a[ci][e]], ri = img0[PLANE + sp_a[ci][e]];
                            v[0] = fma(sp_c[ci][e].x, rr, v[0]);
                            v[0] = fma(-sp_c[ci][e].y, ri, v[0]);
                            v[1] = fma(sp_c[ci][e].x, ri, v[1]);
                            v[1] = fma(sp_c[ci][e].y, rr, v[1]);
                        }
                    if (!SAND && lane < (DIM < 64 ? DIM : 64)) {          // tr R = conj(tr(X' L))
                        v[2] = img0[lane * P + lane];
                        v[3] = img0[PLANE + lane * P + lane];
                    }
                    wave_sum_n(v);
                    if (lane == 0 && emit) {
                        const double zr = v[2], zi = -v[3];
                        const double im = SAND ? v[1] : fma(v[0], zi, v[1] * zr);
                        out[c + (size_t)t * K] = gs * im;
                        if (!SAND && t == N - 1 && c == 0)
                            out[(size_t)K * N] = zr * zr - zi * zi;              // Re(z^2), src/cost_functions.jl:99-101
                    }
                }
            }
            if (SAND && t == N - 1) {                   // the figure of merit's tr(X' L): this slice only, over the workgroup
                double v2[2] = {zr_p, zi_p};
                wave_sum_n(v2);
                if (lane == 0) {
                    s_red[2 * wave] = v2[0];
                    s_red[2 * wave + 1] = v2[1];
                }
                grid_barrier();
                if (threadIdx.x == 0 && emit) {
                    double zr = 0.0, zi = 0.0;
                    for (int w = 0; w < WAVES; ++w) {
                        zr += s_red[2 * w];
                        zi += s_red[2 * w + 1];
                    }
                    const double inv = 1.0 / (double)p.n;
                    const double ar = zr * inv, ai = zi * inv;
                    out[(size_t)K * N] = 1.0 - (ar * ar + ai * ai);              // src/cost_functions.jl:13-17
                }
            }
        } else
        // traces: sum_ij B_c[i][j] R[j][i] = sum over the elements of (B_c^T .* R); kGridGroup controls per reduction, the
        // workgroup's sum in wave order (deterministic)
        for (int c0 = 0; c0 < K; c0 += kGridGroup) {
            double v[kGridRed];
            v[0] = zr_p;
            v[1] = zi_p;
#pragma unroll
            for (int cc = 0; cc < kGridGroup; ++cc) {
                double wr = 0.0, wi = 0.0;
                if (c0 + cc < K) {
                    const GT BT = gt_load(opBT + (size_t)(c0 + cc) * TSZ, tile, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        wr = fma(BT.re[r], R.re[r], wr);
                        wr = fma(-BT.im[r], R.im[r], wr);
                        wi = fma(BT.re[r], R.im[r], wi);
                        wi = fma(BT.im[r], R.re[r], wi);
                    }
                }
                v[2 + 2 * cc] = wr;
                v[3 + 2 * cc] = wi;
            }
            wave_sum_n(v);
            double *red = s_red + (size_t)buf * WAVES * kGridRed;
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < kGridRed; ++q)
                    red[wave * kGridRed + q] = v[q];
            }
            grid_barrier();
            if ((int)threadIdx.x < kGridGroup && c0 + (int)threadIdx.x < K && emit) {
                const int cc = threadIdx.x;
                double zr = 0.0, zi = 0.0, wr = 0.0, wi = 0.0;
                for (int w = 0; w < WAVES; ++w) {
                    zr += red[w * kGridRed];
                    zi += red[w * kGridRed + 1];
                    wr += red[w * kGridRed + 2 + 2 * cc];
                    wi += red[w * kGridRed + 3 + 2 * cc];
                }
                const double im = SAND ? wi : fma(wr, zi, wi * zr);
                out[c0 + cc + (size_t)t * K] = gs * im;
                if (t == N - 1 && c0 + cc == 0) {       // figure of merit at t = N (:77, :94)
                    if (SAND) {
                        const double inv = 1.0 / (double)p.n;
                        const double ar = zr * inv, ai = zi * inv;
                        out[(size_t)K * N] = 1.0 - (ar * ar + ai * ai);          // src/cost_functions.jl:13-17
                    } else {
                        out[(size_t)K * N] = zr * zr - zi * zi;                  // Re(z^2), :99-101
                    }
                }
            }
            buf ^= 1;                                   // (two buffers: the next group's writes meet no reader)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Chunked time axis, step 1: Q_c = P_(hi-1) ... P_lo of chunk c = blockIdx.x of member blockIdx.y (control array blockIdx.z),
// tp_S - 1 products; the boundary scan and the per-chunk chains are grid_chain_kernel itself (launch_grid_nt).
template <int NT>
__global__ __launch_bounds__(64 * NT * NT) void grid_chunk_product_kernel(const TileParams p)
{
    using G_ = GridGeom<NT>;
    constexpr int TSZ = G_::TSZ, PLANE = G_::PLANE;
    extern __shared__ double s_grid[];
    double *img0 = s_grid, *img1 = s_grid + 2 * PLANE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave / NT, J = wave % NT, tile = I * NT + J;
    const int c = blockIdx.x, k = blockIdx.y, N = p.N;
    const size_t kw = (size_t)blockIdx.z * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    const int lo = c * p.tp_S, hi = min(N, lo + p.tp_S);
    GT Q = gt_load(Pk + (size_t)lo * TSZ, tile, lane);
    GT Pt = gt_load(Pk + (size_t)min(lo + 1, hi - 1) * TSZ, tile, lane);
    for (int t = lo + 1; t < hi; ++t) {
        grid_barrier();
        grid_put<NT, false>(img0, Pt, I, J, lane);
        grid_put<NT, true>(img1, Q, I, J, lane);
        grid_barrier();
        Pt = gt_load(Pk + (size_t)min(t + 1, hi - 1) * TSZ, tile, lane);        // next slice's tile in flight under the products
        Q = grid_mma<NT, false, false>(img0, img1, I, J, lane);                 // P_t Q
    }
    gt_store(p.tp_q + (kw * p.tp_chunks + c) * TSZ, tile, lane, Q);
}

// ---------------------------------------------------------------------------------------------------------------------
// Rank-one states at n = 33..64 (sparse control operators): the chain on VECTORS.  X_t = v_t v_t', L_t = w_t w_t' under the
// sandwich (pure-state density operators, the vec(rho) vec(rho)' operands of a Liouville-space transfer -- a three-qubit
// Liouvillian is 64 x 64), or X_t = v_t, L_t = w_t for n x 1 states under left multiplication (test/liou.jl:38-48):
//     v_{t+1} = P_t v_t,   w_t = P_t' w_{t+1},   s = w_t' v_t (the same for every t),   a_c = w_t' B_c v_t,   b_c = v_t' B_c w_t,
//     sandwich  g[c,t] = -dt Im(conj(s) a_c - s b_c),  F = 1 - (|s|^2 / n)^2;   left multiplication  g = -/+ 2 dt Im(a_c conj(s)),
//     F = Re(conj(s)^2)      (src/GRAPE.jl:261-303, src/cost_functions.jl:13-17, :99-111 with the rank-one factors put in)
// -- the numbers of the dense formulas up to rounding (tests compare with the oracle's DENSE evaluation), for two
// matrix-vector products per slice instead of six (three) matrix products: the chain is bound by reading P_t twice.
// One workgroup per (member, control array), wave (I, J) owns tile (I, J) of P_t: a partial product per wave, summed over
// the tile row / column through LDS.  Controls: (coefficient, position) lists as in the dense chain, wave c owns control c.
template <int NT, int SAND>
__global__ __launch_bounds__(64 * NT * NT) void grid_thin_kernel(const TileParams p)
{
    using G_ = GridGeom<NT>;
    constexpr int TSZ = G_::TSZ, DIM = G_::DIM, WAVES = G_::WAVES, P = G_::P;
    __shared__ double2 s_v[2][DIM], s_w[2][DIM], s_part[NT][DIM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave / NT, J = wave % NT, tile = I * NT + J;
    const int lo = lane & 15, hi = lane >> 4, tid = threadIdx.x;
    const int k = blockIdx.x, K = p.K, N = p.N;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    double2 *__restrict__ V = p.states + kw * (size_t)(N + 1) * DIM;              // records v_0 .. v_N
    double *__restrict__ out = p.member_out + ((size_t)blockIdx.y * p.E_members + k) * ((size_t)K * N + 1);
    const double2 *__restrict__ v0 = p.vecs + (size_t)k * 2 * DIM, *__restrict__ wT = v0 + DIM;
    // ------------------------------------------------------------ forward: v_{t+1} = P_t v_t
    if (tid < DIM) {
        s_v[0][tid] = v0[tid];
        V[tid] = v0[tid];
    }
    GT Pt = gt_load(Pk, tile, lane);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < N; ++t) {
        const double2 vj = s_v[cur][16 * J + lo];
        double acc[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                          // row 16 I + 4 r + hi, column 16 J + lo
            acc[2 * r] = fma(Pt.re[r], vj.x, -Pt.im[r] * vj.y);
            acc[2 * r + 1] = fma(Pt.re[r], vj.y, Pt.im[r] * vj.x);
        }
        Pt = gt_load(Pk + (size_t)min(t + 1, N - 1) * TSZ, tile, lane);       // next slice in flight
        row_sum_n(acc);                                                        // over the 16 columns of the tile
        if (lo == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_part[J][16 * I + 4 * r + hi] = make_double2(acc[2 * r], acc[2 * r + 1]);
        }
        grid_barrier();
        if (tid < DIM) {
            double yr = 0.0, yi = 0.0;
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
                yr += s_part[jj][tid].x;
                yi += s_part[jj][tid].y;
            }
            s_v[cur ^ 1][tid] = make_double2(yr, yi);
            V[(size_t)(t + 1) * DIM + tid] = make_double2(yr, yi);
        }
        grid_barrier();
        cur ^= 1;
    }
    // ------------------------------------------------------------ backward: w_t = P_t' w_{t+1}, forms, gradient
    constexpr int NCW = (16 + WAVES - 1) / WAVES;                              // controls a wave may own (K <= 16)
    double2 sp_c[NCW][4];
    int sp_i[NCW][4], sp_j[NCW][4];
    const int sp_ne = p.sp_nz >> 6;
#pragma unroll
    for (int ci = 0; ci < NCW; ++ci)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = wave + ci * WAVES;
            const bool on = c < K && e < sp_ne;
            const size_t at = ((size_t)k * K + (on ? c : 0)) * p.sp_nz + (on ? e : 0) * 64 + lane;
            sp_c[ci][e] = on ? p.sp_coef[at] : make_double2(0.0, 0.0);
            const int addr = on ? p.sp_addr[at] : 0;                           // position of R[j][i] in an image of pitch P
            sp_j[ci][e] = addr / P;
            sp_i[ci][e] = addr - (addr / P) * P;
        }
    const double gs = SAND ? -p.dt : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);
    __syncthreads();                                   // (the forward pass's records have left: this workgroup reads them back)
    if (tid < DIM)
        s_w[0][tid] = wT[tid];
    Pt = gt_load(Pk + (size_t)(N - 1) * TSZ, tile, lane);
    double2 v_next = tid < DIM ? V[(size_t)(N - 1) * DIM + tid] : make_double2(0.0, 0.0);
    grid_barrier();
    int wc = 0;
    for (int t = N - 1; t >= 0; --t) {
        if (tid < DIM) {
            s_v[t & 1][tid] = v_next;                                          // v_t for the forms (double-buffered by slice parity)
            v_next = V[(size_t)max(t - 1, 0) * DIM + tid];                     // (this thread stored it in the forward pass)
        }
        double ar = 0.0, ai = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                          // conj(P[row][col]) w[row], summed over the tile's rows
            const double2 wv = s_w[wc][16 * I + 4 * r + hi];
            ar = fma(Pt.re[r], wv.x, ar);
            ar = fma(Pt.im[r], wv.y, ar);
            ai = fma(Pt.re[r], wv.y, ai);
            ai = fma(-Pt.im[r], wv.x, ai);
        }
        Pt = gt_load(Pk + (size_t)max(t - 1, 0) * TSZ, tile, lane);
        double acc[2] = {ar, ai};
        col_sum_n(acc);                                                        // over the four lane groups (rows 4 r + hi)
        if (hi == 0)
            s_part[I][16 * J + lo] = make_double2(acc[0], acc[1]);
        grid_barrier();
        if (tid < DIM) {
            double yr = 0.0, yi = 0.0;
#pragma unroll
            for (int ii = 0; ii < NT; ++ii) {
                yr += s_part[ii][tid].x;
                yi += s_part[ii][tid].y;
            }
            s_w[wc ^ 1][tid] = make_double2(yr, yi);
        }
        grid_barrier();
        wc ^= 1;
        const double2 *vt = s_v[t & 1], *wt = s_w[wc];
#pragma unroll
        for (int ci = 0; ci < NCW; ++ci) {
            const int c = wave + ci * WAVES;
            if (c < K) {                               // (wave-uniform)
                double v6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};                  // a, b, s
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < sp_ne) {
                        const double2 b = sp_c[ci][e];
                        const double2 wi_ = wt[sp_i[ci][e]], vj_ = vt[sp_j[ci][e]], vi_ = vt[sp_i[ci][e]], wj_ = wt[sp_j[ci][e]];
                        // a += conj(w_i) B_ij v_j
                        const double t1r = b.x * vj_.x - b.y * vj_.y, t1i = b.x * vj_.y + b.y * vj_.x;
                        v6[0] = fma(wi_.x, t1r, v6[0]);
                        v6[0] = fma(wi_.y, t1i, v6[0]);
                        v6[1] = fma(wi_.x, t1i, v6[1]);
                        v6[1] = fma(-wi_.y, t1r, v6[1]);
                        // b += conj(v_i) B_ij w_j
                        const double t2r = b.x * wj_.x - b.y * wj_.y, t2i = b.x * wj_.y + b.y * wj_.x;
                        v6[2] = fma(vi_.x, t2r, v6[2]);
                        v6[2] = fma(vi_.y, t2i, v6[2]);
                        v6[3] = fma(vi_.x, t2i, v6[3]);
                        v6[3] = fma(-vi_.y, t2r, v6[3]);
                    }
                if (lane < (DIM < 64 ? DIM : 64)) {     // s = w' v
                    const double2 wv = wt[lane], vv = vt[lane];
                    v6[4] = wv.x * vv.x + wv.y * vv.y;
                    v6[5] = wv.x * vv.y - wv.y * vv.x;
                }
                wave_sum_n(v6);
                if (lane == 0) {
                    const double sr = v6[4], si = v6[5];
                    double im;
                    if (SAND)                           // Im(conj(s) a - s b)
                        im = (sr * v6[1] - si * v6[0]) - (sr * v6[3] + si * v6[2]);
                    else                                // Im(a conj(s))
                        im = v6[1] * sr - v6[0] * si;
                    out[c + (size_t)t * K] = gs * im;
                    if (t == N - 1 && c == 0) {
                        if (SAND) {
                            const double q = (sr * sr + si * si) / (double)p.n;
                            out[(size_t)K * N] = 1.0 - q * q;
                        } else {
                            out[(size_t)K * N] = sr * sr - si * si;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact control gradient / ADGRAPE functional for n = 33..64: exact_tile.hip's algorithm (read its header and exact_grad.hip's)
// on the workgroup-owned matrices -- from the debug flow's stored P_t, X_t, L_{t+1}:
//     dPhi/dx[c,t] = tr(dP_t[c] W1) [+ conj(tr(dP_t[c] W2))],  UnitaryGate W1 = X_t L_{t+1}',  sandwich W1 = X_t (L_{t+1} P_t)',
//     W2 = (P_t X_t)' L_{t+1};  tr(DF_G[B] W) = tr(DF_G[W] B) for the Taylor-8 + squaring polynomial F: ONE forward-mode
//     derivative per slice in the direction W1 (W2), a control costs a trace.  One workgroup per (member, slice).
template <int NT, bool HA, bool HB>
GRAPE_DEV GT grid_prod(double *img0, double *img1, const GT &a, const GT &b, int I, int J, int lane)
{
    grid_barrier();
    grid_put<NT, HA>(img0, a, I, J, lane);               // left operand [row][k]: plain, or transposed for a conjugate transpose
    grid_put<NT, !HB>(img1, b, I, J, lane);              // right operand [column][k]: transposed, or plain for a conjugate transpose
    grid_barrier();
    return grid_mma<NT, HA, HB>(img0, img1, I, J, lane);
}

// sum over the workgroup of M complex values per wave (wave-summed here), wave order: every thread gets the totals
template <int NT, int M>
GRAPE_DEV void grid_block_sum(double (&v)[M], double *s_red, int wave, int lane)
{
    constexpr int WAVES = NT * NT;
    wave_sum_n(v);
    grid_barrier();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < M; ++q)
            s_red[wave * M + q] = v[q];
    }
    grid_barrier();
#pragma unroll
    for (int q = 0; q < M; ++q) {
        double t = 0.0;
        for (int w = 0; w < WAVES; ++w)
            t += s_red[w * M + q];
        v[q] = t;
    }
}

GRAPE_DEV void gt_axpy(GT &o, double a, const GT &x)
{
    o.re += a * x.re;
    o.im += a * x.im;
}
GRAPE_DEV GT gt_lin2(double a, const GT &x, double b, const GT &y)
{
    GT o;
    o.re = a * x.re + b * y.re;
    o.im = a * x.im + b * y.im;
    return o;
}

template <int NT, int SAND>
__global__ __launch_bounds__(64 * NT * NT) void grid_exact_kernel(const TileParams p, int objective)
{
    using G_ = GridGeom<NT>;
    constexpr int TSZ = G_::TSZ, PLANE = G_::PLANE, DIM = G_::DIM;
    extern __shared__ double s_grid[];
    double *img0 = s_grid, *img1 = s_grid + 2 * PLANE, *s_red = s_grid + 4 * PLANE;      // s_red[WAVES][kGridRed]; s_col behind it
    double *s_col = s_red + NT * NT * kGridRed;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, I = wave / NT, J = wave % NT, tile = I * NT + J;
    const int k = blockIdx.y, t = blockIdx.x, K = p.K, N = p.N;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;             // [A | B_c | B_c^T | Xi | Xt]
    auto prod = [&](const GT &a, const GT &b) { return grid_prod<NT, false, false>(img0, img1, a, b, I, J, lane); };
    const GT P = gt_load(p.props + ((size_t)k * N + t) * TSZ, tile, lane);
    const GT X = gt_load(p.states + ((size_t)k * N + t) * TSZ, tile, lane);
    const GT L = t + 1 < N ? gt_load(p.costates + ((size_t)k * N + t + 1) * TSZ, tile, lane)
                           : gt_load(ops + (size_t)(2 + 2 * K) * TSZ, tile, lane);      // Xt
    // W1, W2 and Phi
    GT W1, W2;
    double ph[2];
    {
        const GT V = prod(P, X);                                                         // V = P X
        if (SAND) {
            const GT Z = prod(L, P);                                                     // Z = L P
            ph[0] = 0.0;
            ph[1] = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                                // Phi = tr((L P)' (P X))
                ph[0] = fma(Z.re[r], V.re[r], ph[0]);
                ph[0] = fma(Z.im[r], V.im[r], ph[0]);
                ph[1] = fma(Z.re[r], V.im[r], ph[1]);
                ph[1] = fma(-Z.im[r], V.re[r], ph[1]);
            }
            W1 = grid_prod<NT, false, true>(img0, img1, X, Z, I, J, lane);               // W1 = X (L P)'
            W2 = grid_prod<NT, true, false>(img0, img1, V, L, I, J, lane);               // W2 = (P X)' L
        } else {
            ph[0] = 0.0;
            ph[1] = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                                // Phi = tr(L' P X)
                ph[0] = fma(L.re[r], V.re[r], ph[0]);
                ph[0] = fma(L.im[r], V.im[r], ph[0]);
                ph[1] = fma(L.re[r], V.im[r], ph[1]);
                ph[1] = fma(-L.im[r], V.re[r], ph[1]);
            }
            W1 = grid_prod<NT, false, true>(img0, img1, X, L, I, J, lane);               // W1 = X L'
            W2 = W1;
        }
        grid_block_sum<NT, 2>(ph, s_red, wave, lane);
    }
    const double phr = ph[0], phi = ph[1];
    // generator (as grid_prop_kernel's own build) and the shared part of the Taylor evaluation
    GT G;
    if (p.variant == 0) {
        G.re = (d4){0, 0, 0, 0};
        G.im = (d4){0, 0, 0, 0};
    } else {
        G = gt_load(ops, tile, lane);
    }
    for (int c = 0; c < K; ++c) {
        const double xv = p.x[c + (size_t)t * K];
        const GT B = gt_load(ops + (size_t)(1 + c) * TSZ, tile, lane);
        gt_axpy(G, xv, B);
    }
    if (p.variant == 0) {
        const GT A = gt_load(ops, tile, lane);
        gt_axpy(G, 1.0, A);
    }
    const double dt = p.dt;
    double cs = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double hr = G.re[r], hi = G.im[r];
        G.re[r] = dt * hi;
        G.im[r] = -dt * hr;
        cs += fabs(G.re[r]) + fabs(G.im[r]);
    }
    cs = swap16_add(cs, cs);
    cs = swap32_add(cs, cs);
    grid_barrier();
    if (lane < 16)
        s_col[I * DIM + 16 * J + lane] = cs;
    grid_barrier();
    double colmax = 0.0;
    if (lane < DIM) {
#pragma unroll
        for (int ii = 0; ii < NT; ++ii)
            colmax += s_col[ii * DIM + lane];
    }
    colmax = wave_max_fast(colmax);
    const int s = p.s_forced >= 0 ? p.s_forced : squarings_for(colmax);
    const double sc = s > 0 ? ldexp(1.0, -s) : 1.0;
    if (s > 0) {
        G.re *= sc;
        G.im *= sc;
    }
    const GT A2 = prod(G, G);
    const GT T1 = gt_lin2(kX1, G, kX2, A2);
    const GT A4 = prod(A2, T1);
    const GT U = gt_lin2(kX3, A2, 1.0, A4);
    GT T2 = gt_lin2(kX5, G, kX6, A2);
    gt_axpy(T2, kX7, A4);
    gt_add_identity(T2, kX4, I, J, lane);
    GT Ps;
    if (s > 0) {                                                   // value at the scaled point, for the squaring chain rule
        Ps = prod(U, T2);
        gt_axpy(Ps, 1.0, G);
        gt_axpy(Ps, kY2, A2);
        gt_add_identity(Ps, 1.0, I, J, lane);
    }
    auto frechet = [&](const GT &W) {
        const GT E = gt_lin2(sc, W, 0.0, W);
        GT dA2 = prod(E, G);
        gt_axpy(dA2, 1.0, prod(G, E));
        const GT dT1 = gt_lin2(kX1, E, kX2, dA2);
        GT dA4 = prod(dA2, T1);
        gt_axpy(dA4, 1.0, prod(A2, dT1));
        const GT dU = gt_lin2(kX3, dA2, 1.0, dA4);
        GT dT2 = gt_lin2(kX5, E, kX6, dA2);
        gt_axpy(dT2, kX7, dA4);
        GT dP = prod(dU, T2);
        gt_axpy(dP, 1.0, prod(U, dT2));
        gt_axpy(dP, 1.0, E);
        gt_axpy(dP, kY2, dA2);
        if (s > 0) {
            GT Pq = Ps;
            for (int i = 0; i < s; ++i) {
                GT tmp = prod(dP, Pq);
                gt_axpy(tmp, 1.0, prod(Pq, dP));
                dP = tmp;
                Pq = prod(Pq, Pq);
            }
        }
        return dP;
    };
    const GT D1 = frechet(W1);
    const bool second = SAND && !p.herm_states;                    // Hermitian X, L: W2 == W1
    GT D2m = D1;
    if (second)
        D2m = frechet(W2);
    double *__restrict__ out = p.member_out + (size_t)k * ((size_t)K * N + 1);
    const double Dn2 = 1.0 / ((double)p.n * (double)p.n);
    const bool c1 = SAND || objective == 1;
    for (int c = 0; c < K; ++c) {
        const GT BT = gt_load(ops + (size_t)(1 + K + c) * TSZ, tile, lane);              // B_c^T: tr(D B_c) = sum D .* B_c^T
        double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[0] = fma(D1.re[r], BT.re[r], v[0]);
            v[0] = fma(-D1.im[r], BT.im[r], v[0]);
            v[1] = fma(D1.re[r], BT.im[r], v[1]);
            v[1] = fma(D1.im[r], BT.re[r], v[1]);
            v[2] = fma(D2m.re[r], BT.re[r], v[2]);
            v[2] = fma(-D2m.im[r], BT.im[r], v[2]);
            v[3] = fma(D2m.re[r], BT.im[r], v[3]);
            v[3] = fma(D2m.im[r], BT.re[r], v[3]);
        }
        grid_block_sum<NT, 4>(v, s_red, wave, lane);
        double dr = dt * v[1], di = -dt * v[0];                                          // B'_c = (-i dt) B_c
        if (SAND) {
            if (second) {
                dr += dt * v[3];
                di -= -dt * v[2];
            } else {
                dr += dr;                                                                // + conj of the same number
                di = 0.0;
            }
        }
        const double g = c1 ? -2.0 * Dn2 * (phr * dr + phi * di) : 2.0 * (phr * dr - phi * di);
        if (threadIdx.x == 0)
            out[c + (size_t)t * K] = g;
    }
    if (t == N - 1 && threadIdx.x == 0)
        out[(size_t)K * N] = c1 ? 1.0 - Dn2 * (phr * phr + phi * phi) : phr * phr - phi * phi;
}

template <int NT>
static hipError_t launch_grid_exact_nt(int sandwich, const TileParams &p, int objective, hipStream_t stream)
{
    const size_t dim = 16 * (size_t)NT;
    const size_t lds = sizeof(double) * (4 * dim * (dim + 2) + (size_t)NT * NT * kGridRed + (size_t)NT * dim);
    const dim3 grid(p.N, p.E), block(64 * NT * NT);
    hipError_t e;
    if (sandwich) {
        e = ensure_dynamic_lds((const void *)grid_exact_kernel<NT, 1>, lds);
        if (e != hipSuccess)
            return e;
        GRAPE_LAUNCH((grid_exact_kernel<NT, 1>), grid, block, lds, stream, p, objective);
    } else {
        e = ensure_dynamic_lds((const void *)grid_exact_kernel<NT, 0>, lds);
        if (e != hipSuccess)
            return e;
        GRAPE_LAUNCH((grid_exact_kernel<NT, 0>), grid, block, lds, stream, p, objective);
    }
    return hipGetLastError();
}

hipError_t launch_grid_exact(int NT, int sandwich, const TileParams &p, int objective, hipStream_t stream)
{
    switch (NT) {
    case 3: return launch_grid_exact_nt<3>(sandwich, p, objective, stream);
    case 4: return launch_grid_exact_nt<4>(sandwich, p, objective, stream);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the expm launches: [ctrl_sum_kernel +] grid_prop_kernel
template <int NT>
static hipError_t launch_grid_prop_nt(const TileParams &p, hipStream_t stream)
{
    TileParams q = p;
    const size_t lds_p = grid_lds_bytes(NT, false);
    // slices per workgroup of the expm kernel: about two rounds of workgroups over the device, at most 64 slices
    const long cus = p.cus > 0 ? p.cus : 256, total = (long)p.N * p.E * p.n_x;
    q.prop_slices = (int)std::min<long>(64, std::max<long>(1, (total + 2 * cus - 1) / (2 * cus)));
    hipError_t e;
    const dim3 pgrid((p.N + q.prop_slices - 1) / q.prop_slices, p.E, p.n_x);
    if (p.hoist == 1) {                                            // member-invariant controls: the control sum once per slice
        e = launch_ctrl_sum(NT, q, stream);
        if (e != hipSuccess)
            return e;
        e = ensure_dynamic_lds((const void *)grid_prop_kernel<NT, true>, lds_p);
        if (e != hipSuccess)
            return e;
        GRAPE_LAUNCH((grid_prop_kernel<NT, true>), pgrid, dim3(64 * NT * NT), lds_p, stream, q);
    } else {
        e = ensure_dynamic_lds((const void *)grid_prop_kernel<NT, false>, lds_p);
        if (e != hipSuccess)
            return e;
        GRAPE_LAUNCH((grid_prop_kernel<NT, false>), pgrid, dim3(64 * NT * NT), lds_p, stream, q);
    }
    return hipGetLastError();
}

// (called by the tile family's launcher too: at 32 x 32 with member-invariant controls and an ensemble that fills the device
// this expm kernel runs C5's 8.2 M propagators in 85.5 ms at 0.72 of the matrix pipe, prop_hoist2_kernel in 96-97 ms at 0.64)
hipError_t launch_grid_prop(int NT, const TileParams &p, hipStream_t stream)
{
    switch (NT) {
    case 1: return launch_grid_prop_nt<1>(p, stream);
    case 2: return launch_grid_prop_nt<2>(p, stream);
    case 3: return launch_grid_prop_nt<3>(p, stream);
    case 4: return launch_grid_prop_nt<4>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int NT>
static hipError_t launch_grid_nt(int sandwich, bool keepl, const TileParams &p, hipStream_t stream)
{
    TileParams q = p;
    const size_t lds_c = grid_lds_bytes(NT, true);
    hipError_t e = launch_grid_prop_nt<NT>(p, stream);
    if (e != hipSuccess)
        return e;
    if (p.ev_mid) {
        e = hipEventRecord(p.ev_mid, stream);
        if (e != hipSuccess)
            return e;
    }
    const dim3 grid(p.E, p.n_x), block(64 * NT * NT);
    if constexpr (NT >= 3) {
        if (p.thin && p.sparse && p.K <= 16 && p.sp_nz <= 256) {       // rank-one states: the chain on vectors
            if (sandwich) GRAPE_LAUNCH((grid_thin_kernel<NT, 1>), grid, block, 0, stream, q);
            else          GRAPE_LAUNCH((grid_thin_kernel<NT, 0>), grid, block, 0, stream, q);
            return hipGetLastError();
        }
    }
    // the chain kernel for these parameters on `cgrid` (costates stored or not)
    auto run_chain = [&](const TileParams &cq, dim3 cgrid, bool kl) -> hipError_t {
        hipError_t ce = hipSuccess;
#define GRAPE_GRID_CHAIN(S, KL, SP)                                                                                      \
    {                                                                                                                    \
        ce = ensure_dynamic_lds((const void *)grid_chain_kernel<NT, S, KL, SP>, lds_c);                                  \
        if (ce != hipSuccess)                                                                                            \
            return ce;                                                                                                   \
        GRAPE_LAUNCH((grid_chain_kernel<NT, S, KL, SP>), cgrid, block, lds_c, stream, cq);                               \
    }
        const int spne = (cq.sparse && cq.K <= 16 && cq.sp_nz <= 256 && NT >= 2) ? (cq.sp_nz <= 64 ? 1 : 4) : 0;
        if constexpr (NT >= 2) {
            if (spne == 1) {
                if (sandwich) {
                    if (kl) GRAPE_GRID_CHAIN(1, true, 1) else GRAPE_GRID_CHAIN(1, false, 1)
                } else {
                    if (kl) GRAPE_GRID_CHAIN(0, true, 1) else GRAPE_GRID_CHAIN(0, false, 1)
                }
                return hipGetLastError();
            }
        }
        if constexpr (NT >= 3) {
            if (spne == 4) {
                if (sandwich) {
                    if (kl) GRAPE_GRID_CHAIN(1, true, 4) else GRAPE_GRID_CHAIN(1, false, 4)
                } else {
                    if (kl) GRAPE_GRID_CHAIN(0, true, 4) else GRAPE_GRID_CHAIN(0, false, 4)
                }
                return hipGetLastError();
            }
        }
        if (sandwich) {
            if (kl) GRAPE_GRID_CHAIN(1, true, 0) else GRAPE_GRID_CHAIN(1, false, 0)
        } else {
            if (kl) GRAPE_GRID_CHAIN(0, true, 0) else GRAPE_GRID_CHAIN(0, false, 0)
        }
#undef GRAPE_GRID_CHAIN
        return hipGetLastError();
    };
    if (p.tp_chunks > 1 && p.tp_q && p.tp_u && p.tp_r) {
        // Chunked time axis (single problems, ensembles far smaller than the device): chunk products -> boundary scan (the chain
        // kernel on the chunk products: states at the chunks' starts into tp_u, costates at their starts into tp_r) -> one
        // workgroup per (member, chunk).  Dependent products per evaluation: (tp_S - 1) + 2..4 C + 3..6 tp_S instead of 3..6 N.
        const size_t lds_q = grid_lds_bytes(NT, false);
        e = ensure_dynamic_lds((const void *)grid_chunk_product_kernel<NT>, lds_q);
        if (e != hipSuccess)
            return e;
        GRAPE_LAUNCH((grid_chunk_product_kernel<NT>), dim3(p.tp_chunks, p.E, p.n_x), block, lds_q, stream, q);
        TileParams sc = q;
        sc.props = q.tp_q;
        sc.N = q.tp_chunks;
        sc.states = q.tp_u;
        sc.costates = q.tp_r;
        sc.tp_chunks = 0;
        sc.tp_window = 0;
        sc.tp_scan = 1;
        e = run_chain(sc, grid, true);
        if (e != hipSuccess)
            return e;
        TileParams w = q;
        w.tp_window = 1;
        w.tp_scan = 0;
        return run_chain(w, dim3(p.E, p.n_x, p.tp_chunks), keepl);
    }
    q.tp_window = 0;
    q.tp_scan = 0;
    return run_chain(q, grid, keepl);
}

hipError_t launch_sweep_grid(int NT, int sandwich, bool keep_costates, const TileParams &p, hipStream_t stream)
{
    switch (NT) {
    case 1: return launch_grid_nt<1>(sandwich, keep_costates, p, stream);     // (GRAPE_GRID=1: the small sizes through these
    case 2: return launch_grid_nt<2>(sandwich, keep_costates, p, stream);     //  kernels, for cross-checks against the tile family)
    case 3: return launch_grid_nt<3>(sandwich, keep_costates, p, stream);
    case 4: return launch_grid_nt<4>(sandwich, keep_costates, p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace grape
