// sweep_thin.hip -- the chain of the tile family (n = 9..16) for states of RANK ONE: gfx950, one wavefront
// per ensemble member, matrix-VECTOR products instead of the dense 16 x 16 chain products.
//
// When the operands of the reference's sweeps are rank one,
//   * State/CoherenceTransfer with Xi = v0 v0', Xt = wT wT'  (pure-state density operators -- also the
//     vec(rho) vec(rho)' operands of a Liouville-space transfer), or
//   * UnitaryGate-style left multiplication of n x 1 states (vec(rho) under Liouvillian superoperators,
//     /root/reference/test/liou.jl:38-48),
// every X_t = v_t v_t' (resp. v_t) and L_t = w_t w_t' (resp. w_t) with
//     v_{t+1} = P_t v_t            (src/GRAPE.jl:226 / :245-246 applied to the factor)
//     w_t     = P_t' w_{t+1}       (src/GRAPE.jl:228 / :248-249)
// and the gradient traces (src/GRAPE.jl:261-303) collapse to two bilinear forms per control,
//     a = w_t' B_c v_t ,  b = v_t' B_c w_t ,  s = w' v  (the same for every t):
//     sandwich     g[c,t] = -dt Im(conj(s) a - s b)          F = 1 - (|s|^2 / n)^2
//     left mult.   g[c,t] = -/+ 2 dt Im(a conj(s))           F = Re(conj(s)^2)
// -- identical to the dense formulas up to rounding (tests compare with the oracle's dense evaluation at the
// 1e-10 bar).  The propagators still come from prop_tile_kernel (Taylor-8 on the FP64 matrix cores, the
// genuinely dense part); this kernel streams them twice (forward, backward): 2 x 4 KB per slice and member,
// HBM-bound, ~1/10 of the dense chain's FMAs.
//
// No layout conversions: with P_t stored in D layout for EVEN t and TRANSPOSED (the D layout of P_t^T) for
// ODD t (prop_tile_kernel does that when TileParams.thin is set), the two vector formats of a wave
//     per-column   lane l holds x[l & 15]
//     gathered     lane l holds x[4r + (l >> 4)], r = 0..3
// alternate by themselves:  forward  even t: per-column in -> sum over the 16 lanes of a row -> gathered out,
//                                    odd  t: gathered in   -> sum over the 4 rows          -> per-column out;
//                           backward even t: gathered in -> per-column out,  odd t: per-column in -> gathered out.
// The bilinear forms take B_c or B_c^T (both dumps are in the operator block) according to the format w_t
// happens to be in; v_t comes back from the forward pass's 256-byte records in whichever format is needed.
#include "grape_kernels.hpp"
#include "tile.hpp"

namespace grape {

namespace {

#ifndef GRAPE_THIN_ABL
#define GRAPE_THIN_ABL 0      // diagnostic ablations: 1 no backward pass, 2 no bilinear forms, 4 no P loads in the backward pass, 8 forward pass = streaming only
#endif
constexpr int kRing = 4;          // backward pass: slices of P in flight per wave (4 KB each)
constexpr int kRingF = 8;         // forward pass (one wave per member runs it alone)

struct Tile1 {                    // one 16 x 16 D-layout dump: 4 complex per lane
    double re[4], im[4];
};

GRAPE_DEV void load_tile(Tile1 &m, const double2 *__restrict__ src, int lane)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2 v = src[r * 64 + lane];
        m.re[r] = v.x;
        m.im[r] = v.y;
    }
}

}  // namespace

// SAND: sandwich formulas (State/CoherenceTransfer), else left multiplication (UnitaryGate, n x 1 states).
// HERMB: every control operator is Hermitian, so b = conj(a) (sandwich only; left multiplication needs a alone).
// INLDS: the member's 2K operator dumps are staged in LDS (else read from global memory / L2 every slice) -- a
// template parameter, not a run-time branch: merged code would wait for ALL vector-memory operations (the
// propagator prefetches too) before every form.
// Both passes keep a ring of propagator tiles in flight.  The steady-state loops contain no branch around a
// load and no per-slice bounds check (prefetch indices are clamped, the ragged group of slices is handled apart):
// only then does the compiler's s_waitcnt placement wait for the OLDEST tile instead of draining the ring.
template <int U>
struct IC {
    static constexpr int value = U;
};

template <int SAND, bool HERMB, bool INLDS>
__global__ __launch_bounds__(64) void chain_thin_kernel(const TileParams p)
{
    constexpr int TSZ = 256;
    extern __shared__ double2 s_thin[];           // (INLDS) this member's [B_c | B_c^T] dumps
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int k = blockIdx.x;
    // time-parallel mode (TileParams.tp_chunks, small ensembles): this wavefront owns the slices [lo, lo + N) of its
    // member; v at the chunk's start, w at its end and s come from chunk_scan_thin_kernel.  Chunks start at multiples
    // of 8, so slice parities and ring slots are those of the whole pulse.
    const int C = p.tp_chunks;
    const int lo = C ? (int)blockIdx.z * p.tp_S : 0;
    const int K = p.K, N = C ? min(p.N, lo + p.tp_S) - lo : p.N;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ opB = ops + TSZ;                 // B_c at opB + c TSZ, B_c^T at opB + (K + c) TSZ
    if (INLDS) {
        for (int i = lane; i < 2 * K * TSZ; i += 64)
            s_thin[i] = opB[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    auto load_op = [&](Tile1 &m, int idx) {
        if (INLDS)
            load_tile(m, s_thin + (size_t)idx * TSZ, lane);
        else
            load_tile(m, opB + (size_t)idx * TSZ, lane);
    };
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + (kw * p.N + lo) * TSZ;
    double2 *__restrict__ V = p.states + (kw * (size_t)(p.N + 1) + lo) * 16;   // records v_0 .. v_N, 16 complex each
    const double2 *__restrict__ v0 = C ? p.tp_vec + (kw * C + blockIdx.z) * 32 : p.vecs + (size_t)k * 32;
    const double2 *__restrict__ wT = v0 + 16;
    double *__restrict__ out_member = p.member_out + ((size_t)blockIdx.y * p.E_members + k) * ((size_t)K * p.N + 1);
    double *__restrict__ out = out_member + (size_t)lo * K;
    const int sel = c & 3;                        // which of a lane's four gathered entries it writes to a record

    // ------------------------------------------------------------------ forward: v_{t+1} = P_t v_t
    // (already done by prop_tile_kernel when it walked the member's slices in order: TileParams.fuse_fwd)
    if (!p.fuse_fwd) {
        Tile1 Pq[kRingF];
#pragma unroll
        for (int u = 0; u < kRingF; ++u)
            load_tile(Pq[u], Pk + (size_t)min(u, N - 1) * TSZ, lane);
        double vr, vi;                            // per-column format
        {
            const double2 t = v0[c];
            vr = t.x;
            vi = t.y;
        }
        double y[8];                              // gathered format: re/im of x[4r + g]
#pragma unroll
        for (int e = 0; e < 8; ++e)
            y[e] = 0.0;
        // every lane stores entry (c & 3) of its four gathered ones (4 lanes per entry: no branch, same data); the pick
        // is arithmetic (0/1 weights) -- a ternary chain on the lane index becomes an indexed array in scratch memory
        const double m0 = sel == 0 ? 1.0 : 0.0, m1 = sel == 1 ? 1.0 : 0.0, m2 = sel == 2 ? 1.0 : 0.0, m3 = sel == 3 ? 1.0 : 0.0;
        auto record_gathered = [&](int t) {
            const double sr = fma(m3, y[6], fma(m2, y[4], fma(m1, y[2], m0 * y[0])));
            const double si = fma(m3, y[7], fma(m2, y[5], fma(m1, y[3], m0 * y[1])));
            V[(size_t)t * 16 + 4 * sel + g] = make_double2(sr, si);
        };
        auto step = [&](auto uc, int t) {
            constexpr int u = decltype(uc)::value;
            const Tile1 P = Pq[u];
            load_tile(Pq[u], Pk + (size_t)min(t + kRingF, N - 1) * TSZ, lane);
            if (GRAPE_THIN_ABL & 8) {             // streaming only
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vr += P.re[r];
                    vi += P.im[r];
                }
            } else
            if ((u & 1) == 0) {                   // even t: D layout of P, reg r = P[4r + g][c]
                V[(size_t)t * 16 + c] = make_double2(vr, vi);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[2 * r] = fma(P.re[r], vr, -P.im[r] * vi);
                    y[2 * r + 1] = fma(P.re[r], vi, P.im[r] * vr);
                }
                row_sum_n(y);
            } else {                              // odd t: D layout of P^T, reg r = P[c][4r + g]
                record_gathered(t);
                double acc[2] = {0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0] = fma(P.re[r], y[2 * r], acc[0]);
                    acc[0] = fma(-P.im[r], y[2 * r + 1], acc[0]);
                    acc[1] = fma(P.re[r], y[2 * r + 1], acc[1]);
                    acc[1] = fma(P.im[r], y[2 * r], acc[1]);
                }
                col_sum_n(acc);
                vr = acc[0];
                vi = acc[1];
            }
        };
        int base = 0;
        for (; base + kRingF <= N; base += kRingF) {
            step(IC<0>(), base);
            step(IC<1>(), base + 1);
            step(IC<2>(), base + 2);
            step(IC<3>(), base + 3);
            step(IC<4>(), base + 4);
            step(IC<5>(), base + 5);
            step(IC<6>(), base + 6);
            step(IC<7>(), base + 7);
        }
        static_assert(kRingF == 8, "unrolled by hand");
        if (base + 0 < N) step(IC<0>(), base);
        if (base + 1 < N) step(IC<1>(), base + 1);
        if (base + 2 < N) step(IC<2>(), base + 2);
        if (base + 3 < N) step(IC<3>(), base + 3);
        if (base + 4 < N) step(IC<4>(), base + 4);
        if (base + 5 < N) step(IC<5>(), base + 5);
        if (base + 6 < N) step(IC<6>(), base + 6);
        // v_N: after an odd last slice it is in per-column format, after an even one gathered
        // (a chunk leaves that record to the next chunk, whose first record it is)
        if (C) {
        } else if ((N & 1) == 0)
            V[(size_t)N * 16 + c] = make_double2(vr, vi);
        else
            record_gathered(N);
    }
    // the records were written by other lanes of this wave than the ones that read them back
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // stores drained, vector L1 invalidated

    // ------------------------------------------------------------------ s = wT' v_N
    double s_re, s_im;
    if (C) {
        s_re = p.tp_z[kw * 128];
        s_im = p.tp_z[kw * 128 + 1];
    } else {
        double z[2] = {0.0, 0.0};
        if (g == 0) {
            const double2 w = wT[c], v = V[(size_t)N * 16 + c];
            z[0] = w.x * v.x + w.y * v.y;         // conj(w) v
            z[1] = w.x * v.y - w.y * v.x;
        }
        row_sum_n(z);
        s_re = __shfl(z[0], 0, 64);
        s_im = __shfl(z[1], 0, 64);
    }
    const double gs = SAND ? -p.dt * (HERMB ? 2.0 : 1.0) : (p.variant == 0 ? -2.0 * p.dt : 2.0 * p.dt);

    // ------------------------------------------------------------------ backward: w_t = P_t' w_{t+1}, gradient
    {
        static_assert(kRing == 4, "the gradient reduction batches the four slices of a ring pass");
        Tile1 Pq[kRing];
        double vq[kRing][8];                      // v_t records in the format slice t needs (even: gathered, odd: per-column)
        const int top = (N - 1) & ~(kRing - 1);
        auto load_v = [&](auto uc, int t) {
            constexpr int u = decltype(uc)::value;
            if ((u & 1) == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 t2 = V[(size_t)t * 16 + 4 * r + g];
                    vq[u][2 * r] = t2.x;
                    vq[u][2 * r + 1] = t2.y;
                }
            } else {
                const double2 t2 = V[(size_t)t * 16 + c];
                vq[u][0] = t2.x;
                vq[u][1] = t2.y;
            }
        };
        auto fill = [&](auto uc) {                // slot u first serves the largest t < N with t & 3 == u
            constexpr int u = decltype(uc)::value;
            const int t = max(top + u < N ? top + u : top + u - kRing, 0);
            load_tile(Pq[u], Pk + (size_t)t * TSZ, lane);
            load_v(uc, t);
        };
        fill(IC<0>());
        fill(IC<1>());
        fill(IC<2>());
        fill(IC<3>());
        double wr = 0.0, wi = 0.0;                // per-column format
        double w4[8];                             // gathered format
        if ((N - 1) & 1) {                        // last slice odd: its input format is per-column
            const double2 t2 = wT[c];
            wr = t2.x;
            wi = t2.y;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                w4[e] = 0.0;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 t2 = wT[4 * r + g];
                w4[2 * r] = t2.x;
                w4[2 * r + 1] = t2.y;
            }
        }
        double q16[16];                           // [slice base + u][control slot i]: per-lane partial of g
        auto step = [&](auto uc, int t) {
            constexpr int u = decltype(uc)::value;
            const Tile1 P = Pq[u];
            double vt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                vt[e] = vq[u][e];
            {
                const int tn = max(t - kRing, 0);
                if (!(GRAPE_THIN_ABL & 4))
                load_tile(Pq[u], Pk + (size_t)tn * TSZ, lane);
                load_v(uc, tn);
            }
            double q4[4] = {0.0, 0.0, 0.0, 0.0};
            auto all_forms = [&](auto form) {     // controls 0..3 go through the batched reduction
                if (GRAPE_THIN_ABL & 2)
                    return;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < K)
                        q4[i] = form(i);
                for (int cc = 4; cc < K; ++cc) {  // many controls: one by one
                    double one[1] = {form(cc)};
                    row_sum_n(one);
                    col_sum_n(one);
                    if (lane == 0)
                        out[cc + (size_t)t * K] = gs * one[0];
                }
            };
            if ((u & 1) == 0) {
                // even t: reg r = P[4r + g][c];  w_t[c] = sum conj(P[4r+g][c]) w[4r+g]: gathered in, per-column out
                double acc[2] = {0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0] = fma(P.re[r], w4[2 * r], acc[0]);
                    acc[0] = fma(P.im[r], w4[2 * r + 1], acc[0]);
                    acc[1] = fma(P.re[r], w4[2 * r + 1], acc[1]);
                    acc[1] = fma(-P.im[r], w4[2 * r], acc[1]);
                }
                col_sum_n(acc);
                wr = acc[0];
                wi = acc[1];
                // forms: w per-column, v gathered.  a = sum_{i,j} conj(w[i]) B[i][j] v[j] with the B^T dump
                // (reg r = B[c][4r + g]);  b = sum conj(v[i]) B[i][j] w[j] with the B dump (reg r = B[4r + g][c])
                all_forms([&](int cc) -> double {
                    Tile1 BT;
                    load_op(BT, K + cc);
                    double ur = 0.0, ui = 0.0;                 // sum_r B[c][4r+g] v[4r+g]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ur = fma(BT.re[r], vt[2 * r], ur);
                        ur = fma(-BT.im[r], vt[2 * r + 1], ur);
                        ui = fma(BT.re[r], vt[2 * r + 1], ui);
                        ui = fma(BT.im[r], vt[2 * r], ui);
                    }
                    const double ar = wr * ur + wi * ui, ai = wr * ui - wi * ur;      // conj(w[c]) u
                    double val = s_re * ai - s_im * ar;                                // Im(conj(s) a)
                    if (SAND && !HERMB) {
                        Tile1 Bm;
                        load_op(Bm, cc);
                        double xr = 0.0, xi = 0.0;             // sum_r conj(v[4r+g]) B[4r+g][c]
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            xr = fma(vt[2 * r], Bm.re[r], xr);
                            xr = fma(vt[2 * r + 1], Bm.im[r], xr);
                            xi = fma(vt[2 * r], Bm.im[r], xi);
                            xi = fma(-vt[2 * r + 1], Bm.re[r], xi);
                        }
                        const double br = xr * wr - xi * wi, bi = xr * wi + xi * wr;  // (..) w[c]
                        val -= s_re * bi + s_im * br;                                  // - Im(s b)
                    }
                    return val;
                });
            } else {
                // odd t: reg r = P[c][4r + g];  w_t[4r+g] = sum_c conj(P[c][4r+g]) w[c]: per-column in, gathered out
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    w4[2 * r] = fma(P.re[r], wr, P.im[r] * wi);
                    w4[2 * r + 1] = fma(P.re[r], wi, -P.im[r] * wr);
                }
                row_sum_n(w4);
                // forms: w gathered, v per-column.  a with the B dump (reg r = B[4r + g][c]);  b with the B^T dump
                const double vcr = vt[0], vci = vt[1];
                all_forms([&](int cc) -> double {
                    Tile1 Bm;
                    load_op(Bm, cc);
                    double ur = 0.0, ui = 0.0;                 // sum_r conj(w[4r+g]) B[4r+g][c]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ur = fma(w4[2 * r], Bm.re[r], ur);
                        ur = fma(w4[2 * r + 1], Bm.im[r], ur);
                        ui = fma(w4[2 * r], Bm.im[r], ui);
                        ui = fma(-w4[2 * r + 1], Bm.re[r], ui);
                    }
                    const double ar = ur * vcr - ui * vci, ai = ur * vci + ui * vcr;  // (..) v[c]
                    double val = s_re * ai - s_im * ar;
                    if (SAND && !HERMB) {
                        Tile1 BT;
                        load_op(BT, K + cc);
                        double xr = 0.0, xi = 0.0;             // sum_r B[c][4r+g] w[4r+g]
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            xr = fma(BT.re[r], w4[2 * r], xr);
                            xr = fma(-BT.im[r], w4[2 * r + 1], xr);
                            xi = fma(BT.re[r], w4[2 * r + 1], xi);
                            xi = fma(BT.im[r], w4[2 * r], xi);
                        }
                        const double br = vcr * xr + vci * xi, bi = vcr * xi - vci * xr;  // conj(v[c]) (..)
                        val -= s_re * bi + s_im * br;
                    }
                    return val;
                });
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                q16[4 * u + i] = q4[i];
        };
        // one reduce-scatter for the four slices of a group: quad (lane >> 2) = 4 u + i ends up with the sum of
        // control i of slice base + u
        auto flush = [&](int base) {
            const double tot = reduce_scatter16(q16);
            const int uu = lane >> 4, ii = (lane >> 2) & 3;
            if ((lane & 3) == 0 && base + uu < N && ii < K)
                out[ii + (size_t)(base + uu) * K] = gs * tot;
        };
        int base = (GRAPE_THIN_ABL & 1) ? -1 : top;
        if (base >= 0 && top + kRing > N) {       // the ragged last group
#pragma unroll
            for (int e = 0; e < 16; ++e)
                q16[e] = 0.0;
            if (top + 3 < N) step(IC<3>(), top + 3);
            if (top + 2 < N) step(IC<2>(), top + 2);
            if (top + 1 < N) step(IC<1>(), top + 1);
            step(IC<0>(), top);
            flush(top);
            base = top - kRing;
        }
        for (; base >= 0; base -= kRing) {
            step(IC<3>(), base + 3);
            step(IC<2>(), base + 2);
            step(IC<1>(), base + 1);
            step(IC<0>(), base);
            flush(base);
        }
    }
    if (lane == 0 && (!C || (int)blockIdx.z == C - 1)) {
        if (SAND) {
            const double z = (s_re * s_re + s_im * s_im) / (double)p.n;   // tr(L' X) / D = |s|^2 / n
            out_member[(size_t)K * p.N] = 1.0 - z * z;
        } else {
            out_member[(size_t)K * p.N] = s_re * s_re - s_im * s_im;      // Re(z^2), z = conj(s)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Time-parallel vector chain for small ensembles (see "Time-parallel unitary chain" in sweep_tile.hip): dense chunk
// products Q_c on the matrix cores (one wavefront per member and chunk), then one wavefront per member carries v
// through the chunk starts (v <- Q_c v) and w back through the chunk ends (w <- Q_c' w), and the chunks run the
// kernel above on their own slices.  The propagators are stored transposed for odd t: an odd slice is a free left
// factor (Z^T W with Z the dump), an even one goes through the layout conversion.
__global__ __launch_bounds__(64) void chunk_product_thin_kernel(const TileParams p)
{
    constexpr int TSZ = 256;
    extern __shared__ double2 s_thin[];
    const int lane = threadIdx.x, k = blockIdx.x, ch = blockIdx.z;
    const int C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const int lo = ch * p.tp_S, hi = min(p.N, lo + p.tp_S);
    const double2 *__restrict__ Pk = p.props + kw * p.N * TSZ;
    TMat<1> W, Y, Pm, Pn;
    tzero(W);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * r + (lane >> 4) == (lane & 15))
            W.re[0][0][r] = 1.0;
    tload(Pm, Pk + (size_t)lo * TSZ, lane);
    for (int t = lo; t < hi; ++t) {                                 // W <- P_t W
        tload(Pn, Pk + (size_t)min(t + 1, p.N - 1) * TSZ, lane);
        if (t & 1) {
            tmul_tn<1, false, false>(Y, Pm, W);                     // the dump is P_t^T
        } else {
            TOp<1> PA;
            to_a_layout(PA, Pm, s_thin, lane);
            tmul_an<1, false, false>(Y, PA, W);
        }
        W = Y;
        Pm = Pn;
    }
    tstore(p.tp_q + (kw * C + ch) * TSZ, W, lane);
}

__global__ __launch_bounds__(64) void chunk_scan_thin_kernel(const TileParams p)
{
    constexpr int TSZ = 256;
    __shared__ double2 s_v[16];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15, k = blockIdx.x;
    const int C = p.tp_chunks;
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Qk = p.tp_q + kw * C * TSZ;
    double2 *__restrict__ vec = p.tp_vec + kw * C * 32;
    const double2 *__restrict__ v0 = p.vecs + (size_t)k * 32, *__restrict__ wT = v0 + 16;
    Tile1 Q, Qn;
    if (blockIdx.z == 0) {                                          // the two scans are independent: one wavefront each
    // forward: v at every chunk start (per-column: lane holds v[c]);  reg r of the dump = Q[4r + g][c]
    double2 v = v0[c];
    load_tile(Q, Qk, lane);
    for (int ch = 0; ch < C; ++ch) {
        load_tile(Qn, Qk + (size_t)min(ch + 1, C - 1) * TSZ, lane);
        if (g == 0)
            vec[(size_t)ch * 32 + c] = v;
        double y[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y[2 * r] = fma(Q.re[r], v.x, -Q.im[r] * v.y);
            y[2 * r + 1] = fma(Q.re[r], v.y, Q.im[r] * v.x);
        }
        row_sum_n(y);                                               // every lane of row g: (Q v)[4r + g]
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c == r)
                s_v[4 * r + g] = make_double2(y[2 * r], y[2 * r + 1]);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        v = s_v[c];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        Q = Qn;
    }
    // s = wT' v_N
    {
        double z[2] = {0.0, 0.0};
        if (g == 0) {
            const double2 w = wT[c];
            z[0] = w.x * v.x + w.y * v.y;
            z[1] = w.x * v.y - w.y * v.x;
        }
        row_sum_n(z);
        const double s_re = __shfl(z[0], 0, 64), s_im = __shfl(z[1], 0, 64);
        if (lane == 0) {
            p.tp_z[kw * 128] = s_re;
            p.tp_z[kw * 128 + 1] = s_im;
        }
    }
    return;
    }
    // backward: w at every chunk end;  (Q' w)[c] = sum_i conj(Q[i][c]) w[i]
    double2 w = wT[c];
    load_tile(Q, Qk + (size_t)(C - 1) * TSZ, lane);
    for (int ch = C - 1; ch >= 0; --ch) {
        load_tile(Qn, Qk + (size_t)max(ch - 1, 0) * TSZ, lane);
        if (g == 0) {
            vec[(size_t)ch * 32 + 16 + c] = w;
            s_v[c] = w;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        double acc[2] = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 wi = s_v[4 * r + g];
            acc[0] = fma(Q.re[r], wi.x, acc[0]);
            acc[0] = fma(Q.im[r], wi.y, acc[0]);
            acc[1] = fma(Q.re[r], wi.y, acc[1]);
            acc[1] = fma(-Q.im[r], wi.x, acc[1]);
        }
        col_sum_n(acc);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        w = make_double2(acc[0], acc[1]);
        Q = Qn;
    }
}

template <int SAND, bool HERMB>
static void launch_thin(const TileParams &q, size_t lds, hipStream_t stream)
{
    const dim3 grid(q.E, q.n_x, q.tp_chunks ? q.tp_chunks : 1);
    if (q.bt_in_lds)
        GRAPE_LAUNCH((chain_thin_kernel<SAND, HERMB, true>), grid, dim3(64), lds, stream, q);
    else
        GRAPE_LAUNCH((chain_thin_kernel<SAND, HERMB, false>), grid, dim3(64), 0, stream, q);
}

hipError_t launch_chain_thin(int sandwich, const TileParams &p, hipStream_t stream)
{
    TileParams q = p;
    if (q.tp_chunks > 1 && !q.fuse_fwd) {
        const dim3 cgrid(p.E, p.n_x, p.tp_chunks);
        GRAPE_LAUNCH(chunk_product_thin_kernel, cgrid, dim3(64), sizeof(double2) * (kTileImage + 1), stream, q);
        GRAPE_LAUNCH(chunk_scan_thin_kernel, dim3(p.E, p.n_x, 2), dim3(64), 0, stream, q);
    } else {
        q.tp_chunks = 0;
    }
    const size_t b_bytes = sizeof(double2) * 2 * (size_t)p.K * 256;
    q.bt_in_lds = b_bytes <= 36 * 1024 ? 1 : 0;                   // four workgroups per CU still fit
    const size_t lds = q.bt_in_lds ? b_bytes : 0;
    if (!sandwich)
        launch_thin<0, true>(q, lds, stream);
    else if (p.herm_ctrl)
        launch_thin<1, true>(q, lds, stream);
    else
        launch_thin<1, false>(q, lds, stream);
    return hipGetLastError();
}

}  // namespace grape
