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

constexpr int kRing = 4;          // slices of P in flight per wave (4 KB each)

// sum over the 16 lanes of a DPP row (lanes sharing l >> 4), result in every lane: rotations by 8, 4, 2, 1
template <int M>
GRAPE_DEV void row_sum_n(double (&v)[M])
{
#define GRAPE_ROR_STEP(CTRL)                                                                                   \
    {                                                                                                          \
        double o[M];                                                                                           \
        _Pragma("unroll") for (int m = 0; m < M; ++m)                                                          \
        {                                                                                                      \
            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[m]), CTRL, 0xF, 0xF, true);         \
            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[m]), CTRL, 0xF, 0xF, true);         \
            o[m] = __hiloint2double(hi, lo);                                                                   \
        }                                                                                                      \
        _Pragma("unroll") for (int m = 0; m < M; ++m) v[m] += o[m];                                            \
    }
    GRAPE_ROR_STEP(0x128)          // row_ror:8
    GRAPE_ROR_STEP(0x124)          // row_ror:4
    GRAPE_ROR_STEP(0x122)          // row_ror:2
    GRAPE_ROR_STEP(0x121)          // row_ror:1
#undef GRAPE_ROR_STEP
}

// sum over the 4 rows (lanes sharing l & 15), result in every lane
template <int M>
GRAPE_DEV void col_sum_n(double (&v)[M])
{
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1) {
        double o[M];
#pragma unroll
        for (int m = 0; m < M; ++m)
            o[m] = __shfl_xor(v[m], d, 64);
#pragma unroll
        for (int m = 0; m < M; ++m)
            v[m] += o[m];
    }
}

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
template <int SAND, bool HERMB>
__global__ __launch_bounds__(64) void chain_thin_kernel(const TileParams p)
{
    constexpr int TSZ = 256;
    extern __shared__ double2 s_thin[];           // (bt_in_lds) this member's [B_c | B_c^T] dumps
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int k = blockIdx.x;
    const int K = p.K, N = p.N;
    const double2 *__restrict__ ops = p.ops + (size_t)k * (2 * K + 3) * TSZ;
    const double2 *__restrict__ opB = ops + TSZ;                 // B_c at opB + c TSZ, B_c^T at opB + (K + c) TSZ
    const bool in_lds = p.bt_in_lds != 0;
    if (in_lds) {
        for (int i = lane; i < 2 * K * TSZ; i += 64)
            s_thin[i] = opB[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    // one operator dump: LDS and global memory keep their own address spaces (a generic pointer would turn these
    // into flat loads, whose completion also waits for the propagator prefetches in flight)
    auto load_op = [&](Tile1 &m, int idx) {
        if (in_lds)
            load_tile(m, s_thin + (size_t)idx * TSZ, lane);
        else
            load_tile(m, opB + (size_t)idx * TSZ, lane);
    };
    const size_t kw = (size_t)blockIdx.y * p.E + k;
    const double2 *__restrict__ Pk = p.props + kw * N * TSZ;
    double2 *__restrict__ V = p.states + kw * (size_t)(N + 1) * 16;       // records v_0 .. v_N, 16 complex each
    const double2 *__restrict__ v0 = p.vecs + (size_t)k * 32, *__restrict__ wT = v0 + 16;
    double *__restrict__ out = p.member_out + ((size_t)blockIdx.y * p.E_members + k) * ((size_t)K * N + 1);

    // ------------------------------------------------------------------ forward: v_{t+1} = P_t v_t
    {
        Tile1 Pq[kRing];
#pragma unroll
        for (int u = 0; u < kRing; ++u)
            if (u < N)
                load_tile(Pq[u], Pk + (size_t)u * TSZ, lane);
        double vr, vi;                            // per-column format
        {
            const double2 t = v0[c];
            vr = t.x;
            vi = t.y;
        }
        double y[8];                              // gathered format: re/im of x[4r + g]
        for (int base = 0; base < N; base += kRing) {
#pragma unroll
            for (int u = 0; u < kRing; ++u) {
                const int t = base + u;
                if (t < N) {
                    const Tile1 P = Pq[u];
                    if (t + kRing < N)
                        load_tile(Pq[u], Pk + (size_t)(t + kRing) * TSZ, lane);
                    if ((u & 1) == 0) {           // even t: D layout of P, reg r = P[4r + g][c]
                        if (g == 0)
                            V[(size_t)t * 16 + c] = make_double2(vr, vi);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            y[2 * r] = fma(P.re[r], vr, -P.im[r] * vi);
                            y[2 * r + 1] = fma(P.re[r], vi, P.im[r] * vr);
                        }
                        row_sum_n(y);
                    } else {                      // odd t: D layout of P^T, reg r = P[c][4r + g]
                        if (c < 4) {
                            const double sr = c == 0 ? y[0] : (c == 1 ? y[2] : (c == 2 ? y[4] : y[6]));
                            const double si = c == 0 ? y[1] : (c == 1 ? y[3] : (c == 2 ? y[5] : y[7]));
                            V[(size_t)t * 16 + 4 * c + g] = make_double2(sr, si);
                        }
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
                }
            }
        }
        // v_N: after an odd last slice it is in per-column format, after an even one gathered
        if ((N & 1) == 0) {
            if (g == 0)
                V[(size_t)N * 16 + c] = make_double2(vr, vi);
        } else if (c < 4) {
            const double sr = c == 0 ? y[0] : (c == 1 ? y[2] : (c == 2 ? y[4] : y[6]));
            const double si = c == 0 ? y[1] : (c == 1 ? y[3] : (c == 2 ? y[5] : y[7]));
            V[(size_t)N * 16 + 4 * c + g] = make_double2(sr, si);
        }
    }
    // the records were written by other lanes of this wave than the ones that read them back
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // stores drained, vector L1 invalidated

    // ------------------------------------------------------------------ s = wT' v_N
    double s_re, s_im;
    {
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
        Tile1 Pq[kRing];
        double vq[kRing][8];                      // v_t records in the format slice t needs (even: gathered, odd: per-column)
        const int top = (N - 1) & ~(kRing - 1);
        auto load_v = [&](int u, int t) {
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
#pragma unroll
        for (int u = 0; u < kRing; ++u) {         // the last kRing slices: t = N-1 .. N-kRing, slot t & 3
            const int t = N - 1 - u;
            if (t >= 0) {
                // slot of slice t is t & (kRing - 1): static only inside the unrolled sub-step loops below, so fill
                // the ring through a switch on the (uniform) slot
                const int slot = t & (kRing - 1);
#pragma unroll
                for (int q = 0; q < kRing; ++q)
                    if (q == slot) {
                        load_tile(Pq[q], Pk + (size_t)t * TSZ, lane);
                        load_v(q, t);
                    }
            }
        }
        double wr = 0.0, wi = 0.0;                // per-column format
        double w4[8];                             // gathered format
        if ((N - 1) & 1) {                        // last slice odd: its input format is per-column
            const double2 t2 = wT[c];
            wr = t2.x;
            wi = t2.y;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 t2 = wT[4 * r + g];
                w4[2 * r] = t2.x;
                w4[2 * r + 1] = t2.y;
            }
        }
        for (int base = top; base >= 0; base -= kRing) {
#pragma unroll
            for (int u = kRing - 1; u >= 0; --u) {
                const int t = base + u;
                if (t < N) {
                    const Tile1 P = Pq[u];
                    double vt[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        vt[e] = vq[u][e];
                    if (t - kRing >= 0) {
                        load_tile(Pq[u], Pk + (size_t)(t - kRing) * TSZ, lane);
                        load_v(u, t - kRing);
                    }
                    double q4[4] = {0.0, 0.0, 0.0, 0.0};
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
                        for (int cc = 0; cc < K; ++cc) {
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
                            if (cc < 4) q4[cc] = val; else { /* more than four controls: reduced one by one below */
                                double one[1] = {val};
                                row_sum_n(one);
                                col_sum_n(one);
                                if (lane == 0) out[cc + (size_t)t * K] = gs * one[0];
                            }
                        }
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
                        for (int cc = 0; cc < K; ++cc) {
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
                            if (cc < 4) q4[cc] = val; else {
                                double one[1] = {val};
                                row_sum_n(one);
                                col_sum_n(one);
                                if (lane == 0) out[cc + (size_t)t * K] = gs * one[0];
                            }
                        }
                    }
                    row_sum_n(q4);
                    col_sum_n(q4);
                    if (lane == 0) {
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)
                            if (cc < K)
                                out[cc + (size_t)t * K] = gs * q4[cc];
                    }
                }
            }
        }
    }
    if (lane == 0) {
        if (SAND) {
            const double z = (s_re * s_re + s_im * s_im) / (double)p.n;   // tr(L' X) / D = |s|^2 / n
            out[(size_t)K * N] = 1.0 - z * z;
        } else {
            out[(size_t)K * N] = s_re * s_re - s_im * s_im;               // Re(z^2), z = conj(s)
        }
    }
}

hipError_t launch_chain_thin(int sandwich, const TileParams &p, hipStream_t stream)
{
    TileParams q = p;
    const size_t b_bytes = sizeof(double2) * 2 * (size_t)p.K * 256;
    q.bt_in_lds = b_bytes <= 36 * 1024 ? 1 : 0;                   // four waves per CU still fit
    const size_t lds = q.bt_in_lds ? b_bytes : 0;
    const dim3 grid(p.E, p.n_x), block(64);
    if (!sandwich)
        hipLaunchKernelGGL((chain_thin_kernel<0, true>), grid, block, lds, stream, q);
    else if (p.herm_ctrl)
        hipLaunchKernelGGL((chain_thin_kernel<1, true>), grid, block, lds, stream, q);
    else
        hipLaunchKernelGGL((chain_thin_kernel<1, false>), grid, block, lds, stream, q);
    return hipGetLastError();
}

}  // namespace grape
