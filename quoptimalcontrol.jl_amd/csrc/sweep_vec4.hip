// sweep_vec4.hip -- n = 4, states 4 x 1, left multiplication: vec(rho) of ONE qubit under a Liouvillian, the workload the
// reference writes out by hand in test/liou.jl:38-48.  gfx950.
//
// The lane-pair kernel (sweep_pair.hip) runs such a problem zero-padded to 4 x 4: with a dissipator the generator is not
// Hermitian, the general flow applies -- seven to eight 4 x 4 products per slice (Taylor-8, chunk product, X_t = Q X_s,
// L_t = P' L_t+1, X L') and P_t AND the in-chunk prefixes through HBM (1 KB per slice: 1.05 GB at E = 1024, N = 1000,
// 0.225-0.265 ms, bound by that traffic).  On vectors the chain itself is cheap -- v_t+1 = P_t v_t and w_t = P_t' w_t+1 are
// 16 complex FMAs each -- but SEQUENTIAL in time, and cutting the time axis needs the matrix products again (chunk products,
// matrix scans).  This kernel keeps the chain sequential and hides it instead:
//
//   workgroup = 4 members, 8 waves.  Wave 0 is the CHAIN wave: DPP row m (16 lanes) carries member m's vector, lane r of
//   the row its component r; a step is sixteen `v_fmac_f64 ... row_newbcast` (the one DPP control the FP64 ALU takes), the
//   matrix rows / columns come from LDS.  Waves 1..7 are WORKERS: 224 lane pairs, 56 per member.
//   forward rounds   workers: P_t = exp(G_t) of tile r (56 slices per member, one per lane pair; cmatp.hpp) -> LDS tile
//                    buffer r & 1, and tile r - 1 from LDS to the propagator array in HBM (coalesced copy);
//                    chain wave: tile r - 1 from LDS, v_t+1 = P_t v_t, records v_t to HBM (64 B per slice)
//   backward rounds  workers: tile j of P_t from HBM -> LDS (the tiles return in reverse order: the last ones written are
//                    the first ones read), and the gradient of the tile the chain finished LAST round -- thread = slice:
//                    a_c = w_t' B_c v_t, g[c,t] = -/+ 2 dt Im(a_c z);  chain wave: w_t = P_t' w_t+1 -> LDS records
//   one __syncthreads() per round, no other synchronisation.  z = v_N' x_t (= tr(X_t' L_t) at every t).
// Per slice: the Taylor-8 products (3 + s), two matrix-vector products, K forms; HBM: P_t written once and read once, the v
// records written once and read once (576 B per slice).  Same mathematics as the reference's general flow
// (src/GRAPE.jl:53-92, :226-228) to rounding; tests hold it to the 1e-10 bar against the oracle and to 1e-12 against the
// padded run of sweep_pair.hip.
//
// Interface: AnyParams (the size-generic family's: operators per member [A | B_1..B_K | Xi | Xt] 4 x 4 column-major as
// uploaded, props = N matrices per (control array, member), states = N matrices per (control array, member) of which this
// kernel writes column 0 (the buffer is zeroed once by the host: X_t = [v_t 0 0 0]), member_out rows).
#include <cstdlib>

#include "cmatp.hpp"
#include "grape_kernels.hpp"

#ifndef GRAPE_V4_ABL
#define GRAPE_V4_ABL 0       // diagnostic ablations (tools/variant.sh; wrong results): 1 no chain steps, 2 no Taylor series, 4 no sweep back,
#endif                       // 8 no reduce-side stores of the gradient

namespace grape {

namespace {

constexpr int kV4Threads = 512;                        // 8 waves
constexpr int kV4Members = 4;                          // members per workgroup = DPP rows of the chain wave
constexpr int kV4Tile = 56;                            // slices per member and round = lane pairs per member (7 waves x 32 / 4)
constexpr int kV4MaxK = 8;

// global element i + 4 j of local element (r, jl) of a lane of parity q (cmatp.hpp's "own block first" row order)
GRAPE_DEV int v4_gelem(int q, int r, int jl)
{
    const int i = (((r >> 1) ^ q) << 1) + (r & 1);
    const int j = 2 * q + jl;
    return i + 4 * j;
}

// y += M x over four columns: x_J broadcast from lane J of the DPP row (see action_thin.hip: act_matvec), four
// independent accumulators; y_re = a0 + a1, y_im = b0 + b1
#define GRAPE_V4_MAC(J, MR, MI)                                                       \
    "v_fmac_f64_dpp %0, %4, %" #MR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, %5, %" #MR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%5, %" #MI " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, %4, %" #MI " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"
GRAPE_DEV void v4_matvec(double &yr, double &yi, double xr, double xi, const double (&mr)[4], const double (&mi)[4])
{
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    // (s_nop 1: the two wait states a DPP read needs behind the VALU write of xr / xi -- the hazard recogniser does not look
    // inside inline asm)
    asm("s_nop 1\n" GRAPE_V4_MAC(0, 6, 10) GRAPE_V4_MAC(1, 7, 11) GRAPE_V4_MAC(2, 8, 12) GRAPE_V4_MAC(3, 9, 13)
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
        : "v"(xr), "v"(xi), "v"(mr[0]), "v"(mr[1]), "v"(mr[2]), "v"(mr[3]), "v"(mi[0]), "v"(mi[1]), "v"(mi[2]), "v"(mi[3]));
    yr = a0 + a1;
    yi = b0 + b1;
}
#undef GRAPE_V4_MAC

constexpr int kV4TileStride = kV4Tile * 16 + 4;        // double2 per member tile: the four members' tiles start 64 B apart in
                                                       // the bank pattern (the chain wave reads all four with one instruction)
struct V4Lds {                                         // offsets in double2 units into the dynamic LDS block
    int ops, raw, ptile, wrec, zed, total;
};
__host__ __device__ inline V4Lds v4_lds(int K)
{
    V4Lds l;
    l.ops = 0;                                         // (-i dt) [A | B_1..B_K] per member: (1 + K) x 16
    l.raw = l.ops + kV4Members * (1 + K) * 16;         // B_1..B_K as uploaded (the forms): K x 16 per member
    l.ptile = l.raw + kV4Members * K * 16;             // two buffers of 4 x 56 propagators
    l.wrec = l.ptile + 2 * kV4Members * kV4TileStride; // two buffers of 4 x 56 costate records (4 entries each)
    l.zed = l.wrec + 2 * kV4Members * kV4Tile * 4;     // per member: four partial products of z, then z
    l.total = l.zed + kV4Members * 8;
    return l;
}

__global__ __launch_bounds__(kV4Threads) void vec4_sweep_kernel(const AnyParams p)
{
    extern __shared__ double2 s_v4[];
    const int K = p.K, N = p.N;
    const V4Lds L = v4_lds(K);
    double2 *s_ops = s_v4 + L.ops, *s_raw = s_v4 + L.raw, *s_P = s_v4 + L.ptile, *s_W = s_v4 + L.wrec, *s_z = s_v4 + L.zed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, k0 = blockIdx.x * kV4Members;
    const double dt = p.dt;
    const double *__restrict__ x = p.x + (size_t)z * K * N;
    const int R = (N + kV4Tile - 1) / kV4Tile;         // tiles
    auto member_ops = [&](int m) { return p.ops + (size_t)min(k0 + m, p.E - 1) * (K + 3) * 16; };
    auto kw_of = [&](int m) { return (size_t)z * p.E + (size_t)min(k0 + m, p.E - 1); };
    auto tile_of = [&](int buf, int m) { return s_P + ((size_t)buf * kV4Members + m) * kV4TileStride; };

    // operator images: (-i dt) [A | B_c] for the propagators, B_c as uploaded for the forms
    for (int i = tid; i < kV4Members * (1 + K) * 16; i += kV4Threads) {
        const int m = i / ((1 + K) * 16), rem = i % ((1 + K) * 16);
        const double2 v = member_ops(m)[rem];
        s_ops[i] = make_double2(dt * v.y, -dt * v.x);
    }
    for (int i = tid; i < kV4Members * K * 16; i += kV4Threads) {
        const int m = i / (K * 16), rem = i % (K * 16);
        s_raw[i] = member_ops(m)[16 + rem];
    }
    __syncthreads();

    // chain wave: DPP row m <-> member m, lane r (0..3) of the row <-> component r (lanes 4..15 mirror lane r & 3: they
    // compute, and where something is stored store, the same values to the same places)
    const int cm = lane >> 4, cr = lane & 3;
    const bool chain_valid = k0 + cm < p.E;
    // workers: pair wp = member wm, slot wq of the tile; the copies walk a member's tile 16 bytes per thread, CP steps each
    const int wp = (wave - 1) * 32 + (lane >> 1), par = lane & 1;
    const int wm = wp / kV4Tile, wq = wp % kV4Tile;
    const int wtid = tid - 64;                         // 0 .. 447 among the workers
    constexpr int WT = kV4Threads - 64, CP = kV4Members * kV4Tile * 16 / WT;         // 3584 / 448 = 8
    static_assert(CP * WT == kV4Members * kV4Tile * 16, "the tile copy is CP whole steps");
    // element i = wtid + u WT of a tile: member i / 896, offset i % 896 inside the member's 56 x 16 entries
    auto copy_out = [&](int T) {                       // tile T: LDS -> propagator array (16 bytes per thread and step, contiguous)
        const int t0 = T * kV4Tile, len = min(kV4Tile, N - t0);
        double2 v[CP];
#pragma unroll
        for (int u = 0; u < CP; ++u) {
            const int i = wtid + u * WT, m = i / (kV4Tile * 16), rem = i % (kV4Tile * 16);
            v[u] = tile_of(T & 1, m)[rem];
        }
#pragma unroll
        for (int u = 0; u < CP; ++u) {
            const int i = wtid + u * WT, m = i / (kV4Tile * 16), rem = i % (kV4Tile * 16);
            if (rem < len * 16 && k0 + m < p.E)
                p.props[(kw_of(m) * N + t0) * 16 + rem] = v[u];
        }
    };

    double vr = 0.0, vi = 0.0;                         // the chain's vector component
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);                 // the chain is the critical path: it wins the issue slot it shares
        const double2 xi0 = member_ops(cm)[(1 + K) * 16 + cr];       // Xi[:, 0]
        vr = xi0.x;
        vi = xi0.y;
        if (chain_valid)                               // X_0
            p.states[(kw_of(cm) * N + 0) * 16 + cr] = xi0;
    }
    // ------------------------------------------------------------ forward rounds, src/GRAPE.jl:53-63
    for (int r = 0; r <= R; ++r) {
        if (wave == 0) {
            if (r >= 1) {
                const int T = r - 1, t0 = T * kV4Tile, len = min(kV4Tile, N - t0);
                const double2 *tile = tile_of(T & 1, cm) + cr;
                double2 *xs = p.states + (kw_of(cm) * N + t0 + 1) * 16 + cr;
                const int nstore = chain_valid ? min(len, N - 1 - t0) : 0;          // X_t+1 = P_t X_t; X_N is not needed
                // row cr of P_t: elements cr + 4 c.  The rows of the next two steps are in flight while a step runs (an LDS read
                // takes longer than the sixteen FMAs of a step: unprefetched, the chain -- not the propagators -- set the pace)
                auto load_row = [&](int q, double (&mr)[4], double (&mi)[4]) {
                    const double2 *e = tile + min(q, len - 1) * 16;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double2 v = e[4 * c];
                        mr[c] = v.x;
                        mi[c] = v.y;
                    }
                };
                auto step = [&](int q, const double (&mr)[4], const double (&mi)[4]) {
                    if (q >= len || (GRAPE_V4_ABL & 1))
                        return;
                    double yr, yi;
                    v4_matvec(yr, yi, vr, vi, mr, mi);
                    vr = yr;
                    vi = yi;
                    if (q < nstore)
                        xs[q * 16] = make_double2(vr, vi);
                };
                double ar_[4], ai_[4], br_[4], bi_[4], cr_[4], ci_[4];
                load_row(0, ar_, ai_);
                load_row(1, br_, bi_);
                for (int q = 0; q < len; q += 3) {
                    load_row(q + 2, cr_, ci_);
                    step(q, ar_, ai_);
                    load_row(q + 3, ar_, ai_);
                    step(q + 1, br_, bi_);
                    load_row(q + 4, br_, bi_);
                    step(q + 2, cr_, ci_);
                }
            }
        } else {
            if (r >= 1)                                // tile r - 1 leaves for HBM first: its stores drain under the Taylor series
                copy_out(r - 1);
            if (r < R) {                               // P_t = exp(G_t), slice t = 56 r + wq of member wm
                const int t = r * kV4Tile + wq;
                if (t < N) {
                    const double2 *sA = s_ops + (size_t)wm * (1 + K) * 16, *sB = sA + 16;
                    PMat<4> G, P;
                    int ge[8];
#pragma unroll
                    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            ge[rr + jl * 4] = v4_gelem(par, rr, jl);
                    if (p.variant != 0) {              // A + B_1 x_1 + ...
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const double2 a = sA[ge[e]];
                            G.re[e] = a.x;
                            G.im[e] = a.y;
                        }
                    }
                    for (int c = 0; c < K; ++c) {
                        const double xc = x[c + (size_t)t * K];
                        if (c == 0 && p.variant == 0) {          // (0 + B_1 x_1 + ...) + A, src/timeevolution.jl:101-108
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const double2 b = sB[ge[e]];
                                G.re[e] = b.x * xc;
                                G.im[e] = b.y * xc;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const double2 b = sB[c * 16 + ge[e]];
                                G.re[e] = fma(b.x, xc, G.re[e]);
                                G.im[e] = fma(b.y, xc, G.im[e]);
                            }
                        }
                    }
                    if (p.variant == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const double2 a = sA[ge[e]];
                            G.re[e] += a.x;
                            G.im[e] += a.y;
                        }
                    }
                    if (GRAPE_V4_ABL & 2)
                        P = G;
                    else
                        pexpm_t8<4, false>(P, G, p.s_forced);
                    double2 *dst = tile_of(r & 1, wm) + wq * 16;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        dst[ge[e]] = make_double2(P.re[e], P.im[e]);
                }
            }
        }
        __syncthreads();
    }
    // z = X_N' Xt = tr(X_t' L_t) at every t (the trace is cyclic): four partial products per member, summed in lane order
    if (wave == 0 && (lane & 15) < 4) {
        const double2 xt = member_ops(cm)[(2 + K) * 16 + cr];
        s_z[cm * 8 + cr] = make_double2(vr * xt.x + vi * xt.y, vr * xt.y - vi * xt.x);      // conj(v) xt
    }
    __syncthreads();
    if (tid < kV4Members) {
        double2 acc = s_z[tid * 8];
        for (int c = 1; c < 4; ++c) {
            acc.x += s_z[tid * 8 + c].x;
            acc.y += s_z[tid * 8 + c].y;
        }
        s_z[tid * 8 + 4] = acc;
        if (k0 + tid < p.E) {                          // figure of merit, src/GRAPE.jl:99-101: Re(z^2)
            double *out = p.member_out + ((size_t)z * p.E_rows + k0 + tid) * ((size_t)K * N + 1);
            out[(size_t)K * N] = acc.x * acc.x - acc.y * acc.y;
        }
    }
    __syncthreads();
    // ------------------------------------------------------------ backward rounds + gradient, :65-92
    const double gs = p.variant == 0 ? -2.0 * dt : 2.0 * dt;
    double wr_ = 0.0, wi_ = 0.0;
    if (wave == 0) {
        const double2 xt = member_ops(cm)[(2 + K) * 16 + cr];        // L_N = Xt[:, 0]
        wr_ = xt.x;
        wi_ = xt.y;
    }
    // The tiles return in reverse order.  A worker keeps the tile that is on its way in registers for a whole round: round j
    // puts tile R - 1 - j (requested in round j - 1) into LDS buffer j & 1, requests tile R - 2 - j, and forms the gradient
    // of tile R + 1 - j (walked by the chain in round j - 1); the chain walks tile R - j.  No round waits for memory.
    double2 fly0, fly1, fly2, fly3, fly4, fly5, fly6, fly7;      // (named registers: as an array they ended up in scratch memory)
    static_assert(CP == 8, "eight entries in flight per worker");
#define GRAPE_V4_REQUEST(T_)                                                                                  \
    do {                                                                                                      \
        const int Tq = max((T_), 0), t0q = Tq * kV4Tile, lenq = min(kV4Tile, N - t0q);                          \
        auto at = [&](int u) {                                                                                \
            const int i = wtid + u * WT, m = i / (kV4Tile * 16), rem = i % (kV4Tile * 16);                    \
            return p.props[(kw_of(m) * N + t0q) * 16 + min(rem, lenq * 16 - 1)];                              \
        };                                                                                                    \
        fly0 = at(0); fly1 = at(1); fly2 = at(2); fly3 = at(3);                                               \
        fly4 = at(4); fly5 = at(5); fly6 = at(6); fly7 = at(7);                                               \
    } while (0)
    if (wave != 0 && !(GRAPE_V4_ABL & 4))
        GRAPE_V4_REQUEST(R - 1);
    for (int j = 0; j <= R + 1 && !(GRAPE_V4_ABL & 4); ++j) {
        if (wave == 0) {
            if (j >= 1 && j <= R) {
                const int T = R - j, t0 = T * kV4Tile, len = min(kV4Tile, N - t0), b = (j - 1) & 1;
                const double2 *tile = tile_of(b, cm) + 4 * cr;
                double2 *rec = s_W + ((size_t)b * kV4Members + cm) * kV4Tile * 4 + cr;
                // column cr of P_t, conjugated: (P')[cr][c] = conj(P[c][cr]); step u walks slice len - 1 - u
                auto load_col = [&](int u, double (&mr)[4], double (&mi)[4]) {
                    const double2 *e = tile + (len - 1 - min(u, len - 1)) * 16;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double2 v = e[c];
                        mr[c] = v.x;
                        mi[c] = -v.y;
                    }
                };
                auto step = [&](int u, const double (&mr)[4], const double (&mi)[4]) {
                    if (u >= len || (GRAPE_V4_ABL & 1))
                        return;
                    double yr, yi;
                    v4_matvec(yr, yi, wr_, wi_, mr, mi);
                    wr_ = yr;
                    wi_ = yi;
                    rec[(len - 1 - u) * 4] = make_double2(wr_, wi_);     // L_t
                };
                double ar_[4], ai_[4], br_[4], bi_[4], cr_[4], ci_[4];
                load_col(0, ar_, ai_);
                load_col(1, br_, bi_);
                for (int u = 0; u < len; u += 3) {
                    load_col(u + 2, cr_, ci_);
                    step(u, ar_, ai_);
                    load_col(u + 3, ar_, ai_);
                    step(u + 1, br_, bi_);
                    load_col(u + 4, br_, bi_);
                    step(u + 2, cr_, ci_);
                }
            }
        } else {
            if (j < R) {                               // tile R - 1 - j has landed: registers -> LDS buffer j & 1; the next one leaves
                auto put = [&](int u, double2 v) {
                    const int i = wtid + u * WT, m = i / (kV4Tile * 16), rem = i % (kV4Tile * 16);
                    tile_of(j & 1, m)[rem] = v;
                };
                put(0, fly0); put(1, fly1); put(2, fly2); put(3, fly3);
                put(4, fly4); put(5, fly5); put(6, fly6); put(7, fly7);
                if (R - 2 - j >= 0)
                    GRAPE_V4_REQUEST(R - 2 - j);
            }
            if (j >= 2) {                              // the gradient of tile R + 1 - j, thread = (member, slice)
                const int T = R + 1 - j, t0 = T * kV4Tile, len = min(kV4Tile, N - t0), b = (j - 2) & 1;
                if (wtid < kV4Members * kV4Tile) {
                    const int m = wtid / kV4Tile, q = wtid % kV4Tile;
                    if (q < len && k0 + m < p.E) {
                        const int t = t0 + q;
                        const double2 *rec = s_W + (((size_t)b * kV4Members + m) * kV4Tile + q) * 4;
                        const double2 *xs = p.states + (kw_of(m) * N + t) * 16;
                        double2 l[4], xv[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            l[c] = rec[c];
                            xv[c] = xs[c];
                        }
                        const double2 zz = s_z[m * 8 + 4];
                        double *out = p.member_out + ((size_t)z * p.E_rows + k0 + m) * ((size_t)K * N + 1);
                        for (int c = 0; c < K; ++c) {
                            const double2 *Bc = s_raw + ((size_t)m * K + c) * 16;
                            double ar = 0.0, ai = 0.0;           // a = sum_ij conj(l_i) B[i][j] x_j
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                double ur = 0.0, ui = 0.0;       // u = sum_i conj(l_i) B[i][jj]
#pragma unroll
                                for (int ii = 0; ii < 4; ++ii) {
                                    const double2 bv = Bc[ii + 4 * jj];
                                    ur = fma(l[ii].x, bv.x, ur);
                                    ur = fma(l[ii].y, bv.y, ur);
                                    ui = fma(l[ii].x, bv.y, ui);
                                    ui = fma(-l[ii].y, bv.x, ui);
                                }
                                ar = fma(ur, xv[jj].x, ar);
                                ar = fma(-ui, xv[jj].y, ar);
                                ai = fma(ur, xv[jj].y, ai);
                                ai = fma(ui, xv[jj].x, ai);
                            }
                            out[c + (size_t)t * K] = gs * fma(ar, zz.y, ai * zz.x);          // Im(a z)
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

#undef GRAPE_V4_REQUEST

}  // namespace

size_t sweep_vec4_lds_bytes(int K) { return sizeof(double2) * (size_t)v4_lds(K).total; }

bool sweep_vec4_serves(int n, int K) { return n == 4 && K >= 1 && K <= kV4MaxK; }

hipError_t launch_sweep_vec4(const AnyParams &p, hipStream_t stream)
{
    const size_t lds = sweep_vec4_lds_bytes(p.K);
    const hipError_t e = ensure_dynamic_lds((const void *)vec4_sweep_kernel, lds);
    if (e != hipSuccess)
        return e;
    const dim3 grid((p.E + kV4Members - 1) / kV4Members, p.n_x);
    GRAPE_LAUNCH_AS("vec4_sweep_kernel", vec4_sweep_kernel, grid, dim3(kV4Threads), lds, stream, p);
    return hipGetLastError();
}

}  // namespace grape
