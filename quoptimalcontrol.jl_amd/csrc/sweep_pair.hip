// sweep_pair.hip -- the GRAPE hot path for even small operators (n = 2, 4), lane-PAIR edition.
//
// Same algorithm, phases and data flows as sweep_small.hip (read its header first: phase A propagators
// + chunk product, phase B lane-chunk scan, phase D backward sweep fused with the gradient traces and the
// figure of merit; MODE_GENERAL / MODE_GENERAL_KEEPL / MODE_UNITARY), with ONE change of decomposition:
// a time chunk belongs to a PAIR of adjacent lanes, each owning half the columns of every matrix
// (cmatp.hpp).  A 4 x 4 complex matrix is 32 VGPRs per lane instead of 64, the kernel needs < 256
// registers instead of ~400 (256 VGPR + 150 AGPR of spill), so TWO waves share every SIMD: the FP64 pipe
// goes from the ~60 % issue rate of a lone wave to ~80 %, LDS / HBM latency hides behind the other wave,
// and the scan's products are shared by the pair.  At the headline config (E = 1024 members on 1024
// SIMDs) a member is W = 2 waves = 64 chunks of 8 slices.
//
// Operators: the two lanes of a pair need DIFFERENT entries of the member's (wave-uniform) operators, so
// the scalar-load trick of sweep_small.hip does not apply; instead the workgroup stages, per member, a
// parity-relative image of [A' | B'_c | B'_c^T | Xi | Xt] in LDS once (2 x (2K+3) x n*n/2 double2 = 2.8 KB
// at n = 4, K = 4) and every lane reads its entries with ds_read_b128 (two distinct addresses per wave
// instruction: broadcast, conflict free).
//
// Workspace layout (chunk-major): element e = i + j n of slice t = c S + jj of member k at
//   ((k S + jj) n^2 + e) CH + c ,   CH = chunks per member = 32 W ;
// a lane stores its own columns (two 512-byte runs per wave instruction) and the backward sweep reads
// the whole P_t per lane (both lanes of a pair read the same 16 bytes: one 512-byte run).
#include "cmatp.hpp"
#include "grape_kernels.hpp"

#ifndef GRAPE_ABL
#define GRAPE_ABL 0          // diagnostic ablation bitmask (tools/ablate.sh); 0 in the product build
#endif
#ifndef GRAPE_PD
#define GRAPE_PD 2           // unitary backward sweep: 0 = load one slice ahead in front of each step; 1 = products / gradient
#endif                       // split, the slice after next loaded as soon as the products have released its buffer
#ifndef GRAPE_PLAYOUT
#define GRAPE_PLAYOUT 0
#endif
#ifndef GRAPE_PD_EARLY
#define GRAPE_PD_EARLY 2     // GRAPE_PD == 2: slices requested before the scan (0, 1, 2)
#endif
#ifndef GRAPE_PD_EARLY_AT
#define GRAPE_PD_EARLY_AT 0  // where: 0 = straight after phase A, 1 = behind the scan's first barrier
#endif
#ifndef GRAPE_MBAR
#define GRAPE_MBAR 2         // scan barriers: 0 = __syncthreads() (all members of the workgroup), 1 = per-member LDS counters,
#endif                       // 2 = workgroup barrier that waits for LDS only (not for the P_t stores / prefetches in flight)

namespace grape {

enum { PMODE_GENERAL = 0, PMODE_GENERAL_KEEPL = 1, PMODE_UNITARY = 2 };
constexpr int kParityPad = 8;          // double2 slots (128 B = half an LDS row of 64 banks) between a member's two parity images

// global element index i + j n of local element (r, jl) as seen by a lane of parity q
template <int N>
GRAPE_DEV int gelem(int q, int r, int jl)
{
    constexpr int NC = N / 2;
    const int i = (((r / NC) ^ q) * NC) + (r % NC);
    const int j = q * NC + jl;
    return i + j * N;
}

// local half (parity-q view) <-> chunk-major workspace
template <int N>
GRAPE_DEV void pstore_ws(double2 *__restrict__ base, size_t stride, const PMat<N> &m, int q)
{
#pragma unroll
    for (int jl = 0; jl < N / 2; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r)
            base[(size_t)gelem<N>(q, r, jl) * stride] = make_double2(m.re[r + jl * N], m.im[r + jl * N]);
}

template <int N>
GRAPE_DEV void pload_ws(PMat<N> &m, const double2 *__restrict__ base, size_t stride, int q)
{
#pragma unroll
    for (int jl = 0; jl < N / 2; ++jl)
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const double2 v = base[(size_t)gelem<N>(q, r, jl) * stride];
            m.re[r + jl * N] = v.x;
            m.im[r + jl * N] = v.y;
        }
}

// parity image of one matrix in LDS: NE = n*n/2 consecutive double2
template <int N>
GRAPE_DEV void pload_lds(PMat<N> &m, const double2 *s)
{
#pragma unroll
    for (int e = 0; e < N * (N / 2); ++e) {
        const double2 v = s[e];
        m.re[e] = v.x;
        m.im[e] = v.y;
    }
}

template <int N>
GRAPE_DEV void pstore_lds(double2 *s, const PMat<N> &m)
{
#pragma unroll
    for (int e = 0; e < N * (N / 2); ++e)
        s[e] = make_double2(m.re[e], m.im[e]);
}

GRAPE_DEV void pstamp(unsigned long long *__restrict__ st, int slot)
{
    if (st) {
        const unsigned long long t = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0)
            st[slot] = t;
    }
}

// A propagator's own half as the eight 16-byte loads deliver it, loaded OUTSIDE the compiler's s_waitcnt bookkeeping (inline asm):
// the compiler drains every outstanding load in front of a loop's back edge, i.e. it waits for the prefetch it has just issued,
// and the backward sweep then exposes one full memory latency per slice (tools: build/abl pd0..pd3 of round 4).  Here the
// kernel counts itself: pbuf_issue() = 8 loads, pbuf_wait<N>() = "all but the N youngest have landed"; the wait takes the
// buffer as an in/out operand so that no read of it can move in front of the wait.
typedef double d2v __attribute__((ext_vector_type(2)));
struct PBuf {
    d2v v[8];          // element e = r + 4 jl of the lane's half (PMat<4> order)
};

// element (r, jl) of a lane of parity q sits (2 ((r >> 1) ^ q) + 8 q) + ((r & 1) + 4 jl) rows of CH*16 bytes into the slice:
// the first part is the lane's (voff[r >> 1], with the chunk's 16 ch bytes), the second is uniform (sb[(r & 1) + 4 jl])
GRAPE_DEV void pbuf_issue(PBuf &b, unsigned voff0, unsigned voff1, const double2 *sb0, const double2 *sb1, const double2 *sb4,
                          const double2 *sb5)
{
    asm volatile("global_load_dwordx4 %0, %8, %10\n\t"
                 "global_load_dwordx4 %1, %8, %11\n\t"
                 "global_load_dwordx4 %2, %9, %10\n\t"
                 "global_load_dwordx4 %3, %9, %11\n\t"
                 "global_load_dwordx4 %4, %8, %12\n\t"
                 "global_load_dwordx4 %5, %8, %13\n\t"
                 "global_load_dwordx4 %6, %9, %12\n\t"
                 "global_load_dwordx4 %7, %9, %13"
                 : "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]), "=&v"(b.v[3]), "=&v"(b.v[4]), "=&v"(b.v[5]), "=&v"(b.v[6]),
                   "=&v"(b.v[7])
                 : "v"(voff0), "v"(voff1), "s"(sb0), "s"(sb1), "s"(sb4), "s"(sb5)
                 : "memory");
}

template <int YOUNGER>
GRAPE_DEV void pbuf_wait(PBuf &b)
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]), "+v"(b.v[4]), "+v"(b.v[5]), "+v"(b.v[6]), "+v"(b.v[7])
                 : "n"(YOUNGER)
                 : "memory");
}

GRAPE_DEV void pbuf_to_mat(PMat<4> &m, const PBuf &b)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        m.re[e] = b.v[e][0];
        m.im[e] = b.v[e][1];
    }
}

// Barrier among the W waves of ONE member (the scan's exchanges never cross members): an LDS counter per member, every wave adds
// one and polls until `target` arrivals.  Unlike __syncthreads() it neither couples the members of a workgroup nor waits for
// this wave's global stores (the P_t of phase A are still draining when the scan starts).  LDS operations of a CU execute in
// order, so a wave that reads the count also sees what the arriving waves wrote before they added.
GRAPE_DEV void member_barrier(unsigned *cnt, unsigned target)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// workgroup barrier for exchanges through LDS only: the wave's own LDS operations drained, then s_barrier (__syncthreads()
// also waits for every global store / load in flight)
GRAPE_DEV void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int N, int SAND>
GRAPE_DEV double pfigure_of_merit(double zr, double zi)
{
    if (SAND) {                                  // 1 - |tr(L'X)/D|^2, cost_functions.jl:13-17
        const double inv = 1.0 / (double)N;
        const double ar = zr * inv, ai = zi * inv;
        return 1.0 - (ar * ar + ai * ai);
    }
    return zr * zr - zi * zi;                    // Re(z^2), cost_functions.jl:99-101
}

// gradient entries of one slice: Re tr(B'_c (z M)) for the K controls, own half + pair sum.
// s_bt: parity image of the transposed scaled controls, matrix c at s_bt + c*NE.
template <int N, int SAND>
GRAPE_DEV void pwrite_gradient(double *out, const double2 *s_bt, int K, const PMat<N> &M, double zr, double zi,
                               double gs, int p)
{
    constexpr int NE = N * (N / 2);
    PMat<N> Z;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (SAND) {
            Z.re[e] = M.re[e];
            Z.im[e] = M.im[e];
        } else {
            Z.re[e] = fma(M.re[e], zr, -M.im[e] * zi);
            Z.im[e] = fma(M.re[e], zi, M.im[e] * zr);
        }
    }
    // operand reads software-pipelined by hand: control c+1's entries are in flight during control c's FMAs
    double2 b0[NE], b1[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e)
        b0[e] = s_bt[e];
    for (int c = 0; c < K; c += 2) {
        if (c + 1 < K) {
#pragma unroll
            for (int e = 0; e < NE; ++e)
                b1[e] = s_bt[(c + 1) * NE + e];
        }
        double re = 0.0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            re = fma(b0[e].x, Z.re[e], re);
            re = fma(-b0[e].y, Z.im[e], re);
        }
        re += pair_swap(re);
        if (p == 0)
            out[c] = gs * re;
        if (c + 1 < K) {
            if (c + 2 < K) {
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    b0[e] = s_bt[(c + 2) * NE + e];
            }
            double r1 = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                r1 = fma(b1[e].x, Z.re[e], r1);
                r1 = fma(-b1[e].y, Z.im[e], r1);
            }
            r1 += pair_swap(r1);
            if (p == 0)
                out[c + 1] = gs * r1;
        }
    }
}

// VEC (general flow, left multiplication, n = 4, at most kVecSlices slices per lane): the states are n x 1 -- column 0 of the
// zero-padded matrices -- and the sweep back runs on VECTORS (phase D, below); phase A then stores no in-chunk prefixes.
constexpr int kVecSlices = 16;
template <int N, int SAND, int MODE, int MAXT, bool XGLDS, bool DUMPW = false, bool VEC = false>
__global__ __launch_bounds__(MAXT) void sweep_pair_kernel(const double2 *__restrict__ ops_all,
                                                          const double *__restrict__ x_all,
                                                          const double *__restrict__ wts_all,
                                                          const SweepParams p)
{
    constexpr int NN = N * N, NC = N / 2, NE = N * NC;
    constexpr int MAXW = MAXT / 64;
    constexpr bool UNI = (MODE == PMODE_UNITARY);
    constexpr bool KEEPL = (MODE == PMODE_GENERAL_KEEPL);
    // the backward sweep counts its own propagator loads (PBuf): only where they are the sweep's ONLY vector-memory operations
    // -- gradient entries staged in LDS (XGLDS), no W_t dump
    constexpr bool PDASM = (GRAPE_PD == 2) && UNI && N == 4 && !DUMPW && XGLDS;
    // dynamic LDS:  s_tot  2*MAXW*NN double2   wave totals of the two scans (both parity halves)
    //               s_ops  MPB * 2 * PS double2, PS = NM * NE + 8: parity images of the members' operators (the pad puts the
    //                      two parities' images half an LDS row apart: a wave reads BOTH addresses in one instruction)
    //               s_xg   MPB*CH*(S*K+1) double      controls in / gradient out, chunk stride odd
    //               s_F    MPB double
    extern __shared__ double2 s_dyn[];
    double2(*s_tot)[MAXW][NN] = reinterpret_cast<double2(*)[MAXW][NN]>(s_dyn);

    const int LT = p.LT, W = LT >> 6, CH = LT >> 1;
    const int mb = __builtin_amdgcn_readfirstlane(threadIdx.x / LT);   // member within the block
    const int L = threadIdx.x - mb * LT;             // lane within the member
    const int lane = L & 63, wave = L >> 6;
    const int par = L & 1;                           // parity within the pair
    const int ch = L >> 1;                           // time chunk of the pair
    const int cw = lane >> 1;                        // chunk within the wave (0..31)
    const int wbase_tot = mb * W;
    const int xi = blockIdx.x / p.BPX;               // which control array (batched evaluation)
    const int bi = blockIdx.x - xi * p.BPX;
    int kl = bi * p.MPB + mb;
    const bool valid = kl < p.E;
    if (!valid)
        kl = p.E - 1;
    const int k = xi * p.E + kl;
    const int K = p.K, Nsl = p.N, S = p.S;
    const size_t stride = (size_t)CH;
    const int NM = 2 * K + 4;                     // images: A', B'_c, B'_c^T, Xi, Xt, and Xi Xt' (built in the prologue)

    unsigned *s_cnt = reinterpret_cast<unsigned *>(s_dyn + 2 * MAXW * NN);      // per-member arrival counters (4 double2 = 16 words)
    double2 *s_ops_all = s_dyn + 2 * MAXW * NN + 4;
    const int PS = NM * NE + kParityPad;          // stride between the two parity images of a member
    double2 *s_ops = s_ops_all + (size_t)mb * 2 * PS;
    const int SK = S * K;
    const size_t nrm_d2 = ((size_t)p.MPB * (K + 1) + 1) / 2;        // double2 slots of the operator-norm table
    double *s_xg_all = XGLDS ? reinterpret_cast<double *>(s_ops_all + (size_t)p.MPB * 2 * PS + nrm_d2)
                             : p.xg_scratch + (size_t)blockIdx.x * ((size_t)p.MPB * CH * (SK + 1) + p.MPB);
    double *s_xg = s_xg_all + (size_t)mb * CH * (SK + 1);
    double *s_F = s_xg_all + (size_t)p.MPB * CH * (SK + 1);
    // (optional) the chunks' last propagators: element e of thread T at s_plast[e * blockDim.x + T]
    double2 *s_plast = reinterpret_cast<double2 *>(s_F + ((p.MPB + 1) & ~1));
    // 1-norm bounds (max column sum of |re| + |im|) of this member's A' and B'_c: |G_t|_1 <= nrm[0] + sum |x_c| nrm[1+c]
    double *s_nrm = reinterpret_cast<double *>(s_ops_all + (size_t)p.MPB * 2 * PS) + (size_t)mb * (K + 1);
    const unsigned magic = p.sk_magic;
    auto chunk_of = [&](int q) { return SK == 1 ? q : (int)__umulhi((unsigned)q, magic); };
    if (threadIdx.x < 16)
        s_cnt[threadIdx.x] = 0u;
    if (p.tune & 7) {                                 // diagnostic / tuning: which waves of a SIMD go first (s_setprio)
        const int sel = p.tune & 7;
        const bool hi = sel == 1 ? (mb & 1) : sel == 2 ? (mb & 2) : sel == 3 ? (blockIdx.x & 1) : sel == 4 ? ((threadIdx.x >> 6) & 1) : false;
        if (hi)
            __builtin_amdgcn_s_setprio(2);
    }
    {
        const double *__restrict__ xsrc = x_all + (size_t)xi * K * Nsl;
        const int KNs = K * Nsl;
        for (int q0 = 0; q0 < KNs; q0 += 8 * LT) {           // 8 loads in flight per lane, then the LDS scatter
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * LT + L;
                v[u] = q < KNs ? xsrc[q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * LT + L;
                if (q < KNs)
                    s_xg[q + chunk_of(q)] = v[u];
            }
        }
        // parity images of this member's operators: image element (q, mat, e = r + jl n)
        const double2 *__restrict__ ops = ops_all + (size_t)kl * (K + 3) * NN;
        for (int idx = L; idx < 2 * NM * NE; idx += LT) {
            const int q = idx / (NM * NE), rem = idx - q * NM * NE;
            const int mat = rem / NE, e = rem - mat * NE;
            if (mat == 2 * K + 3)
                continue;                            // Xi Xt': filled below
            const int r = e % N, jl = e / N;
            const int i = (((r / NC) ^ q) * NC) + (r % NC), j = q * NC + jl;
            int src;
            if (mat <= K)            src = mat * NN + i + j * N;                 // A', B'_c
            else if (mat <= 2 * K)   src = (mat - K) * NN + j + i * N;           // B'_c transposed
            else                     src = (mat - K) * NN + i + j * N;           // Xi, Xt
            s_ops[(size_t)q * PS + rem] = ops[src];
        }
    }
    if (L < 2 * NE) {                                // Xi Xt' (unitary UnitaryGate fix-up), one entry per lane, from the global operators
        const double2 *__restrict__ ops = ops_all + (size_t)kl * (K + 3) * NN;
        const double2 *__restrict__ gXi = ops + (size_t)(1 + K) * NN, *__restrict__ gXt = gXi + NN;
        const int q = L / NE, e = L - q * NE;
        const int r = e % N, jl = e / N;
        const int i = (((r / NC) ^ q) * NC) + (r % NC), j = q * NC + jl;
        double sr = 0.0, si = 0.0;
        for (int kk = 0; kk < N; ++kk) {
            const double2 a = gXi[i + kk * N], b = gXt[j + kk * N];      // Xi[i,k] conj(Xt[j,k])
            sr = fma(a.x, b.x, sr); sr = fma(a.y, b.y, sr);
            si = fma(a.y, b.x, si); si = fma(-a.x, b.y, si);
        }
        s_ops[(size_t)q * PS + (size_t)(2 * K + 3) * NE + e] = make_double2(sr, si);
    }
    __syncthreads();
    if (L <= K) {                                    // one lane per generator: max column sum of |re| + |im|
        double best = 0.0;
        for (int q = 0; q < 2; ++q)
            for (int jl = 0; jl < NC; ++jl) {
                double cs = 0.0;
                for (int r = 0; r < N; ++r) {
                    const double2 v = s_ops[(size_t)q * PS + (size_t)L * NE + r + jl * N];
                    cs += fabs(v.x) + fabs(v.y);
                }
                best = fmax(best, cs);
            }
        s_nrm[L] = best;
    }
    __syncthreads();
    const double2 *sA = s_ops + (size_t)par * PS;               // my parity's images
    const double2 *sB = sA + NE;
    const double2 *sBT = sA + (size_t)(1 + K) * NE;
    const double2 *sXi = sA + (size_t)(1 + 2 * K) * NE;
    const double2 *sXt = sXi + NE;
    const double2 *sXi_o = s_ops + (size_t)(1 - par) * PS + (size_t)(1 + 2 * K) * NE;   // the partner's images
    const double2 *sXt_o = sXi_o + NE;
    const double2 *sXX = sXt + NE, *sXX_o = sXt_o + NE;
    double *xg = s_xg + ch * (SK + 1);
    const size_t wbase = (size_t)k * S * NN * stride + ch;
    // GRAPE_PLAYOUT 1 (experiment): propagators slice-major over the launch's members -- all waves work on the same slice
    // index at the same time, and member-major puts their 16 KB blocks 16 S KB apart
    const size_t pstep = GRAPE_PLAYOUT ? (size_t)p.E * p.n_x * NN * stride : (size_t)NN * stride;
    double2 *__restrict__ Pw = p.props + (GRAPE_PLAYOUT ? (size_t)k * NN * stride + ch : wbase);
    double2 *__restrict__ Xw = p.states + wbase;
    double *__restrict__ out = p.member_out + (size_t)k * ((size_t)K * Nsl + 1);
    const int t0 = ch * S;
    unsigned long long *__restrict__ st =
        p.stamps ? p.stamps + ((size_t)k * W + wave) * kStampSlots : nullptr;
    if (st && lane == 0)
        st[5] = __builtin_amdgcn_s_memrealtime();
    pstamp(st, 0);

    // ---------------------------------------------------------------- phase A
    PMat<N> Q, Q2;
    // returns an upper bound of |G|_1 from the members' operator norms: |A'| + sum_c |x_c| |B'_c|
    auto build = [&](int j, PMat<N> &G) -> double {
        // operator images are fetched one control ahead with clamped (always valid) indices: no branches around
        // the LDS reads; an odd K adds one pass with x = 0
        double nb = s_nrm[0];
        double2 b0[NE], b1[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e)
            b0[e] = sB[e];
        if (p.variant != 0)
            pload_lds(G, sA);
        for (int c = 0; c < K; c += 2) {
            const int c1 = min(c + 1, K - 1), c2 = min(c + 2, K - 1);
            const double x0 = xg[j * K + c];
            const double x1 = (c + 1 < K) ? xg[j * K + c1] : 0.0;
            nb = fma(fabs(x0), s_nrm[1 + c], nb);
            nb = fma(fabs(x1), s_nrm[1 + c1], nb);
#pragma unroll
            for (int e = 0; e < NE; ++e)
                b1[e] = sB[c1 * NE + e];
            if (c == 0 && p.variant == 0) {          // (0 + B_1 x_1): the sum starts here, timeevolution.jl:101-108
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    G.re[e] = b0[e].x * x0;
                    G.im[e] = b0[e].y * x0;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    G.re[e] = fma(b0[e].x, x0, G.re[e]);
                    G.im[e] = fma(b0[e].y, x0, G.im[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < NE; ++e)
                b0[e] = sB[c2 * NE + e];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                G.re[e] = fma(b1[e].x, x1, G.re[e]);
                G.im[e] = fma(b1[e].y, x1, G.im[e]);
            }
        }
        if (p.variant == 0) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const double2 a = sA[e];
                G.re[e] += a.x;
                G.im[e] += a.y;
            }
        }
        return nb;
    };
    // one slice: P_t = exp(G), stored; chunk product Qout = P_t Qin (FIRST: the product starts from the identity)
    auto slice = [&](int j, PMat<N> &G, double nb, const PMat<N> &Qin, PMat<N> &Qout, bool first) {
        PMat<N> P, Ppar;
        __builtin_amdgcn_sched_barrier(0);            // stages stay apart: one long basic block otherwise costs ~50 spilled VGPRs
        if (GRAPE_ABL & 32) P = G; else
        pexpm_t8<N, UNI>(P, G, p.s_forced, nb);
        __builtin_amdgcn_sched_barrier(0);
        if (!(GRAPE_ABL & 2))
        pstore_ws(Pw + (size_t)j * pstep, stride, P, par);
        if (UNI && XGLDS && p.plast_lds && j == S - 1) {          // the backward sweep's first operand stays on chip
#pragma unroll
            for (int e = 0; e < NE; ++e)
                s_plast[e * blockDim.x + threadIdx.x] = make_double2(P.re[e], P.im[e]);
        }
        if (GRAPE_ABL & 4) { Qout = Qin; Qout.re[0] += P.re[1]; } else if (first) {
            Qout = P;
        } else {
            fetch_partner(Ppar, P);
            pmul(Qout, P, Ppar, Qin);
        }
        if (MODE == PMODE_GENERAL && !VEC)            // in-chunk prefix product, read back in phase D
            pstore_ws(Xw + (size_t)j * NN * stride, stride, Qout, par);
        __builtin_amdgcn_sched_barrier(0);
    };
    // every lane of this wave owns S slices inside the pulse (all waves but the last of a ragged decomposition):
    // no validity branches, the chunk product ping-pongs between two register sets (peeling the first slice
    // as well costs ~50 spilled VGPRs with this compiler).  Otherwise: the same steps under per-lane validity.
    if (__all(t0 + S <= Nsl)) {
        PMat<N> G;
        pset_identity(Q);
        int j = 0;
        for (; j + 1 < S; j += 2) {
            double nb = build(j, G);
            slice(j, G, nb, Q, Q2, j == 0);
            nb = build(j + 1, G);
            slice(j + 1, G, nb, Q2, Q, false);
        }
        if (j < S) {
            const double nb = build(j, G);
            slice(j, G, nb, Q, Q2, j == 0);
            Q = Q2;
        }
    } else {
        PMat<N> G;
        pset_identity(Q);
        for (int j = 0; j < S; ++j) {
            if (t0 + j < Nsl) {
                const double nb = build(j, G);
                slice(j, G, nb, Q, Q2, j == 0);
                Q = Q2;
            }
        }
    }

    pstamp(st, 1);
#if GRAPE_PD == 2
    // backward sweep's propagator ring (see PBuf): addresses, and with GRAPE_PD_EARLY the first one or two slices requested
    // BEFORE the scan -- HBM is idle during phase B and the sweep then has two slices fewer to wait for
    PBuf bA, bB;
    const unsigned pd_rowb = (unsigned)CH * 16u;
    const unsigned pd_voff0 = (unsigned)(10 * par) * pd_rowb + (unsigned)ch * 16u;
    const unsigned pd_voff1 = (unsigned)(2 + 6 * par) * pd_rowb + (unsigned)ch * 16u;
    const double2 *pd_Pk = p.props + (size_t)k * (GRAPE_PLAYOUT ? 1 : S) * NN * stride;   // wave-uniform
    auto pd_issue = [&](PBuf &b, int jj) {
        const double2 *s0 = pd_Pk + (size_t)max(jj, 0) * pstep;
        pbuf_issue(b, pd_voff0, pd_voff1, s0, s0 + CH, s0 + 4 * CH, s0 + 5 * CH);
    };
    // (no wait in front: phase A's stores may still be draining; the sweep starts with vmcnt(0), which covers both)
    auto pd_early = [&]() {
        if constexpr (PDASM && GRAPE_PD_EARLY >= 1) {
            const int jm0 = (XGLDS && p.plast_lds) ? S - 2 : S - 1;
            pd_issue(bA, jm0);
            if (GRAPE_PD_EARLY >= 2)
                pd_issue(bB, jm0 - 1);
        }
    };
    if (W == 1 || GRAPE_PD_EARLY_AT == 0)
        pd_early();
#endif
    // ---------------------------------------------------------------- phase B
    // Kogge-Stone over the 32 chunks of a wave (shuffle distance 2d lanes keeps the parity), wave totals
    // through LDS.  Unitary flow: M at the chunk end from the inclusive prefix U and the total T;
    // general flow: state at the chunk start and costate at the chunk end from prefix and suffix.
    PMat<N> Xs, Le;
    double zr = 0.0, zi = 0.0;
    {
        PMat<N> inc = Q, oth, tmp, ipar;
        for (int d = 1; d < 32; d <<= 1) {
            pshfl_up(oth, inc, 2 * d);
            if (cw >= d) {
                fetch_partner(ipar, inc);
                pmul(tmp, inc, ipar, oth);
                inc = tmp;
            }
        }
        if (!UNI) {                                  // exclusive prefix
            pshfl_up(oth, inc, 2);
            if (cw == 0)
                pset_identity(oth);
        }
        if (W > 1) {
            if (cw == 31)
                pstore_lds(&s_tot[0][wbase_tot + wave][par * NE], inc);
            if (GRAPE_MBAR == 1 && UNI)
                member_barrier(&s_cnt[mb], (unsigned)W);
            else if (GRAPE_MBAR == 2 && UNI)
                lds_barrier();
            else
                __syncthreads();
#if GRAPE_PD == 2
            if (GRAPE_PD_EARLY_AT == 1)                 // every wave of the workgroup has left phase A
                pd_early();
#endif
            if (wave > 0) {                          // (wave is uniform: the first wave's prefix is the identity -- nothing to do)
                PMat<N> pre, wt, wtp;
                pload_lds(pre, &s_tot[0][wbase_tot][par * NE]);          // total of wave 0
                for (int w = 1; w < wave; ++w) {
                    pload_lds(wt, &s_tot[0][wbase_tot + w][par * NE]);
                    pload_lds(wtp, &s_tot[0][wbase_tot + w][(1 - par) * NE]);
                    pmul(tmp, wt, wtp, pre);
                    pre = tmp;
                }
                if (UNI) {
                    fetch_partner(ipar, inc);
                    pmul(tmp, inc, ipar, pre);
                    inc = tmp;
                } else {
                    fetch_partner(ipar, oth);
                    pmul(tmp, oth, ipar, pre);
                    oth = tmp;
                }
            }
        }
        if (UNI) {
            if (ch == CH - 1)
                pstore_lds(&s_tot[1][wbase_tot][par * NE], inc);
            if (GRAPE_MBAR == 1)
                member_barrier(&s_cnt[mb], (unsigned)(W > 1 ? 2 * W : W));
            else if (GRAPE_MBAR == 2)
                lds_barrier();
            else
                __syncthreads();
            PMat<N> T, Tp, C0, xi_m, xi_o, xt_m, xt_o;
            pload_lds(T, &s_tot[1][wbase_tot][par * NE]);
            if (SAND) {
                pload_lds(Tp, &s_tot[1][wbase_tot][(1 - par) * NE]);
                pload_lds(xi_m, sXi);
                pload_lds(xi_o, sXi_o);
                pload_lds(xt_m, sXt);
                pload_lds(xt_o, sXt_o);
            }
            if (SAND) {
                PMat<N> Em, Ep;
                pmul(tmp, xt_m, xt_o, T);            // Xt T
                pmul_ah_b(Em, T, Tp, tmp);           // E = T' Xt T
                ptrace_ah_b(zr, zi, xi_m, Em);       // tr(Xi' E) = tr(X_t' L_t) for every t
                fetch_partner(Ep, Em);
                pmul_a_bh(C0, xi_m, xi_o, Em, Ep);   // Xi E'
                pmul_ah_b(tmp, Em, Ep, xi_m);        // E' Xi
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    C0.re[e] -= tmp.re[e];
                    C0.im[e] -= tmp.im[e];
                }
            } else {
                pload_lds(xi_m, sXX);                // Xi Xt', both halves prepared in the prologue
                pload_lds(xi_o, sXX_o);
                pmul(C0, xi_m, xi_o, T);
            }
            fetch_partner(ipar, inc);
            pmul(tmp, inc, ipar, C0);
            PMat<N> tp;
            fetch_partner(tp, tmp);
            pmul_a_bh(Xs, tmp, tp, inc, ipar);       // Xs := M at the chunk end = U C0 U'
        } else {
            PMat<N> xi_m;
            pload_lds(xi_m, sXi);
            fetch_partner(ipar, oth);
            if (SAND) {
                pmul(tmp, oth, ipar, xi_m);
                PMat<N> tp;
                fetch_partner(tp, tmp);
                pmul_a_bh(Xs, tmp, tp, oth, ipar);
            } else {
                pmul(Xs, oth, ipar, xi_m);
            }
        }
    }
    if (!UNI) {
        PMat<N> inc = Q, oth, tmp, opar;
        for (int d = 1; d < 32; d <<= 1) {
            pshfl_down(oth, inc, 2 * d);
            if (cw + d < 32) {
                fetch_partner(opar, oth);
                pmul(tmp, oth, opar, inc);
                inc = tmp;
            }
        }
        pshfl_down(oth, inc, 2);
        if (cw == 31)
            pset_identity(oth);
        if (W > 1) {
            if (cw == 0)
                pstore_lds(&s_tot[1][wbase_tot + wave][par * NE], inc);
            __syncthreads();
            if (wave + 1 < W) {                      // the last wave's suffix is the identity
                PMat<N> post, wt, wtp;
                pload_lds(post, &s_tot[1][wbase_tot + wave + 1][par * NE]);
                for (int w = wave + 2; w < W; ++w) {
                    pload_lds(wt, &s_tot[1][wbase_tot + w][par * NE]);
                    pload_lds(wtp, &s_tot[1][wbase_tot + w][(1 - par) * NE]);
                    pmul(tmp, wt, wtp, post);
                    post = tmp;
                }
                PMat<N> pp;
                fetch_partner(pp, post);
                pmul(tmp, post, pp, oth);
                oth = tmp;
            }
        }
        PMat<N> xt_m;
        pload_lds(xt_m, sXt);
        fetch_partner(opar, oth);
        if (SAND) {
            pmul_ah_b(tmp, oth, opar, xt_m);         // V' Xt
            PMat<N> tp;
            fetch_partner(tp, tmp);
            pmul(Le, tmp, tp, oth);                  // V' Xt V
        } else {
            pmul_ah_b(Le, oth, opar, xt_m);
        }
    }

    const double gs = SAND ? -1.0 : (p.variant == 0 ? -2.0 : 2.0);

    if (UNI) {
        pstamp(st, 2);
        pstamp(st, 3);
        // ------------------------------------------------------------ phase D, unitary flow
        // Every lane loads only its own columns of P_t (8 x 16 B) and takes the other half from its
        // partner by DPP; slice j-1's load is issued BEFORE slice j's products (two register buffers,
        // loop unrolled by two), so the HBM latency hides under ~300 VALU instructions of both waves.
        PMat<N> M = Xs, Mp, tmp, PA, PB, Pp;
        auto step = [&](int j, const PMat<N> &P) {
            const int t = t0 + j;
            if ((GRAPE_ABL & 64) && t < Nsl) {           // loads only: no products, no gradient
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    M.re[e] += P.re[e];
                    M.im[e] += P.im[e];
                }
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    acc += M.re[e] + M.im[e];
                if (par == 0) xg[j * K] = acc;
            } else
            if (t < Nsl) {
                fetch_partner(Pp, P);
                fetch_partner(Mp, M);
                pmul(tmp, M, Mp, P);
                pmul_ah_b(M, P, Pp, tmp);            // M_t = P' M_{t+1} P
                if (!SAND) {                         // z_t = tr(X_t' L_t) = conj(tr M_t)
                    double tr_r, tr_i;
                    ptrace(tr_r, tr_i, M);
                    zr = tr_r;
                    zi = -tr_i;
                }
                if (DUMPW) {                         // exact gradient: W_t = X_t L_{t+1}' = M_t P_t' (P_t unitary), Phi = tr M
                    fetch_partner(Mp, M);
                    pmul_a_bh(tmp, M, Mp, P, Pp);
                    pstore_ws(Xw + (size_t)j * NN * stride, stride, tmp, par);
                    if (t == Nsl - 1 && par == 0) {
                        p.zphi[2 * (size_t)k] = zr;
                        p.zphi[2 * (size_t)k + 1] = -zi;
                    }
                }
                if (GRAPE_ABL & 16) { if (par == 0) xg[j * K] = M.re[0] + zr; } else
                pwrite_gradient<N, SAND>(xg + j * K, sBT, K, M, zr, zi, gs, par);
                if (t == Nsl - 1 && par == 0)
                    s_F[mb] = pfigure_of_merit<N, SAND>(zr, zi);
            }
        };
        if (XGLDS && p.plast_lds) {                      // written by this very thread in phase A
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const double2 v = s_plast[e * blockDim.x + threadIdx.x];
                PA.re[e] = v.x;
                PA.im[e] = v.y;
            }
        } else if (!PDASM) {
            pload_ws(PA, Pw + (size_t)(S - 1) * pstep, stride, par);
        }
        // (Round 4, ISA: the compiler's s_waitcnt pass waits with vmcnt(0) in front of BOTH steps -- for the prefetch it has
        // just issued -- because the second load sits behind `if (j >= 2)` and PA enters the loop from LDS on one path and
        // from HBM on the other.  With both removed the waits are counted, vmcnt(15) .. vmcnt(8), the two buffers really
        // overlap -- and the kernel is 3 us SLOWER (74.3 vs 71.4 us, build/abl variants pd0..pd3, profiles/r04_C3_phaseD.txt):
        // sixteen 1 KB loads in flight per wave instead of eight, 128 KB per CU against a 32 KB L1.  The drain is what
        // paces phase D at the rate HBM delivers; left as it was.)
        int j = S - 1;
#if GRAPE_PD == 1
        // products and gradient of a slice apart: the products are the last readers of the slice's propagator, so the load of
        // the slice after next goes out between them -- unconditionally, from a clamped (always valid) address, so that the
        // compiler's s_waitcnt bookkeeping sees the same history on every path and waits for the OLDER buffer only
        auto products = [&](int jj, const PMat<N> &P) {
            if (t0 + jj < Nsl) {
                fetch_partner(Pp, P);
                fetch_partner(Mp, M);
                pmul(tmp, M, Mp, P);
                pmul_ah_b(M, P, Pp, tmp);            // M_t = P' M_{t+1} P
            }
        };
        auto gradient = [&](int jj) {
            const int t = t0 + jj;
            if (t < Nsl) {
                if (!SAND) {
                    double tr_r, tr_i;
                    ptrace(tr_r, tr_i, M);
                    zr = tr_r;
                    zi = -tr_i;
                }
                pwrite_gradient<N, SAND>(xg + jj * K, sBT, K, M, zr, zi, gs, par);
                if (t == Nsl - 1 && par == 0)
                    s_F[mb] = pfigure_of_merit<N, SAND>(zr, zi);
            }
        };
        if (DUMPW) {
            for (; j >= 0; --j) {
                if (j < S - 1 || !(XGLDS && p.plast_lds))
                    pload_ws(PA, Pw + (size_t)j * pstep, stride, par);
                step(j, PA);
            }
        } else {
            pload_ws(PB, Pw + (size_t)max(j - 1, 0) * NN * stride, stride, par);
            for (; j >= 1; j -= 2) {
                products(j, PA);
                pload_ws(PA, Pw + (size_t)max(j - 2, 0) * NN * stride, stride, par);
                gradient(j);
                products(j - 1, PB);
                pload_ws(PB, Pw + (size_t)max(j - 3, 0) * NN * stride, stride, par);
                gradient(j - 1);
            }
            if (j == 0) {
                products(0, PA);
                gradient(0);
            }
        }
#elif GRAPE_PD == 2
        if constexpr (PDASM) {
            auto products = [&](int jj, const PMat<N> &P) {
                if (t0 + jj < Nsl) {
                    fetch_partner(Pp, P);
                    fetch_partner(Mp, M);
                    pmul(tmp, M, Mp, P);
                    pmul_ah_b(M, P, Pp, tmp);            // M_t = P' M_{t+1} P
                }
            };
            auto gradient = [&](int jj) {
                const int t = t0 + jj;
                if (t < Nsl) {
                    if (!SAND) {
                        double tr_r, tr_i;
                        ptrace(tr_r, tr_i, M);
                        zr = tr_r;
                        zi = -tr_i;
                    }
                    pwrite_gradient<N, SAND>(xg + jj * K, sBT, K, M, zr, zi, gs, par);
                    if (t == Nsl - 1 && par == 0)
                        s_F[mb] = pfigure_of_merit<N, SAND>(zr, zi);
                }
            };
            const bool plast = XGLDS && p.plast_lds;
            const int jm = plast ? S - 2 : S - 1;                 // the first slice that comes from memory
            if (GRAPE_PD_EARLY >= 1) {
                pbuf_wait<0>(bA);                                 // issued before the scan: long landed
                if (GRAPE_PD_EARLY >= 2)
                    pbuf_wait<0>(bB);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // phase A's stores are long done: the count starts at zero
                pd_issue(bA, jm);
            }
            if (GRAPE_PD_EARLY < 2)
                pd_issue(bB, jm - 1);
            if (plast) {
                products(j, PA);
                gradient(j);
            }
            // slice jm in bA, slice jm - 1 in bB (landed or in flight); clamped loads once the slices run out
            for (j = jm; j >= 1; j -= 2) {
                pbuf_wait<8>(bA);
                pbuf_to_mat(PA, bA);
                products(j, PA);
                pd_issue(bA, j - 2);
                gradient(j);
                pbuf_wait<8>(bB);
                pbuf_to_mat(PB, bB);
                products(j - 1, PB);
                pd_issue(bB, j - 3);
                gradient(j - 1);
            }
            if (j == 0) {
                pbuf_wait<8>(bA);
                pbuf_to_mat(PA, bA);
                products(0, PA);
                gradient(0);
            }
            pbuf_wait<0>(bA);                                     // the clamped extra loads land before their registers are reused
            pbuf_wait<0>(bB);
        } else {
            for (; j >= 1; j -= 2) {
                pload_ws(PB, Pw + (size_t)(j - 1) * pstep, stride, par);
                step(j, PA);
                if (j >= 2)
                    pload_ws(PA, Pw + (size_t)(j - 2) * pstep, stride, par);
                step(j - 1, PB);
            }
            if (j == 0)
                step(0, PA);
        }
#else
        for (; j >= 1; j -= 2) {
            if (!(GRAPE_ABL & 8)) pload_ws(PB, Pw + (size_t)(j - 1) * pstep, stride, par);
            step(j, PA);
            if (j >= 2 && !(GRAPE_ABL & 8))
                pload_ws(PA, Pw + (size_t)(j - 2) * pstep, stride, par);
            step(j - 1, PB);
        }
        if (j == 0)
            step(0, PA);
#endif
    } else {
        pstamp(st, 2);
        // ------------------------------------------------------------ phase C (debug flow only)
        if (KEEPL) {
            PMat<N> X = Xs, Po, Pp, Xp, tmp;
            for (int j = 0; j < S; ++j) {
                const int t = t0 + j;
                if (t < Nsl) {
                    pstore_ws(Xw + (size_t)j * NN * stride, stride, X, par);
                    if (j + 1 < S) {
                        pload_ws(Po, Pw + (size_t)j * pstep, stride, par);
                        pload_ws(Pp, Pw + (size_t)j * pstep, stride, 1 - par);
                        if (SAND) {
                            fetch_partner(Xp, X);
                            pmul_a_bh(tmp, X, Xp, Po, Pp);       // X P'
                            pmul(X, Po, Pp, tmp);                // P X P'
                        } else {
                            pmul(tmp, Po, Pp, X);
                            X = tmp;
                        }
                    }
                }
            }
        }

        pstamp(st, 3);
        // ------------------------------------------------------------ phase D, general flow on n x 1 states (VEC)
        // X_t = [x_t 0 0 0], L_t = [w_t 0 0 0]: the chunk's states by a forward pass x_j+1 = P_j x_j from the chunk-start state the
        // scan delivered (kept in registers: 16 slices x 2 entries per lane), then w_j = P_j' w_j+1 backward with the gradient
        // g[c,t] = gs Re(z w_t' B'_c x_t), z = x_t' w_t -- matrix-vector products on the lane's OWN columns of P_j (one 128-byte
        // half per lane and pass: no partner half, no prefixes).  Per slice 3 x 256 B of workspace traffic instead of 4 x 256 B
        // and ~36 complex FMAs per lane instead of three 4 x 4 products.  Own-block-first order as everywhere (cmatp.hpp): a
        // lane's half of a vector = the entries of ITS index block, the partner's half arrives through DPP.
        if constexpr (VEC) {
            static_assert(N == 4 && !SAND && MODE == PMODE_GENERAL, "the vector sweep serves left multiplication at n = 4");
            constexpr int SV = kVecSlices;
            double xr[SV + 1][2], xi_[SV + 1][2];
            {   // column 0 of the padded matrices lives in lane 0 of the pair: rows 0, 1 are its own block, rows 2, 3 lane 1's
                const double a2 = pair_swap(Xs.re[2]), b2 = pair_swap(Xs.im[2]), a3 = pair_swap(Xs.re[3]), b3 = pair_swap(Xs.im[3]);
                xr[0][0] = par ? a2 : Xs.re[0];
                xi_[0][0] = par ? b2 : Xs.im[0];
                xr[0][1] = par ? a3 : Xs.re[1];
                xi_[0][1] = par ? b3 : Xs.im[1];
            }
            double wr[2], wi[2];
            {
                const double a2 = pair_swap(Le.re[2]), b2 = pair_swap(Le.im[2]), a3 = pair_swap(Le.re[3]), b3 = pair_swap(Le.im[3]);
                wr[0] = par ? a2 : Le.re[0];
                wi[0] = par ? b2 : Le.im[0];
                wr[1] = par ? a3 : Le.re[1];
                wi[1] = par ? b3 : Le.im[1];
            }
            PMat<N> Pj;
            // forward pass: the states at the slices of this chunk
#pragma unroll
            for (int j = 0; j < SV; ++j) {
                if (j < S && t0 + j < Nsl) {
                    pload_ws(Pj, Pw + (size_t)j * pstep, stride, par);
                    double yr[4], yi[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {          // y_loc[r] = P_loc[r, 0] x_own[0] + P_loc[r, 1] x_own[1]
                        yr[r] = Pj.re[r] * xr[j][0];
                        yr[r] = fma(-Pj.im[r], xi_[j][0], yr[r]);
                        yr[r] = fma(Pj.re[r + 4], xr[j][1], yr[r]);
                        yr[r] = fma(-Pj.im[r + 4], xi_[j][1], yr[r]);
                        yi[r] = Pj.re[r] * xi_[j][0];
                        yi[r] = fma(Pj.im[r], xr[j][0], yi[r]);
                        yi[r] = fma(Pj.re[r + 4], xi_[j][1], yi[r]);
                        yi[r] = fma(Pj.im[r + 4], xr[j][1], yi[r]);
                    }
                    // my rows 2, 3 are the partner's block: trade them for the partner's contribution to mine
                    xr[j + 1][0] = yr[0] + pair_swap(yr[2]);
                    xi_[j + 1][0] = yi[0] + pair_swap(yi[2]);
                    xr[j + 1][1] = yr[1] + pair_swap(yr[3]);
                    xi_[j + 1][1] = yi[1] + pair_swap(yi[3]);
                } else {
                    xr[j + 1][0] = xr[j][0];
                    xi_[j + 1][0] = xi_[j][0];
                    xr[j + 1][1] = xr[j][1];
                    xi_[j + 1][1] = xi_[j][1];
                }
            }
            // backward pass + gradient
#pragma unroll
            for (int j = SV - 1; j >= 0; --j) {
                if (j < S && t0 + j < Nsl) {
                    const int t = t0 + j;
                    pload_ws(Pj, Pw + (size_t)j * pstep, stride, par);
                    double lr[4], li[4];                   // w in my local row order: own block, then the partner's
                    lr[0] = wr[0]; li[0] = wi[0]; lr[1] = wr[1]; li[1] = wi[1];
                    lr[2] = pair_swap(wr[0]); li[2] = pair_swap(wi[0]); lr[3] = pair_swap(wr[1]); li[3] = pair_swap(wi[1]);
#pragma unroll
                    for (int jl = 0; jl < 2; ++jl) {       // (P' w)[own column jl] = sum_r conj(P_loc[r, jl]) w_loc[r]
                        double ar = 0.0, ai = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ar = fma(Pj.re[r + 4 * jl], lr[r], ar);
                            ar = fma(Pj.im[r + 4 * jl], li[r], ar);
                            ai = fma(Pj.re[r + 4 * jl], li[r], ai);
                            ai = fma(-Pj.im[r + 4 * jl], lr[r], ai);
                        }
                        wr[jl] = ar;
                        wi[jl] = ai;
                    }
                    // z = x_t' w_t (own halves + the partner's)
                    double zr_ = xr[j][0] * wr[0];
                    zr_ = fma(xi_[j][0], wi[0], zr_);
                    zr_ = fma(xr[j][1], wr[1], zr_);
                    zr_ = fma(xi_[j][1], wi[1], zr_);
                    double zi_ = xr[j][0] * wi[0];
                    zi_ = fma(-xi_[j][0], wr[0], zi_);
                    zi_ = fma(xr[j][1], wi[1], zi_);
                    zi_ = fma(-xi_[j][1], wr[1], zi_);
                    zr_ += pair_swap(zr_);
                    zi_ += pair_swap(zi_);
                    // w_t in local row order again (it has just changed)
                    lr[0] = wr[0]; li[0] = wi[0]; lr[1] = wr[1]; li[1] = wi[1];
                    lr[2] = pair_swap(wr[0]); li[2] = pair_swap(wi[0]); lr[3] = pair_swap(wr[1]); li[3] = pair_swap(wi[1]);
                    for (int c = 0; c < K; ++c) {          // a = w' B'_c x over my columns, pair sum; g = gs Re(z a)
                        const double2 *bc = sB + (size_t)c * NE;
                        double ar = 0.0, ai = 0.0;
#pragma unroll
                        for (int jl = 0; jl < 2; ++jl) {
                            double ur = 0.0, ui = 0.0;     // u = sum_r conj(w_loc[r]) B'_loc[r, jl]
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const double2 b = bc[r + 4 * jl];
                                ur = fma(lr[r], b.x, ur);
                                ur = fma(li[r], b.y, ur);
                                ui = fma(lr[r], b.y, ui);
                                ui = fma(-li[r], b.x, ui);
                            }
                            ar = fma(ur, xr[j][jl], ar);
                            ar = fma(-ui, xi_[j][jl], ar);
                            ai = fma(ur, xi_[j][jl], ai);
                            ai = fma(ui, xr[j][jl], ai);
                        }
                        ar += pair_swap(ar);
                        ai += pair_swap(ai);
                        if (par == 0)
                            xg[j * K + c] = gs * fma(zr_, ar, -zi_ * ai);
                    }
                    if (t == Nsl - 1 && par == 0)
                        s_F[mb] = pfigure_of_merit<N, SAND>(zr_, zi_);
                }
            }
        } else {
        // ------------------------------------------------------------ phase D, general flow
        PMat<N> Lc = Le, Po, Pp, X, Xp, M, tmp, Lp;
        for (int j = S - 1; j >= 0; --j) {
            const int t = t0 + j;
            if (t < Nsl) {
                pload_ws(Po, Pw + (size_t)j * pstep, stride, par);
                pload_ws(Pp, Pw + (size_t)j * pstep, stride, 1 - par);
                if (SAND) {
                    fetch_partner(Lp, Lc);
                    pmul(tmp, Lc, Lp, Po);           // L P
                    pmul_ah_b(Lc, Po, Pp, tmp);      // P' L P
                } else {
                    pmul_ah_b(tmp, Po, Pp, Lc);
                    Lc = tmp;
                }
                if (KEEPL) {
                    pload_ws(X, Xw + (size_t)j * NN * stride, stride, par);
                } else if (j > 0) {                  // X_t = Q_{j-1} Xs [Q_{j-1}']
                    PMat<N> Qo, Qp;
                    pload_ws(Qo, Xw + (size_t)(j - 1) * NN * stride, stride, par);
                    pload_ws(Qp, Xw + (size_t)(j - 1) * NN * stride, stride, 1 - par);
                    if (SAND) {
                        pmul(tmp, Qo, Qp, Xs);
                        PMat<N> tp;
                        fetch_partner(tp, tmp);
                        pmul_a_bh(X, tmp, tp, Qo, Qp);
                    } else {
                        pmul(X, Qo, Qp, Xs);
                    }
                } else {
                    X = Xs;
                }
                if (KEEPL)
                    pstore_ws(p.costates + wbase + (size_t)j * NN * stride, stride, Lc, par);
                double zr, zi;
                ptrace_ah_b(zr, zi, X, Lc);          // tr(X' L)
                fetch_partner(Xp, X);
                fetch_partner(Lp, Lc);
                pmul_a_bh(M, X, Xp, Lc, Lp);         // X L'
                if (SAND) {
                    pmul_ah_b(tmp, Lc, Lp, X);       // L' X
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        M.re[e] -= tmp.re[e];
                        M.im[e] -= tmp.im[e];
                    }
                }
                pwrite_gradient<N, SAND>(xg + j * K, sBT, K, M, zr, zi, gs, par);
                if (t == Nsl - 1 && par == 0)
                    s_F[mb] = pfigure_of_merit<N, SAND>(zr, zi);
            }
        }
        }
    }
    // results: LDS -> HBM, lane-contiguous.  (1) this member's unweighted row, (2) the block's weighted partial sum
    __syncthreads();
    const int KN = K * Nsl;
    if (p.member_out && valid) {
        for (int q = L; q < KN; q += LT)
            out[q] = s_xg[q + chunk_of(q)];
        if (L == 0)
            out[KN] = s_F[mb];
    }
    {
        const int nmem = min(p.MPB, p.E - bi * p.MPB);
        const double *__restrict__ wb = wts_all + (size_t)bi * p.MPB;
        double *__restrict__ bout = p.block_out + (size_t)blockIdx.x * (KN + 1);
        const int mstride = CH * (SK + 1);
        for (int q = threadIdx.x; q <= KN; q += blockDim.x) {
            const int off = q + chunk_of(q);
            double acc = 0.0;
            for (int m = 0; m < nmem; ++m) {
                const double v = (q < KN) ? s_xg_all[m * mstride + off] : s_F[m];
                acc = fma(v, wb[m], acc);
            }
            bout[q] = acc;
            if (p.direct_dst)
                p.direct_dst[q] = acc;
        }
    }
    if (p.direct_flag) {               // one workgroup: the result is complete, tell the host (reduce.hip: signal_done)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(p.direct_flag, p.direct_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    pstamp(st, 4);
    if (st && lane == 0) {
        st[6] = __builtin_amdgcn_s_memrealtime();
        st[7] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
}

template <int N>
struct PairTraits;
template <> struct PairTraits<2> { static constexpr int MAXT = 1024; };
template <> struct PairTraits<4> { static constexpr int MAXT = 512; };

int sweep_pair_max_waves(int n)
{
    switch (n) {
    case 2: return PairTraits<2>::MAXT / 64;
    case 4: return PairTraits<4>::MAXT / 64;
    default: return 0;
    }
}

size_t sweep_pair_lds_bytes(int n, int MPB, int LT, int S, int K, bool xg_in_lds, bool plast)
{
    const int maxt = n == 2 ? PairTraits<2>::MAXT : PairTraits<4>::MAXT;
    size_t b = sizeof(double2) * (2 * (size_t)(maxt / 64) * n * n + 4);
    b += sizeof(double2) * ((size_t)MPB * 2 * ((2 * K + 4) * (n * (n / 2)) + kParityPad) + ((size_t)MPB * (K + 1) + 1) / 2);
    if (xg_in_lds)
        b += sizeof(double) * ((size_t)MPB * (LT / 2) * ((size_t)S * K + 1) + ((MPB + 1) & ~1));
    if (plast)
        b += sizeof(double2) * (size_t)MPB * LT * (n * (n / 2));
    return b;
}

template <int N, int SAND, int MODE, bool XGLDS>
static hipError_t plaunch_one(const SweepParams &p0, hipStream_t stream)
{
    constexpr int MAXT = PairTraits<N>::MAXT;
    SweepParams p = p0;
    const dim3 grid(p.BPX * p.n_x), block(p.LT * p.MPB);
    size_t lds = sweep_pair_lds_bytes(N, p.MPB, p.LT, p.S, p.K, XGLDS);
    if (lds > 160 * 1024)
        return hipErrorInvalidConfiguration;
    p.plast_lds = 0;
    {
        static const int tune_env = [] { const char *e = getenv("GRAPE_PAIR_TUNE"); return e ? atoi(e) : 0; }();
        p.tune = tune_env;
    }
    if (MODE == PMODE_UNITARY && XGLDS && p.S > 1) {         // room for the chunks' last propagators?
        const size_t with = sweep_pair_lds_bytes(N, p.MPB, p.LT, p.S, p.K, true, true);
        if (with <= 160 * 1024) {
            p.plast_lds = 1;
            lds = with;
        }
    }
    if constexpr (MODE == PMODE_UNITARY && SAND == 0) {
        if (p.dump_w1) {
            auto kern_w = sweep_pair_kernel<N, SAND, MODE, MAXT, XGLDS, true>;
            if (lds > 64 * 1024) {
                hipError_t e = ensure_dynamic_lds((const void *)kern_w, lds);
                if (e != hipSuccess)
                    return e;
            }
            GRAPE_LAUNCH_AS("sweep_pair_kernel", kern_w, grid, block, lds, stream, p.ops, p.x, p.wts, p);
            return hipGetLastError();
        }
    }
    if (p.dump_w1)
        return hipErrorInvalidConfiguration;
    if constexpr (N == 4 && SAND == 0 && MODE == PMODE_GENERAL && XGLDS) {
        if (p.vec && p.S <= kVecSlices) {                    // n x 1 states: the sweep back on vectors
            auto kern_v = sweep_pair_kernel<N, SAND, MODE, MAXT, XGLDS, false, true>;
            if (lds > 64 * 1024) {
                hipError_t e = ensure_dynamic_lds((const void *)kern_v, lds);
                if (e != hipSuccess)
                    return e;
            }
            GRAPE_LAUNCH_AS("sweep_pair_vec_kernel", kern_v, grid, block, lds, stream, p.ops, p.x, p.wts, p);
            return hipGetLastError();
        }
    }
    auto kern = sweep_pair_kernel<N, SAND, MODE, MAXT, XGLDS>;
    if (lds > 64 * 1024) {
        hipError_t e = ensure_dynamic_lds((const void *)kern, lds);
        if (e != hipSuccess)
            return e;
    }
    GRAPE_LAUNCH_AS("sweep_pair_kernel", kern, grid, block, lds, stream, p.ops, p.x, p.wts, p);
    return hipGetLastError();
}

template <int N, int SAND>
static hipError_t plaunch_ns(int mode, const SweepParams &p, hipStream_t stream)
{
    constexpr int MAXT = PairTraits<N>::MAXT;
    if (p.MPB < 1 || p.LT * p.MPB > MAXT || (p.LT & 63) || (long long)p.S * (p.LT / 2) < p.N)
        return hipErrorInvalidConfiguration;
    const bool lds = p.xg_scratch == nullptr;
    switch (mode) {
    case PMODE_GENERAL:
        return lds ? plaunch_one<N, SAND, PMODE_GENERAL, true>(p, stream) : plaunch_one<N, SAND, PMODE_GENERAL, false>(p, stream);
    case PMODE_GENERAL_KEEPL:
        return lds ? plaunch_one<N, SAND, PMODE_GENERAL_KEEPL, true>(p, stream)
                   : plaunch_one<N, SAND, PMODE_GENERAL_KEEPL, false>(p, stream);
    case PMODE_UNITARY:
        return lds ? plaunch_one<N, SAND, PMODE_UNITARY, true>(p, stream) : plaunch_one<N, SAND, PMODE_UNITARY, false>(p, stream);
    default:
        return hipErrorInvalidValue;
    }
}

hipError_t launch_sweep_pair(int n, int sandwich, int mode, const SweepParams &p, hipStream_t stream)
{
    switch (n * 2 + (sandwich ? 1 : 0)) {
    case 4: return plaunch_ns<2, 0>(mode, p, stream);
    case 5: return plaunch_ns<2, 1>(mode, p, stream);
    case 8: return plaunch_ns<4, 0>(mode, p, stream);
    case 9: return plaunch_ns<4, 1>(mode, p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace grape
