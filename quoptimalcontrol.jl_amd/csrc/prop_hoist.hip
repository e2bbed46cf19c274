// prop_hoist.hip -- pw_prop_save! (src/timeevolution.jl:98-110) of the tile family when the control operators are the
// SAME for every ensemble member (B_gens = k -> [Sx, Sy], test/setup_tests.jl:32; every BASELINE config):
//
//   ctrl_sum_kernel     once per evaluation and slice:  Gc_t = (-i dt) sum_c x[c,t] B_c  as a D-layout dump, and its
//                       norm bound |Gc_t|_1 (max column sum of |re| + |im|);
//   prop_hoist_kernel   per (member, slice):  G = A'_k + Gc_t  with A'_k = (-i dt) A_k prepared at grape_set_operators
//                       (one tile add instead of K tile FMAs + K operator-tile reads), squarings from the bound
//                       |A'_k|_1 + |Gc_t|_1 >= |G|_1 (two scalars: no cross-lane norm reduction), Taylor-8 on the
//                       FP64 matrix cores, P_t dumped in D layout (rank-one chain: transposed for odd t).
//
// The reference sums the controls first and adds A last (variant 0; the static variant adds A first): hoisting the
// control sum keeps that association and only moves the scalar (-i dt) inside the sum -- a rounding-level change.
//
// Why this file exists (profiles/r02_C4_E1024_pmc.json, tools/ubench/pipe_mix.hip): on gfx950 a SIMD does not overlap
// v_mfma_f64_16x16x4 with vector instructions of the waves that issue the MFMAs -- a propagator costs
// 64 cycles x MFMAs + ~4.3 cycles x VALU instructions, whatever the occupancy.  The round-2 kernel issued ~490 vector
// instructions per 38.6 MFMAs (16 x 16) and 2255 per 319 (32 x 32, two thirds of them register moves between the
// VGPR and AGPR halves of a 420-register allocation).  Here every remaining vector instruction is arithmetic the
// algorithm needs: no packing moves in front of LDS stores (ds_write2_b64 takes re and im from separate registers),
// no cross-lane norm, no control FMAs, negations folded into operands, the identity added through a 0/1 register.
#include <algorithm>
#include <cstdlib>

#include "cmat.hpp"          // Taylor-8 coefficients, squarings_for
#include "grape_kernels.hpp"
#include "tile.hpp"

namespace grape {

// ---------------------------------------------------------------------------------------------------------------
// Gc_t and its norm bound: one workgroup per (slice, control array), one wave per tile (a single problem's evaluation
// waits for this pre-pass: the K dependent tile loads of the four tiles run side by side)
template <int NT>
__global__ __launch_bounds__(64 * NT * NT) void ctrl_sum_kernel(const TileParams p)
{
    constexpr int TSZ = NT * NT * 256;
    __shared__ double s_col[NT * NT][16];
    const int lane = threadIdx.x & 63, tile = threadIdx.x >> 6, t = blockIdx.x, z = blockIdx.y;
    const int K = p.K;
    const double2 *__restrict__ B0 = (p.ops_ref ? p.ops_ref : p.ops) + TSZ + tile * 256;     // unit 0: [A | B_1..B_K | ...]
    const double *__restrict__ x = p.x + (size_t)z * K * p.N + (size_t)t * K;
    double2 *__restrict__ out = p.gc + ((size_t)z * p.N + t) * TSZ + tile * 256;
    const double dt = p.dt;
    double hr[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    for (int c = 0; c < K; ++c) {
        const double xv = x[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 b = B0[(size_t)c * TSZ + r * 64 + lane];
            hr[r] = fma(b.x, xv, hr[r]);
            hi[r] = fma(b.y, xv, hi[r]);
        }
    }
    double cs = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double gr = dt * hi[r], gi = -dt * hr[r];            // (-i dt) H
        out[r * 64 + lane] = make_double2(gr, gi);
        cs += fabs(gr) + fabs(gi);
    }
    cs = swap16_add(cs, cs);                                       // rows live on lane >> 4 and r: column sums of this tile
    cs = swap32_add(cs, cs);
    if (NT > 1) {                                                  // (every wave reaches every barrier)
        if (lane < 16)
            s_col[tile][lane] = cs;
        __syncthreads();
        cs = 0.0;                                                  // waves 0..NT-1: column block J summed over the tile rows
        if (tile < NT) {
#pragma unroll
            for (int I = 0; I < NT; ++I)
                cs += s_col[I * NT + tile][lane & 15];
        }
    }
    double colmax = wave_max_fast(cs);
    if (NT > 1) {
        __syncthreads();
        if (lane == 0)
            s_col[tile][0] = colmax;
        __syncthreads();
        colmax = 0.0;
#pragma unroll
        for (int J = 0; J < NT; ++J)
            colmax = fmax(colmax, s_col[J][0]);
    }
    if (lane == 0 && tile == 0)
        p.gcn[(size_t)z * p.N + t] = colmax * (1.0 / kTheta8);     // the expm kernel adds |A'_k|_1 / theta8
}

// ---------------------------------------------------------------------------------------------------------------
// LDS layout conversion without packing moves.  The image of tile.hpp (slot = 68 r + 17 (lane >> 4) + (lane & 15) for
// the D registers; the A operand of k-block kb at slot 68 (rho >> 2) + 17 (rho & 3) + q + 4 kb) is written with
// ds_write2_b64: re and im come from their own register pairs and land in adjacent 8-byte words -- the same
// interleaved double2 image as a ds_write_b128 of a packed quad, minus the eight v_mov_b32 that packing costs.
// Offsets are in 8-byte units and at most 255: rows 2, 3 go through a second base address (+ 2 x 68 slots).
// raw: `im` (or `re`) may be the untouched result of an MFMA.  The compiler's hazard recogniser does not look inside
// inline assembly, so the wait states an LDS store needs behind a v_mfma_f64_16x16x4 that wrote its data register
// (18, CDNA3/4 ISA "XDL write VGPR -> LDS read") are spelled out here.  raw = false is for data a vector instruction
// produced (the recogniser has put the wait in front of THAT instruction).
GRAPE_DEV void img_write_tile(unsigned base, const d4 &re, const d4 &im, bool raw)
{
    if (raw)
        asm volatile("s_nop 15\n\ts_nop 2" ::: "memory");
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" : : "v"(base), "v"(re[0]), "v"(im[0]) : "memory");
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:136 offset1:137" : : "v"(base), "v"(re[1]), "v"(im[1]) : "memory");
    const unsigned base2 = base + 2 * 68 * 16;
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" : : "v"(base2), "v"(re[2]), "v"(im[2]) : "memory");
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:136 offset1:137" : : "v"(base2), "v"(re[3]), "v"(im[3]) : "memory");
}

typedef double d2v __attribute__((ext_vector_type(2)));

// the A operand of one tile: four (re, im) pairs, one per k-block, each an aligned register quad
struct AOp {
    d2v v[4];
};

// the four reads and the wait for them in one statement: the outputs exist only behind the s_waitcnt, so no consumer can
// be scheduled in front of it.  LDS operations of one wave execute in issue order: the reads see the image that the
// same wave's img_write_tile stored just before, without a wait in between.
GRAPE_DEV void img_read_tile(AOp &a, unsigned rd)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                 "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a.v[0]), "=&v"(a.v[1]), "=&v"(a.v[2]), "=&v"(a.v[3])
                 : "v"(rd)
                 : "memory");
}

// the image read back as it was written (D registers r = 0..3 of the tile, re and im packed in one register quad): what a
// 16-byte global store wants -- the LDS round trip packs the pairs without a vector instruction
GRAPE_DEV void img_read_rows(AOp &a, unsigned wr)
{
    const unsigned wr2 = wr + 2 * 68 * 16;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1088\n\tds_read_b128 %2, %5\n\t"
                 "ds_read_b128 %3, %5 offset:1088\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a.v[0]), "=&v"(a.v[1]), "=&v"(a.v[2]), "=&v"(a.v[3])
                 : "v"(wr), "v"(wr2)
                 : "memory");
}

// x += c in the lanes of `mask` only (the diagonal of a D-layout tile): exec is narrowed around ONE v_add_f64 -- no 0/1
// registers, no select
GRAPE_DEV double add_masked(double x, double c, unsigned long long mask)
{
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tv_add_f64 %0, %0, %2\n\ts_mov_b64 exec, %1"
                 : "+v"(x), "=&s"(save)
                 : "s"(c), "s"(mask));
    return x;
}

// LDS word read / write by byte address (the hand-over flag of the fused forward pass: a volatile C++ access would
// go through the flat aperture)
GRAPE_DEV int lds_load_word(unsigned addr)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
GRAPE_DEV void lds_store_word(unsigned addr, int v)
{
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

// a pointer every lane agrees on, moved to scalar registers and typed as GLOBAL memory: accesses then take the SGPR base +
// 32-bit lane offset form of global_load / global_store instead of a 64-bit address per lane kept (or spilled) across
// the slice loop.  (Elements are plain vector types: HIP's double2 class does not live in a named address space.)
typedef __attribute__((address_space(1))) d2v *gptr;
typedef const __attribute__((address_space(1))) d2v *gcptr;
GRAPE_DEV gptr uniform_global(const void *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gptr)(((unsigned long long)hi << 32) | lo);
}

// The waves a SIMD hosts run the same program from the same start, and its arbiter favours the oldest: at C4 the first
// workgroups of the fused-forward grid finished after 53 % of the kernel's run time and the last ones had the SIMDs to
// themselves (one wave alone cannot cover its own LDS and matrix-core latencies).  A priority that rotates with TIME over
// the wave slots -- every wave of a SIMD holds each of the four levels for the same share of the time, never two the
// same level -- lets them progress together: all workgroups end within 10 % of each other, the kernel 5-6 % earlier.
GRAPE_DEV void rotate_priority()
{
    const int slot = __builtin_amdgcn_s_getreg(6148);              // HW_REG_HW_ID[3:0]: wave slot on its SIMD
    const int pr = ((int)(__builtin_amdgcn_s_memtime() >> 16) + slot) & 3;
    if (pr == 0) __builtin_amdgcn_s_setprio(0);
    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

#define GRAPE_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// one 16 x 16 complex tile product  (re, im) = A * W  with A an AOp and W in D layout, three real products
// (tile.hpp tprod), written so that the only vector instructions are the 8 operand sums and the 8 combinations
GRAPE_DEV void tile_prod(d4 &ore, d4 &oim, const AOp &a, const d4 &wre, const d4 &wim)
{
    // vector instructions placed BETWEEN matrix-core instructions cost ~7 cycles each instead of ~4.4 in a burst
    // (tools/ubench/pipe_mix.hip, experiment C): the operand sums go in front of the first chain, the combinations
    // between the chains, and the scheduler is kept from spreading them (sched_barrier)
    double as[4], bs[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        as[kb] = a.v[kb][0] + a.v[kb][1];
        bs[kb] = wre[kb] + wim[kb];
    }
    __builtin_amdgcn_sched_barrier(0);
    d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        t1 = GRAPE_MFMA(a.v[kb][0], wre[kb], t1);
        t2 = GRAPE_MFMA(a.v[kb][1], wim[kb], t2);
    }
    __builtin_amdgcn_sched_barrier(0);
    d4 t3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ore[r] = t1[r] - t2[r];
        t3[r] = -t1[r] - t2[r];                                    // one v_add_f64 with both operands negated
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
        t3 = GRAPE_MFMA(as[kb], bs[kb], t3);
    __builtin_amdgcn_sched_barrier(0);
    oim = t3;
}

// Taylor-8 constants rearranged so that two of the three combinations are a single FMA per element:
//   A4' = A2 (A2 + c1 G)            = A4 / x2
//   A8' = (A4' + c3 A2) (x4 I + x5 G + x6 A2 + c7 A4')   = A8 / x2
//   P   = x2 A8' + (I + G + y2 A2)
constexpr double kC1 = kX1 / kX2, kC3 = kX3 / kX2, kC7 = kX7 * kX2;

constexpr int kHoistWaves = 4;

// NT = 1.  FUSE: rank-one chain with one workgroup per member (sweep_tile.hip, tile_fuse_forward): wave w walks the
// slices t = w mod 4 and the forward vector v_{t+1} = P_t v_t goes from wave to wave through LDS.  Here EVERY slice's
// propagator is converted to its transposed registers (one LDS round trip, no vector instruction) before the wave
// waits for its turn, so the serial hand-over is the short form for all t: gathered reads of v from LDS, 16 FMAs, a
// sum over the four lane rows (two permlane swaps), per-column write.
// HOIST: Gc_t comes from the pre-pass (member-invariant control operators).  !HOIST: the member has its own control
// operators (amplitude-scaled controls of a robustness ensemble, ...): the control sum is formed here from the member's
// operator dumps, G = A'_k + sum_c (dt x[c,t]) (Im B_c, -Re B_c), the squaring bound from |A'_k| + sum_c |x_c| |B'_c| --
// everything behind the H build is the same kernel.
template <bool FUSE, bool HOIST>
__global__ __launch_bounds__(64 * kHoistWaves, 4) void prop_hoist1_kernel(const TileParams p)
{
    constexpr int TSZ = 256, WPB = kHoistWaves;
    extern __shared__ double2 s_hoist[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.y, z = blockIdx.z;
    double2 *img = s_hoist + (size_t)wave * kTileImage;
    double2 *s_vec = s_hoist + (size_t)WPB * kTileImage;           // FUSE: [v even | v odd | flag]
    const unsigned flag_addr = (unsigned)(size_t)(s_vec + 32);     // LDS byte addresses: the low 32 bits of a __shared__ pointer
    const unsigned img_base = (unsigned)(size_t)img;
    const unsigned wr = img_base + 16u * (17 * (lane >> 4) + (lane & 15));
    const unsigned rd = img_base + 16u * (68 * ((lane & 15) >> 2) + 17 * (lane & 3) + (lane >> 4));
    if (FUSE) {
        if (threadIdx.x < 16)
            s_vec[threadIdx.x] = p.vecs[(size_t)k * 32 + threadIdx.x];
        if (threadIdx.x == 0)
            *reinterpret_cast<int *>(s_vec + 32) = 0;
        __syncthreads();
    }
    // A'_k stays in registers for every slice this wave walks; the diagonal of the tile as four lane masks
    d4 Are, Aim;
    unsigned long long dmask[4];
    {
        const gcptr ha = uniform_global(p.ha + (size_t)k * TSZ);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const d2v v = ha[r * 64 + lane];
            Are[r] = v[0];
            Aim[r] = v[1];
            dmask[r] = __builtin_amdgcn_ballot_w64(4 * r + (lane >> 4) == (lane & 15));
        }
    }
    const int K = p.K;
    const int nstride = HOIST ? 1 : 1 + K;                         // ha_norm: [unit] or [unit][|A'|, |B'_1| .. |B'_K|], all / theta8
    const double nA = p.ha_norm[(size_t)k * nstride];
    const double sk = (HOIST && p.ctrl_scale) ? p.ctrl_scale[k] : 1.0;      // B_k = s_k B_0: G = A'_k + s_k Gc_t
    const double *__restrict__ gcn = p.gcn + (size_t)z * p.N;
    const gcptr gc = uniform_global(p.gc + (size_t)z * p.N * TSZ);
    const gcptr Bk = uniform_global(p.ops + ((size_t)k * (2 * K + 3) + 1) * TSZ);      // !HOIST: this member's B_1 .. B_K dumps
    const double *__restrict__ xz = p.x + (size_t)z * K * p.N;
    const gptr props = uniform_global(p.props + ((size_t)z * p.E + k) * (size_t)p.N * TSZ);
    // p.thin == 2 (chain_prop_kernel): P_t AND P_t^T go to memory, both as they are -- the two vector chains then read their
    // rows lane-contiguously
    const gptr props_t = uniform_global((p.thin == 2 ? p.props_t : p.props) + ((size_t)z * p.E + k) * (size_t)p.N * TSZ);
    const gptr V = uniform_global(FUSE ? p.states + ((size_t)z * p.E + k) * (size_t)(p.N + 1) * 16 : p.states);
    const int t_lo = FUSE ? 0 : blockIdx.x * p.prop_slices;
    const int t_hi = FUSE ? p.N : min(p.N, t_lo + p.prop_slices);
    int t = t_lo + wave;
    // The loop is ROTATED (see prop_hoist2_kernel): the next slice's generator is formed at the END of the body, where the
    // wait for its loads has one history -- the loads, then this slice's stores -- and is counted; at the top of the body it
    // also covered the entry path, whose loads are the last vector-memory operations, and waited for the stores' acks.
    d2v gnext[4];
    double bnext = 0.0;                                            // HOIST: the next slice's norm bound, FIRST of its prefetch
    d4 Gre, Gim;
    double bound = 0.0;
    auto prefetch = [&](int tn) {
        if (!HOIST) return;
        const gcptr g1 = gc + (size_t)tn * TSZ;
        bnext = gcn[tn];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            gnext[r] = g1[r * 64 + lane];
    };
    auto form_generator = [&](int tn) {
        if (HOIST) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Gre[r] = fma(sk, gnext[r][0], Are[r]);             // (s_k = 1: the plain sum, bit for bit)
                Gim[r] = fma(sk, gnext[r][1], Aim[r]);
            }
            bound = fma(fabs(sk), bnext, nA);
        } else {
            Gre = Are;
            Gim = Aim;
            bound = nA;
            for (int c = 0; c < K; ++c) {
                const double xv = xz[c + (size_t)tn * K], sx = p.dt * xv;
                bound = fma(fabs(xv), p.ha_norm[(size_t)k * nstride + 1 + c], bound);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const d2v b = Bk[(size_t)c * TSZ + r * 64 + lane];
                    Gre[r] = fma(sx, b[1], Gre[r]);                // (-i dt x)(br + i bi) = dt x (bi - i br)
                    Gim[r] = fma(-sx, b[0], Gim[r]);
                }
            }
        }
    };
    if (t < t_hi) {
        prefetch(t);
        form_generator(t);
    }
    for (; t < t_hi; t += WPB) {
        rotate_priority();
        const int s = p.s_forced >= 0 ? p.s_forced : squarings_from_ratio(bound);          // bounds are stored / theta8
        if (s > 0) {
            const double sc = ldexp(1.0, -s);
            Gre *= sc;
            Gim *= sc;
        }
        AOp opa;
        d4 A2re, A2im, A4re, A4im, Tre, Tim, Ure, Uim;
        img_write_tile(wr, Gre, Gim, false);
        img_read_tile(opa, rd);
        tile_prod(A2re, A2im, opa, Gre, Gim);                      // A2 = G G
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Tre[r] = fma(kC1, Gre[r], A2re[r]);
            Tim[r] = fma(kC1, Gim[r], A2im[r]);
        }
        __builtin_amdgcn_sched_barrier(0);                         // A2im is a raw MFMA result: its first readers are the FMAs above
        img_write_tile(wr, A2re, A2im, false);
        img_read_tile(opa, rd);
        tile_prod(A4re, A4im, opa, Tre, Tim);                      // A4' = A2 (A2 + c1 G)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Ure[r] = fma(kC3, A2re[r], A4re[r]);
            Uim[r] = fma(kC3, A2im[r], A4im[r]);
            Tre[r] = fma(kC7, A4re[r], fma(kX6, A2re[r], kX5 * Gre[r]));
            Tim[r] = fma(kC7, A4im[r], fma(kX6, A2im[r], kX5 * Gim[r]));
            Tre[r] = add_masked(Tre[r], kX4, dmask[r]);
        }
        img_write_tile(wr, Ure, Uim, false);
        img_read_tile(opa, rd);
        d4 Pre, Pim;
        tile_prod(Pre, Pim, opa, Tre, Tim);                        // A8'
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Pre[r] = fma(kX2, Pre[r], fma(kY2, A2re[r], Gre[r]));
            Pim[r] = fma(kX2, Pim[r], fma(kY2, A2im[r], Gim[r]));
            Pre[r] = add_masked(Pre[r], 1.0, dmask[r]);
        }
        // the next slice's control sum: in flight during the squarings (G, A2 and A4' are dead by now), the conversions, the
        // hand-over and the stores below
        const int tn = min(t + WPB, t_hi - 1);                     // (clamped: no branch around the loads)
        prefetch(tn);
        for (int i = 0; i < s; ++i) {
            img_write_tile(wr, Pre, Pim, i > 0);
            img_read_tile(opa, rd);
            d4 Qre, Qim;
            tile_prod(Qre, Qim, opa, Pre, Pim);
            Pre = Qre;
            Pim = Qim;
        }
        // P_t goes to memory through the image: read back as written for P_t itself, transposed for P_t^T (rank-one
        // chain: odd slices are stored transposed, and the fused forward pass multiplies by the transposed registers)
        const bool transposed = p.thin == 1 && (t & 1);
        const gptr dst = props + (size_t)t * TSZ;
        img_write_tile(wr, Pre, Pim, s > 0);
        if (!transposed) {
            AOp pk;
            img_read_rows(pk, wr);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[r * 64 + lane] = pk.v[r];
        }
        if (FUSE || transposed || p.thin == 2)
            img_read_tile(opa, rd);                                // opa.v[r] = {re, im} of P^T's D register r
        if (transposed) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[r * 64 + lane] = opa.v[r];
        }
        if (p.thin == 2) {
            const gptr dst_t = props_t + (size_t)t * TSZ;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst_t[r * 64 + lane] = opa.v[r];
        }
        if (FUSE && p.fuse_fwd == 1) {
            // the serial part of the kernel (N hand-overs per member): between the flag read and the flag write there is
            // only LDS traffic -- no memory fence, the global stores above are not waited for
            const int g = lane >> 4, c = lane & 15;
            const double2 *vb = s_vec + (t & 1) * 16;
            double2 *vn = s_vec + ((t + 1) & 1) * 16;
            while (lds_load_word(flag_addr) != t)
                __builtin_amdgcn_s_sleep(1);
            double acc[2] = {0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {                          // y[c] = sum_j P[c][j] v[j], j = 4 r + g
                const double2 v = vb[4 * r + g];
                acc[0] = fma(opa.v[r][0], v.x, acc[0]);
                acc[0] = fma(-opa.v[r][1], v.y, acc[0]);
                acc[1] = fma(opa.v[r][0], v.y, acc[1]);
                acc[1] = fma(opa.v[r][1], v.x, acc[1]);
            }
            const double2 rec = vb[c];                             // v_t per column: the record of slice t
            col_sum_n(acc);
            if (g == 0)
                vn[c] = make_double2(acc[0], acc[1]);
            __builtin_amdgcn_s_waitcnt(0xc07f);                    // this wave's reads of v_t and its write of v_{t+1} are done
            asm volatile("" ::: "memory");
            lds_store_word(flag_addr, t + 1);
            if (g == 0) {
                V[(size_t)t * 16 + c] = (d2v){rec.x, rec.y};
                if (t == p.N - 1)
                    V[(size_t)p.N * 16 + c] = (d2v){acc[0], acc[1]};
            }
        }
        form_generator(tn);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NT = 2 (n = 17..32): FOUR waves share one propagator, wave (I, J) owns tile (I, J) of every matrix of the polynomial.
//
// Round 2's kernel gave a wave the whole 2 x 2-tile matrix: ~450 live registers, 164 of them on the AGPR side of the
// file where vector instructions cannot reach them -- 1 350 of its 2 255 vector instructions per slice only moved
// data between the two halves (profiles/r02_C5_E4096_pmc.json).  Here a wave keeps 16 registers per matrix, and
// the operands of a product travel through LDS, where the layout conversion had to go anyway:
//   * every factor is written ONCE, by the owners of its tiles, as the padded image of tile.hpp plus a third plane
//     holding re + im (the operand sums of the three-product complex multiplication are formed once per element
//     instead of once per consumer);
//   * read with the transposing pattern an image tile is an A operand, read back as written it is a B operand;
//   * wave (I, J) multiplies  sum_Kt X(I, Kt) Y(Kt, J):  Kt = I first, with its OWN tile of Y from registers, then
//     Kt = 1 - I with Y(1 - I, J) from the image -- 24 matrix-core instructions per product and wave.
// Two workgroup barriers per product: one behind the image writes, one behind the operand reads (all operands are in
// registers before the first MFMA), so the compute phases of the four waves -- and of the second workgroup on the
// CU -- are free to drift apart.  One slice at a time per workgroup; A'_k's tile stays in registers.
constexpr int kImg2Plane = kTileImage * 16;                        // bytes of the interleaved (re, im) plane of one tile
constexpr int kImg2Tile = kImg2Plane + kTileImage * 8 + 8;         // + the sum plane; 16-byte aligned
constexpr int kImg2Matrix = 4 * kImg2Tile;

struct Tile3 {                                                     // one tile of a factor in D layout + its operand sums
    d4 re, im, sm;
};
struct AOp3 {                                                      // A operand of one tile: (re, im) per k-block + the sums
    d2v v[4];
    d2v s01, s23;
};
struct BOp3 {                                                      // B operand of one tile read back from an image
    d2v v[4];
    d2v s01, s23;
};

// tile (re, im, sum) -> image.  `plane`: LDS byte address of the tile's (re, im) plane + this lane's write offset
GRAPE_DEV void img2_write(unsigned plane, unsigned sums, const Tile3 &t)
{
    // rows 0, 1 pairwise (re and im land in adjacent words); rows 2, 3 lie beyond the 8-bit offsets of ds_write2 and go
    // word by word with 16-bit byte offsets: no second base address, no address arithmetic in the slice loop
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" : : "v"(plane), "v"(t.re[0]), "v"(t.im[0]) : "memory");
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:136 offset1:137" : : "v"(plane), "v"(t.re[1]), "v"(t.im[1]) : "memory");
    asm volatile("ds_write_b64 %0, %1 offset:2176" : : "v"(plane), "v"(t.re[2]) : "memory");
    asm volatile("ds_write_b64 %0, %1 offset:2184" : : "v"(plane), "v"(t.im[2]) : "memory");
    asm volatile("ds_write_b64 %0, %1 offset:3264" : : "v"(plane), "v"(t.re[3]) : "memory");
    asm volatile("ds_write_b64 %0, %1 offset:3272" : : "v"(plane), "v"(t.im[3]) : "memory");
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:68" : : "v"(sums), "v"(t.sm[0]), "v"(t.sm[1]) : "memory");
    asm volatile("ds_write2_b64 %0, %1, %2 offset0:136 offset1:204" : : "v"(sums), "v"(t.sm[2]), "v"(t.sm[3]) : "memory");
}

// no wait inside: the caller drains all reads of a product together (img2_wait)
GRAPE_DEV void img2_read_a(AOp3 &a, unsigned plane_rd, unsigned sums_rd)
{
    asm volatile("ds_read_b128 %0, %1" : "=v"(a.v[0]) : "v"(plane_rd) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(a.v[1]) : "v"(plane_rd) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(a.v[2]) : "v"(plane_rd) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:192" : "=v"(a.v[3]) : "v"(plane_rd) : "memory");
    asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:4" : "=v"(a.s01) : "v"(sums_rd) : "memory");
    asm volatile("ds_read2_b64 %0, %1 offset0:8 offset1:12" : "=v"(a.s23) : "v"(sums_rd) : "memory");
}
GRAPE_DEV void img2_read_b(BOp3 &b, unsigned plane_wr, unsigned sums_wr)
{
    asm volatile("ds_read_b128 %0, %1" : "=v"(b.v[0]) : "v"(plane_wr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:1088" : "=v"(b.v[1]) : "v"(plane_wr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:2176" : "=v"(b.v[2]) : "v"(plane_wr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:3264" : "=v"(b.v[3]) : "v"(plane_wr) : "memory");
    asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:68" : "=v"(b.s01) : "v"(sums_wr) : "memory");
    asm volatile("ds_read2_b64 %0, %1 offset0:136 offset1:204" : "=v"(b.s23) : "v"(sums_wr) : "memory");
}
// every register the reads above fill is an in/out operand of the wait: no consumer can be scheduled in front of it
GRAPE_DEV void img2_wait(AOp3 &a0, AOp3 &a1, BOp3 &b)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a0.v[0]), "+v"(a0.v[1]), "+v"(a0.v[2]), "+v"(a0.v[3]), "+v"(a0.s01), "+v"(a0.s23), "+v"(a1.v[0]),
                   "+v"(a1.v[1]), "+v"(a1.v[2]), "+v"(a1.v[3]), "+v"(a1.s01), "+v"(a1.s23), "+v"(b.v[0]), "+v"(b.v[1]),
                   "+v"(b.v[2]), "+v"(b.v[3]), "+v"(b.s01), "+v"(b.s23)
                 :
                 : "memory");
}
GRAPE_DEV void lds_drain_and_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// (ore, oim) = X(I, I) own(I, J) + X(I, 1 - I) Y(1 - I, J): a0 = X(I, I), a1 = X(I, 1 - I) as A operands, `own` this wave's
// tile of Y (registers), b = Y(1 - I, J) read back from the image
GRAPE_DEV void tile2_prod(d4 &ore, d4 &oim, const AOp3 &a0, const AOp3 &a1, const Tile3 &own, const BOp3 &b)
{
    __builtin_amdgcn_sched_barrier(0);                             // vector work stays in bursts outside the MFMA chains
    d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        t1 = GRAPE_MFMA(a0.v[kb][0], own.re[kb], t1);
        t2 = GRAPE_MFMA(a0.v[kb][1], own.im[kb], t2);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        t1 = GRAPE_MFMA(a1.v[kb][0], b.v[kb][0], t1);
        t2 = GRAPE_MFMA(a1.v[kb][1], b.v[kb][1], t2);
    }
    __builtin_amdgcn_sched_barrier(0);
    d4 t3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ore[r] = t1[r] - t2[r];
        t3[r] = -t1[r] - t2[r];
    }
    __builtin_amdgcn_sched_barrier(0);
    t3 = GRAPE_MFMA(a0.s01[0], own.sm[0], t3);
    t3 = GRAPE_MFMA(a0.s01[1], own.sm[1], t3);
    t3 = GRAPE_MFMA(a0.s23[0], own.sm[2], t3);
    t3 = GRAPE_MFMA(a0.s23[1], own.sm[3], t3);
    t3 = GRAPE_MFMA(a1.s01[0], b.s01[0], t3);
    t3 = GRAPE_MFMA(a1.s01[1], b.s01[1], t3);
    t3 = GRAPE_MFMA(a1.s23[0], b.s23[0], t3);
    t3 = GRAPE_MFMA(a1.s23[1], b.s23[1], t3);
    __builtin_amdgcn_sched_barrier(0);
    oim = t3;
}


template <bool HOIST>
__global__ __launch_bounds__(256, 3) void prop_hoist2_kernel(const TileParams p)
{
    constexpr int TSZ = 1024;
    extern __shared__ double2 s_hoist[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int I = wave >> 1, J = wave & 1, tile = wave;            // own tile (I, J) = dump tile I * 2 + J
    const int k = blockIdx.y, z = blockIdx.z;
    const unsigned lds0 = (unsigned)(size_t)s_hoist;
    const unsigned wr_off = 16u * (17 * (lane >> 4) + (lane & 15));
    const unsigned rd_off = 16u * (68 * ((lane & 15) >> 2) + 17 * (lane & 3) + (lane >> 4));
    // image X (A-operand role; both roles for G G and the squarings) and image Y (B-operand role)
    const unsigned imgX = lds0, imgY = lds0 + kImg2Matrix;
    auto plane_of = [&](unsigned img, int t) { return img + (unsigned)t * kImg2Tile; };
    const unsigned own_x_wr = plane_of(imgX, tile) + wr_off, own_x_sm = plane_of(imgX, tile) + kImg2Plane + wr_off / 2;
    const unsigned own_y_wr = plane_of(imgY, tile) + wr_off, own_y_sm = plane_of(imgY, tile) + kImg2Plane + wr_off / 2;
    const unsigned a0_rd = plane_of(imgX, I * 2 + I) + rd_off, a0_sm = plane_of(imgX, I * 2 + I) + kImg2Plane + rd_off / 2;
    const unsigned a1_rd = plane_of(imgX, I * 2 + (1 - I)) + rd_off, a1_sm = plane_of(imgX, I * 2 + (1 - I)) + kImg2Plane + rd_off / 2;
    const int tb = (1 - I) * 2 + J;                                // Y(1 - I, J)
    const unsigned bx_wr = plane_of(imgX, tb) + wr_off, bx_sm = plane_of(imgX, tb) + kImg2Plane + wr_off / 2;
    const unsigned by_wr = plane_of(imgY, tb) + wr_off, by_sm = plane_of(imgY, tb) + kImg2Plane + wr_off / 2;

    unsigned long long dmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        dmask[r] = (I == J) ? __builtin_amdgcn_ballot_w64(4 * r + (lane >> 4) == (lane & 15)) : 0ull;
    // A'_k's tile is re-read from L2 with every slice's control sum (16 registers kept free: three waves per SIMD)
    const gcptr ha = uniform_global(p.ha + (size_t)k * TSZ + tile * 256);
    const int K = p.K;
    const int nstride = HOIST ? 1 : 1 + K;
    const double nA = p.ha_norm[(size_t)k * nstride];
    const double sk = (HOIST && p.ctrl_scale) ? p.ctrl_scale[k] : 1.0;      // B_k = s_k B_0: G = A'_k + s_k Gc_t
    const double *__restrict__ gcn = p.gcn + (size_t)z * p.N;
    const gcptr gc = uniform_global(p.gc + (size_t)z * p.N * TSZ + tile * 256);
    const gcptr Bk = uniform_global(p.ops + ((size_t)k * (2 * K + 3) + 1) * TSZ + tile * 256);   // !HOIST: own tile of B_1 .. B_K
    const double *__restrict__ xz = p.x + (size_t)z * K * p.N;
    const gptr props = uniform_global(p.props + ((size_t)z * p.E + k) * (size_t)p.N * TSZ + tile * 256);
    const int t_lo = blockIdx.x * p.prop_slices, t_hi = min(p.N, t_lo + p.prop_slices);
    // The loop is ROTATED: slice t + 1's generator tile G (and its norm bound) is formed at the END of slice t's body, from
    // loads issued before slice t's squarings.  Its s_waitcnt then sits in one place with one history -- eight loads, then
    // P_t's four stores -- and waits with vmcnt(4); formed at the top of the body (rounds 2-3) the same wait had to cover
    // the entry path too, whose loads are the LAST vector-memory operations, and waited with vmcnt(0): for the stores' acks.
    d2v gnext[4], anext[4];
    double bnext = 0.0;                                            // HOIST: the next slice's norm bound, FIRST of its prefetch
    Tile3 G;
    double bound = 0.0;
    auto prefetch = [&](int tn) {
        const gcptr g1 = gc + (size_t)tn * TSZ;
        if (HOIST) bnext = gcn[tn];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (HOIST) gnext[r] = g1[r * 64 + lane];
            anext[r] = ha[r * 64 + lane];
        }
    };
    auto form_generator = [&](int tn) {
        if (HOIST) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                G.re[r] = fma(sk, gnext[r][0], anext[r][0]);       // (s_k = 1: the plain sum, bit for bit)
                G.im[r] = fma(sk, gnext[r][1], anext[r][1]);
            }
            bound = fma(fabs(sk), bnext, nA);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                G.re[r] = anext[r][0];
                G.im[r] = anext[r][1];
            }
            bound = nA;
            for (int c = 0; c < K; ++c) {
                const double xv = xz[c + (size_t)tn * K], sx = p.dt * xv;
                bound = fma(fabs(xv), p.ha_norm[(size_t)k * nstride + 1 + c], bound);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const d2v b = Bk[(size_t)c * TSZ + r * 64 + lane];
                    G.re[r] = fma(sx, b[1], G.re[r]);
                    G.im[r] = fma(-sx, b[0], G.im[r]);
                }
            }
        }
    };
    if (t_lo < t_hi) {
        prefetch(t_lo);
        form_generator(t_lo);
    }
    for (int t = t_lo; t < t_hi; ++t) {
        rotate_priority();
        const int s = p.s_forced >= 0 ? p.s_forced : squarings_from_ratio(bound);
        if (s > 0) {
            const double sc = ldexp(1.0, -s);
            G.re *= sc;
            G.im *= sc;
        }
        G.sm = G.re + G.im;
        AOp3 a0, a1;
        BOp3 b;
        d4 A2re, A2im, A4re, A4im;
        // ---- A2 = G G: one image in both roles
        img2_write(own_x_wr, own_x_sm, G);
        lds_drain_and_barrier();
        img2_read_a(a0, a0_rd, a0_sm);
        img2_read_a(a1, a1_rd, a1_sm);
        img2_read_b(b, bx_wr, bx_sm);
        img2_wait(a0, a1, b);
        __builtin_amdgcn_s_barrier();
        tile2_prod(A2re, A2im, a0, a1, G, b);
        // ---- A4' = A2 (A2 + c1 G)
        Tile3 X, Y;
        X.re = A2re;
        X.im = A2im;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Y.re[r] = fma(kC1, G.re[r], A2re[r]);
            Y.im[r] = fma(kC1, G.im[r], A2im[r]);
        }
        X.sm = X.re + X.im;
        Y.sm = Y.re + Y.im;
        __builtin_amdgcn_sched_barrier(0);                         // the raw MFMA result A2im has been read by the vector ALU above
        img2_write(own_x_wr, own_x_sm, X);
        img2_write(own_y_wr, own_y_sm, Y);
        lds_drain_and_barrier();
        img2_read_a(a0, a0_rd, a0_sm);
        img2_read_a(a1, a1_rd, a1_sm);
        img2_read_b(b, by_wr, by_sm);
        img2_wait(a0, a1, b);
        __builtin_amdgcn_s_barrier();
        tile2_prod(A4re, A4im, a0, a1, Y, b);
        // ---- A8' = (A4' + c3 A2) (x4 I + x5 G + x6 A2 + c7 A4')
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X.re[r] = fma(kC3, A2re[r], A4re[r]);
            X.im[r] = fma(kC3, A2im[r], A4im[r]);
            Y.re[r] = fma(kC7, A4re[r], fma(kX6, A2re[r], kX5 * G.re[r]));
            Y.im[r] = fma(kC7, A4im[r], fma(kX6, A2im[r], kX5 * G.im[r]));
            Y.re[r] = add_masked(Y.re[r], kX4, dmask[r]);
        }
        X.sm = X.re + X.im;
        Y.sm = Y.re + Y.im;
        __builtin_amdgcn_sched_barrier(0);
        img2_write(own_x_wr, own_x_sm, X);
        img2_write(own_y_wr, own_y_sm, Y);
        lds_drain_and_barrier();
        img2_read_a(a0, a0_rd, a0_sm);
        img2_read_a(a1, a1_rd, a1_sm);
        img2_read_b(b, by_wr, by_sm);
        img2_wait(a0, a1, b);
        __builtin_amdgcn_s_barrier();
        Tile3 P;
        tile2_prod(P.re, P.im, a0, a1, Y, b);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            P.re[r] = fma(kX2, P.re[r], fma(kY2, A2re[r], G.re[r]));
            P.im[r] = fma(kX2, P.im[r], fma(kY2, A2im[r], G.im[r]));
            P.re[r] = add_masked(P.re[r], 1.0, dmask[r]);
        }
        // ---- next slice's control sum in flight BEFORE the squarings (G, A2 and A4' are dead by now: the 32 registers cost
        //      nothing; behind the squarings -- rounds 2-3 -- the loads were issued ~40 instructions ahead of their first use).
        //      The bound goes first: loads return in order.
        const int tn = min(t + 1, t_hi - 1);
        prefetch(tn);
        for (int i = 0; i < s; ++i) {                              // P <- P P
            P.sm = P.re + P.im;
            __builtin_amdgcn_sched_barrier(0);
            img2_write(own_x_wr, own_x_sm, P);
            lds_drain_and_barrier();
            img2_read_a(a0, a0_rd, a0_sm);
            img2_read_a(a1, a1_rd, a1_sm);
            img2_read_b(b, bx_wr, bx_sm);
            img2_wait(a0, a1, b);
            __builtin_amdgcn_s_barrier();
            Tile3 Q;
            tile2_prod(Q.re, Q.im, a0, a1, P, b);
            P.re = Q.re;
            P.im = Q.im;
        }
        // ---- P_t's tile packed through the wave's own region of image Y (nobody reads image Y before the next slice's
        //      second product) and stored
        if (s > 0)
            asm volatile("s_nop 15\n\ts_nop 2" ::: "memory");      // P.im is a raw MFMA result after a squaring
        {
            asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" : : "v"(own_y_wr), "v"(P.re[0]), "v"(P.im[0]) : "memory");
            asm volatile("ds_write2_b64 %0, %1, %2 offset0:136 offset1:137" : : "v"(own_y_wr), "v"(P.re[1]), "v"(P.im[1]) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:2176" : : "v"(own_y_wr), "v"(P.re[2]) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:2184" : : "v"(own_y_wr), "v"(P.im[2]) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:3264" : : "v"(own_y_wr), "v"(P.re[3]) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:3272" : : "v"(own_y_wr), "v"(P.im[3]) : "memory");
            AOp pk;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1088\n\tds_read_b128 %2, %4 offset:2176\n\t"
                         "ds_read_b128 %3, %4 offset:3264\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(pk.v[0]), "=&v"(pk.v[1]), "=&v"(pk.v[2]), "=&v"(pk.v[3])
                         : "v"(own_y_wr)
                         : "memory");
            const gptr dst = props + (size_t)t * TSZ;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[r * 64 + lane] = pk.v[r];
        }
        form_generator(tn);
    }
}

// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_ctrl_sum(int NT, const TileParams &p, hipStream_t stream)
{
    const dim3 grid(p.N, p.n_x), block(64 * NT * NT);
    switch (NT) {
    case 1: GRAPE_LAUNCH((ctrl_sum_kernel<1>), grid, block, 0, stream, p); break;
    case 2: GRAPE_LAUNCH((ctrl_sum_kernel<2>), grid, block, 0, stream, p); break;
    case 3: GRAPE_LAUNCH((ctrl_sum_kernel<3>), grid, block, 0, stream, p); break;      // (sweep_grid.hip, n = 33..64)
    default: GRAPE_LAUNCH((ctrl_sum_kernel<4>), grid, block, 0, stream, p); break;
    }
    return hipGetLastError();
}

// q: the launcher's copy of the parameters with prop_slices / fuse_fwd decided (sweep_tile.hip: launch_nt)
// q.hoist: 1 = member-invariant controls (pre-pass + hoisted kernels), 2 = per-member controls (same kernels, H build inside)
hipError_t launch_prop_hoist(int NT, const TileParams &q, hipStream_t stream)
{
    const bool hoisted = q.hoist == 1;
    if (hoisted) {
        hipError_t e = launch_ctrl_sum(NT, q, stream);
        if (e != hipSuccess)
            return e;
    }
    if (NT == 1) {
        const size_t lds = sizeof(double2) * (kHoistWaves * (size_t)kTileImage + 33);
        const dim3 grid(q.fuse_fwd ? 1 : (q.N + q.prop_slices - 1) / q.prop_slices, q.E, q.n_x), block(64 * kHoistWaves);
        if (q.fuse_fwd) {
            if (hoisted) GRAPE_LAUNCH((prop_hoist1_kernel<true, true>), grid, block, lds, stream, q);
            else         GRAPE_LAUNCH((prop_hoist1_kernel<true, false>), grid, block, lds, stream, q);
        } else {
            if (hoisted) GRAPE_LAUNCH((prop_hoist1_kernel<false, true>), grid, block, lds, stream, q);
            else         GRAPE_LAUNCH((prop_hoist1_kernel<false, false>), grid, block, lds, stream, q);
        }
        return hipGetLastError();
    }
    if (NT == 2) {
        const size_t lds = 2 * (size_t)kImg2Matrix;
        const int per = q.prop_slices;
        const dim3 grid((q.N + per - 1) / per, q.E, q.n_x), block(256);
        if (hoisted) GRAPE_LAUNCH(prop_hoist2_kernel<true>, grid, block, lds, stream, q);
        else         GRAPE_LAUNCH(prop_hoist2_kernel<false>, grid, block, lds, stream, q);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace grape
