// sweep_small.hip -- the GRAPE hot path for small operators (n = 2, 3, 4) on gfx950.
//
// One workgroup per ensemble member, W wavefronts per workgroup; every lane owns S consecutive
// time slices and whole n x n ComplexF64 matrices in VGPRs (cmat.hpp).  The time axis, serial
// in the reference (src/GRAPE.jl:53-75), is cut into 64*W lane chunks and stitched with a
// wavefront-shuffle prefix/suffix product scan, so all lanes work for the whole kernel:
//
//   phase A  pw_prop_save!  (src/timeevolution.jl:98-110): H_t = sum_j B_j x[j,t] + A,
//            P_t = exp(-i dt H_t) (expm_t8), P_t -> HBM, chunk product Q = P_last ... P_first
//   phase B  scan over lanes/waves: U_L = Q_{L-1}...Q_0 (exclusive prefix), V_L = Q_last...Q_{L+1}
//            (exclusive suffix); X at chunk start = U Xi [U'], L at chunk end = V' Xt [V]
//   phase C  evolve_func! forward (src/GRAPE.jl:226 / :245-246) inside the chunk, X_t -> HBM
//   phase D  evolve_func! backward (:228 / :248-249) fused with grad_func! (:261-287) and
//            fom_func (src/cost_functions.jl:99-111): the costates never leave registers;
//            tr(L' B_c X) = sum_ij B_c[i,j] (X L')[j,i]  (the trace_matmul identity,
//            src/grape_tools.jl:66-68), so one product per slice serves all K controls.
//
// HBM traffic = "model S" of BASELINE.md: P_t and X_t make one round trip, 64 n^2 N bytes per
// member, written and read with lane-contiguous 16-byte accesses (1 KiB per wave instruction).
#include "cmat.hpp"
#include "grape_kernels.hpp"

namespace grape {

template <int N>
GRAPE_DEV void load_uniform(CMat<N> &m, const double2 *__restrict__ src)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        const double2 v = src[e];
        m.re[e] = v.x;
        m.im[e] = v.y;
    }
}

template <int N>
GRAPE_DEV void store_ws(double2 *__restrict__ base, size_t stride, const CMat<N> &m)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e)
        base[e * stride] = make_double2(m.re[e], m.im[e]);
}

template <int N>
GRAPE_DEV void load_ws(CMat<N> &m, const double2 *__restrict__ base, size_t stride)
{
#pragma unroll
    for (int e = 0; e < N * N; ++e) {
        const double2 v = base[e * stride];
        m.re[e] = v.x;
        m.im[e] = v.y;
    }
}

// diagnostic phase stamps (GRAPE_FLAG_PHASE_STAMPS): lane 0 of every wave stores the shader
// clock at the phase boundaries into a buffer nothing else reads.  p.stamps is NULL in
// production, so no stamp executes there.
GRAPE_DEV void stamp(unsigned long long *__restrict__ st, int slot)
{
    if (st) {
        const unsigned long long t = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0)
            st[slot] = t;
    }
}

template <int N, int SAND, bool KEEPL, int MAXT>
__global__ __launch_bounds__(MAXT) void sweep_small_kernel(const SweepParams p)
{
    constexpr int NN = N * N;
    constexpr int MAXW = MAXT / 64;
    __shared__ double2 s_tot[2][MAXW][NN];

    const int k = blockIdx.x;
    const int L = threadIdx.x;
    const int lane = L & 63, wave = L >> 6;
    const int LT = p.LT, W = LT >> 6;
    const int K = p.K, Nsl = p.N, S = p.S;
    const size_t stride = (size_t)LT;

    const double2 *__restrict__ ops = p.ops + (size_t)k * (K + 3) * NN;
    const double2 *__restrict__ opB = ops + NN;
    const double2 *__restrict__ opXi = ops + (size_t)(1 + K) * NN;
    const double2 *__restrict__ opXt = opXi + NN;
    const size_t wbase = (size_t)k * S * NN * stride + L;
    double2 *__restrict__ Pw = p.props + wbase;
    double2 *__restrict__ Xw = p.states + wbase;
    double *__restrict__ out = p.member_out + (size_t)k * ((size_t)K * Nsl + 1);
    const int t0 = L * S;
    const double dt = p.dt;
    unsigned long long *__restrict__ st =
        p.stamps ? p.stamps + ((size_t)k * W + wave) * kStampSlots : nullptr;
    if (st && lane == 0)
        st[5] = __builtin_amdgcn_s_memrealtime();
    stamp(st, 0);

    // ---------------------------------------------------------------- phase A
    CMat<N> Q;
    set_identity(Q);
    for (int j = 0; j < S; ++j) {
        const int t = t0 + j;
        if (t < Nsl) {
            CMat<N> G, P;
            if (p.variant == 0) {
#pragma unroll
                for (int e = 0; e < NN; ++e) { G.re[e] = 0.0; G.im[e] = 0.0; }
            } else {
                load_uniform(G, ops);
            }
            for (int c = 0; c < K; ++c) {
                const double xv = p.x[c + (size_t)t * K];
#pragma unroll
                for (int e = 0; e < NN; ++e) {
                    const double2 b = opB[c * NN + e];
                    G.re[e] = fma(b.x, xv, G.re[e]);
                    G.im[e] = fma(b.y, xv, G.im[e]);
                }
            }
            if (p.variant == 0) {
#pragma unroll
                for (int e = 0; e < NN; ++e) {
                    const double2 a = ops[e];
                    G.re[e] += a.x;
                    G.im[e] += a.y;
                }
            }
#pragma unroll
            for (int e = 0; e < NN; ++e) {          // (-i dt) * H
                const double hr = G.re[e], hi = G.im[e];
                G.re[e] = dt * hi;
                G.im[e] = -dt * hr;
            }
            expm_t8(P, G, p.s_forced);
            store_ws(Pw + (size_t)j * NN * stride, stride, P);
            if (j == 0) {
                Q = P;
            } else {
                mul(G, P, Q);
                Q = G;
            }
        }
    }

    stamp(st, 1);
    // ---------------------------------------------------------------- phase B
    CMat<N> Xs, Le;
    {
        CMat<N> inc = Q, oth, tmp;
        for (int d = 1; d < 64; d <<= 1) {
            shfl_up(oth, inc, d);
            if (lane >= d) {
                mul(tmp, inc, oth);
                inc = tmp;
            }
        }
        shfl_up(oth, inc, 1);
        if (lane == 0)
            set_identity(oth);
        if (W > 1) {
            if (lane == 63) {
#pragma unroll
                for (int e = 0; e < NN; ++e)
                    s_tot[0][wave][e] = make_double2(inc.re[e], inc.im[e]);
            }
            __syncthreads();
            CMat<N> pre;
            set_identity(pre);
            for (int w = 0; w < wave; ++w) {
                load_uniform(inc, &s_tot[0][w][0]);
                mul(tmp, inc, pre);
                pre = tmp;
            }
            mul(tmp, oth, pre);
            oth = tmp;
        }
        load_uniform(inc, opXi);
        if (SAND) {
            mul(tmp, oth, inc);
            mul_a_bh(Xs, tmp, oth);
        } else {
            mul(Xs, oth, inc);
        }
    }
    {
        CMat<N> inc = Q, oth, tmp;
        for (int d = 1; d < 64; d <<= 1) {
            shfl_down(oth, inc, d);
            if (lane + d < 64) {
                mul(tmp, oth, inc);
                inc = tmp;
            }
        }
        shfl_down(oth, inc, 1);
        if (lane == 63)
            set_identity(oth);
        if (W > 1) {
            if (lane == 0) {
#pragma unroll
                for (int e = 0; e < NN; ++e)
                    s_tot[1][wave][e] = make_double2(inc.re[e], inc.im[e]);
            }
            __syncthreads();
            CMat<N> post;
            set_identity(post);
            for (int w = wave + 1; w < W; ++w) {
                load_uniform(inc, &s_tot[1][w][0]);
                mul(tmp, inc, post);
                post = tmp;
            }
            mul(tmp, post, oth);
            oth = tmp;
        }
        load_uniform(inc, opXt);
        if (SAND) {
            mul_ah_b(tmp, oth, inc);
            mul(Le, tmp, oth);
        } else {
            mul_ah_b(Le, oth, inc);
        }
    }

    stamp(st, 2);
    // ---------------------------------------------------------------- phase C
    {
        CMat<N> X = Xs, P, tmp;
        for (int j = 0; j < S; ++j) {
            const int t = t0 + j;
            if (t < Nsl) {
                store_ws(Xw + (size_t)j * NN * stride, stride, X);
                if (j + 1 < S) {                       // the chunk's last state is never read
                    load_ws(P, Pw + (size_t)j * NN * stride, stride);
                    if (SAND) {
                        mul_a_bh(tmp, X, P);
                        mul(X, P, tmp);
                    } else {
                        mul(tmp, P, X);
                        X = tmp;
                    }
                }
            }
        }
    }

    stamp(st, 3);
    // ---------------------------------------------------------------- phase D
    {
        CMat<N> Lc = Le, P, X, M, tmp;
        const double gs = SAND ? -dt : (p.variant == 0 ? -2.0 * dt : 2.0 * dt);
        for (int j = S - 1; j >= 0; --j) {
            const int t = t0 + j;
            if (t < Nsl) {
                load_ws(P, Pw + (size_t)j * NN * stride, stride);
                load_ws(X, Xw + (size_t)j * NN * stride, stride);
                if (SAND) {
                    mul(tmp, Lc, P);
                    mul_ah_b(Lc, P, tmp);
                } else {
                    mul_ah_b(tmp, P, Lc);
                    Lc = tmp;
                }
                if (KEEPL)
                    store_ws(p.costates + wbase + (size_t)j * NN * stride, stride, Lc);
                double zr, zi;
                trace_ah_b(zr, zi, X, Lc);           // tr(X' L)
                mul_a_bh(M, X, Lc);                  // X L'
                if (SAND) {
                    mul_ah_b(tmp, Lc, X);            // L' X
#pragma unroll
                    for (int e = 0; e < NN; ++e) {
                        M.re[e] -= tmp.re[e];
                        M.im[e] -= tmp.im[e];
                    }
                }
                for (int c = 0; c < K; ++c) {
                    double wr = 0.0, wi = 0.0;
#pragma unroll
                    for (int jj = 0; jj < N; ++jj)
#pragma unroll
                        for (int ii = 0; ii < N; ++ii) {
                            const double2 b = opB[c * NN + ii + jj * N];
                            const double mr = M.re[jj + ii * N], mi = M.im[jj + ii * N];
                            wr = fma(b.x, mr, wr);
                            wr = fma(-b.y, mi, wr);
                            wi = fma(b.x, mi, wi);
                            wi = fma(b.y, mr, wi);
                        }
                    const double im = SAND ? wi : fma(wr, zi, wi * zr);   // Im(w) | Im(w z)
                    out[c + (size_t)t * K] = gs * im;
                }
                if (t == Nsl - 1) {
                    if (SAND) {
                        const double inv = 1.0 / (double)N;
                        const double ar = zr * inv, ai = zi * inv;
                        out[(size_t)K * Nsl] = 1.0 - (ar * ar + ai * ai);
                    } else {
                        out[(size_t)K * Nsl] = zr * zr - zi * zi;
                    }
                }
            }
        }
    }
    stamp(st, 4);
    if (st && lane == 0)
        st[6] = __builtin_amdgcn_s_memrealtime();
}

template <int N>
struct SmallTraits;
template <> struct SmallTraits<2> { static constexpr int MAXT = 1024; };
template <> struct SmallTraits<3> { static constexpr int MAXT = 512; };
template <> struct SmallTraits<4> { static constexpr int MAXT = 256; };

int sweep_small_max_waves(int n)
{
    switch (n) {
    case 2: return SmallTraits<2>::MAXT / 64;
    case 3: return SmallTraits<3>::MAXT / 64;
    case 4: return SmallTraits<4>::MAXT / 64;
    default: return 0;
    }
}

template <int N>
static hipError_t launch_n(int sandwich, bool keepl, const SweepParams &p, hipStream_t stream)
{
    constexpr int MAXT = SmallTraits<N>::MAXT;
    const dim3 grid(p.E), block(p.LT);
    if (p.LT > MAXT || (p.LT & 63) || (long long)p.S * p.LT < p.N)
        return hipErrorInvalidConfiguration;
    if (sandwich) {
        if (keepl) hipLaunchKernelGGL((sweep_small_kernel<N, 1, true, MAXT>), grid, block, 0, stream, p);
        else       hipLaunchKernelGGL((sweep_small_kernel<N, 1, false, MAXT>), grid, block, 0, stream, p);
    } else {
        if (keepl) hipLaunchKernelGGL((sweep_small_kernel<N, 0, true, MAXT>), grid, block, 0, stream, p);
        else       hipLaunchKernelGGL((sweep_small_kernel<N, 0, false, MAXT>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_sweep_small(int n, int sandwich, bool keep_costates, const SweepParams &p,
                              hipStream_t stream)
{
    switch (n) {
    case 2: return launch_n<2>(sandwich, keep_costates, p, stream);
    case 3: return launch_n<3>(sandwich, keep_costates, p, stream);
    case 4: return launch_n<4>(sandwich, keep_costates, p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace grape
