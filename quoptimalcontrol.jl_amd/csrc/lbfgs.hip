// lbfgs.hip -- device-resident L-BFGS around the GRAPE evaluation.
//
// Stands in for  Optim.optimize(Optim.only_fg!(topt), x0, Optim.LBFGS(), opts)
// (/root/reference/src/solve.jl:138, :244) with every vector on the GPU: x, g, the m = 10 (s, y)
// pairs, the search direction and the trial points never visit the host; per iteration the host
// reads eight scalars (F, |g|_inf, the accepted step, ...) to decide convergence.
//
//   lbfgs_direction_kernel   two-loop recursion (Nocedal & Wright alg. 7.4, initial scaling
//                            gamma = s'y / y'y, Optim's scaleinvH0) -> d, g'd, and the B trial
//                            points x + alpha_j d of the line search, in ONE single-workgroup launch
//   [GRAPE sweep + reduce on the B trial points: one batched evaluation]
//   lbfgs_select_kernel      picks the largest trial step with sufficient decrease (Armijo, c1 = 1e-4)
//                            that also meets the strong Wolfe curvature condition (c2 = 0.9) if any
//                            does, updates x, g, pushes (s, y) when s'y > 0, publishes the scalars
//
// One workgroup of 1024 threads holds a K*N-vector in registers (up to kLbfgsMaxPer elements per
// thread): every dot product is a wave shuffle tree + one LDS round, fixed summation order.
#include "grape_kernels.hpp"

namespace grape {

constexpr int kLbfgsThreads = 1024;

struct BlockSum {
    double *s_part;          // 16 doubles
    __device__ __forceinline__ double operator()(double v) const
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            v += __shfl_xor(v, d, 64);
        __syncthreads();                             // s_part may still be read from the previous sum
        if ((threadIdx.x & 63) == 0)
            s_part[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kLbfgsThreads / 64; ++w)
            t += s_part[w];
        return t;
    }
};

__global__ __launch_bounds__(kLbfgsThreads) void lbfgs_direction_kernel(LbfgsState st, int B, double alpha0)
{
    __shared__ double s_part[kLbfgsThreads / 64];
    __shared__ double s_alpha[64];
    const BlockSum sum{s_part};
    const int KN = st.KN, m = st.m;
    const int n_hist = (int)st.sc[6], head = (int)st.sc[7];
    const double gamma = st.sc[3];
    double q[kLbfgsMaxPer], gg[kLbfgsMaxPer];
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i) {
        const int idx = threadIdx.x + i * kLbfgsThreads;
        gg[i] = idx < KN ? st.g[idx] : 0.0;
        q[i] = gg[i];
    }
    for (int h = 0; h < n_hist; ++h) {               // newest -> oldest
        const int j = (head - h + m) % m;
        const double *__restrict__ Sj = st.S + (size_t)j * KN, *__restrict__ Yj = st.Y + (size_t)j * KN;
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) part = fma(Sj[idx], q[i], part);
        }
        const double a = st.rho[j] * sum(part);
        if (threadIdx.x == 0) s_alpha[h] = a;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) q[i] = fma(-a, Yj[idx], q[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i)
        q[i] *= gamma;
    __syncthreads();
    for (int h = n_hist - 1; h >= 0; --h) {          // oldest -> newest
        const int j = (head - h + m) % m;
        const double *__restrict__ Sj = st.S + (size_t)j * KN, *__restrict__ Yj = st.Y + (size_t)j * KN;
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) part = fma(Yj[idx], q[i], part);
        }
        const double b = st.rho[j] * sum(part);
        const double a = s_alpha[h];
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) q[i] = fma(a - b, Sj[idx], q[i]);
        }
    }
    double part = 0.0;
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i)
        part = fma(-q[i], gg[i], part);
    double dg = sum(part);
    bool reset = !(dg < 0.0);                        // not a descent direction (or NaN): steepest descent, drop the history
    if (reset) {
        part = 0.0;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            q[i] = gg[i];
            part = fma(-gg[i], gg[i], part);
        }
        dg = sum(part);
    }
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i) {
        const int idx = threadIdx.x + i * kLbfgsThreads;
        if (idx < KN) {
            const double di = -q[i];
            st.d[idx] = di;
            const double xi = st.x[idx];
            for (int j = 0; j < B; ++j)
                st.xt[(size_t)j * KN + idx] = fma(ldexp(alpha0, -j), di, xi);     // alpha0, alpha0/2, alpha0/4, ...
        }
    }
    if (threadIdx.x < B)
        st.alphas[threadIdx.x] = ldexp(alpha0, -(int)threadIdx.x);
    if (threadIdx.x == 0) {
        st.sc[2] = dg;
        if (reset) st.sc[6] = 0.0;
    }
}

// mode 0: choose among the B probes (factor-2 ladder search); mode 1: PROBE -- publish phi = F and phi' = g.d of trial
// slot 0 in host_sc[8], host_sc[9] and change nothing (the host runs the HagerZhang logic on those two scalars);
// mode 2: COMMIT trial slot 0 unconditionally (the line search accepted it)
__global__ __launch_bounds__(kLbfgsThreads) void lbfgs_select_kernel(LbfgsState st, int B, int mode, DoneSignal done)
{
    __shared__ double s_part[kLbfgsThreads / 64];
    __shared__ int s_pick;
    const BlockSum sum{s_part};
    const int KN = st.KN, m = st.m, Q = KN + 1;
    const double F0 = st.sc[0], dg0 = st.sc[2];
    double dd[kLbfgsMaxPer];
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i) {
        const int idx = threadIdx.x + i * kLbfgsThreads;
        dd[i] = idx < KN ? st.d[idx] : 0.0;
    }
    // directional derivatives of the B trial points, then the choice (thread 0, broadcast through LDS)
    double dgj[kLbfgsMaxProbes];
    for (int j = 0; j < B; ++j) {
        const double *__restrict__ gj = st.fgt + (size_t)j * Q;
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) part = fma(gj[idx], dd[i], part);
        }
        dgj[j] = sum(part);
    }
    if (mode == 1) {
        if (threadIdx.x == 0) {
            st.host_sc[8] = st.fgt[KN];
            st.host_sc[9] = dgj[0];
            st.host_sc[10] = dg0;                                  // phi'(0) = g.d, left by the direction kernel
            if (done.flag) {
                __threadfence_system();
                __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (threadIdx.x == 0 && mode == 2)
        s_pick = 0;
    if (threadIdx.x == 0 && mode == 0) {
        int armijo = -1, wolfe = -1;
        for (int j = 0; j < B; ++j) {
            const double Fj = st.fgt[(size_t)j * Q + KN];
            const bool ok = (Fj == Fj) && fabs(Fj) < 1.0e300 && Fj <= F0 + st.c1 * st.alphas[j] * dg0;
            if (ok && armijo < 0) armijo = j;
            if (ok && wolfe < 0 && fabs(dgj[j]) <= st.c2 * fabs(dg0)) wolfe = j;
        }
        s_pick = wolfe >= 0 ? wolfe : armijo;
    }
    __syncthreads();
    const int pick = s_pick;
    double sc_out[8];
    if (pick < 0) {                                  // no acceptable step among the probes: the host shrinks and retries
        if (threadIdx.x == 0) st.sc[5] = 1.0;
    } else {
        const double alpha = st.alphas[pick];
        const double *__restrict__ gj = st.fgt + (size_t)pick * Q;
        double sy = 0.0, yy = 0.0, ss = 0.0, gmax = 0.0;
        double gn[kLbfgsMaxPer], yv[kLbfgsMaxPer];
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            gn[i] = idx < KN ? gj[idx] : 0.0;
            yv[i] = idx < KN ? gn[i] - st.g[idx] : 0.0;
            const double si = alpha * dd[i];
            sy = fma(si, yv[i], sy);
            yy = fma(yv[i], yv[i], yy);
            ss = fma(si, si, ss);
            gmax = fmax(gmax, fabs(gn[i]));
        }
        sy = sum(sy);
        yy = sum(yy);
        ss = sum(ss);
        // |g|_inf: max through the same tree (all values >= 0: exact in any order)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            gmax = fmax(gmax, __shfl_xor(gmax, d, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = gmax;
        __syncthreads();
        gmax = 0.0;
#pragma unroll
        for (int w = 0; w < kLbfgsThreads / 64; ++w)
            gmax = fmax(gmax, s_part[w]);
        const bool push = sy > 1e-10 * sqrt(ss * yy) && yy > 0.0;
        int n_hist = (int)st.sc[6], head = (int)st.sc[7];
        if (push) {
            head = n_hist == 0 ? 0 : (head + 1) % m;
            n_hist = n_hist < m ? n_hist + 1 : m;
        }
        double *__restrict__ Sj = st.S + (size_t)head * KN, *__restrict__ Yj = st.Y + (size_t)head * KN;
        const double *__restrict__ xj = st.xt + (size_t)pick * KN;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) {
                if (push) {
                    Sj[idx] = alpha * dd[i];
                    Yj[idx] = yv[i];
                }
                st.x[idx] = xj[idx];
                st.g[idx] = gn[i];
            }
        }
        if (threadIdx.x == 0) {
            if (push) {
                st.rho[head] = 1.0 / sy;
                st.sc[3] = sy / yy;                  // gamma
            }
            st.sc[0] = gj[KN];                       // F
            st.sc[1] = gmax;
            st.sc[4] = alpha;
            st.sc[5] = 0.0;
            st.sc[6] = (double)n_hist;
            st.sc[7] = (double)head;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc_out[i] = st.sc[i];
            st.host_sc[i] = sc_out[i];
        }
        if (done.flag) {
            __threadfence_system();
            __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// first evaluation: fg0 = [g (KN), F] -> g, F, |g|_inf; empty history
__global__ __launch_bounds__(kLbfgsThreads) void lbfgs_init_kernel(LbfgsState st, DoneSignal done)
{
    __shared__ double s_part[kLbfgsThreads / 64];
    const int KN = st.KN;
    double gmax = 0.0;
    for (int idx = threadIdx.x; idx < KN; idx += kLbfgsThreads) {
        const double v = st.fgt[idx];
        st.g[idx] = v;
        gmax = fmax(gmax, fabs(v));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        gmax = fmax(gmax, __shfl_xor(gmax, d, 64));
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        gmax = 0.0;
        for (int w = 0; w < kLbfgsThreads / 64; ++w)
            gmax = fmax(gmax, s_part[w]);
        st.sc[0] = st.fgt[KN];
        st.sc[1] = gmax;
        st.sc[2] = 0.0;
        st.sc[3] = 1.0;
        st.sc[4] = 0.0;
        st.sc[5] = 0.0;
        st.sc[6] = 0.0;
        st.sc[7] = 0.0;
        for (int i = 0; i < 8; ++i)
            st.host_sc[i] = st.sc[i];
        if (done.flag) {
            __threadfence_system();
            __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One wave does an accepted step AND the next direction (Hager-Zhang path): commit trial slot 0 (x, g, the (s, y) pair,
// rho, gamma), two-loop recursion, d, phi'(0) = g.d and the next trial point x + d -- one launch, no workgroup barrier.
// The block-wide version above pays ~1 us per reduction (wave tree + LDS + two barriers), 2 n_hist + 4 of them per
// iteration; a K*N-vector is 16 KB at C3: one wave walks it in 31 strides and a dot product is a DPP/shuffle tree.
// commit == 0: direction only (the first iteration).  The working vector q lives in LDS (K*N <= 16384 doubles).
__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(64) void lbfgs_step_kernel(LbfgsState st, int commit, DoneSignal done)
{
    extern __shared__ double s_q[];
    const int KN = st.KN, m = st.m, lane = threadIdx.x;
    int n_hist = (int)st.sc[6], head = (int)st.sc[7];
    double gamma = st.sc[3];
    if (commit) {
        const double alpha = st.alphas[0];
        const double *__restrict__ gj = st.fgt;
        double sy = 0.0, yy = 0.0, ss = 0.0, gmax = 0.0;
        for (int idx = lane; idx < KN; idx += 64) {
            const double gn = gj[idx], y = gn - st.g[idx], si = alpha * st.d[idx];
            sy = fma(si, y, sy);
            yy = fma(y, y, yy);
            ss = fma(si, si, ss);
            gmax = fmax(gmax, fabs(gn));
        }
        sy = wave_sum64(sy);
        yy = wave_sum64(yy);
        ss = wave_sum64(ss);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            gmax = fmax(gmax, __shfl_xor(gmax, d, 64));
        const bool push = sy > 1e-10 * sqrt(ss * yy) && yy > 0.0;
        if (push) {
            head = n_hist == 0 ? 0 : (head + 1) % m;
            n_hist = n_hist < m ? n_hist + 1 : m;
            gamma = sy / yy;
        }
        double *__restrict__ Sj = st.S + (size_t)head * KN, *__restrict__ Yj = st.Y + (size_t)head * KN;
        for (int idx = lane; idx < KN; idx += 64) {
            const double gn = gj[idx];
            if (push) {
                Sj[idx] = alpha * st.d[idx];
                Yj[idx] = gn - st.g[idx];
            }
            st.x[idx] = st.xt[idx];
            st.g[idx] = gn;
        }
        if (lane == 0) {
            if (push) {
                st.rho[head] = 1.0 / sy;
                st.sc[3] = gamma;
            }
            st.sc[0] = gj[KN];
            st.sc[1] = gmax;
            st.sc[4] = alpha;
            st.sc[5] = 0.0;
            st.sc[6] = (double)n_hist;
            st.sc[7] = (double)head;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // this wave re-reads S, Y, g, rho below
        __builtin_amdgcn_wave_barrier();
    }
    // ---- direction: two-loop recursion on q = g (Nocedal & Wright alg. 7.4), newest -> oldest -> newest
    for (int idx = lane; idx < KN; idx += 64)
        s_q[idx] = st.g[idx];
    double *a_h = s_q + KN;                          // the recursion's alpha_h (m <= 64)
    for (int h = 0; h < n_hist; ++h) {
        const int j = (head - h + m) % m;
        const double *__restrict__ Sj = st.S + (size_t)j * KN, *__restrict__ Yj = st.Y + (size_t)j * KN;
        double part = 0.0;
        for (int idx = lane; idx < KN; idx += 64)
            part = fma(Sj[idx], s_q[idx], part);
        const double a = st.rho[j] * wave_sum64(part);
        if (lane == 0) a_h[h] = a;
        for (int idx = lane; idx < KN; idx += 64)
            s_q[idx] = fma(-a, Yj[idx], s_q[idx]);
    }
    for (int idx = lane; idx < KN; idx += 64)
        s_q[idx] *= gamma;
    for (int h = n_hist - 1; h >= 0; --h) {
        const int j = (head - h + m) % m;
        const double *__restrict__ Sj = st.S + (size_t)j * KN, *__restrict__ Yj = st.Y + (size_t)j * KN;
        double part = 0.0;
        for (int idx = lane; idx < KN; idx += 64)
            part = fma(Yj[idx], s_q[idx], part);
        const double b = st.rho[j] * wave_sum64(part);
        const double ab = a_h[h] - b;
        for (int idx = lane; idx < KN; idx += 64)
            s_q[idx] = fma(ab, Sj[idx], s_q[idx]);
    }
    double part = 0.0;
    for (int idx = lane; idx < KN; idx += 64)
        part = fma(-s_q[idx], st.g[idx], part);
    double dg = wave_sum64(part);
    const bool reset = !(dg < 0.0);                  // not a descent direction (or NaN): steepest descent, drop the history
    if (reset) {
        part = 0.0;
        for (int idx = lane; idx < KN; idx += 64) {
            const double gv = st.g[idx];
            s_q[idx] = gv;
            part = fma(-gv, gv, part);
        }
        dg = wave_sum64(part);
    }
    for (int idx = lane; idx < KN; idx += 64) {
        const double di = -s_q[idx];
        st.d[idx] = di;
        st.xt[idx] = st.x[idx] + di;                 // the Hager-Zhang search starts at alpha = 1 (InitialStatic)
    }
    if (lane == 0) {
        st.alphas[0] = 1.0;
        st.sc[2] = dg;
        if (reset) st.sc[6] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            st.host_sc[i] = st.sc[i];
        if (done.flag) {
            __threadfence_system();
            __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The same fused step for long control arrays (K*N > 512): the 1024-thread workgroup of the kernels above, vectors in
// registers, block-wide reductions -- a single wave would walk 16 KB vectors forty times at one L2 round trip each.
__global__ __launch_bounds__(kLbfgsThreads) void lbfgs_step_block_kernel(LbfgsState st, int commit, DoneSignal done)
{
    __shared__ double s_part[kLbfgsThreads / 64];
    __shared__ double s_alpha[64];
    const BlockSum sum{s_part};
    const int KN = st.KN, m = st.m;
    int n_hist = (int)st.sc[6], head = (int)st.sc[7];
    double gamma = st.sc[3];
    double q[kLbfgsMaxPer], gg[kLbfgsMaxPer];
    if (commit) {
        const double alpha = st.alphas[0];
        const double *__restrict__ gj = st.fgt;
        double sy = 0.0, yy = 0.0, ss = 0.0, gmax = 0.0;
        double yv[kLbfgsMaxPer], sv[kLbfgsMaxPer];
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            gg[i] = idx < KN ? gj[idx] : 0.0;
            yv[i] = idx < KN ? gg[i] - st.g[idx] : 0.0;
            sv[i] = idx < KN ? alpha * st.d[idx] : 0.0;
            sy = fma(sv[i], yv[i], sy);
            yy = fma(yv[i], yv[i], yy);
            ss = fma(sv[i], sv[i], ss);
            gmax = fmax(gmax, fabs(gg[i]));
        }
        sy = sum(sy);
        yy = sum(yy);
        ss = sum(ss);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            gmax = fmax(gmax, __shfl_xor(gmax, d, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = gmax;
        __syncthreads();
        gmax = 0.0;
#pragma unroll
        for (int w = 0; w < kLbfgsThreads / 64; ++w)
            gmax = fmax(gmax, s_part[w]);
        const bool push = sy > 1e-10 * sqrt(ss * yy) && yy > 0.0;
        if (push) {
            head = n_hist == 0 ? 0 : (head + 1) % m;
            n_hist = n_hist < m ? n_hist + 1 : m;
            gamma = sy / yy;
        }
        double *__restrict__ Sj = st.S + (size_t)head * KN, *__restrict__ Yj = st.Y + (size_t)head * KN;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) {
                if (push) {
                    Sj[idx] = sv[i];
                    Yj[idx] = yv[i];
                }
                st.x[idx] = st.xt[idx];
                st.g[idx] = gg[i];
            }
        }
        if (threadIdx.x == 0) {
            if (push) {
                st.rho[head] = 1.0 / sy;
                st.sc[3] = gamma;
            }
            st.sc[0] = gj[KN];
            st.sc[1] = gmax;
            st.sc[4] = alpha;
            st.sc[5] = 0.0;
            st.sc[6] = (double)n_hist;
            st.sc[7] = (double)head;
        }
        __threadfence_block();                       // the history row and rho written above are read back below
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            gg[i] = idx < KN ? st.g[idx] : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i)
        q[i] = gg[i];
    for (int h = 0; h < n_hist; ++h) {               // newest -> oldest
        const int j = (head - h + m) % m;
        const double *__restrict__ Sj = st.S + (size_t)j * KN, *__restrict__ Yj = st.Y + (size_t)j * KN;
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) part = fma(Sj[idx], q[i], part);
        }
        const double a = st.rho[j] * sum(part);
        if (threadIdx.x == 0) s_alpha[h] = a;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) q[i] = fma(-a, Yj[idx], q[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i)
        q[i] *= gamma;
    __syncthreads();
    for (int h = n_hist - 1; h >= 0; --h) {          // oldest -> newest
        const int j = (head - h + m) % m;
        const double *__restrict__ Sj = st.S + (size_t)j * KN, *__restrict__ Yj = st.Y + (size_t)j * KN;
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) part = fma(Yj[idx], q[i], part);
        }
        const double b = st.rho[j] * sum(part);
        const double a = s_alpha[h];
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            const int idx = threadIdx.x + i * kLbfgsThreads;
            if (idx < KN) q[i] = fma(a - b, Sj[idx], q[i]);
        }
    }
    double part = 0.0;
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i)
        part = fma(-q[i], gg[i], part);
    double dg = sum(part);
    const bool reset = !(dg < 0.0);
    if (reset) {
        part = 0.0;
#pragma unroll
        for (int i = 0; i < kLbfgsMaxPer; ++i) {
            q[i] = gg[i];
            part = fma(-gg[i], gg[i], part);
        }
        dg = sum(part);
    }
#pragma unroll
    for (int i = 0; i < kLbfgsMaxPer; ++i) {
        const int idx = threadIdx.x + i * kLbfgsThreads;
        if (idx < KN) {
            const double di = -q[i];
            st.d[idx] = di;
            st.xt[idx] = st.x[idx] + di;
        }
    }
    if (threadIdx.x == 0) {
        st.alphas[0] = 1.0;
        st.sc[2] = dg;
        if (reset) st.sc[6] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            st.host_sc[i] = st.sc[i];
        if (done.flag) {
            __threadfence_system();
            __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// K*N <= 2048, m <= 10 (C3: 2000, Optim's default memory): 512 threads with the WHOLE history in registers -- every
// (s, y) row is fetched at kernel start in logical order newest -> oldest (one L2 round trip for all 20 vectors instead of
// one per pass of the recursion: 80 doubles per thread, no AGPRs), the two-loop recursion then runs on registers with
// eight-wave reductions (one barrier each).  A committed pair is used from the registers it was formed in (it is the
// newest row of the recursion; the oldest stored row is skipped when the buffer was full): nothing is shifted.
constexpr int kRegThreads = 512, kRegPer = 4, kRegM = 10, kRegWaves = kRegThreads / 64;
struct BlockSumReg {
    double *s_part;          // 2 x kRegWaves doubles (ping-pong: one barrier per sum)
    int flip = 0;
    __device__ __forceinline__ double operator()(double v)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            v += __shfl_xor(v, d, 64);
        double *buf = s_part + kRegWaves * flip;
        flip ^= 1;
        if ((threadIdx.x & 63) == 0)
            buf[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kRegWaves; ++w)
            t += buf[w];
        return t;
    }
};

__global__ __launch_bounds__(kRegThreads, 2) void lbfgs_step_reg_kernel(LbfgsState st, int commit, DoneSignal done)
{
    __shared__ double s_part[2 * kRegWaves];
    BlockSumReg sum{s_part};
    const int KN = st.KN, m = st.m;
    const int n_old = (int)st.sc[6], head_old = (int)st.sc[7];
    double gamma = st.sc[3];
    double q[kRegPer], gg[kRegPer];
    double Sh[kRegM][kRegPer], Yh[kRegM][kRegPer], rho[kRegM];
    // the stored history (before a commit): logical row h = physical row (head - h) mod m
#pragma unroll
    for (int h = 0; h < kRegM; ++h) {
        const bool have = h < n_old;
        const int j = have ? (head_old - h + m) % m : 0;
        rho[h] = have ? st.rho[j] : 0.0;
#pragma unroll
        for (int i = 0; i < kRegPer; ++i) {
            const int idx = threadIdx.x + i * kRegThreads;
            const bool ok = have && idx < KN;
            Sh[h][i] = ok ? st.S[(size_t)j * KN + idx] : 0.0;
            Yh[h][i] = ok ? st.Y[(size_t)j * KN + idx] : 0.0;
        }
    }
    double yv[kRegPer], sv[kRegPer], rho_new = 0.0;
    bool push = false;
    int n_rows = n_old;                              // stored rows that take part in the recursion
    if (commit) {
        const double alpha = st.alphas[0];
        const double *__restrict__ gj = st.fgt;
        double sy = 0.0, yy = 0.0, ss = 0.0, gmax = 0.0;
#pragma unroll
        for (int i = 0; i < kRegPer; ++i) {
            const int idx = threadIdx.x + i * kRegThreads;
            gg[i] = idx < KN ? gj[idx] : 0.0;
            yv[i] = idx < KN ? gg[i] - st.g[idx] : 0.0;
            sv[i] = idx < KN ? alpha * st.d[idx] : 0.0;
            sy = fma(sv[i], yv[i], sy);
            yy = fma(yv[i], yv[i], yy);
            ss = fma(sv[i], sv[i], ss);
            gmax = fmax(gmax, fabs(gg[i]));
        }
        sy = sum(sy);
        yy = sum(yy);
        ss = sum(ss);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            gmax = fmax(gmax, __shfl_xor(gmax, d, 64));
        {
            double *buf = s_part + kRegWaves * sum.flip;
            sum.flip ^= 1;
            if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = gmax;
            __syncthreads();
            gmax = 0.0;
#pragma unroll
            for (int w = 0; w < kRegWaves; ++w)
                gmax = fmax(gmax, buf[w]);
        }
        push = sy > 1e-10 * sqrt(ss * yy) && yy > 0.0;
        int head = head_old, n_hist = n_old;
        if (push) {
            head = n_old == 0 ? 0 : (head_old + 1) % m;
            n_hist = n_old < m ? n_old + 1 : m;
            gamma = sy / yy;
            rho_new = 1.0 / sy;
            n_rows = n_old < m ? n_old : m - 1;      // a full buffer drops its oldest row
        }
        double *__restrict__ Sj = st.S + (size_t)head * KN, *__restrict__ Yj = st.Y + (size_t)head * KN;
#pragma unroll
        for (int i = 0; i < kRegPer; ++i) {
            const int idx = threadIdx.x + i * kRegThreads;
            if (idx < KN) {
                if (push) {
                    Sj[idx] = sv[i];
                    Yj[idx] = yv[i];
                }
                st.x[idx] = st.xt[idx];
                st.g[idx] = gg[i];
            }
        }
        if (threadIdx.x == 0) {
            if (push) {
                st.rho[head] = rho_new;
                st.sc[3] = gamma;
            }
            st.sc[0] = gj[KN];
            st.sc[1] = gmax;
            st.sc[4] = alpha;
            st.sc[5] = 0.0;
            st.sc[6] = (double)n_hist;
            st.sc[7] = (double)head;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kRegPer; ++i) {
            const int idx = threadIdx.x + i * kRegThreads;
            gg[i] = idx < KN ? st.g[idx] : 0.0;
            sv[i] = 0.0;
            yv[i] = 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < kRegPer; ++i)
        q[i] = gg[i];
    // newest -> oldest: the pair just committed first, then the stored rows (rows beyond n_rows: no-ops)
    double a_new = 0.0;
    if (push) {
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < kRegPer; ++i)
            part = fma(sv[i], q[i], part);
        a_new = rho_new * sum(part);
#pragma unroll
        for (int i = 0; i < kRegPer; ++i)
            q[i] = fma(-a_new, yv[i], q[i]);
    }
    double a_h[kRegM];
#pragma unroll
    for (int h = 0; h < kRegM; ++h) {
        a_h[h] = 0.0;
        if (h < n_rows) {                            // (uniform: every thread takes the same branch, the barrier inside is safe)
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < kRegPer; ++i)
                part = fma(Sh[h][i], q[i], part);
            a_h[h] = rho[h] * sum(part);
#pragma unroll
            for (int i = 0; i < kRegPer; ++i)
                q[i] = fma(-a_h[h], Yh[h][i], q[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < kRegPer; ++i)
        q[i] *= gamma;
#pragma unroll
    for (int h = kRegM - 1; h >= 0; --h) {           // oldest -> newest
        if (h < n_rows) {
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < kRegPer; ++i)
                part = fma(Yh[h][i], q[i], part);
            const double b = rho[h] * sum(part);
#pragma unroll
            for (int i = 0; i < kRegPer; ++i)
                q[i] = fma(a_h[h] - b, Sh[h][i], q[i]);
        }
    }
    if (push) {
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < kRegPer; ++i)
            part = fma(yv[i], q[i], part);
        const double b = rho_new * sum(part);
#pragma unroll
        for (int i = 0; i < kRegPer; ++i)
            q[i] = fma(a_new - b, sv[i], q[i]);
    }
    double part = 0.0;
#pragma unroll
    for (int i = 0; i < kRegPer; ++i)
        part = fma(-q[i], gg[i], part);
    double dg = sum(part);
    const bool reset = !(dg < 0.0);
    if (reset) {
        part = 0.0;
#pragma unroll
        for (int i = 0; i < kRegPer; ++i) {
            q[i] = gg[i];
            part = fma(-gg[i], gg[i], part);
        }
        dg = sum(part);
    }
#pragma unroll
    for (int i = 0; i < kRegPer; ++i) {
        const int idx = threadIdx.x + i * kRegThreads;
        if (idx < KN) {
            const double di = -q[i];
            st.d[idx] = di;
            st.xt[idx] = (commit ? st.xt[idx] : st.x[idx]) + di;      // (after a commit x == the accepted trial point)
        }
    }
    if (threadIdx.x == 0) {
        st.alphas[0] = 1.0;
        st.sc[2] = dg;
        if (reset) st.sc[6] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            st.host_sc[i] = st.sc[i];
        if (done.flag) {
            __threadfence_system();
            __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_lbfgs_step(const LbfgsState &st, int commit, hipStream_t stream, DoneSignal done)
{
    if (st.KN > 512 && st.KN <= kRegThreads * kRegPer && st.m <= kRegM)
        GRAPE_LAUNCH(lbfgs_step_reg_kernel, dim3(1), dim3(kRegThreads), 0, stream, st, commit, done);
    else if (st.KN <= 512)                           // short control arrays: one wave, no workgroup barrier
        GRAPE_LAUNCH(lbfgs_step_kernel, dim3(1), dim3(64), sizeof(double) * ((size_t)st.KN + 64), stream, st, commit, done);
    else
        GRAPE_LAUNCH(lbfgs_step_block_kernel, dim3(1), dim3(kLbfgsThreads), 0, stream, st, commit, done);
    return hipGetLastError();
}

hipError_t launch_lbfgs_init(const LbfgsState &st, hipStream_t stream, DoneSignal done)
{
    GRAPE_LAUNCH(lbfgs_init_kernel, dim3(1), dim3(kLbfgsThreads), 0, stream, st, done);
    return hipGetLastError();
}

hipError_t launch_lbfgs_direction(const LbfgsState &st, int B, double alpha0, hipStream_t stream)
{
    GRAPE_LAUNCH(lbfgs_direction_kernel, dim3(1), dim3(kLbfgsThreads), 0, stream, st, B, alpha0);
    return hipGetLastError();
}

hipError_t launch_lbfgs_select(const LbfgsState &st, int B, hipStream_t stream, DoneSignal done, int mode)
{
    GRAPE_LAUNCH(lbfgs_select_kernel, dim3(1), dim3(kLbfgsThreads), 0, stream, st, B, mode, done);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the accepted step on MANY workgroups (K N up to 16384, m <= 10).  lbfgs_step_reg_kernel spends 21 us per
// iteration at C3 (rocprofv3, tools/lbfgs_time.py): ONE workgroup pulls the 320 KB of history another workgroup wrote
// through one CU, and the two-loop recursion's 2 m + 4 reductions run one after the other.  Here
//   * lbfgs_dots_kernel runs behind EVERY trial evaluation, off the critical path (the host is reading phi, phi' over PCIe
//     meanwhile): sixteen workgroups form every dot product the step can need -- {g_t, g, d} x {s_j, y_j, d, g}, g_t.g_t,
//     max |g_t| -- as sixteen rows of partial sums;
//   * lbfgs_step_mb_kernel commits: every workgroup adds the sixteen rows (fixed order), runs the recursion ON SCALARS --
//     s_r.q and y_r.r are linear combinations of s_j.g, y_j.g and the Gram matrices S'Y, Y'Y, which grow by the committed
//     pair's row and column (s.y_j = alpha d.y_j, s_j.y = s_j.g_t - s_j.g, y.y_j = y_j.g_t - y_j.g: all direct dot
//     products of vectors in memory) -- and writes its slice of s, y, x, g, d = -(gamma (g - sum a_r y_r) + sum c_r s_r)
//     and x + d.  No reduction, no hand-off between workgroups; phi'(0) = g.d comes out of the same scalars.
// A pair pushed by lbfgs_select_kernel (the ladder an unbracketed search falls back to) has no Gram row: the host goes back
// to the single-workgroup kernels for the rest of that run.
constexpr int kMbVecs = 2 * kLbfgsMbM + 2;          // s_0 .. s_9, y_0 .. y_9, d, g
constexpr int kMbSums = 3 * kMbVecs + 3;            // {g_t, g, d} x vectors, then g_t.g_t, y.y, d.y with y = g_t - g formed per
                                                    // element (69; slot 69: max |g_t|) -- y.y and s.y = alpha d.y from differences
                                                    // of the big dot products lose eps |g|^2 / |y|^2 on short steps
constexpr int kMbThreads = 256;

// v[0..63] per lane -> lane l holds the wave's sum of v[l]: pairwise halving, 32 + 16 + ... + 1 shuffles; one function per
// level keeps every register index a compile-time constant
template <int HALF>
__device__ __forceinline__ void mb_reduce_scatter(double (&v)[kMbSums], int lane)
{
    const bool up = (lane & HALF) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const double send = up ? v[i] : v[i + HALF];
        const double keep = up ? v[i + HALF] : v[i];
        v[i] = keep + __shfl_xor(send, HALF, 64);
    }
}

__global__ __launch_bounds__(kMbThreads) void lbfgs_dots_kernel(LbfgsState st)
{
    __shared__ double s_red[kMbThreads / 64][kLbfgsDotStride];
    const int KN = st.KN, m = st.m, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (KN + kLbfgsDotBlocks - 1) / kLbfgsDotBlocks;
    const int lo = blockIdx.x * per, hi = min(KN, lo + per);
    double acc[kMbSums], amax = 0.0;
#pragma unroll
    for (int k = 0; k < kMbSums; ++k)
        acc[k] = 0.0;
    for (int e = lo + threadIdx.x; e < hi; e += kMbThreads) {
        double vec[kMbVecs];
#pragma unroll
        for (int j = 0; j < kLbfgsMbM; ++j) {
            const size_t at = (size_t)(j < m ? j : 0) * KN + e;
            vec[j] = st.S[at];
            vec[kLbfgsMbM + j] = st.Y[at];
        }
        const double gt = st.fgt[e], g = st.g[e], d = st.d[e];
        vec[2 * kLbfgsMbM] = d;
        vec[2 * kLbfgsMbM + 1] = g;
#pragma unroll
        for (int v = 0; v < kMbVecs; ++v) {
            acc[v] = fma(gt, vec[v], acc[v]);
            acc[kMbVecs + v] = fma(g, vec[v], acc[kMbVecs + v]);
            acc[2 * kMbVecs + v] = fma(d, vec[v], acc[2 * kMbVecs + v]);
        }
        acc[3 * kMbVecs] = fma(gt, gt, acc[3 * kMbVecs]);
        const double yv = gt - g;
        acc[3 * kMbVecs + 1] = fma(yv, yv, acc[3 * kMbVecs + 1]);
        acc[3 * kMbVecs + 2] = fma(d, yv, acc[3 * kMbVecs + 2]);
        amax = fmax(amax, fabs(gt));
    }
    // wave sums: a reduce-scatter for the first 64 (lane l ends up with sum l: 63 shuffles, no branch between them -- one
    // butterfly per value with its `if (lane == 0)` store each ran the 67 shuffle chains one behind the other: 19 us),
    // butterflies for the last three and the maximum
    {
        double tail[kMbSums - 64 + 1];
#pragma unroll
        for (int k = 64; k < kMbSums; ++k)
            tail[k - 64] = acc[k];
        tail[kMbSums - 64] = amax;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
#pragma unroll
            for (int k = 0; k < kMbSums - 64; ++k)
                tail[k] += __shfl_xor(tail[k], dd, 64);
            tail[kMbSums - 64] = fmax(tail[kMbSums - 64], __shfl_xor(tail[kMbSums - 64], dd, 64));
        }
        mb_reduce_scatter<32>(acc, lane);
        mb_reduce_scatter<16>(acc, lane);
        mb_reduce_scatter<8>(acc, lane);
        mb_reduce_scatter<4>(acc, lane);
        mb_reduce_scatter<2>(acc, lane);
        mb_reduce_scatter<1>(acc, lane);
        s_red[wave][lane] = acc[0];
#pragma unroll
        for (int k = 64; k <= kMbSums; ++k)
            s_red[wave][k] = tail[k - 64];           // (every lane holds the same value)
    }
    __syncthreads();
    if (threadIdx.x <= kMbSums) {
        double t = s_red[0][threadIdx.x];
        for (int w = 1; w < kMbThreads / 64; ++w)
            t = threadIdx.x == kMbSums ? fmax(t, s_red[w][threadIdx.x]) : t + s_red[w][threadIdx.x];
        st.dots[(size_t)blockIdx.x * kLbfgsDotStride + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(kMbThreads) void lbfgs_step_mb_kernel(LbfgsState st, double alpha, DoneSignal done)
{
    constexpr int M = kLbfgsMbM, LR = M + 1;
    __shared__ double s_tot[kLbfgsDotStride];
    __shared__ double s_SY[M][M], s_YY[M][M], s_rho[M];
    __shared__ double s_coef[2 * LR + 8];            // a_r, c_r by LOGICAL row (0: the committed pair), then the scalars below
    const int KN = st.KN, m = st.m, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- one batch of loads: nothing here waits for another load
    const double sc_gamma = st.sc[3], sc_n = st.sc[6], sc_head = st.sc[7];
    const double F_trial = st.fgt[KN];                            // (alpha: the accepted step length, from the host -- this kernel
    const int e = blockIdx.x * kMbThreads + threadIdx.x;           //  writes nothing another of its workgroups may still read)
    const bool in = e < KN;
    const int ce = in ? e : 0;
    const double gt = st.fgt[ce], g_old = st.g[ce], d_old = st.d[ce], x_new = st.xt[ce];
    double Sr[M], Yr[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const size_t at = (size_t)(j < m ? j : 0) * KN + ce;
        Sr[j] = st.S[at];
        Yr[j] = st.Y[at];
    }
    if (threadIdx.x <= kMbSums) {
        double part[kLbfgsDotBlocks];
#pragma unroll
        for (int b = 0; b < kLbfgsDotBlocks; ++b)
            part[b] = st.dots[(size_t)b * kLbfgsDotStride + threadIdx.x];
        double t = part[0];
#pragma unroll
        for (int b = 1; b < kLbfgsDotBlocks; ++b)
            t = threadIdx.x == kMbSums ? fmax(t, part[b]) : t + part[b];
        s_tot[threadIdx.x] = t;
    }
    for (int q = threadIdx.x; q < 2 * m * m; q += kMbThreads) {
        const int which = q / (m * m), rem = q - which * m * m;
        (which ? s_YY : s_SY)[rem / m][rem % m] = st.gram[q];
    }
    if ((int)threadIdx.x < m)
        s_rho[threadIdx.x] = st.rho[threadIdx.x];
    __syncthreads();
    const int n_old = (int)sc_n, head_old = (int)sc_head;
    auto row_of = [&](int h) { const int j = head_old - h; return j < 0 ? j + m : j; };    // logical stored row h (0 = newest) -> physical
    // products of the three vectors with the history (physical rows), d and g
    auto P = [&](int a, int v) { return s_tot[a * kMbVecs + v]; };        // a: 0 g_t, 1 g, 2 d;  v: j | M + j | 2M (d) | 2M+1 (g)
    const double gtgt = s_tot[3 * kMbVecs];
    const double sy = alpha * s_tot[3 * kMbVecs + 2];           // s.y = alpha d.(g_t - g), y.y: summed from per-element differences
    const double yy = s_tot[3 * kMbVecs + 1];
    const double ss = alpha * alpha * P(2, 2 * M);
    const bool push = sy > 1e-10 * sqrt(ss * yy) && yy > 0.0;
    const int head = push ? (n_old == 0 ? 0 : (head_old + 1 == m ? 0 : head_old + 1)) : head_old;
    const int n_hist = push ? (n_old < m ? n_old + 1 : m) : n_old;
    const int n_rows = push ? (n_old < m ? n_old : m - 1) : n_old;        // stored rows in the recursion (a full buffer drops its oldest)
    const double gamma = push ? sy / yy : sc_gamma, rho_new = push ? 1.0 / sy : 0.0;
    const int off = push ? 1 : 0, R = n_rows + off;
    // the Gram entries, rho and the products with g_t by LOGICAL row (0: the committed pair, then the stored rows newest first;
    // zero beyond the R rows in use): one entry per thread, so that the recursion below reads LDS at constant indices
    __shared__ double s_LSY[LR][LR], s_LYY[LR][LR], s_lrho[LR], s_lsg[LR], s_lyg[LR];
    for (int q = threadIdx.x; q < LR * LR; q += kMbThreads) {
        const int r1 = q / LR, r2 = q - r1 * LR;
        double vsy = 0.0, vyy = 0.0;
        if (r1 < R && r2 < R) {
            if (r1 < off && r2 < off) {
                vsy = sy;
                vyy = yy;
            } else if (r1 < off) {
                const int j = row_of(r2 - off);
                vsy = alpha * P(2, M + j);                                 // s.y_j = alpha d.y_j
                vyy = P(0, M + j) - P(1, M + j);                           // y.y_j = y_j.g_t - y_j.g
            } else if (r2 < off) {
                const int j = row_of(r1 - off);
                vsy = P(0, j) - P(1, j);                                   // s_j.y = s_j.g_t - s_j.g
                vyy = P(0, M + j) - P(1, M + j);
            } else {
                vsy = s_SY[row_of(r1 - off)][row_of(r2 - off)];
                vyy = s_YY[row_of(r1 - off)][row_of(r2 - off)];
            }
        }
        s_LSY[r1][r2] = vsy;
        s_LYY[r1][r2] = vyy;
        if (r2 == 0) {
            const bool on = r1 < R, fresh = r1 < off;
            const int j = on && !fresh ? row_of(r1 - off) : 0;
            s_lrho[r1] = !on ? 0.0 : (fresh ? rho_new : s_rho[j]);
            s_lsg[r1] = !on ? 0.0 : (fresh ? alpha * P(0, 2 * M) : P(0, j));                 // s_r . g_t
            s_lyg[r1] = !on ? 0.0 : (fresh ? gtgt - P(0, 2 * M + 1) : P(0, M + j));          // y_r . g_t
        }
    }
    __syncthreads();
    if (wave == 0) {
        double a_r[LR], c_r[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) {               // newest -> oldest: a_r = rho_r s_r.q_r,  s_r.q_r = s_r.g - sum_{i<r} a_i s_r.y_i
            double t = s_lsg[r];
#pragma unroll
            for (int i = 0; i < r; ++i)
                t = fma(-a_r[i], s_LSY[r][i], t);
            a_r[r] = s_lrho[r] * t;                  // (rows beyond R: rho = 0)
        }
#pragma unroll
        for (int r = LR - 1; r >= 0; --r) {          // oldest -> newest: b_r = rho_r y_r.r,  c_r = a_r - b_r
            double t = s_lyg[r];
#pragma unroll
            for (int i = 0; i < LR; ++i)
                t = fma(-a_r[i], s_LYY[r][i], t);                  // y_r.q_n,  q_n = g - sum a_i y_i
            t *= gamma;
#pragma unroll
            for (int i = LR - 1; i > r; --i)
                t = fma(c_r[i], s_LSY[i][r], t);                   // + sum_{i older} c_i s_i.y_r
            c_r[r] = a_r[r] - s_lrho[r] * t;                       // (zero beyond R: a_r = 0, rho = 0)
        }
        // phi'(0) of the next search: g.d = -(gamma (g.g - sum a_r y_r.g) + sum c_r s_r.g)
        double w = gtgt, u = 0.0;
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            w = fma(-a_r[r], s_lyg[r], w);
            u = fma(c_r[r], s_lsg[r], u);
        }
        double dg = -fma(gamma, w, u);
        const bool reset = !(dg < 0.0);              // not a descent direction (or NaN): steepest descent, the history is dropped
        if (reset) dg = -gtgt;
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < LR; ++r) {
                s_coef[r] = reset ? 0.0 : a_r[r];
                s_coef[LR + r] = reset ? 0.0 : c_r[r];
            }
            s_coef[2 * LR] = dg;
            s_coef[2 * LR + 1] = reset ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    const double dg = s_coef[2 * LR];
    const bool reset = s_coef[2 * LR + 1] != 0.0;
    // ---- this workgroup's elements
    if (in) {
        const double s_e = alpha * d_old, y_e = gt - g_old;
        double wv = gt, uv = 0.0;
        if (push) {
            wv = fma(-s_coef[0], y_e, wv);
            uv = s_coef[LR] * s_e;
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int hq = head_old - (j < m ? j : 0), h = hq < 0 ? hq + m : hq;    // physical row j -> logical stored row
            const bool on = j < m && h < n_rows && !(push && j == head);
            const double ar = on ? s_coef[h + off] : 0.0, cr = on ? s_coef[LR + h + off] : 0.0;
            wv = fma(-ar, on ? Yr[j] : 0.0, wv);
            uv = fma(cr, on ? Sr[j] : 0.0, uv);
        }
        const double q = reset ? gt : fma(gamma, wv, uv);
        if (push) {
            st.S[(size_t)head * KN + e] = s_e;
            st.Y[(size_t)head * KN + e] = y_e;
        }
        st.x[e] = x_new;
        st.g[e] = gt;
        st.d[e] = -q;
        st.xt[e] = x_new - q;
    }
    if (blockIdx.x == 0) {
        // the committed pair's row and column of the Gram matrices (entries against rows that have left the history are never
        // read again), rho, the scalars of the new iterate -- into sc_out: the other workgroups may still be reading sc
        if (push && (int)threadIdx.x < m) {
            const int j = threadIdx.x, hq = head_old - j, h = hq < 0 ? hq + m : hq;
            double *GSY = st.gram, *GYY = st.gram + (size_t)m * m;
            if (j == head) {
                GSY[(size_t)head * m + head] = sy;
                GYY[(size_t)head * m + head] = yy;
                st.rho[head] = rho_new;
            } else if (h < n_rows) {
                GSY[(size_t)head * m + j] = alpha * P(2, M + j);
                GSY[(size_t)j * m + head] = P(0, j) - P(1, j);
                GYY[(size_t)head * m + j] = P(0, M + j) - P(1, M + j);
                GYY[(size_t)j * m + head] = P(0, M + j) - P(1, M + j);
            }
        }
        if (threadIdx.x == 64) {
            double o[8];
            o[0] = F_trial;
            o[1] = s_tot[kMbSums];                   // |g|_inf of the new iterate
            o[2] = dg;
            o[3] = gamma;
            o[4] = alpha;
            o[5] = 0.0;
            o[6] = reset ? 0.0 : (double)n_hist;
            o[7] = (double)head;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                st.sc_out[i] = o[i];
                st.host_sc[i] = o[i];
            }
            if (done.flag) {
                __threadfence_system();
                __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

hipError_t launch_lbfgs_dots(const LbfgsState &st, hipStream_t stream)
{
    GRAPE_LAUNCH(lbfgs_dots_kernel, dim3(kLbfgsDotBlocks), dim3(kMbThreads), 0, stream, st);
    return hipGetLastError();
}

hipError_t launch_lbfgs_step_mb(const LbfgsState &st, double alpha, hipStream_t stream, DoneSignal done)
{
    GRAPE_LAUNCH(lbfgs_step_mb_kernel, dim3((st.KN + kMbThreads - 1) / kMbThreads), dim3(kMbThreads), 0, stream, st, alpha, done);
    return hipGetLastError();
}

// trial slot 0 <- x + alpha d
__global__ __launch_bounds__(kLbfgsThreads) void lbfgs_trial_kernel(LbfgsState st, double alpha)
{
    for (int idx = threadIdx.x; idx < st.KN; idx += kLbfgsThreads)
        st.xt[idx] = fma(alpha, st.d[idx], st.x[idx]);
    if (threadIdx.x == 0)
        st.alphas[0] = alpha;
}

hipError_t launch_lbfgs_trial(const LbfgsState &st, double alpha, hipStream_t stream)
{
    GRAPE_LAUNCH(lbfgs_trial_kernel, dim3(1), dim3(kLbfgsThreads), 0, stream, st, alpha);
    return hipGetLastError();
}

}  // namespace grape
