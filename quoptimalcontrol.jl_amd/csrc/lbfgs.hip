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
