// done_signal.hpp -- device side of DoneSignal (grape_kernels.hpp): how the last kernel of an evaluation hands [G, F] to the
// host.  Used by the reduce kernels (reduce.hip) and by the forms kernels of action_thin.hip when they close a single
// problem's evaluation themselves.
#pragma once
#include "grape_kernels.hpp"

namespace grape {

// Host-visible completion without waiting for the kernel-end signal: the LAST workgroup of the final
// kernel of an evaluation stores the evaluation's sequence number into coherent pinned host memory,
// after every workgroup's result stores have been released at system scope.  Called by ONE thread per
// workgroup whose own wave made (or, after a workgroup barrier + per-thread fence, covers) the stores.
__device__ __forceinline__ void signal_done(DoneSignal d, unsigned nblocks)
{
    if (!d.flag)
        return;
    __threadfence_system();
    const unsigned prev = atomicAdd(d.counter, 1u);
    if (prev == nblocks - 1) {
        __hip_atomic_store(d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch: behind this kernel
        __threadfence_system();
        __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Result stores of the reduce kernels when the evaluation ends in host memory.  251 workgroups each
// pushing 64 bytes over PCIe and waiting for a system-scope fence cost ~10 us; instead every workgroup
// stores its outputs into a DEVICE staging buffer with sc1 (write-through) stores, drains them
// (s_waitcnt vmcnt(0)), and adds to an agent-scope counter; the workgroup whose add came last reads the
// whole staging buffer with sc1 loads (MI355X guide, inter-workgroup hand-off table, row 1), writes it to
// the mapped host buffer in one coalesced burst and publishes the sequence number.
__device__ __forceinline__ void stage_store(double *stage, int i, double v)
{
    __hip_atomic_store(stage + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // global_store ... sc1
}

__device__ __forceinline__ void publish_via_last_block(DoneSignal d, const double *stage, int n_total, unsigned nblocks)
{
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its own staging stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(d.counter, 1u);
        s_last = prev == nblocks - 1;
        if (s_last)
            __hip_atomic_store(d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last)
        return;
    if (d.probe_out) {
        // the line search's probe (grape_lbfgs): phi = F, phi' = G . dir -- in lbfgs_select_kernel's summation order (1024
        // partial sums idx = v, v + 1024, ...; a 64-lane xor tree per group of 64; the sixteen groups in order), so the
        // iterates do not depend on which kernel formed the scalars
        __shared__ double s_pw[16];
        const int KNp = n_total - 1, nw = (int)(blockDim.x >> 6) > 0 ? (int)(blockDim.x >> 6) : 1, l = (int)(threadIdx.x & 63);
        for (int vw = (int)(threadIdx.x >> 6); vw < 16; vw += nw) {
            // (all loads of a partial sum in flight together, then the multiply-adds in the kernel's order: one at a time
            // the sc1 loads cost a trip to L2 each -- 5 us more per evaluation than the separate kernel they replace)
            double gv[kLbfgsMaxPer], dv[kLbfgsMaxPer];
#pragma unroll
            for (int k = 0; k < kLbfgsMaxPer; ++k) {
                const int idx = vw * 64 + l + 1024 * k;
                gv[k] = idx < KNp ? __hip_atomic_load(stage + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                dv[k] = idx < KNp ? d.probe_dir[idx] : 0.0;
            }
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < kLbfgsMaxPer; ++k)
                if (vw * 64 + l + 1024 * k < KNp)
                    part = fma(gv[k], dv[k], part);
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1)
                part += __shfl_xor(part, dd, 64);
            if (l == 0)
                s_pw[vw] = part;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w)
                t += s_pw[w];
            d.probe_out[0] = __hip_atomic_load(stage + KNp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d.probe_out[1] = t;
            d.probe_out[2] = d.probe_sc[2];
            __threadfence_system();
            __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    for (int i0 = threadIdx.x; i0 < n_total; i0 += 8 * blockDim.x) {       // 8 sc1 loads in flight per thread
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            v[u] = i < n_total ? __hip_atomic_load(stage + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < n_total)
                d.host_out[i] = v[u];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(d.flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ONE problem (E = 1, one control array): the LAST kernel of the flow closes the evaluation (TileParams.fold_fg: the forms
// kernels of action_thin.hip, the unitary chain kernels of sweep_tile.hip / sweep_coop.hip) -- fg[q] = w_0 x member_out[q]
// exactly as reduce_few_kernel forms it (fma(value, w, 0)), written where the result is wanted (the mapped host buffer when
// there is one: a handful of workgroups with a few KB each -- the staging buffer + copy-out of the reduce kernels, built
// for hundreds of workgroups, cost the forms kernel 6 us), then the storing wave of every workgroup releases its stores at
// system scope and the last one publishes the sequence number.  No reduce launch.
__device__ __forceinline__ double *fold_dst(const TileParams &p)
{
    return p.fold_done.flag && p.fold_done.host_out ? p.fold_done.host_out : p.fold_fg;
}
__device__ __forceinline__ void fold_store(const TileParams &p, size_t q, double v)
{
    if (p.fold_fg)
        fold_dst(p)[q] = fma(v, p.fold_wts[0], 0.0);
}
// called by ONE thread of the wave that made the workgroup's fold stores, behind them
__device__ __forceinline__ void fold_publish(const TileParams &p)
{
    if (p.fold_fg)
        signal_done(p.fold_done, gridDim.x * gridDim.y * gridDim.z);
}

}  // namespace grape
