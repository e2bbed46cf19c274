// reduce.hip -- weighted ensemble reduction  F = sum_k w_k F_k,  G = sum_k w_k g_k
// (src/solve.jl:171-186 and :191 of the reference) over the per-member results the sweep
// kernel leaves in HBM.  Two small launches with a fixed summation tree, so results are
// bitwise reproducible run to run (the reference sums k = 1..E sequentially; the tree differs
// from it by O(sqrt(E) * 1e-16), well inside the 1e-10 parity budget).
//
// Pure HBM streaming: E*(K*N+1) doubles read once, lane-contiguous.
#include "grape_kernels.hpp"
#include "done_signal.hpp"

namespace grape {

constexpr int kTileQ = 64;    // outputs per block (one wave-width, coalesced)
constexpr int kSubK = 4;      // member sub-lanes per block

// stage 1: partial[ks][q] = sum over this split's members of w_k * member_out[k][q]
__global__ __launch_bounds__(kTileQ *kSubK) void reduce_stage1(const double *__restrict__ member_out,
                                                              const double *__restrict__ wts,
                                                              double *__restrict__ partial, int E,
                                                              int Q, int per_split)
{
    __shared__ double s_acc[kSubK][kTileQ];
    const int tq = threadIdx.x & (kTileQ - 1), tk = threadIdx.x / kTileQ;
    const int q = blockIdx.x * kTileQ + tq;
    const int ks = blockIdx.y;
    const int k_lo = ks * per_split;
    const int k_hi = min(E, k_lo + per_split);
    double acc = 0.0;
    if (q < Q) {
        for (int k = k_lo + tk; k < k_hi; k += kSubK)
            acc = fma(member_out[(size_t)k * Q + q], wts[k], acc);
    }
    s_acc[tk][tq] = acc;
    __syncthreads();
    if (tk == 0 && q < Q) {
        double s = s_acc[0][tq];
#pragma unroll
        for (int i = 1; i < kSubK; ++i)
            s += s_acc[i][tq];
        partial[(size_t)q * gridDim.y + ks] = s;      // [q][ks]: stage 2 reads it lane-contiguously
    }
}

// stage 2: fg[q] = sum_ks partial[q][ks]; 32 lanes per output, fixed xor-shuffle tree.
constexpr int kMaxSplit = 32;
__global__ __launch_bounds__(256) void reduce_stage2(const double *__restrict__ partial,
                                                     double *__restrict__ fg, int Q, int ksplit, DoneSignal done)
{
    const int ks = threadIdx.x & (kMaxSplit - 1);
    const int q = blockIdx.x * (256 / kMaxSplit) + threadIdx.x / kMaxSplit;
    double v = (q < Q && ks < ksplit) ? partial[(size_t)q * ksplit + ks] : 0.0;
#pragma unroll
    for (int d = kMaxSplit / 2; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    if (done.flag) {
        if (ks == 0 && q < Q)
            stage_store(fg, q, v);
        publish_via_last_block(done, done.stage_base ? done.stage_base : fg, done.stage_base ? done.n_total : Q, gridDim.x);
    } else if (ks == 0 && q < Q) {
        fg[q] = v;
    }
}

// single-stage reduction of NB already-weighted rows (the sweep kernel's workgroup partials):
// a block owns 8 consecutive outputs x 32 row-lanes; every thread walks its rows with stride 32,
// then the 32 partial sums of an output are added in a fixed order through LDS.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const double *__restrict__ rows,
                                                          double *__restrict__ fg, int NB, int Q, DoneSignal done)
{
    __shared__ double s_acc[32][9];
    const int ql = threadIdx.x & 7, bl = threadIdx.x >> 3;
    const int q = blockIdx.x * 8 + ql;
    rows += (size_t)blockIdx.y * NB * Q;             // batched evaluation: one reduction per control array
    fg += (size_t)blockIdx.y * Q;
    double acc = 0.0;
    if (q < Q) {
        // 8 independent loads in flight per thread (a plain `acc += load` loop serialises the round trips);
        // the additions keep the order b = bl, bl + 32, ...
        for (int b0 = bl; b0 < NB; b0 += 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 32 * u;
                v[u] = b < NB ? rows[(size_t)b * Q + q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc += v[u];
        }
    }
    s_acc[bl][ql] = acc;
    __syncthreads();
    if (bl == 0 && q < Q) {
        double s = s_acc[0][ql];
#pragma unroll
        for (int i = 1; i < 32; ++i)
            s += s_acc[i][ql];
        if (done.flag && (done.host_out || done.probe_out))
            stage_store(fg, q, s);
        else
            fg[q] = s;
    }
    if (done.flag && (done.host_out || done.probe_out)) // fg - blockIdx.y * Q: the staging buffer of all n_x reductions
        publish_via_last_block(done, done.stage_base ? done.stage_base : fg - (size_t)blockIdx.y * Q,
                               done.stage_base ? done.n_total : Q * (int)gridDim.y, gridDim.x * gridDim.y);
    else if (done.flag && threadIdx.x == 0)            // fg IS the mapped host buffer: every workgroup pushed its own 64 bytes
        signal_done(done, gridDim.x * gridDim.y);
}

// The same sums in the same order (32 row-lanes per output, rows b = bl, bl + 32, ... ascending, then the 32 partial sums in
// order: bitwise reduce_rows_kernel's results), 32 outputs per workgroup of 1024 threads -- and NO hand-off between
// workgroups: each writes its 256 bytes straight into the mapped host buffer, fences at system scope and stores the
// evaluation's sequence number into ITS OWN host flag; the host waits until every flag shows it.  reduce_rows_kernel's
// publication (staging stores drained, a 251 -> 1 fan-in on a counter, the last workgroup's sc1 loads + copy-out + fence)
// was four dependent trips to memory, 5.5 of its 8.7 us (profiles/r04_C3_phaseD.txt); this one has none of them.
constexpr int kMfOut = 32;
__global__ __launch_bounds__(1024) void reduce_rows_mf_kernel(const double *__restrict__ rows, double *__restrict__ fg, int NB,
                                                              int Q, DoneSignal done)
{
    __shared__ double s_acc[32][kMfOut + 1];
    const int ql = threadIdx.x & (kMfOut - 1), bl = threadIdx.x / kMfOut;
    const int q = blockIdx.x * kMfOut + ql;
    rows += (size_t)blockIdx.y * NB * Q;
    double acc = 0.0;
    if (q < Q) {
        for (int b0 = bl; b0 < NB; b0 += 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 32 * u;
                v[u] = b < NB ? rows[(size_t)b * Q + q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc += v[u];
        }
    }
    s_acc[bl][ql] = acc;
    __syncthreads();
    if (bl == 0) {                                   // lanes 0..31 of the first wave
        if (q < Q) {
            double s = s_acc[0][ql];
#pragma unroll
            for (int i = 1; i < 32; ++i)
                s += s_acc[i][ql];
            fg[(size_t)blockIdx.y * Q + q] = s;
            done.host_out[(size_t)blockIdx.y * Q + q] = s;
        }
        __threadfence_system();                      // the wave's host stores are visible before its flag
        if (ql == 0)
            __hip_atomic_store(done.mflags + blockIdx.y * gridDim.x + blockIdx.x, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int reduce_rows_mflags(int Q, int n_x)
{
    const long long g = (long long)((Q + kMfOut - 1) / kMfOut) * n_x;
    return g >= 1 && g <= kMaxMflags ? (int)g : 0;
}

hipError_t launch_reduce_rows(const double *rows, double *fg, int NB, int Q, int n_x, hipStream_t stream, DoneSignal done)
{
    if (done.flag && done.host_out && done.mflags && !done.probe_out && reduce_rows_mflags(Q, n_x)) {
        GRAPE_LAUNCH(reduce_rows_mf_kernel, dim3((Q + kMfOut - 1) / kMfOut, n_x), dim3(1024), 0, stream, rows, fg, NB, Q, done);
        return hipGetLastError();
    }
    GRAPE_LAUNCH(reduce_rows_kernel, dim3((Q + 7) / 8, n_x), dim3(256), 0, stream, rows, fg, NB, Q, done);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_kernel(const double *__restrict__ src, double *__restrict__ dst, int n,
                                                   DoneSignal done)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        dst[i] = src[i];
    if (done.flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            signal_done(done, gridDim.x);
    }
}

hipError_t launch_copy(const double *src, double *dst, int n, hipStream_t stream, DoneSignal done)
{
    GRAPE_LAUNCH(copy_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, dst, n, done);
    return hipGetLastError();
}

int reduce_ksplit(int E)
{
    // enough blocks to cover the chip (32 q-tiles at C3) without making stage 2 long
    int ks = (E + 31) / 32;
    if (ks > 32) ks = 32;
    if (ks < 1) ks = 1;
    return ks;
}

// few members (one split): fg[q] = sum_k w_k member_out[k][q] in member order, one output per thread, and the
// publication -- one launch instead of two (a single problem's evaluation is 0.1 ms; stage 2 alone took 26 us with
// 31 of its 32 lanes per output idle)
__global__ __launch_bounds__(256) void reduce_few_kernel(const double *__restrict__ member_out, const double *__restrict__ wts,
                                                         double *__restrict__ fg, int E, int Q, DoneSignal done)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    if (q < Q)
        for (int k = 0; k < E; ++k)
            acc = fma(member_out[(size_t)k * Q + q], wts[k], acc);
    if (done.flag) {
        if (q < Q)
            stage_store(fg, q, acc);
        publish_via_last_block(done, done.stage_base ? done.stage_base : fg, done.stage_base ? done.n_total : Q, gridDim.x);
    } else if (q < Q) {
        fg[q] = acc;
    }
}

// the shards of a multi-device context: fg[q] = sum over the shards' rows, in shard order, each row read where it was
// produced (the devices of a group have peer access to each other) -- no staging copies, one launch
__global__ __launch_bounds__(256) void reduce_shards_kernel(ShardRows rows, double *__restrict__ fg, int Q, DoneSignal done)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    if (q < Q) {
        double v[kMaxShards];
#pragma unroll
        for (int g = 0; g < kMaxShards; ++g)
            v[g] = g < rows.n ? rows.p[g][q] : 0.0;           // all loads in flight together
#pragma unroll
        for (int g = 0; g < kMaxShards; ++g)
            acc += v[g];
    }
    if (done.flag && done.host_out) {
        if (q < Q)
            stage_store(fg, q, acc);
        publish_via_last_block(done, fg, Q, gridDim.x);
    } else {
        if (q < Q)
            fg[q] = acc;
        if (done.flag) {
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0)
                signal_done(done, gridDim.x);
        }
    }
}

hipError_t launch_reduce_shards(const ShardRows &rows, double *fg, int Q, hipStream_t stream, DoneSignal done)
{
    GRAPE_LAUNCH(reduce_shards_kernel, dim3((Q + 255) / 256), dim3(256), 0, stream, rows, fg, Q, done);
    return hipGetLastError();
}

// ---- arrive-and-sum (in-process groups) ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shard_arrive_kernel(const ArriveParams p)
{
    __shared__ int s_last;
    const int q = blockIdx.x * 256 + threadIdx.x, G = p.rows.n;
    if (threadIdx.x == 0) {
        // (this shard's row: written by the previous kernel of this stream, written back at its end)
        __threadfence_system();
        const unsigned prev = __hip_atomic_fetch_add(p.arrive + blockIdx.x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
        s_last = prev == (unsigned)(G - 1);
        if (s_last)                                               // the next evaluation's arrivals come behind its publication
            __hip_atomic_store(p.arrive + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (!s_last)
        return;
    __threadfence_system();
    if (q < p.Q) {
        double v[kMaxShards];
#pragma unroll
        for (int g = 0; g < kMaxShards; ++g)                      // all loads in flight together; system scope: never a cached copy
            v[g] = g < G ? __hip_atomic_load(p.rows.p[g] + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int g = 0; g < kMaxShards; ++g)
            acc += v[g];                                          // shard order: reduce_shards_kernel's sum, bit for bit
        if (p.out)
            p.out[q] = acc;
        if (p.done.host_out)
            p.done.host_out[q] = acc;
    }
    if (!p.done.flag)
        return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *finished = p.finished;
        const unsigned prev = __hip_atomic_fetch_add(finished, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(p.done.flag, p.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_shard_arrive(const ArriveParams &p, hipStream_t stream)
{
    GRAPE_LAUNCH(shard_arrive_kernel, dim3((p.Q + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---- mailbox all-reduce between processes ------------------------------------------------------------------------------
size_t ipc_mailbox_bytes(int Q, int n_ranks)
{
    const size_t Qpad = ((size_t)Q + 255) / 256 * 256, nb = Qpad / 256;
    return sizeof(double) * 2 * (size_t)n_ranks * Qpad + sizeof(unsigned long long) * 2 * nb;
}

__global__ __launch_bounds__(256) void ipc_allreduce_kernel(const IpcParams p)
{
    __shared__ int s_ok;
    const int q = blockIdx.x * 256 + threadIdx.x, R = p.n_ranks, nb = gridDim.x;     // (the whole mailbox: Qpad / 256 block columns)
    const size_t slot = ((size_t)p.parity * R + p.rank) * p.Qpad + q, counters = (size_t)2 * R * p.Qpad;
    const double mine = q < p.Q ? p.own_row[q] : 0.0;
    for (int j = 0; j < R; ++j)                                   // my outputs into slot `rank` of every mailbox
        __hip_atomic_store(p.mbox[j] + slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = 0; j < R; ++j) {
            unsigned long long *cnt = reinterpret_cast<unsigned long long *>(p.mbox[j] + counters) + (size_t)p.parity * nb + blockIdx.x;
            (void)__hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned long long *own = reinterpret_cast<const unsigned long long *>(p.mbox[p.rank] + counters) + (size_t)p.parity * nb + blockIdx.x;
        int ok = 0;
        for (long long it = 0; it < p.spin_limit; ++it) {
            if (__hip_atomic_load(own, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= p.target) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        s_ok = ok;
    }
    __syncthreads();
    const int ok = s_ok;
    if (ok) {
        __threadfence_system();
        if (q < p.Q) {
            const double *base = p.mbox[p.rank] + (size_t)p.parity * R * p.Qpad + q;
            double v[kMaxShards];
#pragma unroll
            for (int j = 0; j < kMaxShards; ++j)
                v[j] = j < R ? __hip_atomic_load(base + (size_t)j * p.Qpad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < kMaxShards; ++j)
                acc += v[j];                                      // rank order, as the in-process sum
            if (p.out)
                p.out[q] = acc;
            if (p.done.host_out)
                p.done.host_out[q] = acc;
        }
    } else {
        // gave up: a partial sum must never pass for a result -- NaN where the sum was expected, and the host word
        // the device-pointer path has instead of a completion flag
        if (q < p.Q && p.out)
            p.out[q] = __builtin_nan("");
        if (threadIdx.x == 0 && p.fail_word)
            __hip_atomic_store(p.fail_word, p.target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!p.done.flag)
        return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // counter: low 16 bits = blocks finished, bit 16 on = some block gave up
        const unsigned prev = atomicAdd(p.done.counter, ok ? 1u : 0x10001u);
        if ((prev & 0xffffu) == (unsigned)nb - 1) {
            const bool failed = ((prev >> 16) != 0) || !ok;
            __hip_atomic_store(p.done.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(p.done.flag, failed ? (p.done.seq | kSeqFailed) : p.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_ipc_allreduce(const IpcParams &p, hipStream_t stream)
{
    GRAPE_LAUNCH(ipc_allreduce_kernel, dim3(p.Qpad / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_reduce(const double *member_out, const double *wts, double *partial, double *fg,
                         int E, int Q, int ksplit, hipStream_t stream, DoneSignal done)
{
    if (ksplit == 1 && E <= 32) {
        GRAPE_LAUNCH(reduce_few_kernel, dim3((Q + 255) / 256), dim3(256), 0, stream, member_out, wts, fg, E, Q, done);
        return hipGetLastError();
    }
    const int per_split = (E + ksplit - 1) / ksplit;
    const dim3 g1((Q + kTileQ - 1) / kTileQ, ksplit), b1(kTileQ * kSubK);
    GRAPE_LAUNCH(reduce_stage1, g1, b1, 0, stream, member_out, wts, partial, E, Q, per_split);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    const int per_block = 256 / kMaxSplit;
    GRAPE_LAUNCH(reduce_stage2, dim3((Q + per_block - 1) / per_block), dim3(256), 0, stream, partial,
                       fg, Q, ksplit, done);
    return hipGetLastError();
}

}  // namespace grape
