// grape_api.cpp -- C-ABI host layer of libgrape_hip.so (include/grape_hip.h).
//
// Owns the device workspace (what init_GRAPE allocates on the Julia heap,
// /root/reference/src/grape_tools.jl:4-16), uploads the operators once, and turns one call of
// the reference's (F, G, x) closure (src/solve.jl:164-196) into: sweep kernel -> reduce launch
// [-> one RCCL all-reduce when the ensemble spans several GPUs], all asynchronous on one HIP
// stream per device.  There is NO CPU fallback: without a gfx950 device every entry point fails
// with GRAPE_ERR_NO_DEVICE.
//
// Multi-GPU lives here, behind the C ABI, in two shapes:
//   * one process, n_devices GPUs (grape_config.n_devices/device_ids): the context is a GROUP of
//     per-device shard contexts (contiguous blocks of ceil(E/G) members); ncclCommInitAll once,
//     one grouped ncclAllReduce of K*N+1 doubles per evaluation;
//   * one process per GPU (grape_comm_attach): the context is this rank's shard and joins a
//     communicator built from a unique id the caller distributes (ncclCommInitRank).
// librccl is dlopen'ed on first use, so single-GPU users never pay for loading it.
#include "../../include/grape_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the library itself is dlopen'ed

#include <dlfcn.h>
#include <fcntl.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <new>
#include <utility>
#include <string>
#include <vector>

#include "grape_host.hpp"
#include "grape_kernels.hpp"

// (hipFuncGetAttributes on it tells grape_create whether a code object of this build matches the device)
__global__ void grape_probe_kernel() {}

using grape::SweepParams;
using grape::TileParams;
typedef std::complex<double> cplx;

struct grape_ctx {
    grape_config cfg{};
    int device = 0;
    int compute_units = 0;
    char arch[32] = {0};
    int m = 0;                    // state columns (Xi, Xt are n x m); m < n runs zero-padded to n x n
    int family = 0;               // 0: register-resident small-n kernels, 1: MFMA tile kernels (tile + grid), 2: the size-generic
                                  // kernel of sweep_any.hip (n = 1, n > 64)
    int NT = 0;                   // tile family: tiles per dimension (padded n = 16 NT)
    size_t TSZ = 0;               // tile family: double2 per matrix dump
    bool pack2 = false;           // tile family, n <= 8: two members per 16x16 tile (block diagonal)
    double2 *d_scratch = nullptr; // family 2 (sweep_any.hip): 6 matrices per (control array, member of the workspace chunk, propagator block)
    int any_blocks = 1;           // family 2: workgroups per member of the propagator launch (grape::any_prop_blocks)
    size_t scratch_bytes = 0;
    bool grid = false;            // tile family, n = 33..64 (NT = 3, 4): a workgroup of NT x NT waves per matrix (sweep_grid.hip) -- the
                                  // reference's general flow only; GRAPE_GRID=1 sends smaller sizes there too (cross-checks)
    int EU = 0;                   // tile family: wavefront-level units = members, or member pairs when pack2
    int S = 0, W = 0, LT = 0;
    bool pair = false;            // small family: lane-pair kernel (sweep_pair.hip), a time chunk per two lanes
    int CH = 0;                   // small family: time chunks per member = workspace stride (LT, or LT/2 when pair)
    int MPB = 1, NB = 0;          // small family: members per workgroup, workgroups per control array
    int B = 1;                    // batch capacity: control arrays per grape_eval_batch call
    double *d_block_out = nullptr;
    double *d_xg_scratch = nullptr;   // only when K*N is too long for the LDS staging buffer
    int ksplit = 1;
    size_t ws_elems = 0;          // double2 elements per workspace array
    // member-chunked evaluation (round 5; SURVEY.md section 7 "needs member-chunking", src/solve.jl:166-187 is a serial member
    // loop with no memory cliff): when P_t / X_t / L_t of the whole ensemble exceed the budget, the workspace arrays hold Ec
    // members and an evaluation walks the ensemble in blocks of Ec through them; per-member inputs and result rows stay whole,
    // and the weighted sum runs once over all rows in its fixed order -- a chunked evaluation is bitwise the unchunked one.
    int Ec = 0, EUc = 0;          // members / wavefront-level units per workspace chunk (= n_ensemble / EU when everything fits)
    size_t ws_unit = 0;           // double2 elements of one unit's share of a workspace array
    size_t ws_budget = 0;         // bytes the workspace arrays may take: GRAPE_MAX_WORKSPACE_BYTES, else 0.9 x free memory - the rest
    int ws_arrays = 1;            // arrays the current plan was made for (props [+ costates] [+ states])
    bool ws_invalid = false;      // a re-plan freed props / costates and could not allocate them again: the next
    bool ws_keep_costates = false; //   grape_set_operators plans and allocates afresh (with this costate wish)
    int test_fail_replan = 0;     // tests: that allocation fails this many times (GRAPE_TEST_FAIL_REPLAN)
    int ws_B = 1;                 // control arrays the workspace holds: max_batch, or 1 when that does not fit (a batch then runs
                                  // its arrays one behind the other, as it does on a member-chunked workspace)
    uint64_t bytes = 0;
    // device
    double2 *d_ops = nullptr;
    double *d_wts = nullptr;
    double *d_x = nullptr;
    double *d_fg = nullptr;
    double2 *d_props = nullptr, *d_states = nullptr, *d_costates = nullptr;
    double *d_zphi = nullptr;                  // exact gradient, unitary flow: tr M per member and control array
    bool exact_w1 = false;                     // exact gradient from the unitary flow's W_t dump (UnitaryGate, Hermitian generators, pair kernel)
    double *d_member_out = nullptr;
    double *d_partial = nullptr;
    unsigned long long *d_stamps = nullptr;
    // host
    double *h_stage = nullptr;    // pinned + device-mapped, K*N + 1 doubles: x on the way in
    double *d_h_stage = nullptr;  // device address of h_stage
    double *h_fg = nullptr;       // pinned + device-mapped, K*N + 1 doubles: the reduce kernel writes [G, F] here
    double *d_h_fg = nullptr;     // device address of h_fg
    // host-visible completion of an evaluation: the final kernel's last workgroup publishes `seq` in h_flag
    // ([1]: set by a mailbox exchange of the device-pointer path that gave up; [8 ..]: one flag per workgroup of
    // reduce_rows_mf_kernel, the publication without a device-side fan-in -- mf_wait of them belong to the evaluation in flight)
    unsigned long long *h_flag = nullptr, *d_h_flag = nullptr;
    int mf_wait = 0;
    unsigned *d_done_counter = nullptr;
    bool peer_sum = false;                     // group: [G, F] summed on the first device through peer copies (no RCCL)
    bool peer_direct = false;                  // group: the first device can read every shard's memory (peer access): the sum reads the rows in place
    bool peer_all = false;                     // group: EVERY pair of its devices has peer access (arrive-and-sum: any shard may sum)
    double *d_gather = nullptr;                // group, peer_sum: one row of K*N+1 doubles per shard, on the first device
    hipEvent_t ev_done = nullptr;              // shard of a peer_sum group: its evaluation has finished
    // round 4: the peer sum without stream dependencies -- every shard launches shard_arrive_kernel behind its own reduction
    unsigned *d_arrive = nullptr;              // group: fine-grained counters on the first device (ArriveParams.arrive)
    grape_ctx *group = nullptr;                // shard of a group: the group context
    grape::DoneSignal group_done;              // group: how the evaluation being issued is published (same value for every shard)
    int group_nx = 1;                          // group: control arrays of the evaluation being issued
    // one process per GPU without RCCL (grape_ipc_export / grape_ipc_attach)
    double *d_mbox = nullptr;                  // own mailbox (fine-grained device memory, IPC-exported)
    double *ipc_mbox[grape::kMaxShards] = {};  // every rank's mailbox as this process addresses it
    int ipc_ranks = 0;                         // > 1: attached
    int ipc_alloc_ranks = 0;                   // ranks the own mailbox was sized for
    unsigned long long ipc_evals = 0, ipc_count[2] = {0, 0};
    std::vector<double> lb_alpha;              // grape_lbfgs: accepted step length of every iteration of the last run ...
    std::vector<int32_t> lb_evals;             // ... and the evaluations made up to its end (grape_lbfgs_get_trace)
    bool broken = false;                       // a partial failure left counters / peers out of step: every further evaluation is refused
    bool thin = false;                         // rank-one states: matrix-vector chain (sweep_thin.hip)
    bool herm_ctrl = false;                    // every B_c Hermitian
    double2 *d_vecs = nullptr;                 // thin: per member [v0 | wT], 16 complex each
    bool sparse_ctrl = false;                  // tile family: every B_c has <= kSparseMax non-zeros (sparse gradient traces)
    int sp_nz = 64;                            // list length of the sparse control lists (64 .. 256)
    size_t sp_cap = 0;                         // entries behind d_sp_coef / d_sp_addr
    double2 *d_sp_coef = nullptr;              // [E][K][sp_nz]
    int32_t *d_sp_addr = nullptr;
    int hoist = 0;                             // tile family, prop_hoist.hip kernels: 1 member-invariant controls (control sum formed once per slice), 2 per-member controls
    double2 *d_ha = nullptr;                   // [EU] dumps of A'_k = (-i dt) A_k
    double *d_ha_norm = nullptr;               // [EU] |A'_k|_1 bound / theta8
    double2 *d_gc = nullptr;                   // [B][N] dumps of Gc_t
    double *d_gcn = nullptr;
    bool action = false;                       // rank-one states + member-invariant controls: exp(G) v on vectors (action_thin.hip)
    bool thin_dpp = false;                     // rank-one states, expm kernel + chain_prop_kernel (one DPP matrix-vector product per slice)
    double2 *d_wrec = nullptr;                 // its backward records (the vector flow keeps them in d_props)
    size_t wrec_bytes = 0;
    double2 *d_props_t = nullptr;              // and the P_t^T dumps the expm kernel writes beside P_t
    size_t props_t_bytes = 0;
    double2 *d_act_a = nullptr, *d_act_b = nullptr, *d_act_bf = nullptr, *d_act_g = nullptr;
    double *d_act_an = nullptr, *d_act_gn = nullptr;
    int act_R = 0;                             // sparse rows of the control operators (0: dense forms kernel)
    bool act_shared = true;                    // one set of control operators for every member
    bool ctrl_shared = false;                  // the members' control operators are identical (memcmp)
    int32_t *d_any_sp_i = nullptr;             // size-generic family, sparse shared controls: [eptr | ectl | cptr | caddr]
    double2 *d_any_sp_c = nullptr;             //                                              [ecoef | ccoef]
    size_t any_sp_i_cap = 0, any_sp_c_cap = 0, any_sp_nnz = 0, any_sp_ntouch = 0;   // capacities (elements); non-zeros of the current operators (0: dense), elements they touch
    bool ctrl_scaled = false;                  // B_{k,c} = s_k B_{0,c} with some s_k != 1 (d_ctrl_scale: s_k per member): the
    double *d_ctrl_scale = nullptr;            //   hoisted flows' pre-pass runs on member 0's operators, members scale Gc_t
    size_t act_var_bytes = 0;                  // device bytes of the vector flow's operator buffers (re-sized per upload)
    double *d_act_bn = nullptr;                // per-member controls: [E][K] norm bounds
    double2 *d_act_bs = nullptr;
    int32_t *d_act_bo = nullptr;                   // [B][N] |Gc_t|_1 bound / theta8
    size_t states_bytes = 0;                   // size of d_states (vector records are smaller than state dumps)
    int tp_C = 0, tp_S = 0;                    // tile family, unitary flow, small ensembles: time chunks per unit (0 = sequential chain)
    double2 *d_tp_q = nullptr, *d_tp_r = nullptr, *d_tp_m = nullptr;   // chunk products, products after each chunk, M_N
    double *d_tp_z = nullptr;
    double2 *d_tp_vec = nullptr;               // rank-one chain: v at every chunk's start, w at its end
    size_t tp_cap[6] = {0, 0, 0, 0, 0, 0};     // bytes behind d_tp_q, d_tp_r, d_tp_m, d_tp_vec, d_tp_z, d_tp_a
    int tp_G = 0, tp_g = 0;                    // two-level scan: groups, chunks per group
    double2 *d_tp_a = nullptr;
    bool direct_publish = true;                // GRAPE_DIRECT_PUBLISH=0: always go through the reduce kernel
    bool mf_publish = true;                    // GRAPE_MF_PUBLISH=0: reduce_rows_kernel's staged publication instead of per-workgroup host flags
    unsigned long long seq = 0;
    std::string kernel_log;                    // names of the kernels the last evaluation launched (grape_get_kernel_names)
    int x_upload = 1;             // 0: hipMemcpyAsync, 1: copy kernel reading the mapped staging buffer,
                                  // 2: the host writes x straight into fine-grained device memory (large BAR)
    double *d_x_bar = nullptr;    // mode 2: host-writable device buffer the sweep reads x from
    hipStream_t stream = nullptr;
    bool ops_set = false, evaluated = false;
    bool unitary = false;         // all generators Hermitian -> unitary data flow
    bool herm_states = false;     // all Xi, Xt Hermitian (tile family: one-product commutator)
    // GRAPE_FLAG_TIME_KERNELS: a fixed ring of start/stop event pairs around the sweep launches,
    // created at grape_create; when the ring wraps, the oldest pair is folded into ev_total_ms
    std::vector<hipEvent_t> ev;               // three per ring slot: start, middle (tile family: behind the expm kernel), stop
    std::vector<char> ev_has_mid;             // per slot: the middle event was recorded
    std::vector<float> smp_total, smp_first;  // folded per-launch durations since the last reset (grape_get_kernel_samples)
    uint64_t ev_issued = 0, ev_folded = 0;    // pairs
    uint64_t launches = 0;                    // evaluations enqueued (GRAPE_FLAG_TIME_SAMPLED)
    double ev_total_ms = 0.0;
    int64_t ev_count = 0;
    hipEvent_t ev_dev = nullptr;  // recorded after every grape_eval_device: orders the private stream behind it
    bool dev_pending = false;
    // collective: one all-reduce(sum) of [G, F] per evaluation (src/solve.jl:171-191 across GPUs)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    // group context (n_devices >= 2, or GRAPE_FLAG_FORCE_COLLECTIVE): owns no workspace itself,
    // only the per-device shard contexts
    bool is_group = false;
    std::vector<grape_ctx *> sub;
    std::vector<int> sub_lo;      // first member of every shard
    struct GroupWorker *worker = nullptr;     // shard of a group (not the first): the thread that issues its launches
    double group_tm[5] = {0, 0, 0, 0, 0};     // group: accumulated seconds [x fan-out, shard launches first->last, sum issue, wait, total]
    uint64_t group_tm_n = 0;
    double timeout_s = 600.0;
    double eval_ema_s = 0.0;      // smoothed duration of the last evaluations (long ones sleep through most of it)
    mutable std::string err;
};

static constexpr size_t kEventRing = 256;      // start/stop pairs kept before folding

// where GRAPE_LAUNCH writes the names of the kernels it launches: the log of the evaluation this thread is issuing
static thread_local std::string *g_kernel_log = nullptr;
struct KernelLogScope {
    // append: the launches that close an evaluation behind its sweep kernels (copy-out, the cross-shard sum, the exchange)
    explicit KernelLogScope(std::string *log, bool append = false) { g_kernel_log = log; if (log && !append) log->clear(); }
    ~KernelLogScope() { g_kernel_log = nullptr; }
};
namespace grape {
void log_kernel(const char *expr)
{
    std::string *log = g_kernel_log;
    if (!log || !expr) return;
    while (*expr == '(' || *expr == ' ') ++expr;             // "(chain_tile_split_kernel<1, true>)" -> chain_tile_split_kernel
    const char *end = expr;
    while (*end && *end != '<' && *end != ')' && *end != ' ') ++end;
    if (!log->empty()) log->push_back(';');
    log->append(expr, (size_t)(end - expr));
}
}  // namespace grape

static thread_local std::string g_create_err = "";

static int fail(const grape_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

#define HIP_TRY(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(ctx, e__ == hipErrorOutOfMemory ? GRAPE_ERR_ALLOC : GRAPE_ERR_HIP,     \
                        std::string(#call) + ": " + hipGetErrorString(e__));                   \
    } while (0)

// ------------------------------------------------------------------------------------------
// RCCL, loaded on demand (573 MB shared object: single-GPU users never touch it)
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
static RcclApi g_rccl;

static RcclApi *rccl()
{
    static std::once_flag once;
    std::call_once(once, [] {
        RcclApi &api = g_rccl;
        // a librccl already mapped into the process (e.g. the one a host framework bundles) is
        // found by SONAME and reused
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char *e = dlerror();
            api.err = std::string("cannot load librccl: ") + (e ? e : "?");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(api.handle, name);
            if (!p && ok) { ok = false; api.err = std::string("librccl lacks ") + name; }
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) api.handle = nullptr;
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

#define NCCL_TRY(ctx, call)                                                                    \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess)                                                                \
            return fail(ctx, GRAPE_ERR_COMM, std::string(#call) + ": " + g_rccl.GetErrorString(r__)); \
    } while (0)

// Every entry point selects its shard's device; the CALLER's current device is restored on every return path (a host
// framework in the same thread -- torch tensors, the caller's own streams -- keeps allocating and launching where it was).
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// One issuing thread per shard of an in-process group (all but the first, which the calling thread serves): a group
// evaluation used to be issued shard after shard by ONE host thread -- two launches and an event per shard, ~10 us each,
// so that the eighth shard of a C3 evaluation started ~70 us after the first, twice its own 32 us kernel.  The workers
// spin on a sequence word while evaluations keep coming (an optimiser calls every ~100 us) and go to sleep on a
// condition variable after 2 ms without work.
struct GroupWorker {
    std::thread th;
    std::atomic<uint64_t> req{0}, done{0};
    std::atomic<bool> asleep{false}, stop{false};
    std::mutex mu;
    std::condition_variable cv;
    grape_ctx *shard = nullptr;
    int (*job)(grape_ctx *) = nullptr;             // what to issue for this shard (set before req is bumped)
    int rc = 0;
    // Sleep / wake handshake: the worker publishes `asleep` and THEN re-reads `req`; the poster bumps `req` and THEN reads
    // `asleep`.  Both pairs are store -> load, which release / acquire does not order (the store can sit in the store
    // buffer while the load runs: worker reads the old req, poster reads asleep == false, nobody notifies -- ADVICE r3).
    // All four accesses are sequentially consistent, so at least one side sees the other's store.
    void post(int (*fn)(grape_ctx *))
    {
        job = fn;
        req.fetch_add(1, std::memory_order_seq_cst);
        if (asleep.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lk(mu);
            cv.notify_one();
        }
    }
    // the poster's side: spin until the job has been issued; after `timeout_s` the shard is reported as hung
    // (GRAPE_ERR_TIMEOUT) instead of spinning forever
    int wait(double timeout_s)
    {
        const uint64_t want = req.load(std::memory_order_relaxed);
        unsigned spins = 0;
        timespec t0{0, 0};
        while (done.load(std::memory_order_acquire) != want) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((++spins & 0xffff) == 0) {
                timespec t;
                clock_gettime(CLOCK_MONOTONIC, &t);
                if (!t0.tv_sec && !t0.tv_nsec)
                    t0 = t;
                else if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > timeout_s)
                    return GRAPE_ERR_TIMEOUT;
                {                                             // belt and braces: a sleeping worker with work posted is woken again
                    std::lock_guard<std::mutex> lk(mu);
                    cv.notify_one();
                }
            }
        }
        return rc;
    }
    void run();
};

// diagnostic switches: set and not "0"
static bool env_on(const char *name)
{
    const char *v = std::getenv(name);
    return v && v[0] && !(v[0] == '0' && !v[1]);
}

// ... and the ones that switch a default off: set to exactly "0"
static bool env_off(const char *name)
{
    const char *v = std::getenv(name);
    return v && v[0] == '0' && !v[1];
}

static size_t KN(const grape_ctx *c) { return (size_t)c->cfg.n_controls * c->cfg.n_slices; }

static void free_all(grape_ctx *c)
{
    if (!c) return;
    if (c->worker) {
        GroupWorker *w = c->worker;
        w->stop.store(true);
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->cv.notify_one();
        }
        if (w->th.joinable()) w->th.join();
        delete w;
        c->worker = nullptr;
    }
    for (grape_ctx *s : c->sub) free_all(s);
    c->sub.clear();
    if (c->is_group) {
        if (c->d_gather) { (void)hipSetDevice(c->device); (void)hipFree(c->d_gather); }
        if (c->d_arrive) { (void)hipSetDevice(c->device); (void)hipFree(c->d_arrive); }
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    for (int j = 0; j < c->ipc_ranks; ++j)
        if (c->ipc_mbox[j] && c->ipc_mbox[j] != c->d_mbox) (void)hipIpcCloseMemHandle(c->ipc_mbox[j]);
    if (c->d_mbox) (void)hipFree(c->d_mbox);
    if (c->comm && g_rccl.handle) (void)g_rccl.CommDestroy(c->comm);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->ev_dev) (void)hipEventDestroy(c->ev_dev);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->d_ops); (void)hipFree(c->d_wts); (void)hipFree(c->d_x); (void)hipFree(c->d_fg);
    (void)hipFree(c->d_props); (void)hipFree(c->d_states); (void)hipFree(c->d_costates); (void)hipFree(c->d_zphi);
    (void)hipFree(c->d_ctrl_scale);
    (void)hipFree(c->d_member_out); (void)hipFree(c->d_partial); (void)hipFree(c->d_stamps); (void)hipFree(c->d_block_out); (void)hipFree(c->d_xg_scratch);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_fg) (void)hipHostFree(c->h_fg);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    (void)hipFree(c->d_done_counter);
    (void)hipFree(c->d_scratch);
    (void)hipFree(c->d_vecs);
    (void)hipFree(c->d_sp_coef); (void)hipFree(c->d_sp_addr);
    (void)hipFree(c->d_any_sp_i); (void)hipFree(c->d_any_sp_c);
    (void)hipFree(c->d_tp_q); (void)hipFree(c->d_tp_r); (void)hipFree(c->d_tp_m); (void)hipFree(c->d_tp_z); (void)hipFree(c->d_tp_vec); (void)hipFree(c->d_tp_a);
    (void)hipFree(c->d_x_bar);
    (void)hipFree(c->d_ha); (void)hipFree(c->d_ha_norm); (void)hipFree(c->d_gc); (void)hipFree(c->d_gcn);
    (void)hipFree(c->d_act_a); (void)hipFree(c->d_act_b); (void)hipFree(c->d_act_bf); (void)hipFree(c->d_act_g);
    (void)hipFree(c->d_act_an); (void)hipFree(c->d_act_gn); (void)hipFree(c->d_act_bs); (void)hipFree(c->d_act_bo); (void)hipFree(c->d_act_bn); (void)hipFree(c->d_wrec); (void)hipFree(c->d_props_t);
    delete c;
}

namespace grape {
hipError_t ensure_dynamic_lds(const void *fn, size_t lds)
{
    struct Granted { const void *fn; int dev; size_t lds; };
    static std::mutex mu;
    static std::vector<Granted> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::lock_guard<std::mutex> lock(mu);
    for (Granted &g : table)
        if (g.fn == fn && g.dev == dev) {
            if (lds <= g.lds)
                return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                g.lds = lds;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        table.push_back({fn, dev, lds});
    return e;
}
}  // namespace grape

extern "C" int grape_abi_version(void) { return GRAPE_ABI_VERSION; }

extern "C" const char *grape_last_error(const grape_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

static int validate_config(const grape_config *cfg)
{
    if (cfg->sys_type < GRAPE_UNITARY_GATE || cfg->sys_type > GRAPE_COHERENCE_TRANSFER)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: bad sys_type");
    if (cfg->variant != GRAPE_VARIANT_INPLACE && cfg->variant != GRAPE_VARIANT_STATIC)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: bad variant");
    if (cfg->n < 1 || cfg->n_controls < 1 || cfg->n_slices < 1 || cfg->n_ensemble < 1)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG,
                    "grape_create: n, n_controls, n_slices, n_ensemble must be positive");
    if (!(cfg->duration == cfg->duration))
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: duration is NaN");
    if (cfg->n_state_cols < 0 || cfg->n_state_cols > cfg->n)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: n_state_cols must be in 0..n");
    if (cfg->n_devices < 0 || cfg->n_devices > GRAPE_MAX_DEVICES)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: n_devices must be in 0..8");
    if (cfg->max_batch < 0)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: max_batch is negative");
    if ((uint64_t)cfg->n_controls * (uint64_t)cfg->n_slices >= (1ull << 31))
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: n_controls * n_slices overflows int32");
    const int wmax = grape::sweep_small_max_waves(cfg->n);
    const int nt = grape::tile_count(cfg->n);
    if ((uint64_t)cfg->n > 2048)
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED,
                    "grape_create: operator dimension n=" + std::to_string(cfg->n) + " is beyond what this build indexes (n <= 2048)");
    if (wmax == 0 && nt == 0 && cfg->gradient == GRAPE_GRADIENT_EXACT)
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED, "grape_create: the exact gradient exists for 2 <= n <= 64 (other sizes run the reference's first-order flow)");
    const int m = cfg->n_state_cols ? cfg->n_state_cols : cfg->n;
    if (m != cfg->n && cfg->sys_type != GRAPE_UNITARY_GATE)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG,
                    "grape_create: n x m states with m < n need UnitaryGate (the sandwich X P' is not defined)");
    if (cfg->gradient < 0 || cfg->gradient > 1 || cfg->objective < 0 || cfg->objective > 1)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: bad gradient / objective");
    if (cfg->objective == GRAPE_OBJECTIVE_C1 && cfg->gradient != GRAPE_GRADIENT_EXACT)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG,
                    "grape_create: the C1 functional (ADGRAPE path) comes with the exact gradient: set gradient = 1");
    if ((cfg->flags & GRAPE_FLAG_PHASE_STAMPS) && wmax == 0)
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED, "grape_create: GRAPE_FLAG_PHASE_STAMPS exists for n <= 4 only (the tile kernels write no stamps)");
    return GRAPE_OK;
}

// How many members the workspace arrays hold: all of them when `arrays` arrays of the whole ensemble (x max_batch) fit the
// budget, else as many whole units of work as fit ONE control array's share (a chunked context runs the arrays of a batch
// one behind the other) -- in multiples of the members a workgroup / a wave pairs up, so that a member's arithmetic does
// not depend on where the chunks are cut.  false: not even one granule fits.
static bool plan_chunk(grape_ctx *c, int arrays)
{
    // (every field is committed at the end: a plan that does not fit leaves the context as it was)
    const size_t E = (size_t)c->cfg.n_ensemble, unit_bytes = sizeof(double2) * c->ws_unit * (size_t)arrays;
    const size_t units = c->family == 0 ? E : (size_t)c->EU;
    int ws_B = c->B, Ec, EUc;
    if (unit_bytes * units * (size_t)c->B <= c->ws_budget) {
        Ec = (int)E;
        EUc = c->EU;
    } else {
        ws_B = 1;
        size_t fit = c->ws_budget / unit_bytes;                      // units of one control array
        const size_t gran = c->family == 0 ? (size_t)c->MPB : (c->pack2 ? 1 : 2);
        fit = fit / gran * gran;
        if (fit < gran) return false;
        if (fit > units) fit = units;
        EUc = (int)fit;
        Ec = c->family == 1 && c->pack2 ? (int)std::min(E, 2 * fit) : (int)fit;
    }
    c->ws_arrays = arrays;
    c->ws_B = ws_B;
    c->Ec = Ec;
    c->EUc = EUc;
    c->ws_elems = c->ws_unit * (size_t)(c->family == 0 ? c->Ec : c->EUc);
    return true;
}
static bool chunked(const grape_ctx *c) { return c->Ec < c->cfg.n_ensemble; }
// Could ANY flow of this context end up member-chunked (three arrays of the whole ensemble exceed the budget)?  The chunked
// time axis and the small-ensemble propagator chain keep per-(unit, chunk) buffers of the whole ensemble: they are not chosen
// then (a question that only arises under a test budget: ensembles small enough for those flows are far below any real one).
static bool may_chunk(const grape_ctx *c)
{
    const size_t units = c->family == 0 ? (size_t)c->cfg.n_ensemble : (size_t)c->EU;
    return sizeof(double2) * c->ws_unit * 3 * units * (size_t)c->B > c->ws_budget;
}
static size_t ws_batch(const grape_ctx *c) { return (size_t)c->ws_B; }                    // control arrays the workspace holds

// one device, one contiguous shard of members: the workspace init_GRAPE allocates
static int create_shard(const grape_config *cfg, int dev, grape_ctx **out)
{
    int wmax = grape::sweep_small_max_waves(cfg->n);
    const int nt = grape::tile_count(cfg->n);
    // n = 4 runs the lane-pair kernel (two waves per SIMD); n = 2, 3 the lane-per-chunk kernel.
    // GRAPE_SMALL_KERNEL=pair|lane overrides (parity tests run both).
    bool pair = cfg->n == 4;
    if (const char *sk = std::getenv("GRAPE_SMALL_KERNEL")) {
        if (!std::strcmp(sk, "pair")) pair = true;
        if (!std::strcmp(sk, "lane")) pair = false;
    }
    if (grape::sweep_pair_max_waves(cfg->n) == 0 || wmax == 0) pair = false;
    if (pair) wmax = grape::sweep_pair_max_waves(cfg->n);
    HIP_TRY(nullptr, hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GRAPE_ERR_NO_DEVICE,
                    std::string("grape_create: device is ") + prop.gcnArchName +
                        ", this library carries gfx950 code only");

    {   // the code objects are built for one XNACK setting (Makefile TARGETS, default gfx950:xnack-): say so, instead of
        // failing at the first launch, when this device runs with the other one
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, (const void *)grape_probe_kernel) != hipSuccess) {
            (void)hipGetLastError();
            return fail(nullptr, GRAPE_ERR_NO_DEVICE,
                        std::string("grape_create: no code object of this library matches the device (") + prop.gcnArchName +
                            "); rebuild with `make TARGETS=gfx950` for a device running with XNACK on");
        }
    }
    grape_ctx *c = new (std::nothrow) grape_ctx();
    if (!c) return fail(nullptr, GRAPE_ERR_ALLOC, "grape_create: out of host memory");
    c->cfg = *cfg;
    c->m = cfg->n_state_cols ? cfg->n_state_cols : cfg->n;
    c->device = dev;
    c->compute_units = prop.multiProcessorCount;
    std::snprintf(c->arch, sizeof(c->arch), "%s", prop.gcnArchName);
    if (const char *ts = std::getenv("GRAPE_EVAL_TIMEOUT_S")) {
        const double v = std::atof(ts);
        if (v > 0) c->timeout_s = v;
    }

    // time-axis decomposition: W waves per member, S slices per lane, S * 64 * W >= N.
    // Aim for one wave per SIMD across the chip; more waves per member only when the
    // ensemble alone cannot fill it.
    const int N = cfg->n_slices, E = cfg->n_ensemble;
    c->family = wmax > 0 ? 0 : (nt > 0 ? 1 : 2);              // 2: n = 1 or n > 64 -- the size-generic kernel (sweep_any.hip)
    c->B = cfg->max_batch > 1 ? cfg->max_batch : 1;
    c->NT = nt;
    c->TSZ = (size_t)nt * nt * 256;
    c->pack2 = (c->family == 1 && cfg->n <= 8 && !std::getenv("GRAPE_TILE_NOPACK") &&
                cfg->gradient != GRAPE_GRADIENT_EXACT);        // the exact-gradient kernel works on whole tiles
    c->grid = c->family == 1 && (nt > 2 || (env_on("GRAPE_GRID") && cfg->gradient != GRAPE_GRADIENT_EXACT));
    if (c->grid) c->pack2 = false;
    c->EU = c->pack2 ? (E + 1) / 2 : E;
    c->pair = pair && c->family == 0;
    const int cpw = c->pair ? 32 : 64;                           // time chunks per wave
    int W = cfg->waves_per_member;
    if (W <= 0) {
        // n = 4 lane kernel: one wave per SIMD (256 registers); pair kernel and the n = 2, 3 lane kernels: two (their
        // registers allow it, and the FP64 pipe needs two waves to run near its peak: n = 2, E = 1024, N = 500
        // 28.0 -> 24.4 us, n = 3 48.4 -> 46.1 us)
        const long slots = ((c->pair || cfg->n <= 3) ? 8L : 4L) * c->compute_units;
        const long units = (long)E * c->B;                       // a batch fills the chip like a larger ensemble
        W = (int)((slots + units - 1) / units);
        const int wneed = (N + cpw - 1) / cpw;
        if (W > wneed) W = wneed;
        if (W > 8) W = 8;        // the wave totals of the scan combine sequentially: beyond 8 waves per member that
                                 // costs more than the shorter chunks save (single qubit, N = 1000: 15.2 vs 18.1 us)
        // No more than 16 slices per lane (pair).  An ensemble that fills the device alone got W = 1 whatever N was -- n = 4,
        // E = 4096, N = 2000: 63 slices per pair, the x / g staging no longer fits LDS, 2.01 ms; with W = 4 (16 slices) 1.13 ms;
        // E = 2048 / 4096 at N = 1000: 0.346 / 0.645 -> 0.312 / 0.592 ms with W = 2; E = 1024, N = 2000: 0.352 -> 0.309 with W = 4;
        // n = 2 / 3, E = 4096, N = 4000: 1.07 / 1.86 -> 0.59 / 1.12 ms with W = 4 (round 6, tools/w_sweep.py, profiles/r06_w_sweep.txt).
        // It also keeps n x 1 problems at n = 4 inside the vector sweep's 16 slices per lane.
        if (c->family == 0)
            while (W < 8 && 2 * W <= wmax && (N + cpw * W - 1) / (cpw * W) > 16)
                W *= 2;
    }
    if (wmax > 0 && W > wmax) W = wmax;
    if (W < 1 || c->family != 0) W = 1;
    int S = cfg->slices_per_lane;
    const int smin = (N + cpw * W - 1) / (cpw * W);
    if (S < smin) S = smin;
    c->W = W; c->S = S; c->LT = 64 * W; c->CH = cpw * W;
    if (c->family == 0 && (uint64_t)cfg->n_controls * N * S * cfg->n_controls >= (1ull << 32)) {
        delete c;
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED, "grape_create: n_controls^2 * n_slices^2 too large for the LDS index arithmetic");
    }
    const int wg_waves = c->pair ? 8 : 4;                    // a workgroup fills the CU's 4 SIMDs (x2 for the pair kernel)
    c->MPB = (c->family == 0 && W <= wg_waves) ? wg_waves / W : 1;
    if (const char *ev = std::getenv("GRAPE_MPB")) {           // tuning experiment: members per workgroup
        const int v = std::atoi(ev);
        if (c->family == 0 && v >= 1 && v * W <= wmax) c->MPB = v;
    }
    if (c->MPB > E) c->MPB = E;
    bool xg_in_lds = true;
    if (c->family == 0) {                                    // fit the x/g staging buffer into LDS
        const size_t cap = 150 * 1024;
        auto lds_need = [&](int mpb, bool in_lds) {
            return c->pair ? grape::sweep_pair_lds_bytes(cfg->n, mpb, c->LT, S, cfg->n_controls, in_lds)
                           : grape::sweep_small_lds_bytes(cfg->n, mpb, c->LT, S, cfg->n_controls, in_lds);
        };
        while (c->MPB > 1 && lds_need(c->MPB, true) > cap)
            c->MPB /= 2;
        xg_in_lds = lds_need(c->MPB, true) <= cap;
    }
    c->NB = (E + c->MPB - 1) / c->MPB;
    c->ksplit = grape::reduce_ksplit(E);

    const size_t nn = (size_t)cfg->n * cfg->n, K = cfg->n_controls;
    const size_t Q = KN(c) + 1;
    c->ws_unit = c->family == 0 ? (size_t)S * nn * c->CH : (c->family == 1 ? (size_t)N * c->TSZ : (size_t)N * nn);
    const size_t ops_elems = c->family != 1 ? (size_t)E * (K + 3) * nn : (size_t)c->EU * (2 * K + 3) * c->TSZ;
    const bool exact = cfg->gradient == GRAPE_GRADIENT_EXACT;  // needs every X_t and L_t in HBM: the debug flow
    const bool keepl = (cfg->flags & GRAPE_FLAG_KEEP_COSTATES) != 0 || exact;
    {
        // workspace budget: what is free now minus what is allocated besides the workspace arrays (operators, rows, staging;
        // twice, for the buffers grape_set_operators adds), 90 % of it.  GRAPE_MAX_WORKSPACE_BYTES overrides (tests).
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b == 0) {   // no figure: plan for the whole device and let
            hipDeviceProp_t pr{};                                            // the allocations below decide
            free_b = hipGetDeviceProperties(&pr, c->device) == hipSuccess ? pr.totalGlobalMem : ~(size_t)0 >> 1;
        }
        const size_t rest = 2 * (sizeof(double2) * ops_elems + sizeof(double) * (size_t)E * Q * c->B) + ((size_t)64 << 20);
        c->ws_budget = free_b > rest ? (size_t)(0.9 * (double)(free_b - rest)) : 0;
        if (const char *ev = std::getenv("GRAPE_MAX_WORKSPACE_BYTES")) c->ws_budget = (size_t)std::strtoull(ev, nullptr, 10);
        if (const char *ev = std::getenv("GRAPE_TEST_FAIL_REPLAN")) c->test_fail_replan = std::atoi(ev);
        // optimistic plan: the propagators (and the costates when asked for); grape_set_operators plans again when the flow
        // it chooses stores the forward states as well
        if (!plan_chunk(c, 1 + (keepl ? 1 : 0))) {
            const std::string msg = "grape_create: not even " + std::to_string(c->family == 0 ? c->MPB : 2) +
                                    " members' workspace fits the budget of " + std::to_string(c->ws_budget) + " bytes";
            delete c;
            return fail(nullptr, GRAPE_ERR_ALLOC, msg);
        }
    }

    auto alloc = [&](void **p, size_t bytes) -> hipError_t {
        c->bytes += bytes;
        return hipMalloc(p, bytes);
    };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = alloc((void **)&c->d_ops, sizeof(double2) * ops_elems);
    if (e == hipSuccess) e = alloc((void **)&c->d_wts, sizeof(double) * E);
    const size_t Bn = (size_t)c->B;
    if (e == hipSuccess) e = alloc((void **)&c->d_x, sizeof(double) * KN(c) * Bn);
    if (e == hipSuccess) e = alloc((void **)&c->d_fg, sizeof(double) * Q * Bn);
    if (e == hipSuccess) e = alloc((void **)&c->d_props, sizeof(double2) * c->ws_elems * ws_batch(c));
    if (e == hipSuccess && keepl) e = alloc((void **)&c->d_costates, sizeof(double2) * c->ws_elems * ws_batch(c));
    if (e == hipSuccess && exact && c->family == 0) e = alloc((void **)&c->d_zphi, sizeof(double) * 2 * E * Bn);
    const bool want_rows = c->family != 0 || (cfg->flags & GRAPE_FLAG_MEMBER_RESULTS) || exact;
    if (e == hipSuccess && want_rows) e = alloc((void **)&c->d_member_out, sizeof(double) * E * Q * Bn);
    if (e == hipSuccess) e = alloc((void **)&c->d_partial, sizeof(double) * c->ksplit * Q);
    if (e == hipSuccess && c->family == 0) e = alloc((void **)&c->d_block_out, sizeof(double) * c->NB * Q * Bn);
    if (e == hipSuccess && c->family == 0 && !xg_in_lds)
        e = alloc((void **)&c->d_xg_scratch,
                  sizeof(double) * Bn * c->NB * ((size_t)c->MPB * c->CH * ((size_t)S * K + 1) + c->MPB));
    if (e == hipSuccess && (cfg->flags & GRAPE_FLAG_PHASE_STAMPS)) {
        const size_t sb = sizeof(unsigned long long) * Bn * E * W * grape::kStampSlots;
        e = alloc((void **)&c->d_stamps, sb);
        if (e == hipSuccess) e = hipMemset(c->d_stamps, 0, sb);
    }
    const unsigned hflags = hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable;   // (any shard's device may publish)
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_stage, sizeof(double) * Q * Bn, hflags);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->d_h_stage, c->h_stage, 0);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_fg, sizeof(double) * Q * Bn, hflags);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->d_h_fg, c->h_fg, 0);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_flag, 64 + sizeof(unsigned long long) * grape::kMaxMflags, hflags);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->d_h_flag, c->h_flag, 0);
    if (e == hipSuccess) { std::memset(c->h_flag, 0, 64 + sizeof(unsigned long long) * grape::kMaxMflags); e = alloc((void **)&c->d_done_counter, 64); }
    if (e == hipSuccess) e = hipMemset(c->d_done_counter, 0, 64);
    {
        const char *dp = std::getenv("GRAPE_DIRECT_PUBLISH");
        c->direct_publish = !(dp && dp[0] == '0');
        c->mf_publish = !env_off("GRAPE_MF_PUBLISH");
    }
    {
        // x upload path.  Default: if the device exposes its memory to the CPU (large BAR), the host
        // writes x straight into a fine-grained device buffer -- no upload kernel, no kernel boundary;
        // otherwise (or GRAPE_X_UPLOAD=kernel|memcpy) a copy kernel / hipMemcpyAsync moves the staged x.
        const char *xu = std::getenv("GRAPE_X_UPLOAD");
        int mode = 2;
        if (xu) mode = !std::strcmp(xu, "memcpy") ? 0 : (!std::strcmp(xu, "kernel") ? 1 : 2);
        int large_bar = 0;
        if (mode == 2 && e == hipSuccess &&
            (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess || !large_bar))
            mode = 1;
        if (mode == 2 && e == hipSuccess) {
            void *p = nullptr;
            if (hipExtMallocWithFlags(&p, sizeof(double) * KN(c) * Bn, hipDeviceMallocFinegrained) == hipSuccess && p) {
                // is the buffer writable from the CPU?  read(2) into it fails with EFAULT instead of faulting
                const int fd = open("/dev/zero", O_RDONLY);
                const bool ok = fd >= 0 && read(fd, p, 64) == 64;
                if (fd >= 0) close(fd);
                if (ok) {
                    c->d_x_bar = (double *)p;
                    c->bytes += sizeof(double) * KN(c) * Bn;
                } else {
                    (void)hipFree(p);
                    mode = 1;
                }
            } else {
                (void)hipGetLastError();
                mode = 1;
            }
        }
        c->x_upload = mode;
        if (std::getenv("GRAPE_DEBUG"))
            std::fprintf(stderr, "[grape] x upload mode %d (large_bar=%d)\n", mode, large_bar);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_dev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e == hipSuccess && (cfg->flags & GRAPE_FLAG_TIME_KERNELS)) {
        c->ev.reserve(3 * kEventRing);
        c->ev_has_mid.assign(kEventRing, 0);
        for (size_t i = 0; i < 3 * kEventRing && e == hipSuccess; ++i) {
            hipEvent_t evn = nullptr;
            e = hipEventCreate(&evn);
            if (e == hipSuccess) c->ev.push_back(evn);
        }
    }
    if (e != hipSuccess) {
        std::string msg = std::string("grape_create: device allocation failed: ") + hipGetErrorString(e);
        free_all(c);
        return fail(nullptr, e == hipErrorOutOfMemory ? GRAPE_ERR_ALLOC : GRAPE_ERR_HIP, msg);
    }
    *out = c;
    return GRAPE_OK;
}

// contiguous blocks of ceil(E / G) members; trailing devices may stay unused
static void shard_plan(int E, int G, std::vector<int> &lo)
{
    const int per = (E + G - 1) / G;
    lo.clear();
    for (int s = 0; s < E; s += per) lo.push_back(s);
    lo.push_back(E);
}

extern "C" int grape_create(const grape_config *cfg, grape_ctx **out)
{
    DeviceGuard guard;
    if (out) *out = nullptr;
    if (!cfg || !out) return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: null argument");
    int rc = validate_config(cfg);
    if (rc) return rc;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, GRAPE_ERR_NO_DEVICE, "grape_create: no HIP device visible");
    const bool group = cfg->n_devices >= 2 || (cfg->flags & GRAPE_FLAG_FORCE_COLLECTIVE);
    std::vector<int> devs;
    if (cfg->n_devices >= 2) {
        for (int i = 0; i < cfg->n_devices; ++i) {
            const int d = cfg->device_ids[i];
            if (d < 0 || d >= ndev)
                return fail(nullptr, GRAPE_ERR_NO_DEVICE,
                            "grape_create: device_ids[" + std::to_string(i) + "]=" + std::to_string(d) +
                                " but " + std::to_string(ndev) + " HIP device(s) are visible");
            for (int j = 0; j < i; ++j)
                if (cfg->device_ids[j] == d && !(cfg->flags & GRAPE_FLAG_GROUP_PEER_SUM))
                    return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: device_ids holds a duplicate "
                                                                "(allowed only with GRAPE_FLAG_GROUP_PEER_SUM)");
            devs.push_back(d);
        }
    } else {
        int dev = cfg->n_devices == 1 && cfg->device < 0 ? cfg->device_ids[0] : cfg->device;
        if (dev < 0) {
            if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        }
        if (dev >= ndev)
            return fail(nullptr, GRAPE_ERR_NO_DEVICE, "grape_create: device ordinal out of range");
        devs.push_back(dev);
    }
    if (!group)
        return create_shard(cfg, devs[0], out);

    if (cfg->flags & GRAPE_FLAG_PHASE_STAMPS)
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED, "grape_create: GRAPE_FLAG_PHASE_STAMPS is single-device");
    const bool peer_sum = (cfg->flags & GRAPE_FLAG_GROUP_PEER_SUM) != 0 && cfg->n_devices >= 2;
    RcclApi *api = peer_sum ? nullptr : rccl();
    if (!peer_sum && !api) return fail(nullptr, GRAPE_ERR_COMM, "grape_create: " + g_rccl.err);

    grape_ctx *g = new (std::nothrow) grape_ctx();
    if (!g) return fail(nullptr, GRAPE_ERR_ALLOC, "grape_create: out of host memory");
    g->cfg = *cfg;
    g->is_group = true;
    g->B = cfg->max_batch > 1 ? cfg->max_batch : 1;       // (every shard evaluates the same n_x control arrays per call)
    shard_plan(cfg->n_ensemble, (int)devs.size(), g->sub_lo);
    const int G = (int)g->sub_lo.size() - 1;
    for (int i = 0; i < G; ++i) {
        grape_config sc = *cfg;
        sc.n_ensemble = g->sub_lo[i + 1] - g->sub_lo[i];
        sc.device = devs[i];
        sc.n_devices = 0;
        sc.flags &= ~(GRAPE_FLAG_FORCE_COLLECTIVE | GRAPE_FLAG_GROUP_PEER_SUM);
        grape_ctx *s = nullptr;
        rc = create_shard(&sc, devs[i], &s);
        if (rc) { free_all(g); return rc; }
        g->sub.push_back(s);
    }
    if (peer_sum) {
        g->peer_sum = true;
        g->device = g->sub[0]->device;
        const size_t Q = (size_t)cfg->n_controls * cfg->n_slices + 1;
        if (hipSetDevice(g->device) != hipSuccess ||
            hipMalloc((void **)&g->d_gather, sizeof(double) * Q * G * (size_t)g->B) != hipSuccess) {
            free_all(g);
            return fail(nullptr, GRAPE_ERR_ALLOC, "grape_create: device allocation failed (shard gather buffer)");
        }
        for (int i = 0; i < G; ++i) {
            g->sub[i]->comm_rank = i;
            g->sub[i]->comm_size = G;
            g->sub[i]->group = g;
        }
        // arrive-and-sum counters (one per 256 outputs + one): fine-grained, so that system-scope atomics from every shard's
        // device meet in one place.  Without them (allocation refused) the sum keeps the stream-ordered reduction kernel.
        if (G <= grape::kMaxShards && !env_on("GRAPE_GROUP_STREAM_SUM")) {
            void *pa = nullptr;
            const size_t nb = (Q * (size_t)g->B + 255) / 256 + 1;
            if (hipExtMallocWithFlags(&pa, sizeof(unsigned) * nb, hipDeviceMallocFinegrained) == hipSuccess && pa &&
                hipMemset(pa, 0, sizeof(unsigned) * nb) == hipSuccess && hipDeviceSynchronize() == hipSuccess)
                g->d_arrive = (unsigned *)pa;
            else {
                if (pa) (void)hipFree(pa);
                (void)hipGetLastError();
            }
        }
    } else {
        std::vector<ncclComm_t> comms(G);
        std::vector<int> used(devs.begin(), devs.begin() + G);
        const ncclResult_t nr = api->CommInitAll(comms.data(), G, used.data());
        if (nr != ncclSuccess) {
            free_all(g);
            return fail(nullptr, GRAPE_ERR_COMM, std::string("grape_create: ncclCommInitAll: ") + api->GetErrorString(nr));
        }
        for (int i = 0; i < G; ++i) {
            g->sub[i]->comm = comms[i];
            g->sub[i]->comm_rank = i;
            g->sub[i]->comm_size = G;
            g->sub[i]->group = g;
        }
    }
    // peer access between every pair of distinct devices of the group: x fan-out and the [G, F] rows travel device to
    // device over xGMI instead of being staged through host memory ("already enabled" is fine)
    g->peer_direct = g->peer_all = G <= grape::kMaxShards;
    for (int i = 0; i < G; ++i)
        for (int j = 0; j < G; ++j) {
            const int di = g->sub[i]->device, dj = g->sub[j]->device;
            if (di == dj) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess || !can) {
                if (i == 0) g->peer_direct = false;
                g->peer_all = false;
                continue;
            }
            if (hipSetDevice(di) == hipSuccess) {
                const hipError_t pe = hipDeviceEnablePeerAccess(dj, 0);
                if (pe != hipSuccess) (void)hipGetLastError();     // hipErrorPeerAccessAlreadyEnabled and friends
            }
        }
    // one issuing thread per shard beyond the first (GroupWorker): the shards of an evaluation are launched at once
    for (int i = 1; i < G; ++i) {
        GroupWorker *w = new (std::nothrow) GroupWorker();
        if (!w) { free_all(g); return fail(nullptr, GRAPE_ERR_ALLOC, "grape_create: out of host memory"); }
        w->shard = g->sub[i];
        g->sub[i]->worker = w;
        try {
            w->th = std::thread([w] { w->run(); });
        } catch (...) {
            free_all(g);
            return fail(nullptr, GRAPE_ERR_ALLOC, "grape_create: cannot start the issuing thread of a shard");
        }
    }
    const grape_ctx *s0 = g->sub[0];
    g->device = s0->device; g->compute_units = s0->compute_units; g->family = s0->family;
    g->S = s0->S; g->W = s0->W; g->LT = s0->LT; g->comm_size = G; g->timeout_s = s0->timeout_s;
    std::snprintf(g->arch, sizeof(g->arch), "%s", s0->arch);
    for (const grape_ctx *s : g->sub) g->bytes += s->bytes;
    *out = g;
    return GRAPE_OK;
}

extern "C" int grape_destroy(grape_ctx *ctx)
{
    DeviceGuard guard;
    if (!ctx) return GRAPE_OK;
    if (ctx->is_group) {
        for (grape_ctx *s : ctx->sub) {
            (void)hipSetDevice(s->device);
            (void)hipDeviceSynchronize();
        }
    } else {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
    }
    free_all(ctx);
    return GRAPE_OK;
}

extern "C" int grape_comm_unique_id(grape_comm_id *out)
{
    if (!out) return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_comm_unique_id: null argument");
    static_assert(sizeof(grape_comm_id) == sizeof(ncclUniqueId), "grape_comm_id must hold an ncclUniqueId");
    RcclApi *api = rccl();
    if (!api) return fail(nullptr, GRAPE_ERR_COMM, "grape_comm_unique_id: " + g_rccl.err);
    ncclUniqueId id;
    NCCL_TRY(nullptr, api->GetUniqueId(&id));
    std::memcpy(out->bytes, &id, sizeof(id));
    return GRAPE_OK;
}

extern "C" int grape_comm_attach(grape_ctx *c, const grape_comm_id *id, int32_t rank, int32_t n_ranks)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_comm_attach: bad rank / n_ranks / id");
    if (c->is_group)
        return fail(c, GRAPE_ERR_UNSUPPORTED, "grape_comm_attach: the context already spans devices in-process");
    if (c->comm)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_comm_attach: a communicator is already attached");
    RcclApi *api = rccl();
    if (!api) return fail(c, GRAPE_ERR_COMM, "grape_comm_attach: " + g_rccl.err);
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId nid;
    std::memcpy(&nid, id->bytes, sizeof(nid));
    NCCL_TRY(c, api->CommInitRank(&c->comm, n_ranks, nid, rank));
    c->comm_rank = rank;
    c->comm_size = n_ranks;
    return GRAPE_OK;
}

// ---- one process per GPU without RCCL: mailboxes exchanged through HIP IPC handles (ABI v4) -----------------------------------
extern "C" int grape_ipc_export(grape_ctx *c, int32_t n_ranks, grape_ipc_handle *out)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!out || n_ranks < 1 || n_ranks > grape::kMaxShards)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_ipc_export: n_ranks must be in 1..8, out non-null");
    if (c->is_group)
        return fail(c, GRAPE_ERR_UNSUPPORTED, "grape_ipc_export: the context already spans devices in-process");
    if (c->comm || c->ipc_ranks > 1)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_ipc_export: a communicator is already attached");
    static_assert(sizeof(grape_ipc_handle) >= sizeof(hipIpcMemHandle_t), "grape_ipc_handle must hold a hipIpcMemHandle_t");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->d_mbox && c->ipc_alloc_ranks != n_ranks) {
        (void)hipFree(c->d_mbox);
        c->d_mbox = nullptr;
    }
    const size_t bytes = grape::ipc_mailbox_bytes((int)((KN(c) + 1) * (size_t)c->B), n_ranks);     // (n_x <= max_batch rows per call)
    if (!c->d_mbox) {
        void *pm = nullptr;
        // fine-grained: stores and atomics of the peers' devices are visible to this device's loads (and the other way round)
        // without cache maintenance
        hipError_t e = hipExtMallocWithFlags(&pm, bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess || !pm) {
            (void)hipGetLastError();
            return fail(c, GRAPE_ERR_ALLOC, std::string("grape_ipc_export: fine-grained device allocation failed: ") + hipGetErrorString(e));
        }
        c->d_mbox = (double *)pm;
        c->ipc_alloc_ranks = n_ranks;
        c->bytes += bytes;
    }
    HIP_TRY(c, hipMemset(c->d_mbox, 0, bytes));
    HIP_TRY(c, hipDeviceSynchronize());
    c->ipc_evals = 0;
    c->ipc_count[0] = c->ipc_count[1] = 0;
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, c->d_mbox);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, GRAPE_ERR_COMM, std::string("grape_ipc_export: hipIpcGetMemHandle: ") + hipGetErrorString(e));
    }
    std::memset(out->bytes, 0, sizeof(out->bytes));
    std::memcpy(out->bytes, &h, sizeof(h));
    return GRAPE_OK;
}

extern "C" int grape_ipc_attach(grape_ctx *c, const grape_ipc_handle *handles, int32_t rank, int32_t n_ranks)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!handles || n_ranks < 1 || n_ranks > grape::kMaxShards || rank < 0 || rank >= n_ranks)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_ipc_attach: bad rank / n_ranks / handles");
    if (!c->d_mbox || c->ipc_alloc_ranks != n_ranks)
        return fail(c, GRAPE_ERR_NOT_READY, "grape_ipc_attach: call grape_ipc_export(ctx, n_ranks, ...) first");
    if (c->comm || c->ipc_ranks > 1)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_ipc_attach: a communicator is already attached");
    HIP_TRY(c, hipSetDevice(c->device));
    double *opened[grape::kMaxShards] = {};
    for (int j = 0; j < n_ranks; ++j) {
        if (j == rank) { opened[j] = c->d_mbox; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles[j].bytes, sizeof(h));
        void *pj = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&pj, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess || !pj) {
            (void)hipGetLastError();
            for (int i = 0; i < j; ++i)
                if (opened[i] && opened[i] != c->d_mbox) (void)hipIpcCloseMemHandle(opened[i]);
            return fail(c, GRAPE_ERR_COMM, "grape_ipc_attach: hipIpcOpenMemHandle (rank " + std::to_string(j) + "): " + hipGetErrorString(e));
        }
        opened[j] = (double *)pj;
    }
    for (int j = 0; j < n_ranks; ++j) c->ipc_mbox[j] = opened[j];
    c->ipc_ranks = n_ranks;
    c->comm_rank = rank;
    c->comm_size = n_ranks;
    return GRAPE_OK;
}

// smallest ensemble the vector flow of action_thin.hip is chosen for.  A member's two chains are strictly sequential in
// time: the flow takes the same 1.3 ms at C4's shape for 128 as for 1024 members (one wavefront per member, up to one per
// SIMD), where the expm + chain kernels spread (member, slice) pairs and time chunks over the whole device and are ahead
// for small ensembles.  Measured crossover at C4 (N = 1000): 256 members 1.30 vs 1.27 ms, 320 members 1.31 vs 1.52 ms.
// GRAPE_ACTION_MIN overrides.
static long act_min_units(const grape_ctx *c)
{
    if (const char *e = std::getenv("GRAPE_ACTION_MIN")) return std::atol(e);
    // 9 <= n <= 16: the expm kernel + chain_prop_kernel (one DPP matrix-vector product per slice on the stored propagators)
    // is ahead up to ~240 members since round 4's action_parts_kernel (profiles/r04_C4_flow_crossover.txt: 192 members 0.83
    // against 1.02 ms, 224: 1.04 / 1.02, 256: 1.03 / 1.04, 288: 1.26 / 1.02; round 3: ~350)
    if (c->NT == 2) return 3L * c->compute_units / 8;
    return c->NT == 1 && !c->pack2 ? 15L * c->compute_units / 16 : 9L * c->compute_units / 8;
}

extern "C" int grape_set_operators(grape_ctx *c, const double *A, const double *B, const double *Xi,
                                   const double *Xt, const double *wts)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!A || !B || !Xi || !Xt || !wts)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_set_operators: null argument");
    if (c->is_group) {                                       // hand every device its contiguous member block
        const size_t nn2 = 2 * (size_t)c->cfg.n * c->cfg.n, Kc = c->cfg.n_controls;
        const size_t nm2 = 2 * (size_t)c->cfg.n * (c->cfg.n_state_cols ? c->cfg.n_state_cols : c->cfg.n);
        for (size_t i = 0; i < c->sub.size(); ++i) {
            const size_t lo = (size_t)c->sub_lo[i];
            const int rc = grape_set_operators(c->sub[i], A + lo * nn2, B + lo * Kc * nn2, Xi + lo * nm2,
                                               Xt + lo * nm2, wts + lo);
            if (rc) {
                c->ops_set = false;                              // (some shards hold the new operators, some the old ones)
                return fail(c, rc, c->sub[i]->err);
            }
        }
        c->ops_set = true;
        c->evaluated = false;
        return GRAPE_OK;
    }
    // an upload that fails half way (allocation of a re-planned workspace, a copy) leaves the context NOT READY -- never with
    // the previous upload's flags over freed or partly rewritten buffers
    c->ops_set = false;
    DeviceGuard guard;
    HIP_TRY(c, hipSetDevice(c->device));
    // ordered behind the last evaluation BEFORE any device buffer is touched (the sparse lists, vectors and hoisted
    // generators below are rewritten in place): an evaluation may still run on a caller's stream
    if (c->dev_pending) {
        HIP_TRY(c, hipEventSynchronize(c->ev_dev));
        c->dev_pending = false;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t nn = (size_t)c->cfg.n * c->cfg.n, K = c->cfg.n_controls, E = c->cfg.n_ensemble;
    // n x m states with m < n (UnitaryGate-style left multiplication of m column vectors, e.g. m = 1:
    // a vectorised density matrix under Liouvillian superoperators, test/liou.jl:38-48): run zero-padded
    // to n x n.  Every product keeps the padding columns zero and both traces only add zeros, so F and g
    // are exactly those of the n x m problem (the kernels do n/m times the minimal chain work).
    std::vector<double> xi_pad, xt_pad;
    if (c->m != c->cfg.n) {
        const size_t nm = (size_t)c->cfg.n * c->m;
        try {
            xi_pad.assign(2 * E * nn, 0.0);
            xt_pad.assign(2 * E * nn, 0.0);
        } catch (...) {
            return fail(c, GRAPE_ERR_ALLOC, "grape_set_operators: out of host memory");
        }
        for (size_t k = 0; k < E; ++k) {
            std::memcpy(xi_pad.data() + 2 * k * nn, Xi + 2 * k * nm, sizeof(double) * 2 * nm);
            std::memcpy(xt_pad.data() + 2 * k * nn, Xt + 2 * k * nm, sizeof(double) * 2 * nm);
        }
        Xi = xi_pad.data();
        Xt = xt_pad.data();
    }
    std::vector<double> packed;
    try {
        packed.assign(c->family != 1 ? 2 * E * (K + 3) * nn : 2 * (size_t)c->EU * (2 * K + 3) * c->TSZ, 0.0);
    } catch (...) {
        return fail(c, GRAPE_ERR_ALLOC, "grape_set_operators: out of host memory");
    }
    if (c->family == 0) {
        // per member: [A' | B'_0..B'_{K-1} | Xi | Xt], column-major, with the generators already
        // multiplied by (-i dt): the kernel builds G = -i dt H directly and its gradient traces use
        // dt Im(tr(B M)) = Re(tr(B' M))
        const double dt = c->cfg.duration / c->cfg.n_slices;
        auto scaled = [&](double *dst, const double *src, size_t count) {
            for (size_t e = 0; e < count; ++e) {
                dst[2 * e] = dt * src[2 * e + 1];
                dst[2 * e + 1] = -dt * src[2 * e];
            }
        };
        for (size_t k = 0; k < E; ++k) {
            double *dst = packed.data() + 2 * k * (K + 3) * nn;
            scaled(dst, A + 2 * k * nn, nn);
            scaled(dst + 2 * nn, B + 2 * k * K * nn, K * nn);
            std::memcpy(dst + 2 * (1 + K) * nn, Xi + 2 * k * nn, sizeof(double) * 2 * nn);
            std::memcpy(dst + 2 * (2 + K) * nn, Xt + 2 * k * nn, sizeof(double) * 2 * nn);
        }
    } else if (c->family == 2) {
        // per member: [A | B_0..B_{K-1} | Xi | Xt] as they are (sweep_any.hip applies (-i dt) itself)
        for (size_t k = 0; k < E; ++k) {
            double *dst = packed.data() + 2 * k * (K + 3) * nn;
            std::memcpy(dst, A + 2 * k * nn, sizeof(double) * 2 * nn);
            std::memcpy(dst + 2 * nn, B + 2 * k * K * nn, sizeof(double) * 2 * K * nn);
            std::memcpy(dst + 2 * (1 + K) * nn, Xi + 2 * k * nn, sizeof(double) * 2 * nn);
            std::memcpy(dst + 2 * (2 + K) * nn, Xt + 2 * k * nn, sizeof(double) * 2 * nn);
        }
    } else {
        // per member: [A | B_c | B_c^T | Xi | Xt] as zero-padded D-layout dumps (tile.hpp)
        const int nd = c->cfg.n, NT = c->NT;
        // off: 0, or 8 for the second member of a block-diagonal pair (pack2, n <= 8)
        auto dump = [&](double *dst, const double *M, bool transpose, int off) {
            for (int I = 0; I < NT; ++I)
                for (int J = 0; J < NT; ++J)
                    for (int r = 0; r < 4; ++r)
                        for (int l = 0; l < 64; ++l) {
                            int row = 16 * I + 4 * r + (l >> 4) - off, col = 16 * J + (l & 15) - off;
                            if (row < 0 || col < 0 || row >= nd || col >= nd) continue;
                            if (transpose) std::swap(row, col);
                            const size_t o = 2 * ((size_t)((I * NT + J) * 4 + r) * 64 + l);
                            dst[o] = M[2 * (row + (size_t)nd * col)];
                            dst[o + 1] = M[2 * (row + (size_t)nd * col) + 1];
                        }
        };
        for (size_t k = 0; k < E; ++k) {
            const size_t unit = c->pack2 ? k / 2 : k;
            const int off = c->pack2 ? 8 * (int)(k & 1) : 0;
            double *dst = packed.data() + 2 * unit * (2 * K + 3) * c->TSZ;
            dump(dst, A + 2 * k * nn, false, off);
            for (size_t j = 0; j < K; ++j) {
                dump(dst + 2 * (1 + j) * c->TSZ, B + 2 * (k * K + j) * nn, false, off);
                dump(dst + 2 * (1 + K + j) * c->TSZ, B + 2 * (k * K + j) * nn, true, off);
            }
            dump(dst + 2 * (1 + 2 * K) * c->TSZ, Xi + 2 * k * nn, false, off);
            dump(dst + 2 * (2 + 2 * K) * c->TSZ, Xt + 2 * k * nn, false, off);
        }
    }
    // Data-flow choice: if every generator is Hermitian to rounding, every propagator is
    // unitary and the sweep can carry M_t = P_t' M_{t+1} P_t instead of storing X_t.
    bool herm = !(c->cfg.flags & (GRAPE_FLAG_FORCE_GENERAL | GRAPE_FLAG_KEEP_COSTATES)) &&
                c->cfg.gradient != GRAPE_GRADIENT_EXACT && !c->grid && c->family != 2;     // (sweep_grid.hip, sweep_any.hip: the reference's general flow only)
    const int n = c->cfg.n;
    for (size_t k = 0; k < E && herm; ++k)
        for (size_t m = 0; m < K + 1 && herm; ++m)
            herm = grape_host::hermitian_to_rounding((m == 0) ? A + 2 * k * nn : B + 2 * (k * K + (m - 1)) * nn, n);
    c->unitary = herm;
    // Exact gradient of a UnitaryGate problem on lane pairs: with Hermitian generators the UNITARY flow provides all the
    // exact-gradient kernel needs of the trajectory -- W_t = X_t L_{t+1}' = M_t P_t' per slice and tr M per member -- at
    // 0.09 instead of 0.16 ms for the debug flow that dumps X_t and L_t (C3), and the kernel behind it loads one matrix
    // per slice instead of three.  GRAPE_EXACT_W1=0 / GRAPE_FLAG_KEEP_COSTATES / FORCE_GENERAL keep the debug flow.
    c->exact_w1 = false;
    if (c->cfg.gradient == GRAPE_GRADIENT_EXACT && c->family == 0 && c->pair && c->d_zphi &&
        c->cfg.sys_type == GRAPE_UNITARY_GATE && c->m == c->cfg.n &&
        !(c->cfg.flags & (GRAPE_FLAG_FORCE_GENERAL | GRAPE_FLAG_KEEP_COSTATES)) && !env_off("GRAPE_EXACT_W1")) {
        bool hg = true;
        for (size_t k = 0; k < E && hg; ++k)
            for (size_t mm = 0; mm < K + 1 && hg; ++mm)
                hg = grape_host::hermitian_to_rounding((mm == 0) ? A + 2 * k * nn : B + 2 * (k * K + (mm - 1)) * nn, n);
        c->exact_w1 = hg;
        if (hg) c->unitary = true;
    }
    if (c->cfg.gradient == GRAPE_GRADIENT_EXACT && c->family == 0 && !(c->cfg.flags & GRAPE_FLAG_KEEP_COSTATES)) {
        // the W_t flow writes no costates: their E N n^2 16 B bytes go back (and return if a later upload needs the debug flow)
        const size_t cb = sizeof(double2) * c->ws_elems * ws_batch(c);
        if (c->exact_w1 && c->d_costates) {
            (void)hipFree(c->d_costates);
            c->d_costates = nullptr;
            c->bytes -= cb;
        } else if (!c->exact_w1 && !c->d_costates) {
            HIP_TRY(c, hipMalloc((void **)&c->d_costates, cb));
            c->bytes += cb;
        }
    }
    {                                                        // Hermitian initial / target operators (square states only)
        bool hs = c->m == c->cfg.n;
        for (size_t k = 0; k < E && hs; ++k)
            for (int which = 0; which < 2 && hs; ++which)
                hs = grape_host::hermitian_to_rounding((which ? Xt : Xi) + 2 * k * nn, n);
        c->herm_states = hs;
    }
    // Rank-one states in the single-tile family (n = 9..16): X_t = v_t v_t' (sandwich, Xi = v0 v0', Xt = wT wT') or
    // X_t = v_t (left multiplication of n x 1 states) -- the chain runs on vectors (sweep_thin.hip).
    // GRAPE_FLAG_FORCE_GENERAL / KEEP_COSTATES / the exact gradient keep the dense chain.
    // n = 5..8 (two members per tile) and n = 17..32: there is no expm-based vector chain; rank-one states count only where
    // the vector flow of action_thin.hip takes them (shared controls, an ensemble that fills the device -- decided right
    // here), else the dense chains stay.
    std::vector<double> vecs;
    const size_t VS = 16 * (size_t)c->NT;                    // complex entries of a (zero padded) vector
    const char *act_env = std::getenv("GRAPE_ACTION");
    bool ctrl_shared = true;                                 // the members' control operators are identical
    for (size_t k = 1; k < E && ctrl_shared; ++k)
        ctrl_shared = std::memcmp(B + 2 * k * K * nn, B, sizeof(double) * 2 * K * nn) == 0;
    // ... or member 0's times a scalar per member (B_g(k) = (1 + eps_k) B: amplitude inhomogeneity, src/problems.jl:33-41,
    // test/setup_tests.jl:31-32): every flow built on a per-slice control sum keeps it and scales it per member
    std::vector<double> ctrl_scale;
    c->ctrl_scaled = false;
    if (!ctrl_shared && c->family == 1 && !c->pack2 && !env_off("GRAPE_CTRL_SCALE") &&
        grape_host::controls_scaled(B, E, K, nn, ctrl_scale)) {
        c->ctrl_scaled = true;
        if (!c->d_ctrl_scale) {
            HIP_TRY(c, hipMalloc((void **)&c->d_ctrl_scale, sizeof(double) * E));
            c->bytes += sizeof(double) * E;
        }
        HIP_TRY(c, hipMemcpy(c->d_ctrl_scale, ctrl_scale.data(), sizeof(double) * E, hipMemcpyHostToDevice));
    }
    const bool ctrl_hoistable = ctrl_shared || c->ctrl_scaled;   // ONE control sum per slice serves every member
    c->ctrl_shared = ctrl_shared;
    c->any_sp_nnz = 0;
    if (c->family == 2 && c->cfg.n >= 17 && ctrl_shared && K >= 1 && K <= 256 && !env_on("GRAPE_NO_SPARSE")) {
        // size-generic family: shared control operators with few non-zeros (all of them together at most n^2 / 4: local drives on
        // seven qubits have 7 x 128 of 16 384) -- the H build and the gradient traces walk lists, not K dense operators per slice
        std::vector<int32_t> tidx, tptr, ectl, cptr, caddr;
        std::vector<double> ecoef, ccoef;
        const long nnz = grape_host::build_any_sparse(B, K, n, nn / 4, tidx, tptr, ectl, ecoef, cptr, caddr, ccoef);
        if (nnz > 0) {
            const size_t nt = tidx.size();
            const size_t ni = nt + (nt + 1) + (size_t)nnz + (K + 1) + (size_t)nnz, nc = 2 * (size_t)nnz;
            if (c->any_sp_i_cap < ni) {
                (void)hipFree(c->d_any_sp_i); c->d_any_sp_i = nullptr;
                c->bytes -= sizeof(int32_t) * c->any_sp_i_cap;
                c->any_sp_i_cap = 0;
                HIP_TRY(c, hipMalloc((void **)&c->d_any_sp_i, sizeof(int32_t) * ni));
                c->any_sp_i_cap = ni;
                c->bytes += sizeof(int32_t) * ni;
            }
            if (c->any_sp_c_cap < nc) {
                (void)hipFree(c->d_any_sp_c); c->d_any_sp_c = nullptr;
                c->bytes -= sizeof(double2) * c->any_sp_c_cap;
                c->any_sp_c_cap = 0;
                HIP_TRY(c, hipMalloc((void **)&c->d_any_sp_c, sizeof(double2) * nc));
                c->any_sp_c_cap = nc;
                c->bytes += sizeof(double2) * nc;
            }
            int32_t *di = c->d_any_sp_i;                     // [tidx | tptr | ectl | cptr | caddr]
            HIP_TRY(c, hipMemcpy(di, tidx.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(di + nt, tptr.data(), sizeof(int32_t) * (nt + 1), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(di + 2 * nt + 1, ectl.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(di + 2 * nt + 1 + nnz, cptr.data(), sizeof(int32_t) * (K + 1), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(di + 2 * nt + 1 + nnz + (K + 1), caddr.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->d_any_sp_c, ecoef.data(), sizeof(double2) * nnz, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->d_any_sp_c + nnz, ccoef.data(), sizeof(double2) * nnz, hipMemcpyHostToDevice));
            c->any_sp_ntouch = nt;
            c->any_sp_nnz = (size_t)nnz;
        }
    }
    const bool act_forced = act_env && act_env[0] == '1';
    // shared controls, or -- n <= 16 -- the members' own (at most six: a lane keeps its half rows of them in registers)
    const bool act_ok = (ctrl_hoistable || (c->NT == 1 && K <= 6)) && !(act_env && act_env[0] == '0') &&
                        c->cfg.n_slices <= 4096;             // (per-slice plans live in LDS)
    // (n = 33..64: grid_thin_kernel of sweep_grid.hip, sparse control operators only -- decided below the list build)
    bool thin = c->family == 1 && c->cfg.gradient != GRAPE_GRADIENT_EXACT && (!c->grid || c->NT >= 3) &&
                !(c->cfg.flags & (GRAPE_FLAG_FORCE_GENERAL | GRAPE_FLAG_KEEP_COSTATES)) && !env_on("GRAPE_NO_THIN");
    const bool act_only = c->NT == 2 || c->pack2;            // n = 5..8 (two members per tile) and n = 17..32: vector flow or dense chains
    if (thin && act_only)
        thin = act_ok && (act_forced || (long)E >= act_min_units(c));
    if (thin) {
        const bool sand = c->cfg.sys_type != GRAPE_UNITARY_GATE;
        vecs.assign(E * 4 * VS, 0.0);
        if (!sand) {
            thin = c->m == 1;
            for (size_t k = 0; k < E && thin; ++k)
                for (int i = 0; i < n; ++i)
                    for (int which = 0; which < 2; ++which) {
                        const double *M = (which ? Xt : Xi) + 2 * k * nn;      // zero padded to n x n: column 0
                        vecs[(k * 2 + which) * 2 * VS + 2 * i] = M[2 * i];
                        vecs[(k * 2 + which) * 2 * VS + 2 * i + 1] = M[2 * i + 1];
                    }
        } else {
            thin = c->m == n;
            for (size_t k = 0; k < E && thin; ++k)
                for (int which = 0; which < 2 && thin; ++which) {
                    const double *M = (which ? Xt : Xi) + 2 * k * nn;
                    if (!grape_host::factor_rank_one(M, n, vecs.data() + (k * 2 + which) * 2 * VS)) thin = false;
                }
        }
    }
    // ONE rank-one problem is latency-bound either way, and the dense chunked flows (two-level scan, 4-slice chunks) are
    // ahead of the vector chain's chunked mode there: 0.083 vs 0.100 ms per evaluation for C4's operators, N = 1000
    // (superseded where the propagator chain of action_thin.hip runs: it takes a chunked time axis too -- below)
    // Small ensembles of rank-one problems, down to ONE: expm kernel (P_t and P_t^T stored), chunk products, then
    // chain_prop_kernel twice -- on the chunk products for the vectors at the chunk boundaries, on the propagators with a
    // workgroup per (member, chunk).  GRAPE_DPP_CHUNKS=0 keeps sweep_thin.hip's chunked chain / the dense flows there.
    const char *dpp_env = std::getenv("GRAPE_THIN_DPP"), *hoist_env = std::getenv("GRAPE_HOIST"), *dppc_env = std::getenv("GRAPE_DPP_CHUNKS");
    const long dpp_min = dpp_env && dpp_env[0] == '1' ? 2 : (dpp_env && dpp_env[0] != '0' ? std::atol(dpp_env) : 80);
    bool dpp_small = thin && !c->grid && !act_only && !act_forced && !(dpp_env && dpp_env[0] == '0') && !(hoist_env && hoist_env[0] == '0') &&
                     !(dppc_env && dppc_env[0] == '0') && !env_on("GRAPE_NO_TP") && !env_on("GRAPE_THIN_SINGLE") &&
                     (long)E < std::min(dpp_min, std::getenv("GRAPE_DPP_SMALL_MAX") ? std::atol(std::getenv("GRAPE_DPP_SMALL_MAX")) : 41L) &&
                     c->cfg.n_slices >= 64 && !may_chunk(c);
    // (dense control operators included: their forms run on the matrix cores -- action_forms_mfma_kernel; with the vector-ALU
    // forms kernel, GRAPE_FORMS_VALU=1, this flow loses there: one problem / eight, K = 4: 0.090 / 0.141 ms against 0.086 / 0.127
    // of the flows it replaces, 0.084 / 0.113 with the matrix-core kernel)
    if (dpp_small && !(dppc_env && dppc_env[0] == '1') && env_on("GRAPE_FORMS_VALU")) {
        int rmax = 0;
        for (size_t q = 0; q < (ctrl_shared ? 1 : E) * K && rmax <= 6; ++q)
            for (int row = 0; row < n && rmax <= 6; ++row) {
                int cnt = 0;
                for (int col = 0; col < n; ++col)
                    cnt += (B[2 * (q * nn + row + (size_t)n * col)] != 0.0 || B[2 * (q * nn + row + (size_t)n * col) + 1] != 0.0) ? 1 : 0;
                rmax = std::max(rmax, cnt);
            }
        if (rmax > 6 || env_on("GRAPE_FORMS_DENSE"))
            dpp_small = false;
    }
    if (thin && !c->grid && E == 1 && c->cfg.n_slices >= 64 && !env_on("GRAPE_NO_TP") && !env_on("GRAPE_THIN_SINGLE") && !dpp_small)
        thin = false;
    c->thin = thin;
    {                                                        // Hermitian control operators?
        bool hb = true;
        for (size_t k = 0; k < E && hb; ++k)
            for (size_t m = 0; m < K && hb; ++m)
                hb = grape_host::hermitian_to_rounding(B + 2 * (k * K + m) * nn, n);
        c->herm_ctrl = hb;
    }
    {   // sparse control operators (Pauli-type controls): lists of (B_c[i][j], position of M[j][i]) per member and control
        bool sp = c->family == 1 && !c->pack2 && (!c->grid || c->NT >= 2) && K >= 1 && K <= 16 && !env_on("GRAPE_NO_SPARSE");
        // list length: the longest operator's non-zeros rounded up to whole wavefronts (64 for single Pauli strings up to five
        // qubits; 128 .. 256 for sums of a few of them -- a global drive sum_i X_i on five qubits has 160), while the K
        // lists fit the kernels' LDS budgets beside their images (K x length <= 1536 entries = 30 KB)
        int SM = 64;
        if (sp) {
            size_t most = 0;
            for (size_t m = 0; m < E * K; ++m) {
                size_t cnt = 0;
                for (size_t q = 0; q < nn; ++q)
                    cnt += (B[2 * (m * nn + q)] != 0.0 || B[2 * (m * nn + q) + 1] != 0.0) ? 1 : 0;
                most = std::max(most, cnt);
            }
            SM = (int)std::max<size_t>(64, (most + 63) / 64 * 64);
            if (const char *e = std::getenv("GRAPE_SPARSE_MAX")) sp = most <= (size_t)std::atol(e);      // (tests, A/B timing)
            if (SM > grape::kSparseMax || (size_t)K * SM > 1536) sp = false;
        }
        std::vector<double> coef;
        std::vector<int32_t> addr;
        if (sp)
            sp = grape_host::build_sparse_lists(B, E, K, n, c->grid ? 16 * c->NT + 2 : 16 * c->NT + 1, SM, coef, addr);   // (sweep_grid.hip's images: pitch 16 NT + 2)
        c->sparse_ctrl = sp;
        if (sp) {
            c->sp_nz = SM;
            const size_t need = E * K * (size_t)SM;
            if (c->sp_cap < need) {                          // (the list length may change between uploads)
                (void)hipFree(c->d_sp_coef); c->d_sp_coef = nullptr;
                (void)hipFree(c->d_sp_addr); c->d_sp_addr = nullptr;
                c->bytes -= (sizeof(double2) + sizeof(int32_t)) * c->sp_cap;
                c->sp_cap = 0;
                HIP_TRY(c, hipMalloc((void **)&c->d_sp_coef, sizeof(double2) * need));
                HIP_TRY(c, hipMalloc((void **)&c->d_sp_addr, sizeof(int32_t) * need));
                c->sp_cap = need;
                c->bytes += (sizeof(double2) + sizeof(int32_t)) * need;
            }
            HIP_TRY(c, hipMemcpy(c->d_sp_coef, coef.data(), sizeof(double) * coef.size(), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->d_sp_addr, addr.data(), sizeof(int32_t) * addr.size(), hipMemcpyHostToDevice));
        }
        if (c->grid && thin && !(sp && SM <= 256)) {             // grid_thin_kernel takes its forms from the lists: dense controls keep
            thin = false;                                        // the dense chain
            c->thin = false;
        }
    }
    {   // The expm kernels of prop_hoist.hip (n = 5..32, one member per tile).  hoist = 1: member-invariant control operators
        // (every BASELINE config, every reference test: B_gens = k -> [Sx, Sy], test/setup_tests.jl:32) -- the control sum
        // (-i dt) sum_c x[c,t] B_c is formed once per slice and evaluation by a pre-pass and a (member, slice) adds its own
        // A'_k = (-i dt) A_k.  hoist = 2: the members have their own control operators (amplitude-scaled controls of a
        // robustness ensemble, ...): the same kernels form the sum themselves.  16 x 16 with shared controls: from 8 units
        // on (below, the pre-pass launch costs more than it saves and the round-2 kernel stays).  GRAPE_HOIST=0 keeps
        // prop_tile_kernel, GRAPE_HOIST=1 forces the new kernels for any ensemble size.
        bool hz = c->family == 1 && c->cfg.gradient != GRAPE_GRADIENT_EXACT;
        const char *he = std::getenv("GRAPE_HOIST");
        if (he && he[0] == '0') hz = false;
        bool invariant = hz && !c->pack2 && ctrl_hoistable;      // (block-diagonal member pairs, n <= 8: always the in-kernel sum)
        if (c->grid && !invariant) hz = false;                   // sweep_grid.hip: the pre-pass or its own H build, nothing in between
        // (32 x 32: the new kernel is also the four-waves-per-propagator one -- single problems take it too)
        if (hz && invariant && !(he && he[0] == '1') && c->EU < 8 && c->NT == 1) {
            if (dpp_small) invariant = false;                    // (that flow needs this kernel's two dumps: the in-kernel sum, no pre-pass)
            else hz = false;
        }
        // (32 x 32, one to four units: the in-kernel sum instead of the pre-pass -- nobody shares its output there -- was
        // measured and is 1-3 % slower: 0.162 / 0.222 / 0.344 against 0.160 / 0.216 / 0.333 ms for 1 / 2 / 4 problems)
        c->hoist = hz ? (invariant ? 1 : 2) : 0;
        if (hz) {
            const double dt = c->cfg.duration / c->cfg.n_slices;
            const size_t TSZ = c->TSZ, ns = invariant ? 1 : 1 + K;      // norm bounds per unit: |A'| [, |B'_1| .. |B'_K|]
            std::vector<double> ha, hn;
            try {
                ha.assign(2 * (size_t)c->EU * TSZ, 0.0);
                hn.assign((size_t)c->EU * ns, 0.0);
            } catch (...) {
                return fail(c, GRAPE_ERR_ALLOC, "grape_set_operators: out of host memory");
            }
            const int nd = c->cfg.n, NT = c->NT;
            auto norm1 = [&](const double *M) {                  // max column sum of |re| + |im|, NaN-propagating
                double best = 0.0;
                for (int col = 0; col < nd; ++col) {
                    double cs = 0.0;
                    for (int row = 0; row < nd; ++row)
                        cs += std::fabs(M[2 * (row + (size_t)nd * col)]) + std::fabs(M[2 * (row + (size_t)nd * col) + 1]);
                    if (!(cs <= best)) best = cs;
                }
                return best;
            };
            for (size_t k = 0; k < E; ++k) {
                const double *M = A + 2 * k * nn;
                const size_t unit = c->pack2 ? k / 2 : k;            // pack2: two members per tile, block diagonal
                const int off = c->pack2 ? 8 * (int)(k & 1) : 0;
                double *dst = ha.data() + 2 * unit * TSZ;
                for (int col = 0; col < nd; ++col)
                    for (int row = 0; row < nd; ++row) {
                        const double re = dt * M[2 * (row + (size_t)nd * col) + 1], im = -dt * M[2 * (row + (size_t)nd * col)];
                        const int rr = row + off, cc2 = col + off;
                        const int I = rr >> 4, J = cc2 >> 4, r = (rr & 15) >> 2, l = 16 * (rr & 3) + (cc2 & 15);
                        const size_t o = 2 * ((size_t)((I * NT + J) * 4 + r) * 64 + l);
                        dst[o] = re;
                        dst[o + 1] = im;
                    }
                auto upd = [](double &slot, double v) { if (!(v <= slot)) slot = v; };      // the norm of a block-diagonal pair: the larger block's
                upd(hn[unit * ns], std::fabs(dt) * norm1(M) / grape::kTheta8);
                for (size_t cc = 0; cc + 1 < ns; ++cc)
                    upd(hn[unit * ns + 1 + cc], std::fabs(dt) * norm1(B + 2 * (k * K + cc) * nn) / grape::kTheta8);
            }
            const size_t gc_elems = (size_t)c->B * c->cfg.n_slices * TSZ;
            if (!c->d_ha) {
                c->bytes += sizeof(double2) * ((size_t)c->EU * TSZ + gc_elems) + sizeof(double) * ((size_t)c->EU * (1 + K) + (size_t)c->B * c->cfg.n_slices);
                HIP_TRY(c, hipMalloc((void **)&c->d_ha, sizeof(double2) * (size_t)c->EU * TSZ));
            }
            if (!c->d_ha_norm) HIP_TRY(c, hipMalloc((void **)&c->d_ha_norm, sizeof(double) * (size_t)c->EU * (1 + K)));
            if (!c->d_gc) HIP_TRY(c, hipMalloc((void **)&c->d_gc, sizeof(double2) * gc_elems));
            if (!c->d_gcn) HIP_TRY(c, hipMalloc((void **)&c->d_gcn, sizeof(double) * (size_t)c->B * c->cfg.n_slices));
            HIP_TRY(c, hipMemcpy(c->d_ha, ha.data(), sizeof(double) * ha.size(), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->d_ha_norm, hn.data(), sizeof(double) * hn.size(), hipMemcpyHostToDevice));
        }
    }
    {   // time-parallel chains: fewer units than wavefront slots (one wave per unit and chunk, 4 per CU).
        // Unitary flow: 3 S + C dependent products per evaluation (S slices per chunk, C chunks) instead of 3 N.
        // Rank-one chain: S dense chunk-product steps + 2 S vector steps + 2 C scan steps instead of 2 N vector steps;
        // its chunks start at multiples of 8 slices (the vector formats and prefetch rings follow the slice index).
        c->tp_C = c->tp_S = c->tp_G = c->tp_g = 0;
        const long units = (long)c->EU, N = c->cfg.n_slices, slots = 4L * c->compute_units;
        const bool general = !herm && !thin;                 // non-unitary propagators, full-rank states: prefix AND suffix products
        // Single-tile chains (n <= 16, full-rank states) hold ~124 registers: four waves fit a SIMD, and one wave per SIMD
        // hides none of its latencies -- 16 x 16 gate synthesis, N = 1000, chain kernel per member: 1024 members (one wave per
        // SIMD) 4.17 us, 4096 members (four) 2.68 us.  So up to 32 x CUs (unit, chunk) pairs there, and for the unitary flow
        // chunks up to 16 x CUs units (1024 members: 5.83 -> 3.98 ms per evaluation, 2048: 9.50 -> 7.96, 512: 2.43 -> 2.11;
        // n = 8, 4096 members: 5.69 -> 4.72).  The general flow's two-wave split chain stays ahead from 2 x CUs units on
        // (C4dense, 1024 members: 6.29 ms unchunked, 6.47 .. 6.80 chunked; 256 members 1.96 -> 1.69 with the longer cap).
        // GRAPE_TP_SLOTS=m: m x CUs pairs for every case (tuning).
        long pair_cap = (c->NT == 1 && !thin) ? 32L * c->compute_units : slots, small_cap = (c->NT == 1 && herm && !thin) ? pair_cap : slots;
        // (round 5, with the two-wave chain rewritten: C4dense-shaped general flow, chunked / two-wave chain, ms per evaluation:
        // 512 members 2.98 / 3.93, 576: 3.53 / 3.97, 640: 3.91 / 4.10, 704: 4.29 / 4.12, 768: 4.73 / 4.16 -- the chunked flow up
        // to 2.5 x CUs units)
        if (general && c->NT == 1) small_cap = 5L * c->compute_units;
        if (const char *e = std::getenv("GRAPE_TP_SLOTS")) pair_cap = small_cap = std::max(1L, std::atol(e)) * c->compute_units;
        const bool small = c->family == 1 && !c->grid && 2 * units <= (thin ? slots : small_cap) && !env_on("GRAPE_NO_TP") && !may_chunk(c);
        if (small && !thin && N >= 8) {
            // slices per chunk at the latency optimum: one-level scan 3 S + N / S dependent products (general flow),
            // two-level scan (unitary flow) 3 S + 2 sqrt(N / S); measured optima (tools/tp_sweep.py): 32 x 32, N = 2000:
            // 12 slices, 16 x 16, N = 1000: 4..5
            // general flow (a slice costs ~6 products there): (0.17 sqrt N)^(2/3), 4 at N = 1000
            long s_lat = general ? std::lround(std::pow(0.17 * std::sqrt((double)N), 2.0 / 3.0))
                                 : std::lround(std::cbrt((double)N / 9.0) * (c->NT == 2 ? 2.0 : 1.0));
            if (!general && c->NT == 2) {
                // the four-wave products of sweep_coop.hip (at most 3 x CUs (unit, chunk) pairs): shorter chunks pay while the
                // pairs still fit -- N = 2000, one unit: 167 / 250 / 334 chunks 0.181 / 0.174 / 0.193 ms
                const long s_coop = std::max(2L, std::lround(std::cbrt((double)N / 9.0) * 1.35));
                if (units * ((N + s_coop - 1) / s_coop) <= 3L * c->compute_units) s_lat = s_coop;
            }
            if (env_on("GRAPE_TP_ONE_LEVEL")) s_lat = std::lround(std::sqrt((double)N / 3.0));
            if (s_lat < 2) s_lat = 2;
            if (s_lat < 4 && N >= 64 && !env_on("GRAPE_TP_ONE_LEVEL")) s_lat = 4;
            long C = std::min(pair_cap / units, (N + s_lat - 1) / s_lat);
            if (const char *e = std::getenv("GRAPE_TP_CHUNKS")) C = std::atol(e);
            if (C > N / 2) C = N / 2;
            if (C >= 2) {
                c->tp_S = (int)((N + C - 1) / C);
                c->tp_C = (int)((N + c->tp_S - 1) / c->tp_S);
            }
        } else if (dpp_small) {
            // critical path: S / 4 + 3 dependent 16 x 16 products (0.6 us each; four waves per chunk) + S chain steps (0.3) +
            // C scan steps (0.2) -- S = sqrt(N / 1.75), 24 at N = 1000 (measured, one problem: 16 / 21 / 24 / 32 slices
            // 61.9 / 61.0 / 59.2 / 59.9 us); a workgroup's LDS ring lets two share a compute unit
            long s_lat = std::max(4L, std::lround(std::sqrt((double)N / 1.75)));
            long C = std::min(2L * c->compute_units / units, (N + s_lat - 1) / s_lat);
            if (const char *e = std::getenv("GRAPE_TP_CHUNKS")) C = std::atol(e);
            if (C > N / 2) C = N / 2;
            if (C >= 2) {
                c->tp_S = (int)((N + C - 1) / C);
                c->tp_C = (int)((N + c->tp_S - 1) / c->tp_S);
            }
        } else if (c->family == 1 && c->grid && !thin && c->cfg.gradient != GRAPE_GRADIENT_EXACT && N >= 16 &&
                   2 * units <= (long)c->compute_units && !env_on("GRAPE_NO_TP") && !may_chunk(c)) {
            // n = 33..64 (round 6; VERDICT r5 #4b): a 16-wave workgroup per member walked all N slices -- one 64 x 64 problem,
            // N = 500: 10.7 ms, the same as 16 of them.  Chunks of S slices: (S - 1) chunk products + 2 .. 4 C scan products +
            // 3 .. 6 S chain products in a row instead of 3 .. 6 N: S ~ sqrt(N / 2); at most two (member, chunk) workgroups per CU
            const long s_lat = std::max(4L, std::lround(std::sqrt((double)N / 2.0)));
            long C = std::min(2L * c->compute_units / units, (N + s_lat - 1) / s_lat);
            if (const char *e = std::getenv("GRAPE_TP_CHUNKS")) C = std::atol(e);
            if (C > N / 2) C = N / 2;
            if (C >= 2) {
                c->tp_S = (int)((N + C - 1) / C);
                c->tp_C = (int)((N + c->tp_S - 1) / c->tp_S);
            }
        } else if (small && thin && N >= 32 && 4 * units < slots) {      // C4's shape: 128 members 1.18 -> 0.87 ms, 256 members 1.39 -> 1.53
                                                                          // (and from half a device on, the fused forward pass)
            // measured optimum at C4's shape (N = 1000): 16..32 slices per chunk for 1..16 members (tools/single_open.py)
            long s_lat = 8 * std::max(1L, std::lround(std::sqrt(0.6 * (double)N) / 8.0));
            long C = std::min(slots / units, (N + s_lat - 1) / s_lat);
            if (const char *e = std::getenv("GRAPE_TP_CHUNKS")) C = std::atol(e);
            if (C >= 2) {
                const long S = ((N + C - 1) / C + 7) / 8 * 8;
                if ((N + S - 1) / S >= 2) {
                    c->tp_S = (int)S;
                    c->tp_C = (int)((N + S - 1) / S);
                }
            }
        }
        if (c->tp_C) {
            const size_t tsz = (size_t)c->NT * c->NT * 256, rows = (size_t)c->EU * c->B;
            const size_t dumps = ((general || dpp_small || c->grid) ? 2 : 1) * (size_t)c->tp_C;     // general flow: [Q_c | Q_c^T] and [R_c | U_c^T]
            auto ensure = [&](void **ptr, size_t *cap, size_t bytes) -> hipError_t {
                if (*cap >= bytes)
                    return hipSuccess;
                (void)hipFree(*ptr);
                *ptr = nullptr;
                c->bytes += bytes - *cap;
                *cap = 0;
                const hipError_t e = hipMalloc(ptr, bytes);
                if (e == hipSuccess)
                    *cap = bytes;
                return e;
            };
            HIP_TRY(c, ensure((void **)&c->d_tp_q, &c->tp_cap[0], sizeof(double2) * rows * tsz * dumps));
            HIP_TRY(c, ensure((void **)&c->d_tp_r, &c->tp_cap[1], sizeof(double2) * rows * tsz * dumps));
            HIP_TRY(c, ensure((void **)&c->d_tp_m, &c->tp_cap[2], sizeof(double2) * rows * tsz));
            HIP_TRY(c, ensure((void **)&c->d_tp_vec, &c->tp_cap[3], sizeof(double2) * rows * 32 * ((size_t)c->tp_C + 1)));
            HIP_TRY(c, ensure((void **)&c->d_tp_z, &c->tp_cap[4], sizeof(double) * rows * 128));
            // unitary flow, many chunks: two-level scan over groups of ~sqrt(C) chunks
            c->tp_G = c->tp_g = 0;
            if (!thin && !c->grid && c->tp_C >= 16 && !env_on("GRAPE_TP_ONE_LEVEL")) {
                c->tp_g = (int)std::lround(std::ceil(std::sqrt((double)c->tp_C)));
                c->tp_G = (c->tp_C + c->tp_g - 1) / c->tp_g;
                HIP_TRY(c, ensure((void **)&c->d_tp_a, &c->tp_cap[5], sizeof(double2) * rows * tsz * (general ? 4 : 2) * c->tp_G));
            }
        }
    }
    // Lists longer than a wavefront where they pay: the chain kernels hold one wave per SIMD, and with more than ~40 KB of
    // LDS per workgroup (images 21 KB at 32 x 32 + 20 B per list entry) only three of them fit a compute unit -- C5's shape
    // with two global drives (160 non-zeros, lists of 192): 1024 members 75.6 ms against 66.8 with the dense traces, ONE
    // problem 0.196 against 0.250 ms.  So: launches that leave LDS to spare (at most 3 workgroups per compute unit: single
    // problems, a handful of members), or lists that keep K x length within 896 entries.  GRAPE_SPARSE_MAX=n forces.
    if (c->sparse_ctrl && !c->grid && c->sp_nz > 64 && !std::getenv("GRAPE_SPARSE_MAX") && (size_t)K * c->sp_nz > 896 &&
        (long)c->EU * std::max(1, c->tp_C) > 3L * c->compute_units)
        c->sparse_ctrl = false;
    {   // Rank-one states with member-invariant control operators: the evaluation runs on vectors alone (action_thin.hip) --
        // exp(G_t) applied to the two chains' vectors by its Taylor series, no propagator formed or stored.  Ensembles that
        // fill the device (the chunked flows of small ensembles keep the expm kernel: they need the chunk PRODUCTS);
        // GRAPE_ACTION=0 keeps the expm + chain kernels, GRAPE_ACTION=1 forces the vector flow for any ensemble size.
        bool act = thin && act_ok && !c->grid;
        if (act && !act_only && !act_forced && (long)c->EU < act_min_units(c)) act = false;
        c->action = act;
        // ... and where that flow does not apply (ensembles below its threshold, more than six per-member controls): the
        // propagators of the expm kernel with the same DPP matrix-vector products, one per slice, and the same forms kernels
        // (chain_prop_kernel).  A member's chains are then N dependent products of ~0.3 us instead of sweep_thin.hip's
        // ~1 us ones.  GRAPE_THIN_DPP=0: sweep_thin.hip; =1: this flow for any ensemble the round-3 expm kernel serves.
        const char *de = std::getenv("GRAPE_THIN_DPP");
        // measured at C4's shape against sweep_thin.hip's chain (chunked time axis below 256 members): 64 members 0.42 vs 0.33 ms,
        // 96: 0.51 vs 0.60, 128: 0.60 vs 0.70, 256: 1.03 vs 1.27, 384: 1.50 vs 1.65, 512: 1.82 vs 1.64 (that chain reads P_t
        // once where the forward pass is fused into the expm kernel; this flow writes and reads it twice)
        // (needs the round-3 expm kernel, which writes both dumps: c->hoist != 0, i.e. at least 8 units)
        const bool dpp_chunked = dpp_small && c->tp_C > 1;       // (below dpp_min members: only with the chunked time axis)
        const bool dpp = thin && !c->grid && !act && !act_only && c->hoist != 0 && !may_chunk(c) && !(de && de[0] == '0') && ((long)E >= dpp_min || dpp_chunked) &&
                         ((de && de[0] == '1') || (long)E < 7L * c->compute_units / 4);
        c->thin_dpp = dpp;
        if (act || dpp) {
            if (!(dpp && dpp_chunked))
                c->tp_C = c->tp_S = c->tp_G = c->tp_g = 0;
            const double dt = c->cfg.duration / c->cfg.n_slices;
            const int nd = c->cfg.n;
            const size_t VV = VS * VS;
            c->act_shared = ctrl_shared;
            const size_t nb = ctrl_shared ? 1 : E;               // sets of control operators
            std::vector<double> aa, an, bb, bf, bn;
            try {
                aa.assign(2 * E * 2 * VV, 0.0);
                an.assign(E, 0.0);
                bb.assign(2 * nb * K * 2 * VV, 0.0);
                bf.assign(2 * nb * K * VV, 0.0);
                bn.assign(nb * K, 0.0);
            } catch (...) {
                return fail(c, GRAPE_ERR_ALLOC, "grape_set_operators: out of host memory");
            }
            // dst: [M' | M''] row-major, zero padded to VS x VS, M' = (-i dt) M (column-major n x n input); returns max(|M'|_1, |M'|_inf)
            auto images = [&](double *dst, const double *M) {
                double colsum[32] = {0}, rowsum[32] = {0};
                for (int col = 0; col < nd; ++col)
                    for (int row = 0; row < nd; ++row) {
                        const double re = dt * M[2 * (row + (size_t)nd * col) + 1], im = -dt * M[2 * (row + (size_t)nd * col)];
                        dst[2 * (row * VS + col)] = re;
                        dst[2 * (row * VS + col) + 1] = im;
                        dst[2 * (VV + col * VS + row)] = re;            // (M'')[col][row] = conj(M'[row][col])
                        dst[2 * (VV + col * VS + row) + 1] = -im;
                        colsum[col] += std::fabs(re) + std::fabs(im);
                        rowsum[row] += std::fabs(re) + std::fabs(im);
                    }
                double best = 0.0;
                for (int q = 0; q < 32; ++q) {
                    if (!(colsum[q] <= best)) best = colsum[q];
                    if (!(rowsum[q] <= best)) best = rowsum[q];
                }
                return best;
            };
            for (size_t k = 0; k < E; ++k)
                an[k] = images(aa.data() + 2 * k * 2 * VV, A + 2 * k * nn);
            int rmax = 0;                                        // most non-zeros in a row of any control operator
            for (size_t q = 0; q < nb * K; ++q) {                // q = k K + c (k = 0 only when the controls are shared)
                const double *Bq = B + 2 * q * nn;
                bn[q] = images(bb.data() + 2 * q * 2 * VV, Bq);
                for (int row = 0; row < nd; ++row) {
                    int cnt = 0;
                    for (int col = 0; col < nd; ++col) {
                        const double re = Bq[2 * (row + (size_t)nd * col)], im = Bq[2 * (row + (size_t)nd * col) + 1];
                        bf[2 * (q * VV + row * VS + col)] = re;
                        bf[2 * (q * VV + row * VS + col) + 1] = im;
                        if (re != 0.0 || im != 0.0) ++cnt;
                    }
                    rmax = std::max(rmax, cnt);
                }
            }
            // sparse rows (Pauli-type controls, Liouville-space commutators): (value, column) lists for the forms kernel
            c->act_R = env_on("GRAPE_FORMS_DENSE") ? 0 : rmax <= 0 ? 1 : rmax <= 4 ? rmax : rmax <= 6 ? 6 : 0;
            if (c->act_R) {
                const size_t R = (size_t)c->act_R;
                std::vector<double> bs(2 * nb * K * VS * R, 0.0);
                std::vector<int32_t> bo(nb * K * VS * R, 0);
                for (size_t q = 0; q < nb * K; ++q)
                    for (int row = 0; row < nd; ++row) {
                        size_t o = (q * VS + row) * R;
                        for (int col = 0; col < nd; ++col) {
                            const double re = bf[2 * (q * VV + row * VS + col)], im = bf[2 * (q * VV + row * VS + col) + 1];
                            if (re == 0.0 && im == 0.0) continue;
                            bs[2 * o] = re;
                            bs[2 * o + 1] = im;
                            bo[o] = 1024 * col;
                            ++o;
                        }
                    }
                (void)hipFree(c->d_act_bs); c->d_act_bs = nullptr;       // (R may change between uploads)
                (void)hipFree(c->d_act_bo); c->d_act_bo = nullptr;
                HIP_TRY(c, hipMalloc((void **)&c->d_act_bs, sizeof(double) * bs.size()));
                HIP_TRY(c, hipMalloc((void **)&c->d_act_bo, sizeof(int32_t) * bo.size()));
                HIP_TRY(c, hipMemcpy(c->d_act_bs, bs.data(), sizeof(double) * bs.size(), hipMemcpyHostToDevice));
                HIP_TRY(c, hipMemcpy(c->d_act_bo, bo.data(), sizeof(int32_t) * bo.size(), hipMemcpyHostToDevice));
            }
            const size_t g_elems = (size_t)c->B * c->cfg.n_slices * 3 * VV;   // (16 x 16: six planes of doubles per slice)
            if (!c->d_act_a) {
                c->bytes += sizeof(double2) * (E * 2 * VV + g_elems) + sizeof(double) * (E + (size_t)c->B * c->cfg.n_slices);
                HIP_TRY(c, hipMalloc((void **)&c->d_act_a, sizeof(double2) * E * 2 * VV));
            }
            if (!c->d_act_an) HIP_TRY(c, hipMalloc((void **)&c->d_act_an, sizeof(double) * E));
            (void)hipFree(c->d_act_b); c->d_act_b = nullptr;             // (one set or E sets: may change between uploads)
            (void)hipFree(c->d_act_bf); c->d_act_bf = nullptr;
            (void)hipFree(c->d_act_bn); c->d_act_bn = nullptr;
            HIP_TRY(c, hipMalloc((void **)&c->d_act_b, sizeof(double) * bb.size()));
            HIP_TRY(c, hipMalloc((void **)&c->d_act_bf, sizeof(double) * bf.size()));
            HIP_TRY(c, hipMalloc((void **)&c->d_act_bn, sizeof(double) * bn.size()));
            {                                                    // (buffers whose size follows the upload: keep workspace_bytes honest)
                const size_t now = sizeof(double) * (bb.size() + bf.size() + bn.size()) +
                                   (c->act_R ? (sizeof(double2) + sizeof(int32_t)) * nb * K * VS * (size_t)c->act_R : 0);
                c->bytes += now - c->act_var_bytes;
                c->act_var_bytes = now;
            }
            HIP_TRY(c, hipMemcpy(c->d_act_bn, bn.data(), sizeof(double) * bn.size(), hipMemcpyHostToDevice));
            // (one slice of slack in front and behind: the 16 x 16 chain kernel prefetches one slice past either end of the pulse)
            if (!c->d_act_g) {
                HIP_TRY(c, hipMalloc((void **)&c->d_act_g, sizeof(double2) * (g_elems + 2 * 3 * VV)));
                HIP_TRY(c, hipMemset(c->d_act_g, 0, sizeof(double2) * (g_elems + 2 * 3 * VV)));
            }
            if (!c->d_act_gn) HIP_TRY(c, hipMalloc((void **)&c->d_act_gn, sizeof(double) * (size_t)c->B * c->cfg.n_slices));
            if (dpp) {
                const size_t need = sizeof(double2) * (size_t)c->Ec * ((size_t)c->cfg.n_slices + 1) * VS * ws_batch(c);
                if (c->wrec_bytes < need) {
                    (void)hipFree(c->d_wrec);
                    c->d_wrec = nullptr;
                    c->bytes += need - c->wrec_bytes;
                    c->wrec_bytes = 0;
                    HIP_TRY(c, hipMalloc((void **)&c->d_wrec, need));
                    c->wrec_bytes = need;
                }
                const size_t need_t = sizeof(double2) * c->ws_elems * ws_batch(c);
                if (c->props_t_bytes < need_t) {
                    (void)hipFree(c->d_props_t);
                    c->d_props_t = nullptr;
                    c->bytes += need_t - c->props_t_bytes;
                    c->props_t_bytes = 0;
                    HIP_TRY(c, hipMalloc((void **)&c->d_props_t, need_t));
                    c->props_t_bytes = need_t;
                }
            }
            HIP_TRY(c, hipMemcpy(c->d_act_a, aa.data(), sizeof(double) * aa.size(), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->d_act_an, an.data(), sizeof(double) * an.size(), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->d_act_b, bb.data(), sizeof(double) * bb.size(), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(c->d_act_bf, bf.data(), sizeof(double) * bf.size(), hipMemcpyHostToDevice));
        }
    }
    {
        // The flow is known now.  Does it keep the forward states in a workspace array of its own (general flow, the exact
        // gradient's W_t)?  Then the chunk plan made at grape_create for the propagators alone is made again; when the members
        // per chunk change, the arrays are allocated afresh (the vector flows' records and the chunked time axis only ever
        // serve ensembles far below any budget).
        const bool full_states = !thin && (!herm || c->exact_w1);
        // (ws_invalid: an earlier re-plan freed the arrays and its allocation failed -- plan and allocate again, with the
        // costate wish it had recorded: the pointers no longer tell)
        const bool keep = c->ws_invalid ? c->ws_keep_costates : c->d_costates != nullptr;
        const int arrays = 1 + (keep ? 1 : 0) + (full_states ? 1 : 0);
        const int had = c->Ec, had_B = c->ws_B;
        if (arrays != c->ws_arrays || c->ws_invalid) {
            if (!plan_chunk(c, arrays))
                return fail(c, GRAPE_ERR_ALLOC, "grape_set_operators: the workspace of this data flow does not fit the budget of " +
                                                    std::to_string(c->ws_budget) + " bytes even for one workgroup's members");
            if (c->Ec != had || c->ws_B != had_B || c->ws_invalid) {
                const size_t new_b = sizeof(double2) * c->ws_elems * ws_batch(c);
                if (!c->ws_invalid) {
                    const size_t old_b = sizeof(double2) * c->ws_unit * (size_t)(c->family == 0 ? had : (c->pack2 ? (had + 1) / 2 : had)) * (size_t)had_B;
                    (void)hipFree(c->d_props); c->d_props = nullptr;
                    (void)hipFree(c->d_states); c->d_states = nullptr;
                    c->bytes -= c->states_bytes + old_b;
                    c->states_bytes = 0;
                    if (keep) { (void)hipFree(c->d_costates); c->d_costates = nullptr; c->bytes -= old_b; }
                    c->ws_invalid = true;
                    c->ws_keep_costates = keep;
                    c->ops_set = false;                       // no evaluation on freed arrays, whatever happens below
                }
                if (c->test_fail_replan > 0) {                // test hook (GRAPE_TEST_FAIL_REPLAN at grape_create)
                    --c->test_fail_replan;
                    return fail(c, GRAPE_ERR_ALLOC, "grape_set_operators: workspace allocation failed (GRAPE_TEST_FAIL_REPLAN)");
                }
                HIP_TRY(c, hipMalloc((void **)&c->d_props, new_b));
                c->bytes += new_b;
                if (keep && !c->d_costates) { HIP_TRY(c, hipMalloc((void **)&c->d_costates, new_b)); c->bytes += new_b; }
                c->ws_invalid = false;
            }
        }
        if (chunked(c) && (c->tp_C || c->thin_dpp))
            return fail(c, GRAPE_ERR_ALLOC, "grape_set_operators: a chunked time axis on a member-chunked workspace (budget too small for this ensemble)");
    }
    if (c->family == 2) {
        c->any_blocks = grape::any_prop_blocks(c->cfg.n, c->cfg.n_slices, (long)c->cfg.n_ensemble * c->B, c->compute_units);
        // (every propagator block owns six scratch matrices: at n = 2048 that is 400 MB per block -- no more blocks than 8 GB,
        // or a quarter of the workspace budget, pays for)
        const size_t per_block = sizeof(double2) * 6 * nn * (size_t)c->Ec * ws_batch(c);
        const size_t cap = std::min<size_t>((size_t)8 << 30, std::max<size_t>(c->ws_budget / 4, per_block));
        while (c->any_blocks > 1 && per_block * (size_t)c->any_blocks > cap)
            c->any_blocks = (c->any_blocks + 1) / 2;
        // Chunked time axis (round 6): the chain of a member is sequential in time and holds ONE workgroup (= one compute unit)
        // -- with fewer members than compute units the slices are cut into chunks (chunk products -> boundary scan -> a
        // workgroup per (member, chunk); sweep_any.hip phases 3-5): about one (member, chunk) workgroup per CU, S ~ sqrt(N / 2)
        // slices or more per chunk ((S - 1) + 2..4 C + 3..6 S dependent products instead of 3..6 N).
        c->tp_C = c->tp_S = 0;
        {
            const long units = (long)c->cfg.n_ensemble * c->B, N = c->cfg.n_slices;
            if (c->cfg.n >= 17 && N >= 16 && 2 * units <= (long)c->compute_units && !env_on("GRAPE_NO_TP") && !chunked(c) && !may_chunk(c)) {
                const long s_lat = std::max(4L, std::lround(std::sqrt((double)N / 2.0)));
                long C = std::min((long)c->compute_units / units, (N + s_lat - 1) / s_lat);
                if (const char *e = std::getenv("GRAPE_TP_CHUNKS")) C = std::atol(e);
                if (C > N / 2) C = N / 2;
                // three more matrices per (member, chunk) + a scratch set each: inside the same cap
                while (C >= 2 && (per_block + sizeof(double2) * 3 * nn * (size_t)c->Ec * ws_batch(c)) * (size_t)C > cap)
                    C = (C + 1) / 2;
                if (C >= 2) {
                    c->tp_S = (int)((N + C - 1) / C);
                    c->tp_C = (int)((N + c->tp_S - 1) / c->tp_S);
                }
            }
        }
        if (c->tp_C) {
            const size_t bytes = sizeof(double2) * nn * (size_t)c->cfg.n_ensemble * c->B * (size_t)c->tp_C;
            auto ensure = [&](double2 **ptr, size_t *capb) -> hipError_t {
                if (*capb >= bytes)
                    return hipSuccess;
                (void)hipFree(*ptr);
                *ptr = nullptr;
                c->bytes += bytes - *capb;
                *capb = 0;
                const hipError_t e = hipMalloc((void **)ptr, bytes);
                if (e == hipSuccess)
                    *capb = bytes;
                return e;
            };
            HIP_TRY(c, ensure(&c->d_tp_q, &c->tp_cap[0]));
            HIP_TRY(c, ensure(&c->d_tp_r, &c->tp_cap[1]));
            HIP_TRY(c, ensure(&c->d_tp_m, &c->tp_cap[2]));      // (family 2: the states at the chunks' starts)
        }
        const size_t need = per_block * (size_t)std::max(c->any_blocks, std::max(1, c->tp_C));
        if (c->scratch_bytes < need) {
            (void)hipFree(c->d_scratch);
            c->d_scratch = nullptr;
            c->bytes += need - c->scratch_bytes;
            c->scratch_bytes = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_scratch, need));
            c->scratch_bytes = need;
        }
    }
    if (thin) {
        c->unitary = false;                                  // the thin chain serves Hermitian generators as well
        if (!c->d_vecs) {
            c->bytes += sizeof(double) * E * 4 * VS;
            HIP_TRY(c, hipMalloc((void **)&c->d_vecs, sizeof(double) * E * 4 * VS));
        }
        HIP_TRY(c, hipMemcpy(c->d_vecs, vecs.data(), sizeof(double) * E * 4 * VS, hipMemcpyHostToDevice));
        const size_t rec = sizeof(double2) * (size_t)c->Ec * ((size_t)c->cfg.n_slices + 1) * VS * ws_batch(c);
        if (c->states_bytes < rec) {                         // the forward pass's vector records: N + 1 per member
            (void)hipFree(c->d_states);
            c->d_states = nullptr;
            c->bytes += rec - c->states_bytes;
            c->states_bytes = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_states, rec));
            c->states_bytes = rec;
        }
    } else if ((!herm || c->exact_w1) && c->states_bytes < sizeof(double2) * c->ws_elems * ws_batch(c)) {
        const size_t full = sizeof(double2) * c->ws_elems * ws_batch(c);
        (void)hipFree(c->d_states);
        c->d_states = nullptr;
        c->bytes += full - c->states_bytes;
        c->states_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_states, full));
        c->states_bytes = full;
    }
    HIP_TRY(c, hipMemcpy(c->d_ops, packed.data(), sizeof(double) * packed.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_wts, wts, sizeof(double) * E, hipMemcpyHostToDevice));
    c->ops_set = true;
    c->evaluated = false;
    return GRAPE_OK;
}

static TileParams tile_params(const grape_ctx *c, const double *d_x, int n_x = 1)
{
    TileParams p{};
    p.n_x = n_x;
    p.ops = c->d_ops;
    p.x = d_x;
    p.props = c->d_props;
    p.states = c->d_states;
    p.costates = c->d_costates;
    p.member_out = c->d_member_out;
    p.K = c->cfg.n_controls;
    p.N = c->cfg.n_slices;
    p.E = c->EU;
    p.E_plan = c->EU;                                        // (a member-chunked launch keeps the whole ensemble's flow decisions)
    p.E_members = c->cfg.n_ensemble;
    p.pack2 = c->pack2 ? 1 : 0;
    p.n = c->cfg.n;
    p.s_forced = c->cfg.expm_squarings;
    p.variant = c->cfg.variant;
    p.dt = c->cfg.duration / c->cfg.n_slices;
    p.unitary = c->unitary ? 1 : 0;
    p.herm_states = c->herm_states ? 1 : 0;
    p.thin = c->thin ? (c->thin_dpp ? 2 : 1) : 0;
    p.herm_ctrl = c->herm_ctrl ? 1 : 0;
    p.vecs = c->d_vecs;
    p.cus = c->compute_units;
    p.tp_chunks = c->tp_C;
    p.tp_S = c->tp_S;
    p.tp_q = c->d_tp_q;
    p.tp_r = c->d_tp_r;
    p.tp_m = c->d_tp_m;
    p.tp_z = c->d_tp_z;
    p.tp_vec = c->d_tp_vec;
    p.tp_groups = c->tp_C ? c->tp_G : 0;
    p.tp_gsize = c->tp_g;
    p.tp_a = c->d_tp_a;
    if (c->tp_C && ((!c->unitary && !c->thin) || c->thin_dpp || c->grid)) {   // general flow (always: sweep_grid.hip) / chunked propagator chain: second halves of the dump buffers
        const size_t half = (size_t)c->EU * c->B * c->tp_C * c->NT * c->NT * 256;
        p.tp_qt = c->d_tp_q + half;
        p.tp_u = c->d_tp_r + half;
    }
    p.sparse = c->sparse_ctrl ? 1 : 0;
    p.sp_nz = c->sp_nz;
    p.sp_coef = c->d_sp_coef;
    p.sp_addr = c->d_sp_addr;
    p.hoist = c->hoist;
    p.ha = c->d_ha;
    p.ha_norm = c->d_ha_norm;
    p.gc = c->d_gc;
    p.gcn = c->d_gcn;
    p.action = c->action ? 1 : 0;
    if (c->action) {                                         // the vector flow works on members, whatever the tile packing
        p.E = c->cfg.n_ensemble;
        p.E_plan = c->cfg.n_ensemble;
        p.pack2 = 0;
    }
    p.act_a = c->d_act_a;
    p.act_an = c->d_act_an;
    p.act_b = c->d_act_b;
    p.act_bf = c->d_act_bf;
    p.act_g = c->d_act_g ? c->d_act_g + 3 * (size_t)(c->cfg.n <= 16 ? 256 : 1024) : nullptr;
    p.act_gn = c->d_act_gn;
    p.act_R = (c->action || c->thin_dpp) ? c->act_R : 0;
    p.wrec = c->thin_dpp ? c->d_wrec : c->d_props;
    p.props_t = c->d_props_t;
    p.act_shared = c->act_shared ? 1 : 0;
    p.ctrl_scale = c->ctrl_scaled ? c->d_ctrl_scale : nullptr;
    p.ops_ref = c->d_ops;
    p.act_b_ref = c->d_act_b;
    p.act_bn = c->d_act_bn;
    p.act_bs = c->d_act_bs;
    p.act_bo = c->d_act_bo;
    return p;
}

// forward states X_t available to grape_get_trajectory without the debug flow?
static bool states_stored(const grape_ctx *c)
{
    if (c->exact_w1) return false;                           // exact gradient behind the unitary flow: W_t where the states would be
    if (c->d_costates) return true;                          // debug flow stores everything
    if (c->family == 2) return true;                         // sweep_any.hip stores every X_t
    if (c->family == 0 || c->unitary || c->thin) return false;   // fast small-n flows / unitary / rank-one flows rebuild them
    if (c->grid && !c->thin) return true;                    // sweep_grid.hip stores every X_t (rank-one states: vector records)
    return !grape::tile_chain_is_split(tile_params(c, nullptr), false);
}

// ONE problem whose flow ends in a forms kernel of action_thin.hip or in a unitary chain kernel (chain_tile_unitary_kernel,
// coop_chain_unitary_kernel: every launch path of launch_nt with p.unitary and no stored costates): that kernel closes the
// evaluation (no reduce launch)
static bool tile_folds_reduce(const grape_ctx *c, int n_x)
{
    if (c->family != 1 || c->cfg.n_ensemble != 1 || n_x != 1 || c->cfg.gradient == GRAPE_GRADIENT_EXACT || !c->direct_publish)
        return false;
    return c->action || c->thin_dpp || (c->unitary && !c->thin && !c->d_costates);
}

// a single-device evaluation of ONE control array whose last launch is one of the reduce kernels (reduce.hip): those can
// close grape_lbfgs' line-search probe themselves (DoneSignal::probe_out) -- the others keep lbfgs_select_kernel behind them
static bool eval_ends_in_reduce(const grape_ctx *c)
{
    if (c->is_group || c->comm || c->ipc_ranks > 1) return false;
    return c->family == 0 ? true : !tile_folds_reduce(c, 1);
}

// folds the `count` oldest outstanding event pairs into ev_total_ms (synchronises their stop events)
static int fold_events(grape_ctx *c, uint64_t count)
{
    for (uint64_t i = 0; i < count && c->ev_folded < c->ev_issued; ++i) {
        const size_t slot = (size_t)(c->ev_folded % kEventRing);
        HIP_TRY(c, hipEventSynchronize(c->ev[3 * slot + 2]));
        float ms = 0.f, first = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[3 * slot], c->ev[3 * slot + 2]));
        if (c->ev_has_mid[slot])
            HIP_TRY(c, hipEventElapsedTime(&first, c->ev[3 * slot], c->ev[3 * slot + 1]));
        if (c->smp_total.size() < 65536) {
            c->smp_total.push_back(ms);
            c->smp_first.push_back(first);
        }
        c->ev_total_ms += ms;
        c->ev_count += 1;
        c->ev_folded += 1;
    }
    return GRAPE_OK;
}

// one shard: sweep kernel(s) + the on-device ensemble reduction into d_fg; nothing is synchronised
static int enqueue_eval(grape_ctx *c, const double *d_x, double *d_fg, hipStream_t stream, int n_x = 1,
                        grape::DoneSignal done = grape::DoneSignal())
{
    if ((c->cfg.gradient == GRAPE_GRADIENT_EXACT || n_x > c->ws_B) && n_x > 1) {
        // the stored trajectory (every X_t and L_t of the debug flow) is ONE control array's, and so is a member-chunked
        // workspace: the arrays of a batch run one behind the other on the stream, each through the whole chain; the last
        // reduction publishes them all
        const size_t Qs = KN(c) + 1;
        for (int b = 0; b < n_x; ++b) {
            grape::DoneSignal db;
            if (b == n_x - 1 && done.flag) {
                db = done;
                db.stage_base = d_fg;
                db.n_total = (int)(Qs * n_x);
                db.mflags = nullptr;                         // (one flag per workgroup covers ONE array's outputs)
            }
            const int rc = enqueue_eval(c, d_x + (size_t)b * KN(c), d_fg + (size_t)b * Qs, stream, 1, db);
            if (rc) return rc;
        }
        return GRAPE_OK;
    }
    KernelLogScope log_scope(&c->kernel_log);
    c->mf_wait = 0;
    SweepParams p{};
    p.ops = c->d_ops;
    p.x = d_x;
    p.props = c->d_props;
    p.states = c->d_states;
    p.costates = c->d_costates;
    p.member_out = c->d_member_out;
    p.wts = c->d_wts;
    p.block_out = c->d_block_out;
    p.MPB = c->MPB;
    p.BPX = c->NB;
    p.n_x = n_x;
    p.sk_magic = (uint32_t)((1ull << 32) / ((uint64_t)c->S * c->cfg.n_controls)) + 1u;
    p.stamps = c->d_stamps;
    p.xg_scratch = c->d_xg_scratch;
    p.K = c->cfg.n_controls;
    p.N = c->cfg.n_slices;
    p.E = c->cfg.n_ensemble;
    p.S = c->S;
    p.LT = c->LT;
    p.s_forced = c->cfg.expm_squarings;
    p.variant = c->cfg.variant;
    p.dt = c->cfg.duration / c->cfg.n_slices;                 // src/GRAPE.jl:42
    p.dump_w1 = c->exact_w1 ? 1 : 0;
    p.zphi = c->d_zphi;
    // n x 1 states under left multiplication at n = 4 (vec(rho) of one qubit under a Liouvillian with a dissipator: the general
    // flow): the lane-pair kernel sweeps back on vectors (GRAPE_PAIR_VEC=0: on the padded matrices, as round 5)
    p.vec = (c->family == 0 && c->pair && c->cfg.n == 4 && c->m == 1 && c->cfg.sys_type == GRAPE_UNITARY_GATE &&
             c->cfg.gradient != GRAPE_GRADIENT_EXACT && !env_off("GRAPE_PAIR_VEC")) ? 1 : 0;
    bool timed = (c->cfg.flags & GRAPE_FLAG_TIME_KERNELS) != 0;
    if (timed && (c->cfg.flags & GRAPE_FLAG_TIME_SAMPLED) && (c->launches++ & 7) != 0)
        timed = false;
    hipEvent_t e0 = nullptr, e1 = nullptr, emid = nullptr;
    if (timed) {
        if (c->ev_issued - c->ev_folded == kEventRing) {   // ring full: fold the oldest pair (long finished)
            int rc = fold_events(c, 1);
            if (rc) return rc;
        }
        const size_t slot = (size_t)(c->ev_issued % kEventRing);
        e0 = c->ev[3 * slot];
        e1 = c->ev[3 * slot + 2];
        if (c->family == 1 || (c->family == 2 && c->any_blocks > 1)) emid = c->ev[3 * slot + 1];
        c->ev_has_mid[slot] = emid ? 1 : 0;
        c->ev_issued += 1;
        HIP_TRY(c, hipEventRecord(e0, stream));
    }
    const bool exact = c->cfg.gradient == GRAPE_GRADIENT_EXACT;
    if (exact) p.member_out = nullptr;                       // the sweep's first-order rows are not wanted
    // one workgroup holds the whole ensemble (single problems): its row is [G, F], no reduce launch
    // (done.stage_base: this evaluation is the last array of a batch that runs array by array -- its publication has to
    // copy out the WHOLE staging buffer, which only the reduce kernels do: no self-closing sweep / fold then)
    const bool direct = c->family == 0 && c->NB == 1 && n_x == 1 && !exact && c->direct_publish && !done.probe_out &&
                        !done.stage_base;
    const bool fold = !done.stage_base && tile_folds_reduce(c, n_x);
    if (direct) {
        p.direct_dst = done.flag && done.host_out ? done.host_out : d_fg;
        p.direct_flag = done.flag;
        p.direct_seq = done.seq;
    }
    // The members [lo, lo + cnt) through the sweep / tile kernels (and the exact-gradient kernel behind them).  Unchunked: ONE
    // call for the whole ensemble.  Member-chunked: per-member INPUTS and result ROWS are addressed from member lo on, the
    // workspace arrays from their start -- the kernels index both by the launch's own member number.
    const size_t nn = (size_t)c->cfg.n * c->cfg.n, Kc = (size_t)c->cfg.n_controls, Qrow = KN(c) + 1;
    auto launch_members = [&](int lo, int cnt) -> int {
        if (c->family == 0) {
            SweepParams q = p;
            q.ops = p.ops + (size_t)lo * (Kc + 3) * nn;
            q.wts = p.wts + lo;
            if (p.member_out) q.member_out = p.member_out + (size_t)lo * Qrow;
            q.block_out = p.block_out + (size_t)(lo / c->MPB) * Qrow;
            if (p.zphi) q.zphi = p.zphi + 2 * (size_t)lo;
            q.E = cnt;
            q.BPX = (cnt + c->MPB - 1) / c->MPB;
            const int mode = c->unitary ? 2 : (c->d_costates ? 1 : 0);
            if (c->pair)
                HIP_TRY(c, grape::launch_sweep_pair(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE, mode, q, stream));
            else
                HIP_TRY(c, grape::launch_sweep_small(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE, mode, q, stream));
            if (exact) {                                     // exact gradient + objective from the stored trajectory
                grape::ExactParams xq{};
                xq.ops = q.ops;
                xq.x = d_x;
                xq.props = c->d_props;
                xq.states = c->d_states;
                xq.costates = c->d_costates;
                xq.member_out = c->d_member_out + (size_t)lo * Qrow;
                xq.K = c->cfg.n_controls;
                xq.N = c->cfg.n_slices;
                xq.E = cnt;
                xq.S = c->S;
                xq.CH = c->CH;
                xq.s_forced = c->cfg.expm_squarings;
                xq.variant = c->cfg.variant;
                xq.objective = c->cfg.objective;
                xq.herm_states = c->herm_states ? 1 : 0;
                xq.w1_in = c->exact_w1 ? 1 : 0;
                xq.zphi = q.zphi;
                HIP_TRY(c, grape::launch_exact_grad(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE, xq, stream));
            }
            return GRAPE_OK;
        }
        if (c->family == 2) {
            grape::AnyParams a{};
            a.ops = c->d_ops + (size_t)lo * (Kc + 3) * nn;
            a.x = d_x;
            a.props = c->d_props;
            a.states = c->d_states;
            a.costates = c->d_costates;
            a.scratch = c->d_scratch;
            a.member_out = c->d_member_out + (size_t)lo * Qrow;
            a.n = c->cfg.n;
            a.K = c->cfg.n_controls;
            a.N = c->cfg.n_slices;
            a.E = cnt;
            a.E_rows = c->cfg.n_ensemble;
            a.n_x = n_x;
            a.sand = c->cfg.sys_type != GRAPE_UNITARY_GATE ? 1 : 0;
            a.s_forced = c->cfg.expm_squarings;
            a.variant = c->cfg.variant;
            a.dt = c->cfg.duration / c->cfg.n_slices;
            a.prop_blocks = c->any_blocks;
            a.ev_mid = lo == 0 ? emid : nullptr;
            a.shared_b = c->ctrl_shared ? c->d_ops + nn : nullptr;
            if (c->any_sp_nnz) {
                const size_t z = c->any_sp_nnz;
                const size_t nt = c->any_sp_ntouch;
                a.sp_tidx = c->d_any_sp_i;
                a.sp_tptr = a.sp_tidx + nt;
                a.sp_ectl = a.sp_tptr + (nt + 1);
                a.sp_cptr = a.sp_ectl + z;
                a.sp_caddr = a.sp_cptr + (Kc + 1);
                a.sp_ntouch = (int32_t)nt;
                a.sp_ecoef = c->d_any_sp_c;
                a.sp_ccoef = a.sp_ecoef + z;
            }
            a.tp_chunks = c->tp_C;
            a.tp_S = c->tp_S;
            a.tp_q = c->d_tp_q;
            a.tp_r = c->d_tp_r;
            a.tp_u = c->d_tp_m;
            HIP_TRY(c, grape::launch_sweep_any(a, stream));
            return GRAPE_OK;
        }
        TileParams t = tile_params(c, d_x, n_x);
        if (lo != 0 || cnt != c->cfg.n_ensemble) {
            const size_t lu = c->pack2 ? (size_t)lo / 2 : (size_t)lo, VS = 16 * (size_t)c->NT, VV = VS * VS;
            t.E = c->action ? cnt : (c->pack2 ? (cnt + 1) / 2 : cnt);
            t.E_members = cnt;
            t.ops += lu * (2 * Kc + 3) * c->TSZ;
            t.member_out += (size_t)lo * Qrow;
            if (t.vecs) t.vecs += (size_t)lo * 2 * VS;
            if (t.sp_coef) { t.sp_coef += lu * Kc * c->sp_nz; t.sp_addr += lu * Kc * c->sp_nz; }
            if (t.ha) { t.ha += lu * c->TSZ; t.ha_norm += lu * (c->hoist == 1 ? 1 : 1 + Kc); }
            if (t.act_a) { t.act_a += (size_t)lo * 2 * VV; t.act_an += lo; }
            if (t.ctrl_scale) t.ctrl_scale += lo;
            if (!c->act_shared && t.act_b) {
                t.act_bn += (size_t)lo * Kc;
                t.act_b += (size_t)lo * Kc * 2 * VV;
                t.act_bf += (size_t)lo * Kc * VV;
                if (t.act_bs) { t.act_bs += (size_t)lo * Kc * VS * c->act_R; t.act_bo += (size_t)lo * Kc * VS * c->act_R; }
            }
        }
        t.ev_mid = lo == 0 ? emid : nullptr;
        if (d_fg && fold) {
            t.fold_fg = d_fg;
            t.fold_wts = c->d_wts;
            t.fold_done = done;
        }
        if (c->grid)
            HIP_TRY(c, grape::launch_sweep_grid(c->NT, c->cfg.sys_type != GRAPE_UNITARY_GATE, c->d_costates != nullptr, t, stream));
        else
            HIP_TRY(c, grape::launch_sweep_tile(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE, c->d_costates != nullptr, t, stream));
        if (exact)
            HIP_TRY(c, grape::launch_exact_tile(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE, t, c->cfg.objective, stream));
        return GRAPE_OK;
    };
    for (int lo = 0; lo < c->cfg.n_ensemble; lo += c->Ec) {
        const int rc = launch_members(lo, std::min(c->Ec, c->cfg.n_ensemble - lo));
        if (rc) return rc;
    }
    if (timed) HIP_TRY(c, hipEventRecord(e1, stream));
    if (exact)
        HIP_TRY(c, grape::launch_reduce(c->d_member_out, c->d_wts, c->d_partial, d_fg, p.E, (int)(KN(c) + 1), c->ksplit,
                                        stream, done));
    else if (direct || fold)
        ;                                                    // the sweep / forms kernel has written [G, F] (and the flag)
    else if (c->family == 0) {
        if (done.mflags && done.flag && done.host_out && !done.probe_out)
            c->mf_wait = grape::reduce_rows_mflags((int)(KN(c) + 1), n_x);       // (the launch takes the same decision)
        HIP_TRY(c, grape::launch_reduce_rows(c->d_block_out, d_fg, c->NB, (int)(KN(c) + 1), n_x, stream, done));
    }
    else {
        // one weighted reduction per control array (each reuses d_partial, in stream order); with a host
        // destination only the last one publishes -- it copies out the whole staging buffer
        const size_t Qs = KN(c) + 1;
        for (int b = 0; b < n_x; ++b) {
            grape::DoneSignal db;
            if (b == n_x - 1 && done.flag) {
                db = done;
                if (!db.stage_base) {                        // (set already: the last array of a batch run array by array)
                    db.stage_base = d_fg;
                    db.n_total = (int)(Qs * n_x);
                }
            }
            HIP_TRY(c, grape::launch_reduce(c->d_member_out + (size_t)b * p.E * Qs, c->d_wts, c->d_partial,
                                            d_fg + (size_t)b * Qs, p.E, (int)Qs, c->ksplit, stream, db));
        }
    }
    c->evaluated = true;
    return GRAPE_OK;
}

// in-place (or send -> recv) all-reduce of one shard's [G, F] on `stream`
static int enqueue_allreduce(grape_ctx *c, const double *send, double *recv, hipStream_t stream, int n_x = 1)
{
    NCCL_TRY(c, g_rccl.AllReduce(send, recv, (KN(c) + 1) * (size_t)n_x, ncclDouble, ncclSum, c->comm, stream));
    return GRAPE_OK;
}

// One process per GPU, mailboxes attached (grape_ipc_attach): `row` (this rank's [G, F], complete on `stream`) summed over
// the ranks into `out` (device, nullable) and, with `done`, into the host buffer -- ipc_allreduce_kernel, no RCCL.
static int enqueue_ipc_allreduce(grape_ctx *c, const double *row, double *out, hipStream_t stream, grape::DoneSignal done,
                                 int n_x = 1)
{
    KernelLogScope log_scope(&c->kernel_log, true);
    grape::IpcParams ip{};
    ip.own_row = row;
    ip.Q = (int)((KN(c) + 1) * (size_t)n_x);                 // the n_x rows of a batched call travel as one
    ip.Qpad = (int)(((KN(c) + 1) * (size_t)c->B + 255) / 256 * 256);     // slot size: fixed by max_batch (every block column arrives every call)
    ip.rank = c->comm_rank;
    ip.n_ranks = c->ipc_ranks;
    ip.parity = (int)(c->ipc_evals & 1);
    ip.target = (unsigned long long)c->ipc_ranks * (c->ipc_count[ip.parity] + 1);
    for (int j = 0; j < c->ipc_ranks; ++j) ip.mbox[j] = c->ipc_mbox[j];
    // a poll is an s_sleep of ~2 k cycles (~1 us): give up after the context's timeout, 5 s at least / 120 s at most
    const double lim = std::min(120.0, std::max(5.0, c->timeout_s));
    ip.spin_limit = (long long)(lim * 1e6);
    ip.out = out;
    ip.done = done;
    ip.fail_word = c->d_h_flag + 1;                          // mapped host word: set when a block gives up (device path: nobody reads the flag)
    if (grape::launch_ipc_allreduce(ip, stream) != hipSuccess)
        return fail(c, GRAPE_ERR_HIP, "ipc_allreduce_kernel: launch failed");
    c->ipc_evals += 1;                                       // (counted once the exchange is in the stream: the peers count launches too)
    c->ipc_count[ip.parity] += 1;
    return GRAPE_OK;
}

// A mailbox exchange that gave up (a peer never arrived) has no completion flag to report through when it was issued by the
// device-pointer entry points: ipc_allreduce_kernel then sets the word behind the completion flag (and poisons its output
// with NaN).  Checked wherever the host next touches the context; sticky -- the peers are out of step for good.
static int ipc_check(grape_ctx *c)
{
    if (c->broken)
        return fail(c, GRAPE_ERR_COMM, "the context is unusable: an earlier multi-device evaluation failed part-way (see the error it returned)");
    if (c->ipc_ranks > 1 && c->h_flag) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (((volatile unsigned long long *)c->h_flag)[1] != 0) {
            c->broken = true;
            return fail(c, GRAPE_ERR_COMM, "a mailbox exchange issued through the device-pointer path gave up (a peer is gone or stuck): "
                                           "its [G, F] were poisoned with NaN; the context is unusable");
        }
    }
    return GRAPE_OK;
}

// Blocks until `stream` has drained: busy polls for a few hundred microseconds (the optimiser is
// sequential, so per-call latency is what the caller sees), then sleeps between polls; gives up
// after c->timeout_s (device presumed hung).
static int wait_stream(grape_ctx *c, hipStream_t stream)
{
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto elapsed = [&]() {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return (double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec);
    };
    long nap_ns = 20000;
    for (unsigned it = 0;; ++it) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return GRAPE_OK;
        if (q != hipErrorNotReady) HIP_TRY(c, q);
        if ((it & 63) != 63) continue;
        const double el = elapsed();
        if (el < 500e-6) continue;                          // spin phase
        if (el > c->timeout_s)
            return fail(c, GRAPE_ERR_TIMEOUT, "evaluation did not finish within " + std::to_string(c->timeout_s) +
                                                  " s (GRAPE_EVAL_TIMEOUT_S): device presumed hung");
        timespec nap{0, nap_ns};
        nanosleep(&nap, nullptr);
        if (nap_ns < 1000000) nap_ns *= 2;
    }
}

// x (host) -> where this shard's kernels will read it: straight into device memory through the BAR, or the mapped staging
// buffer that shard_issue's copy launch drains.  Host work only (no HIP call).
static void shard_stage_x(grape_ctx *s, const double *x, int n_x)
{
    const size_t kn = KN(s);
    if (s->x_upload == 2) {     // posted writes through the BAR; the doorbell of the launch that follows them orders them
        std::memcpy(s->d_x_bar, x, sizeof(double) * kn * n_x);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
#if defined(__x86_64__)
        __builtin_ia32_sfence();
#endif
    } else {
        std::memcpy(s->h_stage, x, sizeof(double) * kn * n_x);
    }
}

// the launches of one shard's evaluation of the staged x into `target`, on the shard's private stream
static int shard_issue(grape_ctx *s, int n_x, double *target, bool signal)
{
    HIP_TRY(s, hipSetDevice(s->device));
    if (s->dev_pending) {                                   // order behind the last grape_eval_device
        HIP_TRY(s, hipStreamWaitEvent(s->stream, s->ev_dev, 0));
        s->dev_pending = false;
    }
    const size_t kn = KN(s);
    const double *d_x = s->d_x;
    if (s->x_upload == 2)
        d_x = s->d_x_bar;
    else if (s->x_upload == 1)  // a small kernel pulls x out of the coherent mapped staging buffer
        HIP_TRY(s, grape::launch_copy(s->d_h_stage, s->d_x, (int)(kn * n_x), s->stream));
    else
        HIP_TRY(s, hipMemcpyAsync(s->d_x, s->h_stage, sizeof(double) * kn * n_x, hipMemcpyHostToDevice, s->stream));
    grape::DoneSignal done;
    if (signal) {                  // single GPU: the reduce kernel stages [G, F] in d_fg and its last workgroup
        done.counter = s->d_done_counter;          // writes them to the mapped host buffer + the completion flag
        done.flag = s->d_h_flag;
        done.seq = ++s->seq;
        done.host_out = target;
        if (s->mf_publish) done.mflags = s->d_h_flag + 8;
        target = s->d_fg;
    }
    return enqueue_eval(s, d_x, target, s->stream, n_x, done);
}

static int shard_enqueue_host(grape_ctx *s, const double *x, int n_x, double *target, bool signal)
{
    shard_stage_x(s, x, n_x);
    return shard_issue(s, n_x, target, signal);
}

// a group shard's share of grape_eval: its launches, then the event the sum on the first device waits for
static int shard_issue_group(grape_ctx *s)
{
    int rc = shard_issue(s, s->group ? s->group->group_nx : 1, s->d_fg, false);
    if (rc == GRAPE_OK && s->ev_done && hipEventRecord(s->ev_done, s->stream) != hipSuccess)
        rc = fail(s, GRAPE_ERR_HIP, "hipEventRecord failed");
    return rc;
}

// round 4: the same, ending in the shard's arrival (shard_arrive_kernel on its own stream): whichever shard's blocks arrive
// last sum the rows and publish -- no event, no wait on the first device's stream, no separate reduction launch
static int shard_launch_arrive(grape_ctx *s)
{
    KernelLogScope log_scope(&s->kernel_log, true);
    grape_ctx *g = s->group;
    grape::ArriveParams ap{};
    ap.rows.n = (int)g->sub.size();
    for (size_t i = 0; i < g->sub.size(); ++i) ap.rows.p[i] = g->sub[i]->d_fg;
    ap.Q = (int)((KN(g) + 1) * (size_t)g->group_nx);
    ap.arrive = g->d_arrive;
    ap.finished = g->d_arrive + ((KN(g) + 1) * (size_t)g->B + 255) / 256;
    ap.out = nullptr;
    ap.done = g->group_done;
    if (grape::launch_shard_arrive(ap, s->stream) != hipSuccess)
        return fail(s, GRAPE_ERR_HIP, "shard_arrive_kernel: launch failed");
    return GRAPE_OK;
}
static int shard_issue_group_arrive(grape_ctx *s)
{
    int rc = shard_issue(s, s->group->group_nx, s->d_fg, false);
    if (rc == GRAPE_OK) rc = shard_launch_arrive(s);
    return rc;
}

void GroupWorker::run()
{
    (void)hipSetDevice(shard->device);
    uint64_t seen = 0;
    timespec idle_since;
    clock_gettime(CLOCK_MONOTONIC, &idle_since);
    for (;;) {
        unsigned spins = 0;
        while (req.load(std::memory_order_acquire) == seen) {
            if (stop.load(std::memory_order_relaxed)) return;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((++spins & 4095) == 0) {                     // 2 ms without work: sleep until the next post
                timespec now;
                clock_gettime(CLOCK_MONOTONIC, &now);
                const double idle = (double)(now.tv_sec - idle_since.tv_sec) + 1e-9 * (double)(now.tv_nsec - idle_since.tv_nsec);
                if (idle > 2e-3) {
                    std::unique_lock<std::mutex> lk(mu);
                    asleep.store(true, std::memory_order_seq_cst);
                    cv.wait(lk, [&] { return req.load(std::memory_order_seq_cst) != seen || stop.load(); });
                    asleep.store(false, std::memory_order_seq_cst);
                }
            }
        }
        seen = req.load(std::memory_order_acquire);
        rc = job ? job(shard) : 0;
        done.store(seen, std::memory_order_release);
        clock_gettime(CLOCK_MONOTONIC, &idle_since);
    }
}

// Blocks until the final kernel has published sequence number s->seq in the host flag (see
// reduce.hip: signal_done): a plain load loop on coherent pinned memory, no runtime call on the fast
// path.  Falls back to the stream for error detection and for the timeout.
static int wait_flag(grape_ctx *s)
{
    volatile unsigned long long *flag = s->h_flag;
    const unsigned long long want = s->seq;
    const int mf = s->mf_wait;                               // > 0: one flag per workgroup of the final kernel (h_flag[8 ..])
    auto all_flags = [&]() {
        for (int i = 0; i < mf; ++i)
            if (flag[8 + i] != want) return false;
        return true;
    };
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto elapsed = [&]() {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return (double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec);
    };
    bool napped = false;
    auto done = [&](double el, bool first_look) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        // The estimate only ever learns from waits that SAW the flag unset after the nap (el is then the evaluation's
        // duration to within one poll).  When the flag was already set on wake-up the nap overshot: halve the estimate
        // instead of feeding the nap back into it (one 50 ms hiccup used to cost ~200 slow calls).
        if (napped && first_look)
            s->eval_ema_s *= 0.5;
        else
            s->eval_ema_s = s->eval_ema_s > 0.0 ? 0.75 * s->eval_ema_s + 0.25 * el : el;
        return GRAPE_OK;
    };
    // evaluations known to take milliseconds (16 x 16 and larger): sleep through ~90 % of the expected
    // duration in one go instead of burning a core, then spin for the rest
    if (s->eval_ema_s > 1e-3) {
        const double nap = 0.9 * s->eval_ema_s;
        timespec ts{(time_t)nap, (long)((nap - (double)(time_t)nap) * 1e9)};
        nanosleep(&ts, nullptr);
        napped = true;
    }
    const double spin_until = s->eval_ema_s > 1e-3 ? 1.3 * s->eval_ema_s : 500e-6;
    long nap_ns = 20000;
    for (unsigned it = 0;; ++it) {
        const unsigned long long seen = mf ? (all_flags() ? want : 0ull) : *flag;
        if (seen == want)
            return done(elapsed(), it == 0);
        if (seen == (want | grape::kSeqFailed))
            return fail(s, GRAPE_ERR_COMM, "the ranks' rows did not all arrive (mailbox exchange gave up): a peer is gone or stuck");
        if ((it & 1023) != 1023) continue;
        const double el = elapsed();
        if (el < spin_until) continue;                      // spin phase
        hipError_t q = hipStreamQuery(s->stream);           // a failed kernel never publishes: ask the runtime
        if (q != hipSuccess && q != hipErrorNotReady) HIP_TRY(s, q);
        if (q == hipSuccess && s->group && s->group->peer_all && *flag != want) {
            // arrive-and-sum: the LAST shard to arrive publishes, from its own stream -- this one being idle says nothing
            // before every other shard's stream is idle too
            for (grape_ctx *o : s->group->sub) {
                if (o == s) continue;
                const hipError_t qo = hipStreamQuery(o->stream);
                if (qo != hipSuccess && qo != hipErrorNotReady) HIP_TRY(s, qo);
                if (qo == hipErrorNotReady) { q = qo; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        if (q == hipSuccess && (mf ? !all_flags() : *flag != want))
            return fail(s, GRAPE_ERR_HIP, "evaluation finished without publishing its completion flag");
        if (el > s->timeout_s)
            return fail(s, GRAPE_ERR_TIMEOUT, "evaluation did not finish within " + std::to_string(s->timeout_s) +
                                                  " s (GRAPE_EVAL_TIMEOUT_S): device presumed hung");
        timespec nap{0, nap_ns};
        nanosleep(&nap, nullptr);
        if (nap_ns < 200000) nap_ns *= 2;
    }
}

static int group_fail(grape_ctx *g, grape_ctx *s, int rc) { return fail(g, rc, s->err); }

static int enqueue_peer_sum(grape_ctx *g, double *target, hipStream_t lead_stream, bool shard0_on_lead,
                            grape::DoneSignal done, bool record = true, int n_x = 1);

// device pointers in, device pointers out, nothing synchronised: n_x control arrays (n_x > 1: grape_eval_batch_device) on a
// single device, an attached communicator / mailbox exchange, or an in-process group
// (`done`: single-device contexts without a communicator only -- grape_lbfgs' fused probe, see eval_ends_in_reduce)
static int eval_device_impl(grape_ctx *c, const double *d_x, double *d_fg, void *stream, int n_x,
                            grape::DoneSignal done = grape::DoneSignal())
{
    hipStream_t st = (hipStream_t)stream;
    if (int rc0 = ipc_check(c)) return rc0;
    if (!c->is_group) {
        HIP_TRY(c, hipSetDevice(c->device));
        int rc = enqueue_eval(c, d_x, d_fg, st, n_x, c->ipc_ranks > 1 || c->comm ? grape::DoneSignal() : done);
        if (rc) return rc;
        if (c->ipc_ranks > 1) {
            rc = enqueue_ipc_allreduce(c, d_fg, d_fg, st, grape::DoneSignal(), n_x);
            if (rc) return rc;
        } else if (c->comm) {
            rc = enqueue_allreduce(c, d_fg, d_fg, st, n_x);
            if (rc) return rc;
        }
        HIP_TRY(c, hipEventRecord(c->ev_dev, st));
        c->dev_pending = true;
        return GRAPE_OK;
    }
    // group: d_x, d_fg live on the first device; fan x out, evaluate every shard, one grouped all-reduce
    grape_ctx *s0 = c->sub[0];
    const size_t bytes = sizeof(double) * KN(c) * (size_t)n_x;
    HIP_TRY(c, hipSetDevice(s0->device));
    if (c->sub.size() > 1) HIP_TRY(c, hipEventRecord(s0->ev_dev, st));       // x is ready at this point of `st`
    int rc = enqueue_eval(s0, d_x, s0->d_fg, st, n_x);
    if (rc) return group_fail(c, s0, rc);
    for (size_t i = 1; i < c->sub.size(); ++i) {
        grape_ctx *s = c->sub[i];
        HIP_TRY(c, hipSetDevice(s->device));
        HIP_TRY(c, hipStreamWaitEvent(s->stream, s0->ev_dev, 0));
        HIP_TRY(c, hipMemcpyPeerAsync(s->d_x, s->device, d_x, s0->device, bytes, s->stream));
        rc = enqueue_eval(s, s->d_x, s->d_fg, s->stream, n_x);
        if (rc) return group_fail(c, s, rc);
    }
    if (c->peer_sum) {
        rc = enqueue_peer_sum(c, d_fg, st, true, grape::DoneSignal(), true, n_x);
        if (rc) return rc;
        HIP_TRY(c, hipSetDevice(s0->device));
        HIP_TRY(c, hipEventRecord(s0->ev_dev, st));
        s0->dev_pending = true;
        c->evaluated = true;
        return GRAPE_OK;
    }
    NCCL_TRY(c, g_rccl.GroupStart());
    for (size_t i = 0; i < c->sub.size(); ++i) {
        grape_ctx *s = c->sub[i];
        const ncclResult_t r = g_rccl.AllReduce(s->d_fg, i == 0 ? d_fg : s->d_fg, (KN(c) + 1) * (size_t)n_x, ncclDouble, ncclSum,
                                                s->comm, i == 0 ? st : s->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail(c, GRAPE_ERR_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
        }
    }
    NCCL_TRY(c, g_rccl.GroupEnd());
    HIP_TRY(c, hipSetDevice(s0->device));
    HIP_TRY(c, hipEventRecord(s0->ev_dev, st));
    s0->dev_pending = true;
    c->evaluated = true;
    return GRAPE_OK;
}

extern "C" int grape_eval_device(grape_ctx *c, const double *d_x, double *d_fg, void *stream)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!d_x || !d_fg) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_device: null argument");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, "grape_eval_device: operators not set");
    return eval_device_impl(c, d_x, d_fg, stream, 1);
}

// peer_sum group: every shard's [G, F] (K*N+1 doubles in its d_fg) is copied to the first device behind that shard's
// own evaluation and ONE reduction kernel sums the rows in shard order into `target` (and, with `done`, publishes them
// to the host like the single-GPU path).  `shard0_on_lead`: shard 0 was evaluated on `lead_stream` itself.
static int enqueue_peer_sum(grape_ctx *g, double *target, hipStream_t lead_stream, bool shard0_on_lead,
                            grape::DoneSignal done, bool record, int n_x)
{
    grape_ctx *lead = g->sub[0];
    KernelLogScope log_scope(&lead->kernel_log, true);
    const size_t Q = (KN(g) + 1) * (size_t)n_x;              // (the n_x rows of a batched call are summed as one)
    grape::ShardRows rows{};
    rows.n = (int)g->sub.size();
    const bool direct = g->peer_direct;                      // every shard's row readable from the first device
    for (size_t i = 0; i < g->sub.size(); ++i) {
        grape_ctx *s = g->sub[i];
        rows.p[i] = s->d_fg;
        if (!(i == 0 && shard0_on_lead)) {
            if (record) {                                    // (grape_eval's issuing threads have recorded it themselves)
                HIP_TRY(g, hipSetDevice(s->device));
                HIP_TRY(g, hipEventRecord(s->ev_done, s->stream));
            }
            HIP_TRY(g, hipSetDevice(lead->device));
            HIP_TRY(g, hipStreamWaitEvent(lead_stream, s->ev_done, 0));
        }
    }
    HIP_TRY(g, hipSetDevice(lead->device));
    if (direct && g->sub.size() <= (size_t)grape::kMaxShards) {
        // one kernel on the first device reads the G rows where the shards left them (peer access) and publishes
        HIP_TRY(g, grape::launch_reduce_shards(rows, target, (int)Q, lead_stream, done));
        return GRAPE_OK;
    }
    for (size_t i = 0; i < g->sub.size(); ++i)               // no peer access between some pair: staged copies
        HIP_TRY(g, hipMemcpyPeerAsync(g->d_gather + i * Q, lead->device, g->sub[i]->d_fg, g->sub[i]->device, sizeof(double) * Q, lead_stream));
    HIP_TRY(g, grape::launch_reduce_rows(g->d_gather, target, (int)g->sub.size(), (int)Q, 1, lead_stream, done));
    return GRAPE_OK;
}

// host -> device(s) -> host: the (F, G, x) closure body.  n_x > 1: grape_eval_batch.
static int eval_host(grape_ctx *c, int n_x, const double *x, double *F, double *G, const char *who)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!x) return fail(c, GRAPE_ERR_INVALID_ARG, std::string(who) + ": x is null");
    if (n_x < 1 || n_x > c->B)
        return fail(c, GRAPE_ERR_INVALID_ARG, std::string(who) + ": n_x must be in 1..grape_config.max_batch");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, std::string(who) + ": operators not set");
    const size_t kn = KN(c), Q = kn + 1;
    grape_ctx *lead = c->is_group ? c->sub[0] : c;
    int rc = ipc_check(c);
    if (rc) return rc;
    if (!c->is_group && !c->comm && c->ipc_ranks <= 1) {
        // single GPU: the final reduce kernel writes its result straight into mapped pinned host
        // memory (no D2H copy node) and the host polls the stream
        rc = shard_enqueue_host(c, x, n_x, c->d_h_fg, true);
        if (rc) return rc;
    } else {
        if (c->is_group) {
            // x into every shard's buffer first, then all shards issue AT ONCE: shard 0 from this thread, the others from
            // their own issuing threads (GroupWorker) -- the last shard starts one launch latency behind the first one,
            // not G of them
            timespec ts0, ts1, ts2, ts3;
            clock_gettime(CLOCK_MONOTONIC, &ts0);
            for (grape_ctx *s : c->sub) shard_stage_x(s, x, n_x);
            c->group_nx = n_x;
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            const bool arrive = c->peer_sum && c->d_arrive && c->peer_all;
            if (arrive) {                                    // how this evaluation is published: the same for every shard
                c->group_done = grape::DoneSignal();
                c->group_done.flag = lead->d_h_flag;
                c->group_done.seq = ++lead->seq;
                c->group_done.host_out = lead->d_h_fg;
            }
            for (size_t i = 1; i < c->sub.size(); ++i) c->sub[i]->worker->post(arrive ? shard_issue_group_arrive : shard_issue_group);
            rc = arrive ? shard_issue_group_arrive(lead) : shard_issue(lead, n_x, lead->d_fg, false);
            int rc_w = GRAPE_OK;
            grape_ctx *bad = nullptr;
            bool worker_hung = false;
            for (size_t i = 1; i < c->sub.size(); ++i) {
                const int r = c->sub[i]->worker->wait(c->timeout_s);
                if (r == GRAPE_ERR_TIMEOUT) { worker_hung = true; continue; }   // (the shard's own error text belongs to its thread: not touched here)
                if (r && !rc_w) { rc_w = r; bad = c->sub[i]; }
            }
            if (worker_hung) {
                // a thread is still issuing on its shard: nothing can be drained or reset safely -- the group is retired
                c->broken = true;
                return fail(c, GRAPE_ERR_TIMEOUT, "grape_eval: a shard's issuing thread did not answer within the timeout; the context is unusable");
            }
            if ((rc || rc_w) && arrive) {
                // Some shards have arrived, others never will: the counters would stay below G and the NEXT evaluation's first
                // arrivals would read as the last ones (rows summed before they exist).  Drain every shard, then
                // zero the counters (the `finished` word included); if that cannot be done the group is retired.
                bool clean = true;
                for (grape_ctx *s : c->sub)
                    clean = clean && hipSetDevice(s->device) == hipSuccess && wait_stream(s, s->stream) == GRAPE_OK;
                const size_t nb = ((KN(c) + 1) * (size_t)c->B + 255) / 256 + 1;
                clean = clean && hipSetDevice(lead->device) == hipSuccess &&
                        hipMemset(c->d_arrive, 0, sizeof(unsigned) * nb) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
                if (!clean) { (void)hipGetLastError(); c->broken = true; }
            }
            if (rc) return group_fail(c, lead, rc);
            if (rc_w) return group_fail(c, bad, rc_w);
            clock_gettime(CLOCK_MONOTONIC, &ts2);
            if (arrive) {
                // nothing left to issue: the last blocks to arrive sum and publish
            } else if (c->peer_sum) {
                grape::DoneSignal done;                      // the reduction kernel publishes like the single-GPU path
                done.counter = lead->d_done_counter;
                done.flag = lead->d_h_flag;
                done.seq = ++lead->seq;
                done.host_out = lead->d_h_fg;
                rc = enqueue_peer_sum(c, lead->d_fg, lead->stream, true, done, false, n_x);
                if (rc) return rc;
            } else {
                NCCL_TRY(c, g_rccl.GroupStart());
                for (grape_ctx *s : c->sub) {
                    const ncclResult_t r = g_rccl.AllReduce(s->d_fg, s->d_fg, Q * (size_t)n_x, ncclDouble, ncclSum, s->comm, s->stream);
                    if (r != ncclSuccess) {
                        (void)g_rccl.GroupEnd();
                        return fail(c, GRAPE_ERR_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
                    }
                }
                NCCL_TRY(c, g_rccl.GroupEnd());
            }
            clock_gettime(CLOCK_MONOTONIC, &ts3);
            auto dt = [](const timespec &a, const timespec &b) { return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec); };
            c->group_tm[0] += dt(ts0, ts1);
            c->group_tm[1] += dt(ts1, ts2);
            c->group_tm[2] += dt(ts2, ts3);
            c->group_tm_n += 1;
            c->group_tm[4] -= (double)ts0.tv_sec + 1e-9 * (double)ts0.tv_nsec;      // (+ the end time below)
            c->group_tm[3] -= (double)ts3.tv_sec + 1e-9 * (double)ts3.tv_nsec;
        } else if (c->ipc_ranks > 1) {
            rc = shard_enqueue_host(c, x, n_x, c->d_fg, false);
            if (rc) return rc;
            grape::DoneSignal done;                          // the exchange kernel publishes like the single-GPU path
            done.counter = c->d_done_counter;
            done.flag = c->d_h_flag;
            done.seq = ++c->seq;
            done.host_out = c->d_h_fg;
            rc = enqueue_ipc_allreduce(c, c->d_fg, nullptr, c->stream, done, n_x);
            if (rc) return rc;
        } else {
            rc = shard_enqueue_host(c, x, n_x, c->d_fg, false);
            if (rc) return rc;
            rc = enqueue_allreduce(c, c->d_fg, c->d_fg, c->stream, n_x);
            if (rc) return rc;
        }
        if (!(c->is_group && c->peer_sum) && !(c->ipc_ranks > 1)) {
            HIP_TRY(c, hipSetDevice(lead->device));
            grape::DoneSignal done;
            done.counter = lead->d_done_counter;
            done.flag = lead->d_h_flag;
            done.seq = ++lead->seq;
            KernelLogScope log_scope(&lead->kernel_log, true);
            HIP_TRY(c, grape::launch_copy(lead->d_fg, lead->d_h_fg, (int)(Q * (size_t)n_x), lead->stream, done));
        }
    }
    rc = wait_flag(lead);                                    // [G, F] are in host memory
    if (c->is_group) {
        timespec te;
        clock_gettime(CLOCK_MONOTONIC, &te);
        const double tend = (double)te.tv_sec + 1e-9 * (double)te.tv_nsec;
        c->group_tm[3] += tend;
        c->group_tm[4] += tend;
    }
    if (rc) return c->is_group ? group_fail(c, lead, rc) : rc;
    if (c->is_group) {
        // the other shards' streams finish with the same all-reduce; drain them so that the next call
        // (and the accessors) find every device idle -- they are done or microseconds from it.  (Arrive-and-sum: the
        // publication itself says that every shard's kernels have run -- only exiting blocks of their arrive kernels can be
        // left, and seven hipStreamQuery round trips on streams that have JUST finished cost ~70 us.)
        for (size_t i = 1; i < c->sub.size() && !(c->peer_sum && c->d_arrive && c->peer_all); ++i) {
            rc = wait_stream(c->sub[i], c->sub[i]->stream);
            if (rc) return group_fail(c, c->sub[i], rc);
        }
        c->evaluated = true;
    }
    for (int b = 0; b < n_x; ++b) {
        if (G) std::memcpy(G + (size_t)b * kn, lead->h_fg + (size_t)b * Q, sizeof(double) * kn);
        if (F) F[b] = lead->h_fg[(size_t)b * Q + kn];
    }
    return GRAPE_OK;
}

extern "C" int grape_eval(grape_ctx *c, const double *x, double *F, double *G)
{
    return eval_host(c, 1, x, F, G, "grape_eval");
}

extern "C" int grape_eval_batch(grape_ctx *c, int32_t n_x, const double *x, double *F, double *G)
{
    return eval_host(c, n_x, x, F, G, "grape_eval_batch");
}

extern "C" int grape_eval_batch_device(grape_ctx *c, int32_t n_x, const double *d_x, double *d_fg, void *stream)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!d_x || !d_fg) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_batch_device: null argument");
    if (n_x < 1 || n_x > c->B)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_batch_device: n_x must be in 1..grape_config.max_batch");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, "grape_eval_batch_device: operators not set");
    return eval_device_impl(c, d_x, d_fg, stream, n_x);
}

// group accessors: the shard that owns `member`
static grape_ctx *owner_of(grape_ctx *c, int member, int *local)
{
    for (size_t i = 0; i < c->sub.size(); ++i)
        if (member >= c->sub_lo[i] && member < c->sub_lo[i + 1]) {
            *local = member - c->sub_lo[i];
            return c->sub[i];
        }
    return nullptr;
}

// ------------------------------------------------------------------------------------------
// Device-resident L-BFGS (lbfgs.hip): the optimiser loop of src/solve.jl:138 / :244.
//
// x, g, the (s, y) history, the direction and the trial points live on the (first) device; the host sees scalars only.
// Line search (grape_lbfgs_options.line_search):
//   0, 1  Hager-Zhang (LineSearches.jl's HagerZhang(), Optim's default for LBFGS(): delta 0.1, sigma 0.9, rho 5,
//         epsilon 1e-6, gamma 0.66, psi3 0.1; initial step 1 = InitialStatic) run HERE on phi(alpha), phi'(alpha) -- one
//         evaluation per trial step, whatever the context (groups of devices, attached communicators, no batching);
//         0 accepts the initial step when it already satisfies the (approximate) Wolfe conditions, 1 never does
//         (Optim's literal behaviour with InitialStatic: `mayterminate` stays false)
//   2     the factor-2 ladder of rounds 1-2: `probes` step lengths per batched launch (single-device contexts)
// A Hager-Zhang search that cannot bracket (the reference's UnitaryGate gradient is not the derivative of its figure
// of merit) hands that iteration to the ladder.
namespace {

struct LbfgsRun {
    grape_ctx *c = nullptr, *lead = nullptr;      // lead: the context that owns the vectors, the stream and the host flag
    grape::LbfgsState st{};
    double *h_sc = nullptr;
    int evals = 0;
    bool hung = false;                            // a wait timed out: the device is presumed hung, nothing is synchronised any more
    bool fused_probe = false;                     // phi / phi' of a trial step come from the evaluation's reduce kernel
    bool mb = false;                              // accepted steps run on many workgroups (lbfgs_dots_kernel + lbfgs_step_mb_kernel)

    grape::DoneSignal signal()
    {
        grape::DoneSignal d;
        d.flag = lead->d_h_flag;
        d.seq = ++lead->seq;
        return d;
    }
    int wait()
    {
        int rc = wait_flag(lead);
        if (rc == GRAPE_ERR_TIMEOUT) hung = true;
        if (!rc) rc = ipc_check(c);                   // an exchange of the device path that gave up (its [G, F] are NaN)
        return rc ? (c->is_group ? group_fail(c, lead, rc) : rc) : GRAPE_OK;
    }
    // [G, F] of the n_x control arrays in st.xt -> st.fgt, on the lead stream, nothing synchronised
    int evaluate(int n_x)
    {
        evals += n_x;
        // (groups: fan-out of x, every shard, the grouped all-reduce / peer sum; communicators: the exchange behind the sweep)
        return eval_device_impl(c, st.xt, st.fgt, lead->stream, n_x);
    }
    // trial slot 0: evaluate, publish phi, phi' (and phi'(0)) behind it -- nothing is waited for
    int probe()
    {
        if (fused_probe) {
            // the evaluation's own reduction publishes the scalars (same bits as lbfgs_select_kernel's): one launch less
            // on the dependent chain of every trial step
            grape::DoneSignal d = signal();
            d.counter = lead->d_done_counter;
            d.probe_dir = st.d;
            d.probe_sc = st.sc;
            d.probe_out = st.host_sc + 8;
            evals += 1;
            const int rc = eval_device_impl(c, st.xt, st.fgt, lead->stream, 1, d);
            return rc ? rc : dots();
        }
        int rc = evaluate(1);
        if (rc) return rc;
        HIP_TRY(c, hipSetDevice(lead->device));
        if (grape::launch_lbfgs_select(st, 1, lead->stream, signal(), 1) != hipSuccess)
            return fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed");
        return dots();
    }
    // every dot product an accepted step can need, behind the evaluation and off the critical path (nobody waits for it)
    int dots()
    {
        if (!mb) return GRAPE_OK;
        HIP_TRY(c, hipSetDevice(lead->device));
        if (grape::launch_lbfgs_dots(st, lead->stream) != hipSuccess)
            return fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed");
        return GRAPE_OK;
    }
    // phi(alpha), phi'(alpha) along the current direction; `have_trial`: slot 0 already holds x + alpha d
    int phi(double alpha, bool have_trial, double &f, double &df)
    {
        HIP_TRY(c, hipSetDevice(lead->device));
        if (!have_trial && grape::launch_lbfgs_trial(st, alpha, lead->stream) != hipSuccess)
            return fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed");
        int rc = probe();
        if (rc) return rc;
        rc = wait();
        if (rc) return rc;
        f = h_sc[8];
        df = h_sc[9];
        return GRAPE_OK;
    }
};

// LineSearches.jl HagerZhang on scalars.  Returns 0 with the accepted step (which is the LAST evaluated point unless
// *reeval is set), 1 when the search failed (no bracket / iteration limit), or a negative grape_status.
struct HagerZhang {
    LbfgsRun &run;
    double phi0, dphi0, phi_lim;
    std::vector<double> al, va, sl;               // evaluated steps, values, slopes; index 0 = (0, phi0, dphi0)
    double last_alpha = -1.0;
    int budget;
    bool seeded = false;                          // the first trial (alpha0) has been evaluated already: seed_f, seed_df, dphi0 are set
    double seed_f = 0.0, seed_df = 0.0;
    // strict: LineSearches.jl's control flow to the letter (line_search = 1; oracle/optim_lbfgs.py restates it and
    // tests/test_gpu_lbfgs.py compares the two step by step): `linesearchmax` counts passes of the bracketing and the
    // secant^2 loops (a bisection inside one of them is not counted), a bracketing point is never accepted without a
    // secant step, a bracket of width <= eps(b) or a flat one returns its lower end even when that is alpha = 0.
    bool strict = false;
    int ls_max = 50, iter = 1;
    static constexpr double delta = 0.1, sigma = 0.9, rho = 5.0, epsilon = 1e-6, gamma = 0.66, psi3 = 0.1;

    int eval(double c_, bool have_trial = false)
    {
        double f, df;
        if (budget-- <= 0) return 1;
        const int rc = run.phi(c_, have_trial, f, df);
        if (rc) return rc;
        al.push_back(c_); va.push_back(f); sl.push_back(df);
        last_alpha = c_;
        return 0;
    }
    bool wolfe(size_t i) const
    {
        const double c_ = al[i], f = va[i], df = sl[i];
        const bool w1 = delta * dphi0 >= (f - phi0) / c_ && df >= sigma * dphi0;
        const bool w2 = (2.0 * delta - 1.0) * dphi0 >= df && df >= sigma * dphi0 && f <= phi_lim;
        return w1 || w2;
    }
    static double secant(double a, double b, double da, double db) { return (a * db - b * da) / (db - da); }
    // HZ U3: bisection on [a, b] with phi'(a) < 0, phi(a) <= phi_lim, phi'(b) < 0, phi(b) > phi_lim
    int bisect(size_t &ia, size_t &ib)
    {
        while (al[ib] - al[ia] > std::numeric_limits<double>::epsilon() * al[ib]) {
            const int rc = eval(0.5 * (al[ia] + al[ib]));
            if (rc) return rc;
            const size_t id = al.size() - 1;
            if (sl[id] >= 0.0) { ib = id; return 0; }
            if (va[id] <= phi_lim) ia = id; else ib = id;
        }
        return 0;
    }
    // HZ U0-U3
    int update(size_t ia, size_t ib, size_t ic, size_t &oa, size_t &ob)
    {
        oa = ia; ob = ib;
        if (al[ic] < al[ia] || al[ic] > al[ib]) return 0;
        if (sl[ic] >= 0.0) { ob = ic; return 0; }
        if (va[ic] <= phi_lim) { oa = ic; return 0; }
        ob = ic;
        return bisect(oa, ob);
    }
    int secant2(size_t ia, size_t ib, bool &iswolfe, size_t &oa, size_t &ob)
    {
        iswolfe = false;
        double c_ = secant(al[ia], al[ib], sl[ia], sl[ib]);
        if (!(c_ == c_) || std::isinf(c_)) { oa = ia; ob = ib; return 1; }
        int rc = eval(c_);
        if (rc) return rc;
        size_t ic = al.size() - 1;
        if (wolfe(ic)) { iswolfe = true; oa = ob = ic; return 0; }
        size_t iA, iB;
        rc = update(ia, ib, ic, iA, iB);
        if (rc) return rc;
        const double a = al[iA], b = al[iB];
        bool second = false;
        if (iB == ic) { c_ = secant(al[ib], al[iB], sl[ib], sl[iB]); second = true; }
        else if (iA == ic) { c_ = secant(al[ia], al[iA], sl[ia], sl[iA]); second = true; }
        if (second && a <= c_ && c_ <= b) {
            rc = eval(c_);
            if (rc) return rc;
            ic = al.size() - 1;
            if (wolfe(ic)) { iswolfe = true; oa = ob = ic; return 0; }
            size_t nA, nB;
            rc = update(iA, iB, ic, nA, nB);
            if (rc) return rc;
            iA = nA; iB = nB;
        }
        oa = iA; ob = iB;
        return 0;
    }
    // alpha0 has been written to trial slot 0 by the direction kernel
    int search(double alpha0, bool mayterminate, double &alpha_out, double &phi_out, bool &reeval)
    {
        reeval = false;
        phi_lim = phi0 + epsilon * std::fabs(phi0);
        double c_ = alpha0;
        al.assign(1, 0.0); va.assign(1, phi0); sl.assign(1, 0.0);
        int rc = 0;
        if (seeded) {
            --budget;
            al.push_back(c_); va.push_back(seed_f); sl.push_back(seed_df);
            last_alpha = c_;
        } else {
            rc = eval(c_, true);                                   // the probe also publishes phi'(0) = g.d of the direction kernel
            if (rc) return rc;
            dphi0 = run.h_sc[10];
        }
        sl[0] = dphi0;
        if (!(dphi0 < 0.0)) return 1;
        for (int it = 0; !(std::isfinite(va.back()) && std::isfinite(sl.back())); ++it) {   // shrink out of a non-finite region
            if (it >= 52) return 1;
            al.pop_back(); va.pop_back(); sl.pop_back();
            c_ *= psi3;
            rc = eval(c_);
            if (rc) return rc;
        }
        auto accept = [&](size_t i) {
            alpha_out = al[i]; phi_out = va[i];
            reeval = al[i] != last_alpha;
            return 0;
        };
        if (mayterminate && wolfe(al.size() - 1)) return accept(al.size() - 1);
        // bracketing, HZ B0-B3
        size_t ia = 0, ib = 1;
        bool bracketed = false;
        iter = 1;
        while (!bracketed && (!strict || iter < ls_max)) {
            const size_t ic = al.size() - 1;
            if (sl[ic] >= 0.0) {                                   // B1: reached the upward slope
                ib = ic;
                for (size_t i = ib; i-- > 0;)
                    if (va[i] <= phi_lim) { ia = i; break; }
                bracketed = true;
            } else if (va[ic] > phi_lim) {                         // B2: over the crest, slope still downward: bisect
                ia = 0; ib = ic;
                rc = bisect(ia, ib);
                if (rc) return rc;
                bracketed = true;
            } else {                                               // B3: still going downhill: expand
                const double cold = c_;
                c_ *= rho;
                if (!(c_ < 1e30)) return 1;
                rc = eval(c_);
                if (rc) return rc;
                for (int it = 0; !(std::isfinite(va.back()) && std::isfinite(sl.back())); ++it) {   // back towards the last finite point
                    if (it >= 52) return 1;
                    al.pop_back(); va.pop_back(); sl.pop_back();
                    c_ = 0.5 * (cold + c_);
                    rc = eval(c_);
                    if (rc) return rc;
                }
            }
            ++iter;
        }
        if (!bracketed) return 1;
        if (!strict && ia != ib && wolfe(ib) && ib != 0) return accept(ib);   // (a bracketing point may already do)
        while (!strict || iter < ls_max) {
            const double a = al[ia], b = al[ib];
            if (b - a <= std::numeric_limits<double>::epsilon() * b)
                return (ia == 0 && !strict) ? 1 : accept(ia);
            bool iswolfe;
            size_t iA, iB;
            rc = secant2(ia, ib, iswolfe, iA, iB);
            if (rc) return rc;
            if (iswolfe) return accept(iA);
            if (al[iB] - al[iA] < gamma * (b - a)) {
                if (std::nextafter(va[ia], INFINITY) >= va[ib] && std::nextafter(va[iA], INFINITY) >= va[iB])
                    return (iA == 0 && !strict) ? 1 : accept(iA);  // flat to the last bit
                ia = iA; ib = iB;
            } else {                                               // secant converges too slowly: bisect
                rc = eval(0.5 * (al[iA] + al[iB]));
                if (rc) return rc;
                rc = update(iA, iB, al.size() - 1, ia, ib);
                if (rc) return rc;
            }
            ++iter;
        }
        return 1;                                                  // LineSearchException: linesearchmax passes without convergence
    }
};

}  // namespace

extern "C" int grape_lbfgs(grape_ctx *c, const double *x0, const grape_lbfgs_options *opts, double *x_min,
                           grape_lbfgs_result *result)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!x0 || !x_min || !result) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_lbfgs: null argument");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, "grape_lbfgs: operators not set");
    const size_t kn = KN(c), Q = kn + 1;
    if (kn > (size_t)grape::kLbfgsMaxPer * 1024)
        return fail(c, GRAPE_ERR_UNSUPPORTED, "grape_lbfgs: n_controls * n_slices > 16384");
    grape_lbfgs_options o{};
    if (opts) o = *opts;
    const int m = o.memory > 0 ? o.memory : 10;
    const int max_it = o.max_iterations > 0 ? o.max_iterations : 1000;
    const double g_tol = o.g_tol >= 0.0 ? o.g_tol : 1e-8;
    const double f_tol = o.f_tol > 0.0 ? o.f_tol : 0.0;
    const int max_ls = o.max_linesearch > 0 ? o.max_linesearch : 50;
    if (m > 64) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_lbfgs: memory must be <= 64");
    if (o.line_search < 0 || o.line_search > 2) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_lbfgs: line_search must be 0, 1 or 2");
    const bool multi = c->is_group || c->comm != nullptr || c->ipc_ranks > 1;
    if (multi && o.line_search == 2 && c->B < 2)
        return fail(c, GRAPE_ERR_UNSUPPORTED, "grape_lbfgs: the batched ladder search (line_search = 2) needs max_batch >= 2 "
                                              "(multi-device contexts default to the Hager-Zhang search)");
    grape_ctx *lead = c->is_group ? c->sub[0] : c;
    int B = 1;
    if (o.line_search == 2 || (!multi && o.probes > 1)) {        // (multi-device contexts: batched probes only on request)
        B = o.probes;
        if (B <= 0) {
            // probing several step lengths multiplies the sweep's work: free while the ensemble leaves the chip
            // mostly empty, not when it already fills it
            const long waves = (long)c->cfg.n_ensemble * c->W;
            B = waves * 4 <= 8L * c->compute_units ? 4 : (waves * 2 <= 8L * c->compute_units ? 2 : 1);
        }
        if (B > c->B) B = c->B;
        if (B > grape::kLbfgsMaxProbes) B = grape::kLbfgsMaxProbes;
        if (B < 1) B = 1;
    }
    HIP_TRY(c, hipSetDevice(lead->device));
    if (lead->dev_pending) {
        HIP_TRY(c, hipStreamWaitEvent(lead->stream, lead->ev_dev, 0));
        lead->dev_pending = false;
    }
    // workspace (freed on return): vectors + history + trial points/results + scalars
    const size_t n_dbl = 3 * kn + 2 * (size_t)m * kn + m + (size_t)B * kn + (size_t)B * Q + grape::kLbfgsMaxProbes + 8 +
                         8 + 2 * (size_t)m * m + (size_t)grape::kLbfgsDotBlocks * grape::kLbfgsDotStride;
    double *buf = nullptr, *h_sc = nullptr, *d_h_sc = nullptr;
    HIP_TRY(c, hipMalloc((void **)&buf, sizeof(double) * n_dbl));
    hipError_t he = hipHostMalloc((void **)&h_sc, sizeof(double) * 16, hipHostMallocMapped | hipHostMallocCoherent);
    if (he == hipSuccess) he = hipHostGetDevicePointer((void **)&d_h_sc, h_sc, 0);
    if (he != hipSuccess) {
        (void)hipFree(buf);
        if (h_sc) (void)hipHostFree(h_sc);
        return fail(c, GRAPE_ERR_ALLOC, std::string("grape_lbfgs: ") + hipGetErrorString(he));
    }
    LbfgsRun run;
    run.c = c;
    run.lead = lead;
    run.h_sc = h_sc;
    {
        const char *fp = std::getenv("GRAPE_LBFGS_FUSED_PROBE");   // 0: lbfgs_select_kernel behind every trial evaluation (A/B, tests)
        run.fused_probe = eval_ends_in_reduce(c) && !(fp && fp[0] == '0');
    }
    grape::LbfgsState &st = run.st;
    double *p = buf;
    st.x = p; p += kn;
    st.g = p; p += kn;
    st.d = p; p += kn;
    st.S = p; p += (size_t)m * kn;
    st.Y = p; p += (size_t)m * kn;
    st.rho = p; p += m;
    st.xt = p; p += (size_t)B * kn;
    st.fgt = p; p += (size_t)B * Q;
    st.alphas = p; p += grape::kLbfgsMaxProbes;
    st.sc = p; p += 8;
    st.sc_out = p; p += 8;
    st.gram = p; p += 2 * (size_t)m * m;
    st.dots = p;
    st.host_sc = d_h_sc;
    {
        const char *me = std::getenv("GRAPE_LBFGS_MB");              // 0: the single-workgroup step kernels throughout (A/B, tests)
        run.mb = o.line_search != 2 && m <= grape::kLbfgsMbM && !(me && me[0] == '0');
    }
    st.c1 = 1e-4;
    st.c2 = 0.9;
    st.KN = (int32_t)kn;
    st.m = m;
    int rc = GRAPE_OK;
    auto cleanup = [&](int code) {
        // after a timeout the device is presumed hung: synchronising would block for ever and freeing memory a running
        // kernel may still write is no better -- the buffers are leaked and the error returned
        if (run.hung) return code;
        (void)hipSetDevice(lead->device);
        (void)hipStreamSynchronize(lead->stream);
        (void)hipFree(buf);
        (void)hipHostFree(h_sc);
        return code;
    };
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    // f(x0), g(x0): x0 goes to the trial slot, the evaluation's [g, F] to the iterate
    if (hipMemcpyAsync(st.x, x0, sizeof(double) * kn, hipMemcpyHostToDevice, lead->stream) != hipSuccess ||
        hipMemcpyAsync(st.xt, st.x, sizeof(double) * kn, hipMemcpyDeviceToDevice, lead->stream) != hipSuccess)
        return cleanup(fail(c, GRAPE_ERR_HIP, "grape_lbfgs: upload of x0 failed"));
    rc = run.evaluate(1);
    if (rc) return cleanup(rc);
    if (hipSetDevice(lead->device) != hipSuccess) return cleanup(fail(c, GRAPE_ERR_HIP, "grape_lbfgs: hipSetDevice failed"));
    if (grape::launch_lbfgs_init(st, lead->stream, run.signal()) != hipSuccess)
        return cleanup(fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed"));
    rc = run.wait();
    if (rc) return cleanup(rc);
    int it = 0, status = 2, hz_fallbacks = 0;
    c->lb_alpha.clear();
    c->lb_evals.clear();
    double F = h_sc[0], gnorm = h_sc[1];
    if (gnorm <= g_tol) status = 0;
    // the factor-2 ladder search of one iteration (trial points of the first ladder already written when `first_written`)
    auto ladder = [&](bool &accepted) -> int {
        // Step lengths: start at 1 (Optim: InitialStatic(alpha = 1)); every launch probes B lengths in a
        // factor-2 ladder.  Ladders tried in turn until one holds an acceptable step:
        //   [1 .. 2^-(B-1)], then the next B LARGER lengths [2^B .. 2], then the next B smaller, ...
        // (expansion matters for the reference's UnitaryGate conventions, where g is not the gradient of
        // the reported figure of merit -- SURVEY.md App. C #2 -- and small steps along -g may not descend).
        int tried = 0, shrink = 0, grow = 0;
        accepted = false;
        while (tried < max_ls) {
            const bool up = (shrink > grow);                  // alternate: down, up, down, up, ...
            const int top = up ? (grow + 1) * B : -shrink * B;    // exponent of the ladder's largest length
            if (up) ++grow; else ++shrink;
            if (top > 40 && -shrink * B < -60) break;         // both directions exhausted: no acceptable step exists
            if (top > 40 || top < -60) { tried += B; continue; }
            const double alpha0 = std::ldexp(1.0, top);
            HIP_TRY(c, hipSetDevice(lead->device));
            if (grape::launch_lbfgs_direction(st, B, alpha0, lead->stream) != hipSuccess)
                return fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed");
            int r = run.evaluate(B);
            if (r) return r;
            HIP_TRY(c, hipSetDevice(lead->device));
            if (grape::launch_lbfgs_select(st, B, lead->stream, run.signal(), 0) != hipSuccess)
                return fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed");
            r = run.wait();
            if (r) return r;
            tried += B;
            if (h_sc[5] == 0.0) { accepted = true; break; }
        }
        return GRAPE_OK;
    };
    auto launch_step = [&](int commit, bool with_signal, double alpha_acc = 1.0) -> int {
        HIP_TRY(c, hipSetDevice(lead->device));
        const grape::DoneSignal ds = with_signal ? run.signal() : grape::DoneSignal();
        if (commit && run.mb) {
            // (the dot products of the accepted trial point are in st.dots: lbfgs_dots_kernel ran behind its evaluation)
            if (grape::launch_lbfgs_step_mb(st, alpha_acc, lead->stream, ds) != hipSuccess)
                return fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed");
            std::swap(st.sc, st.sc_out);                      // the new iterate's scalars: what every later launch reads
            return GRAPE_OK;
        }
        if (grape::launch_lbfgs_step(st, commit, lead->stream, ds) != hipSuccess)
            return fail(c, GRAPE_ERR_HIP, "grape_lbfgs: launch failed");
        return GRAPE_OK;
    };
    auto launch_probe = [&]() -> int { return run.probe(); };  // trial slot 0: evaluate, publish phi, phi' (and phi'(0))
    if (o.line_search == 2) {
        while (status == 2 && it < max_it) {
            const double F_prev = F;
            bool accepted = false;
            rc = ladder(accepted);
            if (rc) return cleanup(rc);
            if (!accepted) { status = 3; break; }
            ++it;
            F = h_sc[0];
            gnorm = h_sc[1];
            if (gnorm <= g_tol) { status = 0; break; }
            if (f_tol > 0.0 && std::fabs(F - F_prev) <= f_tol * std::fabs(F)) { status = 1; break; }
        }
    } else if (status == 2) {
        // Hager-Zhang, pipelined: an accepted step is committed and the next direction formed by ONE single-wave kernel
        // (lbfgs_step_kernel) that the host does not wait for -- the evaluation of the next trial point x + d is queued
        // right behind it, and the committed iterate's F and |g| arrive with that evaluation's probe.  (When the
        // iterate turns out to have converged, that one evaluation was speculative.)
        rc = launch_step(0, false);
        if (rc == GRAPE_OK) rc = launch_probe();
        if (rc) return cleanup(rc);
        bool committed = false;                              // h_sc[0..1] hold an iterate the convergence tests have not seen
        double F_before = F;
        int same_f = 0;
        for (;;) {
            rc = run.wait();
            if (rc) return cleanup(rc);
            if (committed) {
                F = h_sc[0];
                gnorm = h_sc[1];
                committed = false;
                if (gnorm <= g_tol) { status = 0; break; }
                if (f_tol > 0.0 && std::fabs(F - F_before) <= f_tol * std::fabs(F)) { status = 1; break; }
                // (Optim with f_abstol = f_reltol = 0: "f converged" = no change at all in two successive iterations)
                same_f = (o.line_search == 1 && F == F_before) ? same_f + 1 : 0;
                if (same_f > 1) { status = 1; break; }
                if (it >= max_it) break;
            }
            const bool strict = o.line_search == 1;
            // (strict: the evaluation count is bounded by LineSearches' own rule -- max_ls passes; 64 bisections of a pass at
            // most -- the budget is only a backstop)
            HagerZhang hz{run, F, h_sc[10], 0.0, {}, {}, {}, -1.0, strict ? 64 * max_ls : max_ls};
            hz.strict = strict;
            hz.ls_max = max_ls;
            hz.seeded = true;
            hz.seed_f = h_sc[8];
            hz.seed_df = h_sc[9];
            double alpha = 1.0, fa = F;
            bool reeval = false;
            const int hr = hz.search(1.0, o.line_search == 0, alpha, fa, reeval);
            if (hr < 0) return cleanup(hr);
            F_before = F;
            if (hr == 0 && alpha == 0.0) {
                // (strict mode only) the search returned the lower end of a bracket that has collapsed onto alpha = 0: Optim
                // takes the zero step, finds x unchanged and stops with "x converged" (x_abstol = 0 is met by equality)
                ++it;
                c->lb_alpha.push_back(0.0);
                c->lb_evals.push_back(run.evals);
                status = 4;
                break;
            }
            if (hr == 0) {
                if (reeval) {                                      // the accepted step is not the one evaluated last
                    double f2, df2;
                    rc = run.phi(alpha, false, f2, df2);
                    if (rc) return cleanup(rc);
                }
                ++it;
                c->lb_alpha.push_back(alpha);
                c->lb_evals.push_back(run.evals);
                committed = true;
                if (it >= max_it) {                                // the last iterate: commit, wait, no further trial
                    rc = launch_step(1, true, alpha);
                    if (rc) return cleanup(rc);
                    continue;                                      // (the wait at the top reads its F and |g|)
                }
                rc = launch_step(1, false, alpha);
                if (rc == GRAPE_OK) rc = launch_probe();
                if (rc) return cleanup(rc);
            } else if (strict) {                                   // LineSearchException: Optim stops here (line search failed)
                status = 3;
                break;
            } else {                                               // no bracket: the ladder search takes this iteration
                ++hz_fallbacks;
                run.mb = false;                                    // the ladder's pair will have no Gram row: single-workgroup steps from here on
                bool accepted = false;
                if (multi) { status = 3; break; }
                rc = ladder(accepted);
                if (rc) return cleanup(rc);
                if (!accepted) { status = 3; break; }
                ++it;
                F = h_sc[0];
                gnorm = h_sc[1];
                if (gnorm <= g_tol) { status = 0; break; }
                if (f_tol > 0.0 && std::fabs(F - F_before) <= f_tol * std::fabs(F)) { status = 1; break; }
                if (it >= max_it) break;
                rc = launch_step(0, false);
                if (rc == GRAPE_OK) rc = launch_probe();
                if (rc) return cleanup(rc);
            }
        }
    }
    if (hipSetDevice(lead->device) != hipSuccess) return cleanup(fail(c, GRAPE_ERR_HIP, "grape_lbfgs: hipSetDevice failed"));
    if (hipMemcpyAsync(lead->h_fg, st.x, sizeof(double) * kn, hipMemcpyDeviceToHost, lead->stream) != hipSuccess ||
        hipStreamSynchronize(lead->stream) != hipSuccess)
        return cleanup(fail(c, GRAPE_ERR_HIP, "grape_lbfgs: download of the minimiser failed"));
    std::memcpy(x_min, lead->h_fg, sizeof(double) * kn);
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    result->minimum = F;
    result->g_norm = gnorm;
    result->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    result->iterations = it;
    result->evaluations = run.evals;
    result->status = status;
    result->probes = B;
    result->line_search = o.line_search;
    result->ladder_fallbacks = hz_fallbacks;
    c->evaluated = true;
    return cleanup(GRAPE_OK);
}

extern "C" int grape_lbfgs_get_trace(const grape_ctx *c, double *alphas, int32_t *evals, int32_t capacity, int32_t *count)
{
    if (!c || !count) return GRAPE_ERR_INVALID_ARG;
    const size_t n = c->lb_alpha.size();
    *count = (int32_t)n;
    for (size_t i = 0; i < n && (int64_t)i < (int64_t)capacity; ++i) {
        if (alphas) alphas[i] = c->lb_alpha[i];
        if (evals) evals[i] = c->lb_evals[i];
    }
    return GRAPE_OK;
}

extern "C" int grape_get_member_results(grape_ctx *c, double *foms, double *grads)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!c->evaluated) return fail(c, GRAPE_ERR_NOT_READY, "grape_get_member_results: no evaluation yet");
    if (c->is_group) {
        const size_t kn = KN(c);
        for (size_t i = 0; i < c->sub.size(); ++i) {
            const size_t lo = (size_t)c->sub_lo[i];
            const int rc = grape_get_member_results(c->sub[i], foms ? foms + lo : nullptr, grads ? grads + lo * kn : nullptr);
            if (rc) return group_fail(c, c->sub[i], rc);
        }
        return GRAPE_OK;
    }
    if (!c->d_member_out)
        return fail(c, GRAPE_ERR_NOT_READY, "grape_get_member_results: create the context with GRAPE_FLAG_MEMBER_RESULTS");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    const size_t kn = KN(c), Q = kn + 1, E = c->cfg.n_ensemble;
    std::vector<double> h(E * Q);
    HIP_TRY(c, hipMemcpy(h.data(), c->d_member_out, sizeof(double) * E * Q, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < E; ++k) {
        if (foms) foms[k] = h[k * Q + kn];
        if (grads) std::memcpy(grads + k * kn, h.data() + k * Q, sizeof(double) * kn);
    }
    return GRAPE_OK;
}

// gathers one member's slab from the lane-major workspace layout into (n,n,count) col-major
static int fetch_slab(grape_ctx *c, const double2 *d_ws, int member, cplx *out, bool odd_transposed = false)
{
    if (c->family == 1) {
        const size_t N = c->cfg.n_slices, TSZ = c->TSZ;
        const int n = c->cfg.n, NT = c->NT;
        const int unit = c->pack2 ? member / 2 : member, off = c->pack2 ? 8 * (member & 1) : 0;
        std::vector<cplx> h(N * TSZ);
        HIP_TRY(c, hipMemcpy(h.data(), d_ws + (size_t)unit * N * TSZ, sizeof(cplx) * h.size(),
                             hipMemcpyDeviceToHost));
        for (size_t t = 0; t < N; ++t)
            for (int I = 0; I < NT; ++I)
                for (int J = 0; J < NT; ++J)
                    for (int r = 0; r < 4; ++r)
                        for (int l = 0; l < 64; ++l) {
                            int row = 16 * I + 4 * r + (l >> 4) - off, col = 16 * J + (l & 15) - off;
                            if (odd_transposed && (t & 1)) std::swap(row, col);      // rank-one chain: P_t^T dumps
                            if (row >= 0 && col >= 0 && row < n && col < n)
                                out[t * n * n + row + (size_t)n * col] =
                                    h[t * TSZ + (size_t)((I * NT + J) * 4 + r) * 64 + l];
                        }
        return GRAPE_OK;
    }
    if (c->family == 2) {                                    // plain N x (n x n) per member
        const size_t nn2 = (size_t)c->cfg.n * c->cfg.n, N2 = c->cfg.n_slices;
        HIP_TRY(c, hipMemcpy(out, d_ws + (size_t)member * N2 * nn2, sizeof(cplx) * N2 * nn2, hipMemcpyDeviceToHost));
        return GRAPE_OK;
    }
    const size_t nn = (size_t)c->cfg.n * c->cfg.n, S = c->S, LT = c->CH, N = c->cfg.n_slices;   // LT: chunks per member
    std::vector<cplx> h(S * nn * LT);
    HIP_TRY(c, hipMemcpy(h.data(), d_ws + (size_t)member * S * nn * LT, sizeof(cplx) * h.size(),
                         hipMemcpyDeviceToHost));
    for (size_t t = 0; t < N; ++t) {
        const size_t L = t / S, j = t % S;
        for (size_t e = 0; e < nn; ++e)
            out[t * nn + e] = h[(j * nn + e) * LT + L];
    }
    return GRAPE_OK;
}

extern "C" int grape_get_trajectory(grape_ctx *c, int32_t member, double *props, double *states,
                                    double *costates)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!c->evaluated) return fail(c, GRAPE_ERR_NOT_READY, "grape_get_trajectory: no evaluation yet");
    if (member < 0 || member >= c->cfg.n_ensemble)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_get_trajectory: member out of range");
    if (c->is_group) {
        int local = 0;
        grape_ctx *s = owner_of(c, member, &local);
        const int rc = grape_get_trajectory(s, local, props, states, costates);
        return rc ? group_fail(c, s, rc) : GRAPE_OK;
    }
    if (chunked(c))
        return fail(c, GRAPE_ERR_NOT_READY,
                    "grape_get_trajectory: the workspace of this context holds " + std::to_string(c->Ec) + " of its " +
                    std::to_string(c->cfg.n_ensemble) + " members at a time (grape_info.member_chunk): no stored trajectory to return");
    if ((costates && !c->d_costates) || ((costates || states) && c->exact_w1))
        return fail(c, GRAPE_ERR_NOT_READY,
                    "grape_get_trajectory: costates need GRAPE_FLAG_KEEP_COSTATES at grape_create (this exact-gradient flow "
                    "stores neither states nor costates)");
    if (props && c->action)
        return fail(c, GRAPE_ERR_NOT_READY,
                    "grape_get_trajectory: this flow applies exp(G_t) to vectors and forms no propagators; create the "
                    "context with GRAPE_FLAG_KEEP_COSTATES");
    if (states && !states_stored(c))
        return fail(c, GRAPE_ERR_NOT_READY,
                    "grape_get_trajectory: forward states are stored only by the debug flow; create the "
                    "context with GRAPE_FLAG_KEEP_COSTATES");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    const int n = c->cfg.n;
    const size_t nn = (size_t)n * n, N = c->cfg.n_slices, K = c->cfg.n_controls;
    std::vector<cplx> P(N * nn);
    int rc = fetch_slab(c, c->d_props, member, P.data(), c->thin && !c->thin_dpp && !c->grid);
    if (rc) return rc;
    if (props) std::memcpy(props, P.data(), sizeof(cplx) * N * nn);
    // n x m states: the workspace holds them zero-padded to n x n; hand out the first m columns
    std::vector<cplx> xs_full, ls_full;
    double *states_user = states, *costates_user = costates;
    if (c->m != n) {
        if (states) { xs_full.resize((N + 1) * nn); states = reinterpret_cast<double *>(xs_full.data()); }
        if (costates) { ls_full.resize((N + 1) * nn); costates = reinterpret_cast<double *>(ls_full.data()); }
    }
    auto narrow = [&](const std::vector<cplx> &full, double *user) {
        const size_t nm = (size_t)n * c->m;
        for (size_t t = 0; t <= N; ++t)
            std::memcpy(reinterpret_cast<cplx *>(user) + t * nm, full.data() + t * nn, sizeof(cplx) * nm);
    };
    if (states) {
        cplx *X = reinterpret_cast<cplx *>(states);
        rc = fetch_slab(c, c->d_states, member, X);
        if (rc) return rc;
        // the final state X_N is never needed by the gradient, so the kernel does not form
        // it; complete the reference's fwd_state_store[N+1] here (debug accessor only).
        const cplx *Pl = P.data() + (N - 1) * nn, *Xl = X + (N - 1) * nn;
        cplx *Xn = X + N * nn;
        std::vector<cplx> tmp(nn);
        const bool sand = c->cfg.sys_type != GRAPE_UNITARY_GATE;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                cplx s = 0.0;
                for (int k = 0; k < n; ++k)
                    s += sand ? Xl[i + k * n] * std::conj(Pl[j + k * n]) : Pl[i + k * n] * Xl[k + j * n];
                (sand ? tmp[i + j * n] : Xn[i + j * n]) = s;
            }
        if (sand)
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) {
                    cplx s = 0.0;
                    for (int k = 0; k < n; ++k) s += Pl[i + k * n] * tmp[k + j * n];
                    Xn[i + j * n] = s;
                }
    }
    if (costates) {
        cplx *Lc = reinterpret_cast<cplx *>(costates);
        rc = fetch_slab(c, c->d_costates, member, Lc);
        if (rc) return rc;
        if (c->family != 1) {
            HIP_TRY(c, hipMemcpy(Lc + N * nn, c->d_ops + (size_t)member * (K + 3) * nn + (K + 2) * nn,
                                 sizeof(cplx) * nn, hipMemcpyDeviceToHost));
        } else {
            std::vector<cplx> h(c->TSZ);
            const int unit = c->pack2 ? member / 2 : member, off = c->pack2 ? 8 * (member & 1) : 0;
            HIP_TRY(c, hipMemcpy(h.data(), c->d_ops + ((size_t)unit * (2 * K + 3) + 2 * K + 2) * c->TSZ,
                                 sizeof(cplx) * c->TSZ, hipMemcpyDeviceToHost));
            for (int I = 0; I < c->NT; ++I)
                for (int J = 0; J < c->NT; ++J)
                    for (int r = 0; r < 4; ++r)
                        for (int l = 0; l < 64; ++l) {
                            const int row = 16 * I + 4 * r + (l >> 4) - off, col = 16 * J + (l & 15) - off;
                            if (row >= 0 && col >= 0 && row < n && col < n)
                                Lc[N * nn + row + (size_t)n * col] = h[(size_t)((I * c->NT + J) * 4 + r) * 64 + l];
                        }
        }
    }
    if (c->m != n) {
        if (states_user) narrow(xs_full, states_user);
        if (costates_user) narrow(ls_full, costates_user);
    }
    return GRAPE_OK;
}

extern "C" int grape_get_kernel_time(grape_ctx *c, double *total_ms, int64_t *launches, int32_t reset)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (c->is_group) {                                       // the shards run concurrently: report the slowest device
        double worst = 0.0;
        int64_t n = 0;
        for (grape_ctx *s : c->sub) {
            double ms = 0.0;
            int64_t cnt = 0;
            const int rc = grape_get_kernel_time(s, &ms, &cnt, reset);
            if (rc) return group_fail(c, s, rc);
            if (ms > worst) worst = ms;
            n = cnt;
        }
        if (total_ms) *total_ms = worst;
        if (launches) *launches = n;
        return GRAPE_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = fold_events(c, c->ev_issued - c->ev_folded);
    if (rc) return rc;
    if (total_ms) *total_ms = c->ev_total_ms;
    if (launches) *launches = c->ev_count;
    if (reset) { c->ev_total_ms = 0.0; c->ev_count = 0; c->smp_total.clear(); c->smp_first.clear(); }
    return GRAPE_OK;
}

extern "C" int grape_get_kernel_names(const grape_ctx *c, char *buf, int32_t capacity)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    const grape_ctx *s = c->is_group && !c->sub.empty() ? c->sub[0] : c;
    const std::string &log = s->kernel_log;
    if (buf && capacity > 0) {
        const size_t n = std::min(log.size(), (size_t)capacity - 1);
        std::memcpy(buf, log.data(), n);
        buf[n] = 0;
    }
    return (int)log.size() + 1;
}

extern "C" int grape_get_kernel_samples(grape_ctx *c, double *total_ms, double *first_ms, int64_t capacity, int64_t *count)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (c->is_group) {                                       // the shards run concurrently: the first device stands for them
        const int rc = grape_get_kernel_samples(c->sub[0], total_ms, first_ms, capacity, count);
        return rc ? group_fail(c, c->sub[0], rc) : GRAPE_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = fold_events(c, c->ev_issued - c->ev_folded);
    if (rc) return rc;
    const int64_t have = (int64_t)c->smp_total.size();
    if (count) *count = have;
    const int64_t n = capacity < have ? (capacity < 0 ? 0 : capacity) : have;
    for (int64_t i = 0; i < n; ++i) {                        // the most recent n, oldest first
        if (total_ms) total_ms[i] = c->smp_total[(size_t)(have - n + i)];
        if (first_ms) first_ms[i] = c->smp_first[(size_t)(have - n + i)];
    }
    return GRAPE_OK;
}

extern "C" int grape_get_group_timing(grape_ctx *c, double *out, int32_t reset)
{
    if (!c || !out) return GRAPE_ERR_INVALID_ARG;
    const double n = c->group_tm_n ? (double)c->group_tm_n : 1.0;
    out[0] = (double)c->group_tm_n;
    for (int i = 0; i < 5; ++i) out[1 + i] = 1e6 * c->group_tm[i] / n;
    if (reset) {
        c->group_tm_n = 0;
        for (double &v : c->group_tm) v = 0.0;
    }
    return GRAPE_OK;
}

extern "C" int grape_get_phase_stamps(grape_ctx *c, uint64_t *out, int64_t capacity, int64_t *count)
{
    DeviceGuard guard;
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (c->is_group || !c->d_stamps || !c->evaluated)
        return fail(c, GRAPE_ERR_NOT_READY, "grape_get_phase_stamps: needs GRAPE_FLAG_PHASE_STAMPS and an evaluation");
    const int64_t total = (int64_t)c->cfg.n_ensemble * c->W * grape::kStampSlots;
    if (count) *count = total;
    if (out && capacity > 0) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipDeviceSynchronize());
        const int64_t nget = capacity < total ? capacity : total;
        HIP_TRY(c, hipMemcpy(out, c->d_stamps, sizeof(uint64_t) * nget, hipMemcpyDeviceToHost));
    }
    return GRAPE_OK;
}

extern "C" int grape_get_info(const grape_ctx *c, grape_info *info)
{
    if (!c || !info) return GRAPE_ERR_INVALID_ARG;
    std::memset(info, 0, sizeof(*info));
    info->abi_version = GRAPE_ABI_VERSION;
    info->device = c->device;
    info->compute_units = c->compute_units;
    info->slices_per_lane = c->S;
    info->waves_per_member = c->W;
    info->expm_squarings = c->cfg.expm_squarings;
    info->kernel_family = c->family;
    info->unitary_flow = c->unitary ? 1 : 0;
    info->expm_theta = grape::kTheta8;
    info->workspace_bytes = c->bytes;
    std::snprintf(info->arch, sizeof(info->arch), "%s", c->arch);
    info->n_devices = c->is_group ? (int32_t)c->sub.size() : 1;
    info->comm_size = c->comm_size;
    info->comm_rank = c->comm_rank;
    info->members_first_device = c->is_group ? c->sub[0]->cfg.n_ensemble : c->cfg.n_ensemble;
    if (c->is_group) info->unitary_flow = c->sub[0]->unitary ? 1 : 0;
    info->lane_pair = (c->is_group ? c->sub[0]->pair : c->pair) ? 1 : 0;
    info->states_stored = states_stored(c->is_group ? c->sub[0] : c) ? 1 : 0;
    info->rank_one_chain = (c->is_group ? c->sub[0]->thin : c->thin) ? 1 : 0;
    {
        const grape_ctx *s0 = c->is_group ? c->sub[0] : c;
        info->sparse_controls = (s0->sparse_ctrl || s0->any_sp_nnz) ? 1 : 0;
    }
    {
        const grape_ctx *s0 = c->is_group ? c->sub[0] : c;
        info->fused_forward = (s0->thin && !s0->action && tile_fuse_forward(tile_params(s0, nullptr, 1)) == 1) ? 1 : 0;
        info->time_chunks = s0->tp_C;
        info->hoisted_controls = s0->hoist == 1 ? 1 : 0;
        info->expm_action = s0->action ? 1 : 0;
        info->prop_chain = s0->thin_dpp ? 1 : 0;
        info->member_chunk = s0->Ec;
        info->workspace_budget_bytes = s0->ws_budget;
        info->scaled_controls = s0->ctrl_scaled ? 1 : 0;
        info->propagator_blocks = s0->family == 2 ? s0->any_blocks : 0;
    }
    return GRAPE_OK;
}
