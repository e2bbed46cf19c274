/*
 * grape_hip.h -- C ABI of libgrape_hip.so, the MI355X (gfx950) GRAPE propagator/gradient
 * engine that replaces the body of the (F, G, x) closure QuOptimalControl.jl hands to Optim.
 *
 * The reference has no FFI of its own (pure Julia); the seam is the closure `topt` built in
 *   solve(::Problem, ::GRAPE)          /root/reference/src/solve.jl:63-143  (closure :75-100)
 *   solve(::EnsembleProblem, ::GRAPE)  /root/reference/src/solve.jl:145-250 (closure :164-196)
 * whose body is  _fom_and_gradient_GRAPE!  (src/GRAPE.jl:25-96)  looped over the ensemble.
 * Each entry point below names the reference code it stands in for.  INTEGRATION.md shows the
 * Julia `ccall` glue (julia/GrapeHIP.jl) and the Python ctypes binding that mirror it.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  Every function returns a grape_status
 *     (0 = OK, negative = error); grape_last_error() gives the message.  No exceptions cross.
 *   - complex numbers are interleaved {re, im} doubles == Julia ComplexF64 == double _Complex.
 *   - matrices are column-major (Julia): element (i,j) of an n x n matrix at i + j*n.
 *   - x and G are (K, N) column-major Float64: x[j,i] at j + i*K  (what Optim hands in/out).
 *   - host-pointer arguments are only read/written during the call; the library keeps no
 *     caller pointer after return (Julia: GC.@preserve for the duration of the ccall).
 *   - every entry point selects the device(s) of its context and restores the calling thread's current HIP device before
 *     it returns (a host framework in the same thread keeps allocating and launching where it was).
 *   - one evaluation in flight per context (like the reference's closure, which shares one
 *     evolve_store, src/solve.jl:162); distinct contexts are independent.
 */
#ifndef GRAPE_HIP_H
#define GRAPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRAPE_ABI_VERSION 6

typedef enum grape_status {
    GRAPE_OK = 0,
    GRAPE_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad enum           */
    GRAPE_ERR_UNSUPPORTED = -2,   /* operator dimension / option this build has no kernel for */
    GRAPE_ERR_NO_DEVICE = -3,     /* no HIP device / wrong architecture                 */
    GRAPE_ERR_HIP = -4,           /* a HIP runtime call failed (message has the detail) */
    GRAPE_ERR_NOT_READY = -5,     /* grape_eval before grape_set_operators              */
    GRAPE_ERR_ALLOC = -6,         /* host or device allocation failed                   */
    GRAPE_ERR_TIMEOUT = -7,       /* the device did not finish an evaluation within the time limit
                                     (GRAPE_EVAL_TIMEOUT_S, default 600 s): device presumed hung */
    GRAPE_ERR_COMM = -8           /* RCCL could not be loaded / a collective call failed */
} grape_status;

/* src/problems.jl:8-10.  CoherenceTransfer dispatches exactly like StateTransfer
 * (src/GRAPE.jl:197,236,276,294; src/cost_functions.jl:104). */
typedef enum grape_sys_type {
    GRAPE_UNITARY_GATE = 0,
    GRAPE_STATE_TRANSFER = 1,
    GRAPE_COHERENCE_TRANSFER = 2
} grape_sys_type;

/* GRAPE(isinplace=true)  -> _fom_and_gradient_GRAPE!  (src/GRAPE.jl:25-96):   H = (sum_j B_j x_j) + A,
 *                           UnitaryGate gradient sign +i (src/GRAPE.jl:272)
 * GRAPE(isinplace=false) -> _fom_and_gradient_sGRAPE  (src/GRAPE.jl:103-166): H = A + sum_j B_j x_j,
 *                           UnitaryGate gradient sign -i (src/GRAPE.jl:290) */
typedef enum grape_variant {
    GRAPE_VARIANT_INPLACE = 0,
    GRAPE_VARIANT_STATIC = 1
} grape_variant;

enum {
    GRAPE_FLAG_KEEP_COSTATES = 1 << 0,  /* debug: also store every costate L_t so that
                                           grape_get_trajectory can return them         */
    GRAPE_FLAG_TIME_KERNELS = 1 << 1,   /* record HIP events around the sweep kernel of every
                                           evaluation (see grape_get_kernel_time)       */
    GRAPE_FLAG_PHASE_STAMPS = 1 << 2,   /* diagnostic build of the sweep: every wave stamps the
                                           shader clock at its phase boundaries
                                           (see grape_get_phase_stamps); never for timing runs */
    GRAPE_FLAG_MEMBER_RESULTS = 1 << 4, /* also leave every member's unweighted (F_k, g_k) in HBM for
                                           grape_get_member_results (the reference's
                                           gradient[k,:,:] intermediate); off by default: F and G
                                           do not need it and it costs E*(K*N+1) doubles of writes */
    GRAPE_FLAG_FORCE_GENERAL = 1 << 3,  /* always use the general data flow (forward states stored
                                           in HBM, as the reference does), even when every
                                           generator is Hermitian and the cheaper unitary flow
                                           applies.  KEEP_COSTATES implies it.          */
    GRAPE_FLAG_TIME_SAMPLED = 1 << 6,   /* with TIME_KERNELS: record the event pair on every 8th evaluation only
                                           (an event pair costs ~5 us of a ~90 us host->host call)  */
    GRAPE_FLAG_GROUP_PEER_SUM = 1 << 7, /* multi-device contexts (n_devices >= 2): sum the shards' [G, F] on the first device
                                           through peer copies and one reduction kernel (fixed shard order) instead of the
                                           RCCL all-reduce; librccl is not loaded, and device_ids may repeat a device
                                           (several shards on one GPU: the way the sharding logic is tested on a one-GPU
                                           machine)                                                                    */
    GRAPE_FLAG_FORCE_COLLECTIVE = 1 << 5 /* create the RCCL communicator and run the all-reduce of
                                           [G, F] even when the context spans ONE device (a
                                           1-rank collective: exercises the multi-GPU code path
                                           on a single-GPU machine; testing/diagnostics)  */
};

#define GRAPE_MAX_DEVICES 8

/* Mirrors what solve() unpacks: Problem fields (src/problems.jl:19-28: sys_type, T,
 * n_controls), the integrator's n_slices (src/timeevolution.jl:11-14), EnsembleProblem.n_ens
 * (src/problems.jl:33-41) -- the members this context owns: the whole ensemble for a
 * single-process caller (the library shards it over `n_devices` GPUs itself), or this rank's
 * shard when one process per GPU is used with grape_comm_attach. */
typedef struct grape_config {
    int32_t sys_type;          /* grape_sys_type                                     */
    int32_t variant;           /* grape_variant                                      */
    int32_t n;                 /* operator dimension (d, or d*d for Liouvillians): any n >= 1 (ABI v5; src/GRAPE.jl:25-96 is
                                  size-generic).  n = 2..4 run in registers, 5..32 and 33..64 on the FP64 matrix cores, n = 1 and
                                  n > 64 (up to 2048) through a plain size-generic kernel: correct, not fast */
    int32_t n_controls;        /* K                                                  */
    int32_t n_slices;          /* N                                                  */
    int32_t n_ensemble;        /* E owned by this context (1 for a plain Problem)    */
    double  duration;          /* T                                                  */
    int32_t device;            /* HIP device ordinal, -1 = current device            */
    int32_t flags;             /* GRAPE_FLAG_*                                       */
    /* tuning; 0 = choose automatically */
    int32_t slices_per_lane;   /* S: consecutive time slices one lane owns           */
    int32_t waves_per_member;  /* W: wavefronts that share one member's time axis    */
    int32_t expm_squarings;    /* <0 = per slice from the generator norm; >=0 forces s */
    int32_t max_batch;         /* control arrays one grape_eval_batch call may carry; 0 or 1 = no batching */
    /* ---- ABI v2 ---- */
    int32_t n_state_cols;      /* m: Xi, Xt are n x m (src/problems.jl:23-24 puts no constraint on the
                                  shape); 0 = n (square, every reference test).  m < n (e.g. m = 1,
                                  vectorised density matrices as in test/liou.jl:38-48) needs UnitaryGate */
    int32_t n_devices;         /* 0 or 1: one GPU (`device`).  2..8: the ensemble axis is sharded inside the
                                  library over device_ids[0..n_devices) in contiguous blocks of ceil(E/G)
                                  members (src/solve.jl:166-187 is the loop being split) and every
                                  evaluation ends in ONE RCCL all-reduce of the K*N+1 doubles [G, F]
                                  (src/solve.jl:171-186, :191) */
    int32_t device_ids[GRAPE_MAX_DEVICES];   /* HIP ordinals, used when n_devices >= 2 */
    int32_t gradient;          /* grape_gradient: 0 = the reference's first-order grad_func! (src/GRAPE.jl:261-303),
                                  1 = exact derivative of the objective (what ADGRAPE gets from Zygote,
                                  src/GRAPE.jl:12-20; cf. expm_exact_gradient, src/grape_tools.jl:26-57); 2 <= n <= 64.  Runs behind the
                                  debug flow (every X_t, L_t stored: grape_get_trajectory returns them), except for
                                  UnitaryGate problems with Hermitian generators at n = 2 or 4 (the lane-pair kernel),
                                  which take the unitary flow (grape_info.unitary_flow = 1, states_stored = 0:
                                  grape_get_trajectory serves propagators only) unless GRAPE_FLAG_KEEP_COSTATES is set */
    int32_t objective;         /* grape_objective: 0 = fom_func (src/cost_functions.jl:99-111),
                                  1 = the ADGRAPE functional C1(Xt, U Xi [U']) for every system type
                                  (src/solve.jl:268-291, :317-361); needs gradient = 1 */
} grape_config;

typedef enum grape_gradient { GRAPE_GRADIENT_REFERENCE = 0, GRAPE_GRADIENT_EXACT = 1 } grape_gradient;
typedef enum grape_objective { GRAPE_OBJECTIVE_FOM = 0, GRAPE_OBJECTIVE_C1 = 1 } grape_objective;

typedef struct grape_info {
    int32_t abi_version;
    int32_t device;
    int32_t compute_units;
    int32_t slices_per_lane;       /* S in use                                          */
    int32_t waves_per_member;      /* W in use                                          */
    int32_t expm_squarings;        /* forced s, or -1 = per slice from the generator norm */
    int32_t kernel_family;         /* 0 = register-resident small-n, 1 = LDS/MFMA tile (n = 5..64), 2 = size-generic (n = 1, n > 64) */
    int32_t unitary_flow;          /* 1 after grape_set_operators found every A_k, B_jk Hermitian
                                      (propagators unitary): no forward-state round trip  */
    double  expm_theta;            /* norm threshold below which no scaling/squaring is done */
    uint64_t workspace_bytes;      /* device bytes owned by the context                 */
    char    arch[32];              /* gcnArchName of the device                         */
    /* ---- ABI v2 ---- */
    int32_t n_devices;             /* GPUs this context spans (in-library sharding)      */
    int32_t comm_size;             /* ranks of the RCCL communicator the all-reduce runs on (1 = none) */
    int32_t comm_rank;
    int32_t members_first_device;  /* members owned by device_ids[0] (the largest shard)  */
    int32_t lane_pair;             /* 1: the lane-pair small-n kernel (two lanes per time chunk, two waves per SIMD) */
    int32_t states_stored;         /* 1: grape_get_trajectory can return the forward states (after set_operators);
                                      0: the flow in use rebuilds them on the fly -- ask for GRAPE_FLAG_KEEP_COSTATES */
    int32_t rank_one_chain;        /* 1 after grape_set_operators found rank-one states (Xi = v v', Xt = w w' under the
                                      sandwich, or n x 1 states) where a vector flow exists: n = 9..16; n = 5..8 and 17..32
                                      with member-invariant controls on large ensembles; n = 33..64 with sparse control
                                      operators: the sweeps run on vectors; GRAPE_FLAG_FORCE_GENERAL keeps the dense chain */
    int32_t sparse_controls;       /* 1: every control operator has at most 64 non-zeros (Pauli-type controls; up to 256 --
                                      sums of a few Pauli strings, global drives -- where the longer lists pay) and the
                                      kernels that support it read (coefficient, position) lists instead of dense
                                      operators for the gradient traces */
    int32_t fused_forward;         /* rank-one chain, single evaluations: 1 when the forward vector pass runs inside the
                                      expm kernel (ensembles of at least 2 x compute_units members), so every propagator is
                                      read from HBM once instead of twice */
    int32_t time_chunks;           /* n = 5..32, fewer members than wavefront slots: the time axis
                                      of every member is cut into this many chunks evaluated in parallel (0 = one
                                      wavefront walks all slices) */
    /* ---- ABI v3 ---- */
    int32_t hoisted_controls;      /* 1 after grape_set_operators found the control operators B_c identical for every
                                      member (B_gens = k -> [Sx, Sy], test/setup_tests.jl:32) in the n = 5..32 family:
                                      the control sum sum_c x[c,t] B_c of src/timeevolution.jl:105-107 is formed once per
                                      slice and evaluation instead of once per (member, slice) */
    int32_t expm_action;           /* 1: rank-one states (rank_one_chain) on an ensemble that fills the device, with control
                                      operators shared by the members (n = 5..32) or up to six of the members' own
                                      (n <= 16): exp(G_t) is applied to the two chains' vectors by its
                                      Taylor series (matrix-vector products only); no propagator is formed, so
                                      grape_get_trajectory has none to return -- GRAPE_FLAG_KEEP_COSTATES keeps the dense flow */
    int32_t prop_chain;            /* 1: rank-one states (rank_one_chain), 9 <= n <= 16, ensembles below expm_action's threshold
                                      (down to one problem): the expm kernel stores P_t and P_t^T and both vector chains run on
                                      them with one matrix-vector product per slice (chain_prop_kernel); with time_chunks >= 2
                                      on a chunked time axis (a workgroup per member and chunk) */
    /* ---- ABI v5 ---- */
    int32_t member_chunk;          /* members the workspace arrays (P_t, X_t, L_t) hold at a time: n_ensemble when everything
                                      fits, fewer when the ensemble's workspace exceeds 0.9 x the free device memory (or
                                      GRAPE_MAX_WORKSPACE_BYTES): an evaluation then walks the ensemble in blocks of this
                                      many members (src/solve.jl:166-187 is a serial loop with no such limit) -- same
                                      results bit for bit; grape_get_trajectory is not available on such a context */
    int32_t reserved0;
    /* ---- ABI v6 ---- */
    uint64_t workspace_budget_bytes; /* what the workspace arrays may take: 0.9 x the device memory free at grape_create minus the
                                      other buffers, or GRAPE_MAX_WORKSPACE_BYTES -- with member_chunk, what makes a run's
                                      chunk plan reproducible on another device / another day */
    int32_t scaled_controls;       /* 1 after grape_set_operators found the members' control operators to be member 0's times one
                                      real factor per member, B_{k,c} = s_k B_{0,c} (EnsembleProblem.B_g with amplitude
                                      inhomogeneity, src/problems.jl:33-41): the flows built on the per-slice control sum keep
                                      it (hoisted_controls = 1) and every member scales it by its s_k */
    int32_t propagator_blocks;     /* kernel_family 2, n >= 17: workgroups per member of the propagator launch (1: one launch
                                      forms the propagators and walks the chain) */
} grape_info;

/* Opaque RCCL bootstrap token (ncclUniqueId), see grape_comm_unique_id / grape_comm_attach. */
typedef struct grape_comm_id {
    char bytes[128];
} grape_comm_id;

/* Opaque handle of a rank's exchange mailbox (hipIpcMemHandle_t), see grape_ipc_export / grape_ipc_attach. */
typedef struct grape_ipc_handle {
    char bytes[64];
} grape_ipc_handle;

typedef struct grape_ctx grape_ctx;

/* ABI version of the loaded library (== GRAPE_ABI_VERSION of the header it was built from). */
int grape_abi_version(void);

/* Replaces init_GRAPE (src/grape_tools.jl:4-16) + init_ensemble's allocation
 * (src/tools.jl:42-53): creates the device workspace (propagators, forward states,
 * per-member gradients) for the given shape.  *out is NULL on failure. */
int grape_create(const grape_config *cfg, grape_ctx **out);

/* Frees everything the context owns (Julia: finalizer). NULL is accepted. */
int grape_destroy(grape_ctx *ctx);

/* One process per GPU (the layout torch.distributed / MPI launchers produce): every rank creates a
 * context for ITS contiguous member shard, rank 0 calls grape_comm_unique_id and ships the 128
 * bytes to the other ranks by any means, then every rank calls grape_comm_attach.  From then on
 * each grape_eval / grape_eval_device on that context ends in the one all-reduce(sum) of
 * [G, F] over the ranks (RCCL over xGMI, enqueued on the evaluation's own stream) that completes
 * src/solve.jl:171-191, so every rank returns the full-ensemble F and G.  Collective: all ranks
 * must call grape_comm_attach, and later the evaluations, in the same order.
 * librccl is loaded lazily (dlopen) by these two calls and by multi-device contexts only. */
int grape_comm_unique_id(grape_comm_id *out);
int grape_comm_attach(grape_ctx *ctx, const grape_comm_id *id, int32_t rank, int32_t n_ranks);

/* ABI v4.  The same one-process-per-GPU layout WITHOUT librccl: the K*N+1 doubles of src/solve.jl:171-191's sum travel
 * through mailboxes in device memory that the ranks open in each other through HIP IPC.  Every rank calls
 * grape_ipc_export(ctx, n_ranks, &h) (allocates its mailbox, returns 64 opaque bytes), the ranks exchange the handles by any
 * means (all-gather over the launcher's control plane), then every rank calls grape_ipc_attach(ctx, handles[n_ranks], rank,
 * n_ranks).  From then on grape_eval / grape_eval_device / grape_lbfgs on that context end in ipc_allreduce_kernel: each rank
 * stores its row into its slot of every rank's mailbox (one hop over xGMI), waits -- bounded -- until its own mailbox holds
 * all n_ranks rows, sums them in rank order (bitwise the sum every other rank, and an in-process group with the same
 * shards, gets) and publishes to its host as a single-GPU evaluation does.  A rank that never shows up turns into
 * GRAPE_ERR_COMM on the others after GRAPE_EVAL_TIMEOUT_S (5 s at least, 120 s at most), never into a hang.  The blocking
 * entry points return that code themselves; the device-pointer entry points (grape_eval_device, grape_eval_batch_device, the
 * evaluations inside grape_lbfgs) cannot -- for them an exchange that gave up writes NaN over [G, F] (never a partial sum)
 * and sets a host-visible word: grape_lbfgs stops with GRAPE_ERR_COMM at its next wait, and every later call on the context
 * returns GRAPE_ERR_COMM (the peers are out of step: the context is retired).  Ranks may share
 * a GPU (tests).  n_ranks <= 8; mutually exclusive with grape_comm_attach; collective like it. */
int grape_ipc_export(grape_ctx *ctx, int32_t n_ranks, grape_ipc_handle *out);
int grape_ipc_attach(grape_ctx *ctx, const grape_ipc_handle *handles, int32_t rank, int32_t n_ranks);

/* Uploads the per-member operators once -- what init_ensemble (src/tools.jl:42-53) produces by
 * calling A_g(k), B_g(k), XiG(k), XtG(k), packed contiguously by the glue:
 *   A  c128 (n,n,E)     B  c128 (n,n,K,E)     Xi, Xt  c128 (n,m,E)     wts  f64 (E)
 * (wts: EnsembleProblem.wts, src/problems.jl:40; pass {1.0} for a plain Problem). */
int grape_set_operators(grape_ctx *ctx, const double *A, const double *B, const double *Xi,
                        const double *Xt, const double *wts);

/* The closure body, src/solve.jl:164-196 (E>1) / :75-100 (E=1):
 *   F = sum_k w_k F_k ,  G[c,t] = sum_k w_k g_k[c,t]   with (F_k, g_k) = _fom_and_gradient_GRAPE!.
 * x: host (K,N) f64.  F (nullable): host f64.  G (nullable): host (K,N) f64 -- Optim passes
 * `nothing` for the one it does not need (src/solve.jl:189-195).  Blocks until F/G are written.
 * Non-finite x propagates NaN like the reference (no trapping). */
int grape_eval(grape_ctx *ctx, const double *x, double *F, double *G);

/* Same evaluation with device-resident input/output, asynchronous on `stream`
 * (a hipStream_t; NULL = the default stream):
 *   d_x   device (K,N) f64            d_fg  device f64[K*N + 1] = { G (K,N col-major), F }
 * This is the entry point the multi-GPU host layer uses: each rank evaluates its member shard
 * and a single all-reduce(sum) of d_fg over ranks completes src/solve.jl:171-191.
 * Nothing is synchronised; errors detectable at enqueue time are returned.
 * Stream ordering: the context has ONE workspace.  The library orders a later grape_eval /
 * grape_set_operators behind the last grape_eval_device (event wait), but two grape_eval_device
 * calls on DIFFERENT streams must be ordered by the caller.  Multi-device contexts: d_x and d_fg
 * live on device_ids[0]; the library fans x out to the other devices (peer copies). */
int grape_eval_device(grape_ctx *ctx, const double *d_x, double *d_fg, void *stream);

/* Extension beyond the reference (multi-start optimisation / line-search batches):
 * n_x <= max_batch independent control arrays evaluated against the same ensemble in ONE launch.
 *   x  host f64 (K,N,n_x)      F  host f64[n_x] (nullable)      G  host f64 (K,N,n_x) (nullable)
 * Entry b is exactly what grape_eval(ctx, x[:,:,b]) returns; with n_x = 1 the two calls are the same.
 * Needs grape_config.max_batch >= n_x (the workspace is sized for max_batch control arrays).  ABI v4: multi-device contexts
 * and attached communicators batch too -- every shard evaluates the n_x arrays, their rows cross the devices as one sum.
 * gradient = GRAPE_GRADIENT_EXACT batches as well: the stored trajectory is one control array's, so the arrays run one behind
 * the other on the stream (same results, one call, one completion). */
int grape_eval_batch(grape_ctx *ctx, int32_t n_x, const double *x, double *F, double *G);

/* Device-pointer form: d_x (K,N,n_x), d_fg f64[(K*N + 1) * n_x] = n_x blocks of { G, F }. */
int grape_eval_batch_device(grape_ctx *ctx, int32_t n_x, const double *d_x, double *d_fg, void *stream);

/* Device-resident L-BFGS: stands in for
 *     Optim.optimize(Optim.only_fg!(topt), x0, Optim.LBFGS(), optim_options)       src/solve.jl:138, :244
 * with x, g, the (s, y) history and the line-search trial points kept on the GPU; per evaluation the host
 * reads two scalars (phi, phi').  Optim's LBFGS() defaults are mirrored: memory m = 10, initial inverse-Hessian
 * scaling s'y / y'y, initial step 1 (InitialStatic), g_tol = 1e-8 on |g|_inf, f_tol = x_tol = 0, 1000 iterations, and
 * the line search is Hager-Zhang with LineSearches.jl's constants (delta 0.1, sigma 0.9, rho 5, epsilon 1e-6,
 * gamma 0.66, psi3 0.1, at most 50 evaluations): bracketing, secant^2 and bisection on phi(alpha), phi'(alpha), one
 * GRAPE evaluation per trial step.  line_search:
 *   0  Hager-Zhang; the initial step is taken at once when it satisfies the (approximate) Wolfe conditions
 *   1  Hager-Zhang exactly as Optim runs it behind InitialStatic (`mayterminate` false: the initial step is never
 *      accepted without a second evaluation).  ABI v5: LineSearches.jl's control flow to the letter -- max_linesearch counts
 *      passes of its bracketing / secant^2 loops as `linesearchmax` does (bisections inside a pass are not counted), a
 *      collapsed or flat bracket returns its lower end even at alpha = 0 (status 4: Optim then stops with "x converged"),
 *      a search that runs out of passes ends the run with status 3 as Optim's LineSearchException does (Optim takes the
 *      step of the exception's alpha first; this loop does not), two successive iterations without any change of F end it
 *      with status 1.  oracle/optim_lbfgs.py restates Optim.jl's LBFGS + LineSearches.jl's HagerZhang in NumPy and
 *      tests/test_gpu_lbfgs.py holds this mode to it step length by step length.
 *   2  the factor-2 ladder of ABI v2: `probes` step lengths alpha, alpha/2, ... per BATCHED launch
 *      (grape_config.max_batch >= probes), the largest with sufficient decrease (c1 = 1e-4), preferring the strong Wolfe
 *      curvature condition (c2 = 0.9)
 * Multi-device contexts (n_devices >= 2) and contexts with an attached communicator / mailbox exchange default to modes 0
 * and 1 (one sharded evaluation per trial step); ABI v4: with max_batch >= 2 they take mode 2 as well -- the B probes of a
 * ladder are ONE batched sharded evaluation with ONE exchange of B rows.  The vectors live on the first device (every
 * rank's device).  With grape_comm_attach / grape_ipc_attach all ranks must call grape_lbfgs together (they take identical
 * decisions on identical [G, F]).
 * A Hager-Zhang search that cannot bracket -- the reference's UnitaryGate gradient is not the derivative of its figure
 * of merit -- hands that iteration to the ladder (single-device) or ends with status 3.
 * The gradient is whatever the GRAPE evaluation returns, with the reference's conventions. */
typedef struct grape_lbfgs_options {
    int32_t memory;            /* m; 0 = 10                                              */
    int32_t max_iterations;    /* 0 = 1000                                               */
    double  g_tol;             /* < 0 = 1e-8; stop when |g|_inf <= g_tol                 */
    double  f_tol;             /* stop when |f - f_prev| <= f_tol |f|  (Optim f_tol; 0 = off) */
    int32_t max_linesearch;    /* evaluations per line search before giving up; 0 = 50   */
    int32_t probes;            /* ladder search: step lengths per launch, 1..8; 0 = automatic */
    /* ---- ABI v3 ---- */
    int32_t line_search;       /* 0, 1 Hager-Zhang (see above), 2 ladder                 */
    int32_t reserved;
} grape_lbfgs_options;

typedef struct grape_lbfgs_result {
    double  minimum;           /* Optim's res.minimum                                    */
    double  g_norm;            /* |g|_inf at the minimiser                               */
    double  seconds;           /* wall time of the whole optimisation                    */
    int32_t iterations;
    int32_t evaluations;       /* control arrays evaluated (probes count individually)   */
    int32_t status;            /* 0 g_tol reached, 1 f_tol reached, 2 max_iterations, 3 line search failed,
                                  4 (ABI v5, line_search = 1) zero step: Optim's "x converged" with x_tol = 0 */
    int32_t probes;            /* step lengths per launch actually used                  */
    /* ---- ABI v3 ---- */
    int32_t line_search;       /* the mode that ran                                      */
    int32_t ladder_fallbacks;  /* iterations whose Hager-Zhang search could not bracket  */
} grape_lbfgs_result;

/* x0: host (K,N) f64 initial controls (Problem.guess); x_min: host (K,N) f64, receives res.minimizer.
 * opts may be NULL (all defaults). */
int grape_lbfgs(grape_ctx *ctx, const double *x0, const grape_lbfgs_options *opts, double *x_min,
                grape_lbfgs_result *result);

/* ABI v5.  The last grape_lbfgs run on this context, per iteration: the accepted step length and the number of control
 * arrays evaluated up to the end of that iteration (what Optim's trace shows as `alpha` and f_calls).  *count = iterations
 * recorded; at most `capacity` entries are written (alphas / evals may be NULL). */
int grape_lbfgs_get_trace(const grape_ctx *ctx, double *alphas, int32_t *evals, int32_t capacity, int32_t *count);

/* Debug/parity accessors (valid after an evaluation; needs GRAPE_FLAG_MEMBER_RESULTS; after a batched
 * evaluation they refer to control array 0):
 * per-member unweighted results, as the reference's `gradient[k,:,:]` and the F_k summands:
 *   foms  host f64[E] (nullable)      grads  host f64 (K,N,E) (nullable) */
int grape_get_member_results(grape_ctx *ctx, double *foms, double *grads);

/* The stores the reference keeps per member (src/grape_tools.jl:4-16), for parity tests:
 *   props     c128 (n,n,N)     propagators[t],  t = 0..N-1
 *   states    c128 (n,m,N+1)   fwd_state_store[t], t = 0..N   (states[0] = Xi); the fast flows rebuild
 *                              them on the fly (grape_info.states_stored == 0): then only under
 *                              GRAPE_FLAG_KEEP_COSTATES, else GRAPE_ERR_NOT_READY.
 *   costates  c128 (n,m,N+1)   bwd_costate_store[t], t = 0..N (costates[N] = Xt);
 *                              needs GRAPE_FLAG_KEEP_COSTATES, else GRAPE_ERR_NOT_READY.
 * Any of the three may be NULL. */
int grape_get_trajectory(grape_ctx *ctx, int32_t member, double *props, double *states,
                         double *costates);

/* Sum and count of the sweep kernel's HIP-event durations recorded since the last reset
 * (GRAPE_FLAG_TIME_KERNELS).  Synchronises the recorded events.  reset != 0 clears them. */
int grape_get_kernel_time(grape_ctx *ctx, double *total_ms, int64_t *launches, int32_t reset);

/* The individual durations behind grape_get_kernel_time since its last reset (at most 65536 are kept): the most recent
 * min(capacity, *count) of them, oldest first.  total_ms[i] = all sweep kernels of one evaluation; first_ms[i] (nullable)
 * = the part in front of the chain kernels (n = 5..32: control-sum pre-pass + expm kernel, pw_prop_save!,
 * src/timeevolution.jl:98-110; 0 for n <= 4, where one kernel does everything).  Multi-device contexts report the
 * first device.  *count (nullable) receives the number of durations available. */
int grape_get_kernel_samples(grape_ctx *ctx, double *total_ms, double *first_ms, int64_t capacity, int64_t *count);

/* ABI v4.  The kernels the most recent evaluation launched (grape_eval / grape_eval_device / one batch), by name and in launch
 * order, ';'-separated and NUL-terminated in buf (at most capacity bytes; buf may be NULL): the names rocprofv3 prints, without
 * namespace and template arguments -- e.g. "action_rows_kernel;action_parts_kernel;action_forms_sparse_kernel;reduce_stage1;
 * reduce_stage2".  Which kernels serve src/GRAPE.jl:25-96 depends on what grape_set_operators found (grape_info) and on the
 * ensemble size; benchmarks label their lines with this instead of guessing.  Multi-device contexts report the first device.
 * Returns the buffer size the complete list needs (> 0), or a negative status. */
int grape_get_kernel_names(const grape_ctx *ctx, char *buf, int32_t capacity);

/* Host-side cost of the sharded grape_eval of a multi-device context (n_devices >= 2), means over the evaluations since the
 * last reset: out[0] = evaluations, then microseconds: out[1] writing x into every shard's buffer, out[2] first to last
 * shard launch (the issue skew: one issuing thread per shard), out[3] issuing the sum (peer copies + reduction, or
 * the grouped ncclAllReduce + publication), out[4] waiting for [G, F], out[5] the whole call.  out: double[6]. */
int grape_get_group_timing(grape_ctx *ctx, double *out, int32_t reset);

/* Diagnostic (GRAPE_FLAG_PHASE_STAMPS): the stamps of the last evaluation, 8 uint64 per wave,
 * waves ordered (member, wave-in-member): [0..4] shader clock at start / after propagators /
 * after scan / after forward sweep / end, [5],[6] 100 MHz real-time counter at start / end,
 * [7] where the wave ran: HW_REG_XCC_ID << 32 | HW_REG_HW_ID.
 * out: host uint64[capacity]; *count receives the number of values available. */
int grape_get_phase_stamps(grape_ctx *ctx, uint64_t *out, int64_t capacity, int64_t *count);

int grape_get_info(const grape_ctx *ctx, grape_info *info);

/* Message of the last error on this context (ctx == NULL: of the last failed grape_create
 * on the calling thread).  Never NULL. */
const char *grape_last_error(const grape_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* GRAPE_HIP_H */
