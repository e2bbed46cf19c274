// grape_kernels.hpp -- launch interface between the C-ABI host layer (grape_api.cpp) and the
// gfx950 kernels (sweep_small.hip, reduce.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace grape {

// Every kernel launch of the library goes through GRAPE_LAUNCH: the launchers' own names, in launch order, are what
// grape_get_kernel_names reports for the last evaluation (bench.py labels its lines with them; they are the names rocprofv3
// prints, without template arguments).  log_kernel is a no-op on threads that are not inside an evaluation's launches.
void log_kernel(const char *expr);
#define GRAPE_LAUNCH(KERNEL, ...)                        \
    do {                                                 \
        ::grape::log_kernel(#KERNEL);                    \
        hipLaunchKernelGGL(KERNEL, __VA_ARGS__);         \
    } while (0)
#define GRAPE_LAUNCH_AS(NAME, KERNEL, ...)               \
    do {                                                 \
        ::grape::log_kernel(NAME);                       \
        hipLaunchKernelGGL(KERNEL, __VA_ARGS__);         \
    } while (0)


// Scaling threshold of the expm: a generator of norm bound theta <= kTheta8 goes through the degree-8 Taylor polynomial
// unscaled, larger ones are halved s times first and the result squared s times.  The truncation error of the
// polynomial is theta^9 / 9! = 3.7e-16 at 0.08 (absolute, |P| ~ 1) -- the size of one rounding error of its three
// products, and 7e-13 if it added up coherently over 2000 slices, against the 1e-10 parity bar.  (Rounds 1-2 used
// 0.05, the bound for a backward error of 2^-53 RELATIVE TO |G|; every slice of C4 and C5 sits below 0.08, 26 % / 40 %
// of them above 0.05: one squaring -- a fourth matrix product -- each, and in C4 all of them in the far-detuned
// members, whose workgroups then set the kernel's run time.)
constexpr double kTheta8 = 0.08;

// optional host-visible completion signal of an evaluation's FINAL kernel (reduce.hip: signal_done);
// flag == nullptr: none (device-pointer entry points, intermediate kernels)
struct DoneSignal {
    unsigned *counter = nullptr;               // device: workgroups of the final kernel that have finished
    unsigned long long *flag = nullptr;        // coherent pinned host memory (device address)
    unsigned long long seq = 0;                // value to publish
    double *host_out = nullptr;                // reduce kernels: mapped host destination of [G, F]; the kernel's
                                               // `fg` argument is then a DEVICE staging buffer that the last
                                               // workgroup copies out in one coalesced burst before publishing
    const double *stage_base = nullptr;        // reduce_stage2 of a batched evaluation: start of the whole staging
    int n_total = 0;                           // buffer and its length (all n_x blocks), copied by the last launch
    // grape_lbfgs' line-search probe closed by the reduction itself (done_signal.hpp): instead of copying [G, F] out, the
    // last workgroup publishes phi = F and phi' = G . probe_dir (and phi'(0) = probe_sc[2]) in probe_out[0..2]
    const double *probe_dir = nullptr;         // device: the search direction (K N doubles)
    const double *probe_sc = nullptr;          // device: the L-BFGS scalars (LbfgsState::sc)
    double *probe_out = nullptr;               // mapped host memory (device address): LbfgsState::host_sc + 8
    // round 5: publication WITHOUT a device-side fan-in (reduce_rows_mf_kernel): workgroup b writes its outputs straight to
    // host_out, fences at system scope and stores `seq` into mflags[b] (mapped host memory); the host waits for all of them
    unsigned long long *mflags = nullptr;
};
constexpr int kMaxMflags = 1000;               // flags behind grape_ctx::h_flag (8 KB: [0] flag, [1] exchange failure, [8..] these)
// workgroups (= host flags) reduce_rows_mf_kernel publishes with for Q outputs x n_x control arrays; 0: not applicable
int reduce_rows_mflags(int Q, int n_x);

constexpr int kStampSlots = 8;   // [0..4] shader clock at phase boundaries, [5],[6] 100 MHz real time

// Device-side view of one context.  All complex data is interleaved double2 {re, im}.
struct SweepParams {
    // inputs
    const double2 *ops;   // per member: [A | B_0..B_{K-1} | Xi | Xt], each n*n col-major  (E blocks)
    const double *x;      // (K, N) col-major controls, shared by all members
    // workspace (lane-major "chunk" layout):
    //   element e of slice t = L*S + j of member k at  ((k*S + j)*n*n + e)*LT + L
    double2 *props;       // P_t
    double2 *states;      // X_t (state BEFORE slice t), t = 0..N-1
    double2 *costates;    // L_t (debug only, GRAPE_FLAG_KEEP_COSTATES), same layout
    // outputs
    double *member_out;   // (K*N + 1) per member: unweighted g_k (K,N col-major), then F_k; may be NULL
    const double *wts;    // ensemble weights w_k
    double *block_out;    // (K*N + 1) per workgroup: sum over its members of w_k * [g_k, F_k]
    unsigned long long *stamps;   // diagnostic (NULL in production): kStampSlots per (member, wave)
    double *xg_scratch;   // NULL: controls/gradient staged in LDS; else per-workgroup global scratch
    int32_t K, N, E;
    int32_t S;            // slices per lane
    int32_t LT;           // lanes per member = 64 * W
    int32_t MPB;          // members per workgroup (MPB * LT threads)
    int32_t BPX;          // workgroups per control array = ceil(E / MPB)
    int32_t n_x;          // control arrays evaluated by this launch (grid = BPX * n_x workgroups)
    uint32_t sk_magic;    // floor(2^32 / (S*K)) + 1: q / (S*K) == __umulhi(q, sk_magic)
    int32_t s_forced;     // expm squarings, -1 = per slice from the norm
    int32_t variant;      // 0 in-place, 1 static
    int32_t plast_lds;    // pair kernel, set by its launcher: every chunk's LAST propagator also stays in LDS, where
                          // the backward sweep (which starts with it) reads it instead of HBM
    double dt;
    // single workgroup (BPX == 1, n_x == 1: the single-problem closure, src/solve.jl:63-143): its weighted row IS
    // [G, F], so the sweep kernel writes it to `direct_dst` itself (device buffer or mapped host memory) and, when
    // direct_flag is set, publishes direct_seq there -- no reduce launch.  nullptr: off
    double *direct_dst;
    unsigned long long *direct_flag;
    unsigned long long direct_seq;
    // exact gradient of a UnitaryGate problem with Hermitian generators (pair kernel, unitary flow): the backward sweep also
    // stores W_t = X_t L_{t+1}' = M_t P_t' per slice into `states` (unused by this flow) and tr(M) per member into `zphi`
    // -- all that exact_pair_kernel needs of the trajectory (no debug flow, no X_t / L_t dumps)
    int32_t dump_w1;
    double *zphi;         // (2 per member and control array) Phi = tr(L' X) = tr M
    int32_t tune;         // pair kernel, set by its launcher from GRAPE_PAIR_TUNE (tuning experiments; 0 in the product)
    int32_t vec;          // pair kernel, general flow, left multiplication, n = 4: the states are n x 1 (column 0 of the padded
                          // matrices) -- the sweep back runs on vectors and phase A stores no in-chunk prefixes
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of once per launch: the driver call sits
// on the host's critical path in front of an evaluation's launches (grape_api.cpp; thread-safe: group contexts launch from one
// thread per shard).  Returns hipSuccess without a call when `lds` is within what the kernel was already granted there.
hipError_t ensure_dynamic_lds(const void *fn, size_t lds);

// sandwich == 0: UnitaryGate chain; 1: State/CoherenceTransfer sandwich chain.
// mode: 0 general flow, 1 general flow + costates stored (debug), 2 unitary flow (all
// generators Hermitian).  Returns hipSuccess or the launch error.  n must be 2, 3 or 4.
hipError_t launch_sweep_small(int n, int sandwich, int mode, const SweepParams &p, hipStream_t stream);
int sweep_small_max_waves(int n);   // W limit of the register-resident kernel for this n
// dynamic LDS a workgroup of the small-n sweep needs for this decomposition
size_t sweep_small_lds_bytes(int n, int MPB, int LT, int S, int K, bool xg_in_lds);

// lane-pair edition of the same kernel (sweep_pair.hip, n = 2, 4): a time chunk is shared by two adjacent
// lanes, LT/2 chunks per member; same SweepParams, workspace stride LT/2 instead of LT
hipError_t launch_sweep_pair(int n, int sandwich, int mode, const SweepParams &p, hipStream_t stream);
int sweep_pair_max_waves(int n);    // 0: no pair kernel for this n
size_t sweep_pair_lds_bytes(int n, int MPB, int LT, int S, int K, bool xg_in_lds, bool plast = false);

// ---- tile (MFMA) family, n = 5..32, zero-padded to 16*NT ------------------------------------
// All matrices are "D-layout dumps" (tile.hpp): (16 NT)^2 double2 each, TSZ = NT*NT*256.
struct TileParams {
    const double2 *ops;   // per member: [A | B_1..B_K | B_1^T..B_K^T | Xi | Xt]  ((2K+3) dumps)
    const double *x;      // (K, N) col-major controls
    double2 *props;       // P_t dump at (k*N + t)*TSZ
    double2 *states;      // X_t dump, same indexing
    double2 *costates;    // L_t dump (debug, GRAPE_FLAG_KEEP_COSTATES)
    double *member_out;   // as in SweepParams
    int32_t K, N, E, n;   // E = wavefront-level units: members, or member PAIRS when pack2
    int32_t pack2;        // n <= 8: two members share one 16x16 tile as a block-diagonal pair
    int32_t E_members;    // true member count (rows of member_out)
    int32_t E_plan;       // units of the WHOLE ensemble (= E unless this launch is one chunk of a member-chunked evaluation):
                          // what the launchers' flow decisions look at, so that a chunked run takes the unchunked run's kernels
    int32_t s_forced, variant;
    int32_t bt_in_lds;    // set by the launcher: the K transposed control operators are cached in LDS
    int32_t stage_ops;    // set by the launcher: prop kernel stages the generators in LDS
    int32_t prop_slices;  // set by the launcher: slices one workgroup of the prop kernel walks
    int32_t unitary;      // every generator Hermitian: chain kernel carries M_t = P' M P, no stored states
    int32_t herm_states;  // every Xi, Xt Hermitian (density operators): [X, L'] = Y - Y' with one product
    int32_t split_at;     // set by the launcher: first slice of the second wave (chain_tile_split_kernel)
    int32_t split_expm;   // set by the launcher: chain_tile_split_kernel forms P_t itself in its phase 1 (no expm kernel ran)
    int32_t n_x;          // control arrays evaluated by this launch (batched evaluation); x is (K, N, n_x)
    int32_t thin;         // rank-one states: 1 = sweep_thin.hip's matrix-vector chain (the prop kernel then stores P_t
                          // TRANSPOSED for odd t; `states` holds the forward pass's vector records, N + 1 per member);
                          // 2 = chain_prop_kernel of action_thin.hip (P_t stored as it is, records element-major in
                          // `states` / `wrec`, bilinear forms by the action_forms kernels)
    int32_t herm_ctrl;    // every control operator B_c Hermitian (thin chain: one bilinear form per control)
    const double2 *vecs;  // thin: per member [v0 | wT], 16 complex each, zero padded
    int32_t cus;          // compute units of the device
    // time-parallel unitary chain (small ensembles, sweep_tile.hip): the N slices of a unit are cut into tp_chunks chunks
    // of tp_S slices, one wavefront per chunk; 0 = the sequential chain (one wavefront walks all N slices)
    int32_t tp_chunks, tp_S;
    int32_t tp_window;    // sweep_grid.hip (round 6): 1 = grid_chain_kernel works on ONE chunk of the time axis per workgroup
                          // (blockIdx.z), its first state from tp_u, its last costate from tp_r (written by the boundary scan)
    int32_t tp_scan;      // 1 = this launch IS the boundary scan (the chain kernel on the chunk products): no output rows
    double2 *tp_q;        // [control array][unit][chunk] chunk products Q_c = P_hi-1 ... P_lo (D-layout dumps)
    double2 *tp_r;        // same shape: R_c = Q_C-1 ... Q_c+1, the product of everything after chunk c
    double2 *tp_m;        // [control array][unit]: M_N
    // unitary flow, many chunks: two-level scan.  Groups of tp_gsize consecutive chunks; tp_r then holds the product of
    // the chunks after c INSIDE its group, tp_a the product of the groups after group j (tp_groups of them; 0 = one level)
    int32_t tp_groups, tp_gsize;
    double2 *tp_a;        // [control array][unit][group]: A_j, followed by the group products themselves (scan input)
    double2 *tp_qt;       // general flow only (else null): Q_c^T dumps, and
    double2 *tp_u;        //   U_c^T, U_c = Q_c-1 ... Q_0 the product of everything before chunk c
    double *tp_z;         // [control array][unit][lane][2]: tr(X_N' L_N) of the lane's member (sandwich); rank-one chain: s
    double2 *tp_vec;      // rank-one chain: [control array][unit][chunk][v at the chunk's start | w at its end], 16 complex each
    int32_t fuse_fwd;     // set by the launcher (thin): one workgroup of prop_tile_kernel walks ALL slices of a member and
                          // runs the forward vector chain v_{t+1} = P_t v_t on the propagators it still holds in
                          // registers, writing the records; chain_thin_kernel then only runs its backward pass
                          // (P_t is read from HBM once instead of twice)
    // sparse control operators (every B_c of every member has at most kSparseMax = 256 non-zeros -- Pauli-type controls, sums of them):
    // per member and control sp_nz entries, zero padded: sp_coef = B_c[i][j], sp_addr = position of M[j][i]
    // in the wave's LDS image of M (row j, column i, row stride 16 NT + 1).  The gradient traces tr(B_c M_t) then
    // read 64 entries instead of a dense transposed operator per control and slice.
    int32_t sparse;
    int32_t sp_nz;            // list length of this context: 64, 128, 192 or 256 (the longest operator's non-zeros, rounded up:
                              // a lane owns sp_nz / 64 entries of every list -- sums of a few Pauli strings, global drives)
    const double2 *sp_coef;   // [unit][K][sp_nz]
    const int32_t *sp_addr;   // [unit][K][sp_nz]
    // member-invariant control operators (prop_hoist.hip): the control sum is formed once per slice and evaluation
    int32_t hoist;            // set by the host layer: every member has the same B_c (and the ensemble is worth a pre-pass)
    const double2 *ha;        // [unit] D-layout dumps of A'_k = (-i dt) A_k
    const double *ha_norm;    // [unit] |A'_k|_1 bound (max column sum of |re| + |im|) / theta8
    double2 *gc;              // [control array][slice] D-layout dumps of Gc_t = (-i dt) sum_c x[c,t] B_c
    double *gcn;              // [control array][slice] |Gc_t|_1 bound / theta8
    // rank-one states + member-invariant controls (action_thin.hip): exp(G_t) applied to the chains' vectors, no propagators
    int32_t action;           // set by the host layer
    const double2 *act_a;     // [unit][2][256] row-major [A'_k | A'_k'] (conjugate transpose), zero padded to 16 x 16
    const double *act_an;     // [unit] max(|A'_k|_1, |A'_k|_inf)
    double2 *props_t;         // thin == 2: P_t^T dumps, same indexing as props
    double2 *wrec;            // backward chain's records w_0 .. w_N, element-major (the vector flow: = props; thin == 2: own buffer)
    // Per-member control operators that are member 0's times a scalar, B_{k,c} = s_k B_{0,c} (amplitude inhomogeneity,
    // EnsembleProblem.B_g, src/problems.jl:33-41): the hoisted flows keep their pre-pass -- Gc_t from member 0's operators
    // (ops_ref / act_b_ref: the WHOLE ensemble's first unit, also in a member-chunked launch) -- and a member forms
    // G = A'_k + s_k Gc_t, |G| <= |A'_k| + |s_k| |Gc_t|.  ctrl_scale: s_k per member of this launch; nullptr: all ones
    const double *ctrl_scale;
    const double2 *ops_ref;
    const double2 *act_b_ref;
    int32_t act_shared;       // 1: one set of control operators for every member (pre-pass forms the control sums); 0: per member
                              //    (n <= 16, K <= 6): act_b / act_bf / act_bs / act_bo carry a leading member index
    const double *act_bn;     // (per-member controls) [unit][K] max(|B'_kc|_1, |B'_kc|_inf)
    const double2 *act_b;     // [K][2][256] row-major [B'_c | B'_c'], B'_c = (-i dt) B_c
    const double2 *act_bf;    // [K][256] row-major B_c (the gradient's bilinear forms)
    int32_t act_R;            // > 0: every row of every B_c has at most act_R non-zeros (1, 2, 3, 4 or 6) -- the forms read
    const double2 *act_bs;    //      [K][16][act_R] values B_c[i][j], zero padded, and
    const int32_t *act_bo;    //      [K][16][act_R] the byte offsets 1024 j of their columns in the kernel's LDS tile
    double2 *act_g;           // [control array][slice][2][256]: [Gc_t | Gc_t'], written by the pre-pass (16 x 16, shared controls:
                              //    six planes of 256 doubles [re | im | -im] of Gc_t, then of Gc_t': action_parts_kernel)
    double *act_gn;           // [control array][slice] max(|Gc_t|_1, |Gc_t|_inf)
    // ONE problem (E = 1, one control array) on a flow that ends in a forms kernel of action_thin.hip: that kernel writes the
    // weighted row [G, F] itself and its last workgroup publishes it -- no reduce launch.  fold_fg: destination / staging
    double *fold_fg;          // nullptr: off
    const double *fold_wts;
    DoneSignal fold_done;
    hipEvent_t ev_mid;        // timing (GRAPE_FLAG_TIME_KERNELS): recorded behind the expm kernel, in front of the chain kernels; or null
    double dt;
};
constexpr int kSparseMax = 256;   // longest list (non-zeros of one control operator); a context's own length: TileParams.sp_nz
// the chain over rank-one states (n = 9..16, one member per wavefront); called by launch_sweep_tile when p.thin
hipError_t launch_chain_thin(int sandwich, const TileParams &p, hipStream_t stream);
// rank-one states, member-invariant controls: the evaluation on vectors (action_thin.hip); called by launch_sweep_tile when p.action
hipError_t launch_action_thin(int sandwich, const TileParams &p, hipStream_t stream);
// rank-one states, n = 9..16, propagators from the expm kernel (p.thin == 2): both vector chains on DPP FMACs, then the forms
hipError_t launch_chain_prop(int sandwich, const TileParams &p, hipStream_t stream);
// sweep_coop.hip: four waves per product for the chunked unitary flow of few 32 x 32 units (single problems, small ensembles)
bool coop_applies(const TileParams &p, int sandwich, bool keepl);
hipError_t launch_coop_chunk_product(const TileParams &q, hipStream_t stream);
hipError_t launch_coop_scan_group(const TileParams &q, hipStream_t stream);
hipError_t launch_coop_scan(int sandwich, const TileParams &q, hipStream_t stream);
hipError_t launch_coop_chain_unitary(int sandwich, const TileParams &q, hipStream_t stream);
int tile_count(int n);    // NT for this n (0: not a tile-family size)
int tile_fuse_forward(const TileParams &p);   // thin: forward vector pass runs inside prop_tile_kernel for this launch?
hipError_t launch_sweep_tile(int n, int sandwich, bool keep_costates, const TileParams &p, hipStream_t stream);
// ---- any other operator dimension (n = 1, n > 64): sweep_any.hip, plain vector FP64, the reference's general flow ----------
struct AnyParams {
    const double2 *ops;       // per member [A | B_1..B_K | Xi | Xt], n x n column-major each
    const double *x;          // (K, N, n_x)
    double2 *props, *states;  // N matrices per (control array, member)
    double2 *costates;        // nullable (GRAPE_FLAG_KEEP_COSTATES)
    double2 *scratch;         // 6 matrices per (control array, member, propagator block)
    double *member_out;       // rows of K N + 1 doubles; row of (control array z, member k) at (z * E_rows + k)
    int32_t n, K, N, E, E_rows, n_x, sand, s_forced, variant;
    double dt;
    // Few members (fewer than the device has compute units -- the rule at these sizes): the propagators, which do not depend
    // on one another, are formed by prop_blocks workgroups per member (blockIdx.z: slices [z spb, (z + 1) spb)) in a launch of
    // their own (phase 1), the chain -- sequential in time -- by one workgroup per member behind it (phase 2).  prop_blocks <=
    // 1: one launch does both (phase 0).
    int32_t prop_blocks, phase;
    // Chunked time axis (tp_chunks > 1; round 6): phase 3 = chunk products Q_c (workgroup per (member, chunk): blockIdx.z) into
    // tp_q, phase 4 = boundary scan (the chain on the chunk products: states at the chunks' starts -> tp_u, costates at their
    // starts -> tp_r, no output rows), phase 5 = the chain on the slices of chunk blockIdx.z.  The scratch matrices are per
    // (control array, member, max(prop_blocks, tp_chunks)).
    int32_t tp_chunks, tp_S;
    double2 *tp_q, *tp_u, *tp_r;
    hipEvent_t ev_mid;        // nullable: recorded behind the propagator launch (prop_blocks > 1 only)
    int32_t abl;              // diagnostic ablation mask (GRAPE_ANY_ABL; wrong results): 1 no MFMAs, 2 no traces, 4 no operand fetch,
                              // 8 no H build, 16 no norm, 32 no Taylor products, 64 no stores of the H build's first pass
    const double2 *shared_b;  // nullable: the members' control operators are identical -- member 0's [B_1..B_K] for everybody
                              // (K n^2 16 B that stay in L2 instead of E times as much streamed from memory per slice)
    // nullable (all or none): shared control operators with few non-zeros (grape_host::build_any_sparse) -- the H build and the
    // gradient traces walk these lists instead of K dense n x n operators per slice
    const int32_t *sp_tidx, *sp_tptr, *sp_ectl, *sp_cptr, *sp_caddr;     // touched elements, their entries; by control
    const double2 *sp_ecoef, *sp_ccoef;
    int32_t sp_ntouch;
};
hipError_t launch_sweep_any(const AnyParams &p, hipStream_t stream);
int any_prop_blocks(int n, int N, long units, int cus);    // how many propagator blocks per member the launcher will use

// n = 33..64 (NT = 3, 4; sweep_grid.hip): a workgroup of NT x NT waves per matrix, the reference's general flow
hipError_t launch_sweep_grid(int NT, int sandwich, bool keep_costates, const TileParams &p, hipStream_t stream);
hipError_t launch_grid_prop(int NT, const TileParams &p, hipStream_t stream);
hipError_t launch_grid_exact(int NT, int sandwich, const TileParams &p, int objective, hipStream_t stream);   // exact gradient, n = 33..64     // its expm launches alone (P_t dumps as the tile family's)
// Gc_t = (-i dt) sum_c x[c,t] B_c per slice and control array + its norm bound (prop_hoist.hip; member-invariant controls)
hipError_t launch_ctrl_sum(int NT, const TileParams &p, hipStream_t stream);
// prop_hoist.hip: control-sum pre-pass + the expm kernel on A'_k + Gc_t; q = the launcher's parameters (prop_slices, fuse_fwd set)
hipError_t launch_prop_hoist(int NT, const TileParams &q, hipStream_t stream);
bool tile_chain_is_split(const TileParams &p, bool keep_costates);   // the two-wave time-split chain: no full X_t store

// G[q] = sum_k w_k member_out[k][q]  for q in [0, Q)  (Q = K*N + 1; the last entry is F).
// partial: scratch of ksplit*Q doubles.  Deterministic (fixed summation tree).
hipError_t launch_reduce(const double *member_out, const double *wts, double *partial, double *fg,
                         int E, int Q, int ksplit, hipStream_t stream, DoneSignal done = DoneSignal());
int reduce_ksplit(int E);
// fg[q] = sum_b rows[b][q] over NB already-weighted rows (one launch, fixed summation tree).
// n_x > 1: independent reductions, rows [x*NB, (x+1)*NB) -> fg + x*Q
hipError_t launch_reduce_rows(const double *rows, double *fg, int NB, int Q, int n_x, hipStream_t stream,
                              DoneSignal done = DoneSignal());

// fg[q] = sum_g rows.p[g][q] over the G <= 8 shards of a multi-device context, in shard order; the rows are read in place
// (peer access), no staging copies
constexpr int kMaxShards = 8;
struct ShardRows {
    const double *p[kMaxShards];
    int n;
};
hipError_t launch_reduce_shards(const ShardRows &rows, double *fg, int Q, hipStream_t stream, DoneSignal done = DoneSignal());

// ---- the cross-shard sum without a waiting kernel or a stream dependency (round 4) --------------------------------------
// In-process group: every shard launches shard_arrive_kernel on its OWN stream behind its reduction (no event, no
// hipStreamWaitEvent on the first device's stream, no reduce_shards launch).  Block b of shard g adds 1 to arrive[b]
// (system scope; fine-grained memory on the first device); the block whose add was the G-th sums outputs
// [256 b, 256 b + 256) over the G rows IN SHARD ORDER -- read where the shards' reductions left them (peer access; a row is
// complete and written back at its producer's kernel boundary, before any block of that shard arrives) -- stores them to
// `out` / the host buffer, and the last of those summing blocks publishes the evaluation.  Nobody waits for anybody:
// several shards on one GPU (tests) cannot deadlock on a shared hardware queue.
struct ArriveParams {
    ShardRows rows;
    int Q;
    unsigned *arrive;             // [ceil(Q / 256)] arrivals per block column
    unsigned *finished;           // summing blocks that have finished (behind the counters of the largest batch)
    double *out;                  // device result on the first device (nullable)
    DoneSignal done;              // host publication (host_out, flag, seq); counter unused
};
hipError_t launch_shard_arrive(const ArriveParams &p, hipStream_t stream);

// One process per GPU without RCCL (grape_ipc_export / grape_ipc_attach): every rank owns a MAILBOX in fine-grained device
// memory that its peers have opened through hipIpcOpenMemHandle -- [2 parities][n_ranks][Qpad] doubles, then
// [2][ceil(Q / 256)] arrival counters.  Block b of rank r stores its 256 outputs of the rank's row into slot r of EVERY mailbox
// (one xGMI hop each), releases them at system scope and adds 1 to that block column's counter in every mailbox; then
// waits until its OWN mailbox shows the arrivals of all n_ranks (bounded spin), sums the n_ranks slots in rank order --
// bitwise what the in-process group and every other rank get -- and stores / publishes like the single-GPU path.  Slots
// alternate with the evaluation's parity: a rank that runs ahead writes the other half while a slow rank still reads.
struct IpcParams {
    const double *own_row;        // this rank's [G, F], complete (previous kernel of the stream)
    int Q, Qpad, rank, n_ranks, parity;
    double *mbox[kMaxShards];     // every rank's mailbox (own included), as this process addresses it
    unsigned long long target;    // arrivals the own mailbox's counters must show: n_ranks x evaluations of this parity so far
    long long spin_limit;         // polls (with s_sleep) before giving up: the evaluation is then published as failed
    double *out;                  // device result (nullable)
    DoneSignal done;              // host publication; done.counter: agent-scope block counter of this device
    unsigned long long *fail_word;    // mapped host memory (nullable): set non-zero by a block that gave up -- the only failure
                                      // channel of the device-pointer path, which publishes no flag; `out` is NaN then
};
constexpr unsigned long long kSeqFailed = 1ull << 63;      // flag value = seq | kSeqFailed: the exchange gave up
size_t ipc_mailbox_bytes(int Q, int n_ranks);
hipError_t launch_ipc_allreduce(const IpcParams &p, hipStream_t stream);

// ---- exact gradient / functional path (exact_grad.hip), n <= 4 ---------------------------------------
struct ExactParams {
    const double2 *ops;       // as SweepParams.ops (prescaled generators)
    const double *x;          // (K, N)
    const double2 *props, *states, *costates;   // the sweep's debug-flow stores (chunk-major workspace layout)
    double *member_out;       // (K*N + 1) per member: exact gradient, then the objective
    int32_t K, N, E;
    int32_t S, CH;            // workspace decomposition: slice t = L*S + j of member k at ((k*S + j)*n*n + e)*CH + L
    int32_t s_forced, variant;
    int32_t objective;        // 0: the GRAPE figure of merit (fom_func), 1: C1 functional for every system type (ADGRAPE)
    int32_t herm_states;      // every Xi, Xt Hermitian: the sandwich's two derivative directions coincide
    int32_t w1_in;            // UnitaryGate, pair kernel: `states` holds W_t = X_t L_{t+1}' (SweepParams::dump_w1), `zphi` Phi
    const double *zphi;
};
hipError_t launch_exact_grad(int n, int sandwich, const ExactParams &p, hipStream_t stream);
// the same for the tile family (exact_tile.hip): reads the debug flow's props / states / costates dumps of TileParams
struct TileParams;
hipError_t launch_exact_tile(int n, int sandwich, const TileParams &p, int objective, hipStream_t stream);

// ---- device-resident L-BFGS (lbfgs.hip) -----------------------------------------------------------
constexpr int kLbfgsMaxPer = 16;       // vector elements per thread of the 1024-thread workgroup: K*N <= 16384
constexpr int kLbfgsMaxProbes = 8;     // trial step lengths per launch
struct LbfgsState {
    double *x, *g, *d;        // K*N each: iterate, its gradient, search direction
    double *S, *Y, *rho;      // m x K*N history (circular), m curvature reciprocals
    double *xt;               // B x K*N trial points x + alpha_j d   (the batched evaluation's input)
    double *fgt;              // B x (K*N + 1) trial results { g_j, F_j } (the batched evaluation's output)
    double *alphas;           // B trial step lengths
    double *sc;               // 8 scalars: F, |g|_inf, g'd, gamma, accepted alpha, status (1: no acceptable probe), n_hist, head
    double *host_sc;          // mapped host mirror of sc (8 doubles) + [8] phi, [9] phi' of the last probe
    // the multi-workgroup step (lbfgs_dots_kernel + lbfgs_step_mb_kernel; nullptr: not in use)
    double *sc_out;           // the step writes the scalars of the NEXT iterate here (the host swaps sc and sc_out behind it)
    double *gram;             // S'Y then Y'Y, m x m each, by physical row: grown by the committed pair's row and column
    double *dots;             // kLbfgsDotBlocks x kLbfgsDotStride partial sums of the last trial point's dot products
    double c1, c2;
    int32_t KN, m;
};
hipError_t launch_lbfgs_init(const LbfgsState &st, hipStream_t stream, DoneSignal done);
hipError_t launch_lbfgs_direction(const LbfgsState &st, int B, double alpha0, hipStream_t stream);
hipError_t launch_lbfgs_select(const LbfgsState &st, int B, hipStream_t stream, DoneSignal done, int mode = 0);
hipError_t launch_lbfgs_trial(const LbfgsState &st, double alpha, hipStream_t stream);
// the multi-workgroup step: every dot product of {g_t, g, d} with {s_j, y_j, d, g} (+ g_t.g_t, |g_t|_inf) behind an
// evaluation, by kLbfgsDotBlocks workgroups (nobody waits for it) ...
constexpr int kLbfgsDotBlocks = 16, kLbfgsDotStride = 128, kLbfgsMbM = 10;
hipError_t launch_lbfgs_dots(const LbfgsState &st, hipStream_t stream);
// ... and the commit + next direction + next trial point from those numbers and the Gram matrices: scalars recomputed by
// every workgroup, vectors element-wise -- no reduction, no hand-off between workgroups
hipError_t launch_lbfgs_step_mb(const LbfgsState &st, double alpha, hipStream_t stream, DoneSignal done = DoneSignal());
// one wave: commit trial slot 0 (commit != 0), then the next direction, phi'(0) and the trial point x + d
hipError_t launch_lbfgs_step(const LbfgsState &st, int commit, hipStream_t stream, DoneSignal done = DoneSignal());

// dst[i] = src[i], i < n: moves the all-reduced [G, F] into mapped pinned host memory (one small launch
// instead of a D2H copy node: the host polls the stream)
hipError_t launch_copy(const double *src, double *dst, int n, hipStream_t stream, DoneSignal done = DoneSignal());

}  // namespace grape
