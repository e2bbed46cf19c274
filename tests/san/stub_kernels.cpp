// Host-only stand-ins for the kernel launchers of libgrape_hip.so, for the CPU sanitizer build of the C-ABI
// host layer (tests/test_sanitizers.py).  No device code exists in that build: every launcher reports
// hipErrorNoDevice; the sizing helpers answer like the real ones so that grape_create's validation and
// planning logic runs unchanged.  TEST INFRASTRUCTURE ONLY.
#include "../../quoptimalcontrol.jl_amd/csrc/grape_kernels.hpp"

namespace grape {
hipError_t launch_sweep_small(int, int, int, const SweepParams &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_sweep_pair(int, int, int, const SweepParams &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_sweep_tile(int, int, bool, const TileParams &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_sweep_grid(int, int, bool, const TileParams &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_grid_prop(int, const TileParams &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_sweep_any(const AnyParams &, hipStream_t) { return hipErrorNoDevice; }
int any_prop_blocks(int n, int N, long units, int cus) { return (n < 17 || units >= 2L * cus) ? 1 : (int)std::max(1L, std::min<long>((2L * cus + units - 1) / units, std::max(1, N / 4))); }
int reduce_rows_mflags(int Q, int n_x) { const long long g = (long long)((Q + 31) / 32) * n_x; return g >= 1 && g <= kMaxMflags ? (int)g : 0; }
hipError_t launch_reduce(const double *, const double *, double *, double *, int, int, int, hipStream_t, DoneSignal) { return hipErrorNoDevice; }
hipError_t launch_reduce_rows(const double *, double *, int, int, int, hipStream_t, DoneSignal) { return hipErrorNoDevice; }
hipError_t launch_copy(const double *, double *, int, hipStream_t, DoneSignal) { return hipErrorNoDevice; }
hipError_t launch_reduce_shards(const ShardRows &, double *, int, hipStream_t, DoneSignal) { return hipErrorNoDevice; }
hipError_t launch_shard_arrive(const ArriveParams &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_ipc_allreduce(const IpcParams &, hipStream_t) { return hipErrorNoDevice; }
size_t ipc_mailbox_bytes(int Q, int n_ranks) { const size_t Qpad = ((size_t)Q + 255) / 256 * 256; return 8 * 2 * (size_t)n_ranks * Qpad + 8 * 2 * (Qpad / 256); }
int sweep_small_max_waves(int n) { return n == 2 ? 16 : (n == 3 ? 8 : (n == 4 ? 4 : 0)); }
int sweep_pair_max_waves(int n) { return n == 2 ? 16 : (n == 4 ? 8 : 0); }
size_t sweep_small_lds_bytes(int n, int MPB, int LT, int S, int K, bool x) { return 16 * (size_t)n * n * 32 + (x ? 8 * ((size_t)MPB * LT * ((size_t)S * K + 1) + MPB) : 0); }
size_t sweep_pair_lds_bytes(int n, int MPB, int LT, int S, int K, bool x, bool) { return 16 * (size_t)n * n * 32 + 16 * (size_t)MPB * 2 * (2 * K + 3) * n * n / 2 + (x ? 8 * ((size_t)MPB * (LT / 2) * ((size_t)S * K + 1) + MPB) : 0); }
int tile_count(int n) { return n <= 4 ? 0 : (n <= 16 ? 1 : (n <= 32 ? 2 : (n <= 48 ? 3 : (n <= 64 ? 4 : 0)))); }
bool tile_chain_is_split(const TileParams &, bool) { return false; }
int tile_fuse_forward(const TileParams &) { return 0; }
hipError_t launch_lbfgs_init(const LbfgsState &, hipStream_t, DoneSignal) { return hipErrorNoDevice; }
hipError_t launch_lbfgs_direction(const LbfgsState &, int, double, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_lbfgs_select(const LbfgsState &, int, hipStream_t, DoneSignal, int) { return hipErrorNoDevice; }
hipError_t launch_lbfgs_trial(const LbfgsState &, double, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_lbfgs_dots(const LbfgsState &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_lbfgs_step_mb(const LbfgsState &, double, hipStream_t, DoneSignal) { return hipErrorNoDevice; }
hipError_t launch_lbfgs_step(const LbfgsState &, int, hipStream_t, DoneSignal) { return hipErrorNoDevice; }
hipError_t launch_exact_grad(int, int, const ExactParams &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_exact_tile(int, int, const TileParams &, int, hipStream_t) { return hipErrorNoDevice; }
int reduce_ksplit(int E) { int ks = (E + 31) / 32; return ks > 32 ? 32 : (ks < 1 ? 1 : ks); }
}  // namespace grape
