// grape_api.cpp -- C-ABI host layer of libgrape_hip.so (include/grape_hip.h).
//
// Owns the device workspace (what init_GRAPE allocates on the Julia heap,
// /root/reference/src/grape_tools.jl:4-16), uploads the operators once, and turns one call of
// the reference's (F, G, x) closure (src/solve.jl:164-196) into: sweep kernel -> 2 reduce
// launches, all asynchronous on one HIP stream.  There is NO CPU fallback: without a gfx950
// device every entry point fails with GRAPE_ERR_NO_DEVICE.
#include "../../include/grape_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <string>
#include <vector>

#include "grape_kernels.hpp"

using grape::SweepParams;
using grape::TileParams;
typedef std::complex<double> cplx;

struct grape_ctx {
    grape_config cfg{};
    int device = 0;
    int compute_units = 0;
    char arch[32] = {0};
    int family = 0;               // 0: register-resident small-n kernels, 1: MFMA tile kernels
    int NT = 0;                   // tile family: tiles per dimension (padded n = 16 NT)
    size_t TSZ = 0;               // tile family: double2 per matrix dump
    bool pack2 = false;           // tile family, n <= 8: two members per 16x16 tile (block diagonal)
    int EU = 0;                   // tile family: wavefront-level units = members, or member pairs when pack2
    int S = 0, W = 0, LT = 0;
    int MPB = 1, NB = 0;          // small family: members per workgroup, workgroups per control array
    int B = 1;                    // batch capacity: control arrays per grape_eval_batch call
    double *d_block_out = nullptr;
    double *d_xg_scratch = nullptr;   // only when K*N is too long for the LDS staging buffer
    int ksplit = 1;
    size_t ws_elems = 0;          // double2 elements per workspace array
    uint64_t bytes = 0;
    // device
    double2 *d_ops = nullptr;
    double *d_wts = nullptr;
    double *d_x = nullptr;
    double *d_fg = nullptr;
    double2 *d_props = nullptr, *d_states = nullptr, *d_costates = nullptr;
    double *d_member_out = nullptr;
    double *d_partial = nullptr;
    unsigned long long *d_stamps = nullptr;
    // host
    double *h_stage = nullptr;    // pinned, K*N + 1 doubles: x on the way in
    double *h_fg = nullptr;       // pinned + device-mapped, K*N + 1 doubles: the reduce kernel writes [G, F] here
    double *d_h_fg = nullptr;     // device address of h_fg
    hipStream_t stream = nullptr;
    bool ops_set = false, evaluated = false;
    bool unitary = false;         // all generators Hermitian -> unitary data flow
    std::vector<hipEvent_t> ev;   // start/stop pairs of the sweep kernel
    size_t ev_used = 0;
    double ev_total_ms = 0.0;
    int64_t ev_count = 0;
    mutable std::string err;
};

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

static size_t KN(const grape_ctx *c) { return (size_t)c->cfg.n_controls * c->cfg.n_slices; }

static void free_all(grape_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->d_ops); (void)hipFree(c->d_wts); (void)hipFree(c->d_x); (void)hipFree(c->d_fg);
    (void)hipFree(c->d_props); (void)hipFree(c->d_states); (void)hipFree(c->d_costates);
    (void)hipFree(c->d_member_out); (void)hipFree(c->d_partial); (void)hipFree(c->d_stamps); (void)hipFree(c->d_block_out); (void)hipFree(c->d_xg_scratch);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_fg) (void)hipHostFree(c->h_fg);
    delete c;
}

extern "C" int grape_abi_version(void) { return GRAPE_ABI_VERSION; }

extern "C" const char *grape_last_error(const grape_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

extern "C" int grape_create(const grape_config *cfg, grape_ctx **out)
{
    if (out) *out = nullptr;
    if (!cfg || !out) return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: null argument");
    if (cfg->sys_type < GRAPE_UNITARY_GATE || cfg->sys_type > GRAPE_COHERENCE_TRANSFER)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: bad sys_type");
    if (cfg->variant != GRAPE_VARIANT_INPLACE && cfg->variant != GRAPE_VARIANT_STATIC)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: bad variant");
    if (cfg->n < 1 || cfg->n_controls < 1 || cfg->n_slices < 1 || cfg->n_ensemble < 1)
        return fail(nullptr, GRAPE_ERR_INVALID_ARG,
                    "grape_create: n, n_controls, n_slices, n_ensemble must be positive");
    if (!(cfg->duration == cfg->duration))
        return fail(nullptr, GRAPE_ERR_INVALID_ARG, "grape_create: duration is NaN");
    const int wmax = grape::sweep_small_max_waves(cfg->n);
    const int nt = grape::tile_count(cfg->n);
    if (wmax == 0 && nt == 0)
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED,
                    "grape_create: operator dimension n=" + std::to_string(cfg->n) +
                        " has no kernel in this build (supported: 2..32)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, GRAPE_ERR_NO_DEVICE, "grape_create: no HIP device visible");
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev)
        return fail(nullptr, GRAPE_ERR_NO_DEVICE, "grape_create: device ordinal out of range");
    HIP_TRY(nullptr, hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !std::getenv("GRAPE_HIP_ANY_ARCH"))
        return fail(nullptr, GRAPE_ERR_NO_DEVICE,
                    std::string("grape_create: device is ") + prop.gcnArchName +
                        ", this library carries gfx950 code only");

    grape_ctx *c = new (std::nothrow) grape_ctx();
    if (!c) return fail(nullptr, GRAPE_ERR_ALLOC, "grape_create: out of host memory");
    c->cfg = *cfg;
    c->device = dev;
    c->compute_units = prop.multiProcessorCount;
    std::snprintf(c->arch, sizeof(c->arch), "%s", prop.gcnArchName);

    // time-axis decomposition: W waves per member, S slices per lane, S * 64 * W >= N.
    // Aim for one wave per SIMD across the chip; more waves per member only when the
    // ensemble alone cannot fill it.
    const int N = cfg->n_slices, E = cfg->n_ensemble;
    c->family = wmax > 0 ? 0 : 1;
    c->B = cfg->max_batch > 1 ? cfg->max_batch : 1;
    if (c->B > 1 && c->family != 0) {
        delete c;
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED, "grape_create: max_batch > 1 needs operator dimension n <= 4 in this build");
    }
    c->NT = nt;
    c->TSZ = (size_t)nt * nt * 256;
    c->pack2 = (c->family == 1 && cfg->n <= 8 && !std::getenv("GRAPE_TILE_NOPACK") && !std::getenv("GRAPE_TILE_MFMA4"));
    c->EU = c->pack2 ? (E + 1) / 2 : E;
    int W = cfg->waves_per_member;
    if (W <= 0) {
        const long simds = 4L * c->compute_units;
        const long units = (long)E * c->B;                       // a batch fills the chip like a larger ensemble
        W = (int)((simds + units - 1) / units);
        const int wneed = (N + 63) / 64;
        if (W > wneed) W = wneed;
    }
    if (wmax > 0 && W > wmax) W = wmax;
    if (W < 1 || c->family == 1) W = 1;
    int S = cfg->slices_per_lane;
    const int smin = (N + 64 * W - 1) / (64 * W);
    if (S < smin) S = smin;
    c->W = W; c->S = S; c->LT = 64 * W;
    if (c->family == 0 && (uint64_t)cfg->n_controls * N * S * cfg->n_controls >= (1ull << 32)) {
        delete c;
        return fail(nullptr, GRAPE_ERR_UNSUPPORTED, "grape_create: n_controls^2 * n_slices^2 too large for the LDS index arithmetic");
    }
    c->MPB = (c->family == 0 && W <= 4) ? 4 / W : 1;         // fill the 4 SIMDs of a CU per workgroup
    if (const char *ev = std::getenv("GRAPE_MPB")) {           // tuning experiment: members per workgroup
        const int v = std::atoi(ev);
        if (c->family == 0 && v >= 1 && v * W <= wmax) c->MPB = v;
    }
    if (c->MPB > E) c->MPB = E;
    bool xg_in_lds = true;
    if (c->family == 0) {                                    // fit the x/g staging buffer into LDS
        const size_t cap = 150 * 1024;
        while (c->MPB > 1 && grape::sweep_small_lds_bytes(cfg->n, c->MPB, c->LT, S, cfg->n_controls, true) > cap)
            c->MPB /= 2;
        xg_in_lds = grape::sweep_small_lds_bytes(cfg->n, c->MPB, c->LT, S, cfg->n_controls, true) <= cap;
    }
    c->NB = (E + c->MPB - 1) / c->MPB;
    c->ksplit = grape::reduce_ksplit(E);

    const size_t nn = (size_t)cfg->n * cfg->n, K = cfg->n_controls;
    const size_t Q = KN(c) + 1;
    c->ws_elems = c->family == 0 ? (size_t)E * S * nn * c->LT : (size_t)c->EU * N * c->TSZ;
    const size_t ops_elems = c->family == 0 ? (size_t)E * (K + 3) * nn : (size_t)c->EU * (2 * K + 3) * c->TSZ;
    const bool keepl = (cfg->flags & GRAPE_FLAG_KEEP_COSTATES) != 0;

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
    if (e == hipSuccess) e = alloc((void **)&c->d_props, sizeof(double2) * c->ws_elems * Bn);
    if (e == hipSuccess && keepl) e = alloc((void **)&c->d_costates, sizeof(double2) * c->ws_elems * Bn);
    const bool want_rows = c->family == 1 || (cfg->flags & GRAPE_FLAG_MEMBER_RESULTS);
    if (e == hipSuccess && want_rows) e = alloc((void **)&c->d_member_out, sizeof(double) * E * Q * Bn);
    if (e == hipSuccess) e = alloc((void **)&c->d_partial, sizeof(double) * c->ksplit * Q);
    if (e == hipSuccess && c->family == 0) e = alloc((void **)&c->d_block_out, sizeof(double) * c->NB * Q * Bn);
    if (e == hipSuccess && c->family == 0 && !xg_in_lds)
        e = alloc((void **)&c->d_xg_scratch,
                  sizeof(double) * Bn * c->NB * ((size_t)c->MPB * c->LT * ((size_t)S * K + 1) + c->MPB));
    if (e == hipSuccess && (cfg->flags & GRAPE_FLAG_PHASE_STAMPS))
        e = alloc((void **)&c->d_stamps, sizeof(unsigned long long) * Bn * E * W * grape::kStampSlots);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_stage, sizeof(double) * Q * Bn, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_fg, sizeof(double) * Q * Bn, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->d_h_fg, c->h_fg, 0);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        std::string msg = std::string("grape_create: device allocation failed: ") + hipGetErrorString(e);
        free_all(c);
        return fail(nullptr, e == hipErrorOutOfMemory ? GRAPE_ERR_ALLOC : GRAPE_ERR_HIP, msg);
    }
    *out = c;
    return GRAPE_OK;
}

extern "C" int grape_destroy(grape_ctx *ctx)
{
    if (!ctx) return GRAPE_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_all(ctx);
    return GRAPE_OK;
}

extern "C" int grape_set_operators(grape_ctx *c, const double *A, const double *B, const double *Xi,
                                   const double *Xt, const double *wts)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!A || !B || !Xi || !Xt || !wts)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_set_operators: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nn = (size_t)c->cfg.n * c->cfg.n, K = c->cfg.n_controls, E = c->cfg.n_ensemble;
    std::vector<double> packed;
    try {
        packed.assign(c->family == 0 ? 2 * E * (K + 3) * nn : 2 * (size_t)c->EU * (2 * K + 3) * c->TSZ, 0.0);
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
                !(c->family == 1 && std::getenv("GRAPE_TILE_MFMA4"));     // the 4x4x4 experiment has no unitary flow
    const int n = c->cfg.n;
    for (size_t k = 0; k < E && herm; ++k) {
        for (size_t m = 0; m < K + 1 && herm; ++m) {
            const double *M = (m == 0) ? A + 2 * k * nn : B + 2 * (k * K + (m - 1)) * nn;
            double scale = 0.0, dev = 0.0;
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) {
                    const double re = M[2 * (i + j * n)], im = M[2 * (i + j * n) + 1];
                    const double tr = M[2 * (j + i * n)], ti = M[2 * (j + i * n) + 1];
                    scale = std::fmax(scale, std::fmax(std::fabs(re), std::fabs(im)));
                    dev = std::fmax(dev, std::fmax(std::fabs(re - tr), std::fabs(im + ti)));
                }
            if (!(dev <= 4e-16 * scale)) herm = false;      // also false on NaN
        }
    }
    c->unitary = herm;
    if (!herm && !c->d_states) {
        c->bytes += sizeof(double2) * c->ws_elems * c->B;
        HIP_TRY(c, hipMalloc((void **)&c->d_states, sizeof(double2) * c->ws_elems * c->B));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_ops, packed.data(), sizeof(double) * packed.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_wts, wts, sizeof(double) * E, hipMemcpyHostToDevice));
    c->ops_set = true;
    c->evaluated = false;
    return GRAPE_OK;
}

static int enqueue_tile(grape_ctx *c, const double *d_x, hipStream_t stream)
{
    TileParams p{};
    p.ops = c->d_ops;
    p.x = d_x;
    p.props = c->d_props;
    p.states = c->d_states;
    p.costates = c->d_costates;
    p.member_out = c->d_member_out;
    p.K = c->cfg.n_controls;
    p.N = c->cfg.n_slices;
    p.E = c->EU;
    p.E_members = c->cfg.n_ensemble;
    p.pack2 = c->pack2 ? 1 : 0;
    p.n = c->cfg.n;
    p.s_forced = c->cfg.expm_squarings;
    p.variant = c->cfg.variant;
    p.dt = c->cfg.duration / c->cfg.n_slices;
    p.unitary = c->unitary ? 1 : 0;
    // default: the v_mfma_f64_16x16x4 kernels (sweep_tile.hip).  GRAPE_TILE_MFMA4=1 selects the
    // v_mfma_f64_4x4x4_4b variant (sweep_tile4.hip): correct, but measured 15-50 % slower so far.
    static const bool mfma4 = std::getenv("GRAPE_TILE_MFMA4") != nullptr;
    if (!mfma4)
        HIP_TRY(c, grape::launch_sweep_tile(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE,
                                            c->d_costates != nullptr, p, stream));
    else
        HIP_TRY(c, grape::launch_sweep_tile4(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE,
                                             c->d_costates != nullptr, p, stream));
    return GRAPE_OK;
}

static int enqueue_eval(grape_ctx *c, const double *d_x, double *d_fg, hipStream_t stream, int n_x = 1)
{
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
    const bool timed = (c->cfg.flags & GRAPE_FLAG_TIME_KERNELS) != 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        if (c->ev_used + 2 > c->ev.size()) {
            for (int i = 0; i < 2; ++i) {
                hipEvent_t e;
                HIP_TRY(c, hipEventCreate(&e));
                c->ev.push_back(e);
            }
        }
        e0 = c->ev[c->ev_used];
        e1 = c->ev[c->ev_used + 1];
        c->ev_used += 2;
        HIP_TRY(c, hipEventRecord(e0, stream));
    }
    if (c->family == 0) {
        const int mode = c->unitary ? 2 : (c->d_costates ? 1 : 0);
        HIP_TRY(c, grape::launch_sweep_small(c->cfg.n, c->cfg.sys_type != GRAPE_UNITARY_GATE, mode, p, stream));
    } else {
        int rc = enqueue_tile(c, d_x, stream);
        if (rc) return rc;
    }
    if (timed) HIP_TRY(c, hipEventRecord(e1, stream));
    if (c->family == 0)
        HIP_TRY(c, grape::launch_reduce_rows(c->d_block_out, d_fg, c->NB, (int)(KN(c) + 1), n_x, stream));
    else
        HIP_TRY(c, grape::launch_reduce(c->d_member_out, c->d_wts, c->d_partial, d_fg, p.E,
                                        (int)(KN(c) + 1), c->ksplit, stream));
    c->evaluated = true;
    return GRAPE_OK;
}

extern "C" int grape_eval_device(grape_ctx *c, const double *d_x, double *d_fg, void *stream)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!d_x || !d_fg) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_device: null argument");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, "grape_eval_device: operators not set");
    HIP_TRY(c, hipSetDevice(c->device));
    return enqueue_eval(c, d_x, d_fg, (hipStream_t)stream);
}

extern "C" int grape_eval(grape_ctx *c, const double *x, double *F, double *G)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!x) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval: x is null");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, "grape_eval: operators not set");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t kn = KN(c);
    std::memcpy(c->h_stage, x, sizeof(double) * kn);
    HIP_TRY(c, hipMemcpyAsync(c->d_x, c->h_stage, sizeof(double) * kn, hipMemcpyHostToDevice, c->stream));
    // the final reduce kernel writes its 16 KB result straight into mapped pinned host memory (no
    // D2H copy node), and the host polls the stream instead of sleeping on an interrupt: the
    // optimiser is sequential, so per-call latency is what the Julia side sees
    int rc = enqueue_eval(c, c->d_x, c->d_h_fg, c->stream);
    if (rc) return rc;
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(c, q);
    }
    if (G) std::memcpy(G, c->h_fg, sizeof(double) * kn);
    if (F) *F = c->h_fg[kn];
    return GRAPE_OK;
}

extern "C" int grape_eval_batch_device(grape_ctx *c, int32_t n_x, const double *d_x, double *d_fg, void *stream)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!d_x || !d_fg) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_batch_device: null argument");
    if (n_x < 1 || n_x > c->B)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_batch_device: n_x must be in 1..grape_config.max_batch");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, "grape_eval_batch_device: operators not set");
    HIP_TRY(c, hipSetDevice(c->device));
    return enqueue_eval(c, d_x, d_fg, (hipStream_t)stream, n_x);
}

extern "C" int grape_eval_batch(grape_ctx *c, int32_t n_x, const double *x, double *F, double *G)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!x) return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_batch: x is null");
    if (n_x < 1 || n_x > c->B)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_eval_batch: n_x must be in 1..grape_config.max_batch");
    if (!c->ops_set) return fail(c, GRAPE_ERR_NOT_READY, "grape_eval_batch: operators not set");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t kn = KN(c), Q = kn + 1;
    std::memcpy(c->h_stage, x, sizeof(double) * kn * n_x);
    HIP_TRY(c, hipMemcpyAsync(c->d_x, c->h_stage, sizeof(double) * kn * n_x, hipMemcpyHostToDevice, c->stream));
    int rc = enqueue_eval(c, c->d_x, c->d_h_fg, c->stream, n_x);
    if (rc) return rc;
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(c, q);
    }
    for (int b = 0; b < n_x; ++b) {
        if (G) std::memcpy(G + (size_t)b * kn, c->h_fg + (size_t)b * Q, sizeof(double) * kn);
        if (F) F[b] = c->h_fg[(size_t)b * Q + kn];
    }
    return GRAPE_OK;
}

extern "C" int grape_get_member_results(grape_ctx *c, double *foms, double *grads)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!c->evaluated) return fail(c, GRAPE_ERR_NOT_READY, "grape_get_member_results: no evaluation yet");
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
static int fetch_slab(grape_ctx *c, const double2 *d_ws, int member, cplx *out)
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
                            const int row = 16 * I + 4 * r + (l >> 4) - off, col = 16 * J + (l & 15) - off;
                            if (row >= 0 && col >= 0 && row < n && col < n)
                                out[t * n * n + row + (size_t)n * col] =
                                    h[t * TSZ + (size_t)((I * NT + J) * 4 + r) * 64 + l];
                        }
        return GRAPE_OK;
    }
    const size_t nn = (size_t)c->cfg.n * c->cfg.n, S = c->S, LT = c->LT, N = c->cfg.n_slices;
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
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!c->evaluated) return fail(c, GRAPE_ERR_NOT_READY, "grape_get_trajectory: no evaluation yet");
    if (member < 0 || member >= c->cfg.n_ensemble)
        return fail(c, GRAPE_ERR_INVALID_ARG, "grape_get_trajectory: member out of range");
    if (costates && !c->d_costates)
        return fail(c, GRAPE_ERR_NOT_READY,
                    "grape_get_trajectory: costates need GRAPE_FLAG_KEEP_COSTATES at grape_create");
    if (states && !c->d_costates && (c->family == 0 || c->unitary))
        return fail(c, GRAPE_ERR_NOT_READY,
                    "grape_get_trajectory: forward states are stored only by the debug flow; create the "
                    "context with GRAPE_FLAG_KEEP_COSTATES");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    const int n = c->cfg.n;
    const size_t nn = (size_t)n * n, N = c->cfg.n_slices, K = c->cfg.n_controls;
    std::vector<cplx> P(N * nn);
    int rc = fetch_slab(c, c->d_props, member, P.data());
    if (rc) return rc;
    if (props) std::memcpy(props, P.data(), sizeof(cplx) * N * nn);
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
        if (c->family == 0) {
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
    return GRAPE_OK;
}

extern "C" int grape_get_kernel_time(grape_ctx *c, double *total_ms, int64_t *launches, int32_t reset)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        HIP_TRY(c, hipEventSynchronize(c->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->ev_total_ms += ms;
        c->ev_count += 1;
    }
    c->ev_used = 0;
    if (total_ms) *total_ms = c->ev_total_ms;
    if (launches) *launches = c->ev_count;
    if (reset) { c->ev_total_ms = 0.0; c->ev_count = 0; }
    return GRAPE_OK;
}

extern "C" int grape_get_phase_stamps(grape_ctx *c, uint64_t *out, int64_t capacity, int64_t *count)
{
    if (!c) return GRAPE_ERR_INVALID_ARG;
    if (!c->d_stamps || !c->evaluated)
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
    info->expm_theta = 0.05;
    info->workspace_bytes = c->bytes;
    std::snprintf(info->arch, sizeof(info->arch), "%s", c->arch);
    return GRAPE_OK;
}
