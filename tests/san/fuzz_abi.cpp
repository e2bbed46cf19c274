// Deterministic fuzz of the C ABI's argument handling under AddressSanitizer + UBSan (CPU build, no GPU):
// random grape_config structs (valid-looking, borderline and garbage fields) through grape_create, and the
// other entry points with null / dangling-free arguments.  Every call must return a grape_status (never
// crash, never touch freed or out-of-bounds memory) and leave a non-empty message on failure.
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "grape_hip.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static int32_t pick(const int32_t *v, int n) { return v[rnd() % n]; }

int main(int argc, char **argv)
{
    const long iters = argc > 1 ? atol(argv[1]) : 20000;
    static const int32_t dims[] = {-1, 0, 1, 2, 3, 4, 5, 8, 16, 17, 32, 33, 64, 1 << 20, INT32_MAX, INT32_MIN};
    static const int32_t small[] = {-3, -1, 0, 1, 2, 3, 4, 7, 8, 9, 100, 1000, 100000, INT32_MAX, INT32_MIN};
    long counts[16] = {0};
    for (long it = 0; it < iters; ++it) {
        grape_config c;
        if (rnd() % 8 == 0) {                            // pure garbage bytes
            uint64_t *w = reinterpret_cast<uint64_t *>(&c);
            for (size_t i = 0; i < sizeof(c) / 8; ++i) w[i] = rnd();
        } else {
            std::memset(&c, 0, sizeof(c));
            c.sys_type = (int32_t)(rnd() % 5) - 1;
            c.variant = (int32_t)(rnd() % 4) - 1;
            c.n = pick(dims, 16);
            c.n_controls = pick(small, 15);
            c.n_slices = pick(small, 15);
            c.n_ensemble = pick(small, 15);
            c.duration = (rnd() % 7 == 0) ? __builtin_nan("") : (double)(int64_t)(rnd() % 2000) / 100.0 - 1.0;
            c.device = (int32_t)(rnd() % 6) - 2;
            c.flags = (int32_t)(rnd() % 256);
            c.slices_per_lane = pick(small, 15);
            c.waves_per_member = pick(small, 15);
            c.expm_squarings = (int32_t)(rnd() % 70) - 5;
            c.max_batch = pick(small, 15);
            c.n_state_cols = pick(dims, 16);
            c.n_devices = (int32_t)(rnd() % 12) - 2;
            for (int i = 0; i < GRAPE_MAX_DEVICES; ++i) c.device_ids[i] = (int32_t)(rnd() % 12) - 2;
        }
        grape_ctx *ctx = reinterpret_cast<grape_ctx *>(0x1);   // must be overwritten with NULL on failure
        const int rc = grape_create(&c, &ctx);
        if (rc == GRAPE_OK) {                            // only possible with a GPU: release it again
            if (!ctx) { std::fprintf(stderr, "OK without a context\n"); return 2; }
            grape_destroy(ctx);
        } else {
            if (ctx) { std::fprintf(stderr, "failure left a context pointer\n"); return 2; }
            if (rc > 0 || rc < GRAPE_ERR_COMM) { std::fprintf(stderr, "unknown status %d\n", rc); return 2; }
            if (!grape_last_error(nullptr)[0]) { std::fprintf(stderr, "empty message for status %d\n", rc); return 2; }
        }
        counts[rc <= 0 && rc > -16 ? -rc : 15]++;
    }
    // null / misuse paths of the remaining entry points
    double F = 0, G[4] = {0}, x[4] = {0};
    int64_t n = 0;
    uint64_t u = 0;
    grape_info inf;
    grape_comm_id id;
    int bad = 0;
    bad |= grape_create(nullptr, nullptr) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_destroy(nullptr) != GRAPE_OK;
    bad |= grape_set_operators(nullptr, x, x, x, x, x) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_eval(nullptr, x, &F, G) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_eval_device(nullptr, x, G, nullptr) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_eval_batch(nullptr, 1, x, &F, G) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_eval_batch_device(nullptr, 1, x, G, nullptr) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_get_member_results(nullptr, &F, G) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_get_trajectory(nullptr, 0, x, x, x) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_get_kernel_time(nullptr, &F, &n, 0) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_get_phase_stamps(nullptr, &u, 1, &n) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_get_info(nullptr, &inf) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_comm_attach(nullptr, &id, 0, 1) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_comm_unique_id(nullptr) != GRAPE_ERR_INVALID_ARG;
    grape_ipc_handle ih;
    char names[8];
    bad |= grape_ipc_export(nullptr, 2, &ih) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_ipc_attach(nullptr, &ih, 0, 1) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_get_kernel_names(nullptr, names, 8) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_lbfgs(nullptr, x, nullptr, G, nullptr) != GRAPE_ERR_INVALID_ARG;
    bad |= grape_abi_version() != GRAPE_ABI_VERSION;
    if (bad) { std::fprintf(stderr, "a null-argument call returned the wrong status\n"); return 3; }
    std::printf("fuzz ok: %ld configs; status histogram:", iters);
    for (int i = 0; i < 10; ++i) std::printf(" [%d]=%ld", -i, counts[i]);
    std::printf("\n");
    return 0;
}
