// host_latency.cpp -- what a compiled caller (the Julia ccall) sees: grape_eval in a tight loop, no Python.
// Builds a C3-shaped ensemble (4x4 UnitaryGate, K=4, N=500, E=1024 by default) with synthetic Hermitian
// generators, calls grape_eval `iters` times and prints the mean/median latency, with and without
// GRAPE_FLAG_TIME_KERNELS, plus the sweep kernel's own HIP-event time.
//   hipcc -O2 -o build/host_latency tools/cbench/host_latency.cpp -Iinclude -Lquoptimalcontrol.jl_amd -lgrape_hip
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "grape_hip.h"

typedef std::complex<double> cplx;

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int n = 4, K = 4, N = argc > 2 ? atoi(argv[2]) : 500, E = argc > 1 ? atoi(argv[1]) : 1024;
    const int iters = argc > 3 ? atoi(argv[3]) : 2000;
    std::vector<cplx> A((size_t)E * n * n), B((size_t)E * K * n * n), Xi((size_t)E * n * n), Xt((size_t)E * n * n);
    std::vector<double> wts(E, 1.0 / E), x((size_t)K * N), G((size_t)K * N);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5; };
    auto herm = [&](cplx *M, double scale) {
        for (int j = 0; j < n; ++j)
            for (int i = 0; i <= j; ++i) {
                cplx v(rnd() * scale, i == j ? 0.0 : rnd() * scale);
                M[i + j * n] = v;
                M[j + i * n] = std::conj(v);
            }
    };
    for (int k = 0; k < E; ++k) {
        herm(&A[(size_t)k * n * n], 3.0);
        for (int c = 0; c < K; ++c) herm(&B[((size_t)k * K + c) * n * n], 1.0);
        for (int i = 0; i < n; ++i) { Xi[(size_t)k * n * n + i + i * n] = 1.0; Xt[(size_t)k * n * n + i + ((i + 1) % n) * n] = 1.0; }
    }
    for (auto &v : x) v = rnd() + 0.5;
    for (int timed = 0; timed < 2; ++timed) {
        grape_config cfg{};
        cfg.sys_type = GRAPE_UNITARY_GATE; cfg.n = n; cfg.n_controls = K; cfg.n_slices = N; cfg.n_ensemble = E;
        cfg.duration = 2.0; cfg.device = -1; cfg.expm_squarings = -1; cfg.flags = timed ? GRAPE_FLAG_TIME_KERNELS : 0;
        grape_ctx *ctx = nullptr;
        if (grape_create(&cfg, &ctx)) { fprintf(stderr, "create: %s\n", grape_last_error(nullptr)); return 1; }
        if (grape_set_operators(ctx, (double *)A.data(), (double *)B.data(), (double *)Xi.data(), (double *)Xt.data(), wts.data())) {
            fprintf(stderr, "set: %s\n", grape_last_error(ctx)); return 1; }
        double F = 0;
        for (int i = 0; i < 50; ++i) grape_eval(ctx, x.data(), &F, G.data());
        std::vector<double> lat(iters);
        const double t0 = now_us();
        for (int i = 0; i < iters; ++i) {
            const double a = now_us();
            if (grape_eval(ctx, x.data(), &F, G.data())) { fprintf(stderr, "eval: %s\n", grape_last_error(ctx)); return 1; }
            lat[i] = now_us() - a;
        }
        const double total = now_us() - t0;
        std::sort(lat.begin(), lat.end());
        double kms = 0; int64_t kn = 0;
        grape_get_kernel_time(ctx, &kms, &kn, 1);
        printf("E=%d N=%d timed_flag=%d: grape_eval mean %.2f us  median %.2f  p10 %.2f  p90 %.2f  (%.0f evals/s)  sweep kernel %.2f us  F=%.6f\n",
               E, N, timed, total / iters, lat[iters / 2], lat[iters / 10], lat[iters * 9 / 10], 1e6 * iters / total,
               kn ? 1e3 * kms / kn : 0.0, F);
        grape_destroy(ctx);
    }
    return 0;
}
