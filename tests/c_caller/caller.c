/* A plain-C caller of libgrape_hip.so, doing what the Julia glue does through ccall: pack the
 * ensemble's operators, grape_create / grape_set_operators / grape_eval, read F and G.  Test infrastructure
 * (tests/test_gpu_c_caller.py compiles it with gcc against include/grape_hip.h and compares the printed
 * numbers with the oracle on the same inputs).
 *
 * Problem: the reference's single-qubit ensemble (test/state_transfer_tests.jl:40-67 / README.md:10-30):
 * StateTransfer, 2x2, A_k = (1 + 0.1 (k - 2)) Sz, B = [Sx, Sy], rho_0 = |0><0| -> |1><1|, n_ens members,
 * weights 1/n_ens; the control array comes from a small linear congruential generator printed in the output. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grape_hip.h"

int main(int argc, char **argv)
{
    const int n = 2, K = 2, N = argc > 1 ? atoi(argv[1]) : 10, E = argc > 2 ? atoi(argv[2]) : 3;
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    const double T = 1.0;
    double *A = calloc((size_t)2 * n * n * E, sizeof(double));
    double *B = calloc((size_t)2 * n * n * K * E, sizeof(double));
    double *Xi = calloc((size_t)2 * n * n * E, sizeof(double));
    double *Xt = calloc((size_t)2 * n * n * E, sizeof(double));
    double *w = calloc((size_t)E, sizeof(double));
    double *x = calloc((size_t)K * N, sizeof(double)), *G = calloc((size_t)K * N, sizeof(double)), F = 0.0;
    /* complex (re, im) interleaved, column-major: element (i, j) of matrix m at 2 * (m * n * n + i + j * n) */
    for (int k = 0; k < E; ++k) {
        const double s = 1.0 + 0.1 * (k - E / 2);
        double *a = A + (size_t)2 * n * n * k;
        a[2 * (0 + 0 * n)] = 0.5 * s;                         /* Sz = diag(1/2, -1/2) */
        a[2 * (1 + 1 * n)] = -0.5 * s;
        double *bx = B + (size_t)2 * n * n * (0 + K * k), *by = B + (size_t)2 * n * n * (1 + K * k);
        bx[2 * (1 + 0 * n)] = 0.5;                            /* Sx */
        bx[2 * (0 + 1 * n)] = 0.5;
        by[2 * (1 + 0 * n) + 1] = 0.5;                        /* Sy = [[0, -i/2], [i/2, 0]] */
        by[2 * (0 + 1 * n) + 1] = -0.5;
        Xi[(size_t)2 * n * n * k + 2 * (0 + 0 * n)] = 1.0;    /* |0><0| */
        Xt[(size_t)2 * n * n * k + 2 * (1 + 1 * n)] = 1.0;    /* |1><1| */
        w[k] = 1.0 / E;
    }
    unsigned long long state = 12345;
    for (int i = 0; i < K * N; ++i) {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        x[i] = (double)(state >> 11) / 9007199254740992.0;   /* [0, 1) */
    }

    grape_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.sys_type = GRAPE_STATE_TRANSFER;
    cfg.variant = variant;
    cfg.n = n;
    cfg.n_controls = K;
    cfg.n_slices = N;
    cfg.n_ensemble = E;
    cfg.duration = T;
    cfg.device = -1;
    cfg.expm_squarings = -1;
    grape_ctx *ctx = NULL;
    int rc = grape_create(&cfg, &ctx);
    if (rc) { fprintf(stderr, "grape_create: %d %s\n", rc, grape_last_error(NULL)); return 1; }
    rc = grape_set_operators(ctx, A, B, Xi, Xt, w);
    if (rc) { fprintf(stderr, "grape_set_operators: %d %s\n", rc, grape_last_error(ctx)); return 1; }
    rc = grape_eval(ctx, x, &F, G);
    if (rc) { fprintf(stderr, "grape_eval: %d %s\n", rc, grape_last_error(ctx)); return 1; }
    double F_only = 0.0;
    rc = grape_eval(ctx, x, &F_only, NULL);                   /* Optim asks for F alone, src/solve.jl:189-195 */
    if (rc || F_only != F) { fprintf(stderr, "F-only evaluation differs\n"); return 1; }
    grape_info info;
    rc = grape_get_info(ctx, &info);
    if (rc) { fprintf(stderr, "grape_get_info: %d\n", rc); return 1; }
    printf("abi %d arch %s family %d\n", info.abi_version, info.arch, info.kernel_family);
    printf("F %.17g\n", F);
    for (int i = 0; i < K * N; ++i)
        printf("x %.17g G %.17g\n", x[i], G[i]);
    grape_destroy(ctx);
    free(A); free(B); free(Xi); free(Xt); free(w); free(x); free(G);
    return 0;
}
