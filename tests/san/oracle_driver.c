/* Runs the C oracle on seeded random problems of every supported shape under ASan + UBSan
 * (tests/test_sanitizers.py): exercises all three system types, both variants, every Pade branch
 * (norms 0.01 .. 40) and the ensemble entry point with OpenMP.  Exit 0 = no sanitizer report. */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int oracle_expm(int n, const void *A, void *out);
int oracle_member_eval(int st, int variant, int n, int K, int N, double T, const void *A, const void *B,
                       const void *Xi, const void *Xt, const void *x, double *fom, void *grad, void *props,
                       void *states, void *costates);
int oracle_ensemble_eval(int st, int variant, int n, int K, int N, int E, double T, const void *A, const void *B,
                         const void *Xi, const void *Xt, const void *wts, const void *x, double *F, void *G,
                         void *foms, void *grads, int n_threads);

static unsigned long long s = 88172645463325252ull;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; }

int main(void)
{
    const int dims[] = {2, 3, 4, 7, 16};
    for (int di = 0; di < 5; ++di) {
        const int n = dims[di], K = 1 + di % 3, N = 5 + di, E = 3;
        const size_t nn = (size_t)n * n;
        double _Complex *A = malloc(sizeof(*A) * E * nn), *B = malloc(sizeof(*B) * E * K * nn);
        double _Complex *Xi = malloc(sizeof(*Xi) * E * nn), *Xt = malloc(sizeof(*Xt) * E * nn);
        double _Complex *P = malloc(sizeof(*P) * N * nn), *X = malloc(sizeof(*X) * (N + 1) * nn), *L = malloc(sizeof(*L) * (N + 1) * nn);
        double *x = malloc(sizeof(double) * K * N), *g = malloc(sizeof(double) * K * N * (E + 1)), *foms = malloc(sizeof(double) * E);
        double wts[3] = {0.2, 0.3, 0.5};
        for (double scale = 0.01; scale < 50.0; scale *= 8.0) {
            for (size_t i = 0; i < E * nn; ++i) { A[i] = scale * (rnd() + I * rnd()); Xi[i] = rnd() + I * rnd(); Xt[i] = rnd() + I * rnd(); }
            for (size_t i = 0; i < (size_t)E * K * nn; ++i) B[i] = rnd() + I * rnd();
            for (int i = 0; i < K * N; ++i) x[i] = rnd();
            if (oracle_expm(n, A, P)) return 2;
            for (int st = 0; st < 3; ++st)
                for (int variant = 0; variant < 2; ++variant) {
                    double F;
                    if (oracle_member_eval(st, variant, n, K, N, 1.0, A, B, Xi, Xt, x, &F, g, P, X, L)) return 3;
                    if (oracle_member_eval(st, variant, n, K, N, 1.0, A, B, Xi, Xt, x, &F, g, NULL, NULL, NULL)) return 3;
                    if (oracle_ensemble_eval(st, variant, n, K, N, E, 1.0, A, B, Xi, Xt, wts, x, &F, g, foms, g + K * N, 2)) return 4;
                    if (oracle_ensemble_eval(st, variant, n, K, N, E, 1.0, A, B, Xi, Xt, wts, x, &F, g, NULL, NULL, 1)) return 4;
                }
        }
        free(A); free(B); free(Xi); free(Xt); free(P); free(X); free(L); free(x); free(g); free(foms);
    }
    puts("oracle sanitizer run ok");
    return 0;
}
