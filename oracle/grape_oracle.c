/*
 * grape_oracle.c -- CPU restatement of the QuOptimalControl.jl GRAPE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.  The product path
 * (libgrape_hip.so) never links, loads or calls this file.
 *
 * PARITY STATUS: "parity unpinned" per evaluation.  The reference is Julia and
 * Julia is absent from the build container and from the GPU box; the reference's
 * tests (test/state_transfer_tests.jl, test/unitary_gate_tests.jl) hold no golden
 * vectors, only "converged minimum <= floor + 1e-6" asserts from unseeded random
 * starts.  This oracle is pinned by (1) those convergence asserts re-run through it
 * (tests/test_oracle_reference_cases.py), (2) the known answers derivable from
 * test/setup_tests.jl (C1(rho,rho)=0.75, C1(U,U)=0), (3) a 50-digit mpmath
 * restatement (oracle/make_golden.py -> tests/golden/) and (4) an independent
 * NumPy/SciPy restatement (oracle/grape_numpy.py, different expm algorithm).
 *
 * What each function follows (paths relative to /root/reference):
 *   oracle_expm            LinearAlgebra.exp! (Julia stdlib, NOT in the reference tree;
 *                          Higham 2005 scaling-and-squaring Pade),
 *                          called from src/timeevolution.jl:108 and :53.
 *                          Deviations, rounding-level only: no gebal balancing, no
 *                          Hermitian eigen shortcut.
 *   oracle_member_eval     _fom_and_gradient_GRAPE!   src/GRAPE.jl:25-96   (variant 0)
 *                          _fom_and_gradient_sGRAPE   src/GRAPE.jl:103-166 (variant 1)
 *     propagators          pw_prop_save!  src/timeevolution.jl:98-110 (variant 0)
 *                          pw_evolve_save src/timeevolution.jl:45-57  (variant 1)
 *     sweeps               evolve_func!   src/GRAPE.jl:216-251 / evolve_func :178-209
 *     gradient             grad_func!     src/GRAPE.jl:261-287 / grad_func   :289-303
 *     figure of merit      fom_func       src/cost_functions.jl:99-111, C1 :13-17
 *     commutator           src/tools.jl:17-19
 *   oracle_ensemble_eval   closure topt in solve(::EnsembleProblem, ::GRAPE),
 *                          src/solve.jl:164-196 (F accumulated in k order, :171-186;
 *                          G = sum(gradient .* wts, dims=1), :191)
 *
 * Layout: every matrix is column-major (Julia), element (i,j) at i + j*n, complex as
 * interleaved {re, im} doubles (binary compatible with Julia ComplexF64).
 * x and G are (K, N) column-major: x[j,i] at j + i*K.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cplx;

enum { ORACLE_UG = 0, ORACLE_ST = 1, ORACLE_CT = 2 };

/* ---------------------------------------------------------------- small dense helpers */

/* C = A * B, all n x n column-major.  Plain triple loop, k innermost-summed in order. */
static void mm(int n, const cplx *A, const cplx *B, cplx *C)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            cplx s = 0.0;
            for (int k = 0; k < n; ++k)
                s += A[i + k * n] * B[k + j * n];
            C[i + j * n] = s;
        }
}

/* C = A' * B  (A' = conjugate transpose) */
static void mm_ah_b(int n, const cplx *A, const cplx *B, cplx *C)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            cplx s = 0.0;
            for (int k = 0; k < n; ++k)
                s += conj(A[k + i * n]) * B[k + j * n];
            C[i + j * n] = s;
        }
}

/* C = A * B' */
static void mm_a_bh(int n, const cplx *A, const cplx *B, cplx *C)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            cplx s = 0.0;
            for (int k = 0; k < n; ++k)
                s += A[i + k * n] * conj(B[j + k * n]);
            C[i + j * n] = s;
        }
}

static cplx trace(int n, const cplx *A)
{
    cplx s = 0.0;
    for (int i = 0; i < n; ++i)
        s += A[i + i * n];
    return s;
}

static double opnorm1(int n, const cplx *A)
{
    double best = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i)
            s += cabs(A[i + j * n]);
        if (s > best || s != s)
            best = s;
    }
    return best;
}

/* Solve M X = R in place (R <- X), n x n system with n right-hand sides, LU with
 * partial pivoting (pivot by |re|+|im| like LAPACK izamax/zgetf2 behind gesv!). */
static int gesv(int n, cplx *M, cplx *R)
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(creal(M[k + k * n])) + fabs(cimag(M[k + k * n]));
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(creal(M[i + k * n])) + fabs(cimag(M[i + k * n]));
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0)
            return -1;
        if (p != k) {
            for (int j = 0; j < n; ++j) {
                cplx t = M[k + j * n]; M[k + j * n] = M[p + j * n]; M[p + j * n] = t;
                t = R[k + j * n]; R[k + j * n] = R[p + j * n]; R[p + j * n] = t;
            }
        }
        cplx inv = 1.0 / M[k + k * n];
        for (int i = k + 1; i < n; ++i) {
            cplx f = M[i + k * n] * inv;
            M[i + k * n] = f;
            for (int j = k + 1; j < n; ++j)
                M[i + j * n] -= f * M[k + j * n];
            for (int j = 0; j < n; ++j)
                R[i + j * n] -= f * R[k + j * n];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = n - 1; i >= 0; --i) {
            cplx s = R[i + j * n];
            for (int k = i + 1; k < n; ++k)
                s -= M[i + k * n] * R[k + j * n];
            R[i + j * n] = s / M[i + i * n];
        }
    return 0;
}

/* ---------------------------------------------------------------- expm (Higham 2005) */

static const double PADE3[]  = {120., 60., 12., 1.};
static const double PADE5[]  = {30240., 15120., 3360., 420., 30., 1.};
static const double PADE7[]  = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
static const double PADE9[]  = {17643225600., 8821612800., 2075673600., 302702400., 30270240.,
                                2162160., 110880., 3960., 90., 1.};
static const double PADE13[] = {64764752532480000., 32382376266240000., 7771770303897600.,
                                1187353796428800., 129060195264000., 10559470521600.,
                                670442572800., 33522128640., 1323241920., 40840800., 960960.,
                                16380., 182., 1.};

/* out = exp(Ain).  Returns 0, or -1 if the Pade denominator is singular.
 * Branch thresholds and evaluation order as in Julia's exp! (see header). */
int oracle_expm(int n, const cplx *Ain, cplx *out)
{
    const size_t nn = (size_t)n * n;
    cplx *w = (cplx *)malloc(sizeof(cplx) * nn * 8);
    if (!w)
        return -2;
    cplx *A = w, *A2 = w + nn, *P = w + 2 * nn, *U = w + 3 * nn, *V = w + 4 * nn,
         *T = w + 5 * nn, *A4 = w + 6 * nn, *A6 = w + 7 * nn;
    memcpy(A, Ain, sizeof(cplx) * nn);
    double nA = opnorm1(n, A);
    int rc = 0;
    if (nA <= 2.1) {
        const double *C;
        int nc;
        if (nA > 0.95)       { C = PADE9; nc = 10; }
        else if (nA > 0.25)  { C = PADE7; nc = 8; }
        else if (nA > 0.015) { C = PADE5; nc = 6; }
        else                 { C = PADE3; nc = 4; }
        mm(n, A, A, A2);
        for (size_t e = 0; e < nn; ++e) { P[e] = 0.0; U[e] = 0.0; V[e] = 0.0; }
        for (int i = 0; i < n; ++i) { P[i + i * n] = 1.0; U[i + i * n] = C[1]; V[i + i * n] = C[0]; }
        for (int k = 1; k <= nc / 2 - 1; ++k) {
            mm(n, P, A2, T);                       /* P *= A2 */
            memcpy(P, T, sizeof(cplx) * nn);
            for (size_t e = 0; e < nn; ++e) {
                U[e] += C[2 * k + 1] * P[e];
                V[e] += C[2 * k] * P[e];
            }
        }
        mm(n, A, U, T);                            /* U = A * U */
        for (size_t e = 0; e < nn; ++e) {
            out[e] = V[e] + T[e];                  /* X = V + U */
            P[e] = V[e] - T[e];                    /* V - U     */
        }
        rc = gesv(n, P, out);
    } else {
        double s = log2(nA / 5.4);
        int si = 0;
        if (s > 0) {
            si = (int)ceil(s);
            double sc = ldexp(1.0, si);
            for (size_t e = 0; e < nn; ++e) A[e] /= sc;
        }
        const double *CC = PADE13;
        mm(n, A, A, A2);
        mm(n, A2, A2, A4);
        mm(n, A2, A4, A6);
        for (size_t e = 0; e < nn; ++e)
            T[e] = CC[13] * A6[e] + CC[11] * A4[e] + CC[9] * A2[e];
        mm(n, A6, T, U);
        for (size_t e = 0; e < nn; ++e)
            U[e] += CC[7] * A6[e] + CC[5] * A4[e] + CC[3] * A2[e];
        for (int i = 0; i < n; ++i) U[i + i * n] += CC[1];
        mm(n, A, U, P);                            /* U = A * (...) -> P */
        for (size_t e = 0; e < nn; ++e)
            T[e] = CC[12] * A6[e] + CC[10] * A4[e] + CC[8] * A2[e];
        mm(n, A6, T, V);
        for (size_t e = 0; e < nn; ++e)
            V[e] += CC[6] * A6[e] + CC[4] * A4[e] + CC[2] * A2[e];
        for (int i = 0; i < n; ++i) V[i + i * n] += CC[0];
        for (size_t e = 0; e < nn; ++e) {
            out[e] = V[e] + P[e];
            T[e] = V[e] - P[e];
        }
        rc = gesv(n, T, out);
        for (int t = 0; t < si && rc == 0; ++t) {  /* X *= X */
            mm(n, out, out, T);
            memcpy(out, T, sizeof(cplx) * nn);
        }
    }
    free(w);
    return rc;
}

/* ---------------------------------------------------------------- one member */

size_t oracle_member_workspace_bytes(int n, int N)
{
    return sizeof(cplx) * (size_t)n * n * ((size_t)3 * N + 2 + 6);
}

/* One member's figure of merit and K x N gradient, in the reference's operation order.
 * work: oracle_member_workspace_bytes(n, N) bytes (propagators[N], states[N+1],
 * costates[N+1], 6 scratch matrices).  Optional outs (may be NULL): props (n,n,N),
 * states (n,n,N+1), costates (n,n,N+1).  Returns 0 or a negative error. */
int oracle_member_eval_ws(int sys_type, int variant, int n, int K, int N, double T,
                          const cplx *A, const cplx *B, const cplx *Xi, const cplx *Xt,
                          const double *x, double *fom, double *grad,
                          cplx *props_out, cplx *states_out, cplx *costates_out, void *work)
{
    const size_t nn = (size_t)n * n;
    cplx *props = (cplx *)work;
    cplx *states = props + nn * N;
    cplx *costates = states + nn * (N + 1);
    cplx *H = costates + nn * (N + 1);
    cplx *store = H + nn, *t1 = store + nn, *t2 = t1 + nn, *t3 = t2 + nn, *t4 = t3 + nn;
    const int sandwich = (sys_type != ORACLE_UG);

    const double dt = T / N;                                   /* GRAPE.jl:42 */
    memcpy(states, Xi, sizeof(cplx) * nn);                     /* :44 */
    memcpy(costates + nn * N, Xt, sizeof(cplx) * nn);          /* :45 */

    const cplx mi_dt = CMPLX(-0.0, -1.0) * dt;                 /* (-1.0im * dt) */
    for (int i = 0; i < N; ++i) {
        if (variant == 0) {                                    /* timeevolution.jl:101-108 */
            for (size_t e = 0; e < nn; ++e) H[e] = 0.0;
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e)
                    H[e] = H[e] + B[e + nn * j] * x[j + (size_t)i * K];
            for (size_t e = 0; e < nn; ++e) t1[e] = mi_dt * (H[e] + A[e]);
        } else {                                               /* timeevolution.jl:49-53 */
            for (size_t e = 0; e < nn; ++e) H[e] = A[e];
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e)
                    H[e] = H[e] + B[e + nn * j] * x[j + (size_t)i * K];
            for (size_t e = 0; e < nn; ++e) t1[e] = mi_dt * H[e];
        }
        int rc = oracle_expm(n, t1, props + nn * i);
        if (rc) return rc;
    }

    for (int t = 0; t < N; ++t) {                              /* GRAPE.jl:53-63 */
        const cplx *P = props + nn * t;
        if (!sandwich) {
            mm(n, P, states + nn * t, states + nn * (t + 1));  /* :226 */
        } else {
            mm_a_bh(n, states + nn * t, P, store);             /* :245 */
            mm(n, P, store, states + nn * (t + 1));            /* :246 */
        }
    }
    for (int t = N - 1; t >= 0; --t) {                         /* GRAPE.jl:65-75 */
        const cplx *P = props + nn * t;
        if (!sandwich) {
            mm_ah_b(n, P, costates + nn * (t + 1), costates + nn * t);   /* :228 */
        } else {
            mm(n, costates + nn * (t + 1), P, store);          /* :248 */
            mm_ah_b(n, P, store, costates + nn * t);           /* :249 */
        }
    }

    for (int c = 0; c < K; ++c) {                              /* GRAPE.jl:79-92 */
        const cplx *Bc = B + nn * c;
        for (int t = 0; t < N; ++t) {
            const cplx *X = states + nn * t, *L = costates + nn * t;
            double g;
            if (!sandwich) {
                mm_ah_b(n, X, L, store);                       /* :271 store = X' L */
                mm_ah_b(n, L, Bc, t1);                         /* (L' * B) * X      */
                mm(n, t1, X, t2);
                cplx tr1 = trace(n, t2);
                /* in-place: (1.0im*dt) :272 ; static: (-1.0im*dt) :290 */
                const cplx idt = (variant == 0 ? CMPLX(0.0, 1.0) : CMPLX(-0.0, -1.0)) * dt;
                g = 2.0 * creal(idt * tr1 * trace(n, store));
            } else {
                mm(n, Bc, X, t1);                              /* tools.jl:17-19 */
                mm(n, X, Bc, t2);
                for (size_t e = 0; e < nn; ++e) t3[e] = t1[e] - t2[e];
                mm_ah_b(n, L, t3, store);                      /* :285 */
                const cplx idt = CMPLX(0.0, 1.0) * dt;
                for (size_t e = 0; e < nn; ++e) t4[e] = idt * store[e];
                g = creal(trace(n, t4));                       /* :286 */
            }
            grad[c + (size_t)t * K] = g;
        }
    }

    {                                                          /* GRAPE.jl:77,94: t = N (1-based) */
        const cplx *X = states + nn * (N - 1), *L = costates + nn * (N - 1);
        if (!sandwich) {                                       /* cost_functions.jl:99-101 */
            mm_ah_b(n, X, L, store);
            cplx z = trace(n, store);
            *fom = creal(z * z);
        } else {                                               /* :103-111 -> C1 :13-17 */
            mm_ah_b(n, L, X, store);
            cplx z = trace(n, store) / (double)n;
            *fom = 1.0 - (creal(z) * creal(z) + cimag(z) * cimag(z));
        }
    }

    if (props_out) memcpy(props_out, props, sizeof(cplx) * nn * N);
    if (states_out) memcpy(states_out, states, sizeof(cplx) * nn * (N + 1));
    if (costates_out) memcpy(costates_out, costates, sizeof(cplx) * nn * (N + 1));
    return 0;
}

int oracle_member_eval(int sys_type, int variant, int n, int K, int N, double T,
                       const cplx *A, const cplx *B, const cplx *Xi, const cplx *Xt,
                       const double *x, double *fom, double *grad,
                       cplx *props_out, cplx *states_out, cplx *costates_out)
{
    void *work = malloc(oracle_member_workspace_bytes(n, N));
    if (!work) return -2;
    int rc = oracle_member_eval_ws(sys_type, variant, n, K, N, T, A, B, Xi, Xt, x, fom, grad,
                                   props_out, states_out, costates_out, work);
    free(work);
    return rc;
}

/* ---------------------------------------------------------------- ensemble closure */

/* F = sum_k w_k F_k (k ascending), G = sum_k w_k g_k  -- src/solve.jl:164-196.
 * A: (n,n,E)  B: (n,n,K,E)  Xi,Xt: (n,n,E)  wts: E  x: (K,N)  G: (K,N).
 * member_grads (nullable): (K,N,E) unweighted per-member gradients; member_foms
 * (nullable): E unweighted foms.  n_threads<=1: the reference's serial loop; >1: OpenMP
 * over members (CPU-baseline variant B2), the weighted sums still taken in k order. */
int oracle_ensemble_eval(int sys_type, int variant, int n, int K, int N, int E, double T,
                         const cplx *A, const cplx *B, const cplx *Xi, const cplx *Xt,
                         const double *wts, const double *x, double *F, double *G,
                         double *member_foms, double *member_grads, int n_threads)
{
    const size_t nn = (size_t)n * n, KN = (size_t)K * N;
    double *grads = member_grads ? member_grads : (double *)malloc(sizeof(double) * KN * E);
    double *foms = member_foms ? member_foms : (double *)malloc(sizeof(double) * E);
    if (!grads || !foms) return -2;
    int rc_all = 0;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        void *work = malloc(oracle_member_workspace_bytes(n, N));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int k = 0; k < E; ++k) {
            int rc = work ? oracle_member_eval_ws(sys_type, variant, n, K, N, T, A + nn * k,
                                                  B + nn * K * k, Xi + nn * k, Xt + nn * k, x,
                                                  foms + k, grads + KN * k, NULL, NULL, NULL, work)
                          : -2;
            if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
                rc_all = rc;
            }
        }
        free(work);
    }
    if (rc_all == 0) {
        double f = 0.0;
        for (int k = 0; k < E; ++k) f += foms[k] * wts[k];     /* solve.jl:171-186 */
        if (F) *F = f;
        if (G) {
            for (size_t q = 0; q < KN; ++q) {                  /* solve.jl:191 */
                double s = 0.0;
                for (int k = 0; k < E; ++k) s += grads[q + KN * k] * wts[k];
                G[q] = s;
            }
        }
    }
    if (!member_grads) free(grads);
    if (!member_foms) free(foms);
    return rc_all;
}

/* ---------------------------------------------------------------- n x m states (UnitaryGate dispatch)
 *
 * The reference's UnitaryGate methods (evolve_func! src/GRAPE.jl:216-230, grad_func! :261-273,
 * fom_func src/cost_functions.jl:99-101) only ever left-multiply the states by n x n propagators and take
 * traces of m x m products, and init_GRAPE allocates `similar(Xi)` (src/grape_tools.jl:6-8): nothing
 * requires Xi, Xt to be square.  m = 1 is the vectorised density matrix evolved by Liouvillian
 * superoperators that test/liou.jl:38-48 writes out by hand.  Same operation order as
 * oracle_member_eval_ws with rectangular products.  Xi, Xt: (n, m) column-major. */
static void mm_rect(int n, int m, const cplx *P, const cplx *X, cplx *out)          /* (n x n)(n x m) */
{
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
            cplx s = 0.0;
            for (int k = 0; k < n; ++k)
                s += P[i + k * n] * X[k + j * n];
            out[i + j * n] = s;
        }
}

static void mm_ah_rect(int n, int m, const cplx *P, const cplx *L, cplx *out)       /* (n x n)' (n x m) */
{
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
            cplx s = 0.0;
            for (int k = 0; k < n; ++k)
                s += conj(P[k + i * n]) * L[k + j * n];
            out[i + j * n] = s;
        }
}

static cplx trace_xh_y(int n, int m, const cplx *X, const cplx *Y)                  /* tr(X' Y), both n x m */
{
    cplx tr = 0.0;                                       /* diagonal of the m x m product, column by column */
    for (int j = 0; j < m; ++j) {
        cplx s = 0.0;
        for (int k = 0; k < n; ++k)
            s += conj(X[k + j * n]) * Y[k + j * n];
        tr += s;
    }
    return tr;
}

int oracle_member_eval_rect(int variant, int n, int m, int K, int N, double T, const cplx *A, const cplx *B,
                            const cplx *Xi, const cplx *Xt, const double *x, double *fom, double *grad,
                            cplx *props_out, cplx *states_out, cplx *costates_out)
{
    const size_t nn = (size_t)n * n, nm = (size_t)n * m;
    cplx *props = (cplx *)malloc(sizeof(cplx) * (nn * N + 2 * nm * (N + 1) + 2 * nn + nm));
    if (!props) return -2;
    cplx *states = props + nn * N, *costates = states + nm * (N + 1);
    cplx *H = costates + nm * (N + 1), *t1 = H + nn, *bx = t1 + nn;
    const double dt = T / N;
    memcpy(states, Xi, sizeof(cplx) * nm);
    memcpy(costates + nm * N, Xt, sizeof(cplx) * nm);
    const cplx mi_dt = CMPLX(-0.0, -1.0) * dt;
    int rc = 0;
    for (int i = 0; i < N && rc == 0; ++i) {
        if (variant == 0) {
            for (size_t e = 0; e < nn; ++e) H[e] = 0.0;
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e)
                    H[e] = H[e] + B[e + nn * j] * x[j + (size_t)i * K];
            for (size_t e = 0; e < nn; ++e) t1[e] = mi_dt * (H[e] + A[e]);
        } else {
            for (size_t e = 0; e < nn; ++e) H[e] = A[e];
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e)
                    H[e] = H[e] + B[e + nn * j] * x[j + (size_t)i * K];
            for (size_t e = 0; e < nn; ++e) t1[e] = mi_dt * H[e];
        }
        rc = oracle_expm(n, t1, props + nn * i);
    }
    if (rc) { free(props); return rc; }
    for (int t = 0; t < N; ++t)                                   /* GRAPE.jl:226 */
        mm_rect(n, m, props + nn * t, states + nm * t, states + nm * (t + 1));
    for (int t = N - 1; t >= 0; --t)                              /* GRAPE.jl:228 */
        mm_ah_rect(n, m, props + nn * t, costates + nm * (t + 1), costates + nm * t);
    for (int c = 0; c < K; ++c)                                   /* GRAPE.jl:79-92, :261-273 */
        for (int t = 0; t < N; ++t) {
            const cplx *X = states + nm * t, *L = costates + nm * t;
            mm_rect(n, m, B + nn * c, X, bx);                     /* B X ; tr(L' B X) = tr(L' (B X)) */
            const cplx tr1 = trace_xh_y(n, m, L, bx);
            const cplx tr2 = trace_xh_y(n, m, X, L);              /* tr(X' L) */
            const cplx idt = (variant == 0 ? CMPLX(0.0, 1.0) : CMPLX(-0.0, -1.0)) * dt;
            grad[c + (size_t)t * K] = 2.0 * creal(idt * tr1 * tr2);
        }
    {
        const cplx z = trace_xh_y(n, m, states + nm * (N - 1), costates + nm * (N - 1));
        *fom = creal(z * z);                                      /* cost_functions.jl:99-101 */
    }
    if (props_out) memcpy(props_out, props, sizeof(cplx) * nn * N);
    if (states_out) memcpy(states_out, states, sizeof(cplx) * nm * (N + 1));
    if (costates_out) memcpy(costates_out, costates, sizeof(cplx) * nm * (N + 1));
    free(props);
    return 0;
}

/* ---------------------------------------------------------------- exact gradient / ADGRAPE functional
 *
 * The reference's ADGRAPE path (src/solve.jl:255-361, src/GRAPE.jl:12-20) minimises
 *     functional(x) = sum_k w_k C1(Xt_k, U Xi_k U')  (StateTransfer, :268-278 / :317-339)
 *                     sum_k w_k C1(Xt_k, U Xi_k)     (UnitaryGate,   :280-291 / :342-361)
 * with U = pw_evolve(...) = prod_t exp((-i dt)(A + sum_j B_j x[j,t]))  (src/timeevolution.jl:28-39) and takes
 * its gradient from Zygote, i.e. the EXACT derivative.  Zygote is a third-party AD package absent here; what it
 * returns is restated as the analytic derivative
 *     dPhi/dx[c,t] = tr(L_{t+1}' dP_t[c] X_t)                         (UnitaryGate)
 *                  = tr(L_{t+1}' (dP_t[c] X_t P_t' + P_t X_t dP_t[c]'))   (StateTransfer)
 * where dP_t[c] is the Frechet derivative of exp at -i dt H_t in direction -i dt B_c, taken from the
 * block-triangular identity  exp([[G, E], [0, G]]) = [[e^G, dexp_G(E)], [0, e^G]]  (Higham, Functions of
 * Matrices, thm 4.12) with the SAME Pade expm as everywhere else in this file -- a different algorithm from the
 * device's differentiated Taylor polynomial.  objective 0: the GRAPE figure of merit (fom_func) with its exact
 * gradient; objective 1: the C1 functional above for every system type.  tests/ pin this against central
 * finite differences of the objective and against 50-digit mpmath fixtures. */
int oracle_member_exact(int objective, int sys_type, int variant, int n, int K, int N, double T, const cplx *A,
                        const cplx *B, const cplx *Xi, const cplx *Xt, const double *x, double *fom, double *grad)
{
    const size_t nn = (size_t)n * n, n2 = (size_t)2 * n, nn4 = n2 * n2;
    const int sandwich = (sys_type != ORACLE_UG);
    cplx *w = (cplx *)malloc(sizeof(cplx) * (nn * N + 2 * nn * (N + 1) + 8 * nn + 2 * nn4));
    if (!w) return -2;
    cplx *props = w, *states = props + nn * N, *costates = states + nn * (N + 1);
    cplx *G = costates + nn * (N + 1), *t1 = G + nn, *t2 = t1 + nn, *dP = t2 + nn, *W1 = dP + nn, *W2 = W1 + nn,
         *Y = W2 + nn, *Y2 = Y + nn, *big = Y2 + nn, *bigout = big + nn4;
    const double dt = T / N;
    const cplx mi_dt = CMPLX(-0.0, -1.0) * dt;
    int rc = 0;
    memcpy(states, Xi, sizeof(cplx) * nn);
    memcpy(costates + nn * N, Xt, sizeof(cplx) * nn);
    for (int i = 0; i < N && rc == 0; ++i) {
        if (variant == 0) {
            for (size_t e = 0; e < nn; ++e) G[e] = 0.0;
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e) G[e] = G[e] + B[e + nn * j] * x[j + (size_t)i * K];
            for (size_t e = 0; e < nn; ++e) t1[e] = mi_dt * (G[e] + A[e]);
        } else {
            for (size_t e = 0; e < nn; ++e) G[e] = A[e];
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e) G[e] = G[e] + B[e + nn * j] * x[j + (size_t)i * K];
            for (size_t e = 0; e < nn; ++e) t1[e] = mi_dt * G[e];
        }
        rc = oracle_expm(n, t1, props + nn * i);
    }
    for (int t = 0; t < N && rc == 0; ++t) {
        const cplx *P = props + nn * t;
        if (!sandwich) mm(n, P, states + nn * t, states + nn * (t + 1));
        else { mm_a_bh(n, states + nn * t, P, t1); mm(n, P, t1, states + nn * (t + 1)); }
    }
    for (int t = N - 1; t >= 0 && rc == 0; --t) {
        const cplx *P = props + nn * t;
        if (!sandwich) mm_ah_b(n, P, costates + nn * (t + 1), costates + nn * t);
        else { mm(n, costates + nn * (t + 1), P, t1); mm_ah_b(n, P, t1, costates + nn * t); }
    }
    if (rc) { free(w); return rc; }
    /* Phi = tr(Xt' X_N) */
    cplx Phi = 0.0;
    for (size_t e = 0; e < nn; ++e) Phi += conj(Xt[e]) * states[nn * N + e];
    const int c1_obj = sandwich || objective == 1;
    const double D2 = 1.0 / ((double)n * (double)n);
    *fom = c1_obj ? 1.0 - D2 * (creal(Phi) * creal(Phi) + cimag(Phi) * cimag(Phi)) : creal(conj(Phi) * conj(Phi));
    for (int t = 0; t < N && rc == 0; ++t) {
        const cplx *P = props + nn * t, *X = states + nn * t, *Ln = costates + nn * (t + 1);
        /* the generator of this slice again (same summation order as above) */
        if (variant == 0) {
            for (size_t e = 0; e < nn; ++e) G[e] = 0.0;
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e) G[e] = G[e] + B[e + nn * j] * x[j + (size_t)t * K];
            for (size_t e = 0; e < nn; ++e) G[e] = mi_dt * (G[e] + A[e]);
        } else {
            for (size_t e = 0; e < nn; ++e) G[e] = A[e];
            for (int j = 0; j < K; ++j)
                for (size_t e = 0; e < nn; ++e) G[e] = G[e] + B[e + nn * j] * x[j + (size_t)t * K];
            for (size_t e = 0; e < nn; ++e) G[e] = mi_dt * G[e];
        }
        if (sandwich) {
            mm_a_bh(n, X, P, Y);            /* X P'          */
            mm_a_bh(n, Y, Ln, W1);          /* X P' L'       */
            mm_ah_b(n, P, Ln, Y2);          /* P' L          */
            mm_ah_b(n, X, Y2, W2);          /* X' P' L       */
        } else {
            mm_a_bh(n, X, Ln, W1);          /* X L'          */
        }
        for (int c = 0; c < K && rc == 0; ++c) {
            for (size_t e = 0; e < nn4; ++e) big[e] = 0.0;
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) {
                    big[i + j * n2] = G[i + j * n];
                    big[(i + n) + (j + n) * n2] = G[i + j * n];
                    big[i + (j + n) * n2] = mi_dt * B[nn * c + i + j * n];
                }
            rc = oracle_expm(2 * n, big, bigout);
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) dP[i + j * n] = bigout[i + (j + n) * n2];
            cplx dPhi = 0.0;                 /* tr(dP W1) */
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) dPhi += dP[i + j * n] * W1[j + i * n];
            if (sandwich) {
                cplx b = 0.0;
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i < n; ++i) b += dP[i + j * n] * W2[j + i * n];
                dPhi += conj(b);
            }
            grad[c + (size_t)t * K] = c1_obj ? -2.0 * D2 * creal(conj(Phi) * dPhi) : 2.0 * creal(Phi * dPhi);
        }
    }
    free(w);
    return rc;
}

/* C1(KT, KN) = 1 - |tr(KT' KN)/D|^2   -- src/cost_functions.jl:13-17 */
double oracle_C1(int n, const cplx *KT, const cplx *KN)
{
    cplx s = 0.0;
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k)
            s += conj(KT[k + j * n]) * KN[k + j * n];
    s /= (double)n;
    return 1.0 - (creal(s) * creal(s) + cimag(s) * cimag(s));
}
