"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): a NumPy restatement of the optimiser the reference hands its
closure to,

    Optim.optimize(Optim.only_fg!(topt), x0, Optim.LBFGS(), optim_options)          src/solve.jl:138, :244

i.e. Optim.jl's `LBFGS()` with its defaults -- m = 10, `alphaguess = LineSearches.InitialStatic()` (alpha = 1, which never
sets `mayterminate`), `linesearch = LineSearches.HagerZhang()` (delta 0.1, sigma 0.9, rho 5, epsilon 1e-6, gamma 0.66,
psi3 0.1, linesearchmax 50), `scaleinvH0 = true`, flat manifold, no preconditioner -- driven by any f/g callable.  Optim.jl
and LineSearches.jl are third-party to the reference tree (Project.toml names Optim without pinning a version; there is no
Manifest and no Julia in this image), so this file restates their PUBLISHED algorithm function by function:

    Optim.jl       src/multivariate/solvers/first_order/l_bfgs.jl   (twoloop!, update_state!, update_h!)
                   src/utilities/perform_linesearch.jl               (perform_linesearch!, reset on a non-descent direction)
                   src/multivariate/optimize/optimize.jl             (the iteration loop, g-norm convergence on |g|_inf)
    LineSearches.jl src/hagerzhang.jl                                (the search, secant2!, update!, bisect!, satisfies_wolfe)
                   src/initialguess.jl                               (InitialStatic)

It pins `grape_lbfgs(line_search = 1)` iterate by iterate (tests/test_gpu_lbfgs.py::test_iterates_match_the_host_restatement):
accepted step lengths, evaluation counts, iterates.  Parity status as for the oracle: unpinned against a run of Optim.jl
itself (none can be made here)."""
import math

import numpy as np

EPS = np.finfo(np.float64).eps


class LineSearchFailure(Exception):
    def __init__(self, msg, alpha):
        super().__init__(msg)
        self.alpha = alpha


class HagerZhang:
    """LineSearches.jl src/hagerzhang.jl, v7: `(ls::HagerZhang)(phidphi, c, phi_0, dphi_0)`."""

    def __init__(self, delta=0.1, sigma=0.9, alphamax=math.inf, rho=5.0, epsilon=1e-6, gamma=0.66, linesearchmax=50, psi3=0.1):
        self.delta, self.sigma, self.alphamax, self.rho = delta, sigma, alphamax, rho
        self.epsilon, self.gamma, self.linesearchmax, self.psi3 = epsilon, gamma, linesearchmax, psi3
        self.mayterminate = False          # Ref{Bool}: InitialStatic never sets it (InitialQuadratic / InitialHagerZhang do)

    def satisfies_wolfe(self, c, phi_c, dphi_c, phi_0, dphi_0, phi_lim):
        wolfe1 = self.delta * dphi_0 >= (phi_c - phi_0) / c and dphi_c >= self.sigma * dphi_0
        wolfe2 = (2 * self.delta - 1) * dphi_0 >= dphi_c >= self.sigma * dphi_0 and phi_c <= phi_lim
        return wolfe1 or wolfe2

    @staticmethod
    def secant(a, b, dphi_a, dphi_b):
        return (a * dphi_b - b * dphi_a) / (dphi_b - dphi_a)

    def bisect(self, phidphi, al, va, sl, ia, ib, phi_lim):
        """HZ, stage U3 (theta = 0.5)"""
        a, b = al[ia], al[ib]
        assert sl[ia] < 0 and va[ia] <= phi_lim and sl[ib] < 0 and va[ib] > phi_lim and b > a
        while b - a > EPS * abs(b):                          # eps(b)
            d = (a + b) / 2
            phi_d, gphi = phidphi(d)
            assert math.isfinite(phi_d) and math.isfinite(gphi)
            al.append(d); va.append(phi_d); sl.append(gphi)
            idx = len(al) - 1
            if gphi >= 0:
                return ia, idx                               # replace b, return
            if phi_d <= phi_lim:
                a, ia = d, idx                               # replace a, keep bisecting until dphi_b > 0
            else:
                b, ib = d, idx
        return ia, ib

    def update(self, phidphi, al, va, sl, ia, ib, ic, phi_lim):
        """HZ, stages U0-U3"""
        a, b = al[ia], al[ib]
        assert sl[ia] < 0 and va[ia] <= phi_lim and sl[ib] >= 0 and b > a
        c, phi_c, dphi_c = al[ic], va[ic], sl[ic]
        if c < a or c > b:
            return ia, ib                                    # outside the bracketing interval
        if dphi_c >= 0:
            return ia, ic                                    # replace b with a closer point
        if phi_c <= phi_lim:
            return ic, ib                                    # replace a
        return self.bisect(phidphi, al, va, sl, ia, ic, phi_lim)

    def secant2(self, phidphi, al, va, sl, ia, ib, phi_lim):
        phi_0, dphi_0 = va[0], sl[0]
        a, b, dphi_a, dphi_b = al[ia], al[ib], sl[ia], sl[ib]
        assert dphi_a < 0 and dphi_b >= 0
        c = self.secant(a, b, dphi_a, dphi_b)
        assert math.isfinite(c)
        phi_c, dphi_c = phidphi(c)
        assert math.isfinite(phi_c) and math.isfinite(dphi_c)
        al.append(c); va.append(phi_c); sl.append(dphi_c)
        ic = len(al) - 1
        if self.satisfies_wolfe(c, phi_c, dphi_c, phi_0, dphi_0, phi_lim):
            return True, ic, ic
        iA, iB = self.update(phidphi, al, va, sl, ia, ib, ic, phi_lim)
        a, b = al[iA], al[iB]
        if iB == ic:                                         # b was updated: make sure a is too
            c = self.secant(al[ib], al[iB], sl[ib], sl[iB])
        elif iA == ic:                                       # a was updated: do it for b too
            c = self.secant(al[ia], al[iA], sl[ia], sl[iA])
        if (iA == ic or iB == ic) and a <= c <= b:
            phi_c, dphi_c = phidphi(c)
            assert math.isfinite(phi_c) and math.isfinite(dphi_c)
            al.append(c); va.append(phi_c); sl.append(dphi_c)
            ic = len(al) - 1
            if self.satisfies_wolfe(c, phi_c, dphi_c, phi_0, dphi_0, phi_lim):
                return True, ic, ic
            iA, iB = self.update(phidphi, al, va, sl, iA, iB, ic, phi_lim)
        return False, iA, iB

    def __call__(self, phidphi, c, phi_0, dphi_0):
        """returns (alpha, phi(alpha)); raises LineSearchFailure as LineSearches.jl throws LineSearchException"""
        delta, sigma, rho, gamma, psi3 = self.delta, self.sigma, self.rho, self.gamma, self.psi3
        alphamax = self.alphamax
        if not (math.isfinite(phi_0) and math.isfinite(dphi_0)):
            raise LineSearchFailure("Value and slope at step length = 0 must be finite.", 0.0)
        if dphi_0 >= EPS * abs(phi_0):
            raise LineSearchFailure("Search direction is not a direction of descent.", 0.0)
        elif dphi_0 >= 0:
            return 0.0, phi_0
        iterfinitemax = math.ceil(-math.log2(EPS))
        al, va, sl = [0.0], [phi_0], [dphi_0]
        phi_lim = phi_0 + self.epsilon * abs(phi_0)
        assert c >= 0
        if c <= EPS:
            return 0.0, phi_0
        assert math.isfinite(c) and c <= alphamax
        phi_c, dphi_c = phidphi(c)
        iterfinite = 1
        while not (math.isfinite(phi_c) and math.isfinite(dphi_c)) and iterfinite < iterfinitemax:
            self.mayterminate = False
            iterfinite += 1
            c *= psi3
            phi_c, dphi_c = phidphi(c)
        if not (math.isfinite(phi_c) and math.isfinite(dphi_c)):
            self.mayterminate = False
            return 0.0, phi_0
        al.append(c); va.append(phi_c); sl.append(dphi_c)
        # a c generated by quadratic interpolation may terminate at once (never behind InitialStatic)
        if self.mayterminate and self.satisfies_wolfe(c, phi_c, dphi_c, phi_0, dphi_0, phi_lim):
            self.mayterminate = False
            return c, phi_c
        # initial bracketing, HZ stages B0-B3
        isbracketed, ia, ib, it = False, 0, 1, 1
        while not isbracketed and it < self.linesearchmax:
            if dphi_c >= 0:
                # the upward slope: this is b; the last earlier point with a small enough value is a
                ib = len(al) - 1
                for i in range(ib - 1, -1, -1):
                    if va[i] <= phi_lim:
                        ia = i
                        break
                isbracketed = True
            elif va[-1] > phi_lim:
                # higher value, downward slope: over the crest -- bisect
                ib, ia = len(al) - 1, 0
                ia, ib = self.bisect(phidphi, al, va, sl, ia, ib, phi_lim)
                isbracketed = True
            else:
                # still going downhill: expand
                cold, phi_cold = c, phi_c
                if np.nextafter(cold, math.inf) >= alphamax:
                    self.mayterminate = False
                    return cold, phi_cold
                c *= rho
                if c > alphamax:
                    c = alphamax
                phi_c, dphi_c = phidphi(c)
                iterfinite = 1
                while not (math.isfinite(phi_c) and math.isfinite(dphi_c)) and c > np.nextafter(cold, math.inf) and iterfinite < iterfinitemax:
                    alphamax = c
                    iterfinite += 1
                    c = (cold + c) / 2
                    phi_c, dphi_c = phidphi(c)
                if not (math.isfinite(phi_c) and math.isfinite(dphi_c)):
                    return cold, phi_cold
                al.append(c); va.append(phi_c); sl.append(dphi_c)
            it += 1
        while it < self.linesearchmax:
            a, b = al[ia], al[ib]
            assert b > a
            if b - a <= EPS * abs(b):
                self.mayterminate = False
                return a, va[ia]
            iswolfe, iA, iB = self.secant2(phidphi, al, va, sl, ia, ib, phi_lim)
            if iswolfe:
                self.mayterminate = False
                return al[iA], va[iA]
            A, B = al[iA], al[iB]
            assert B > A
            if B - A < gamma * (b - a):
                if np.nextafter(va[ia], math.inf) >= va[ib] and np.nextafter(va[iA], math.inf) >= va[iB]:
                    self.mayterminate = False                # so flat that secant did nothing useful
                    return A, va[iA]
                ia, ib = iA, iB
            else:
                c = (A + B) / 2                              # secant converges too slowly: bisection
                phi_c, dphi_c = phidphi(c)
                assert math.isfinite(phi_c) and math.isfinite(dphi_c)
                al.append(c); va.append(phi_c); sl.append(dphi_c)
                ia, ib = self.update(phidphi, al, va, sl, iA, iB, len(al) - 1, phi_lim)
            it += 1
        raise LineSearchFailure(f"Linesearch failed to converge, reached maximum iterations {self.linesearchmax}.", al[ia])


def lbfgs(fg, x0, m=10, iterations=1000, g_tol=1e-8, linesearch=None):
    """Optim.optimize(only_fg!(fg), x0, LBFGS(m = m), Options(iterations = iterations, g_tol = g_tol)) with f_tol = x_tol = 0.
    fg(x) -> (f, g) for x of x0's shape.  Returns a dict: minimizer, minimum, g_norm, iterations, evaluations, status and
    `trace` -- per iteration (alpha accepted, cumulative evaluations, f, |g|_inf, x copy)."""
    ls = linesearch or HagerZhang()
    shape = np.shape(x0)
    x = np.array(x0, dtype=np.float64).reshape(-1).copy()
    n = x.size
    calls = {"n": 0, "x": None, "f": None, "g": None}

    def value_gradient(xv):
        # NLSolversBase caches the last evaluated point: asking again for the same x costs nothing
        if calls["x"] is not None and np.array_equal(calls["x"], xv):
            return calls["f"], calls["g"]
        f, g = fg(xv.reshape(shape))
        calls["n"] += 1
        calls["x"], calls["f"], calls["g"] = xv.copy(), float(f), np.array(g, dtype=np.float64).reshape(-1).copy()
        return calls["f"], calls["g"]

    f_x, g = value_gradient(x)
    g = g.copy()
    dx_hist, dg_hist, rho = np.zeros((m, n)), np.zeros((m, n)), np.zeros(m)
    pseudo_iteration = 0
    trace = []
    status = "max_iterations"
    if np.abs(g).max() <= g_tol:                                   # initial_convergence
        return {"minimizer": x.reshape(shape), "minimum": f_x, "g_norm": float(np.abs(g).max()), "iterations": 0,
                "evaluations": calls["n"], "status": "g_tol", "trace": trace}
    it = 0
    counter_f_tol = 0
    s = np.zeros(n)
    while it < iterations:
        it += 1
        # ---- update_state!: direction (twoloop!), line search, step
        pseudo_iteration += 1
        lower, upper = pseudo_iteration - m, pseudo_iteration - 1
        q = g.copy()
        alpha_tl = np.zeros(m)
        for index in range(upper, lower - 1, -1):
            if index < 1:
                continue
            i = (index - 1) % m                                    # mod1(index, m), 0-based
            alpha_tl[i] = rho[i] * np.dot(dx_hist[i], q)
            q -= alpha_tl[i] * dg_hist[i]
        if pseudo_iteration > 1:                                   # scaleinvH0: Nocedal & Wright (7.20)
            i = (upper - 1) % m
            scaling = np.dot(dx_hist[i], dg_hist[i]) / np.dot(dg_hist[i], dg_hist[i])
            s = scaling * q
        else:
            s = q.copy()                                           # identity preconditioner
        for index in range(lower, upper + 1):
            if index < 1:
                continue
            i = (index - 1) % m
            beta = rho[i] * np.dot(dg_hist[i], s)
            s += dx_hist[i] * (alpha_tl[i] - beta)
        s = -s
        g_previous = g.copy()
        # perform_linesearch!
        dphi_0 = float(np.dot(g, s))
        if dphi_0 >= 0:                                            # reset_search_direction!
            pseudo_iteration = 1
            s = -g
            dphi_0 = float(np.dot(g, s))
        phi_0 = f_x
        alpha0 = 1.0                                               # InitialStatic(alpha = 1.0, scaled = false)

        def phidphi(a, x=x, s=s):
            f, gg = value_gradient(x + a * s)
            return f, float(np.dot(gg, s))
        ls_ok = True
        try:
            alpha, _ = ls(phidphi, alpha0, phi_0, dphi_0)
        except LineSearchFailure as ex:
            alpha, ls_ok = ex.alpha, False
        dx = alpha * s
        x_previous, f_x_previous = x, f_x
        x = x + dx
        if not ls_ok:
            # optimize.jl: `ls_success = !update_state!(...); if !ls_success break end` -- the step of the exception's alpha is
            # taken, nothing is evaluated any more
            trace.append({"alpha": float(alpha), "evaluations": calls["n"], "f": calls["f"], "g_norm": float("nan"), "x": x.copy()})
            status = "linesearch_failed"
            break
        # ---- update_g!
        f_x, g_new = value_gradient(x)
        g = g_new.copy()
        gnorm = float(np.abs(g).max())
        trace.append({"alpha": float(alpha), "evaluations": calls["n"], "f": f_x, "g_norm": gnorm, "x": x.copy()})
        # ---- assess_convergence with Optim.Options() defaults: x_abstol = x_reltol = f_abstol = f_reltol = 0, g_abstol = g_tol,
        # successive_f_tol = 1.  With zero tolerances `<=` still fires on an EXACT zero: a step of length 0 is "x converged",
        # two successive iterations without any change of f are "f converged".
        x_converged = float(np.abs(x - x_previous).max()) <= 0.0
        f_converged = abs(f_x - f_x_previous) <= 0.0
        counter_f_tol = counter_f_tol + 1 if f_converged else 0
        if x_converged or gnorm <= g_tol or counter_f_tol > 1:
            status = "x_tol" if x_converged else ("g_tol" if gnorm <= g_tol else "f_tol")
            break
        # ---- update_h!
        dg = g - g_previous
        sy = float(np.dot(dx, dg))
        if sy == 0.0:                                              # isinf(1 / dot(dx, dg))
            pseudo_iteration = 0
            continue
        idx = (pseudo_iteration - 1) % m
        dx_hist[idx], dg_hist[idx], rho[idx] = dx, dg, 1.0 / sy
    return {"minimizer": x.reshape(shape), "minimum": f_x, "g_norm": float(np.abs(g).max()), "iterations": it,
            "evaluations": calls["n"], "status": status, "trace": trace}
