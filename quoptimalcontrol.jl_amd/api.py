"""Host-side mirror of the reference's user surface for the GRAPE path.

Names, fields and argument meaning follow /root/reference so that the parity tests read like
the reference's own tests (test/state_transfer_tests.jl, test/unitary_gate_tests.jl):

  Problem(B, A, Xi, Xt, T, n_controls, guess, sys_type)        src/problems.jl:19-28
  EnsembleProblem(prob, n_ens, A_g, B_g, XiG, XtG, wts)        src/problems.jl:33-41
  StateTransfer() / UnitaryGate() / CoherenceTransfer()        src/problems.jl:8-10
  GRAPE(n_slices=..., isinplace=True, optim_options=...)       src/solve.jl:33-42
  solve(prob, alg) -> SolutionResult / EnsembleSolutionResult  src/solve.jl:63-143, :145-250
  init_ensemble(ens)                                           src/tools.jl:42-53
  C1(KT, KN)                                                   src/cost_functions.jl:13-17

What differs, on purpose: the body of the (F, G, x) closure is one call into libgrape_hip.so
(engine.GrapeEngine) instead of _fom_and_gradient_GRAPE!, and the L-BFGS driver is SciPy's
(the optimiser is outside the hot path; Optim.jl is a third-party dependency of the
reference).  Generator callables A_g/B_g/XiG/XtG receive the 1-based member index k, as in
Julia.
"""
from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import json

import numpy as np

from .engine import GrapeEngine


class _SysType:
    name = ""

    def __repr__(self):
        return f"{self.name}()"

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(self.name)


class StateTransfer(_SysType):
    name = "StateTransfer"


class UnitaryGate(_SysType):
    name = "UnitaryGate"


class CoherenceTransfer(_SysType):
    name = "CoherenceTransfer"


@dataclass
class Problem:
    B: Any                 # list of K control operators (n x n)
    A: Any                 # drift operator (n x n)
    Xi: Any                # initial state / operator
    Xt: Any                # target state / operator
    T: float               # pulse duration
    n_controls: int
    guess: Any             # (K, N) initial controls
    sys_type: _SysType


@dataclass
class EnsembleProblem:
    prob: Problem
    n_ens: int
    A_g: Callable[[int], Any]
    B_g: Callable[[int], Any]
    XiG: Callable[[int], Any]
    XtG: Callable[[int], Any]
    wts: Any


@dataclass
class GRAPE:
    n_slices: int
    expm_method: str = "fast"      # stored and never read, like the reference (src/solve.jl:39-41,66)
    isinplace: bool = True
    optim_options: dict = field(default_factory=dict)
    device: int = -1               # HIP device ordinal (new: the reference has no devices)
    optimizer: str = "host"        # "host": SciPy L-BFGS-B drives grape_eval (stand-in for Optim.jl on the host);
                                   # "device": grape_lbfgs, the library's device-resident L-BFGS (Hager-Zhang line search;
                                   # optim_options["line_search"]: "hagerzhang" | "optim" | "ladder")
    devices: Optional[list] = None  # HIP ordinals: the library shards the ensemble over them (grape_config.n_devices /
                                    # device_ids, as julia/GrapeHIP.jl's `devices`) and all-reduces [G, F] once per evaluation
    peer_sum: bool = False         # with `devices`: GRAPE_FLAG_GROUP_PEER_SUM (sum on the first device, ids may repeat)


@dataclass
class ADGRAPE:
    """src/solve.jl:44-52: the functional path.  The reference differentiates
    functional(x) = sum_k w_k C1(Xt_k, U Xi_k [U']) (src/solve.jl:268-361, U = pw_evolve) with Zygote; here the same
    functional and its exact gradient come from the device (objective "c1", gradient "exact")."""
    n_slices: int
    expm_method: str = "fast"
    optim_options: dict = field(default_factory=dict)
    device: int = -1
    optimizer: str = "host"
    devices: Optional[list] = None
    peer_sum: bool = False


@dataclass
class SolutionResult:
    result: Any
    fidelity: float
    opti_pulses: Any
    problem: Problem
    alg: GRAPE


@dataclass
class EnsembleSolutionResult:
    result: Any
    fidelity: float
    opti_pulses: Any
    problem: EnsembleProblem
    alg: GRAPE


def C1(KT, KN):
    """Density-matrix infidelity, src/cost_functions.jl:13-17."""
    KT = np.asarray(KT, complex)
    KN = np.asarray(KN, complex)
    D = KT.shape[0]
    return 1.0 - abs(np.trace(KT.conj().T @ KN) / D) ** 2


def init_ensemble(ens):
    """src/tools.jl:42-53: one Problem per member with A, B, Xi, Xt from the generators."""
    out = []
    for k in range(1, ens.n_ens + 1):
        p = ens.prob
        out.append(Problem(B=ens.B_g(k), A=ens.A_g(k), Xi=ens.XiG(k), Xt=ens.XtG(k), T=p.T,
                           n_controls=p.n_controls, guess=p.guess, sys_type=p.sys_type))
    return out


def _pack(problems):
    A = np.array([np.asarray(p.A, complex) for p in problems])
    B = np.array([[np.asarray(b, complex) for b in p.B] for p in problems])
    Xi = np.array([np.asarray(p.Xi, complex) for p in problems])
    Xt = np.array([np.asarray(p.Xt, complex) for p in problems])
    return A, B, Xi, Xt


def make_engine(prob, alg, **engine_kw):
    """Build the device context for a Problem (E = 1) or an EnsembleProblem."""
    if isinstance(prob, EnsembleProblem):
        members = init_ensemble(prob)
        wts = np.asarray(prob.wts, dtype=np.float64)
    else:
        members = [prob]
        wts = np.ones(1)
    first = members[0]
    if len(first.B) != first.n_controls:
        raise ValueError("n_controls does not match the number of control operators")
    A, B, Xi, Xt = _pack(members)
    if getattr(alg, "devices", None):
        from .engine import FLAG_GROUP_PEER_SUM
        engine_kw = dict(engine_kw, devices=list(alg.devices))
        if getattr(alg, "peer_sum", False):
            engine_kw["flags"] = engine_kw.get("flags", 0) | FLAG_GROUP_PEER_SUM
    if isinstance(alg, ADGRAPE):            # pw_evolve adds A first (src/timeevolution.jl:32-35): the static summation order
        return GrapeEngine(first.sys_type.name, A, B, Xi, Xt, wts, first.T, alg.n_slices, variant=1, device=alg.device,
                           gradient="exact", objective="c1", **engine_kw)
    return GrapeEngine(first.sys_type.name, A, B, Xi, Xt, wts, first.T, alg.n_slices,
                       variant=0 if alg.isinplace else 1, device=alg.device, **engine_kw)


def fom_and_gradient(prob, alg, x, engine=None):
    """One call of the closure `topt(F, G, x)` (src/solve.jl:75-100 / :164-196)."""
    own = engine is None
    eng = engine or make_engine(prob, alg)
    try:
        return eng.eval(x)
    finally:
        if own:
            eng.close()


def _json_default(v):
    """Values `json` does not know inside alg.optim_options: NumPy scalars / arrays become numbers / lists, anything else
    (callables, objects) its repr -- a solve that succeeded must stay savable (the reference's BSON takes any value)."""
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (set, frozenset, tuple)):
        return list(v)
    return repr(v)


def save(solres, file_path):
    """src/tools.jl:59-72: every field of a SolutionResult / EnsembleSolutionResult but the optimiser's own result object
    (the reference drops it too: `fieldnames(...)[2:end]`) into one file -- NumPy's .npz here instead of BSON.  The
    problem's operators and states are stored as arrays; generator closures of an EnsembleProblem are stored as the
    members they generate."""
    prob = solres.problem
    ens = isinstance(prob, EnsembleProblem)
    base = prob.prob if ens else prob
    alg = solres.alg
    data = {
        "kind": np.array("ensemble" if ens else "single"),
        "fidelity": np.array(float(solres.fidelity)),
        "opti_pulses": np.asarray(solres.opti_pulses, dtype=np.float64),
        "sys_type": np.array(base.sys_type.name), "T": np.array(float(base.T)), "n_controls": np.array(int(base.n_controls)),
        "A": np.asarray(base.A, complex), "B": np.asarray(base.B, complex), "Xi": np.asarray(base.Xi, complex),
        "Xt": np.asarray(base.Xt, complex), "guess": np.asarray(base.guess, dtype=np.float64),
        "alg_kind": np.array(type(alg).__name__), "n_slices": np.array(int(alg.n_slices)),
        "isinplace": np.array(bool(getattr(alg, "isinplace", True))),
        # the whole alg struct, as the reference's save does (src/tools.jl:59-72): a re-solve from a loaded result must use
        # the optimiser, tolerances and devices of the saved run, not the defaults
        "alg_fields": np.array(json.dumps({"expm_method": alg.expm_method, "optim_options": alg.optim_options,
                                            "device": int(alg.device), "optimizer": alg.optimizer,
                                            "devices": None if alg.devices is None else [int(v) for v in alg.devices],
                                            "peer_sum": bool(alg.peer_sum)}, default=_json_default)),
    }
    if ens:
        members = init_ensemble(prob)
        data.update(n_ens=np.array(int(prob.n_ens)), wts=np.asarray(prob.wts, dtype=np.float64),
                    A_members=np.array([np.asarray(m.A, complex) for m in members]),
                    B_members=np.array([np.asarray(m.B, complex) for m in members]),
                    Xi_members=np.array([np.asarray(m.Xi, complex) for m in members]),
                    Xt_members=np.array([np.asarray(m.Xt, complex) for m in members]))
    with open(file_path, "wb") as io:
        np.savez(io, **data)


def load(file_path):
    """src/tools.jl:75-85 (whose own version passes four arguments to a five-field struct): the SolutionResult back,
    with `result = None` as the reference intends."""
    d = np.load(file_path, allow_pickle=False)
    st = {"StateTransfer": StateTransfer, "UnitaryGate": UnitaryGate, "CoherenceTransfer": CoherenceTransfer}[str(d["sys_type"])]()
    base = Problem(B=list(d["B"]), A=d["A"], Xi=d["Xi"], Xt=d["Xt"], T=float(d["T"]), n_controls=int(d["n_controls"]),
                   guess=d["guess"], sys_type=st)
    alg_cls = ADGRAPE if str(d["alg_kind"]) == "ADGRAPE" else GRAPE
    extra = json.loads(str(d["alg_fields"])) if "alg_fields" in d.files else {}      # (files of earlier rounds: defaults)
    alg = alg_cls(n_slices=int(d["n_slices"]), **extra) if alg_cls is ADGRAPE else GRAPE(n_slices=int(d["n_slices"]),
                                                                                           isinplace=bool(d["isinplace"]), **extra)
    if str(d["kind"]) == "ensemble":
        Am, Bm, Xim, Xtm = d["A_members"], d["B_members"], d["Xi_members"], d["Xt_members"]
        ens = EnsembleProblem(prob=base, n_ens=int(d["n_ens"]), A_g=lambda k: Am[k - 1], B_g=lambda k: list(Bm[k - 1]),
                              XiG=lambda k: Xim[k - 1], XtG=lambda k: Xtm[k - 1], wts=d["wts"])
        return EnsembleSolutionResult(None, float(d["fidelity"]), d["opti_pulses"], ens, alg)
    return SolutionResult(None, float(d["fidelity"]), d["opti_pulses"], base, alg)


def pulse_to_file(pulse, file_path, duration=None):
    """src/tools.jl:90-104: write a (K, N) pulse as delimited text with time going down the file
    (`writedlm(io, pulse')`, tab separated); with `duration`, a leading time column
    `range(0, duration, length=N)` like the reference's 3-argument method (which takes N = length(pulse),
    i.e. a single-control pulse)."""
    P = np.atleast_2d(np.asarray(pulse, dtype=np.float64))
    rows = P.T
    if duration is not None:
        t = np.linspace(0.0, float(duration), rows.shape[0])
        rows = np.column_stack([t, rows])
    with open(file_path, "w") as io:
        for r in rows:
            io.write("\t".join(repr(float(v)) for v in r) + "\n")


def pulse_from_file(file_path, has_time_column=False):
    """inverse of pulse_to_file: returns the (K, N) pulse (and the time column if present)."""
    data = np.loadtxt(file_path, delimiter="\t", ndmin=2)
    if has_time_column:
        return np.ascontiguousarray(data[:, 1:].T), data[:, 0].copy()
    return np.ascontiguousarray(data.T)


def _lbfgs(topt, x0, options):
    """Stand-in for Optim.optimize(Optim.only_fg!(topt), x0, LBFGS(), opts) (src/solve.jl:138)."""
    from scipy.optimize import minimize

    shape = x0.shape
    opts = {"maxiter": int(options.get("iterations", 1000)), "gtol": float(options.get("g_tol", 1e-8)),
            "ftol": float(options.get("f_tol", 0.0)) or 1e-15, "maxls": 40}

    def fun(xflat):
        F, G = topt(xflat.reshape(shape))
        return F, G.reshape(-1)

    res = minimize(fun, np.asarray(x0, float).reshape(-1), jac=True, method="L-BFGS-B", options=opts)
    res.minimum = float(res.fun)
    res.minimizer = res.x.reshape(shape)
    return res


def _device_lbfgs(eng, x0, options):
    """grape_lbfgs with Optim-style options (iterations, g_tol, f_tol); result mimics the fields solve() reads."""
    from types import SimpleNamespace

    x_min, info = eng.lbfgs(x0, iterations=int(options.get("iterations", 0)), g_tol=float(options.get("g_tol", -1.0)),
                            f_tol=float(options.get("f_tol", 0.0)), line_search=options.get("line_search", "hagerzhang"),
                            probes=int(options.get("probes", 0)))
    return SimpleNamespace(minimum=info["minimum"], minimizer=x_min, x=x_min.reshape(-1), fun=info["minimum"],
                           nit=info["iterations"], nfev=info["evaluations"], message=info["message"],
                           success=info["status"] in (0, 1), device_lbfgs=info)


def solve(prob, alg: Optional[GRAPE] = None, engine=None):
    """solve(::Problem, ::GRAPE) / solve(::EnsembleProblem, ::GRAPE)."""
    if alg is None:
        raise TypeError("solve(prob) without an algorithm has no integrator in the reference either "
                        "(src/solve.jl:57,66); pass GRAPE(n_slices=...)")
    own = engine is None
    device_opt = getattr(alg, "optimizer", "host") == "device"
    batched = device_opt and not isinstance(alg, ADGRAPE) and not getattr(alg, "devices", None) and \
        alg.optim_options.get("line_search") == "ladder"          # the ladder search probes several step lengths per launch
    eng = engine or make_engine(prob, alg, **({"max_batch": 4} if batched else {}))
    try:
        guess = np.asarray((prob.prob if isinstance(prob, EnsembleProblem) else prob).guess, float)
        if device_opt:
            res = _device_lbfgs(eng, guess, alg.optim_options)
        else:
            res = _lbfgs(lambda x: eng.eval(x), guess, alg.optim_options)
    finally:
        if own:
            eng.close()
    cls = EnsembleSolutionResult if isinstance(prob, EnsembleProblem) else SolutionResult
    return cls(res, res.minimum, res.minimizer, prob, alg)
