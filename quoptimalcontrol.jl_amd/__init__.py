"""quoptimalcontrol.jl_amd -- MI355X-native GRAPE propagator/gradient engine.

Host-side mirror of QuOptimalControl.jl's Problem / EnsembleProblem / GRAPE / solve()
surface over the C-ABI library libgrape_hip.so (csrc/, include/grape_hip.h).  Only the
hot path named in BASELINE.json:north_star lives here; see DESIGN.md.

The directory name contains a dot (it is the project's name), so import it through the
root-level shim:  `import quoptimalcontrol_jl_amd as qoc`.
"""
from . import workloads  # noqa: F401
from .engine import GrapeEngine, GrapeError, library_path, load_library  # noqa: F401
from .api import (  # noqa: F401
    ADGRAPE, GRAPE, CoherenceTransfer, EnsembleProblem, EnsembleSolutionResult, Problem, SolutionResult,
    StateTransfer, UnitaryGate, C1, init_ensemble, solve, fom_and_gradient, pulse_to_file, pulse_from_file, save, load,
)

__all__ = [
    "workloads", "GrapeEngine", "GrapeError", "library_path", "load_library", "GRAPE", "ADGRAPE",
    "CoherenceTransfer", "EnsembleProblem", "EnsembleSolutionResult", "Problem", "SolutionResult",
    "StateTransfer", "UnitaryGate", "C1", "init_ensemble", "solve", "fom_and_gradient", "pulse_to_file", "pulse_from_file", "save", "load",
]
