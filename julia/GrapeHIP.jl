# GrapeHIP.jl -- the reference-side binding a QuOptimalControl.jl maintainer would add to route
# the GRAPE hot path through libgrape_hip.so (include/grape_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build container and the GPU box have no Julia.  The
# file is kept thin on purpose and mirrors, call for call, quoptimalcontrol.jl_amd/engine.py +
# api.py (which ARE tested, tests/test_gpu_golden_and_api.py), so that behaviour is pinned there.
#
# What it replaces in /root/reference:
#   * the body of the closure `topt` in  solve(::Problem, ::GRAPE)          src/solve.jl:75-100
#                                    and solve(::EnsembleProblem, ::GRAPE)  src/solve.jl:164-196
#     (i.e. the loop over _fom_and_gradient_GRAPE!, src/GRAPE.jl:25-96, and the weighted sums)
#   * init_GRAPE's workspace                                                src/grape_tools.jl:4-16
# What it keeps: Problem / EnsembleProblem / init_ensemble / Optim.LBFGS / the result structs.
#
# Usage:   include("GrapeHIP.jl"); using .GrapeHIP
#          sol = solve(ens_prob, GRAPE_HIP(n_slices = 500))
module GrapeHIP

using QuOptimalControl
using QuOptimalControl: Problem, EnsembleProblem, StateTransfer, UnitaryGate, CoherenceTransfer,
                        SolutionResult, EnsembleSolutionResult, init_ensemble
using Optim
import QuOptimalControl: solve

export GRAPE_HIP, ADGRAPE_HIP, GrapeContext

const libgrape = get(ENV, "LIBGRAPE_HIP", "libgrape_hip.so")

# struct grape_config (include/grape_hip.h) -- field order and types must match exactly
struct GrapeConfig
    sys_type::Int32
    variant::Int32
    n::Int32
    n_controls::Int32
    n_slices::Int32
    n_ensemble::Int32
    duration::Float64
    device::Int32
    flags::Int32
    slices_per_lane::Int32
    waves_per_member::Int32
    expm_squarings::Int32
    max_batch::Int32
    n_state_cols::Int32              # ABI v2: 0 = square states
    n_devices::Int32                 # ABI v2: 0/1 = one GPU; 2..8 = in-library sharding + RCCL all-reduce
    device_ids::NTuple{8,Int32}
    gradient::Int32                  # 0 = reference first-order gradient, 1 = exact (2 <= n <= 64)
    objective::Int32                 # 0 = fom_func, 1 = C1 functional of the ADGRAPE path
end

sys_code(::UnitaryGate) = Int32(0)
sys_code(::StateTransfer) = Int32(1)
sys_code(::CoherenceTransfer) = Int32(2)

"New algorithm tag next to `GRAPE` (src/solve.jl:33-42)."
Base.@kwdef struct GRAPE_HIP{OPTS}
    n_slices::Int
    isinplace::Bool = true           # selects the in-place / static formula variant (sign, sum order)
    device::Int = -1
    devices::Vector{Int} = Int[]     # 2..8 HIP ordinals: the ensemble is sharded over them inside the library
    peer_sum::Bool = false           # GRAPE_FLAG_GROUP_PEER_SUM: sum the shards on devices[1] by peer copies, no RCCL
    optim_options::OPTS = Optim.Options()
end

const GRAPE_FLAG_GROUP_PEER_SUM = Int32(1 << 7)
cfg_flags(alg::GRAPE_HIP) = alg.peer_sum ? GRAPE_FLAG_GROUP_PEER_SUM : Int32(0)
cfg_flags(alg) = Int32(0)

"""
Counterpart of `ADGRAPE` (src/solve.jl:44-52): the functional C1(Xt, U Xi [U']) of src/solve.jl:268-361 with its
EXACT gradient from the device (grape_config.gradient = 1, objective = 1) instead of a Zygote tape.
"""
Base.@kwdef struct ADGRAPE_HIP{OPTS}
    n_slices::Int
    device::Int = -1
    devices::Vector{Int} = Int[]
    optim_options::OPTS = Optim.Options()
end

# per-algorithm settings of grape_config: (variant, gradient, objective)
cfg_mode(alg::GRAPE_HIP) = (alg.isinplace ? Int32(0) : Int32(1), Int32(0), Int32(0))
cfg_mode(::ADGRAPE_HIP) = (Int32(1), Int32(1), Int32(1))      # pw_evolve adds A first (src/timeevolution.jl:32-35)

mutable struct GrapeContext
    handle::Ptr{Cvoid}
    K::Int
    N::Int
    function GrapeContext(members::Vector{<:Problem}, wts::Vector{Float64}, alg)
        p1 = members[1]
        n = size(p1.A, 1)
        m = size(p1.Xi, 2)                 # n x m states (m < n: e.g. a vectorised density matrix, test/liou.jl)
        K, N, E = p1.n_controls, alg.n_slices, length(members)
        nd = length(alg.devices)
        ids = ntuple(i -> i <= nd ? Int32(alg.devices[i]) : Int32(0), 8)
        variant, gradient, objective = cfg_mode(alg)
        cfg = GrapeConfig(sys_code(p1.sys_type), variant, n, K, N, E, Float64(p1.T),
                          nd == 1 ? alg.devices[1] : alg.device, cfg_flags(alg), 0, 0, -1, 0, m == n ? 0 : m, nd > 1 ? nd : 0, ids,
                          gradient, objective)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:grape_create, libgrape), Cint, (Ref{GrapeConfig}, Ref{Ptr{Cvoid}}), cfg, h)
        rc == 0 || error("grape_create: ", unsafe_string(ccall((:grape_last_error, libgrape), Cstring, (Ptr{Cvoid},), C_NULL)))
        ctx = new(h[], K, N)
        finalizer(c -> ccall((:grape_destroy, libgrape), Cint, (Ptr{Cvoid},), c.handle), ctx)
        # pack what init_ensemble produced (src/tools.jl:42-53) into contiguous column-major arrays
        A  = Array{ComplexF64}(undef, n, n, E)
        B  = Array{ComplexF64}(undef, n, n, K, E)
        Xi = Array{ComplexF64}(undef, n, m, E)
        Xt = Array{ComplexF64}(undef, n, m, E)
        for (k, p) in enumerate(members)
            A[:, :, k] .= p.A
            for j in 1:K
                B[:, :, j, k] .= p.B[j]
            end
            Xi[:, :, k] .= p.Xi
            Xt[:, :, k] .= p.Xt
        end
        check(ctx, ccall((:grape_set_operators, libgrape), Cint,
                         (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{Float64}),
                         ctx.handle, A, B, Xi, Xt, wts))
        ctx
    end
end

check(ctx::GrapeContext, rc) =
    rc == 0 || error("libgrape_hip: ", unsafe_string(ccall((:grape_last_error, libgrape), Cstring, (Ptr{Cvoid},), ctx.handle)))

"One call of the reference's closure body: returns F, fills G in place (either may be `nothing`)."
function fom_and_gradient!(ctx::GrapeContext, G, x::Matrix{Float64}; want_F = true)
    size(x) == (ctx.K, ctx.N) || throw(DimensionMismatch("x must be (n_controls, n_slices)"))
    F = Ref{Float64}(NaN)
    GC.@preserve x G begin
        gptr = G === nothing ? Ptr{Float64}(C_NULL) : pointer(G)
        fptr = want_F ? Base.unsafe_convert(Ptr{Float64}, F) : Ptr{Float64}(C_NULL)
        check(ctx, ccall((:grape_eval, libgrape), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                         ctx.handle, x, fptr, gptr))
    end
    F[]
end

"The kernels the last evaluation launched, in launch order (grape_get_kernel_names, ABI v4)."
function kernel_names(ctx::GrapeContext)
    need = ccall((:grape_get_kernel_names, libgrape), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), ctx.handle, C_NULL, 0)
    need > 0 || return String[]
    buf = Vector{UInt8}(undef, need)
    ccall((:grape_get_kernel_names, libgrape), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), ctx.handle, buf, need)
    filter(!isempty, split(unsafe_string(pointer(buf)), ';'))
end

"Accepted step length and cumulative evaluation count of every iteration of the last `grape_lbfgs` run (ABI v5)."
function lbfgs_trace(ctx::GrapeContext)
    n = Ref{Int32}(0)
    check(ctx, ccall((:grape_lbfgs_get_trace, libgrape), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Int32, Ptr{Int32}),
                     ctx.handle, C_NULL, C_NULL, 0, n))
    alphas, evals = Vector{Float64}(undef, n[]), Vector{Int32}(undef, n[])
    check(ctx, ccall((:grape_lbfgs_get_trace, libgrape), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Int32, Ptr{Int32}),
                     ctx.handle, alphas, evals, n[], n))
    alphas, evals
end

"""
One process per GPU without librccl (ABI v4): `allgather` is any function that returns every rank's 64 bytes in rank
order as one `Vector{UInt8}` (e.g. `h -> MPI.Allgather(h, comm)`).  Afterwards `fom_and_gradient!` on every rank returns
the full-ensemble F and G: the ranks' rows meet in mailboxes in device memory (src/solve.jl:171-191's sum).
"""
function attach_ipc!(ctx::GrapeContext, rank::Integer, nranks::Integer, allgather)
    h = Vector{UInt8}(undef, 64)
    check(ctx, ccall((:grape_ipc_export, libgrape), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt8}), ctx.handle, nranks, h))
    all = allgather(h)
    length(all) == 64 * nranks || error("attach_ipc!: allgather must return 64 * nranks bytes")
    check(ctx, ccall((:grape_ipc_attach, libgrape), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), ctx.handle, all, rank, nranks))
    ctx
end

# the (F, G, x) protocol of Optim.only_fg!, as in src/solve.jl:94-99 and :189-195
function make_topt(ctx::GrapeContext)
    (F, G, x) -> begin
        fom = fom_and_gradient!(ctx, G, x; want_F = F !== nothing)
        F !== nothing ? fom : nothing
    end
end

function solve(prob::Problem, alg::Union{GRAPE_HIP,ADGRAPE_HIP})
    ctx = GrapeContext([prob], [1.0], alg)
    res = Optim.optimize(Optim.only_fg!(make_topt(ctx)), prob.guess, Optim.LBFGS(), alg.optim_options)   # src/solve.jl:138
    SolutionResult(res, res.minimum, res.minimizer, prob, alg)                                            # src/solve.jl:139
end

function solve(ens::EnsembleProblem, alg::Union{GRAPE_HIP,ADGRAPE_HIP})
    members = init_ensemble(ens)                                                                          # src/solve.jl:150
    ctx = GrapeContext(members, Vector{Float64}(ens.wts), alg)
    res = Optim.optimize(Optim.only_fg!(make_topt(ctx)), members[1].guess, Optim.LBFGS(), alg.optim_options)  # :244
    EnsembleSolutionResult(res, res.minimum, res.minimizer, ens, alg)                                     # :245
end

end # module
