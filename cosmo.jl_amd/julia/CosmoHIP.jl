# CosmoHIP.jl -- the `ccall` layer that drops libcosmo_hip.so under the unchanged COSMO.jl front-end.
#
# NOT EXECUTED in the build container (Julia is not installed there); it is kept deliberately thin and mechanical: every
# function is a one-to-one wrapper of an entry point of include/cosmo_hip.h, whose semantics are exercised through the
# identical C ABI by the Python ctypes binding (cosmo.jl_amd/_ffi.py) in tests/.  See INTEGRATION.md.
#
# Three plugin surfaces of the reference are served (SURVEY.md 8b):
#   1. AbstractKKTSolver      : HipCGKKTSolver / HipMINRESKKTSolver      (fine-grained; one host<->device trip per solve!)
#   2. AbstractConvexSet      : HipProjection wraps the composite projection (fine-grained)
#   3. COSMO.optimize!(model) : optimize_hip!(model) keeps chordal_decomposition!/setup!/the epilogue in Julia and runs the
#                               `while` loop (src/solver.jl:137-176) device-resident.  This is the performance path.
module CosmoHIP

using COSMO, SparseArrays, LinearAlgebra
import COSMO: AbstractKKTSolver, solve!, update_rho!, free_memory!

# Two builds of the SAME sources and the SAME symbol names (include/cosmo_hip.h: cosmo_hip_real): libcosmo_hip.so for
# COSMO.Model{Float64}, libcosmo_hip_f32.so (-DCOSMO_HIP_REAL_FLOAT) for COSMO.Model{Float32} (src/types.jl:348).  Every data array
# crosses the ABI as Ptr{T}; scalars (settings, residuals, times) are Cdouble in both.
const LIB = Ref{String}(joinpath(@__DIR__, "..", "libcosmo_hip.so"))
const LIB32 = Ref{String}(joinpath(@__DIR__, "..", "libcosmo_hip_f32.so"))
const HipFloat = Union{Float32, Float64}
libpath(::Type{Float64}) = LIB[]
libpath(::Type{Float32}) = LIB32[]

# ---- mirrors of the ABI structs: GENERATED from include/cosmo_hip.h by tools/gen_abi_structs.py (Params, AccelParams, ResultC, MAX_RHO_UPDATES)
include("abi_structs.jl")

const KKT_CG, KKT_MINRES_REDUCED, KKT_MINRES, KKT_CG_SR, KKT_CG_JACOBI = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)   # KKT_CG_SR: opt-in single-reduction CG; KKT_CG_JACOBI: opt-in Jacobi-preconditioned CG (assembled operator only)
const STATUS = (:Undetermined, :Solved, :Max_iter_reached, :Unsolved, :Primal_infeasible, :Dual_infeasible, :Time_limit_reached)

# the struct mirrors of abi_structs.jl were generated for ABI_VERSION: a library that reports another version would read / write them with a
# different layout (a stale .so reads garbage into safeguarding_iter; a newer one writes past the caller's ResultC)
function check_abi(::Type{T}) where {T <: HipFloat}
    v = ccall((:cosmo_hip_version, libpath(T)), Int32, ())
    v == ABI_VERSION || error("$(libpath(T)) reports ABI version $v, CosmoHIP.jl was generated for $ABI_VERSION: rebuild the library or regenerate abi_structs.jl")
    return nothing
end

mutable struct Handle{T <: HipFloat}
    ptr::Ptr{Cvoid}
    function Handle{T}(device::Integer = 0) where {T <: HipFloat}
        check_abi(T)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:cosmo_hip_create, libpath(T)), Int32, (Ref{Ptr{Cvoid}}, Int32), ref, device)
        rc == 0 || error("cosmo_hip_create failed with code $rc (no MI355X visible?)")
        h = new{T}(ref[])
        finalizer(destroy!, h)      # free_memory! / GC both end here; cosmo_hip_destroy is idempotent
        return h
    end
end

Handle(device::Integer = 0) = Handle{Float64}(device)
lib(::Handle{T}) where {T} = libpath(T)

function destroy!(h::Handle)
    h.ptr == C_NULL && return
    ccall((:cosmo_hip_destroy, lib(h)), Int32, (Ptr{Cvoid},), h.ptr)
    h.ptr = C_NULL
    nothing
end

# the reference signals errors with exceptions (e.g. src/linear_solver/kktsolver.jl:304); so does the glue
function check(h::Handle, rc::Int32)
    rc == 0 && return
    msg = unsafe_string(ccall((:cosmo_hip_last_error, lib(h)), Cstring, (Ptr{Cvoid},), h.ptr))
    error("libcosmo_hip error $rc: $msg")
end

cone_type(::COSMO.ZeroSet) = Int32(0)
cone_type(::COSMO.Nonnegatives) = Int32(1)
cone_type(::COSMO.Box) = Int32(2)
cone_type(::COSMO.SecondOrderCone) = Int32(3)
cone_type(::COSMO.PsdCone) = Int32(4)
cone_type(::COSMO.PsdConeTriangle{T, T}) where {T} = Int32(5)
cone_type(::COSMO.PsdConeTriangle{T, Complex{T}}) where {T} = Int32(10)
cone_type(::COSMO.ExponentialCone) = Int32(6)
cone_type(::COSMO.DualExponentialCone) = Int32(7)
cone_type(::COSMO.PowerCone) = Int32(8)
cone_type(::COSMO.DualPowerCone) = Int32(9)
cone_param(s::COSMO.PowerCone) = s.α
cone_param(s::COSMO.DualPowerCone) = s.primal_cone.α
cone_param(s) = 0.0
# any other subtype of AbstractConvexCone is a user-defined cone (docs/src/literate/custom_cone.jl): its own project! /
# in_dual / in_pol_recc methods run on the host, called back by the library on this task's thread (COSMO_HIP_CUSTOM)
cone_type(::COSMO.AbstractConvexCone) = Int32(11)
cone_type(C) = error("set type $(typeof(C)) is outside the MI355X hot path (SURVEY.md 8a)")

# C-callable thunks of the three generic functions; `user` is a pointer to a Ref{Any} holding the cone object
function _custom_project(x::Ptr{T}, dim::Int64, user::Ptr{Cvoid})::Cvoid where {T <: HipFloat}
    cone = unsafe_pointer_to_objref(user)[]
    COSMO.project!(unsafe_wrap(Array, x, dim), cone)
    return
end
function _custom_in_dual(x::Ptr{T}, dim::Int64, tol::Cdouble, user::Ptr{Cvoid})::Int32 where {T <: HipFloat}
    cone = unsafe_pointer_to_objref(user)[]
    return Int32(COSMO.in_dual(unsafe_wrap(Array, x, dim), cone, T(tol)))
end
function _custom_in_pol_recc(x::Ptr{T}, dim::Int64, tol::Cdouble, user::Ptr{Cvoid})::Int32 where {T <: HipFloat}
    cone = unsafe_pointer_to_objref(user)[]
    return Int32(COSMO.in_pol_recc(unsafe_wrap(Array, x, dim), cone, T(tol)))
end

# install the callbacks of every user cone; the returned Refs must stay alive as long as the handle is used (GC.@preserve)
function set_custom_cones!(h::Handle{T}, C::COSMO.CompositeConvexSet{T}) where {T <: HipFloat}
    keep = Any[]
    for (k, s) in enumerate(C.sets)
        cone_type(s) == Int32(11) || continue
        r = Ref{Any}(s); push!(keep, r)
        S = typeof(s)
        fproj = @cfunction(_custom_project, Cvoid, (Ptr{T}, Int64, Ptr{Cvoid}))
        fdual = hasmethod(COSMO.in_dual, Tuple{Vector{T}, S, T}) ? @cfunction(_custom_in_dual, Int32, (Ptr{T}, Int64, Cdouble, Ptr{Cvoid})) : C_NULL
        frecc = hasmethod(COSMO.in_pol_recc, Tuple{Vector{T}, S, T}) ? @cfunction(_custom_in_pol_recc, Int32, (Ptr{T}, Int64, Cdouble, Ptr{Cvoid})) : C_NULL
        check(h, ccall((:cosmo_hip_set_custom_cone, lib(h)), Int32, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
            h.ptr, k - 1, fproj, fdual, frecc, pointer_from_objref(r)))
    end
    return keep
end

# cosmo_hip_set_problem takes SparseMatrixCSC{T,Int64} untouched: colptr / rowval are already 1-based Int64
function set_problem!(h::Handle{T}, P::SparseMatrixCSC{T, Int64}, A::SparseMatrixCSC{T, Int64}, q::Vector{T}, b::Vector{T}) where {T <: HipFloat}
    m, n = size(A)
    GC.@preserve P A q b begin
        check(h, ccall((:cosmo_hip_set_problem, lib(h)), Int32,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{T}, Ptr{Int64}, Ptr{Int64}, Ptr{T}, Ptr{T}, Ptr{T}),
            h.ptr, n, m, P.colptr, P.rowval, P.nzval, A.colptr, A.rowval, A.nzval, q, b))
    end
end

function set_cones!(h::Handle{T}, C::COSMO.CompositeConvexSet{T}) where {T <: HipFloat}
    types = Int32[cone_type(s) for s in C.sets]
    dims = Int64[s.dim for s in C.sets]
    bl = T[]; bu = T[]
    for s in C.sets
        if s isa COSMO.Box
            append!(bl, s.l); append!(bu, s.u)          # already E-scaled by scale!(::Box) (src/convexset.jl:863-867)
        end
    end
    params = T[cone_param(s) for s in C.sets]            # alpha of the power cones (src/convexset.jl:607-618)
    GC.@preserve types dims bl bu params begin
        check(h, ccall((:cosmo_hip_set_cones_ex, lib(h)), Int32, (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int64}, Ptr{T}, Ptr{T}, Ptr{T}),
            h.ptr, length(types), types, dims, isempty(bl) ? C_NULL : pointer(bl), isempty(bu) ? C_NULL : pointer(bu), params))
    end
end

# scale_ruiz! (src/scaling.jl:21-116) on the device-resident UNSCALED problem; fills ws.sm so that the unchanged
# reverse_scaling! / update! keep working.  Call between set_cones! and set_params!.
function scale_ruiz!(h::Handle{T}, ws::COSMO.Workspace{T}) where {T <: HipFloat}
    s = ws.settings
    D = ws.sm.D.diag; E = ws.sm.E.diag; c = Ref{Cdouble}(1.0)
    GC.@preserve D E check(h, ccall((:cosmo_hip_scale_ruiz, lib(h)), Int32, (Ptr{Cvoid}, Int64, Cdouble, Cdouble, Ptr{T}, Ptr{T}, Ref{Cdouble}),
        h.ptr, s.scaling, s.MIN_SCALING, s.MAX_SCALING, D, E, c))
    ws.sm.Dinv.diag .= one(T) ./ D; ws.sm.Einv.diag .= one(T) ./ E
    ws.sm.c[] = T(c[]); ws.sm.cinv[] = one(T) / T(c[])
    nothing
end


# settings.accelerator is an OptionsFactory{<:AbstractAccelerator} (src/settings.jl:96,136,148-150).  The device builds the
# reference's default AndersonAccelerator{T, Type2{QRDecomp}, RestartedMemory, NoRegularizer} (kind 1) and, since round 6, the variants of
# docs/src/acceleration.md:23-26 with broyden type Type1 / Type2{NormalEquations} and RestartedMemory / RollingMemory (kinds 2 .. 5 of
# include/cosmo_hip.h); EmptyAccelerator maps to "none"; anything else (a regulariser, Type2{QRDecomp} with a rolling memory) is rejected
# (error) rather than silently replaced.
function accel_kind(::Type{AT}, ::Type{T}) where {AT, T}
    CA = COSMO.CA                                   # const CA = COSMOAccelerators (src/accelerator_interface.jl:3)
    AT == CA.AndersonAccelerator{T, CA.Type2{CA.QRDecomp}, CA.RestartedMemory, CA.NoRegularizer} && return Int32(1)
    AT == CA.AndersonAccelerator{T, CA.Type1, CA.RestartedMemory, CA.NoRegularizer} && return Int32(2)
    AT == CA.AndersonAccelerator{T, CA.Type1, CA.RollingMemory, CA.NoRegularizer} && return Int32(3)
    AT == CA.AndersonAccelerator{T, CA.Type2{CA.NormalEquations}, CA.RestartedMemory, CA.NoRegularizer} && return Int32(4)
    AT == CA.AndersonAccelerator{T, CA.Type2{CA.NormalEquations}, CA.RollingMemory, CA.NoRegularizer} && return Int32(5)
    error("accelerator $(AT) is not built on the MI355X path; use AndersonAccelerator{T, BT, MT, NoRegularizer} with BT in (Type2{QRDecomp} [RestartedMemory only], Type1, Type2{NormalEquations}) or EmptyAccelerator")
end
function accel_params_from(settings::COSMO.Settings{T}) where {T <: HipFloat}
    AT = settings.accelerator.ObjectType
    AT <: COSMO.EmptyAccelerator && return nothing
    kind = accel_kind(AT, T)
    kw = settings.accelerator.kwargs
    act = get(kw, :activation_reason, COSMO.ImmediateActivation())
    start = act isa COSMO.IterActivation ? act.start_iter : 2
    acc = act isa COSMO.AccuracyActivation ? Float64(act.start_accuracy) : -1.0    # src/accelerator_interface.jl:14-21
    return AccelParams(kind, Int32(get(kw, :mem, 10)), Int32(get(kw, :min_mem, 3)), Int32(settings.safeguard ? 1 : 0), Int64(start),
                       Float64(settings.safeguard_tol), 1e4, acc)
end
function set_accelerator!(h::Handle{T}, settings::COSMO.Settings{T}) where {T <: HipFloat}
    p = accel_params_from(settings)
    p === nothing && return nothing
    check(h, ccall((:cosmo_hip_set_accelerator, lib(h)), Int32, (Ptr{Cvoid}, Ref{AccelParams}), h.ptr, Ref(p)))
    nothing
end

# the settings struct of the ABI is Float64 in both libraries: a Float32 setting converts exactly (Params' constructor converts)
function params_from(settings::COSMO.Settings{T}, kkt_kind::Int32; tol_constant = 1.0, tol_exponent = 1.5, setup_time = 0.0) where {T <: HipFloat}
    s = settings
    Params(s.sigma, s.alpha, s.rho, s.eps_abs, s.eps_rel, s.eps_prim_inf, s.eps_dual_inf, tol_constant, tol_exponent,
           s.RHO_MIN, s.RHO_MAX, s.RHO_TOL, s.RHO_EQ_OVER_RHO_INEQ, s.adaptive_rho_tolerance, s.COSMO_INFTY * s.MIN_SCALING,
           s.time_limit, s.max_iter, min(s.adaptive_rho_max_adaptions, typemax(Int64) >> 1), kkt_kind, s.check_termination,
           s.check_infeasibility, s.adaptive_rho ? 1 : 0, s.adaptive_rho_interval, s.scaling != 0 ? 1 : 0, s.obj_true, s.obj_true_tol,
           s.adaptive_rho_fraction, Float64(setup_time))      # setup_time: optimize_hip! hands it over through cosmo_hip_set_setup_time once setup! has ended; optimize_hip_batch! puts its measured set-up phase here
end

function set_params!(h::Handle{T}, p::Params, rho_vec::Union{Vector{T}, Nothing}) where {T <: HipFloat}
    GC.@preserve rho_vec begin
        check(h, ccall((:cosmo_hip_set_params, lib(h)), Int32, (Ptr{Cvoid}, Ref{Params}, Ptr{T}), h.ptr, Ref(p),
            rho_vec === nothing ? C_NULL : pointer(rho_vec)))
    end
end

# ---------------------------------------------------------------------------------------------------------------------
# 1. AbstractKKTSolver plugin (src/linear_solver/kktsolver.jl:5-11).  Use as
#       settings = COSMO.Settings(kkt_solver = with_options(CosmoHIP.HipCGKKTSolver, device = 0))
#    The constructor receives the scaled P, A, sigma and ws.ρvec from _make_kkt_solver! (src/setup.jl:1-7).
# ---------------------------------------------------------------------------------------------------------------------
mutable struct HipKKTSolver{T <: HipFloat} <: AbstractKKTSolver
    h::Handle{T}
    m::Int; n::Int
    function HipKKTSolver(P::SparseMatrixCSC{T, Int64}, A::SparseMatrixCSC{T, Int64}, sigma::T, rho;
                          kind::Int32 = KKT_CG, device::Integer = 0, tol_constant = 1.0, tol_exponent = 1.5) where {T <: HipFloat}
        m, n = size(A)
        h = Handle{T}(device)
        set_problem!(h, P, A, zeros(T, n), zeros(T, m))    # q, b are not needed by solve!
        types = Int32[1]; dims = Int64[m]                    # cones are irrelevant for solve!: declare one Nonnegatives(m)
        GC.@preserve types dims check(h, ccall((:cosmo_hip_set_cones, lib(h)), Int32,
            (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int64}, Ptr{T}, Ptr{T}), h.ptr, 1, types, dims, C_NULL, C_NULL))
        p = params_from(COSMO.Settings{T}(sigma = sigma), kind; tol_constant = tol_constant, tol_exponent = tol_exponent)
        rv = isa(rho, Number) ? fill(T(rho), m) : Vector{T}(rho)
        set_params!(h, p, rv)
        new{T}(h, m, n)
    end
end
HipCGKKTSolver(P, A, sigma, rho; kwargs...) = HipKKTSolver(P, A, sigma, rho; kind = KKT_CG, kwargs...)
HipMINRESKKTSolver(P, A, sigma, rho; kwargs...) = HipKKTSolver(P, A, sigma, rho; kind = KKT_MINRES, kwargs...)
# OPT-IN, no reference counterpart (COSMO calls cg! without a preconditioner, src/linear_solver/kktsolver_indirect.jl:70): IterativeSolvers' preconditioned
# recurrence with Pl = Diagonal(diag(P + sigma I + A' rho A)); set_params! fails where the reduced operator cannot be assembled
HipCGJacobiKKTSolver(P, A, sigma, rho; kwargs...) = HipKKTSolver(P, A, sigma, rho; kind = KKT_CG_JACOBI, kwargs...)

# called from admm_x! (src/solver.jl:52): lhs = ws.sol, rhs = ws.ls, both length n+m and caller owned
function solve!(S::HipKKTSolver{T}, lhs::AbstractVector{T}, rhs::AbstractVector{T}) where {T <: HipFloat}
    GC.@preserve lhs rhs check(S.h, ccall((:cosmo_hip_kkt_solve, lib(S.h)), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{Int64}),
        S.h.ptr, pointer(lhs), pointer(rhs), C_NULL))
    return lhs
end
# called from update_rho_vec! (src/parameters.jl:85-89)
function update_rho!(S::HipKKTSolver{T}, rho::Vector{T}) where {T <: HipFloat}
    GC.@preserve rho check(S.h, ccall((:cosmo_hip_update_rho, lib(S.h)), Int32, (Ptr{Cvoid}, Ptr{T}), S.h.ptr, rho))
end
# called from optimize! exit (src/solver.jl:200,206-208)
free_memory!(S::HipKKTSolver) = destroy!(S.h)

# ---------------------------------------------------------------------------------------------------------------------
# 2. projection plugin: project!(s::SplitVector, C::CompositeConvexSet) (src/convexset.jl:885-891) on the device
# ---------------------------------------------------------------------------------------------------------------------
# diagnostics of the matrix-sign PSD path and of the opt-in single-launch CG (cosmo_hip_polar_stats / cosmo_hip_cg_persist_stats)
function polar_stats(h::Handle)
    out = zeros(Int64, 16)
    check(h, ccall((:cosmo_hip_polar_stats, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, out))
    return (large_cones = out[1], batch_cones = out[2], tile_side = out[3], k_split = out[4], products_last_large = out[9], fallback_rounds = out[10],
            verified = out[11], products_last_batch = out[12], schedule_steps = out[13], unverified = out[14], projections = out[15], err_max = out[16] * 1e-18)
end
function cg_persist_stats(h::Handle)
    out = zeros(Int64, 8)
    check(h, ccall((:cosmo_hip_cg_persist_stats, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, out))
    return (enabled = out[1] != 0, workgroups = out[2], launches = out[3], fallbacks = out[4])
end

function polar_dataflow_stats(h::Handle)   # the batch's main schedule as one persistent dependency-driven launch (cosmo_hip_polar_dataflow_stats)
    out = zeros(Float64, 8)
    check(h, ccall((:cosmo_hip_polar_dataflow_stats, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Cdouble}), h.ptr, out))
    return (enabled = Int(out[1]), launches = Int(out[2]), products_per_launch = Int(out[3]), timed_launches = Int(out[4]), avg_launch_seconds = out[5],
            flops_per_launch = out[6], workgroups = Int(out[7]), tiles_per_product = Int(out[8]))
end
polar_dataflow_reset_timing(h::Handle) = check(h, ccall((:cosmo_hip_polar_dataflow_reset_timing, lib(h)), Int32, (Ptr{Cvoid},), h.ptr))
function polar_depth_stats(h::Handle)      # per-cone lifting depth of the sign iteration (opt-in COSMO_HIP_POLAR_ADAPT=1)
    out = zeros(Int64, 8)
    check(h, ccall((:cosmo_hip_polar_depth_stats, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, out))
    return (adaptive = out[1] != 0, depth_min = out[2], depth_max = out[3], depth_mean = out[4] / 1000, weighted_products = out[5] / 1000,
            failed_verifications = out[6], downward_probes = out[7], projections = out[8])
end
function time_krylov(h::Handle, reps::Integer)      # measurement hook: seconds per Krylov iteration incl. kernel boundaries, algorithmic bytes, launches
    t = Ref{Cdouble}(0.0); b = Ref{Cdouble}(0.0); nl = Ref{Int32}(0)
    check(h, ccall((:cosmo_hip_time_krylov, lib(h)), Int32, (Ptr{Cvoid}, Int32, Ref{Cdouble}, Ref{Cdouble}, Ref{Int32}), h.ptr, Int32(reps), t, b, nl))
    return (seconds = t[], bytes = b[], launches = nl[])
end

kkt_recurrence(h::Handle) = unsafe_string(ccall((:cosmo_hip_kkt_recurrence, lib(h)), Cstring, (Ptr{Cvoid},), h.ptr))   # which Krylov recurrence / kernels the handle's KKT solves run
function fold_stats(h::Handle)      # assembled reduced CG operator (cosmo_hip_fold_stats)
    out = zeros(Int64, 6)
    check(h, ccall((:cosmo_hip_fold_stats, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, out))
    return (enabled = out[1] != 0, nnz = out[2], terms = out[3], tiles = out[4], factored_rows = out[5], stored_entries = out[6])
end
function polar_streamk_stats(h::Handle)      # stream-K product of the large PSD cones (cosmo_hip_polar_streamk_stats)
    out = zeros(Int64, 4)
    check(h, ccall((:cosmo_hip_polar_streamk_stats, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, out))
    return (enabled = out[1], workgroups = out[2], classes = out[3], timeouts = out[4])
end

# ---- sharding over the GPUs of one node (one Julia process per GPU; the unique id travels through Distributed / MPI) -----------------
# comm_unique_id() on rank 0, comm_init!(h, rank, nranks, id) on every rank, then EITHER set_cone_shard! (projections only: one all-gather
# of the projected s per iteration) OR set_row_shard! (the rank keeps its cones AND their rows of A / s / mu / rho: one all-reduce of an
# n-vector per iteration, src/linear_solver/kktsolver_indirect.jl:52-54).  first_cone: nranks + 1 zero-based cone boundaries.
function comm_unique_id(::Type{T} = Float64) where {T <: HipFloat}
    id = zeros(UInt8, 128)
    rc = ccall((:cosmo_hip_comm_unique_id, T === Float32 ? LIB32[] : LIB[]), Int32, (Ptr{UInt8},), id)
    rc == 0 || error("cosmo_hip_comm_unique_id failed ($rc)")
    return id
end
comm_init!(h::Handle, rank::Integer, nranks::Integer, id::Vector{UInt8}) =
    check(h, ccall((:cosmo_hip_comm_init, lib(h)), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), h.ptr, Int32(rank), Int32(nranks), id))
set_cone_shard!(h::Handle, first_cone::Vector{Int64}) =
    check(h, ccall((:cosmo_hip_set_cone_shard, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, first_cone))
set_row_shard!(h::Handle, first_cone::Vector{Int64}) =
    check(h, ccall((:cosmo_hip_set_row_shard, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, first_cone))
# known-answer all-reduce of `count` reals through the loop's exchange path (collective: every rank calls it); compare `hash` across the ranks
function comm_allreduce_check(h::Handle, count::Integer)
    out = zeros(Int64, 6)
    check(h, ccall((:cosmo_hip_comm_allreduce_check, lib(h)), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}), h.ptr, Int64(count), out))
    return (exact_mismatches = out[1], inexact_outside_bound = out[2], hash = out[3], transport = out[4], nranks = out[5], rccl_version_code = out[6])
end
function comm_stats(h::Handle)
    out = zeros(Int64, 8)
    check(h, ccall((:cosmo_hip_comm_stats_ex, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, out))
    return (nranks = out[1], rank = out[2], collectives = out[3], transport = out[4], mode = out[5], bytes = out[6], allreduces = out[7], allreduce_elems = out[8])
end

function project_hip!(h::Handle{T}, s::COSMO.SplitVector{T}) where {T <: HipFloat}
    d = s.data
    GC.@preserve d check(h, ccall((:cosmo_hip_project, lib(h)), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{Int64}, Ptr{Int32}), h.ptr, d, C_NULL, C_NULL))
    return nothing
end

# ---------------------------------------------------------------------------------------------------------------------
# 3. the coarse path: COSMO.optimize! with the while-loop on the MI355X.  Everything outside src/solver.jl:128-176 is the
#    reference's own code, called unchanged.
# ---------------------------------------------------------------------------------------------------------------------
# psd_projection = :sign (default: verified matrix-sign iteration above side 16) or :eigen (the eigendecomposition-based projection of
# src/convexset.jl:163-189, 243-263 by Jacobi eigensolvers at every side: exact nnz_lambda, several times slower) -- cosmo_hip_set_psd_projection
function optimize_hip!(ws::COSMO.Workspace{T}; device::Integer = 0, kkt_kind::Int32 = KKT_CG, tol_constant = 1.0, tol_exponent = 1.5, psd_projection::Symbol = :sign) where {T <: HipFloat}
    !ws.states.IS_ASSEMBLED && throw(ErrorException("The model has to be assembled! / set! before optimize!() can be called."))
    solver_time_start = time()
    settings = ws.settings
    if settings.decompose                                                     # src/solver.jl:88-94
        if !ws.states.IS_CHORDAL_DECOMPOSED
            ws.times.graph_time = @elapsed COSMO.chordal_decomposition!(ws)
        elseif ws.ci.decompose
            COSMO.pre_allocate_variables!(ws)
        end
    end
    if !ws.states.IS_SCALED                                                   # :99-101
        ws.sm = (settings.scaling > 0) ? COSMO.ScaleMatrices{T}(ws.p.model_size[1], ws.p.model_size[2]) : COSMO.ScaleMatrices{T}()
    end
    # setup! without the CPU KKT factorisation: scaling, row ranges, classification, rho vector (src/setup.jl:18-42)
    settings_nokkt = settings
    ws.times.setup_time = @elapsed begin
        COSMO.allocate_set_memory!(ws)
        if settings.scaling != 0 && !ws.states.IS_SCALED
            COSMO.scale_ruiz!(ws); ws.states.IS_SCALED = true
        else
            COSMO.scale_variables!(ws.vars.x, ws.vars.μ, ws.vars.s, ws.sm.Dinv, ws.sm.Einv, ws.sm.E, ws.sm.c)
        end
        ws.row_ranges = COSMO.get_set_indices(ws.p.C.sets)
        COSMO.classify_constraints!(ws)
        !ws.states.IS_OPTIMIZED && COSMO.set_rho_vec!(ws)
    end
    m, n = ws.p.model_size
    h = Handle{T}(device)
    set_problem!(h, SparseMatrixCSC(ws.p.P), SparseMatrixCSC(ws.p.A), ws.p.q, Vector(ws.p.b))
    set_cones!(h, ws.p.C)
    custom_refs = set_custom_cones!(h, ws.p.C)                                 # user cones: callbacks into their project! methods
    psd_projection in (:sign, :eigen) || error("psd_projection: :sign or :eigen")
    psd_projection == :eigen && check(h, ccall((:cosmo_hip_set_psd_projection, lib(h)), Int32, (Ptr{Cvoid}, Int32), h.ptr, Int32(1)))
    set_params!(h, params_from(settings, kkt_kind; tol_constant = tol_constant, tol_exponent = tol_exponent), ws.ρvec)
    sc = settings.scaling != 0
    D = sc ? ws.sm.D.diag : ones(T, n); Dinv = sc ? ws.sm.Dinv.diag : ones(T, n); E = sc ? ws.sm.E.diag : ones(T, m); Einv = sc ? ws.sm.Einv.diag : ones(T, m)
    GC.@preserve D Dinv E Einv check(h, ccall((:cosmo_hip_set_scaling_full, lib(h)), Int32,
        (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Cdouble, Cdouble), h.ptr, D, Dinv, E, Einv, ws.sm.c[], ws.sm.cinv[]))
    set_accelerator!(h, settings)                                              # _make_accelerator! (src/setup.jl:10-16,44-49)
    x0 = ws.vars.x; s0 = ws.vars.s.data; mu0 = ws.vars.μ
    GC.@preserve x0 s0 mu0 check(h, ccall((:cosmo_hip_set_iterates, lib(h)), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}),
        h.ptr, x0, s0, mu0))                                                  # src/solver.jl:128-129
    ws.states.IS_OPTIMIZED = true
    auto_rho = settings.adaptive_rho && settings.adaptive_rho_interval == 0       # the automatic interval (src/solver.jl:244-256) is measured against setup_time
    auto_rho && check(h, ccall((:cosmo_hip_set_setup_time, lib(h)), Int32, (Ptr{Cvoid}, Cdouble), h.ptr, Float64(ws.times.setup_time)))
    res = Ref{ResultC}()
    GC.@preserve custom_refs check(h, ccall((:cosmo_hip_optimize, lib(h)), Int32, (Ptr{Cvoid}, Ref{ResultC}), h.ptr, res))   # src/solver.jl:137-176
    r = res[]
    if auto_rho                                                                    # the reference writes the chosen interval into the settings (:249-254)
        ri = zeros(Int64, 2)
        check(h, ccall((:cosmo_hip_get_rho_interval, lib(h)), Int32, (Ptr{Cvoid}, Ptr{Int64}), h.ptr, ri))
        settings.adaptive_rho_interval = Int(ri[1])
    end
    w = ws.vars.w; wp = ws.vars.w_prev; sd = ws.vars.s.data; mu = ws.vars.μ  # x is a view of w_prev (src/types.jl:274)
    GC.@preserve w wp sd mu check(h, ccall((:cosmo_hip_get_iterates, lib(h)), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
        h.ptr, w, wp, sd, mu))
    ws.ρ = T(r.rho)
    resize!(ws.rho_updates, 0); append!(ws.rho_updates, T.(collect(r.rho_updates)[1:min(r.n_rho_updates, MAX_RHO_UPDATES)]))
    ws.times.iter_time = r.iter_time; ws.times.proj_time = r.proj_time
    status = STATUS[r.status + 1]
    res_info = COSMO.ResultInfo(T(r.r_prim), T(r.r_dual), T(r.max_norm_prim), T(r.max_norm_dual), ws.rho_updates)
    settings.scaling != 0 && COSMO.reverse_scaling!(ws)                      # src/solver.jl:179-181
    if ws.ci.decompose                                                        # :184-190
        COSMO.reverse_decomposition!(ws, settings)
        y = -ws.vars.μ
    else
        @. ws.utility_vars.vec_m = -ws.vars.μ
        y = ws.utility_vars.vec_m
    end
    ws.times.solver_time = time() - solver_time_start
    destroy!(h)
    return COSMO.Result{T}(ws.vars.x, y, ws.vars.s.data, T(r.cost), Int(r.iter), Int(r.safeguarding_iter), status, res_info, ws.times)
end

# ---------------------------------------------------------------------------------------------------------------------
# Batches of independent problems (BASELINE config 3 and beyond): what a user loop `for ws in models; COSMO.optimize!(ws); end` computes
# (src/solver.jl:78), solved concurrently -- one persistent workgroup per problem.  The problems may differ in shape and cones: the library
# partitions them into classes of identical (n, m, cone table), builds one batch per class and solves all classes at once
# (cosmo_hip_batch_group_*, csrc/batch_group.hip).  Supported cones: ZeroSet, Nonnegatives, Box, SecondOrderCone, PsdCone / PsdConeTriangle of
# side <= 64, the exponential / power cones and their duals; CG KKT solver; EmptyAccelerator or the default AndersonAccelerator (mem <= 16): the
# accelerated loop runs inside the persistent kernels.  ONE Settings object for all problems (that of the first model).
# Per problem the unchanged reference code does the scaling / classification (setup!, src/setup.jl:18-42) and the epilogue
# (src/solver.jl:167-201); the loop of src/solver.jl:137-176 runs on the device for all problems at once.
# ---------------------------------------------------------------------------------------------------------------------
function optimize_hip_batch!(models::Vector{COSMO.Workspace{T}}; device::Integer = 0, kkt_kind::Int32 = KKT_CG, tol_constant = 1.0, tol_exponent = 1.5) where {T <: HipFloat}
    isempty(models) && return COSMO.Result{T}[]
    LIBT = libpath(T)
    settings = models[1].settings
    t_start = time()
    for ws in models
        m, n = ws.p.model_size
        if !ws.states.IS_SCALED
            ws.sm = (settings.scaling > 0) ? COSMO.ScaleMatrices{T}(m, n) : COSMO.ScaleMatrices{T}()
        end
        COSMO.allocate_set_memory!(ws)
        if settings.scaling != 0 && !ws.states.IS_SCALED
            COSMO.scale_ruiz!(ws); ws.states.IS_SCALED = true
        else
            COSMO.scale_variables!(ws.vars.x, ws.vars.μ, ws.vars.s, ws.sm.Dinv, ws.sm.Einv, ws.sm.E, ws.sm.c)
        end
        ws.row_ranges = COSMO.get_set_indices(ws.p.C.sets)
    end
    gptr = Ref{Ptr{Cvoid}}(C_NULL)
    check_abi(T)
    rc = ccall((:cosmo_hip_batch_group_create, LIBT), Int32, (Ref{Ptr{Cvoid}}, Int32, Int64), gptr, device, length(models))
    rc == 0 || error("cosmo_hip_batch_group_create failed (code $rc)")
    g = gptr[]
    gcheck(rc) = rc == 0 || error(unsafe_string(ccall((:cosmo_hip_batch_group_last_error, LIBT), Cstring, (Ptr{Cvoid},), g)))
    try
        for (k, ws) in enumerate(models)
            m, n = ws.p.model_size
            P = SparseMatrixCSC(ws.p.P); A = SparseMatrixCSC(ws.p.A); q = ws.p.q; bv = Vector(ws.p.b)
            GC.@preserve P A q bv gcheck(ccall((:cosmo_hip_batch_group_set_problem, LIBT), Int32,
                (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{T}, Ptr{Int64}, Ptr{Int64}, Ptr{T}, Ptr{T}, Ptr{T}),
                g, k - 1, n, m, P.colptr, P.rowval, P.nzval, A.colptr, A.rowval, A.nzval, q, bv))
            bl = T[]; bu = T[]
            for s in ws.p.C.sets
                if s isa COSMO.Box
                    append!(bl, s.l); append!(bu, s.u)
                end
            end
            types = Int32[cone_type(s) for s in ws.p.C.sets]; dims = Int64[s.dim for s in ws.p.C.sets]
            cparams = T[cone_param(s) for s in ws.p.C.sets]        # alpha of the power cones
            GC.@preserve types dims bl bu cparams gcheck(ccall((:cosmo_hip_batch_group_set_cones, LIBT), Int32,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int32}, Ptr{Int64}, Ptr{T}, Ptr{T}, Ptr{T}),
                g, k - 1, length(types), types, dims, isempty(bl) ? C_NULL : pointer(bl), isempty(bu) ? C_NULL : pointer(bu), cparams))
            if settings.scaling != 0
                Dm = ws.sm.D.diag; Dinv = ws.sm.Dinv.diag; Em = ws.sm.E.diag; Einv = ws.sm.Einv.diag
                GC.@preserve Dm Dinv Em Einv gcheck(ccall((:cosmo_hip_batch_group_set_scaling_full, LIBT), Int32, (Ptr{Cvoid}, Int64, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Cdouble, Cdouble),
                    g, k - 1, Dm, Dinv, Em, Einv, ws.sm.c[], ws.sm.cinv[]))
            end
        end
        ap = accel_params_from(settings)                     # _make_accelerator! (src/setup.jl:10-16) for every problem; before set_params
        ap === nothing || gcheck(ccall((:cosmo_hip_batch_group_set_accelerator, LIBT), Int32, (Ptr{Cvoid}, Ref{AccelParams}), g, Ref(ap)))
        # ws.times.setup_time of this call's set-up phase (solver.jl:246: what the automatic rho interval of the members that run on their own handles is measured against)
        prm = Ref(params_from(settings, kkt_kind; tol_constant = tol_constant, tol_exponent = tol_exponent, setup_time = time() - t_start))
        gcheck(ccall((:cosmo_hip_batch_group_set_params, LIBT), Int32, (Ptr{Cvoid}, Ref{Params}), g, prm))   # classes + classify_constraints! + set_rho_vec! per problem
        for (k, ws) in enumerate(models)
            x0 = ws.vars.x; s0 = ws.vars.s.data; mu0 = ws.vars.μ
            GC.@preserve x0 s0 mu0 gcheck(ccall((:cosmo_hip_batch_group_set_iterates, LIBT), Int32, (Ptr{Cvoid}, Int64, Ptr{T}, Ptr{T}, Ptr{T}), g, k - 1, x0, s0, mu0))
        end
        results_c = Vector{ResultC}(undef, length(models))
        gcheck(ccall((:cosmo_hip_batch_group_optimize, LIBT), Int32, (Ptr{Cvoid}, Ptr{ResultC}), g, results_c))
        out = COSMO.Result{T}[]
        for (k, ws) in enumerate(models)
            r = results_c[k]
            w = ws.vars.w; wp = ws.vars.w_prev; sd = ws.vars.s.data; mu = ws.vars.μ
            GC.@preserve w wp sd mu gcheck(ccall((:cosmo_hip_batch_group_get_iterates, LIBT), Int32, (Ptr{Cvoid}, Int64, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
                g, k - 1, w, wp, sd, mu))
            ws.states.IS_OPTIMIZED = true
            ws.ρ = T(r.rho)
            if ws.settings.adaptive_rho && ws.settings.adaptive_rho_interval == 0      # the reference writes the chosen interval into the settings (src/solver.jl:249-254)
                ri = zeros(Int64, 2)
                gcheck(ccall((:cosmo_hip_batch_group_get_rho_interval, LIBT), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}), g, k - 1, ri))
                ri[1] > 0 && (ws.settings.adaptive_rho_interval = Int(ri[1]))
            end
            resize!(ws.rho_updates, 0); append!(ws.rho_updates, T.(collect(r.rho_updates)[1:min(r.n_rho_updates, MAX_RHO_UPDATES)]))
            ws.times.iter_time = r.iter_time
            res_info = COSMO.ResultInfo(T(r.r_prim), T(r.r_dual), T(r.max_norm_prim), T(r.max_norm_dual), ws.rho_updates)
            settings.scaling != 0 && COSMO.reverse_scaling!(ws)
            @. ws.utility_vars.vec_m = -ws.vars.μ
            ws.times.solver_time = time() - t_start
            push!(out, COSMO.Result{T}(copy(ws.vars.x), copy(ws.utility_vars.vec_m), copy(ws.vars.s.data), T(r.cost), Int(r.iter), Int(r.safeguarding_iter), STATUS[r.status + 1], res_info, ws.times))
        end
        return out
    finally
        ccall((:cosmo_hip_batch_group_destroy, LIBT), Int32, (Ptr{Cvoid},), g)
    end
end

end # module
