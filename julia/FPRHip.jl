# FPRHip.jl -- Julia host shim for libfpr_hip.so (include/fpr.h).
#
# STATUS: written to the C ABI, NOT executed in this repository's CI: neither the build container nor
# the MI355X boxes have a `julia` binary (DESIGN.md, "Host language").  tests/test_abi.py checks that
# every symbol of include/fpr.h is bound below; the Python mirror (finalprojectrepo.jl_amd/) drives the
# same ABI in every test and benchmark.
#
# Purpose: drop-in behind scripts-part1/part1.jl and scripts-part2/part2.jl of
# ntselepidis/FinalProjectRepo.jl.  Replace the preamble
#
#     using CUDA; using ParallelStencil; using ParallelStencil.FiniteDifferences3D
#     @init_parallel_stencil(CUDA, Float64, 3)
#
# by
#
#     include("FPRHip.jl"); using .FPRHip
#     @init_parallel_stencil(AMDGPU, Float64, 3)
#
# and keep the rest of the scripts unchanged:
#   * `@parallel [blocks threads shmem=...] kernel(args...)` drops the launch geometry and calls the HIP kernel of the
#     same name through `ccall`;
#   * the kernel DEFINITIONS the solver files carry next to their host loops -- `@parallel_indices (ix, iy[, iz]) function
#     kernel(...) ... end` (13 of them) and `@parallel function diffusion_3D_step_τ!(...) ... end` -- expand to nothing:
#     this module already defines every one of those names (KERNELS below), an unknown name is an error.  Their bodies
#     (`@sharedMem`, `@threadIdx`, `@blockDim`, `@sync_threads`, `CUDA.@atomic`, `@all`, `@inn`, `@d_xi` ...) are never
#     expanded, so the solver files can be `include`d as they are once their `using` block is replaced;
#   * `include_reference(@__MODULE__, "multigrid.jl")` goes one step further: it includes a reference file WITHOUT any
#     edit, dropping its `using CUDA / ParallelStencil / ImplicitGlobalGrid` lines on the way (and, with `fast = true`,
#     the host functions this module provides natively: the V-cycle as one stream of launches instead of one `ccall`
#     per kernel).
# AMDGPU.jl is used only for device-array allocation (ROCArray) and device selection.
module FPRHip

using AMDGPU
import MPI      # bootstrap only: broadcast of the 128-byte RCCL unique id and the node-local rank (init_global_grid)

export @init_parallel_stencil, @reset_parallel_stencil, @parallel, @parallel_indices, @zeros, @ones, @rand, @synchronize,
       @hide_communication, Data, include_reference,
       diffusion_3D_step_τ, diffusion_3D_step_τ_shared_memory, diffusion_3D_step_τ!,
       compute_flux!, compute_dHdτ!, update_H!, dist_norm_L2, init_local_gaussian_device!,
       residual_2DPoisson!, residual_2DPoisson_shmem!, residual_2DPoisson_wrapper!, iteration_2DPoisson!,
       restrict!, restrict_wrapper!, prolongate!, prolongate_with_atomic!, prolongate_wrapper!,
       matrix_free_matvec_prod!, matrix_free_matvec_prod_shmem!, matrix_free_matvec_prod_wrapper!, cg!,
       Vcycle_2DPoisson!, MGsolve_2DPoisson!, MGOpt, CoarseSolver_t, jacobi, conjugate_gradient,
       ExecutionPolicy_t, serial, parallel, parallel_shmem, preallocate_buffers,
       apply_boundary_conditions!, apply_boundary_conditions_dirichlet!, apply_boundary_conditions_neumann!,
       compute_velocity!, compute_Ra_dTdx!, compute_diffusion2d!, compute_advection2d_x!, compute_advection2d_y!,
       halo_pack!, halo_unpack!, fpr_version,
       init_global_grid, finalize_global_grid, select_device, update_halo!, gather!, nx_g, ny_g, nz_g, x_g, y_g, z_g,
       halo_exchange_begin!, halo_exchange_end!, halo_exchange_comm!, allreduce_sum!,
       diffusion_3D_step_τ2_halo!, diffusion_3D_step_τ3_halo!, can_step_τ3_halo, join_pair!, alloc_fields, alloc_vcycle_fields, provide_arena!, provide_arena_coarse!

const libfpr = get(ENV, "FPR_HIP_LIB", joinpath(@__DIR__, "..", "finalprojectrepo.jl_amd", "lib", "libfpr_hip.so"))

module Data
    using AMDGPU
    const Number = Float64
    const Array = ROCArray{Float64}
end

const CTX = Ref{Ptr{Cvoid}}(C_NULL)

struct FPRError <: Exception
    code::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:fpr_last_error, libfpr), Cstring, (Ptr{Cvoid},), CTX[]))
    rc == -3 && error("ERROR:not a power of 2")            # multigrid.jl:95-97
    rc == -4 && throw(AssertionError(msg))                 # multigrid.jl:45-46
    throw(FPRError(rc, msg))
end

ctx() = (CTX[] == C_NULL && init_context(); CTX[])

const COMM_STREAM = Ref{Any}(nothing)   # keeps the second HIPStream alive for the lifetime of the context

raw_stream(s) = reinterpret(Ptr{Cvoid}, s.stream)   # AMDGPU.HIPStream -> hipStream_t

"""
Create the library context on `device` (0-based).  The COMPUTE stream handed to the library is AMDGPU.jl's current
task stream, so the broadcasts and reductions the reference's host code runs between kernels (`Ht .= Hτ`,
`u_f .= u_f - corr_f`, `sum(rhs.^2)`, `residual_H*dt`, `@zeros` fills) and the library's kernels are ordered on ONE
stream; the comm stream is a second HIPStream owned by this module.  Call from the task that runs the solver.
"""
function init_context(device::Integer = AMDGPU.device_id(AMDGPU.device()) - 1)
    destroy_context()
    AMDGPU.device!(AMDGPU.devices()[device + 1])
    h = Ref{Ptr{Cvoid}}(C_NULL)
    COMM_STREAM[] = AMDGPU.HIPStream(:high)
    rc = ccall((:fpr_ctx_create, libfpr), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Ptr{Cvoid}), h, device,
               raw_stream(AMDGPU.stream()), raw_stream(COMM_STREAM[]))
    rc == 0 || throw(FPRError(rc, "fpr_ctx_create failed (no HIP device? there is no CPU fallback)"))
    CTX[] = h[]
    return nothing
end

function destroy_context()
    if CTX[] != C_NULL
        if CTX2[] != C_NULL
            ccall((:fpr_ctx_destroy, libfpr), Cint, (Ptr{Cvoid},), CTX2[])
            CTX2[] = C_NULL
        end
        ccall((:fpr_ctx_destroy, libfpr), Cint, (Ptr{Cvoid},), CTX[])
        CTX[] = C_NULL
        GRID[] = nothing
        HAS_COMM[] = false
    end
    return nothing
end

fpr_version() = unsafe_string(ccall((:fpr_version, libfpr), Cstring, ()))
set_option(key::String, v::Integer) = check(ccall((:fpr_set_option, libfpr), Cint, (Ptr{Cvoid}, Cstring, Clong), ctx(), key, v))
get_option(key::String) = ccall((:fpr_get_option, libfpr), Clong, (Ptr{Cvoid}, Cstring), ctx(), key)
kernel_timer(on::Bool) = check(ccall((:fpr_kernel_timer, libfpr), Cint, (Ptr{Cvoid}, Cint), ctx(), on))
function kernel_timer_read(kind::Integer = -1)   # FPR_KT_*: 0 step, 1 fused step pair, 2 / 3 / 4 finest MG passes (pre, post, seam), -1 all
    ms = Ref{Cdouble}(0); n = Ref{Clong}(0)
    check(ccall((:fpr_kernel_timer_read, libfpr), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Clong}), ctx(), kind, ms, n))
    return ms[], n[]
end
stream_wait(waiter::Integer, signaller::Integer) = check(ccall((:fpr_stream_wait, libfpr), Cint, (Ptr{Cvoid}, Cint, Cint), ctx(), waiter, signaller))
"Split the device for a decomposed run: stream 1 (comm) on `k` compute units, stream 2 (core) on the others; 0 undoes it."
reserve_comm_cus(k::Integer) = check(ccall((:fpr_reserve_comm_cus, libfpr), Cint, (Ptr{Cvoid}, Cint), ctx(), k))
comm_cus() = ccall((:fpr_comm_cus, libfpr), Cint, (Ptr{Cvoid},), ctx())
function stream_handle(sel::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fpr_stream_handle, libfpr), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), ctx(), sel, h))
    return h[]
end
last_coarse_iters() = ccall((:fpr_last_coarse_iters, libfpr), Clong, (Ptr{Cvoid},), ctx())

# ---- ParallelStencil surface ---------------------------------------------------------------------
macro init_parallel_stencil(args...)
    :(FPRHip.init_context())
end
macro reset_parallel_stencil()
    :(FPRHip.destroy_context())
end
# Every kernel the reference defines with @parallel_indices / @parallel function (14 definitions: part1_kernel_programming.jl:46,75;
# part1_array_programming.jl:9; multigrid.jl:173,191,330,403,427; krylov.jl:7,16; part2.jl:90,99,107,116,128) -- each is a function of
# this module bound to one HIP kernel (or, for the array-programming step, to the three split kernels).
const KERNELS = Set{Symbol}([
    :diffusion_3D_step_τ, :diffusion_3D_step_τ_shared_memory, Symbol("diffusion_3D_step_τ!"),
    Symbol("residual_2DPoisson!"), Symbol("residual_2DPoisson_shmem!"), Symbol("restrict!"), Symbol("prolongate_with_atomic!"),
    Symbol("prolongate!"), Symbol("matrix_free_matvec_prod!"), Symbol("matrix_free_matvec_prod_shmem!"),
    Symbol("compute_velocity!"), Symbol("compute_Ra_dTdx!"), Symbol("compute_diffusion2d!"), Symbol("compute_advection2d_x!"),
    Symbol("compute_advection2d_y!")])

"Name defined by a function-definition expression (long or short form, `where` / return-type annotations, macro prefixes), else `nothing`."
function fdef_name(ex)
    ex isa Expr || return nothing
    ex.head === :macrocall && return fdef_name(ex.args[end])          # `@views function ...`, `@inline function ...`
    if !(ex.head === :function || (ex.head === :(=) && ex.args[1] isa Expr && ex.args[1].head in (:call, :where, :(::))))
        return nothing
    end
    sig = ex.args[1]
    while sig isa Expr && sig.head in (:where, :(::))
        sig = sig.args[1]
    end
    (sig isa Expr && sig.head === :call) || return nothing
    f = sig.args[1]
    if f isa Expr && f.head === :.                                     # `Mod.f(...) = ...`
        f = f.args[end] isa QuoteNode ? f.args[end].value : f.args[end]
    end
    return f isa Symbol ? f : nothing
end

function swallow_kernel_definition(macroname, ex)
    name = fdef_name(ex)
    name === nothing && error("$macroname: expected a kernel definition (`function f(args...) ... end`)")
    name in KERNELS || error("$macroname: FPRHip has no HIP kernel named `$name` -- the hot path it replaces is the set $(sort!(collect(KERNELS)))")
    return nothing          # the definition is dropped: the module's function of that name stands in for it
end

"""
`@parallel_indices (ix, iy[, iz]) function kernel(args...) ... end`: a ParallelStencil kernel DEFINITION.  The reference keeps
its kernels in the same files as the host loops; here the definition expands to nothing -- `kernel` is this module's function
of the same name (one `ccall` into libfpr_hip.so) -- and a kernel this module does not provide is an error.
"""
macro parallel_indices(args...)
    swallow_kernel_definition("@parallel_indices", args[end])
end

"""
`@parallel [blocks threads shmem=n | ranges] f(args...)` -> `f(args...)` (launch geometry is the library's);
`@parallel function f(args...) ... end` (part1_array_programming.jl:9): a kernel definition, swallowed like `@parallel_indices`.
"""
macro parallel(args...)
    fdef_name(args[end]) === nothing || return swallow_kernel_definition("@parallel", args[end])
    esc(args[end])
end

# ---- including the reference's solver files without editing them ------------------------------------------
const ABSENT_PACKAGES = (:CUDA, :ParallelStencil, :ImplicitGlobalGrid)     # this module stands in for them
# host functions / types of the solver files that this module provides natively (dropped with fast = true); functions as
# (name, number of positional arguments): part1_utils.jl's 3-argument apply_boundary_conditions! and part2_utils.jl's 1-argument one are
# different functions, both provided here
const NATIVE_HOST = Set{Tuple{Symbol,Int}}([
    (Symbol("MGsolve_2DPoisson!"), 7), (Symbol("Vcycle_2DPoisson!"), 9), (Symbol("iteration_2DPoisson!"), 6),
    (Symbol("residual_2DPoisson_wrapper!"), 6), (Symbol("restrict_wrapper!"), 4), (Symbol("prolongate_wrapper!"), 4),
    (Symbol("matrix_free_matvec_prod_wrapper!"), 5), (Symbol("cg!"), 7), (:preallocate_buffers, 2),
    (Symbol("apply_boundary_conditions!"), 1), (Symbol("apply_boundary_conditions!"), 3), (Symbol("apply_boundary_conditions_dirichlet!"), 1),
    (Symbol("apply_boundary_conditions_neumann!"), 1), (:dist_norm_L2, 2)])
const NATIVE_TYPES = Set{Symbol}([:ExecutionPolicy_t, :CoarseSolver_t, :MGOpt])

function positional_arity(ex)
    ex isa Expr || return -1
    ex.head === :macrocall && return positional_arity(ex.args[end])
    sig = ex.args[1]
    while sig isa Expr && sig.head in (:where, :(::))
        sig = sig.args[1]
    end
    (sig isa Expr && sig.head === :call) || return -1
    return count(a -> !(a isa Expr && a.head === :parameters), sig.args[2:end])
end

uses_absent_package(ex) = ex isa Expr && ex.head in (:using, :import) &&
    any(a -> begin
            path = a isa Expr && a.head === :(:) ? a.args[1] : a          # `using A: b, c`
            path isa Expr && path.head === :. && !isempty(path.args) && path.args[1] in ABSENT_PACKAGES
        end, ex.args)

function native_type_definition(ex)
    ex isa Expr || return false
    ex.head === :struct && return (n = ex.args[2]; n = n isa Expr ? n.args[1] : n; n in NATIVE_TYPES)
    if ex.head === :macrocall && ex.args[1] === Symbol("@enum")
        n = ex.args[findfirst(a -> !(a isa LineNumberNode), ex.args[2:end]) + 1]
        return n in NATIVE_TYPES
    end
    return false
end

"""
    include_reference(mod, path; fast = true)

`include` one of the reference's solver files (scripts-part1/part1_kernel_programming.jl, part1_array_programming.jl,
scripts-part2/multigrid.jl, krylov.jl, part2.jl ...) into `mod` AS IT IS: `using` / `import` lines of CUDA, ParallelStencil and
ImplicitGlobalGrid are dropped (this module stands in for them; `mod` must have done `using .FPRHip`), nested `include(...)`s go
through the same filter, kernel definitions are swallowed by `@parallel_indices` / `@parallel`.  `fast = true` also drops the
host functions, enums and `MGOpt` this module provides natively (NATIVE_HOST, NATIVE_TYPES), so that `MGsolve_2DPoisson!`,
`Vcycle_2DPoisson!`, `cg!` ... are the library's fused paths instead of the reference's kernel-by-kernel loops.
"""
function include_reference(mod::Module, path::AbstractString; fast::Bool = true)
    dir = dirname(abspath(path))
    Base.include(mod, path) do ex
        uses_absent_package(ex) && return nothing
        if ex isa Expr && ex.head === :call && ex.args[1] === :include && length(ex.args) == 2
            return :($(include_reference)($mod, joinpath($dir, $(ex.args[2])); fast = $fast))
        end
        if fast
            native_type_definition(ex) && return nothing
            name = fdef_name(ex)
            name !== nothing && (name, positional_arity(ex)) in NATIVE_HOST && return nothing
        end
        return ex
    end
end

"""
`@hide_communication (bx, by, bz) begin @parallel kernel(args...); update_halo!(A...) end` (part1_kernel_programming.jl:185-188):
ParallelStencil computes the boundary slabs first, then the inner points while `update_halo!` runs.  Here, for the one kernel
the reference uses it with (`diffusion_3D_step_τ`), the block becomes `step_hide_communication(args...)`: the kernel on the thin
boxes next to the faces that have a neighbour, the exchange of those planes on the comm stream, the kernel on the rest of the
interior beside it, join.  Any other body runs as written (kernel, then `update_halo!`).  The boundary width is the library's
(one cell: all a 7-point stencil needs).  Deviation kept from the Python mirror (DESIGN 2): the planes of the buffer the kernel
WROTE are exchanged -- the reference's `update_halo!(Hτ)` names the old buffer, whose halos are therefore one iteration behind.
"""
macro hide_communication(args...)
    body = args[end]
    stmts = body isa Expr && body.head === :block ? [a for a in body.args if !(a isa LineNumberNode)] : Any[body]
    if length(stmts) == 2 && stmts[1] isa Expr && stmts[1].head === :macrocall && stmts[1].args[1] === Symbol("@parallel") &&
       stmts[2] isa Expr && stmts[2].head === :call && stmts[2].args[1] === Symbol("update_halo!")
        call = stmts[1].args[end]
        if call isa Expr && call.head === :call && call.args[1] === :diffusion_3D_step_τ && length(call.args) == 13
            return esc(:(FPRHip.step_hide_communication($(call.args[2:end]...))))
        end
    end
    esc(body)
end

"One pseudo-iteration with neighbours: boundary boxes, exchange of the written buffer's planes beside the interior box, join."
function step_hide_communication(Ht::DA, Hτ::DA, Hτ2::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
    join_pair!()
    faces = GRID[] === nothing ? Int[] : neighbour_faces()
    if isempty(faces)
        return diffusion_3D_step_τ(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
    end
    boxes, inner = boundary_boxes(size(Ht), faces)
    for (lo, hi) in boxes
        diffusion_3D_step_τ_box(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo, hi)
    end
    halo_exchange_begin!(Hτ2)
    if comm_cus() > 0        # a split device: the interior on the core stream, whose units the exchange does not compete for
        stream_wait(2, 0)
        diffusion_3D_step_τ_box(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, inner[1], inner[2]; stream_sel = 2)
        stream_wait(0, 2)
    else
        diffusion_3D_step_τ_box(Ht, Hτ, Hτ2, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, inner[1], inner[2])
    end
    halo_exchange_end!(Hτ2)
    return nothing
end

macro zeros(dims...)
    esc(:(AMDGPU.zeros(Float64, $(dims...))))
end
macro ones(dims...)
    esc(:(AMDGPU.ones(Float64, $(dims...))))
end
macro rand(dims...)
    esc(:(AMDGPU.rand(Float64, $(dims...))))
end
"`@synchronize()`: the library's two streams (the compute stream IS AMDGPU.jl's task stream) and AMDGPU.jl itself."
macro synchronize()
    quote
        FPRHip.check(ccall((:fpr_synchronize, FPRHip.libfpr), Cint, (Ptr{Cvoid},), FPRHip.ctx()))
        FPRHip.AMDGPU.synchronize()
    end
end

const DA = ROCArray{Float64}
p(A::DA) = Ptr{Cdouble}(UInt(pointer(A)))

# ---- Part 1 (scripts-part1/part1_kernel_programming.jl, part1_array_programming.jl, part1_utils.jl) ----
function diffusion_3D_step_τ(Ht::DA, Hτ::DA, Hτ2::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz)
    nx, ny, nz = size(Ht)
    check(ccall((:fpr_diffusion3d_step, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble),
                ctx(), p(Ht), p(Hτ), p(Hτ2), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz))
    return nothing
end
const diffusion_3D_step_τ_shared_memory = diffusion_3D_step_τ   # part1_kernel_programming.jl:75-97

"Fused update + local sum((dHdτ*scale)^2) into the device scalar `sumsq` (1-element ROCArray)."
function diffusion_3D_step_τ_norm(Ht::DA, Hτ::DA, Hτ2::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, sumsq::DA)
    nx, ny, nz = size(Ht)
    check(ccall((:fpr_diffusion3d_step_norm, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}),
                ctx(), p(Ht), p(Hτ), p(Hτ2), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, p(sumsq)))
end

"Fused update + norm handed to the host (one stream synchronisation per pseudo-iteration)."
function diffusion_3D_step_τ_norm_host(Ht::DA, Hτ::DA, Hτ2::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale)
    nx, ny, nz = size(Ht); out = Ref{Cdouble}(0)
    check(ccall((:fpr_diffusion3d_step_norm_host, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}),
                ctx(), p(Ht), p(Hτ), p(Hτ2), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale, out))
    return out[]
end

"""
Native single-rank host loop (part1_kernel_programming.jl:166-204); returns (iters, errs, swapped).  `Hτ3`: a third work
array owned by the caller (`nothing` = one iteration per launch); with it pairs of iterations run as fused launches.
"""
function diffusion_3D_solve!(Ht::DA, Hτ::DA, Hτ2::DA, Hτ3::Union{DA,Nothing}, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, dt, total_N, nt;
                             tol = 1e-8, iter_max = 100000, fixed_iters = 0, check_every = 1)
    nx, ny, nz = size(Ht); its = zeros(Clong, max(nt, 1)); errs = zeros(Cdouble, max(nt, 1)); sw = Ref{Cint}(0)
    check(ccall((:fpr_diffusion3d_solve, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cdouble, Clong, Clong, Cint,
                 Ptr{Clong}, Ptr{Cdouble}, Ptr{Cint}),
                ctx(), p(Ht), p(Hτ), p(Hτ2), Hτ3 === nothing ? Ptr{Cdouble}(C_NULL) : p(Hτ3), p(dHdτ), nx, ny, nz,
                dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, dt, total_N, nt, tol, iter_max, fixed_iters, check_every, its, errs, sw))
    return its[1:nt], errs[1:nt], sw[] != 0
end

"Sub-box form for boundary/interior splitting (0-based lo/hi); role of @hide_communication."
function diffusion_3D_step_τ_box(Ht::DA, Hτ::DA, Hτ2::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz,
                                 lo::NTuple{3,Int}, hi::NTuple{3,Int}; scale = 0.0, sumsq::Union{DA,Nothing} = nothing, stream_sel = 0)
    nx, ny, nz = size(Ht)
    lo3 = Cint[lo...]; hi3 = Cint[hi...]
    check(ccall((:fpr_diffusion3d_step_box, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cint}, Ptr{Cint}, Cdouble, Ptr{Cdouble}, Cint),
                ctx(), p(Ht), p(Hτ), p(Hτ2), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3, scale,
                sumsq === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq), stream_sel))
end

"Two loop trips (part1_kernel_programming.jl:179-192) in one pass: step(Hτ -> Hmid); step(Hmid -> Hout), Hmid never written."
can_step_τ2(Ht::DA, Hτ::DA, Hmid::DA, Hout::DA, dHdτ::DA) =
    ccall((:fpr_diffusion3d_can_step2, libfpr), Cint,
          (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint),
          ctx(), p(Ht), p(Hτ), p(Hmid), p(Hout), p(dHdτ), size(Ht)...) == 1
function diffusion_3D_step_τ2(Ht::DA, Hτ::DA, Hmid::DA, Hout::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
                              scale = 0.0, sumsq2::Union{DA,Nothing} = nothing)
    nx, ny, nz = size(Ht)
    check(ccall((:fpr_diffusion3d_step2, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}),
                ctx(), p(Ht), p(Hτ), p(Hmid), p(Hout), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale,
                sumsq2 === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq2)))
end
"Three loop trips (part1_kernel_programming.jl:179-192) in one pass: Hτ -> Hout, the reference's two ping-pong buffers in either order; no third buffer."
can_step_τ3(Ht::DA, Hτ::DA, Hout::DA, dHdτ::DA) =
    ccall((:fpr_diffusion3d_can_step3, libfpr), Cint,
          (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint),
          ctx(), p(Ht), p(Hτ), p(Hout), p(dHdτ), size(Ht)...) == 1
function diffusion_3D_step_τ3(Ht::DA, Hτ::DA, Hout::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
                              scale = 0.0, sumsq3::Union{DA,Nothing} = nothing)
    nx, ny, nz = size(Ht)
    check(ccall((:fpr_diffusion3d_step3, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}),
                ctx(), p(Ht), p(Hτ), p(Hout), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, scale,
                sumsq3 === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq3)))
end
function diffusion_3D_step_τ2_box(Ht::DA, Hτ::DA, Hmid::DA, Hout::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz,
                                  lo::NTuple{3,Int}, hi::NTuple{3,Int}; scale = 0.0, sumsq2::Union{DA,Nothing} = nothing, stream_sel = 0)
    nx, ny, nz = size(Ht)
    lo3 = Cint[lo...]; hi3 = Cint[hi...]
    check(ccall((:fpr_diffusion3d_step2_box, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cint}, Ptr{Cint}, Cdouble, Ptr{Cdouble}, Cint),
                ctx(), p(Ht), p(Hτ), p(Hmid), p(Hout), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3, scale,
                sumsq2 === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq2), stream_sel))
end

"As diffusion_3D_step_τ2_box plus a second, disjoint z-range [zlo2, zhi2) with the same x/y extent in the same launch."
function diffusion_3D_step_τ2_box2(Ht::DA, Hτ::DA, Hmid::DA, Hout::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz,
                                   lo::NTuple{3,Int}, hi::NTuple{3,Int}, zlo2::Int, zhi2::Int; scale = 0.0,
                                   sumsq2::Union{DA,Nothing} = nothing, stream_sel = 0)
    nx, ny, nz = size(Ht)
    lo3 = Cint[lo...]; hi3 = Cint[hi...]
    check(ccall((:fpr_diffusion3d_step2_box2, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cint}, Ptr{Cint}, Cint, Cint, Cdouble, Ptr{Cdouble}, Cint),
                ctx(), p(Ht), p(Hτ), p(Hmid), p(Hout), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3,
                zlo2, zhi2, scale, sumsq2 === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq2), stream_sel))
end

"""
The CORE box of a decomposed run's fused pair, leaving `reserve_cus` compute units without a workgroup of it: the shell
launches and RCCL's send / receive kernels of the pair run beside it on the comm stream (`stream_sel = 1` launches,
`halo_exchange_comm!`); role of `@hide_communication` (part1_kernel_programming.jl:185-188) for two iterations at once.
"""
function diffusion_3D_step_τ2_core(Ht::DA, Hτ::DA, Hmid::DA, Hout::DA, dHdτ::DA, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz,
                                   lo::NTuple{3,Int}, hi::NTuple{3,Int}; scale = 0.0, sumsq2::Union{DA,Nothing} = nothing,
                                   stream_sel = 0, reserve_cus = 32, accumulate = true)
    nx, ny, nz = size(Ht)
    lo3 = Cint[lo...]; hi3 = Cint[hi...]
    check(ccall((:fpr_diffusion3d_step2_core, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cint}, Ptr{Cint}, Cdouble, Ptr{Cdouble}, Cint, Cint, Cint),
                ctx(), p(Ht), p(Hτ), p(Hmid), p(Hout), p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz, lo3, hi3, scale,
                sumsq2 === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq2), stream_sel, reserve_cus, accumulate))
end

function compute_flux!(qx::DA, qy::DA, qz::DA, Hτ::DA, D, dx, dy, dz)        # part1_array_programming.jl:10-12
    nx, ny, nz = size(Hτ)
    check(ccall((:fpr_diffusion3d_flux, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cdouble, Cdouble, Cdouble, Cdouble),
                ctx(), p(qx), p(qy), p(qz), p(Hτ), nx, ny, nz, D, dx, dy, dz))
end
function compute_dHdτ!(dHdτ::DA, Hτ::DA, Ht::DA, qx::DA, qy::DA, qz::DA, dt, dx, dy, dz)   # :14-15
    nx, ny, nz = size(Hτ)
    check(ccall((:fpr_diffusion3d_dHdtau, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble),
                ctx(), p(dHdτ), p(Hτ), p(Ht), p(qx), p(qy), p(qz), nx, ny, nz, dt, dx, dy, dz))
end
function update_H!(Hτ::DA, dHdτ::DA, dτ)                                          # :16
    nx, ny, nz = size(Hτ)
    check(ccall((:fpr_diffusion3d_update, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cdouble),
                ctx(), p(Hτ), p(dHdτ), nx, ny, nz, dτ))
end
"part1_array_programming.jl:9-18 with clean (barrier-separated) semantics."
function diffusion_3D_step_τ!(Ht, Hτ, dHdτ, dt, dτ, qx, qy, qz, dx, dy, dz, D)
    compute_flux!(qx, qy, qz, Hτ, D, dx, dy, dz)
    compute_dHdτ!(dHdτ, Hτ, Ht, qx, qy, qz, dt, dx, dy, dz)
    update_H!(Hτ, dHdτ, dτ)
end

function sumsq_scaled(x::DA, scale = 1.0)
    out = Ref{Cdouble}(0)
    check(ccall((:fpr_sumsq_scaled, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t, Cdouble, Ptr{Cdouble}), ctx(), p(x), length(x), scale, out))
    return out[]
end
sumsq_scaled_dev!(out::DA, x::DA, scale = 1.0) =
    check(ccall((:fpr_sumsq_scaled_dev, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t, Cdouble, Ptr{Cdouble}), ctx(), p(x), length(x), scale, p(out)))
function dot_dev(x::DA, y::DA)
    out = Ref{Cdouble}(0)
    check(ccall((:fpr_dot, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t, Ptr{Cdouble}), ctx(), p(x), p(y), length(x), out))
    return out[]
end
function absmax(x::DA)
    out = Ref{Cdouble}(0)
    check(ccall((:fpr_absmax, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t, Ptr{Cdouble}), ctx(), p(x), length(x), out))
    return out[]
end
"MPI.Allreduce!(x, +, comm_cart) of one host Float64 (part1_utils.jl:38) over RCCL; identity on a single rank."
function allreduce_sum1(x::Float64)
    r = Ref{Cdouble}(x)
    check(ccall((:fpr_allreduce_sum1, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), ctx(), r))
    return r[]
end
"In-place sum over all ranks of a small device vector (e.g. the norms of several iterations); no host sync."
function allreduce_sum!(x::DA)   # comm stream, ordered behind / ahead of the compute stream (all RCCL calls share one stream)
    join_pair!()
    stream_wait(1, 0)
    check(ccall((:fpr_allreduce_sum_dev, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint), ctx(), p(x), length(x), 1))
    stream_wait(0, 1)
end
"part1_utils.jl:36-40 with the scale of its call site folded in: dist_norm_L2(residual_H*dt, comm) == dist_norm_L2(residual_H, comm; scale=dt)."
function dist_norm_L2(Rh::DA, comm_cart; scale = 1.0)
    sq = sumsq_scaled(Rh, scale)                                  # part1_utils.jl:37
    comm_cart === nothing || (sq = allreduce_sum1(sq))            # part1_utils.jl:38
    return sqrt(sq)
end
copy_device!(dst::DA, src::DA) = check(ccall((:fpr_copy, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t), ctx(), p(dst), p(src), length(dst)))

"""
    alloc_fields(count, dims...; pool = count + 7, pairs = nothing, trial = nothing, spacer_bytes = nothing) -> Vector{ROCArray{Float64}}

Field arrays placed for the streaming kernels (twin of `finalprojectrepo.jl_amd/placement.py`, DESIGN.md 3): on MI355X two arrays that a
kernel streams at equal offsets get in each other's way when their allocations carry the same placement label (a property of the physical
pages: the fused diffusion launch takes 0.76 ms at 512^3 on arrays that differ, 0.85-0.91 ms on arrays of one class).  The measurement and
the search are ONE call into the library (`fpr_placement_rank`, include/fpr.h); the host only allocates the pool -- the first `count`
plainly, as `@zeros` would, the rest behind untouched spacers -- and frees what was not chosen.  `pairs`: 1-based positions streamed
together (default all); `trial(arrays) -> ms`: the caller's own kernel as the judge (optional; the candidates as allocated are always among
its trials).  The candidates' contents are scratch during the search; nothing but the returned arrays stays allocated.  Replaces nothing in the
reference: `@zeros` (part1_kernel_programming.jl:134-142) keeps working without it -- and the three-iteration launch of a single rank does not
need it at all (DESIGN.md 3).
"""
function alloc_fields(count::Integer, dims::Integer...; pool::Integer = count + 7, pairs = nothing, trial = nothing, spacer_bytes = nothing)
    nbytes = 8 * prod(dims)
    (nbytes < (256 << 20) || count < 2) && return [AMDGPU.zeros(Float64, dims...) for _ in 1:count]
    spacer = spacer_bytes === nothing ? max(4 << 30, 3 * nbytes) : spacer_bytes
    cands, spacers = DA[], Any[]
    for i in 1:pool
        (spacer > 0 && i > count) && push!(spacers, ROCArray{UInt8}(undef, spacer))      # reserved, never touched
        push!(cands, AMDGPU.zeros(Float64, dims...))
    end
    ptrs = Ptr{Cvoid}[Ptr{Cvoid}(p(A)) for A in cands]
    flat = pairs === nothing ? Cint[] : Cint[x - 1 for q in pairs for x in q]
    chosen, report = zeros(Cint, count), zeros(Cdouble, 16)
    judge = (_user::Ptr{Cvoid}, idx::Ptr{Cint}, n::Cint) -> Cdouble(trial([cands[unsafe_load(idx, i) + 1] for i in 1:n]))
    cb = trial === nothing ? C_NULL : @cfunction($judge, Cdouble, (Ptr{Cvoid}, Ptr{Cint}, Cint))
    AMDGPU.synchronize()
    GC.@preserve cands ptrs flat chosen report cb begin
        check(ccall((:fpr_placement_rank, libfpr), Cint,
                    (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cint, Csize_t, Cint, Ptr{Cint}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cdouble}),
                    ctx(), ptrs, length(cands), prod(dims), count, flat, length(flat) ÷ 2,
                    cb === C_NULL ? C_NULL : Base.unsafe_convert(Ptr{Cvoid}, cb), C_NULL, chosen, report))
    end
    out = DA[cands[c + 1] for c in chosen]
    for (i, A) in enumerate(cands); (i - 1) in chosen || AMDGPU.unsafe_free!(A); end
    for q in spacers; AMDGPU.unsafe_free!(q); end
    foreach(A -> fill_device!(A, 0.0), out)
    return out
end
fill_device!(dst::DA, v) = check(ccall((:fpr_fill, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Csize_t), ctx(), p(dst), v, length(dst)))
fill_on!(dst::DA, v, stream_sel) = check(ccall((:fpr_fill_on, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Csize_t, Cint), ctx(), p(dst), v, length(dst), stream_sel))
add_on!(dst::DA, src::DA, stream_sel) = check(ccall((:fpr_add_on, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t, Cint), ctx(), p(dst), p(src), length(dst), stream_sel))
function init_local_gaussian_device!(H::DA, center, dx, dy, dz, coords)
    nx, ny, nz = size(H)
    check(ccall((:fpr_init_gaussian3d, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                ctx(), p(H), nx, ny, nz, dx, dy, dz, center[1], center[2], center[3], coords[1], coords[2], coords[3]))
    return H
end
"update_halo! building blocks: face = 2*dim + side (0-based); buffers are ROCArray{Float64} planes."
halo_pack!(buf::DA, A::DA, face; stream_sel = 0) = (n = size(A);
    check(ccall((:fpr_halo_pack3d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Ptr{Cdouble}, Cint), ctx(), p(A), n[1], n[2], n[3], face, p(buf), stream_sel)))
halo_unpack!(A::DA, buf::DA, face; stream_sel = 0) = (n = size(A);
    check(ccall((:fpr_halo_unpack3d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Ptr{Cdouble}, Cint), ctx(), p(A), n[1], n[2], n[3], face, p(buf), stream_sel)))

# ---- ImplicitGlobalGrid surface (part1_kernel_programming.jl:100-101,117,121,182,187,223,225) over RCCL ----
# One process per GPU.  MPI.jl is used for two bootstrap steps only (node-local rank, broadcast of the RCCL unique
# id); every byte of the data path (halo planes, the norm's all-reduce, gather!) travels through the library.
const GRID = Ref{Any}(nothing)   # (me, dims, nprocs, coords, n = (nx,ny,nz), periods, comm)
const HAS_COMM = Ref(false)      # fpr_comm_init done on the current context
grid() = (GRID[] === nothing && error("init_global_grid has not been called"); GRID[])

# The library's exchange code over a host-staged transport (fpr_comm_init_hosted, include/fpr.h): MPI carries the bytes, as in the
# reference, while pack / unpack kernels, face order and choreography stay the library's.  For rehearsals with several ranks on one
# card (RCCL refuses that) -- `init_global_grid(...; hosted_transport = true)`; rates through it are not measurements.
const HOSTED_PENDING = Any[]     # (request, buffer) of sends not yet complete
function hosted_send(comm_ptr::Ptr{Cvoid}, peer::Cint, buf::Ptr{Cvoid}, nbytes::Csize_t)::Cint
    comm = unsafe_pointer_to_objref(comm_ptr)::MPI.Comm
    data = copy(unsafe_wrap(Array, Ptr{UInt8}(buf), Int(nbytes)))
    push!(HOSTED_PENDING, (MPI.Isend(data, comm; dest = Int(peer), tag = 7), data))
    filter!(q -> !MPI.Test(q[1]), HOSTED_PENDING)
    return Cint(0)
end
function hosted_recv(comm_ptr::Ptr{Cvoid}, peer::Cint, buf::Ptr{Cvoid}, nbytes::Csize_t)::Cint
    comm = unsafe_pointer_to_objref(comm_ptr)::MPI.Comm
    MPI.Recv!(unsafe_wrap(Array, Ptr{UInt8}(buf), Int(nbytes)), comm; source = Int(peer), tag = 7)
    return Cint(0)
end
function hosted_allreduce(comm_ptr::Ptr{Cvoid}, x::Ptr{Cdouble}, count::Cint)::Cint
    comm = unsafe_pointer_to_objref(comm_ptr)::MPI.Comm
    MPI.Allreduce!(unsafe_wrap(Array, x, Int(count)), +, comm)
    return Cint(0)
end
const HOSTED_COMM = Ref{Any}(nothing)    # keeps the MPI.Comm alive whose address the library holds
function comm_init_hosted_mpi(me::Integer, np::Integer, comm)
    HOSTED_COMM[] = comm
    send = @cfunction(hosted_send, Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Csize_t))
    recv = @cfunction(hosted_recv, Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Csize_t))
    allr = @cfunction(hosted_allreduce, Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint))
    check(ccall((:fpr_comm_init_hosted, libfpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                ctx(), me, np, send, recv, allr, pointer_from_objref(comm)))
end

"`select_device()`: bind this rank to GPU (node-local rank mod device count); returns the 0-based device id."
function select_device()
    loc = MPI.Comm_split_type(MPI.COMM_WORLD, MPI.COMM_TYPE_SHARED, MPI.Comm_rank(MPI.COMM_WORLD))
    dev = MPI.Comm_rank(loc) % length(AMDGPU.devices())
    if CTX[] == C_NULL || AMDGPU.device_id(AMDGPU.device()) - 1 != dev
        GRID[] === nothing || error("select_device() after init_global_grid must not change the device")
        init_context(dev)
    end
    return dev
end

"""
`init_global_grid(nx, ny, nz; dimx, dimy, dimz, periodx, periody, periodz, init_MPI, quiet)` ->
`(me, dims, nprocs, coords, comm_cart)` as ImplicitGlobalGrid returns them.  Unlike IGG the device is bound HERE
(the RCCL communicator belongs to a device), so the `select_device()` call that follows in the reference's solvers
(part1_kernel_programming.jl:121) finds the device already selected.
"""
function init_global_grid(nx::Integer, ny::Integer, nz::Integer; dimx = 0, dimy = 0, dimz = 0, periodx = 0, periody = 0,
                          periodz = 0, init_MPI = true, quiet = false, kwargs...)
    init_MPI && !MPI.Initialized() && MPI.Init()
    comm = MPI.COMM_WORLD
    me, np = MPI.Comm_rank(comm), MPI.Comm_size(comm)
    select_device()
    id = zeros(UInt8, 128)                                       # FPR_UNIQUE_ID_BYTES
    me == 0 && (ccall((:fpr_comm_get_unique_id, libfpr), Cint, (Ptr{UInt8},), id) == 0 || error("fpr_comm_get_unique_id failed"))
    MPI.Bcast!(id, 0, comm)
    # a communicator whenever this context has none yet (a single rank needs one too: a periodic dimension makes it its own
    # neighbour, and the planes then travel through ncclSend / ncclRecv like between ranks), or when the world changed
    if !HAS_COMM[] || ccall((:fpr_comm_size, libfpr), Cint, (Ptr{Cvoid},), ctx()) != np || ccall((:fpr_comm_rank, libfpr), Cint, (Ptr{Cvoid},), ctx()) != me
        check(ccall((:fpr_comm_finalize, libfpr), Cint, (Ptr{Cvoid},), ctx()))
        if get(kwargs, :hosted_transport, false)
            comm_init_hosted_mpi(me, np, comm)      # ranks sharing a card / no RCCL: the bytes travel through MPI, host-staged
        else
            check(ccall((:fpr_comm_init, libfpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx(), me, np, id))
        end
        HAS_COMM[] = true
    end
    me_o = Ref{Cint}(0); np_o = Ref{Cint}(0); dims = zeros(Cint, 3); coords = zeros(Cint, 3)
    check(ccall((:fpr_grid_init, libfpr), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}),
                ctx(), nx, ny, nz, dimx, dimy, dimz, periodx, periody, periodz, me_o, dims, np_o, coords))
    GRID[] = (me = Int(me_o[]), dims = Int.(dims), nprocs = Int(np_o[]), coords = Int.(coords), n = (Int(nx), Int(ny), Int(nz)),
              periods = (periodx, periody, periodz), comm = comm)
    quiet || me != 0 || println("Global grid: $(nx_g())x$(ny_g())x$(nz_g()) (nprocs: $np, dims: $(dims[1])x$(dims[2])x$(dims[3]))")
    return GRID[].me, GRID[].dims, GRID[].nprocs, GRID[].coords, comm
end

"`finalize_global_grid(; finalize_MPI)`: releases the RCCL communicator and the library's pack buffers."
function finalize_global_grid(; finalize_MPI = true)
    CTX[] == C_NULL || check(ccall((:fpr_comm_finalize, libfpr), Cint, (Ptr{Cvoid},), CTX[]))
    GRID[] = nothing
    HAS_COMM[] = false
    finalize_MPI && MPI.Initialized() && !MPI.Finalized() && MPI.Finalize()
    return nothing
end

function grid_sizes_g()
    ng = zeros(Cint, 3); nb = zeros(Cint, 6)
    check(ccall((:fpr_grid_info, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}), ctx(), ng, nb))
    return Int.(ng), Int.(nb)
end
nx_g() = grid_sizes_g()[1][1]      # dims[1]*(nx-2)+2  (part1_kernel_programming.jl:117)
ny_g() = grid_sizes_g()[1][2]
nz_g() = grid_sizes_g()[1][3]
"Global coordinate of local index `i` (1-based) in dimension `d` of array `A` (ImplicitGlobalGrid's x_g/y_g/z_g; part1_utils.jl:5-7)."
function coord_g(d::Int, i::Integer, dd::Real, A)
    g = grid(); n = g.n[d]
    x0 = 0.5 * (n - size(A, d)) * dd                       # staggered arrays sit half a cell in
    x = (g.coords[d] * (n - 2) + (i - 1)) * dd + x0
    if g.periods[d] != 0
        # ImplicitGlobalGrid, periodic dimension [3P-memory, SURVEY 8c]: the first cell of the global problem is a ghost
        # cell, so everything shifts one cell to the left and wraps around the global extent n_g * dd
        ng = grid_sizes_g()[1][d]
        x -= dd
        x > (ng - 1) * dd && (x -= ng * dd)
        x < 0 && (x += ng * dd)
    end
    return x
end
x_g(ix::Integer, dx::Real, A) = coord_g(1, ix, dx, A)
y_g(iy::Integer, dy::Real, A) = coord_g(2, iy, dy, A)
z_g(iz::Integer, dz::Real, A) = coord_g(3, iz, dz, A)

"`update_halo!(A...)` (part1_kernel_programming.jl:182,187): ordered on the compute stream, no host synchronisation."
function update_halo!(As::DA...)
    join_pair!()
    for A in As
        n = size(A)
        check(ccall((:fpr_halo_exchange3d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint), ctx(), p(A), n[1], n[2], n[3]))
    end
    return nothing
end
"Split form: kernels launched between begin and end overlap the transfers (role of `@hide_communication`)."
halo_exchange_begin!(A::DA; faces = 63) = (n = size(A);
    check(ccall((:fpr_halo_exchange3d_begin, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Cint), ctx(), p(A), n[1], n[2], n[3], faces)))
halo_exchange_end!(A::DA; faces = 63) = (n = size(A);
    check(ccall((:fpr_halo_exchange3d_end, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Cint), ctx(), p(A), n[1], n[2], n[3], faces)))
"The whole exchange (pack, one RCCL group, unpack) in the COMM stream's order: a link of the shell chain of a fused pair."
halo_exchange_comm!(A::DA; faces = 63) = (n = size(A);
    check(ccall((:fpr_halo_exchange3d_comm, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Cint), ctx(), p(A), n[1], n[2], n[3], faces)))

# ---- iterations BETWEEN ranks: the overlapped single step (@hide_communication) and the fused pair (one library call) ------
"Faces (0-based, 2*dim + side) of this rank that have a neighbour."
neighbour_faces() = [f for f in 0:5 if grid_sizes_g()[2][f + 1] >= 0]

"""
Thin boxes (0-based `[lo, hi)`) of the interior cells next to faces with a neighbour -- z faces peeled first, then y, then
x, so the boxes are disjoint -- and the remaining core box.
"""
function boundary_boxes(n::NTuple{3,Int}, faces)
    lo = [1, 1, 1]; hi = [n[1] - 1, n[2] - 1, n[3] - 1]
    boxes = Tuple{NTuple{3,Int},NTuple{3,Int}}[]
    for d in (3, 2, 1), side in (0, 1)
        f = 2 * (d - 1) + side
        (f in faces && hi[d] - lo[d] >= 1) || continue
        blo = copy(lo); bhi = copy(hi)
        if side == 0
            bhi[d] = lo[d] + 1; lo[d] += 1
        else
            blo[d] = hi[d] - 1; hi[d] -= 1
        end
        push!(boxes, (Tuple(blo), Tuple(bhi)))
    end
    return boxes, (Tuple(lo), Tuple(hi))
end


"""
    diffusion_3D_step_τ2_halo!(Ht, Hτ, Hτ2, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz; scale, sumsq2, join)

Two trips through the loop body of part1_kernel_programming.jl:179-192 on a rank WITH neighbours, halos of `Hout` refreshed --
what `@hide_communication` + `update_halo!` do there for one iteration; ONE call of the library (`fpr_diffusion3d_step2_halo`;
`GlobalGrid.step2_begin/middle/end` in finalprojectrepo.jl_amd/grid.py is the same choreography in phases).  The device is split
(`reserve_comm_cus`): the core of
the local grid runs as ONE fused launch on the core stream; on the comm stream, beside it, run the single steps on the
one-cell shell (level 1 into `Hτ2`), the exchange of `Hτ2`'s planes, the fused launches on the shell boxes and the exchange of
the new field (the shell next to an x-neighbour in compact strips of the library's own, csrc/diffusion3d_xstrip.hpp).  `join = false`
leaves the pair on those two streams (the next pair continues from there and reuses the strips); call `join_pair!()` before anything
else reads OR WRITES the fields or `sumsq2`.  `dHdτ = nothing`: the residual is not stored.  Same results as two single steps with
`update_halo!` of the new buffer after each.
"""
function diffusion_3D_step_τ2_halo!(Ht::DA, Hτ::DA, Hτ2::DA, Hout::DA, dHdτ::Union{DA,Nothing}, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
                                    scale = 0.0, sumsq2::Union{DA,Nothing} = nothing, join::Bool = true)
    nx, ny, nz = size(Ht)
    check(ccall((:fpr_diffusion3d_step2_halo, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cint),
                ctx(), p(Ht), p(Hτ), p(Hτ2), p(Hout), dHdτ === nothing ? Ptr{Cdouble}(C_NULL) : p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz,
                D_dx, D_dy, D_dz, scale, sumsq2 === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq2), join ? 1 : 0))
    return nothing
end

"""
    diffusion_3D_step_τ3_halo!(Ht, Hτ, Hout, dHdτ, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz; scale, sumsq3, join)

Three trips through the loop body of part1_kernel_programming.jl:179-192 on a rank of a z-slab decomposition (`dimz = N`, neighbours on
z-faces only), halos of `Hout` refreshed: ONE call of the library (`fpr_diffusion3d_step3_halo`).  `Hτ` and `Hout` are the reference's
two ping-pong buffers in either order.  The core planes run as one launch of the three-step kernel; the two planes next to each z-face
go through three rounds of single-step launches with the exchange of one plane per face after each -- one exchange per iteration, as in
the reference.  `can_step_τ3_halo` says whether the grid and the arrays qualify (else pairs: `diffusion_3D_step_τ2_halo!`).
"""
can_step_τ3_halo(Ht::DA, Hτ::DA, Hout::DA, dHdτ::Union{DA,Nothing}) =
    ccall((:fpr_diffusion3d_can_step3_halo, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint),
          ctx(), p(Ht), p(Hτ), p(Hout), dHdτ === nothing ? Ptr{Cdouble}(C_NULL) : p(dHdτ), size(Ht)...) == 1

function diffusion_3D_step_τ3_halo!(Ht::DA, Hτ::DA, Hout::DA, dHdτ::Union{DA,Nothing}, dτ, _dt, _dx, _dy, _dz, D_dx, D_dy, D_dz;
                                    scale = 0.0, sumsq3::Union{DA,Nothing} = nothing, join::Bool = true)
    nx, ny, nz = size(Ht)
    check(ccall((:fpr_diffusion3d_step3_halo, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cint),
                ctx(), p(Ht), p(Hτ), p(Hout), dHdτ === nothing ? Ptr{Cdouble}(C_NULL) : p(dHdτ), nx, ny, nz, dτ, _dt, _dx, _dy, _dz,
                D_dx, D_dy, D_dz, scale, sumsq3 === nothing ? Ptr{Cdouble}(C_NULL) : p(sumsq3), join ? 1 : 0))
    return nothing
end

"Order the compute stream behind a pair that `diffusion_3D_step_τ2_halo!(...; join = false)` left on the core / comm streams."
join_pair!() = CTX[] == C_NULL ? nothing : (check(ccall((:fpr_diffusion3d_join, libfpr), Cint, (Ptr{Cvoid},), CTX[])); nothing)

"`gather!(A, A_global)` (part1_kernel_programming.jl:223): A_global (host, rank 0; `nothing` elsewhere) receives all local arrays."
function gather!(A::DA, A_global::Union{Array{Float64,3},Nothing})
    n = size(A)
    check(ccall((:fpr_gather3d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Ptr{Cdouble}), ctx(), p(A), n[1], n[2], n[3],
                A_global === nothing ? Ptr{Cdouble}(C_NULL) : pointer(A_global)))
    return nothing
end
gather!(A::Array{Float64,3}, A_global) = gather!(ROCArray(A), grid().me == 0 ? A_global : nothing)   # the reference passes Array(Ht)

"""
`apply_boundary_conditions!(H, coords, dims)` (part1_utils.jl:14-34) as written: the 0-based Cartesian `coords` are compared
with `1` and with `dims`, so on a single rank it does nothing and on rank coordinate 1 it zeroes the LOW plane -- an internal
halo plane when dims > 1 (SURVEY 7 quirk; the Python mirror reproduces it on one rank and skips it between ranks).
"""
function apply_boundary_conditions!(H::ROCArray{Float64,3}, coords, dims)
    coords[1] == 1 && (H[1, :, :] .= 0.0)
    coords[2] == 1 && (H[:, 1, :] .= 0.0)
    coords[3] == 1 && (H[:, :, 1] .= 0.0)
    coords[1] == dims[1] && (H[end, :, :] .= 0.0)
    coords[2] == dims[2] && (H[:, end, :] .= 0.0)
    coords[3] == dims[3] && (H[:, :, end] .= 0.0)
    return nothing
end

# The driver `diffusion_3D_array_programming` (part1_array_programming.jl:20-92) is NOT restated here: `include_reference(mod,
# "scripts-part1/part1_array_programming.jl")` runs the reference's own, on `diffusion_3D_step_τ!` / `update_halo!` / `dist_norm_L2` /
# `gather!` above (julia/test/runtests.jl, module Part1).

# ---- Part 2 (scripts-part2/multigrid.jl, krylov.jl, part2_utils.jl) -----------------------------------
@enum ExecutionPolicy_t serial parallel parallel_shmem          # part2_utils.jl:4-8
@enum CoarseSolver_t jacobi conjugate_gradient                  # multigrid.jl:10-13
mutable struct MGOpt                                            # multigrid.jl:16-22
    coarse_solve_size::Int
    coarse_solver::CoarseSolver_t
    execution_policy::ExecutionPolicy_t
    MGOpt() = new(5, jacobi, parallel_shmem)
end
"multigrid.jl:25-38: the level arena is owned by the library context; a token keeps the call signature."
preallocate_buffers(nx, ny) = Dict{Symbol,Any}(:nx => nx, :ny => ny)
const ARENA_REFS = Dict{Tuple{Int,Int},Any}()      # keeps buffers handed to the library alive
"""
    provide_arena!(nx, ny, tmp, tmp2)

The finest level's two ping-pong partners of an (nx, ny) hierarchy from the caller (`nothing` = the library's own again): what
`prealloc_dict` is for in multigrid.jl:49-51.  Lets a host that places its field arrays (`alloc_fields`) place these two with `u` and `f`.
"""
function provide_arena!(nx::Integer, ny::Integer, tmp::Union{DA,Nothing}, tmp2::Union{DA,Nothing})
    pt(A) = A === nothing ? Ptr{Cdouble}(C_NULL) : p(A)
    check(ccall((:fpr_mg_arena_provide, libfpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}), ctx(), nx, ny, pt(tmp), pt(tmp2)))
    ARENA_REFS[(Int(nx), Int(ny))] = (tmp, tmp2)
    return nothing
end
"`provide_arena_coarse!(nx, ny, res_c, corr_c, corr_c2)`: the three arrays of the first coarse level too (all three or `nothing` x 3)."
function provide_arena_coarse!(nx::Integer, ny::Integer, res_c::Union{DA,Nothing}, corr_c::Union{DA,Nothing}, corr_c2::Union{DA,Nothing})
    pt(A) = A === nothing ? Ptr{Cdouble}(C_NULL) : p(A)
    check(ccall((:fpr_mg_arena_provide_coarse, libfpr), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), ctx(), nx, ny,
                pt(res_c), pt(corr_c), pt(corr_c2)))
    ARENA_REFS[(-Int(nx), -Int(ny))] = (res_c, corr_c, corr_c2)
    return nothing
end
"""
    x, b = alloc_vcycle_fields(nx, ny; trial = nothing)

`x`, `b` for `MGsolve_2DPoisson!` on an (nx, ny) grid AND the arrays of the library's arena the passes over the finest grid stream beside them,
laid out the way that decides the seam pass's mode (4097^2: 99 us against 113, INTEGRATION.md 5): the two ping-pong partners of the finest
level must differ in placement class from `f` and from the first coarse level's three arrays; the partners may share a class and so may
everything else.  Two 1 GiB allocations that copy fastest among a pool (`alloc_fields(2, ...)`: different classes) -- `x`, `b` and the
coarse level are windows of the first, the partners windows of the second.  `trial(x, b) -> ms` (a timed solve) lets the library try both
orientations and the plain pair.  Grids whose seven arrays do not fit two such allocations get plain arrays.
"""
function alloc_vcycle_fields(nx::Integer, ny::Integer; trial = nothing)
    nf, ncx, ncy = nx * ny, 1 + (nx - 1) ÷ 2, 1 + (ny - 1) ÷ 2
    blk = 1 << 27                                              # doubles per allocation (1 GiB)
    step = (nf + (1 << 21)) & ~((1 << 21) - 1)                 # windows start on 16 MiB boundaries
    cstep = (ncx * ncy + (1 << 19)) & ~((1 << 19) - 1)
    (2 * step + 3 * cstep > blk || nf * 8 < (64 << 20)) && return AMDGPU.zeros(Float64, nx, ny), AMDGPU.zeros(Float64, nx, ny)
    win(B, off, m, n) = reshape(view(B, off + 1:off + m * n), m, n)
    function lay(P, Q)
        x, b = win(P, 0, nx, ny), win(P, step, nx, ny)
        cs = [win(P, 2 * step + k * cstep, ncx, ncy) for k in 0:2]
        t1, t2 = win(Q, 0, nx, ny), win(Q, step, nx, ny)
        foreach(A -> fill_device!(A, 0.0), cs)
        provide_arena!(nx, ny, t1, t2)
        provide_arena_coarse!(nx, ny, cs...)
        return x, b
    end
    judge = trial === nothing ? nothing : (arrs -> trial(lay(arrs[1], arrs[2])...))
    P, Q = alloc_fields(2, blk; pool = 12, pairs = [(1, 2)], trial = judge)
    return lay(P, Q)
end
policy_ok(pol) = (pol in (parallel, parallel_shmem)) || error()    # multigrid.jl:233-236

function residual_2DPoisson!(u::DA, f::DA, h::Float64, c::Float64, res::DA)
    nx, ny = size(u)
    check(ccall((:fpr_residual2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Cint),
                ctx(), p(u), p(f), h, c, p(res), nx, ny))
end
const residual_2DPoisson_shmem! = residual_2DPoisson!
residual_2DPoisson_wrapper!(u_f, rhs, h, c, res_f, execution_policy) = (policy_ok(execution_policy); residual_2DPoisson!(u_f, rhs, h, c, res_f))

function iteration_2DPoisson!(u::DA, f::DA, h, c, res::DA, execution_policy; alpha = 4.0 / 5.0)
    policy_ok(execution_policy)
    nx, ny = size(u); rms = Ref{Cdouble}(0)
    check(ccall((:fpr_jacobi2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Cint, Cdouble, Ptr{Cdouble}),
                ctx(), p(u), p(f), h, c, p(res), nx, ny, alpha, rms))
    return rms[]
end

function restrict_wrapper!(fine::DA, coarse::DA, apply_BCs, execution_policy)
    nx, ny = size(fine)
    check(ccall((:fpr_restrict2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint), ctx(), p(fine), p(coarse), nx, ny, apply_BCs))
end
restrict!(fine, coarse) = restrict_wrapper!(fine, coarse, false, parallel)   # note: also zeroes the coarse boundary
function prolongate_wrapper!(coarse::DA, fine::DA, apply_BCs, execution_policy)
    nx, ny = size(fine)
    check(ccall((:fpr_prolongate2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint), ctx(), p(coarse), p(fine), nx, ny, apply_BCs))
end
prolongate!(coarse, fine) = prolongate_wrapper!(coarse, fine, false, parallel)
const prolongate_with_atomic! = prolongate!
correct!(u_f::DA, corr_f::DA) = check(ccall((:fpr_axmy2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t), ctx(), p(u_f), p(corr_f), length(u_f)))

function matrix_free_matvec_prod!(T::DA, hx, hy, c, dT2::DA)
    nx, ny = size(T)
    check(ccall((:fpr_laplace_apply2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Cint), ctx(), p(T), hx, hy, c, p(dT2), nx, ny))
end
const matrix_free_matvec_prod_shmem! = matrix_free_matvec_prod!
function matrix_free_matvec_prod_wrapper!(pp, hx, hy, c, p_hat; execution_policy = parallel_shmem)
    policy_ok(execution_policy); matrix_free_matvec_prod!(pp, hx, hy, c, p_hat); @synchronize()
end

function cg!(x_in::DA, b::DA, hx, hy, c, tol, Nmax; execution_policy = parallel_shmem, verbose = false)
    nx, ny = size(b); rms = Ref{Cdouble}(0); it = Ref{Cint}(0)
    check(ccall((:fpr_cg2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cint}),
                ctx(), p(x_in), p(b), hx, hy, c, tol, Nmax, nx, ny, rms, it))
    return rms[]
end

apply_boundary_conditions!(T::DA) = check(ccall((:fpr_bc2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint), ctx(), p(T), size(T, 1), size(T, 2)))
apply_boundary_conditions_dirichlet!(T::DA) = check(ccall((:fpr_bc_dirichlet2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint), ctx(), p(T), size(T, 1), size(T, 2)))
apply_boundary_conditions_neumann!(T::DA) = check(ccall((:fpr_bc_neumann2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint), ctx(), p(T), size(T, 1), size(T, 2)))

function Vcycle_2DPoisson!(u_f::DA, rhs::DA, h::Float64, c::Float64, tol::Float64, coarse_solve_size::Int, coarse_solver::CoarseSolver_t,
                           execution_policy::ExecutionPolicy_t, apply_BCs::Bool; prealloc_dict = nothing)
    policy_ok(execution_policy)
    nx, ny = size(u_f); rms = Ref{Cdouble}(0)
    check(ccall((:fpr_vcycle2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cint, Cint, Ptr{Cdouble}),
                ctx(), p(u_f), p(rhs), h, c, tol, coarse_solve_size, Int(coarse_solver), apply_BCs, nx, ny, rms))
    return rms[]
end

function MGsolve_2DPoisson!(u::DA, f::DA, h::Float64, c::Float64, tol::Float64, niters::Int, apply_BCs::Bool; opt = MGOpt(), verbose = false, prealloc_dict = nothing)
    policy_ok(opt.execution_policy)
    nx, ny = size(u)
    rms = Ref{Cdouble}(0); ncyc = Ref{Cint}(0); frms = Ref{Cdouble}(0); conv = Ref{Cint}(0); hist = zeros(Cdouble, max(niters, 1))
    check(ccall((:fpr_mgsolve2d, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}),
                ctx(), p(u), p(f), h, c, tol, niters, apply_BCs, opt.coarse_solve_size, Int(opt.coarse_solver), nx, ny, rms, ncyc, hist, frms, conv))
    verbose && foreach(i -> println("$(i) $(hist[i] / frms[])"), 1:ncyc[])
    conv[] == 0 && @warn "V-cycle multigrid failed to converge within" niters "iterations."   # multigrid.jl:78-80
    return rms[]
end

# ---- NEXT 8f-1 (scripts-part2/part2.jl:90-137) ---------------------------------------------------------
compute_velocity!(S::DA, hx, hy, vx::DA, vy::DA) = check(ccall((:fpr_compute_velocity2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint), ctx(), p(S), hx, hy, p(vx), p(vy), size(S, 1), size(S, 2)))
compute_Ra_dTdx!(Ra, hx, T::DA, out::DA) = check(ccall((:fpr_compute_Ra_dTdx2d, libfpr), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint), ctx(), Ra, hx, p(T), p(out), size(T, 1), size(T, 2)))
compute_diffusion2d!(T::DA, hx, hy, k, dT2::DA) = check(ccall((:fpr_compute_diffusion2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cint, Cint), ctx(), p(T), hx, hy, k, p(dT2), size(T, 1), size(T, 2)))
compute_advection2d_x!(T::DA, hx, vx::DA, dTx::DA) = check(ccall((:fpr_compute_advection2d_x, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint), ctx(), p(T), hx, p(vx), p(dTx), size(T, 1), size(T, 2)))
compute_advection2d_y!(T::DA, hy, vy::DA, dTy::DA) = check(ccall((:fpr_compute_advection2d_y, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint), ctx(), p(T), hy, p(vy), p(dTy), size(T, 1), size(T, 2)))

"Pass 1 of the fused step around the V-cycle: velocity (part2.jl:190) + the three maxima of compute_dt (:76-87, :193-196)."
function velocity_and_maxima!(S::DA, hx, hy; vx::Union{DA,Nothing} = nothing, vy::Union{DA,Nothing} = nothing)
    m = zeros(Cdouble, 3)
    check(ccall((:fpr_ns_velocity_max2d, libfpr), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}),
                ctx(), p(S), hx, hy, vx === nothing ? Ptr{Cdouble}(C_NULL) : p(vx), vy === nothing ? Ptr{Cdouble}(C_NULL) : p(vy),
                size(S, 1), size(S, 2), m))
    return m[1], m[2], m[3]
end
"Pass 2: every pointwise term of part2.jl:202-214 and the right-hand sides of :220 / :225 (beta > 0) or the Euler update of :229-230."
step_rhs!(T_out::DA, W_out::DA, T::DA, W::DA, S::DA, hx, hy, Ra, Pr, k, beta, dt) =
    check(ccall((:fpr_ns_rhs2d, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cint, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble,
                 Ptr{Cdouble}, Ptr{Cdouble}),
                ctx(), p(T), p(W), p(S), hx, hy, size(T, 1), size(T, 2), Ra, Pr, k, beta, dt, p(T_out), p(W_out)))

const CTX2 = Ref{Ptr{Cvoid}}(C_NULL)
"A second context on the default context's device (own streams, own multigrid arena): the W solve of a time step runs on it."
function second_ctx()
    if CTX2[] == C_NULL
        c = ctx()
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:fpr_ctx_create, libfpr), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Ptr{Cvoid}), h, AMDGPU.device_id(AMDGPU.device()) - 1, C_NULL, C_NULL)
        rc == 0 || throw(FPRError(rc, "fpr_ctx_create failed for the second context"))
        CTX2[] = h[]
        # a solve that runs beside another one must not depend on workgroups being resident together
        for (k, v) in (("cg_fused", 2), ("mg_jacobi_persist", 0))
            ccall((:fpr_set_option, libfpr), Cint, (Ptr{Cvoid}, Cstring, Clong), CTX2[], k, v)
        end
        c == C_NULL && error("no context")
    end
    return CTX2[]
end

"""
    navier_stokes_step!(S, T, W, T_rhs, W_rhs, opt; dt_dif, coarse_solve_size = 5, coarse_solver = 0) -> (dt, info)

The loop body of `navier_stokes_2D` (part2.jl:186-226) for `opt.beta > 0` as ONE library call (`fpr_ns_step2d`): S solve, velocities
and maxima, `compute_dt`, boundary conditions of T, the pointwise terms and right-hand sides, then the T and W solves side by side
on two contexts.  Same T, W, S, dt as the piecewise calls; `info` = cycles of the S, T, W solves and their converged flags.
"""
function navier_stokes_step!(S::DA, T::DA, W::DA, T_rhs::DA, W_rhs::DA, opt; dt_dif, coarse_solve_size = 5, coarse_solver = 0)
    dt = Ref{Cdouble}(0.0)
    info = zeros(Cint, 6)
    check(ccall((:fpr_ns_step2d, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Cdouble,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cint}),
                ctx(), second_ctx(), p(S), p(T), p(W), p(T_rhs), p(W_rhs), size(T, 1), size(T, 2), opt.Ra, opt.Pr, opt.k, opt.beta,
                opt.a_adv, dt_dif, opt.tol, opt.niters, coarse_solve_size, coarse_solver, dt, info))
    return dt[], info
end

"""
    navier_stokes_run!(S, T, W, T_rhs, W_rhs, opt, sim_time, max_steps; dt_dif, ...) -> (sim_time, steps, dt_last, unconverged)

`while sim_time < opt.ttot` (part2.jl:182) around `navier_stokes_step!` inside the library (`fpr_ns_run2d`), at most `max_steps` steps.
"""
function navier_stokes_run!(S::DA, T::DA, W::DA, T_rhs::DA, W_rhs::DA, opt, sim_time, max_steps; dt_dif, coarse_solve_size = 5, coarse_solver = 0)
    t = Ref{Cdouble}(sim_time); dt = Ref{Cdouble}(0.0); steps = Ref{Cint}(0); bad = Ref{Cint}(0)
    check(ccall((:fpr_ns_run2d, libfpr), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Cdouble,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}),
                ctx(), second_ctx(), p(S), p(T), p(W), p(T_rhs), p(W_rhs), size(T, 1), size(T, 2), opt.Ra, opt.Pr, opt.k, opt.beta,
                opt.a_adv, dt_dif, opt.tol, opt.niters, coarse_solve_size, coarse_solver, opt.ttot, max_steps, t, steps, dt, bad))
    return t[], Int(steps[]), dt[], Int(bad[])
end

end # module
