# julia/test/runtests.jl -- the reference's own test suite (test/runtests.jl:6-9) run against libfpr_hip.so through FPRHip.jl.
#
#     FPR_REFERENCE_DIR=/path/to/FinalProjectRepo.jl julia --project=<env with AMDGPU, MPI, BSON, Test> julia/test/runtests.jl
#
# STATUS: like FPRHip.jl itself, unexecuted in this repository (no `julia` in the build container or on the MI355X boxes); what
# tests/test_julia_shim.py can check without a toolchain it checks: blocks balanced, every FPRHip name used here exported by the
# shim, every reference file and fixture named here present.  The Python suite (tests/test_gpu_*.py) runs the same four checks
# through the same C ABI on every round.
#
# Each part restates one file of the reference's test/ directory.  The reference's solver files are included UNEDITED through
# `include_reference` (its `using CUDA / ParallelStencil / ImplicitGlobalGrid` lines dropped, kernel definitions swallowed, the
# host functions FPRHip provides natively dropped); each part lives in a module of its own because part1_utils.jl and
# part2_utils.jl both define `apply_boundary_conditions!` with different meanings.
using Test

include(joinpath(@__DIR__, "..", "FPRHip.jl"))

const REF = get(ENV, "FPR_REFERENCE_DIR", joinpath(@__DIR__, "..", "..", ".."))
const GOLDEN = joinpath(@__DIR__, "..", "..", "tests", "golden")      # byte-identical copies of test/reftest-files/*
isdir(joinpath(REF, "scripts-part1")) || error("set FPR_REFERENCE_DIR to a checkout of ntselepidis/FinalProjectRepo.jl")

# ---------------------------------------------------------------------------------------------------------------------
# test/part1.jl:24-40 -- the three 3D diffusion drivers at 32^3 against reftest-files/test_1.bson (atol 1e-5)
# ---------------------------------------------------------------------------------------------------------------------
module Part1
using Test, BSON
import MPI
using ..FPRHip
import ..REF, ..GOLDEN
const USE_GPU = true
@init_parallel_stencil(AMDGPU, Float64, 3)
include_reference(@__MODULE__, joinpath(REF, "scripts-part1", "part1_kernel_programming.jl"))
include_reference(@__MODULE__, joinpath(REF, "scripts-part1", "part1_array_programming.jl"))

comp(d1, d2) = keys(d1) == keys(d2) && all([isapprox(d1[k], d2[k], atol = 1e-5) for k in keys(d1)])

function run()
    MPI.Init()
    ref = BSON.load(joinpath(GOLDEN, "test_1.bson"))
    ref = Dict(:X => ref[:X], :H => ref[:H])
    Xc_g, H_g = diffusion_3D_array_programming(nx = 32, ny = 32, nz = 32, init_and_finalize_MPI = false, verbose = false)
    inds = Int.(ceil.(LinRange(1, length(Xc_g), 12)))
    d_array = Dict(:X => Xc_g[inds], :H => H_g[inds, inds, 15])
    Xc_g, H_g, _ = diffusion_3D_kernel_programming(nx = 32, ny = 32, nz = 32, use_shared_memory = true, init_and_finalize_MPI = false, verbose = false)
    d_kernel_shared_memory = Dict(:X => Xc_g[inds], :H => H_g[inds, inds, 15])
    Xc_g, H_g, _ = diffusion_3D_kernel_programming(nx = 32, ny = 32, nz = 32, use_shared_memory = false, init_and_finalize_MPI = false, verbose = false)
    d_kernel = Dict(:X => Xc_g[inds], :H => H_g[inds, inds, 15])
    if MPI.Comm_rank(MPI.COMM_WORLD) == 0
        @testset "Ref-file" begin
            @test comp(ref, d_array)
            @test comp(ref, d_kernel)
            @test comp(ref, d_kernel_shared_memory)
        end
    end
    @reset_parallel_stencil()
end
end

# ---------------------------------------------------------------------------------------------------------------------
# test/multigrid.jl:30-138 -- MGsolve over policies x coarse sizes x grids x coarse solvers, the Jacobi iteration, the residual
# against the sparse 5-point matrix
# ---------------------------------------------------------------------------------------------------------------------
module Multigrid
using Test, LinearAlgebra
using ..FPRHip
import ..REF
const USE_GPU = true
@init_parallel_stencil(AMDGPU, Float64, 2)
include_reference(@__MODULE__, joinpath(REF, "scripts-part2", "multigrid.jl"))

function run()
    @testset "Test backslash" begin
        nx = 6; ny = 6
        h = 1 / (ny - 1)
        x = rand((nx - 2) * (ny - 2))
        A = stencil_5pt(nx - 2, ny - 2) / h^2
        xhat = A \ (A * x)
        @test norm(x - xhat) / norm(x) < 1e-10
    end
    @testset "Test Multigrid with policy=$(execution_policy) and coarse_solver=$(solver) and coarse_solve_size = $((2^l)+1) and nx=ny=$((2^k)+1) on MI355X." for execution_policy in
            [parallel, parallel_shmem], l in 2:3, k in 7:10, solver in [jacobi, conjugate_gradient]
        n = (2^k) + 1
        h = 1 / (n - 1); c = 0.0; tol = 1e-6
        opt = MGOpt()
        opt.execution_policy = execution_policy
        opt.coarse_solve_size = (2^l) + 1
        opt.coarse_solver = solver
        inn = CartesianIndices((2:n-1, 2:n-1))
        xref = zeros(n, n)
        xref[inn] .= rand(n - 2, n - 2)
        b = zeros(n, n)
        b_ = zeros(n - 2, n - 2)
        A = stencil_5pt(n - 2, n - 2) / h^2
        b_[:] .= A * xref[inn][:]
        b[inn] .= b_
        xhat = @zeros(n, n)
        r_rms = MGsolve_2DPoisson!(xhat, Data.Array(b), h, c, tol, 20, false; opt = opt, verbose = false)
        @synchronize()
        @test r_rms < tol * sqrt(sum(b .^ 2) / (n * n))
    end
    @testset "Test Jacobi solver" for execution_policy in [parallel, parallel_shmem]
        n = 33
        h = 1 / (n - 1); c = 0.0; tol = 1e-6; Nmax = 10000
        inn = CartesianIndices((2:n-1, 2:n-1))
        xref = rand(n, n)
        xref[1, :] .= 0.0; xref[n, :] .= 0.0; xref[:, 1] .= 0.0; xref[:, n] .= 0.0
        b = zeros(n, n)
        b_ = zeros(n - 2, n - 2)
        A = stencil_5pt(n - 2, n - 2) / h^2
        b_[:] .= A * xref[inn][:]
        b[inn] .= b_
        tolb = tol * sqrt(sum(b .^ 2) / (n * n))
        res_buf = @zeros(n, n)
        bd = Data.Array(b)
        xhat = @zeros(n, n)
        for i = 1:Nmax
            res_rms = iteration_2DPoisson!(xhat, bd, h, c, res_buf, execution_policy)
            @synchronize()
            res_rms < tolb && break
        end
        @test norm(xref - Array(xhat)) / norm(xref) < tolb
    end
    @testset "Test residual_2DPoisson" for execution_policy in [parallel, parallel_shmem]
        n = 64
        h = 1 / (n - 1); c = 3.1415
        inn = CartesianIndices((2:n-1, 2:n-1))
        u_cpu = rand(n, n)
        u_cpu[1, :] .= 0.0; u_cpu[n, :] .= 0.0; u_cpu[:, 1] .= 0.0; u_cpu[:, n] .= 0.0
        f_cpu = rand(n, n)
        u = Data.Array(u_cpu); f = Data.Array(f_cpu)
        res = @zeros(n, n)
        residual_2DPoisson_wrapper!(u, f, h, c, res, execution_policy)
        @synchronize()
        A = stencil_5pt(n - 2, n - 2) / h^2 - c * I
        res_ = zeros(n - 2, n - 2)
        res_[:] .= A * u_cpu[inn][:] - f_cpu[inn][:]
        @test res_ ≈ Array(res)[inn]
    end
    @reset_parallel_stencil()
end
end

# ---------------------------------------------------------------------------------------------------------------------
# test/krylov.jl:19-36 -- cg! on the 66 x 66 Helmholtz problem under both execution policies
# ---------------------------------------------------------------------------------------------------------------------
module Krylov
using Test, LinearAlgebra
using ..FPRHip
import ..REF
const USE_GPU = true
@init_parallel_stencil(AMDGPU, Float64, 2)
include_reference(@__MODULE__, joinpath(REF, "scripts-part2", "krylov.jl"))

function run()
    @testset "Test Krylov solver" for execution_policy in [parallel, parallel_shmem]
        n = 66
        h = 1 / (n - 1); c = 3.14; tol = 1e-6; Nmax = 1000
        b_cpu = ones(n, n)
        b_cpu[1, :] .= 0.0; b_cpu[n, :] .= 0.0; b_cpu[:, 1] .= 0.0; b_cpu[:, n] .= 0.0
        b = Data.Array(b_cpu)
        xhat = @zeros(n, n)
        res_rms = cg!(xhat, b, h, h, c, tol, Nmax, execution_policy = execution_policy)
        @test res_rms < tol * sqrt(sum(b_cpu .^ 2) / n^2)
    end
    @reset_parallel_stencil()
end
end

# ---------------------------------------------------------------------------------------------------------------------
# test/part2.jl:4-38 -- the Navier-Stokes driver at 257 x 65 against the FORTRAN fields T / W / S (sizes, then values to 1e-8)
# ---------------------------------------------------------------------------------------------------------------------
module Part2
using Test
using ..FPRHip
import ..REF, ..GOLDEN
const USE_GPU = true
include_reference(@__MODULE__, joinpath(REF, "scripts-part2", "part2.jl"))

comp_size(lhs, rhs) = size(lhs) == size(rhs)
comp_val(nx, ny; atol = 1e-8) = (lhs, rhs) -> all(abs.(lhs[2:nx-1, 2:ny-1] - rhs[2:nx-1, 2:ny-1]) .< atol)
fixture(name) = open(load, joinpath(GOLDEN, "fortran", name), "r")       # `load`: part2_utils.jl:11-19 (Int32 nx, ny, then Float64 column-major)

function run()
    opt = SimIn_t()
    opt.nx, opt.ny = 257, 65
    opt.tol = 1.0e-12
    opt.W_init_strategy = W_from_file
    # part2.jl reads reftest-files/fortran/Winit.bin relative to its own directory: run from the reference's test/ directory
    sim_results = cd(() -> navier_stokes_2D(; opt = opt, verbose = false, do_vis = false, testmode = true), joinpath(REF, "test"))
    @testset "Test against FORTRAN reference implementation: array sizes." begin
        @test comp_size(fixture("T.bin"), Array(sim_results.T))
        @test comp_size(fixture("W.bin"), Array(sim_results.W))
        @test comp_size(fixture("S.bin"), Array(sim_results.S))
    end
    @testset "Test against FORTRAN reference implementation: array values." begin
        cv = comp_val(opt.nx, opt.ny, atol = 1e-8)
        @test cv(fixture("T.bin"), Array(sim_results.T))
        @test cv(fixture("W.bin"), Array(sim_results.W))
        @test cv(fixture("S.bin"), Array(sim_results.S))
    end
    @reset_parallel_stencil()
end
end

Part1.run()
Multigrid.run()
Krylov.run()
Part2.run()
