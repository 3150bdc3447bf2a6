# NS3DShim.jl — reference-side binding of libns3d.so (include/ns3d.h).
#
# What it replaces, so that the BODIES of the reference drivers run as written on an MI355X
# (scripts/NavierStokes3D_gpu.jl:12-173 `runme`, scripts/NavierStokes3D_multi_gpu.jl:287-536 `run_navierstokes3D`):
#
#   ParallelStencil.jl        @init_parallel_stencil, @parallel, @parallel_indices, @zeros, Data.Array / Data.Number
#                             (gpu.jl:2-8, multi.jl:2-8, :343-360, :370; every `@parallel kernel!(…)` call site)
#   the 14/16 kernel bodies   multi.jl:36-281 / gpu.jl:177-368 — same names, same positional arguments, each one `ccall`
#   ImplicitGlobalGrid.jl     init_global_grid, nx_g/ny_g/nz_g, x_g/y_g/z_g, update_halo!, gather!, finalize_global_grid
#                             (multi.jl:9, :325, :328-338, :363-367, :371…, :399-403, :534); the topology is
#                             MPI_Dims_create's unless dimx/dimy/dimz are given, as in ImplicitGlobalGrid
#
# What changes in a script: ONLY its header.  Lines multi.jl:1-13 (resp. gpu.jl:1-10) become
#
#     include("julia/NS3DShim.jl"); using .NS3DShim
#     @init_parallel_stencil(AMDGPU, Float64, 3)
#     import MPI                      # multi.jl only: max_g (multi.jl:21) keeps its MPI.Allreduce
#     using Printf                    # + Plots/MAT only if do_vis/do_save are used
#
# and everything from multi.jl:15 / gpu.jl:12 on is included unchanged: the scripts' own `@parallel function …` and
# `@parallel_indices … function …` DEFINITIONS are swallowed by the macros below (they expand to `nothing`, so their
# @all/@inn/@d_xa bodies are never looked at and cannot shadow the ccall methods exported here), plain helper functions
# (backtrack!, lerp, set_bc_Vel!, set_bc_Pr!, save_array, max_g) are defined as written, and every `@parallel kernel!(…)`
# / `@parallel (ranges…) kernel!(…)` CALL becomes a call of the exported method (the library derives the launch range
# from the array sizes itself).
#
# STATUS: NOT EXECUTED.  There is no Julia toolchain in the build container or on the GPU box (SURVEY.md §8c), so this
# file has been checked on paper only, against multi.jl:1-13,36-102,325-373 and gpu.jl:1-10,175-368.  Every C entry point
# it binds is exercised through the identical C ABI by the Python/ctypes host layer (navierstokes3d_amd/kernels.py,
# mgpu.py) and its GPU parity tests.  `init_global_grid(nx,ny,nz; dimx=1, dimy=1)` gives the z-slabs the fused
# `pt_solve_slab!` needs; without keywords the topology is ImplicitGlobalGrid's default (results depend on it: damp uses the
# local nx, multi.jl:340, and advect! clamps at local array ends).
# Arrays are AMDGPU.jl `ROCArray{T,3}` (packed, column-major: the layout ns3d.h requires), passed as device pointers.
module NS3DShim

using AMDGPU
import MPI

export @init_parallel_stencil, @parallel, @parallel_indices, @zeros, Data
export update_τ!, predict_V!, set_cylinder!, update_∇V!, update_dPrdτ!, update_Pr!, compute_res!, correct_V!, advect!
export bc_x!, bc_y!, bc_z!, bc_zV!, bc_xhydstatic!, bc_x_Vx!, bc_x_Pr!, bc_xVx!, bc_xVyz!
export init_global_grid, finalize_global_grid, nx_g, ny_g, nz_g, x_g, y_g, z_g, update_halo!, gather!
export pt_solve!, pt_solve_slab!, maxabs, copy_advect!, predict_fused!, poisson_direct!, time_step!, reserve_cus!, StepFields, StepParams

const libns3d = get(ENV, "NS3D_LIB", joinpath(@__DIR__, "..", "navierstokes3d_amd", "libns3d.so"))
const NS3D_STRICT, NS3D_FAST, NS3D_ASYNC = Cint(0), Cint(1), Cint(2)
const NS3D_UNIQUE_ID_BYTES = 128

struct Ns3dError <: Exception
    msg::String
end
lasterror() = unsafe_string(ccall((:ns3d_last_error, libns3d), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : throw(Ns3dError("libns3d status $rc: $(lasterror())"))

# ---- state: one kernel context per process (= per rank, as in the reference) ---------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)      # ns3d_ctx*  (owned here, or by MGPU[] once init_global_grid has run)
const MGPU = Ref{Ptr{Cvoid}}(C_NULL)     # ns3d_mgpu*
const OWNS_CTX = Ref(false)
const MODE = Ref(NS3D_STRICT)
const GRID = Ref((nx = 0, ny = 0, nz = 0, me = 0, dims = (1, 1, 1), coords = (0, 0, 0)))

"The element type chosen by `@init_parallel_stencil` (ParallelStencil's `Data` module)."
module Data
    using AMDGPU
    const Number = Float64
    const Array = AMDGPU.ROCArray{Float64}
end

function _init_ctx(device::Integer)
    CTX[] != C_NULL && return nothing
    flags = MODE[]                      # blocking calls, like `@parallel` (no NS3D_ASYNC)
    CTX[] = ccall((:ns3d_create, libns3d), Ptr{Cvoid}, (Cint, Cint), device, flags)
    CTX[] == C_NULL && throw(Ns3dError("ns3d_create failed: $(lasterror())"))
    OWNS_CTX[] = true
    return nothing
end

"""
    @init_parallel_stencil(backend, Float64, 3)

Replaces ParallelStencil's macro (gpu.jl:4-8, multi.jl:4-8).  The backend symbol is ignored (there is one backend: HIP on
gfx950); the element type must be Float64 (`_f64` entry points; an `_f32` shim is the same file with Float32/`_f32`).
`ENV["NS3D_MODE"] = "fast"` selects reciprocal constants + FMA instead of the bit-exact STRICT arithmetic.
"""
macro init_parallel_stencil(backend, T, ndims)
    quote
        $(esc(T)) === Float64 || error("NS3DShim binds the _f64 entry points")
        $(esc(ndims)) == 3 || error("NS3DShim is 3-D")
        NS3DShim.MODE[] = get(ENV, "NS3D_MODE", "strict") == "fast" ? NS3DShim.NS3D_FAST : NS3DShim.NS3D_STRICT
        nothing
    end
end

"`@zeros(nx,ny,nz)` (multi.jl:343-360): a zero-filled device array of the stencil's element type."
macro zeros(dims...)
    :(AMDGPU.zeros(Float64, $(map(esc, dims)...)))
end

_is_definition(ex) = ex isa Expr && (ex.head === :function || (ex.head === :(=) && ex.args[1] isa Expr && ex.args[1].head in (:call, :where)))

"""
    @parallel kernel!(args…)   /   @parallel (ranges…) kernel!(args…)   /   @parallel function kernel!(…) … end

A call is forwarded to the method exported by this module (the ranges are dropped: libns3d derives the launch range from
the array extents).  A DEFINITION — the scripts' own kernels, multi.jl:36-102 — expands to `nothing`.
"""
macro parallel(args...)
    ex = args[end]
    _is_definition(ex) && return nothing
    return esc(ex)
end
"`@parallel_indices (ix,iy,iz) function kernel!(…) … end` (multi.jl:108-150,217-281): definitions only → `nothing`."
macro parallel_indices(args...)
    return nothing
end

# every ccall below is preceded by AMDGPU.synchronize(): array operations of the script (broadcasts, uploads) run on
# AMDGPU.jl's stream, the kernels on the context's own stream and block until done — the two never overlap, exactly the
# "synchronous to the caller" behaviour of `@parallel` in the reference [upstream].
_sync() = AMDGPU.synchronize()
_ctx() = (CTX[] == C_NULL && _init_ctx(AMDGPU.device_id(AMDGPU.device()) - 1); CTX[])
const PF = Ptr{Float64}
ptr(A) = convert(PF, pointer(A))
_cint3(A) = (Cint(size(A, 1)), Cint(size(A, 2)), Cint(size(A, 3)))

# ---- kernels: names and argument order of multi.jl:36-281 / gpu.jl:177-368 --------------------------------------
"update_τ!  multi.jl:36-44 / gpu.jl:177-185"
function update_τ!(τxx, τyy, τzz, τxy, τxz, τyz, Vx, Vy, Vz, μ, dx, dy, dz)
    nx, ny, nz = _cint3(τxx); _sync()
    check(ccall((:ns3d_update_tau_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF, PF, PF, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(τxx), ptr(τyy), ptr(τzz), ptr(τxy), ptr(τxz), ptr(τyz), ptr(Vx), ptr(Vy), ptr(Vz),
                μ, dx, dy, dz, nx, ny, nz))
end
"predict_V!  multi.jl:50-55 / gpu.jl:187-192"
function predict_V!(Vx, Vy, Vz, τxx, τyy, τzz, τxy, τxz, τyz, ρ, g, dt, dx, dy, dz)
    nx, ny, nz = _cint3(τxx); _sync()
    check(ccall((:ns3d_predict_V_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF, PF, PF, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble,
                 Cint, Cint, Cint),
                _ctx(), ptr(Vx), ptr(Vy), ptr(Vz), ptr(τxx), ptr(τyy), ptr(τzz), ptr(τxy), ptr(τxz), ptr(τyz),
                ρ, g, dt, dx, dy, dz, nx, ny, nz))
end
"set_cylinder!, multi.jl form (19 arguments)  multi.jl:249-281"
function set_cylinder!(C, Vx, Vy, Vz, a2, b2, ox, oy, sinβ, cosβ, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz)
    nx, ny, nz = _cint3(C); _sync()
    check(ccall((:ns3d_set_cylinder_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble,
                 Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(C), ptr(Vx), ptr(Vy), ptr(Vz),
                a2, b2, ox, oy, sinβ, cosβ, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz, nx, ny, nz))
end
"set_cylinder!, gpu.jl form (16 arguments, incl. its `yc = yv + dx/2`)  gpu.jl:336-368"
function set_cylinder!(C, Vx, Vy, Vz, a2, b2, ox, oy, sinβ, cosβ, lx, ly, lz, dx, dy, dz)
    nx, ny, nz = _cint3(C); _sync()
    check(ccall((:ns3d_set_cylinder_local_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF,
                 Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble,
                 Cint, Cint, Cint),
                _ctx(), ptr(C), ptr(Vx), ptr(Vy), ptr(Vz),
                a2, b2, ox, oy, sinβ, cosβ, lx, ly, lz, dx, dy, dz, nx, ny, nz))
end
"update_∇V!  multi.jl:61-64 / gpu.jl:194-197"
function update_∇V!(∇V, Vx, Vy, Vz, dx, dy, dz)
    nx, ny, nz = _cint3(∇V); _sync()
    check(ccall((:ns3d_update_divV_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(∇V), ptr(Vx), ptr(Vy), ptr(Vz), dx, dy, dz, nx, ny, nz))
end
"update_dPrdτ!  multi.jl:70-73 / gpu.jl:199-202"
function update_dPrdτ!(Pr, dPrdτ, ∇V, ρ, dt, dτ, damp, dx, dy, dz)
    nx, ny, nz = _cint3(Pr); _sync()
    check(ccall((:ns3d_update_dPrdtau_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(Pr), ptr(dPrdτ), ptr(∇V), ρ, dt, dτ, damp, dx, dy, dz, nx, ny, nz))
end
"update_Pr!  multi.jl:79-82 / gpu.jl:204-207"
function update_Pr!(Pr, dPrdτ, dτ)
    nx, ny, nz = _cint3(Pr); _sync()
    check(ccall((:ns3d_update_Pr_f64, libns3d), Cint, (Ptr{Cvoid}, PF, PF, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(Pr), ptr(dPrdτ), dτ, nx, ny, nz))
end
"compute_res!  multi.jl:88-91 / gpu.jl:209-212"
function compute_res!(Rp, Pr, ∇V, ρ, dt, dx, dy, dz)
    nx, ny, nz = _cint3(Pr); _sync()
    check(ccall((:ns3d_compute_res_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(Rp), ptr(Pr), ptr(∇V), ρ, dt, dx, dy, dz, nx, ny, nz))
end
"correct_V!  multi.jl:97-102 / gpu.jl:214-219"
function correct_V!(Vx, Vy, Vz, Pr, dt, ρ, dx, dy, dz)
    nx, ny, nz = _cint3(Pr); _sync()
    check(ccall((:ns3d_correct_V_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(Vx), ptr(Vy), ptr(Vz), ptr(Pr), dt, ρ, dx, dy, dz, nx, ny, nz))
end
"advect!  multi.jl:217-243 / gpu.jl:308-334 (faithful: third branch back-tracks Vy, Vz is never advected)"
function advect!(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, dt, dx, dy, dz; faithful::Bool = true)
    nx, ny, nz = _cint3(C); _sync()
    check(ccall((:ns3d_advect_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF, PF, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cint),
                _ctx(), ptr(Vx), ptr(Vx_o), ptr(Vy), ptr(Vy_o), ptr(Vz), ptr(Vz_o), ptr(C), ptr(C_o),
                dt, dx, dy, dz, nx, ny, nz, faithful ? 1 : 0))
end
"""
    copy_advect!(Vx_new, Vx, Vy_new, Vy, Vz_new, Vz, C_new, C, dt, dx, dy, dz; faithful = true)

Optional, same results as `Vx_o .= Vx; …; advect!(…)` (multi.jl:475-476 / gpu.jl:141-142) without the four copies: reads the CURRENT
fields, writes COMPLETE new fields into buffers of their own (the entries `advect!` leaves alone written through); the caller swaps
the names afterwards (`Vx, Vx_o = Vx_o, Vx` …).  `Vz_new` may be `Vz` itself in faithful mode (Vz is never advected there).
"""
function copy_advect!(Vx_new, Vx, Vy_new, Vy, Vz_new, Vz, C_new, C, dt, dx, dy, dz; faithful::Bool = true)
    nx, ny, nz = _cint3(C); _sync()
    check(ccall((:ns3d_copy_advect_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF, PF, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cint),
                _ctx(), ptr(Vx_new), ptr(Vx), ptr(Vy_new), ptr(Vy), ptr(Vz_new), ptr(Vz), ptr(C_new), ptr(C),
                dt, dx, dy, dz, nx, ny, nz, faithful ? 1 : 0))
end

"""
    predict_fused!(Vx_new, Vy_new, Vz_new, Vx, Vy, Vz, μ, ρ, g, dt, dx, dy, dz)

Optional, same results as `update_τ!(τ…, Vx, Vy, Vz, μ, …); predict_V!(Vx, Vy, Vz, τ…, ρ, g, dt, …)` (multi.jl:449,451 /
gpu.jl:121-122) for a driver that does not look at the stress arrays: reads the velocities, writes COMPLETE predicted fields into
buffers of their own (e.g. `Vx_o, Vy_o, Vz_o`, idle at that point of the time step); the caller swaps the names afterwards.  The
stresses are evaluated on the fly and not stored; multi.jl:450's `update_halo!(τxx,τyy,τzz)` goes with them (every rank computes
those values itself from velocities that are consistent after multi.jl:477).
"""
function predict_fused!(Vx_new, Vy_new, Vz_new, Vx, Vy, Vz, μ, ρ, g, dt, dx, dy, dz)
    nx, ny, nz = size(Vx, 1) - 1, size(Vx, 2), size(Vx, 3); _sync()
    check(ccall((:ns3d_predict_fused_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, PF, PF, PF, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(Vx_new), ptr(Vy_new), ptr(Vz_new), ptr(Vx), ptr(Vy), ptr(Vz), μ, ρ, g, dt, dx, dy, dz, nx, ny, nz))
end

# boundary-plane kernels: the array's own extents are passed (they act on Pr, Vx, Vy and Vz alike)
"bc_x!  multi.jl:108-112 / gpu.jl:221-225"
bc_x!(A) = (s = _cint3(A); _sync(); check(ccall((:ns3d_bc_x_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Cint, Cint, Cint), _ctx(), ptr(A), s[1], s[2], s[3])))
"bc_y!  multi.jl:118-122 / gpu.jl:227-231"
bc_y!(A) = (s = _cint3(A); _sync(); check(ccall((:ns3d_bc_y_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Cint, Cint, Cint), _ctx(), ptr(A), s[1], s[2], s[3])))
"bc_z!  multi.jl:128-132 / gpu.jl:233-237"
bc_z!(A) = (s = _cint3(A); _sync(); check(ccall((:ns3d_bc_z_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Cint, Cint, Cint), _ctx(), ptr(A), s[1], s[2], s[3])))
"bc_zV!  gpu.jl:239-243"
bc_zV!(A) = (s = _cint3(A); _sync(); check(ccall((:ns3d_bc_zV_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Cint, Cint, Cint), _ctx(), ptr(A), s[1], s[2], s[3])))
"bc_xhydstatic!  gpu.jl:257-261"
function bc_xhydstatic!(A, dz, nz, g, ρ)
    s = _cint3(A); _sync()
    check(ccall((:ns3d_bc_xhydstatic_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Cdouble, Cint, Cdouble, Cdouble, Cint, Cint, Cint),
                _ctx(), ptr(A), dz, nz, g, ρ, s[1], s[2], s[3]))
end
"bc_x_Vx!  multi.jl:138-141"
bc_x_Vx!(A, V) = (s = _cint3(A); _sync(); check(ccall((:ns3d_bc_x_Vx_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Cdouble, Cint, Cint, Cint), _ctx(), ptr(A), V, s[1], s[2], s[3])))
"bc_x_Pr!  multi.jl:147-150"
bc_x_Pr!(A, val) = (s = _cint3(A); _sync(); check(ccall((:ns3d_bc_x_Pr_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Cdouble, Cint, Cint, Cint), _ctx(), ptr(A), val, s[1], s[2], s[3])))
# gpu.jl:245-255 define bc_xVx! / bc_xVyz!, whose only call sites are commented out (gpu.jl:266,270,274): dead code in
# the reference, bound here to an error so that reviving those lines fails loudly instead of silently doing nothing
bc_xVx!(args...) = error("bc_xVx! is dead code in the reference (gpu.jl:266) and has no libns3d entry point")
bc_xVyz!(args...) = error("bc_xVyz! is dead code in the reference (gpu.jl:270,274) and has no libns3d entry point")

"`maximum(abs.(A))` on the device without the temporary (NaN-propagating like Julia's `maximum`; multi.jl:466, gpu.jl:132)"
function maxabs(A)
    out = Ref{Cdouble}(0.0); _sync()
    check(ccall((:ns3d_max_abs_f64, libns3d), Cint, (Ptr{Cvoid}, PF, Clong, Ref{Cdouble}), _ctx(), ptr(A), length(A), out))
    return out[]
end

# ---- the fused inner loop (optional; same iterates as multi.jl:458-471 / gpu.jl:126-137) ---------------------------
struct PtParams                 # struct ns3d_pt_params (include/ns3d.h), field for field
    rho::Cdouble; dt::Cdouble; dtau::Cdouble; damp::Cdouble
    dx::Cdouble; dy::Cdouble; dz::Cdouble
    nx::Cint; ny::Cint; nz::Cint
    bc_kind::Cint; owns_outlet::Cint
    outlet_val::Cdouble; g::Cdouble
    z_lo_is_halo::Cint; z_hi_is_halo::Cint
end
"""
    pt_solve!(Pr, dPrdτ, ∇V, ρ, dt, dτ, damp, dx, dy, dz; bc_kind, owns_outlet, g, εit, niter, nchk, ly, psc) -> (iters, errs)

One rank: `ns3d_pt_solve_f64` — fused sweeps (two to four PT iterations per pass over memory), residual check every `nchk`
iterations with `err = max|Rp|·ly²/psc`, exit on `err < εit || !isfinite(err)`.  `bc_kind = 0`: multi.jl's set_bc_Pr!
(:175-181), `1`: gpu.jl's (:281-286).
"""
function pt_solve!(Pr, dPrdτ, ∇V, ρ, dt, dτ, damp, dx, dy, dz; bc_kind = 0, owns_outlet = true, g = 0.0, εit = 1e-3,
                   niter, nchk, ly, psc)
    nx, ny, nz = _cint3(Pr)
    p = Ref(PtParams(ρ, dt, dτ, damp, dx, dy, dz, nx, ny, nz, bc_kind, owns_outlet ? 1 : 0, 0.0, g, 0, 0))
    cap = niter ÷ max(nchk, 1) + 1
    hist = Vector{Cdouble}(undef, cap); it = Ref{Cint}(0); nchecks = Ref{Cint}(0); _sync()
    check(ccall((:ns3d_pt_solve_f64, libns3d), Cint,
                (Ptr{Cvoid}, PF, PF, PF, Ref{PtParams}, Cdouble, Cint, Cint, Cdouble, Cdouble, Ref{Cint}, Ptr{Cdouble}, Cint,
                 Ref{Cint}),
                _ctx(), ptr(Pr), ptr(dPrdτ), ptr(∇V), p, εit, niter, nchk, ly^2, psc, it, hist, cap, nchecks))
    return Int(it[]), hist[1:nchecks[]]
end
"""
    poisson_direct!(Pr, dPrdτ, ∇V, ρ, dt, dx, dy, dz; bc_kind, owns_outlet, g)

OUTSIDE PARITY (an option the reference does not have): instead of iterating multi.jl:458-471 / gpu.jl:126-137 to `err < εit`, solve
the discrete problem that loop converges to — `∇²Pr = ρ/dt·∇V` with set_bc_Pr!'s boundary cells — directly (`ns3d_poisson_direct_f64`:
exact diagonalisation of the box Laplacian, six fp64 matrix products on the matrix cores).  `Pr` gets the solution and its boundary
cells, `dPrdτ` zeros.  One rank.  255×153×153: ≈ 0.6 ms against ≈ 60 ms for the loop's 2 280 iterations.
"""
function poisson_direct!(Pr, dPrdτ, ∇V, ρ, dt, dx, dy, dz; bc_kind = 0, owns_outlet = true, g = 0.0)
    nx, ny, nz = _cint3(Pr)
    p = Ref(PtParams(ρ, dt, 0.0, 0.0, dx, dy, dz, nx, ny, nz, bc_kind, owns_outlet ? 1 : 0, 0.0, g, 0, 0)); _sync()
    check(ccall((:ns3d_poisson_direct_f64, libns3d), Cint, (Ptr{Cvoid}, PF, PF, PF, Ref{PtParams}),
                _ctx(), ptr(Pr), ptr(dPrdτ), ptr(∇V), p))
end
mutable struct StepFields          # struct ns3d_step_fields (include/ns3d.h): device pointers, IN/OUT (the fused step swaps X and X_o)
    Pr::PF; dPrdtau::PF; divV::PF
    Vx::PF; Vy::PF; Vz::PF; Vx_o::PF; Vy_o::PF; Vz_o::PF; C::PF; C_o::PF
    txx::PF; tyy::PF; tzz::PF; txy::PF; txz::PF; tyz::PF
end
struct StepParams                  # struct ns3d_step_params (include/ns3d.h), field for field
    script::Cint; nx::Cint; ny::Cint; nz::Cint
    mu::Cdouble; rho::Cdouble; g::Cdouble; dt::Cdouble; dtau::Cdouble; damp::Cdouble; dx::Cdouble; dy::Cdouble; dz::Cdouble
    eps::Cdouble; niter::Cint; nchk::Cint; err_mul::Cdouble; err_div::Cdouble
    a2::Cdouble; b2::Cdouble; ox::Cdouble; oy::Cdouble; sinb::Cdouble; cosb::Cdouble
    xco_g::Cdouble; yco_g::Cdouble; zco_g::Cdouble
    lx::Cdouble; ly::Cdouble; lz::Cdouble
    owns_inlet::Cint; owns_outlet::Cint; vin::Cdouble
    faithful::Cint; pressure::Cint; write_stress::Cint
end
"""
    time_step!(f::StepFields, p::StepParams) -> (iters, errs)

The whole time step multi.jl:449-477 (one rank; `p.script = 0`) or gpu.jl:121-142 (`p.script = 1`) in ONE library call
(`ns3d_time_step_f64`): predictor, set_cylinder!, update_∇V!, the pressure loop with its residual read-backs, correct_V!,
set_cylinder!, set_bc_Vel!, {X_o .= X; advect!} in their fused forms, enqueued from C.  `f` holds the device pointers of the
reference's arrays and is updated in place: the step swaps the roles of `Vx`/`Vx_o`, `Vy`/`Vy_o`, `C`/`C_o` (and `Vz`/`Vz_o` unless
`faithful`) instead of copying — re-wrap your arrays from `f` afterwards (or keep working through `f`).
"""
function time_step!(f::StepFields, p::StepParams)
    cap = p.niter ÷ max(p.nchk, 1) + 1
    hist = Vector{Cdouble}(undef, cap); it = Ref{Cint}(0); nchecks = Ref{Cint}(0); _sync()
    check(ccall((:ns3d_time_step_f64, libns3d), Cint,
                (Ptr{Cvoid}, Ref{StepFields}, Ref{StepParams}, Ref{Cint}, Ptr{Cdouble}, Cint, Ref{Cint}),
                _ctx(), f, Ref(p), it, hist, cap, nchecks))
    return Int(it[]), hist[1:nchecks[]]
end
"""
    reserve_cus!(n)

Leave `n` compute units out of this context's launches (`ns3d_reserve_cus`): room for RCCL's send/recv kernels beside a sweep that
would otherwise hold every CU — the device-side counterpart of the `b_width` the reference reserves "for comm / comp overlap"
(multi.jl:326) and never uses.  Results do not depend on it.
"""
reserve_cus!(n::Integer) = check(ccall((:ns3d_reserve_cus, libns3d), Cint, (Ptr{Cvoid}, Cint), _ctx(), n))
"""
    pt_solve_slab!(Pr, dPrdτ, ∇V, ρ, dt, dτ, damp, dx, dy, dz; …) -> (iters, errs)

The same loop on a rank of the global grid after `init_global_grid` (`ns3d_pt_solve_slab_f64`; `owns_outlet` = the GLOBAL
x-hi face carries the outlet rule, the library applies it on the ranks that hold it).  On a topology decomposed in x or y (the
default of `init_global_grid`): deep ghosts in every decomposed direction, up to four iterations per pass, ghost layers exchanged
x, y, z in turn.  On z-slabs (`init_global_grid(…; dimx=1, dimy=1)`): two ghost planes per seam, seam planes
swept first, their RCCL exchange behind the interior sweep, global residual by ncclAllReduce.  Collective over the ranks.
"""
function pt_solve_slab!(Pr, dPrdτ, ∇V, ρ, dt, dτ, damp, dx, dy, dz; owns_outlet = true, g = 0.0, εit = 1e-3, niter, nchk, ly, psc)
    MGPU[] == C_NULL && error("pt_solve_slab! needs init_global_grid")
    nx, ny, nz = _cint3(Pr)
    p = Ref(PtParams(ρ, dt, dτ, damp, dx, dy, dz, nx, ny, nz, 0, owns_outlet ? 1 : 0, 0.0, g, 0, 0))
    cap = niter ÷ max(nchk, 1) + 1
    hist = Vector{Cdouble}(undef, cap); it = Ref{Cint}(0); nchecks = Ref{Cint}(0)
    P_, D_, R_ = [ptr(Pr)], [ptr(dPrdτ)], [ptr(∇V)]; _sync()
    GC.@preserve P_ D_ R_ check(ccall((:ns3d_pt_solve_slab_f64, libns3d), Cint,
                (Ptr{Cvoid}, Ptr{PF}, Ptr{PF}, Ptr{PF}, Ref{PtParams}, Cdouble, Cint, Cint, Cdouble, Cdouble, Ref{Cint},
                 Ptr{Cdouble}, Cint, Ref{Cint}),
                MGPU[], P_, D_, R_, p, εit, niter, nchk, ly^2, psc, it, hist, cap, nchecks))
    return Int(it[]), hist[1:nchecks[]]
end

# ---- ImplicitGlobalGrid's surface (multi.jl:325-373, 399-403, 528-534) ----------------------------------------------
"""
    me, dims = init_global_grid(nx, ny, nz; dimx=0, dimy=0, dimz=0)

multi.jl:325.  One MPI rank per GPU.  The topology is ImplicitGlobalGrid's: `MPI.Dims_create!` over the entries left 0
(`ns3d_dims_create`), ranks in `MPI.Cart_create` order (last dimension fastest); `dimx=1, dimy=1` gives z-slabs.  MPI is
initialised, rank 0 makes an RCCL unique id (`ns3d_mgpu_unique_id`) and broadcasts it, every rank joins the communicator
(`ns3d_mgpu_create_rank_cart`, collective) and takes its kernel context from it.  With one rank every halo call below is
a no-op, as in the reference's own test.
"""
function init_global_grid(nx::Integer, ny::Integer, nz::Integer; dimx::Integer = 0, dimy::Integer = 0, dimz::Integer = 0,
                          quiet::Bool = true)
    MPI.Initialized() || MPI.Init()
    comm = MPI.COMM_WORLD
    me, P = MPI.Comm_rank(comm), MPI.Comm_size(comm)
    device = me % length(AMDGPU.devices())
    AMDGPU.device!(AMDGPU.devices()[device + 1])
    dims = Cint[dimx, dimy, dimz]
    check(ccall((:ns3d_dims_create, libns3d), Cint, (Cint, Ptr{Cint}), P, dims))
    id = Vector{UInt8}(undef, NS3D_UNIQUE_ID_BYTES)
    me == 0 && check(ccall((:ns3d_mgpu_unique_id, libns3d), Cint, (Ptr{UInt8},), id))
    MPI.Bcast!(id, 0, comm)
    MGPU[] = ccall((:ns3d_mgpu_create_rank_cart, libns3d), Ptr{Cvoid},
                   (Ptr{Cint}, Cint, Cint, Ptr{UInt8}, Cint, Cint, Cint, Cint), dims, me, device, id, nx, ny, nz, MODE[])
    MGPU[] == C_NULL && throw(Ns3dError("ns3d_mgpu_create_rank_cart failed: $(lasterror())"))
    OWNS_CTX[] && CTX[] != C_NULL && ccall((:ns3d_destroy, libns3d), Cvoid, (Ptr{Cvoid},), CTX[])
    CTX[] = ccall((:ns3d_mgpu_ctx, libns3d), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), MGPU[], 0)     # owned by the ns3d_mgpu
    OWNS_CTX[] = false
    coords = Cint[0, 0, 0]
    check(ccall((:ns3d_mgpu_coords, libns3d), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}), MGPU[], 0, coords))
    GRID[] = (nx = Int(nx), ny = Int(ny), nz = Int(nz), me = me, dims = (Int(dims[1]), Int(dims[2]), Int(dims[3])),
              coords = (Int(coords[1]), Int(coords[2]), Int(coords[3])))
    return me, Int.(dims)
end
"finalize_global_grid()  multi.jl:534"
function finalize_global_grid(; finalize_MPI::Bool = true)
    if MGPU[] != C_NULL
        _sync()
        ccall((:ns3d_mgpu_destroy, libns3d), Cvoid, (Ptr{Cvoid},), MGPU[])
        MGPU[] = C_NULL; CTX[] = C_NULL
    end
    finalize_MPI && !MPI.Finalized() && MPI.Finalize()
    return nothing
end
# n_g = dims·(n − overlap) + overlap with overlap 2 [upstream ImplicitGlobalGrid]
nx_g() = GRID[].dims[1] * (GRID[].nx - 2) + 2
ny_g() = GRID[].dims[2] * (GRID[].ny - 2) + 2
nz_g() = GRID[].dims[3] * (GRID[].nz - 2) + 2
# x_g(ix,dx,A) = (coord·(n−2) + ix−1)·dx + 0.5·(n − size(A,dim))·dx   [upstream, non-periodic]
_g(i, d, sizeA, n, coord) = (coord * (n - 2) + (i - 1)) * d + 0.5 * (n - sizeA) * d
x_g(ix::Integer, dx, A) = _g(ix, dx, size(A, 1), GRID[].nx, GRID[].coords[1])
y_g(iy::Integer, dy, A) = _g(iy, dy, size(A, 2), GRID[].ny, GRID[].coords[2])
z_g(iz::Integer, dz, A) = _g(iz, dz, size(A, 3), GRID[].nz, GRID[].coords[3])

"update_halo!(A…)  multi.jl:371,373,450,453,455,460,462,182,167,477 → ns3d_update_halo_f64 (x, y, z in turn; z planes as they lie, x / y faces packed by a kernel)"
function update_halo!(A...)
    (MGPU[] == C_NULL || prod(GRID[].dims) == 1) && return nothing
    ptrs = PF[ptr(a) for a in A]
    ext = Cint[]
    for a in A
        append!(ext, _cint3(a))
    end
    _sync()
    GC.@preserve ptrs ext check(ccall((:ns3d_update_halo_f64, libns3d), Cint, (Ptr{Cvoid}, Ptr{PF}, Ptr{Cint}, Cint),
                                      MGPU[], ptrs, ext, length(A)))
    return nothing
end

"""
    gather!(A_inn, A_v)

multi.jl:399-403,528-532: the script passes HOST arrays (`Array(A)[2:end-1,2:end-1,2:end-1]`), so this is ImplicitGlobalGrid's
host gather: one `MPI.Gather!` of the rank blocks, which the root then places by rank coordinates (z-slab blocks of a
column-major array are already in place).  (For device arrays `ns3d_gather_f64` strips the halo and gathers over RCCL;
see mgpu.py for its use.)
"""
function gather!(A_inn::Array, A_v; root::Integer = 0)
    dims = GRID[].dims
    P = prod(dims)
    if P == 1
        A_v .= A_inn
        return nothing
    end
    # ImplicitGlobalGrid requires size(A_v) == dims .* size(A_inn) and errors otherwise; multi.jl's Vz pair (nz-1 planes per
    # rank into dims[3]·(nz-2)+1) violates that for more than one z rank — the reference itself cannot gather Vz there (the
    # unmodified multi.jl therefore cannot return its staggered fields on more than one rank in the staggered dimension).
    # The verdict is the ROOT's (only it holds A_v) and every rank must hear it BEFORE the collective: a root that throws while
    # the others already sit in MPI.Gather! leaves them blocked for good.
    ok = Ref{Cint}(1)
    if GRID[].me == root
        ok[] = size(A_v) == dims .* size(A_inn) ? 1 : 0
    end
    MPI.Bcast!(ok, root, MPI.COMM_WORLD)
    ok[] == 1 || error("gather!: size(A_v) must be dims .* size(A_inn) (rank $(GRID[].me): local block $(size(A_inn)), dims $(dims))")
    if GRID[].me != root
        MPI.Gather!(A_inn, nothing, root, MPI.COMM_WORLD)
        return nothing
    end
    n = length(A_inn)
    if dims[1] == 1 && dims[2] == 1
        MPI.Gather!(A_inn, MPI.UBuffer(vec(A_v), n), root, MPI.COMM_WORLD)
        return nothing
    end
    blocks = Vector{eltype(A_inn)}(undef, n * P)
    MPI.Gather!(A_inn, MPI.UBuffer(blocks, n), root, MPI.COMM_WORLD)
    bx, by, bz = size(A_inn)
    for q in 0:P-1                                    # MPI_Cart_coords: last dimension fastest
        cx, cy, cz = q ÷ (dims[2] * dims[3]), (q ÷ dims[3]) % dims[2], q % dims[3]
        A_v[cx*bx+1:(cx+1)*bx, cy*by+1:(cy+1)*by, cz*bz+1:(cz+1)*bz] .= reshape(view(blocks, q*n+1:(q+1)*n), bx, by, bz)
    end
    return nothing
end

function __init__()
    atexit() do
        OWNS_CTX[] && CTX[] != C_NULL && ccall((:ns3d_destroy, libns3d), Cvoid, (Ptr{Cvoid},), CTX[])
    end
end

end # module
