# NS3DShim.jl — reference-side binding of libns3d.so (include/ns3d.h).
#
# Drop-in for the kernel layer of scripts/NavierStokes3D_gpu.jl:175-368 and
# scripts/NavierStokes3D_multi_gpu.jl:15-281: the same function names and positional argument lists, each a `ccall`
# into the hand-written HIP kernels, plus an `@parallel` macro that simply forwards the call (ParallelStencil's
# launch machinery is replaced by the library's own launch code), so the reference time loops
# gpu.jl:119-142 / multi.jl:446-477 run unmodified on an MI355X:
#
#     # instead of:  using ParallelStencil; @init_parallel_stencil(CUDA, Float64, 3)
#     include("NS3DShim.jl"); using .NS3DShim
#     NS3DShim.init!(device = 0, mode = :strict)      # replaces @init_parallel_stencil
#     Pr = NS3DShim.zeros(nx, ny, nz)                 # replaces @zeros
#
# NOT EXECUTED IN THE BUILD CONTAINER: no Julia toolchain exists there (SURVEY.md §8c).  The entry points it binds
# are exercised through the identical C ABI by the Python/ctypes host layer (navierstokes3d_amd/kernels.py) and its
# GPU parity tests.  Arrays: AMDGPU.jl `ROCArray{Float64,3}` (column-major, packed — exactly the layout ns3d.h
# requires), passed as raw device pointers.
module NS3DShim

using AMDGPU

const libns3d = get(ENV, "NS3D_LIB", joinpath(@__DIR__, "..", "navierstokes3d_amd", "libns3d.so"))
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
const NS3D_STRICT, NS3D_FAST, NS3D_ASYNC = Cint(0), Cint(1), Cint(2)

struct Ns3dError <: Exception
    msg::String
end
lasterror() = unsafe_string(ccall((:ns3d_last_error, libns3d), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing : throw(Ns3dError("libns3d status $rc: $(lasterror())"))

"Replaces `@init_parallel_stencil(CUDA, Float64, 3)` (gpu.jl:4-8)."
function init!(; device::Integer = 0, mode::Symbol = :strict)
    flags = mode == :fast ? NS3D_FAST : NS3D_STRICT      # blocking calls, like @parallel
    CTX[] = ccall((:ns3d_create, libns3d), Ptr{Cvoid}, (Cint, Cint), device, flags)
    CTX[] == C_NULL && throw(Ns3dError("ns3d_create failed: $(lasterror())"))
    # launch on AMDGPU.jl's task-local stream so that broadcasts (`Vx_o .= Vx`) stay ordered with the kernels
    check(ccall((:ns3d_set_stream, libns3d), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), CTX[], AMDGPU.stream().stream))
    return nothing
end
finalize!() = (ccall((:ns3d_destroy, libns3d), Cvoid, (Ptr{Cvoid},), CTX[]); CTX[] = C_NULL; nothing)

"Replaces `@zeros(nx,ny,nz)`."
zeros(dims::Integer...) = AMDGPU.zeros(Float64, dims...)

"`@parallel kernel!(args...)` and `@parallel (ranges...) kernel!(args...)`: the library derives the launch range itself."
macro parallel(args...)
    esc(args[end])
end
macro parallel_indices(args...)   # kernel *definitions* come from this module, not from the script
    nothing
end

const P = Ptr{Float64}
ptr(A) = Base.unsafe_convert(P, A)
const D = Cdouble

# ---- kernels: names and argument order of multi.jl:36-281 / gpu.jl:177-368 --------------------------------------
function update_τ!(τxx, τyy, τzz, τxy, τxz, τyz, Vx, Vy, Vz, μ, dx, dy, dz)
    nx, ny, nz = size(τxx)
    check(ccall((:ns3d_update_tau_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, P, P, P, P, P, P, D, D, D, D, Cint, Cint, Cint),
                CTX[], ptr(τxx), ptr(τyy), ptr(τzz), ptr(τxy), ptr(τxz), ptr(τyz), ptr(Vx), ptr(Vy), ptr(Vz), μ, dx, dy, dz, nx, ny, nz))
end
function predict_V!(Vx, Vy, Vz, τxx, τyy, τzz, τxy, τxz, τyz, ρ, g, dt, dx, dy, dz)
    nx, ny, nz = size(τxx)
    check(ccall((:ns3d_predict_V_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, P, P, P, P, P, P, D, D, D, D, D, D, Cint, Cint, Cint),
                CTX[], ptr(Vx), ptr(Vy), ptr(Vz), ptr(τxx), ptr(τyy), ptr(τzz), ptr(τxy), ptr(τxz), ptr(τyz), ρ, g, dt, dx, dy, dz, nx, ny, nz))
end
function update_∇V!(∇V, Vx, Vy, Vz, dx, dy, dz)
    nx, ny, nz = size(∇V)
    check(ccall((:ns3d_update_divV_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, P, D, D, D, Cint, Cint, Cint),
                CTX[], ptr(∇V), ptr(Vx), ptr(Vy), ptr(Vz), dx, dy, dz, nx, ny, nz))
end
function update_dPrdτ!(Pr, dPrdτ, ∇V, ρ, dt, dτ, damp, dx, dy, dz)
    nx, ny, nz = size(Pr)
    check(ccall((:ns3d_update_dPrdtau_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, D, D, D, D, D, D, D, Cint, Cint, Cint),
                CTX[], ptr(Pr), ptr(dPrdτ), ptr(∇V), ρ, dt, dτ, damp, dx, dy, dz, nx, ny, nz))
end
function update_Pr!(Pr, dPrdτ, dτ)
    nx, ny, nz = size(Pr)
    check(ccall((:ns3d_update_Pr_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, D, Cint, Cint, Cint), CTX[], ptr(Pr), ptr(dPrdτ), dτ, nx, ny, nz))
end
function compute_res!(Rp, Pr, ∇V, ρ, dt, dx, dy, dz)
    nx, ny, nz = size(Pr)
    check(ccall((:ns3d_compute_res_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, D, D, D, D, D, Cint, Cint, Cint),
                CTX[], ptr(Rp), ptr(Pr), ptr(∇V), ρ, dt, dx, dy, dz, nx, ny, nz))
end
function correct_V!(Vx, Vy, Vz, Pr, dt, ρ, dx, dy, dz)
    nx, ny, nz = size(Pr)
    check(ccall((:ns3d_correct_V_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, P, D, D, D, D, D, Cint, Cint, Cint),
                CTX[], ptr(Vx), ptr(Vy), ptr(Vz), ptr(Pr), dt, ρ, dx, dy, dz, nx, ny, nz))
end
for (jl, c) in ((:bc_x!, :ns3d_bc_x_f64), (:bc_y!, :ns3d_bc_y_f64), (:bc_z!, :ns3d_bc_z_f64), (:bc_zV!, :ns3d_bc_zV_f64))
    @eval $jl(A) = check(ccall(($(QuoteNode(c)), libns3d), Cint, (Ptr{Cvoid}, P, Cint, Cint, Cint), CTX[], ptr(A), size(A)...))
end
bc_xhydstatic!(A, dz, nz, g, ρ) = check(ccall((:ns3d_bc_xhydstatic_f64, libns3d), Cint, (Ptr{Cvoid}, P, D, Cint, D, D, Cint, Cint, Cint), CTX[], ptr(A), dz, nz, g, ρ, size(A)...))
bc_x_Vx!(A, V) = check(ccall((:ns3d_bc_x_Vx_f64, libns3d), Cint, (Ptr{Cvoid}, P, D, Cint, Cint, Cint), CTX[], ptr(A), V, size(A)...))
bc_x_Pr!(A, val) = check(ccall((:ns3d_bc_x_Pr_f64, libns3d), Cint, (Ptr{Cvoid}, P, D, Cint, Cint, Cint), CTX[], ptr(A), val, size(A)...))
function advect!(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, dt, dx, dy, dz)
    nx, ny, nz = size(C)
    check(ccall((:ns3d_advect_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, P, P, P, P, P, D, D, D, D, Cint, Cint, Cint, Cint),
                CTX[], ptr(Vx), ptr(Vx_o), ptr(Vy), ptr(Vy_o), ptr(Vz), ptr(Vz_o), ptr(C), ptr(C_o), dt, dx, dy, dz, nx, ny, nz, 1))
end
# multi.jl form (19 arguments) and gpu.jl form (16 arguments) of set_cylinder!
function set_cylinder!(C, Vx, Vy, Vz, a2, b2, ox, oy, sinβ, cosβ, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz)
    nx, ny, nz = size(C)
    check(ccall((:ns3d_set_cylinder_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, P, ntuple(_ -> D, 15)..., Cint, Cint, Cint),
                CTX[], ptr(C), ptr(Vx), ptr(Vy), ptr(Vz), a2, b2, ox, oy, sinβ, cosβ, xco_g, yco_g, zco_g, lx, ly, lz, dx, dy, dz, nx, ny, nz))
end
function set_cylinder!(C, Vx, Vy, Vz, a2, b2, ox, oy, sinβ, cosβ, lx, ly, lz, dx, dy, dz)
    nx, ny, nz = size(C)
    check(ccall((:ns3d_set_cylinder_local_f64, libns3d), Cint, (Ptr{Cvoid}, P, P, P, P, ntuple(_ -> D, 12)..., Cint, Cint, Cint),
                CTX[], ptr(C), ptr(Vx), ptr(Vy), ptr(Vz), a2, b2, ox, oy, sinβ, cosβ, lx, ly, lz, dx, dy, dz, nx, ny, nz))
end
"`maximum(abs.(Rp))` (multi.jl:466, gpu.jl:132) without the temporary."
function maxabs(A)
    out = Ref{Cdouble}(0)
    check(ccall((:ns3d_max_abs_f64, libns3d), Cint, (Ptr{Cvoid}, P, Clong, Ref{Cdouble}), CTX[], ptr(A), length(A), out))
    out[]
end

export @parallel, @parallel_indices, update_τ!, predict_V!, update_∇V!, update_dPrdτ!, update_Pr!, compute_res!,
       correct_V!, bc_x!, bc_y!, bc_z!, bc_zV!, bc_xhydstatic!, bc_x_Vx!, bc_x_Pr!, advect!, set_cylinder!, maxabs
end # module
