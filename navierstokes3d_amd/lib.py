"""ctypes binding of libns3d.so (include/ns3d.h).  There is NO CPU fallback: if the HIP library is missing
or no GPU is present, every entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NS3D_LIB") or os.path.join(_HERE, "libns3d.so")   # NS3D_LIB: A/B-test another build

NS3D_OK = 0
NS3D_ERR_ARG, NS3D_ERR_HIP, NS3D_ERR_STATE, NS3D_ERR_RCCL = 1, 2, 3, 4
NS3D_UNIQUE_ID_BYTES = 128
NS3D_STRICT, NS3D_FAST, NS3D_ASYNC, NS3D_IEEE_DIV = 0x0, 0x1, 0x2, 0x4
NS3D_BC_MULTI, NS3D_BC_GPU = 0, 1


class Ns3dError(RuntimeError):
    pass


class PtParams(C.Structure):
    """struct ns3d_pt_params (include/ns3d.h)."""
    _fields_ = [
        ("rho", C.c_double), ("dt", C.c_double), ("dtau", C.c_double), ("damp", C.c_double),
        ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("bc_kind", C.c_int), ("owns_outlet", C.c_int),
        ("outlet_val", C.c_double), ("g", C.c_double),
        ("z_lo_is_halo", C.c_int), ("z_hi_is_halo", C.c_int),
    ]


class StepFields(C.Structure):
    """struct ns3d_step_fields (include/ns3d.h): device pointers, in/out (the fused step swaps X and X_o)."""
    _fields_ = [(n, C.c_void_p) for n in ("Pr", "dPrdtau", "divV", "Vx", "Vy", "Vz", "Vx_o", "Vy_o", "Vz_o", "C", "C_o",
                                          "txx", "tyy", "tzz", "txy", "txz", "tyz")]


class StepParams(C.Structure):
    """struct ns3d_step_params (include/ns3d.h)."""
    _fields_ = [("script", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int)] + \
               [(n, C.c_double) for n in ("mu", "rho", "g", "dt", "dtau", "damp", "dx", "dy", "dz", "eps")] + \
               [("niter", C.c_int), ("nchk", C.c_int), ("err_mul", C.c_double), ("err_div", C.c_double)] + \
               [(n, C.c_double) for n in ("a2", "b2", "ox", "oy", "sinb", "cosb", "xco_g", "yco_g", "zco_g", "lx", "ly", "lz")] + \
               [("owns_inlet", C.c_int), ("owns_outlet", C.c_int), ("vin", C.c_double), ("faithful", C.c_int), ("pressure", C.c_int),
                ("write_stress", C.c_int)]


_P, _D, _I, _L = C.c_void_p, C.c_double, C.c_int, C.c_long

# name → argument ctypes after the leading ctx pointer (same for _f64 and _f32)
SIGNATURES = {
    "update_tau": [_P] * 9 + [_D] * 4 + [_I] * 3,
    "predict_V": [_P] * 9 + [_D] * 6 + [_I] * 3,
    "predict_fused": [_P] * 6 + [_D] * 7 + [_I] * 3,
    "set_cylinder": [_P] * 4 + [_D] * 15 + [_I] * 3,
    "set_cylinder_local": [_P] * 4 + [_D] * 12 + [_I] * 3,
    "update_divV": [_P] * 4 + [_D] * 3 + [_I] * 3,
    "update_dPrdtau": [_P] * 3 + [_D] * 7 + [_I] * 3,
    "update_Pr": [_P] * 2 + [_D] + [_I] * 3,
    "compute_res": [_P] * 3 + [_D] * 5 + [_I] * 3,
    "max_abs": [_P, _L, C.POINTER(_D)],
    "correct_V": [_P] * 4 + [_D] * 5 + [_I] * 3,
    "bc_x": [_P] + [_I] * 3,
    "bc_y": [_P] + [_I] * 3,
    "bc_z": [_P] + [_I] * 3,
    "bc_zV": [_P] + [_I] * 3,
    "bc_xhydstatic": [_P, _D, _I, _D, _D] + [_I] * 3,
    "bc_x_Vx": [_P, _D] + [_I] * 3,
    "bc_x_Pr": [_P, _D] + [_I] * 3,
    "copy": [_P, _P, _L],
    "advect": [_P] * 8 + [_D] * 4 + [_I] * 4,
    "copy_advect": [_P] * 8 + [_D] * 4 + [_I] * 4,
    "set_bc_Pr": [_P, _I, _I, _D, _D, _I, _D, _D] + [_I] * 3,
    "set_bc_Vel": [_P] * 3 + [_I, _I, _D] + [_I] * 3,
    "pt_iterate": [_P] * 3 + [C.POINTER(PtParams), _I],
    "poisson_direct": [_P] * 3 + [C.POINTER(PtParams)],
    "pt_sweep": [_P] * 4 + [C.POINTER(PtParams), _I, _I],
    "pt_sweep2": [_P] * 5 + [C.POINTER(PtParams), _I, _I],
    "plan_pt": [_P] * 5 + [C.POINTER(PtParams), _I, _I],
    "pt_sweepn": [_I] + [_P] * 5 + [C.POINTER(PtParams), _I, _I],
    "residual_max": [_P] * 2 + [C.POINTER(PtParams), C.POINTER(_D)],
    "selftest_exact_div": [_D, _L, C.c_ulonglong, C.POINTER(_L)],
    "pt_solve": [_P] * 3 + [C.POINTER(PtParams), _D, _I, _I, _D, _D, C.POINTER(_I), C.POINTER(_D), _I, C.POINTER(_I)],
    "time_step": [C.POINTER(StepFields), C.POINTER(StepParams), C.POINTER(_I), C.POINTER(_D), _I, C.POINTER(_I)],
}
CONTEXT_SYMBOLS = ["ns3d_version", "ns3d_last_error", "ns3d_create", "ns3d_destroy", "ns3d_flags",
                   "ns3d_set_stream", "ns3d_use_own_stream", "ns3d_get_stream", "ns3d_sync", "ns3d_reserve_cus", "ns3d_reserved_cus", "ns3d_set_pt_variant",
                   "ns3d_set_pt2_variant", "ns3d_set_ptn_variant", "ns3d_set_pt_depth", "ns3d_set_graph_mode", "ns3d_set_autotune", "ns3d_last_pt2_variant", "ns3d_last_ptn_variant", "ns3d_last_pt_depth",
                   "ns3d_arith_build", "ns3d_cached_graphs", "ns3d_set_persist_mode", "ns3d_persist_faults"]


_PP = C.POINTER(C.c_void_p)      # T *const *  — one device pointer per local rank (field-major for field lists)
# multi-GPU layer (first argument ns3d_mgpu*): name → (restype, argtypes)
MGPU_SYMBOLS = {
    "ns3d_mgpu_create": (_P, [_I, C.POINTER(_I), _I, _I, _I, _I]),
    "ns3d_mgpu_create_cart": (_P, [C.POINTER(_I), C.POINTER(_I), _I, _I, _I, _I]),
    "ns3d_dims_create": (_I, [_I, C.POINTER(_I)]),
    "ns3d_mgpu_reserve_cus": (_I, [_P, _I]),
    "ns3d_mgpu_set_interior_chunks": (_I, [_P, _I]),
    "ns3d_mgpu_unique_id": (_I, [C.c_char_p]),
    "ns3d_mgpu_create_rank": (_P, [_I, _I, _I, C.c_char_p, _I, _I, _I, _I]),
    "ns3d_mgpu_create_rank_cart": (_P, [C.POINTER(_I), _I, _I, C.c_char_p, _I, _I, _I, _I]),
    "ns3d_mgpu_destroy": (None, [_P]),
    "ns3d_mgpu_world": (_I, [_P]),
    "ns3d_mgpu_nlocal": (_I, [_P]),
    "ns3d_mgpu_rank": (_I, [_P, _I]),
    "ns3d_mgpu_ctx": (_P, [_P, _I]),
    "ns3d_mgpu_nz_g": (_I, [_P]),
    "ns3d_mgpu_dims": (_I, [_P, C.POINTER(_I)]),
    "ns3d_mgpu_coords": (_I, [_P, _I, C.POINTER(_I)]),
    "ns3d_mgpu_n_g": (_I, [_P, C.POINTER(_I)]),
    "ns3d_mgpu_transport": (C.c_char_p, [_P]),
    "ns3d_mgpu_rccl_ranks": (_I, [_P]),
    "ns3d_mgpu_pass_depth": (_I, [_P]),
    "ns3d_mgpu_ghost_depth": (_I, [_P]),
    "ns3d_mgpu_sync": (_I, [_P]),
    "ns3d_max_g": (_I, [_P, C.POINTER(_D), C.POINTER(_D)]),
    "ns3d_mgpu_set_temporal": (_I, [_P, _I]),
    "ns3d_slab_iterate": (_I, [_P, _I]),
    "ns3d_slab_plan": (_I, [_P]),
    "ns3d_slab_residual": (_I, [_P, C.POINTER(_D)]),
}
MGPU_SIGNATURES = {   # typed (_f64/_f32), after the leading ns3d_mgpu*
    "update_halo": [_PP, C.POINTER(_I), _I],
    "gather": [_PP, _I, _I, _I, _P],
    "slab_load": [_PP, _PP, _PP, C.POINTER(PtParams)],
    "slab_store": [_PP, _PP],
    "advect_wide": [_PP] * 8 + [_D] * 4 + [_I],
    "pt_solve_slab": [_PP, _PP, _PP, C.POINTER(PtParams), _D, _I, _I, _D, _D, C.POINTER(_I), C.POINTER(_D), _I,
                      C.POINTER(_I)],
}


def exported_symbols():
    """Every symbol include/ns3d.h declares."""
    out = list(CONTEXT_SYMBOLS) + list(MGPU_SYMBOLS)
    for n in list(SIGNATURES) + list(MGPU_SIGNATURES):
        out += ["ns3d_%s_f64" % n, "ns3d_%s_f32" % n]
    return out


_LIB = None


def load():
    """Load libns3d.so (after torch, so that it binds to the HIP runtime already in the process)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise Ns3dError("libns3d.so is not built (run `python -m navierstokes3d_amd.build`); "
                        "navierstokes3d_amd has no CPU or PyTorch fallback path")
    import torch  # noqa: F401  (loads libamdhip64.so.7 first)
    lib = C.CDLL(LIB_PATH)
    lib.ns3d_version.restype = _I
    lib.ns3d_last_error.restype = C.c_char_p
    lib.ns3d_create.restype = _P
    lib.ns3d_create.argtypes = [_I, _I]
    lib.ns3d_destroy.restype = None
    lib.ns3d_destroy.argtypes = [_P]
    lib.ns3d_flags.argtypes = [_P]
    lib.ns3d_set_stream.argtypes = [_P, _P]
    lib.ns3d_use_own_stream.argtypes = [_P]
    lib.ns3d_get_stream.restype = _P
    lib.ns3d_get_stream.argtypes = [_P]
    lib.ns3d_sync.argtypes = [_P]
    lib.ns3d_reserve_cus.argtypes = [_P, _I]
    lib.ns3d_reserved_cus.argtypes = [_P]
    lib.ns3d_set_pt_variant.argtypes = [_P, _I]
    lib.ns3d_set_pt2_variant.argtypes = [_P, _I]
    lib.ns3d_set_ptn_variant.argtypes = [_P, _I]
    lib.ns3d_set_pt_depth.argtypes = [_P, _I]
    lib.ns3d_set_graph_mode.argtypes = [_P, _I]
    lib.ns3d_set_autotune.argtypes = [_P, _I]
    lib.ns3d_last_pt2_variant.argtypes = [_P]
    lib.ns3d_last_pt2_variant.restype = _I
    lib.ns3d_last_ptn_variant.argtypes = [_P]
    lib.ns3d_last_ptn_variant.restype = _I
    lib.ns3d_last_pt_depth.argtypes = [_P]
    lib.ns3d_last_pt_depth.restype = _I
    lib.ns3d_set_persist_mode.argtypes = [_P, _I]
    lib.ns3d_set_persist_mode.restype = _I
    lib.ns3d_persist_faults.argtypes = [_P]
    lib.ns3d_persist_faults.restype = _I
    lib.ns3d_cached_graphs.argtypes = [_P]
    lib.ns3d_cached_graphs.restype = _I
    lib.ns3d_arith_build.argtypes = [_P, _D, _D, _D]
    lib.ns3d_arith_build.restype = _I
    for name, args in SIGNATURES.items():
        for suf in ("f64", "f32"):
            fn = getattr(lib, "ns3d_%s_%s" % (name, suf))
            fn.restype = _I
            fn.argtypes = [_P] + args
    for name, (res, args) in MGPU_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, args in MGPU_SIGNATURES.items():
        for suf in ("f64", "f32"):
            fn = getattr(lib, "ns3d_%s_%s" % (name, suf))
            fn.restype = _I
            fn.argtypes = [_P] + args
    _LIB = lib
    return lib


def last_error():
    return load().ns3d_last_error().decode(errors="replace")


def check(rc):
    if rc != NS3D_OK:
        raise Ns3dError("libns3d status %d: %s" % (rc, last_error()))
