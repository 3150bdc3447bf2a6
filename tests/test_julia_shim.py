"""julia/NS3DShim.jl cannot be executed (no Julia toolchain here or on the GPU box).  What CAN be checked mechanically is
checked here: every `ccall((:ns3d_…, libns3d), Ret, (ArgTypes…), args…)` in the shim against the prototype the C preprocessor
expands out of include/ns3d.h — symbol exists, return class, number and class (pointer / double / int / 64-bit int) of every
argument, as many actual arguments as declared types — and the field-for-field layout of `struct PtParams` against
`ns3d_pt_params` (and against the ctypes mirror the GPU tests drive the same entry points through)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "julia", "NS3DShim.jl")
HEADER = os.path.join(ROOT, "include", "ns3d.h")


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "([{"
        depth -= ch in ")]}"
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def closing(s, i):
    depth = 0
    for j in range(i, len(s)):
        depth += s[j] == "("
        if s[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced")


def shim_source():
    return re.sub(r"#[^\n]*", "", open(SHIM, encoding="utf-8").read())


def julia_ccalls():
    src, calls = shim_source(), []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libns3d\)\s*,", src):
        start = src.index("(", m.start())
        parts = split_top(src[start + 1:closing(src, start)])
        ret, types, args = parts[1], parts[2], parts[3:]
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        calls.append((m.group(1), ret, split_top(types[1:-1]), args))
    return calls


def jclass(t):
    t = t.strip()
    if t == "Cint":
        return "int"
    if t == "Cdouble":
        return "double"
    if t in ("Clong", "Clonglong", "Culonglong"):
        return "int64"
    if t == "Cvoid":
        return "void"
    if t in ("PF", "Cstring") or t.startswith("Ptr{") or t.startswith("Ref{"):
        return "ptr"
    raise ValueError("unmapped Julia C type " + t)


def cclass(p):
    p = p.strip()
    if "*" in p:
        return "ptr"
    if p.startswith("double"):
        return "double"
    if p.startswith("int"):
        return "int"
    if p.startswith("long") or p.startswith("unsigned"):
        return "int64"
    raise ValueError("unmapped C parameter " + p)


def header_prototypes():
    txt = subprocess.run(["gcc", "-E", "-P", HEADER], capture_output=True, text=True, check=True).stdout
    txt = re.sub(r"\s+", " ", txt)
    protos = {}
    for m in re.finditer(r"([\w ]+?[\s\*]+)(ns3d_\w+)\s*\(([^()]*)\)\s*;", txt):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        pl = [] if params in ("", "void") else [cclass(p) for p in split_top(params)]
        protos[name] = ("ptr" if "*" in ret else "void" if ret.endswith("void") else "int", pl)
    return protos, txt


def test_every_ccall_matches_its_prototype():
    protos, _ = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 30 and len(protos) >= 100
    for name, ret, types, args in calls:
        assert name in protos, "the shim binds %s, which include/ns3d.h does not declare" % name
        cret, cparams = protos[name]
        assert jclass(ret) == cret, (name, ret, cret)
        assert [jclass(t) for t in types] == cparams, (name, types, cparams)
        assert len(args) == len(types), (name, len(args), len(types))
    bound = {c[0] for c in calls}
    # the reference-signature kernels and the grid layer are all bound
    for need in ("update_tau", "predict_V", "set_cylinder", "update_divV", "update_dPrdtau", "update_Pr", "compute_res", "correct_V",
                 "advect", "bc_x", "bc_y", "bc_z", "pt_solve", "pt_solve_slab", "update_halo", "max_abs"):
        assert "ns3d_%s_f64" % need in bound, need
    for need in ("ns3d_create", "ns3d_last_error", "ns3d_dims_create", "ns3d_mgpu_unique_id", "ns3d_mgpu_create_rank_cart",
                 "ns3d_mgpu_ctx", "ns3d_mgpu_coords", "ns3d_mgpu_destroy"):
        assert need in bound, need


def test_ptparams_struct_layout():
    _, txt = header_prototypes()
    body = re.search(r"typedef struct ns3d_pt_params \{(.*?)\} ns3d_pt_params;", txt).group(1)
    cfields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(" ", 1)
        cfields += [(n.strip(), ctype) for n in names.split(",")]
    jbody = re.search(r"struct PtParams(.*?)\nend", shim_source(), re.S).group(1)
    jfields = [(n, {"Cdouble": "double", "Cint": "int"}[t]) for n, t in re.findall(r"(\w+)::(\w+)", jbody)]
    assert jfields == cfields
    from navierstokes3d_amd import lib as L
    import ctypes as C
    pyfields = [(n, {C.c_double: "double", C.c_int: "int"}[t]) for n, t in L.PtParams._fields_]
    assert pyfields == cfields


def test_step_struct_layouts():
    """ns3d_step_fields / ns3d_step_params (ns3d_time_step, round 4): the header, the Julia mirrors and the ctypes mirrors agree field for
    field (name, order, class)."""
    import ctypes as C
    from navierstokes3d_amd import lib as L
    _, txt = header_prototypes()
    for cname, jname, py in (("ns3d_step_fields", "StepFields", L.StepFields), ("ns3d_step_params", "StepParams", L.StepParams)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), txt).group(1)
        cfields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.split(" ", 1)
            for n in names.split(","):
                n = n.strip()
                cfields.append((n.lstrip("*"), "ptr" if n.startswith("*") else {"double": "double", "int": "int"}[ctype]))
        jbody = re.search(r"struct %s(.*?)\nend" % jname, shim_source(), re.S).group(1)
        jfields = [(n, {"Cdouble": "double", "Cint": "int", "PF": "ptr"}[t]) for n, t in re.findall(r"(\w+)::(\w+)", jbody)]
        assert jfields == cfields, cname
        pyfields = [(n, {C.c_double: "double", C.c_int: "int", C.c_void_p: "ptr"}[t]) for n, t in py._fields_]
        assert pyfields == cfields, cname


def test_definitions_are_swallowed_and_calls_forwarded():
    """The two macros the scripts' kernels go through: `@parallel function …` / `@parallel_indices (…) function …` must expand
    to nothing (their bodies use ParallelStencil macros that do not exist here), calls must be forwarded."""
    src = shim_source()
    assert re.search(r"_is_definition\(ex\)\s*=", src)
    i = src.index("macro parallel(")
    body = src[i:src.index("\nend", i)]
    assert "_is_definition" in body and "nothing" in body
    i = src.index("macro parallel_indices(")                      # only ever applied to definitions (multi.jl:108-281)
    assert "nothing" in src[i:src.index("\nend", i)]
    for name in ("init_parallel_stencil", "zeros"):
        assert "macro %s(" % name in src
    for fn in ("init_global_grid", "finalize_global_grid", "nx_g", "ny_g", "nz_g", "x_g", "y_g", "z_g", "update_halo!", "gather!"):
        assert re.search(r"(function\s+%s\(|\n%s\()" % (re.escape(fn), re.escape(fn)), src), fn


def shim_methods():
    """name → list of (n_positional_min, varargs) of the shim's function definitions (long and short form)"""
    src = shim_source()
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*(?:function\s+)?([\w!∇τ]+)\(", src):
        name = m.group(1)
        start = m.end() - 1
        end = closing(src, start)
        after = src[end + 1:end + 40].lstrip()
        is_long = src[m.start():m.end()].lstrip().startswith("function")
        if not is_long and not after.startswith("="):
            continue                                               # a call, not a definition
        params = src[start + 1:end].split(";")[0]
        pl = [p for p in split_top(params) if p.strip()]
        varargs = any(p.strip().endswith("...") for p in pl)
        out.setdefault(name, []).append((len(pl) - (1 if varargs else 0), varargs))
    return out


def test_the_shim_serves_every_call_site_of_the_reference_scripts():
    """tests/golden/reference_call_sites.json (oracle/extract_call_sites.py: every `@parallel kernel!(…)` call and every
    ImplicitGlobalGrid call of multi.jl and gpu.jl, as data) against the shim's method table: the callee exists and takes that
    many positional arguments.  (`@parallel (ranges) f!(…)` forwards f!(…): the ranges are dropped by the macro.)"""
    import json
    sites = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_call_sites.json"), encoding="utf-8"))
    meths = shim_methods()
    assert len(sites["parallel_calls"]) >= 40 and len(sites["grid_calls"]) >= 50
    for c in sites["parallel_calls"] + sites["grid_calls"]:
        name, n = c["callee"], c["nargs"]
        assert name in meths, "%s (%s:%d) has no method in the shim" % (name, c["script"], c["line"])
        assert any((n == k and not va) or (va and n >= k) for k, va in meths[name]), (name, n, meths[name], c["script"], c["line"])
    src = shim_source()
    for mac in sites["macros"]:
        if mac.startswith("@"):
            assert "macro %s(" % mac[1:] in src, mac
        else:
            assert "module Data" in src and mac.split(".")[1] in src, mac


def test_the_python_mirror_serves_the_same_call_sites():
    """navierstokes3d_amd/kernels.py mirrors the reference's kernel names (ASCII: τ→tau, ∇V→divV, no `!`) with the same
    positional arguments: every `@parallel kernel!(…)` call site of the two scripts binds to its mirror's signature."""
    import inspect
    import json
    from navierstokes3d_amd import kernels as K
    sites = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_call_sites.json"), encoding="utf-8"))
    for c in sites["parallel_calls"]:
        name = c["callee"].rstrip("!").replace("τ", "tau").replace("∇V", "divV")
        fn = getattr(K, name, None)
        assert fn is not None, (c["callee"], name)
        inspect.signature(fn).bind(*([None] * c["nargs"]))           # raises TypeError when the arity does not fit
