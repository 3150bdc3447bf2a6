"""Mechanical evaluation of the reference's kernel SOURCE TEXT with NumPy — test infrastructure, generation time only.

The hand-written oracle (ns3d_oracle.c, numpy_ref.py) restates the reference's kernels; this module removes the hand from
the loop for the kernels that are pure stencil statements: it reads the `@parallel function …` and `@parallel_indices …
function …` definitions straight out of scripts/NavierStokes3D_multi_gpu.jl and scripts/NavierStokes3D_gpu.jl (in the
container where /root/reference exists), rewrites each statement token by token into a NumPy expression and executes it.
What it assumes is ParallelStencil's published macro table and launch rule [upstream, SURVEY.md App. A]:

    @all(A)   = A[ix,iy,iz]                       @inn(A)  = A[ix+1,iy+1,iz+1]
    @d_xa(A)  = A[ix+1,iy,iz]-A[ix,iy,iz]         @d_xi(A) = A[ix+1,iy+1,iz+1]-A[ix,iy+1,iz+1]          (likewise y, z)
    @d2_xi(A) = (A[ix+2,iy+1,iz+1]-A[ix+1,iy+1,iz+1])-(A[ix+1,iy+1,iz+1]-A[ix,iy+1,iz+1])              (likewise y, z)
    `@all(X) = rhs` runs for ix ≤ size(X,1) …; `@inn(X) = rhs` for ix ≤ size(X,1)−2 …   (@within)
    `@parallel (r1,r2) f!(…)` of an `@parallel_indices (i,j)` kernel runs the body for (i,j) ∈ r1×r2

plus Julia's own rules for what is left: `2μ` is `(2*μ)`, `+ - * /` associate to the left with the usual precedence (as in
Python), IEEE double arithmetic without contraction (as in NumPy), 1-based indices with `end`.  The macro `@∇V()` is expanded
from its definition in the script.  Nothing of the reference's text is stored: generate_goldens() writes INPUT SEEDS and OUTPUT
ARRAYS only (tests/golden/jl_eval_kernels.npz); tests/test_oracle.py compares the C oracle with them bit for bit, and
re-runs the evaluation itself where the reference is present.
"""
import os
import re

import numpy as np

REF = "/root/reference"
SCRIPTS = {"multi": "scripts/NavierStokes3D_multi_gpu.jl", "gpu": "scripts/NavierStokes3D_gpu.jl"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def available(ref=REF):
    return all(os.path.exists(os.path.join(ref, p)) for p in SCRIPTS.values())


def _py_names(s):
    return s.replace("∇V", "divV")


def parse_kernels(script, ref=REF):
    """name → dict(kind='stencil'|'indices', args=[…], ivars=[…], body=[statement lines], divmacro=str)"""
    lines = open(os.path.join(ref, SCRIPTS[script]), encoding="utf-8").read().split("\n")
    divmacro = None
    for ln in lines:
        m = re.match(r"\s*macro ∇V\(\)\s*esc\(:\((.*)\)\)\s*end", ln)
        if m:
            divmacro = m.group(1).strip()
    out, i = {}, 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*@parallel function ([\w!∇τ]+)\((.*)\)\s*$", ln)
        mi = re.match(r"\s*@parallel_indices \((.*?)\) function ([\w!∇τ]+)\((.*)\)\s*$", ln)
        if m or mi:
            name = m.group(1) if m else mi.group(2)
            args = [a.strip() for a in (m.group(2) if m else mi.group(3)).split(",")]
            body, i = [], i + 1
            while not re.match(r"^\s*end\s*$", lines[i]) or _depth(body) > 0:
                code = lines[i].split("#")[0].rstrip()
                if code.strip() and not re.match(r"^\s*return\b", code):
                    body.append(code.strip())
                i += 1
            out[name] = dict(kind="stencil" if m else "indices", args=args, ivars=[v.strip() for v in mi.group(1).split(",")] if mi else [],
                             body=body, divmacro=divmacro)
        i += 1
    return out


def _depth(body):
    d = 0
    for b in body:
        if re.match(r"^(if|for|while)\b", b):
            d += 1
        if re.match(r"^end\b", b):
            d -= 1
    return d


# ---- stencil kernels ---------------------------------------------------------------------------------------------
_OFFS = {  # macro → list of (sign, (ox,oy,oz)) terms, evaluated left to right; d2 handled separately
    "all": [(+1, (0, 0, 0))], "inn": [(+1, (1, 1, 1))],
    "d_xa": [(+1, (1, 0, 0)), (-1, (0, 0, 0))], "d_ya": [(+1, (0, 1, 0)), (-1, (0, 0, 0))], "d_za": [(+1, (0, 0, 1)), (-1, (0, 0, 0))],
    "d_xi": [(+1, (1, 1, 1)), (-1, (0, 1, 1))], "d_yi": [(+1, (1, 1, 1)), (-1, (1, 0, 1))], "d_zi": [(+1, (1, 1, 1)), (-1, (1, 1, 0))],
}
_D2 = {"d2_xi": ((2, 1, 1), (1, 1, 1), (0, 1, 1)), "d2_yi": ((1, 2, 1), (1, 1, 1), (1, 0, 1)), "d2_zi": ((1, 1, 2), (1, 1, 1), (1, 1, 0))}


def _macro(n):
    def sl(A, o):
        idx = tuple(slice(o[d], o[d] + n[d]) for d in range(3))
        v = A[idx]
        if v.shape != tuple(n):
            raise IndexError("BoundsError: offsets %r of an array %r over the range %r" % (o, A.shape, n))
        return v

    def m(name, A):
        if name in _D2:
            a, b, c = _D2[name]
            return (sl(A, a) - sl(A, b)) - (sl(A, b) - sl(A, c))
        terms = _OFFS[name]
        v = sl(A, terms[0][1])
        for s, o in terms[1:]:
            v = v - sl(A, o) if s < 0 else v + sl(A, o)
        return v
    return m


def _rewrite_expr(rhs, divmacro):
    rhs = rhs.replace("@∇V()", "(" + divmacro + ")")
    rhs = _py_names(rhs)
    rhs = re.sub(r"(?<![\w.])(\d+(?:\.\d+)?)([^\W\d_eE]\w*)", r"(\1*\2)", rhs)        # 2μ → (2*μ)
    rhs = re.sub(r"@(\w+)\(([\w]+)\)", r'_m("\1", \2)', rhs)
    return rhs


def run_stencil(k, values):
    """values: dict arg name → array / scalar (arrays are modified in place, like the kernel)"""
    env = {_py_names(a): values[a] for a in k["args"]}
    for st in k["body"]:
        m = re.match(r"@(all|inn)\(([\w∇τ]+)\)\s*=\s*(.*)$", st)
        if not m:
            raise ValueError("statement not understood: " + st)
        X = env[_py_names(m.group(2))]
        n = X.shape if m.group(1) == "all" else tuple(s - 2 for s in X.shape)
        if min(n) <= 0:
            continue
        env["_m"] = _macro(n)
        val = eval(_rewrite_expr(m.group(3), k["divmacro"]), {"__builtins__": {}}, env)
        o = 0 if m.group(1) == "all" else 1
        X[o:o + n[0], o:o + n[1], o:o + n[2]] = val


# ---- @parallel_indices kernels with explicit index statements ---------------------------------------------------------
def _rewrite_indexed(code, arrays):
    def fix(m):
        name, inside = m.group(1), m.group(2)
        if name not in arrays:
            return m.group(0)
        parts = [p.strip() for p in inside.split(",")]
        return name + "[" + ", ".join("(%s)-1" % re.sub(r"\bend\b", "%s.shape[%d]" % (name, d), p) for d, p in enumerate(parts)) + "]"
    return re.sub(r"([\w∇τ]+)\[([^\[\]]*)\]", fix, _py_names(code))


def run_indices(k, values, ranges):
    """ranges: one (lo, hi) 1-based inclusive pair per index variable — the `(1:size(A,2),1:size(A,3))` of the call site"""
    env = {_py_names(a): values[a] for a in k["args"]}
    arrays = {n for n, v in env.items() if isinstance(v, np.ndarray)}
    stmts = []
    for st in k["body"]:
        if re.match(r"^(if|for|while|end)\b", st):
            raise ValueError("control flow is not evaluated here: " + st)
        lhs, rhs = st.split("=", 1)
        stmts.append(compile(_rewrite_indexed(lhs.strip(), arrays) + " = " + _rewrite_indexed(rhs.strip(), arrays), "<jl>", "exec"))
    import itertools
    for tup in itertools.product(*[range(lo, hi + 1) for lo, hi in ranges]):
        for v, val in zip(k["ivars"], tup):
            env[v] = val
        for c in stmts:
            exec(c, {"__builtins__": {}}, env)


# ---- golden vectors ---------------------------------------------------------------------------------------------
def rnd(seed, shape):
    return np.asfortranarray(np.random.default_rng(seed).uniform(-1.0, 1.0, size=shape))


def cases(script, grid):
    """(kernel name, ordered argument dict factory) for every kernel evaluated; seeds fixed → inputs reproducible anywhere"""
    nx, ny, nz = grid
    c, s, i3 = (nx, ny, nz), (nx - 1, ny - 1, nz - 1), (nx - 2, ny - 2, nz - 2)
    vx, vy, vz = (nx + 1, ny, nz), (nx, ny + 1, nz), (nx, ny, nz + 1)
    sc = dict(μ=1e-3, ρ=1000.0, g=9.81, dt=0.013, dτ=0.009, damp=2.0 / nx, dx=1.0 / nx, dy=0.6 / ny, dz=0.7 / nz)
    A = lambda seed, shp: rnd(1000 * seed + nx + 7 * ny + 13 * nz, shp)
    out = [
        ("update_τ!", dict(τxx=A(1, c), τyy=A(2, c), τzz=A(3, c), τxy=A(4, s), τxz=A(5, s), τyz=A(6, s), Vx=A(7, vx), Vy=A(8, vy), Vz=A(9, vz),
                           μ=sc["μ"], dx=sc["dx"], dy=sc["dy"], dz=sc["dz"])),
        ("predict_V!", dict(Vx=A(7, vx), Vy=A(8, vy), Vz=A(9, vz), τxx=A(1, c), τyy=A(2, c), τzz=A(3, c), τxy=A(4, s), τxz=A(5, s), τyz=A(6, s),
                            ρ=sc["ρ"], g=sc["g"], dt=sc["dt"], dx=sc["dx"], dy=sc["dy"], dz=sc["dz"])),
        ("update_∇V!", {"∇V": A(10, c), "Vx": A(7, vx), "Vy": A(8, vy), "Vz": A(9, vz), "dx": sc["dx"], "dy": sc["dy"], "dz": sc["dz"]}),
        ("update_dPrdτ!", {"Pr": A(11, c), "dPrdτ": A(12, i3), "∇V": A(10, c), "ρ": sc["ρ"], "dt": sc["dt"], "dτ": sc["dτ"], "damp": sc["damp"],
                           "dx": sc["dx"], "dy": sc["dy"], "dz": sc["dz"]}),
        ("update_Pr!", {"Pr": A(11, c), "dPrdτ": A(12, i3), "dτ": sc["dτ"]}),
        ("compute_res!", {"Rp": A(13, i3), "Pr": A(11, c), "∇V": A(10, c), "ρ": sc["ρ"], "dt": sc["dt"], "dx": sc["dx"], "dy": sc["dy"], "dz": sc["dz"]}),
        ("correct_V!", dict(Vx=A(7, vx), Vy=A(8, vy), Vz=A(9, vz), Pr=A(11, c), dt=sc["dt"], ρ=sc["ρ"], dx=sc["dx"], dy=sc["dy"], dz=sc["dz"])),
    ]
    for q, shp in enumerate((c, vx, vy, vz)):
        out += [("bc_x!", dict(A=A(20 + q, shp))), ("bc_y!", dict(A=A(24 + q, shp))), ("bc_z!", dict(A=A(28 + q, shp)))]
    if script == "multi":
        out += [("bc_x_Vx!", dict(A=A(32, vx), V=1.25)), ("bc_x_Pr!", dict(A=A(33, c), val=0.375))]
    else:
        out += [("bc_zV!", dict(A=A(34, vx))), ("bc_zV!", dict(A=A(35, vy))),
                ("bc_xhydstatic!", dict(A=A(40, c), dz=sc["dz"], nz=nz, g=sc["g"], ρ=sc["ρ"]))]
    return out


def launch_ranges(name, A):
    """the ranges of the scripts' own call sites: every index of the two dimensions the kernel does not pin"""
    fixed = {"x": 0, "y": 1, "z": 2}[name.split("_")[1][0]]
    return [(1, A.shape[d]) for d in range(3) if d != fixed]


GRIDS = [(9, 7, 6), (13, 6, 8)]


def evaluate_all(ref=REF):
    """{key: output array} for every case, key = script/grid/case index/kernel/argument"""
    res = {}
    for script in SCRIPTS:
        ks = parse_kernels(script, ref)
        for grid in GRIDS:
            for q, (name, vals) in enumerate(cases(script, grid)):
                k = ks[name]
                if k["kind"] == "stencil":
                    run_stencil(k, vals)
                else:
                    run_indices(k, vals, launch_ranges(name, vals["A"]))
                for a, v in vals.items():
                    if isinstance(v, np.ndarray) and v.ndim == 3:
                        res["%s/%dx%dx%d/%02d/%s/%s" % (script, grid[0], grid[1], grid[2], q, name, a)] = v
    return res


def generate_goldens():
    res = evaluate_all()
    res.update(evaluate_all2())
    path = os.path.join(ROOT, "tests", "golden", "jl_eval_kernels.npz")
    np.savez_compressed(path, **{k: v for k, v in res.items()})
    print(len(res), "arrays →", path)



# =====================================================================================================================
# Part 2 — the kernels and host functions that are plain Julia: set_cylinder!, advect!/backtrack!/lerp, set_bc_Vel!,
# set_bc_Pr!.  A line-by-line transpiler for the subset of Julia they use:
#   tuple assignment, `if … end`, `&&`, comparisons, 1-based indexing with size()/checkbounds(), clamp, floor(Int,·), `x % 1`
#   (Julia's rem: sign of the dividend), Bool in arithmetic, one-line function definitions, `@parallel [ranges] f!(…)` launches
#   (without ranges: 1:max over the array arguments of size(·,d) — ParallelStencil's rule for @parallel_indices kernels),
#   update_halo!(…) = no-op (one rank).
# Arithmetic runs on Python floats (IEEE doubles, no contraction) in source order.
# =====================================================================================================================
import itertools
import math


class _End:
    """`end` inside an index: the extent of that dimension (minus k)"""

    def __init__(self, k=0):
        self.k = k

    def __sub__(self, k):
        return _End(self.k + k)


class OneBased:
    """a Julia array view of a NumPy array: 1-based, bounds-checked"""

    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def _ix(self, idx):
        idx = idx if isinstance(idx, tuple) else (idx,)
        idx = tuple(self.a.shape[d] - i.k if isinstance(i, _End) else i for d, i in enumerate(idx))
        for d, i in enumerate(idx):
            if not (isinstance(i, (int, np.integer)) and 1 <= i <= self.a.shape[d]):
                raise IndexError("BoundsError: %r in an array of size %r" % (idx, self.a.shape))
        return tuple(int(i) - 1 for i in idx)

    def __getitem__(self, idx):
        return float(self.a[self._ix(idx)])

    def __setitem__(self, idx, v):
        self.a[self._ix(idx)] = v


def _jl_size(A, d):
    return A.shape[d - 1]


def _jl_inb(A, *idx):
    return all(1 <= i <= A.shape[d] for d, i in enumerate(idx))


def _jl_floor_int(x):
    if not (-9.223372036854775808e18 <= x < 9.223372036854775808e18):      # Julia: InexactError (NaN included)
        raise OverflowError("InexactError: Int64(%r) — the reference program would stop here" % x)
    return int(math.floor(x))


def _jl_clamp(x, lo, hi):
    return lo if x < lo else (hi if x > hi else x)


def _jl_rem(a, b):
    return math.fmod(a, b)


def _expr(code):
    code = code.replace("&&", " and ").replace("||", " or ")
    code = re.sub(r"floor\(\s*Int\s*,", "_floor_int(", code)
    code = re.sub(r"checkbounds\(\s*Bool\s*,", "_inb(", code)
    code = re.sub(r"\bclamp\(", "_clamp(", code)
    code = re.sub(r"\bsize\(", "_size(", code)
    code = re.sub(r"([\w.]+|\([^()]*\))\s*%\s*1\b", r"_rem(\1, 1)", code)
    code = re.sub(r"([\w∇τ]+)!\(", r"\1_b(", code)
    code = re.sub(r"(?<![\w.])(\d+(?:\.\d+)?)([^\W\d_eE]\w*)", r"(\1*\2)", code)
    return _py_names(code)


def transpile_function(header, body):
    """header: 'name(args)'; body: stripped lines → Python source of a function `name_b(args)`"""
    name, args = re.match(r"([\w!∇τ]+)\((.*)\)", header).groups()
    src = ["def %s(%s):" % (_expr(name + "(")[:-1], _py_names(args))]
    ind = 1
    for ln in body:
        if re.match(r"^return\b", ln):
            continue
        if ln == "end":
            ind -= 1
            continue
        m = re.match(r"^if\s+(.*)$", ln)
        if m:
            src.append("    " * ind + "if " + _expr(m.group(1)) + ":")
            ind += 1
            continue
        if ln == "else":
            src.append("    " * (ind - 1) + "else:")
            continue
        m = re.match(r"^@parallel\s+(\(.*?\)\s+)?([\w!∇τ]+)\((.*)\)\s*$", ln)
        if m:
            rng = "None" if not m.group(1) else "[" + ", ".join(
                "(%s, %s)" % tuple(_expr(p) for p in r.split(":")) for r in _split(m.group(1).strip()[1:-1])) + "]"
            src.append("    " * ind + "_launch(%r, %s, [%s])" % (m.group(2), rng, _py_names(m.group(3))))
            continue
        if re.match(r"^update_halo!\(", ln):
            src.append("    " * ind + "pass")
            continue
        src.append("    " * ind + _expr(ln))
    if len(src) == 1:
        src.append("    pass")
    return "\n".join(src)


def _split(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "(["
        depth -= ch in ")]"
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Script:
    """every kernel and helper of one reference script, executable"""

    def __init__(self, script, ref=REF):
        self.kernels = parse_kernels(script, ref)
        lines = open(os.path.join(ref, SCRIPTS[script]), encoding="utf-8").read().split("\n")
        self.env = {"_size": _jl_size, "_inb": _jl_inb, "_floor_int": _jl_floor_int, "_clamp": _jl_clamp, "_rem": _jl_rem,
                    "_launch": self.launch, "end": _End(), "__builtins__": {}}
        self.source = {}
        i = 0
        while i < len(lines):
            ln = lines[i].split("#")[0].rstrip()
            m1 = re.match(r"^(?:@inline\s+)?([\w!]+\([^=]*\))\s*=\s*(.+)$", ln)                      # lerp(a,b,t) = …
            m2 = re.match(r"^(?:@inline\s+)?function\s+([\w!∇τ]+\(.*\))\s*$", ln)                    # plain functions
            m3 = re.match(r"^@parallel_indices\s+\((.*?)\)\s+function\s+([\w!∇τ]+)\((.*)\)\s*$", ln)  # index kernels
            if m1 and not ln.startswith("max_g") and "MPI" not in ln:
                self._define(m1.group(1), ["return_ = 0"], ret=m1.group(2))
            elif m2 or m3:
                body, depth, i = [], 0, i + 1
                while True:
                    code = lines[i].split("#")[0].strip()
                    if code == "end" and depth == 0:
                        break
                    if re.match(r"^(if|for|while)\b", code):
                        depth += 1
                    if code == "end":
                        depth -= 1
                    if code:
                        body.append(code)
                    i += 1
                if m3:
                    self._define("%s(%s, %s)" % (m3.group(2), m3.group(1), m3.group(3)), body)
                elif re.match(r"(backtrack!|set_bc_Vel!|set_bc_Pr!)\(", m2.group(1)):
                    self._define(m2.group(1), body)
            i += 1

    def _define(self, header, body, ret=None):
        src = transpile_function(header, [] if ret else body)
        if ret:
            src = src.replace("    pass", "    return " + _expr(ret))
        name = re.match(r"def (\w+)\(", src).group(1)
        self.source[name] = src
        exec(src, self.env)

    def launch(self, name, ranges, args):
        k = self.kernels[name]
        if k["kind"] == "stencil":
            run_stencil(k, dict(zip(k["args"], [a.a if isinstance(a, OneBased) else a for a in args])))
            return
        arrays = [a for a in args if isinstance(a, OneBased)]
        if ranges is None:
            ranges = [(1, max(a.shape[d] for a in arrays if len(a.shape) > d)) for d in range(len(k["ivars"]))]
        fn = self.env[_expr(name + "(")[:-1]]
        for tup in itertools.product(*[range(lo, hi + 1) for lo, hi in ranges]):
            fn(*tup, *args)

    def call(self, name, *args):
        """name!(args…) as the script would call it; NumPy arrays are wrapped as 1-based views and modified in place"""
        wrapped = [OneBased(a) if isinstance(a, np.ndarray) else a for a in args]
        if name in self.kernels:
            self.launch(name, None, wrapped)
        else:
            self.env[_expr(name + "(")[:-1]](*wrapped)


def cases2(script, grid):
    """(callee, ordered argument list) for the plain-Julia kernels and the host BC sequences"""
    nx, ny, nz = grid
    c, vx, vy, vz = (nx, ny, nz), (nx + 1, ny, nz), (nx, ny + 1, nz), (nx, ny, nz + 1)
    A = lambda seed, shp: rnd(5000 * seed + nx + 7 * ny + 13 * nz, shp)
    dx, dy, dz = 1.0 / nx, 0.6 / ny, 0.7 / nz
    lx, ly, lz = 1.0, 0.6, 0.7
    cyl = [(0.12 * lx) ** 2, (0.2 * lx) ** 2, -0.1 * lx, 0.05 * lx, math.sin(0.3), math.cos(0.3)]
    out = []
    if script == "multi":
        out.append(("set_cylinder!", [A(1, c), A(2, vx), A(3, vy), A(4, vz)] + cyl + [-(lx - dx) / 2, -(ly - dy) / 2, -(lz - dz) / 2, lx, ly, lz, dx, dy, dz]))
        for inlet, q in ((-lx / 2, 0), (0.0, 1)):
            out.append(("set_bc_Vel!", [A(5 + q, vx), A(7 + q, vy), A(9 + q, vz), inlet, lx, 1.25]))
        for outlet, q in ((lx / 2, 0), (0.0, 1)):
            out.append(("set_bc_Pr!", [A(11 + q, c), outlet, lx, 0.375]))
    else:
        out.append(("set_cylinder!", [A(1, c), A(2, vx), A(3, vy), A(4, vz)] + cyl + [lx, ly, lz, dx, dy, dz]))
        out.append(("set_bc_Vel!", [A(5, vx), A(7, vy), A(9, vz), rnd(77, (nz,))]))
        out.append(("set_bc_Pr!", [A(11, c), dz, nz, 9.81, 1000.0]))
    for cfl, q in ((0.4, 0), (1.0, 1), (2.3, 2)):        # |δ| < 1, up to 1, and beyond (the clamps act)
        V = [A(20 + 3 * q, vx) * cfl, A(21 + 3 * q, vy) * cfl, A(22 + 3 * q, vz) * cfl]
        dt = min(dx, dy, dz)
        out.append(("advect!", [A(30 + q, vx), V[0], A(33 + q, vy), V[1], A(36 + q, vz), V[2], A(39 + q, c), A(42 + q, c), dt, dx, dy, dz]))
    return out


def evaluate_all2(ref=REF):
    res = {}
    for script in SCRIPTS:
        sc = Script(script, ref)
        for grid in GRIDS:
            for q, (name, args) in enumerate(cases2(script, grid)):
                sc.call(name, *args)
                for j, v in enumerate(args):
                    if isinstance(v, np.ndarray) and v.ndim == 3:
                        res["%s/%dx%dx%d/p2_%02d/%s/%d" % (script, grid[0], grid[1], grid[2], q, name, j)] = v
    return res


# =====================================================================================================================
# Part 3 — the two DRIVERS, from their source text: the setup block (physics, numerics, allocation, initial conditions) and the
# time loop of run_navierstokes3D (multi.jl:288-373, 446-477) and runme (gpu.jl:13-88, 119-142), one rank, no plots / files.
# The same line-by-line transpiler with a few more constructs: `for v = a:b`, one-line `if … end`, `;`, `^`, comprehensions,
# `@zeros`, LinRange (element i = (1−t)·a + t·b, t = (i−1)/(n−1)), `.=`, push!, break, sincos, ceil(Int,·), maximum(abs.(·)).
# ImplicitGlobalGrid for ONE rank [upstream]: n_g() = n, x_g(i,dx,A) = (i−1)·dx + ½(n − size(A,1))·dx, update_halo! = no-op,
# max_g = maximum.  Lines that print, plot or save are not part of the blocks taken.
# =====================================================================================================================
class _LinRange:
    def __init__(self, a, b, n):
        self.a, self.b, self.n = float(a), float(b), int(n)
        self.shape = (self.n,)

    def __getitem__(self, i):
        t = (i - 1) / (self.n - 1)
        return (1 - t) * self.a + t * self.b


def _expr2(code):
    code = re.sub(r"Float64\[\]", "[]", code)
    code = re.sub(r"push!\(\s*(\w+)\s*,", r"_push(\1,", code)
    code = code.replace("@zeros(", "_zeros(").replace("Data.Array(", "_ident(").replace("LinRange(", "_LinRange(")
    code = re.sub(r"ceil\(\s*Int\s*,", "_ceil_int(", code)
    code = re.sub(r"abs\.\(", "_absdot(", code)
    code = re.sub(r"\b(maximum|max_g)\(", "_maximum(", code)
    code = re.sub(r"\bsincos\(", "_sincos(", code)
    code = re.sub(r"\bsqrt\(", "_sqrt(", code)
    code = re.sub(r"\bisfinite\(", "_isfinite(", code)
    code = re.sub(r"\bInf\b", "_inf", code).replace("π", "_pi")
    code = code.replace("^", "**")
    code = _expr(code)
    code = re.sub(r"!(?!=)", " not ", code)
    return code


def _comprehension(code):
    """[EXPR for i=R1,j=R2,…]  →  _compr(lambda i,j,…: EXPR, [(lo,hi),…])   (ranges `a:b`, possibly parenthesised)"""
    m = re.search(r"\[(.*)\sfor\s(.*)\]", code)
    if not m:
        return code
    its = _split(m.group(2))
    names = [q.split("=")[0].strip() for q in its]
    rngs = []
    for q in its:
        lo, hi = q.split("=", 1)[1].strip().split(":")
        rngs.append("(%s, %s)" % (lo.strip(" ("), hi.strip()))
    return code[:m.start()] + "_compr(lambda %s: %s, [%s])" % (", ".join(names), m.group(1).strip(), ", ".join(rngs)) + code[m.end():]


def _statement(st):
    """one Julia statement → one line of Python, or None (skipped)"""
    st = st.strip()
    if not st or re.match(r"^(return\b|[\w∇τεβ]+$)", st):                       # bare identifiers (`nx`, `nt`) are no-ops
        return None
    m = re.match(r"^@parallel\s+(\(.*?\)\s+)?([\w!∇τ]+)\((.*)\)\s*$", st)
    if m:
        return "_launch(%r, None, [%s])" % (m.group(2), _py_names(m.group(3)))
    if re.match(r"^update_halo!\(", st):
        return None
    m = re.match(r"^([\w∇τ]+)\[(.*?)\]\s*\.=\s*(.*)$", st)                       # Vy[1,:,:] .= vin
    if m:
        parts = ", ".join("None" if q.strip() == ":" else q.strip() for q in m.group(2).split(","))
        return "_setslice(%s, (%s), %s)" % (_py_names(m.group(1)), parts, _expr2(m.group(3)))
    m = re.match(r"^([\w∇τ]+)\s*\.=\s*([\w∇τ]+)$", st)                           # Vx_o .= Vx
    if m:
        return "_copy(%s, %s)" % (_py_names(m.group(1)), _py_names(m.group(2)))
    if st == "break":
        return "break"
    return _expr2(_comprehension(st))


def transpile_block(lines, name, args, hook=None):
    src, ind = ["def %s(%s):" % (name, ", ".join(args))], 1
    for raw in lines:
        ln = raw.split("#")[0].strip()
        if not ln:
            continue
        if re.search(r"println|@printf", ln):
            continue
        if ln == "end":
            ind -= 1
            if hook and ind == 1:
                src.append("        " + hook)
            continue
        m = re.match(r"^for\s+(\w+)\s*=\s*(.+?):(.+)$", ln)
        if m:
            src.append("    " * ind + "for %s in range(%s, (%s) + 1):" % (m.group(1), _expr2(m.group(2)), _expr2(m.group(3))))
            ind += 1
            continue
        m = re.match(r"^if\s+(.*?)\s+(break)\s+end$", ln)                         # one-line `if cond break end`
        if m:
            src.append("    " * ind + "if %s:" % _expr2(m.group(1)))
            src.append("    " * (ind + 1) + "break")
            continue
        m = re.match(r"^if\s+(.*)$", ln)
        if m:
            src.append("    " * ind + "if %s:" % _expr2(m.group(1)))
            ind += 1
            continue
        for st in ln.split(";"):
            py = _statement(st)
            if py:
                src.append("    " * ind + py)
    src.append("    return locals()" if not hook else "    return None")
    return "\n".join(src)


class Driver(Script):
    """run_navierstokes3D / runme evaluated from the script's text (one rank; do_vis = do_save = do_print = false)"""

    def __init__(self, script, ref=REF):
        super().__init__(script, ref)
        self.script = script
        lines = open(os.path.join(ref, SCRIPTS[script]), encoding="utf-8").read().split("\n")
        head = next(i for i, l in enumerate(lines) if re.match(r"^@views function (run_navierstokes3D|runme)\(", l))
        stop = next(i for i in range(head, len(lines)) if re.match(r"^\s*(#init stuff|if do_save\b)", lines[i])
                    or "# Initialization for saving" in lines[i])
        loop = next(i for i in range(stop, len(lines)) if re.match(r"^\s*for it = 1:nt", lines[i]))
        cut = next(i for i in range(loop, len(lines)) if re.match(r"^\s*(# Visualization|if \(?do_vis)", lines[i]))
        self.setup_lines = lines[head + 1:stop]
        self.loop_lines = lines[loop:cut] + ["end"]
        self.grid = {}
        e = self.env
        e.update({"_zeros": lambda *s: OneBased(np.zeros(s, order="F")), "_ident": lambda x: x, "_LinRange": _LinRange,
                  "_ceil_int": lambda x: int(math.ceil(x)), "_absdot": lambda A: np.abs(A.a), "_maximum": lambda a: float(np.max(a)),
                  "_sincos": lambda x: (math.sin(x), math.cos(x)), "_sqrt": math.sqrt, "_isfinite": math.isfinite,
                  "_inf": math.inf, "_pi": math.pi, "max": max, "min": min, "range": range, "locals": locals,
                  "_push": lambda lst, v: lst.append(v), "_compr": self._compr, "_setslice": self._setslice,
                  "_copy": lambda A, B: A.a.__setitem__(Ellipsis, B.a), "init_global_grid": self._init_grid,
                  "nx_g": lambda: self.grid["n"][0], "ny_g": lambda: self.grid["n"][1], "nz_g": lambda: self.grid["n"][2],
                  "x_g": lambda i, d, A: self._xg(0, i, d, A), "y_g": lambda i, d, A: self._xg(1, i, d, A),
                  "z_g": lambda i, d, A: self._xg(2, i, d, A)})

    def _init_grid(self, nx, ny, nz):
        self.grid["n"] = (nx, ny, nz)
        return 0, [1, 1, 1]

    def _xg(self, dim, i, d, A):
        n = self.grid["n"][dim]
        return (0 * (n - 2) + (i - 1)) * d + 0.5 * (n - A.shape[dim]) * d

    @staticmethod
    def _compr(fn, rngs):
        shape = tuple(hi - lo + 1 for lo, hi in rngs)
        out = np.zeros(shape, order="F")
        for tup in itertools.product(*[range(lo, hi + 1) for lo, hi in rngs]):
            out[tuple(t - lo for t, (lo, _) in zip(tup, rngs))] = fn(*tup)
        return OneBased(out)

    @staticmethod
    def _setslice(A, parts, v):
        A.a[tuple(slice(None) if q is None else q - 1 for q in parts)] = v

    def run(self, nx, nt, niter_cap=None):
        """returns (final state dict name → array, [iterations per step], [err history per step])"""
        over = {"nx": nx, "nt": nt}
        setup = [l for l in self.setup_lines if not re.match(r"^\s*(nx|nt)\s*=\s*\d", l)]      # gpu.jl's literals nx = 255, nt = 10000
        src = transpile_block(setup, "_setup", ["nx", "nt", "do_vis", "do_save", "do_print"])
        self.source["_setup"] = src
        exec(src, self.env)
        st = self.env["_setup"](nx, nt, False, False, False)
        st.update(over)
        if niter_cap is not None:
            st["niter"] = min(st["niter"], niter_cap)
        st.setdefault("me", 0)
        iters, errs = [], []
        self.env["_hook"] = lambda it, it_inner, err_evo: (iters.append(it_inner), errs.append(list(err_evo)))
        names = sorted(k for k in st if re.match(r"^[^\W\d]\w*$", k) and k not in ("range", "locals", "max", "min", "iter"))
        lsrc = transpile_block(self.loop_lines, "_loop", names, hook="_hook(it, iter, err_evo)")
        self.source["_loop"] = lsrc
        exec(lsrc, self.env)
        self.env["_loop"](*[st[k] for k in names])
        return {k: v.a for k, v in st.items() if isinstance(v, OneBased)}, iters, errs, st


DRIVER_CASES = [("multi", 20, 3, None), ("multi", 36, 3, None), ("gpu", 16, 2, 60), ("gpu", 20, 2, None)]
FIELDS = ["Pr", "dPrdτ", "C", "C_o", "τxx", "τyy", "τzz", "τxy", "τxz", "τyz", "Vx", "Vy", "Vz", "Vx_o", "Vy_o", "Vz_o", "∇V", "Rp"]
SCALARS = ["dx", "dy", "dz", "dt", "dτ", "damp", "niter", "nchk", "psc", "g", "a2", "b2", "ox", "oy", "sinβ", "cosβ", "lx", "ly", "lz",
           "εit", "ny", "nz"]


def evaluate_drivers(ref=REF):
    """{key: array} — final fields, derived scalars, iterations per step and error histories of every DRIVER_CASES run"""
    res = {}
    for script, nx, nt, cap in DRIVER_CASES:
        fields, iters, errs, st = Driver(script, ref).run(nx, nt, cap)
        pre = "%s/nx%d_nt%d/" % (script, nx, nt)
        for f in FIELDS:
            res[pre + "field/" + f] = fields[_py_names(f)]
        res[pre + "scalars"] = np.array([float(st[k]) for k in SCALARS])
        res[pre + "iters"] = np.array(iters, dtype=np.int64)
        res[pre + "errs"] = np.array([e for es in errs for e in es], dtype=np.float64)
        res[pre + "errs_per_step"] = np.array([len(es) for es in errs], dtype=np.int64)
    return res


def generate_driver_goldens():
    res = evaluate_drivers()
    path = os.path.join(ROOT, "tests", "golden", "jl_eval_drivers.npz")
    np.savez_compressed(path, **res)
    print(len(res), "arrays →", path)


def field_digest(a):
    """sha256 of the array's bytes in memory order of a column-major copy (what a bit-for-bit comparison needs, in 64 characters)"""
    import hashlib
    return hashlib.sha256(np.asfortranarray(a).tobytes(order="F")).hexdigest()


CONFIG_A = ("multi", 63, 3)      # BASELINE configs[0]'s grid (63×38×38), the first three time steps: 37, 259, 296 PT iterations


def generate_config_a_digest():
    """the reference's own CPU-runnable configuration from its text (≈1–2 min of Python): digests instead of 13 MB of arrays"""
    import json
    script, nx, nt = CONFIG_A
    fields, iters, errs, st = Driver(script).run(nx, nt)
    out = {"script": script, "nx": nx, "nt": nt, "iters": iters, "errs_hex": [[float(e).hex() for e in es] for es in errs],
           "scalars_hex": {k: float(st[k]).hex() for k in SCALARS}, "shape": {f: list(fields[_py_names(f)].shape) for f in FIELDS},
           "sha256": {f: field_digest(fields[_py_names(f)]) for f in FIELDS},
           "max_abs": {f: float(np.abs(fields[_py_names(f)]).max()) for f in FIELDS}}
    # what test/test3D.jl:12-17 samples — Pr of run_navierstokes3D(nx=63, nt=1), halo-stripped, at inds_x × inds_y × inds_z —
    # from the committed script's own text (the fixture in that test holds 0.2 … 0.6 at its hot spot: it is stale)
    f1, it1, _, _ = Driver(script).run(63, 1)
    Pv = f1["Pr"][1:-1, 1:-1, 1:-1]
    ix, iy, iz = [31, 38, 50, 51], [2, 5, 19, 31], [12, 13, 23, 23]
    out["test3D_nt1"] = {"iters": it1, "Pr_samples": [[[float(Pv[x - 1, y - 1, z - 1]) for x in ix] for y in iy] for z in iz],
                         "max_abs_Pr": float(np.abs(f1["Pr"]).max())}
    path = os.path.join(ROOT, "tests", "golden", "jl_eval_config_a.json")
    json.dump(out, open(path, "w"), indent=1, ensure_ascii=False)
    print("config A digest →", path, iters)


if __name__ == "__main__":
    import sys
    generate_goldens()
    generate_driver_goldens()
    if "--config-a" in sys.argv:
        generate_config_a_digest()
