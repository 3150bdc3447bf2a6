"""Generates tests/golden/reference_call_sites.json: the CALL SITES the reference's two scripts make into what a drop-in has to
provide — every `@parallel [ranges] kernel!(args…)` call and every ImplicitGlobalGrid / ParallelStencil name they use — as
data: script, line, callee, number of positional arguments, whether launch ranges are given.  TEST INFRASTRUCTURE: the shim
(julia/NS3DShim.jl) is checked against this list without Julia and without the reference at test time.

    python oracle/extract_call_sites.py [/root/reference]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "([{"
        depth -= ch in ")]}"
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def closing(s, i):
    depth = 0
    for j in range(i, len(s)):
        depth += s[j] == "("
        if s[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    raise ValueError


GRID_NAMES = ["init_global_grid", "finalize_global_grid", "update_halo!", "gather!", "nx_g", "ny_g", "nz_g", "x_g", "y_g", "z_g"]


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {"parallel_calls": [], "grid_calls": [], "macros": {}}
    for script in ("NavierStokes3D_multi_gpu.jl", "NavierStokes3D_gpu.jl"):
        lines = open(os.path.join(ref, "scripts", script), encoding="utf-8").read().split("\n")
        for ln, raw in enumerate(lines, 1):
            line = raw.split("#")[0]
            m = re.search(r"@parallel\s+(\([^@]*?\)\s+)?([\w!∇τ]+)\(", line)
            if m and "function" not in line:
                start = line.index("(", m.end() - 1)
                args = split_top(line[start + 1:closing(line, start)])
                out["parallel_calls"].append({"script": script, "line": ln, "callee": m.group(2), "nargs": len(args),
                                              "ranges": bool(m.group(1))})
            for name in GRID_NAMES:
                for g in re.finditer(r"(?<![\w!])" + re.escape(name) + r"\(", line):
                    if re.search(r"function\s+$", line[:g.start()]):
                        continue
                    start = g.end() - 1
                    args = split_top(line[start + 1:closing(line, start)])
                    out["grid_calls"].append({"script": script, "line": ln, "callee": name, "nargs": len(args)})
            for mac in ("@init_parallel_stencil", "@zeros", "@parallel_indices", "@parallel", "Data.Array", "Data.Number"):
                if mac in line:
                    out["macros"][mac] = out["macros"].get(mac, 0) + 1
    path = os.path.join(ROOT, "tests", "golden", "reference_call_sites.json")
    json.dump(out, open(path, "w"), indent=1, ensure_ascii=False)
    print(len(out["parallel_calls"]), "kernel call sites,", len(out["grid_calls"]), "grid calls →", path)


if __name__ == "__main__":
    main()
