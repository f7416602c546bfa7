#!/usr/bin/env python3
"""Names the reference's hot-path files use and define -- DATA for tests/test_julia_shim.py, no source text.

Scans the seven SURVEY 8a files under /root/reference (run in the build container; the GPU box has no reference) and writes
tests/golden/reference_surface.json: every macro used (with the number of uses OUTSIDE kernel-definition bodies), every
kernel defined with @parallel_indices / @parallel function, every macro and top-level function the files define themselves,
and which names of the ParallelStencil / ImplicitGlobalGrid surface they call."""
import json, os, re, sys

REF = "/root/reference"
FILES = ["scripts-part1/part1_kernel_programming.jl", "scripts-part1/part1_array_programming.jl", "scripts-part1/part1_utils.jl",
         "scripts-part2/multigrid.jl", "scripts-part2/krylov.jl", "scripts-part2/part2_utils.jl", "scripts-part2/part2.jl"]
SURFACE = ["init_global_grid", "finalize_global_grid", "select_device", "update_halo!", "gather!", "nx_g", "ny_g", "nz_g", "x_g", "y_g",
           "z_g", "Data.Array", "Data.Number"]

out = {"files": FILES, "macros": {}, "kernels": [], "macros_defined": [], "functions_defined": [], "surface_calls": []}
for f in FILES:
    lines = open(os.path.join(REF, f), encoding="utf-8").read().split("\n")
    in_kernel = [False] * len(lines)
    i = 0
    while i < len(lines):
        m = re.match(r"^@parallel(_indices)?\s*(\([^)]*\)\s*)?function\s+([^\s(]+)\s*\(", lines[i])
        if m:
            out["kernels"].append({"name": m.group(3), "file": f, "line": i + 1, "form": "@parallel_indices" if m.group(1) else "@parallel function"})
            j = i
            while j < len(lines) and not re.match(r"^end\b", lines[j]):
                in_kernel[j] = True
                j += 1
            in_kernel[i] = False      # the definition line itself (its macro must be understood)
            i = j
        i += 1
    for i, l in enumerate(lines):
        code = l.split("#")[0]
        for m in re.finditer(r"@([A-Za-z_][A-Za-z_0-9]*)", code):
            e = out["macros"].setdefault(m.group(1), {"uses": 0, "outside_kernel_bodies": 0})
            e["uses"] += 1
            e["outside_kernel_bodies"] += not in_kernel[i]
        m = re.match(r"^macro\s+([A-Za-z_][A-Za-z_0-9]*)", code)
        if m: out["macros_defined"].append(m.group(1))
        m = re.match(r"^(?:@views\s+)?function\s+([^\s(]+)\s*\(", code)
        if m: out["functions_defined"].append({"name": m.group(1), "file": f})
        for s in SURFACE:
            if re.search(r"(?<![A-Za-z_0-9.])" + re.escape(s) + r"\s*[\(\[]", code) and s not in out["surface_calls"]:
                out["surface_calls"].append(s)
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_surface.json")
json.dump(out, open(dst, "w"), indent=1, ensure_ascii=False, sort_keys=True)
print(dst, len(out["kernels"]), "kernels", len(out["macros"]), "macros")
