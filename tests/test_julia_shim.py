"""julia/FPRHip.jl against the surface the reference's seven hot-path files use -- no GPU, no julia.

tests/golden/reference_surface.json holds NAMES only (tools/make_reference_surface.py reads them off /root/reference in the
build container): every macro the files use, how often outside kernel-definition bodies, every kernel they define with
@parallel_indices / `@parallel function`, the macros / functions they define themselves and the ParallelStencil /
ImplicitGlobalGrid names they call.  The shim is unexecuted here (no julia in the image), so this is a static check that
nothing the files need is missing: with it, the only edit to a reference file is its `using` block (or none at all through
`include_reference`)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = open(os.path.join(ROOT, "julia", "FPRHip.jl"), encoding="utf-8").read()
SURF = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_surface.json"), encoding="utf-8"))

# macros of Base / the standard library (Printf, Logging) the files use outside kernel bodies
BASE_MACROS = {"views", "assert", "warn", "debug", "enum", "sprintf", "static", "isdefined", "__DIR__", "__FILE__"}


def shim_macros():
    return set(re.findall(r"^macro\s+([A-Za-z_][A-Za-z_0-9]*)", SHIM, flags=re.M))


def shim_exports():
    m = re.search(r"^export\s+(.*?)\n\n", SHIM, flags=re.M | re.S)
    return {t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip()}


def shim_defined():
    names = set(re.findall(r"^function\s+([^\s(]+)\s*\(", SHIM, flags=re.M))
    names |= set(re.findall(r"^const\s+([^\s=]+)\s*=", SHIM, flags=re.M))
    names |= set(re.findall(r"^([A-Za-z_][^\s(=]*)\(.*\)\s*=", SHIM, flags=re.M))
    return names


def shim_kernel_set():
    m = re.search(r"const KERNELS = Set\{Symbol\}\(\[(.*?)\]\)", SHIM, flags=re.S)
    body = m.group(1)
    return set(re.findall(r'Symbol\("([^"]+)"\)', body)) | set(re.findall(r"(?<![A-Za-z_\"]):([A-Za-z_][^\s,\]]*)", body))


def test_every_kernel_definition_of_the_reference_is_swallowed_and_provided():
    kernels = {k["name"] for k in SURF["kernels"]}
    assert len(SURF["kernels"]) == 15 and len(kernels) == 15          # 14 definitions by @parallel_indices + 1 by `@parallel function`
    assert sum(k["form"] == "@parallel_indices" for k in SURF["kernels"]) == 14
    assert shim_kernel_set() == kernels, (shim_kernel_set() ^ kernels)
    defined, exported = shim_defined(), shim_exports()
    for k in kernels:
        assert k in defined, "FPRHip.jl swallows the definition of %s but does not define it" % k
        assert k in exported, "FPRHip.jl does not export %s" % k
    # the definition forms themselves
    assert {"parallel", "parallel_indices"} <= shim_macros()
    assert "swallow_kernel_definition(\"@parallel_indices\"" in SHIM and "swallow_kernel_definition(\"@parallel\"" in SHIM


def test_every_macro_used_outside_kernel_bodies_is_understood():
    mine = shim_macros()
    exported = shim_exports()
    for name, use in SURF["macros"].items():
        if use["outside_kernel_bodies"] == 0:
            continue        # only ever inside a swallowed kernel body (@threadIdx, @sharedMem, @all, @qx ...): never expanded
        if name in BASE_MACROS or name in SURF["macros_defined"]:
            continue
        assert name in mine, "the reference uses @%s outside kernel bodies; FPRHip.jl has no such macro" % name
        assert "@" + name in exported, "FPRHip.jl does not export @%s" % name
    # macros that exist only inside kernel bodies must really be unreachable: none of them is defined or needed
    body_only = {n for n, u in SURF["macros"].items() if u["outside_kernel_bodies"] == 0}
    assert {"threadIdx", "blockDim", "sharedMem", "sync_threads", "atomic", "all", "inn"} <= body_only


def test_parallelstencil_and_implicitglobalgrid_surface_is_exported():
    exported = shim_exports()
    for name in SURF["surface_calls"]:
        head = name.split(".")[0]
        assert head in exported, "the reference calls %s; FPRHip.jl does not export %s" % (name, head)
    assert "module Data" in SHIM and "const Array = ROCArray{Float64}" in SHIM


def test_include_reference_drops_only_what_the_shim_provides():
    ref_funcs = {f["name"] for f in SURF["functions_defined"]}
    m = re.search(r"const NATIVE_HOST = Set\{Tuple\{Symbol,Int\}\}\(\[(.*?)\]\)", SHIM, flags=re.S)
    native = set(re.findall(r'Symbol\("([^"]+)"\)', m.group(1))) | set(re.findall(r"\(:([A-Za-z_][A-Za-z_0-9]*),", m.group(1)))
    assert native, "NATIVE_HOST not found"
    defined = shim_defined()
    for n in native:
        assert n in ref_funcs, "NATIVE_HOST names %s, which the reference files do not define" % n
        assert n in defined, "include_reference(fast=true) drops %s but FPRHip.jl does not define it" % n
    # the files' own driver functions must survive the filter
    for keep in ("diffusion_3D_kernel_programming", "navier_stokes_2D", "init_local_gaussian", "linear_interpolate_3D", "compute_dt"):
        assert keep in ref_funcs and keep not in native
    for pkg in ("CUDA", "ParallelStencil", "ImplicitGlobalGrid"):
        assert ":" + pkg in SHIM.split("const ABSENT_PACKAGES")[1].split("\n")[0]


# ----------------------------------------------------------------------------------------------------------------------
# Every `ccall((:fpr_x, libfpr), Ret, (types...), args...)` of the shim against the prototype in include/fpr.h: the shim is
# unexecuted here, and a Cint where the header has a double (or one argument too few) is silent memory corruption on a
# maintainer's machine.  Also: every block closes, every exported name is defined.
# ----------------------------------------------------------------------------------------------------------------------
def _strip_c_comments(txt):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", txt, flags=re.S))


def _split_top(s, sep=","):
    """Split at `sep` outside (), [] and {}."""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        out.append("".join(cur))
    return [x.strip() for x in out]


def _c_type_class(decl, is_return=False):
    """A C parameter declaration -> the class of Julia types that may stand for it."""
    d = re.sub(r"\bconst\b", " ", decl).strip()
    if "[" in d or "*" in d:
        # (an array parameter `const int lo[3]` is a pointer)
        base = re.sub(r"[\*\[].*", "", d).split()
        base = base[0] if base else ""
        stars = d.count("*") + (1 if "[" in d else 0)
        if base == "char":
            return "cstring"
        if base == "double":
            return "ptr_double" if stars == 1 else "ptr_ptr"
        if base == "int":
            return "ptr_int"
        if base == "long":
            return "ptr_long"
        if base == "void" and stars == 1:
            return "ptr_any"                                  # void*: raw bytes, any Ptr{T}
        return "ptr_void" if stars == 1 else "ptr_ptr"      # fpr_ctx*, void**
    base = d.split()[0] if not is_return else d
    if base.split()[0].endswith("_fn"):
        return "ptr_void"                                     # a function-pointer typedef (fpr_place_trial_fn): Ptr{Cvoid} from @cfunction
    return {"int": "int", "long": "long", "double": "double", "size_t": "size_t", "void": "void"}[base.split()[0]]


JULIA_OK = {
    "int": {"Cint"}, "long": {"Clong"}, "double": {"Cdouble"}, "size_t": {"Csize_t"}, "void": {"Cvoid"},
    "cstring": {"Cstring", "Ptr{UInt8}", "Ptr{Cchar}"},
    "ptr_double": {"Ptr{Cdouble}", "Ptr{Float64}", "Ptr{Cvoid}"},       # (a NULL-able double* may be passed as C_NULL :: Ptr{Cvoid})
    "ptr_int": {"Ptr{Cint}"}, "ptr_long": {"Ptr{Clong}"},
    "ptr_void": {"Ptr{Cvoid}"},
    "ptr_ptr": {"Ptr{Ptr{Cvoid}}", "Ptr{Ptr{Cdouble}}", "Ptr{Ptr{Float64}}", "Ptr{Cvoid}"},
}


def header_prototypes():
    txt = _strip_c_comments(open(os.path.join(ROOT, "include", "fpr.h")).read())
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(fpr_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        ps = [] if params in ("", "void") else _split_top(params)
        protos[name] = (_c_type_class(ret, True) if "*" not in ret else ("cstring" if "char" in ret else "ptr_void"),
                        [_c_type_class(p) for p in ps], ps)
    return protos


def shim_ccalls():
    """(name, return type, [argument types], number of values passed, line) of every ccall into libfpr."""
    out = []
    for m in re.finditer(r"ccall\(\(:(fpr_[A-Za-z0-9_]+),\s*(?:FPRHip\.)?libfpr\)", SHIM):
        # the balanced argument list of this ccall
        i = SHIM.index("(", m.start())
        depth, j = 0, i
        while True:
            depth += SHIM[j] == "("
            depth -= SHIM[j] == ")"
            if depth == 0:
                break
            j += 1
        body = re.sub(r"#[^\n]*", "", SHIM[i + 1:j])
        parts = _split_top(body)
        ret, types = parts[1], parts[2]
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        tl = [t for t in _split_top(types[1:-1]) if t]
        vals = parts[3:]
        splats = sum(v.endswith("...") for v in vals)           # `size(Ht)...`: as many values as the array has dimensions
        out.append((m.group(1), ret, tl, (len(vals), splats), SHIM.count("\n", 0, m.start()) + 1))
    return out


def test_every_ccall_matches_its_prototype_in_the_header():
    protos = header_prototypes()
    calls = shim_ccalls()
    assert len(protos) >= 77 and len(calls) >= 76
    bound = set()
    for name, ret, types, nvalues, line in calls:
        assert name in protos, "FPRHip.jl:%d calls %s, which include/fpr.h does not declare" % (line, name)
        pret, pclasses, pdecl = protos[name]
        where = "FPRHip.jl:%d ccall(:%s)" % (line, name)
        assert ret in JULIA_OK[pret], "%s returns %s, the header says %s" % (where, ret, pret)
        assert len(types) == len(pclasses), "%s passes %d types, the prototype has %d parameters" % (where, len(types), len(pclasses))
        nv, splats = nvalues
        if splats == 0:
            assert nv == len(types), "%s: %d types but %d values" % (where, len(types), nv)
        else:       # a splatted size(...) stands for 2 or 3 Cint in a row
            assert nv - splats + 2 * splats <= len(types) <= nv - splats + 3 * splats, "%s: %d types, %d values (%d splatted)" % (
                where, len(types), nv, splats)
        for k, (jt, pc) in enumerate(zip(types, pclasses)):
            if pc == "ptr_any":
                assert re.fullmatch(r"Ptr\{[A-Za-z0-9{}]+\}", jt), "%s: argument %d is %s, the header declares `%s`" % (where, k + 1, jt, pdecl[k])
                continue
            assert jt in JULIA_OK[pc], "%s: argument %d is %s, the header declares `%s`" % (where, k + 1, jt, pdecl[k])
        bound.add(name)
    # every compute entry point of the header is bound by at least one ccall (names in comments do not count)
    missing = sorted(set(protos) - bound)
    assert not missing, "declared in include/fpr.h but never ccall'ed by FPRHip.jl: %s" % missing


def _julia_code_lines():
    """The shim without comments, strings and docstrings (good enough for counting block keywords)."""
    txt = re.sub(r'"""(?:.|\n)*?"""', '""', SHIM)
    txt = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', txt)
    txt = re.sub(r"#=(?:.|\n)*?=#", "", txt)
    txt = re.sub(r"#[^\n]*", "", txt)
    return txt


def julia_block_scan(txt):
    """Token scan of Julia source (comments and strings already removed): returns the list of problems found -- an `end` without
    an opener, a bracket closed by the wrong kind, openers left at the end.  `for` / `if` inside brackets are generators and
    comprehensions (no `end`); `end` inside [] is an index."""
    openers = {"function", "if", "for", "while", "let", "begin", "module", "baremodule", "macro", "quote", "try", "do", "struct"}
    stack, problems = [], []
    line = 1
    prev = ""
    for m in re.finditer(r"\n|[A-Za-z_\u00a0-\uffff][A-Za-z_0-9!\u00a0-\uffff]*|[()\[\]{}]|:(?=[A-Za-z_])|\S", txt):
        t = m.group(0)
        if t == "\n":
            line += 1
            continue
        if t in "([{":
            stack.append((t, line))
        elif t in ")]}":
            want = {")": "(", "]": "[", "}": "{"}[t]
            if not stack or stack[-1][0] != want:
                problems.append("line %d: `%s` closes %r" % (line, t, stack[-1] if stack else None))
            else:
                stack.pop()
        elif prev == ":" or prev == ".":
            pass                                            # :end, :if (quoted symbols), x.end
        elif t in openers:
            in_bracket = bool(stack) and stack[-1][0] in "([{"
            if in_bracket and t in ("for", "if"):
                pass                                        # generator / comprehension
            elif t == "struct" and prev == "mutable":
                stack.append((t, line))
            else:
                stack.append((t, line))
        elif t == "end":
            if any(k == "[" for k, _ in reversed(stack[-3:])) and stack and stack[-1][0] in "([":
                pass                                        # a[end], a[(end - 1)]
            elif not stack or stack[-1][0] in "([{":
                problems.append("line %d: `end` inside %r" % (line, stack[-1] if stack else None))
            else:
                stack.pop()
        prev = t
    problems += ["%s opened at line %d never closed" % kv for kv in stack]
    return problems


def test_blocks_are_balanced():
    assert julia_block_scan("function f(x)\n  if x > 0\n a[end] = [i for i in 1:3 if i > 1]\n end\nend\n") == []
    assert julia_block_scan("function f(x)\n  if x > 0\n return 1\nend\n") != []          # (the scan does notice a missing end)
    assert julia_block_scan("f(x) = (x,\n") != []
    problems = julia_block_scan(_julia_code_lines())
    assert not problems, "FPRHip.jl: " + "; ".join(problems[:5])


def test_every_exported_name_is_defined():
    exported = shim_exports()
    defined = shim_defined() | {"@" + m for m in shim_macros()}
    defined |= set(re.findall(r"^(?:mutable\s+)?struct\s+([A-Za-z_][A-Za-z_0-9]*)", SHIM, flags=re.M))
    defined |= set(re.findall(r"^module\s+([A-Za-z_][A-Za-z_0-9]*)", SHIM, flags=re.M))
    defined |= set(re.findall(r"^\s*@enum\s+([A-Za-z_][A-Za-z_0-9]*)", SHIM, flags=re.M))
    for m in re.finditer(r"^\s*@enum\s+[A-Za-z_][A-Za-z_0-9]*\s+(.*)$", SHIM, flags=re.M):
        defined |= set(re.findall(r"([A-Za-z_][A-Za-z_0-9]*)", m.group(1)))
    defined |= set(re.findall(r"^([A-Za-z_][A-Za-z_0-9!]*)\s*=", SHIM, flags=re.M))
    defined |= set(re.findall(r"^(?:@inline\s+)?function\s+([A-Za-z_][^\s(]*)", SHIM, flags=re.M))
    defined |= set(re.findall(r"^(?:@inline\s+)?([A-Za-z_][A-Za-z_0-9!τ]*)\(", SHIM, flags=re.M))
    missing = sorted(n for n in exported if n not in defined)
    assert not missing, "exported by FPRHip.jl but not defined in it: %s" % missing


# ----------------------------------------------------------------------------------------------------------------------
# julia/test/runtests.jl: the reference's own suite (test/runtests.jl:6-9) restated against the shim.  Unexecuted like the
# shim; what can be checked without a toolchain: its blocks close, every FPRHip name it uses is exported, every reference file it
# includes is one of the seven hot-path files, every function of the reference it calls survives include_reference or is provided
# natively, every fixture it reads is committed.
# ----------------------------------------------------------------------------------------------------------------------
JTEST = open(os.path.join(ROOT, "julia", "test", "runtests.jl"), encoding="utf-8").read()


def _strip_julia(txt):
    txt = re.sub(r'"""(?:.|\n)*?"""', '""', txt)
    keep = re.findall(r'"((?:\\.|[^"\\\n])*)"', txt)
    txt = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', txt)
    txt = re.sub(r"#=(?:.|\n)*?=#", "", txt)
    txt = re.sub(r"#[^\n]*", "", txt)
    return txt, keep


def test_julia_runtests_blocks_are_balanced_and_parts_are_the_references_four():
    code, _ = _strip_julia(JTEST)
    problems = julia_block_scan(code)
    assert not problems, "julia/test/runtests.jl: " + "; ".join(problems[:5])
    assert re.findall(r"^module\s+(\w+)", code, flags=re.M) == ["Part1", "Multigrid", "Krylov", "Part2"]     # test/runtests.jl:6-9, in its order
    assert [m for m in re.findall(r"^(\w+)\.run\(\)", code, flags=re.M)] == ["Part1", "Multigrid", "Krylov", "Part2"]
    assert code.count("@testset") >= 7 and code.count("@test ") >= 14


def test_julia_runtests_uses_only_what_the_shim_exports_and_the_reference_defines():
    code, strings = _strip_julia(JTEST)
    exported = shim_exports()
    # macros: Base / Test ones, or exported by the shim
    for mac in set(re.findall(r"@([A-Za-z_][A-Za-z_0-9]*)", code)):
        if mac in {"test", "testset", "__DIR__", "__MODULE__", "views", "show"}:
            continue
        assert "@" + mac in exported, "runtests.jl uses @%s, which FPRHip.jl does not export" % mac
    # reference files included: only the seven hot-path files, each through include_reference
    files = [s_ for s_ in strings if re.fullmatch(r"[A-Za-z_0-9]+\.jl", s_) and s_ not in ("FPRHip.jl",)]
    assert files and code.count("include_reference(") == len(files)
    surf_files = {os.path.basename(f): f for f in SURF["files"]}
    for f in files:
        assert f in surf_files, f
    # names of the hot path the tests call: provided natively by the shim (exported) or defined by the reference files they include
    ref_funcs = {f["name"] for f in SURF["functions_defined"]}
    called = set(re.findall(r"(?<![\w.@:])([A-Za-z_][A-Za-z_0-9]*!?)\(", code))
    hot = {n for n in called if n in ref_funcs or n in exported}
    for must in ("diffusion_3D_array_programming", "diffusion_3D_kernel_programming", "MGsolve_2DPoisson!", "iteration_2DPoisson!",
                 "residual_2DPoisson_wrapper!", "cg!", "navier_stokes_2D", "stencil_5pt", "MGOpt", "include_reference"):
        assert must in hot, "runtests.jl does not exercise %s" % must
    m = re.search(r"const NATIVE_HOST = Set\{Tuple\{Symbol,Int\}\}\(\[(.*?)\]\)", SHIM, flags=re.S)
    native = set(re.findall(r'Symbol\("([^"]+)"\)', m.group(1))) | set(re.findall(r"\(:([A-Za-z_][A-Za-z_0-9]*),", m.group(1)))
    for n in hot:
        assert n in exported or (n in ref_funcs and n not in native) or n in ref_funcs, n
    # the drivers must be the REFERENCE's (the shim no longer carries a copy of part1_array_programming.jl's driver)
    assert "function diffusion_3D_array_programming" not in SHIM and "diffusion_3D_array_programming" not in exported
    for name in ("MGOpt", "jacobi", "conjugate_gradient", "parallel", "parallel_shmem", "Data"):
        assert name in exported, name
    # fixtures: the committed copies of the reference's reftest-files
    for fx in ("test_1.bson", "T.bin", "W.bin", "S.bin"):
        assert fx in strings
        assert os.path.exists(os.path.join(ROOT, "tests", "golden", fx)) or os.path.exists(os.path.join(ROOT, "tests", "golden", "fortran", fx))
    # tolerances as the reference's tests state them (test/part1.jl:22, test/multigrid.jl:34,64, test/krylov.jl:23, test/part2.jl:31)
    assert "atol = 1e-5" in JTEST and "atol = 1e-8" in JTEST and JTEST.count("tol = 1e-6") >= 3
