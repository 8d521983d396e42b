"""CPU test: every `ccall` of cosmo.jl_amd/julia/CosmoHIP.jl against the declaration of the same symbol in include/cosmo_hip.h.

Julia is not in this image, so the glue cannot be executed here; what CAN be checked mechanically is the failure mode the glue is most
exposed to -- silent drift between a `ccall` signature and the C prototype (an argument added, an Int32 where the header says
int64_t, a Float64 buffer where the header takes `cosmo_hip_real*`).  The test parses both files, matches symbols, and compares
argument COUNT, the return type and the CLASS of every argument (handle / pointer-to-real / pointer-to-int32 / -int64 / -uint8 /
scalar int32 / int64 / double / pointer-to-double / struct pointer / C string / callback)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cosmo_hip.h")
GLUE = os.path.join(ROOT, "cosmo.jl_amd", "julia", "CosmoHIP.jl")


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_class(decl):
    d = re.sub(r"/\*.*?\*/", "", decl).strip()
    d = re.sub(r"\[[^\]]*\]", "*", d)                       # `int64_t out[4]` is a pointer parameter
    if "(*" in d:
        return "callback"
    ptr = d.count("*")
    base = re.sub(r"\bconst\b", "", d)
    base = re.sub(r"\*.*$", "", base).strip() if ptr else " ".join(base.split()[:-1]) or base
    base = base.strip()
    if ptr == 0:
        b0 = base.split()[0]
        if b0.endswith("_fn"):
            return "callback"                               # function-pointer typedefs (cosmo_hip_project_fn, ...)
        return {"int32_t": "i32", "int64_t": "i64", "double": "f64", "int": "i32", "cosmo_hip_real": "real"}[b0]
    if ptr == 2:
        return "ptrptr"
    b = base.split()[0] if base else ""
    return {"cosmo_hip_handle": "handle", "cosmo_hip_batch": "handle", "cosmo_hip_batch_group": "handle", "cosmo_hip_real": "p_real", "int32_t": "p_i32", "int64_t": "p_i64", "double": "p_f64",
            "uint8_t": "p_u8", "char": "cstr", "void": "p_void", "cosmo_hip_params": "p_struct", "cosmo_hip_result": "p_struct",
            "cosmo_hip_accel_params": "p_struct"}.get(b, "p_other:" + b)


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*(?:COSMO_HIP_API\s+)?(int32_t|const char\s*\*|void)\s*(cosmo_hip_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [c_class(a) for a in split_top(args)]
        protos[name] = ("cstr" if "char" in ret else ("i32" if ret == "int32_t" else "void"), params)
    return protos


def jl_class(t):
    t = t.strip()
    table = {"Int32": "i32", "Cint": "i32", "Int64": "i64", "Cdouble": "f64", "Float64": "f64", "Cstring": "cstr", "Ptr{Cvoid}": "p_void", "Ref{Ptr{Cvoid}}": "ptrptr",
             "Ptr{T}": "p_real", "Ptr{Int32}": "p_i32", "Ptr{Int64}": "p_i64", "Ref{Cdouble}": "p_f64", "Ptr{Cdouble}": "p_f64", "Ptr{Float64}": "p_f64",
             "Ptr{UInt8}": "p_u8", "Ref{Params}": "p_struct", "Ref{ResultC}": "p_struct", "Ptr{ResultC}": "p_struct", "Ref{AccelParams}": "p_struct",
             "Ref{Int64}": "p_i64", "Ref{Int32}": "p_i32"}
    return table.get(t, "unknown:" + t)


def glue_ccalls():
    src = open(GLUE).read()
    src = "\n".join(ln.split("#")[0] if "ccall" not in ln.split("#")[0] and False else ln for ln in src.splitlines())
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+)\s*,", src):
        name = m.group(1)
        j = m.end(); depth = 1                                 # close the (:sym, lib) tuple
        while depth:
            depth += src[j] in "({["
            depth -= src[j] in ")}]"
            j += 1
        m2 = re.match(r"\s*,\s*(\w+(?:\{[^}]*\})?)\s*,\s*\(", src[j:])
        assert m2, "cannot parse the ccall of %s" % name
        ret = m2.group(1)
        i = j + m2.end(); depth = 1; j = i
        while depth:
            depth += src[j] in "({["
            depth -= src[j] in ")}]"
            j += 1
        types = split_top(src[i:j - 1])
        calls.append((name, ret, [t for t in types if t]))
    return calls


COMPATIBLE = {("handle", "p_void"), ("p_void", "p_void"), ("callback", "p_void"), ("ptrptr", "ptrptr")}


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    calls = glue_ccalls()
    assert len(protos) >= 60 and len(calls) >= 30
    problems = []
    for name, ret, types in calls:
        if name not in protos:
            problems.append("%s: not declared in include/cosmo_hip.h" % name); continue
        cret, cparams = protos[name]
        if jl_class(ret) != cret:
            problems.append("%s: return %s vs C %s" % (name, ret, cret))
        if len(types) != len(cparams):
            problems.append("%s: %d ccall argument types vs %d C parameters" % (name, len(types), len(cparams))); continue
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            jc = jl_class(jt)
            if jc == ct or (ct, jc) in COMPATIBLE:
                continue
            problems.append("%s: argument %d is %s in the glue (%s) but %s in the header" % (name, k + 1, jt, jc, ct))
    assert not problems, "\n".join(problems)


def test_the_hot_path_entry_points_are_all_bound():
    """The coarse loop and the two fine-grained plugin surfaces of SURVEY 8b must be reachable from the glue."""
    bound = {c[0] for c in glue_ccalls()}
    for name in ("cosmo_hip_create", "cosmo_hip_destroy", "cosmo_hip_set_problem", "cosmo_hip_set_cones", "cosmo_hip_set_params", "cosmo_hip_update_rho",
                 "cosmo_hip_kkt_solve", "cosmo_hip_project", "cosmo_hip_set_iterates", "cosmo_hip_optimize", "cosmo_hip_get_iterates", "cosmo_hip_last_error"):
        assert name in bound, name
