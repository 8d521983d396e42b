"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/cosmo_hip.h declares, the ctypes
table matches the header, the library fails loudly without a GPU, and the host-side mirror of the reference interface
(assemble!, setup!/scale_ruiz!, update!) agrees with the oracle's restatement."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import cosmo_jl_amd as cj
from oracle import cosmo_oracle as O
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "cosmo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cosmo_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    names = _header_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(cj._ffi.LIB_PATH)
    for nm in names:
        assert hasattr(lib, nm), "libcosmo_hip.so does not export %s" % nm
    assert set(cj._ffi.SIGNATURES) == set(names)              # the binding covers the whole header, nothing else
    assert cj.load_library().cosmo_hip_version() == cj._ffi.ABI_VERSION == 1004


def test_the_dynamic_symbol_table_is_the_c_abi_and_nothing_else():
    """Both libraries are built with -fvisibility=hidden + a linker version script (csrc/exports.map): `nm -D --defined-only` lists exactly the
    COSMO_HIP_API functions of include/cosmo_hip.h -- no mangled internals (`_Z10cosmo_failP16cosmo_hip_handle...`), no kernel stubs or handle
    variables, no weak template instantiations that the two libraries (same names, one process) would otherwise have to keep apart with -Bsymbolic."""
    import shutil
    import subprocess
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    names = set(_header_functions())
    hdr = open(os.path.join(ROOT, "include", "cosmo_hip.h")).read()
    assert len(re.findall(r"(?m)^COSMO_HIP_API ", hdr)) == len(names)            # every prototype of the header carries the export attribute
    for path in (cj._ffi.LIB_PATH, cj._ffi.LIB_PATH_F32):
        out = subprocess.run([nm, "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        syms = [ln.split() for ln in out.splitlines() if ln.strip()]
        assert {t[-1] for t in syms} == names, (path, sorted({t[-1] for t in syms} ^ names)[:10])
        assert {t[-2] for t in syms} == {"T"}, path                                # functions only


def test_float32_library_exports_the_same_symbols_and_real_pointers_follow_the_header():
    """libcosmo_hip_f32.so (cosmo_hip_real = float, -DCOSMO_HIP_REAL_FLOAT) exports the same names; the binding's `_PR` placeholders
    sit exactly where the header says `cosmo_hip_real*`, everything that stays `double*` (times, residual scalars, coefficient
    tables) is `_PD` in both libraries."""
    names = _header_functions()
    lib = ctypes.CDLL(cj._ffi.LIB_PATH_F32)
    for nm in names:
        assert hasattr(lib, nm), "libcosmo_hip_f32.so does not export %s" % nm
    assert cj.load_library(np.float32).cosmo_hip_version() == cj._ffi.ABI_VERSION == 1004 and cj.load_library(np.float32) is not cj.load_library()
    src = open(os.path.join(ROOT, "include", "cosmo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for nm, (_, args) in cj._ffi.SIGNATURES.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % nm, src, re.S)
        assert m, nm
        params = [a.strip() for a in m.group(1).split(",")] if m.group(1).strip() not in ("", "void") else []
        assert len(params) == len(args), (nm, params, args)
        for prm, a in zip(params, args):
            is_real_ptr = bool(re.search(r"cosmo_hip_real\s*\*", prm))
            is_double_ptr = bool(re.search(r"\bdouble\s*\*", prm)) or bool(re.search(r"\bdouble\s+\w+\[", prm))
            assert (a is cj._ffi._PR) == is_real_ptr, (nm, prm)
            assert (a is cj._ffi._PD) == is_double_ptr, (nm, prm)
    f64 = cj.load_library(); f32 = cj.load_library(np.float32)
    assert f64.cosmo_hip_kkt_solve.argtypes[1] is cj._ffi._PD and f32.cosmo_hip_kkt_solve.argtypes[1] is cj._ffi._PF
    assert f32.cosmo_hip_residuals.argtypes[1] is cj._ffi._PD            # scalars stay double


def test_chordal_library_exports_every_header_symbol():
    src = open(os.path.join(ROOT, "include", "cosmo_chordal.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(cosmo_chordal_[a-z0-9_]+)\s*\(", src)))
    assert len(names) >= 10
    lib = cj._chordal.load_library()
    for nm in names:
        assert hasattr(lib, nm), "libcosmo_chordal.so does not export %s" % nm
    assert set(cj._chordal.SIGNATURES) == set(names)
    assert ctypes.sizeof(cj._chordal.Options) == 3 * 4 + 4 + 8 + 8  # int32 x 3, padding, pointer, int32 + padding


def test_struct_layouts_match_header():
    # field order/size of the two ABI structs (guards against silent drift between header and binding)
    src = open(os.path.join(ROOT, "include", "cosmo_hip.h")).read()
    body = re.search(r"typedef struct cosmo_hip_params \{(.*?)\} cosmo_hip_params;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, rest = decl.split(None, 1)
        fields += [f.strip() for f in rest.split(",")]
    assert fields == [f[0] for f in cj._ffi.Params._fields_]
    assert ctypes.sizeof(cj._ffi.Params) == 16 * 8 + 2 * 8 + 6 * 4 + 2 * 8 + 2 * 8  # ... + obj_true, obj_true_tol + adaptive_rho_fraction, setup_time (ABI 1002)
    assert ctypes.sizeof(cj._ffi.ResultStruct) == 2 * 4 + 3 * 8 + 8 * 8 + 64 * 8 + 8      # ... + safeguarding_iter


def test_generated_struct_mirrors_agree_with_the_header_and_a_compiled_probe(tmp_path):
    """Params / AccelParams / ResultC of the Julia glue and the ctypes binding are GENERATED from the header
    (tools/gen_abi_structs.py); the committed files are up to date, and header layout == ctypes layout == offsetof() of a C probe."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("gen_abi_structs", os.path.join(ROOT, "tools", "gen_abi_structs.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    text = open(G.HEADER).read()
    jl, py = G.render(text)
    assert open(G.OUT_JL).read() == jl and open(G.OUT_PY).read() == py, "stale generated mirrors: run python tools/gen_abi_structs.py"
    lay = G.layout(text)
    probe = ["#include <stdio.h>", "#include <stddef.h>", '#include "cosmo_hip.h"', "int main(void) {"]
    for cname, (rows, size) in lay.items():
        probe.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for name, _, _, _ in rows:
            probe.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, name, cname, name))
    probe += ["  return 0;", "}"]
    src = tmp_path / "probe.c"; exe = tmp_path / "probe"
    src.write_text("\n".join(probe))
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    ct = {"cosmo_hip_params": cj._ffi.Params, "cosmo_hip_accel_params": cj._ffi.AccelParams, "cosmo_hip_result": cj._ffi.ResultStruct}
    for cname, (rows, size) in lay.items():
        assert int(got[cname]) == size == ctypes.sizeof(ct[cname]), cname
        assert [r[0] for r in rows] == [f[0] for f in ct[cname]._fields_]
        for name, _, off, sz in rows:
            assert int(got["%s.%s" % (cname, name)]) == off == getattr(ct[cname], name).offset, (cname, name)
            assert getattr(ct[cname], name).size == sz
    # the Julia mirror lists the same fields in the same order with the same primitive types (Julia lays isbits structs out like C)
    for cname, jname, _ in G.STRUCTS:
        body = re.search(r"struct %s .*?\n(.*?)\nend" % jname, jl, re.S).group(1)
        jf = [ln.strip().split("::") for ln in body.splitlines()]
        assert [f[0] for f in jf] == [r[0] for r in lay[cname][0]]
        for (fname, jt), (_, ctyp, _, sz) in zip(jf, lay[cname][0]):
            base = G.CTYPE[ctyp][0]
            assert jt == base or jt == "NTuple{%d, %s}" % (sz // G.CTYPE[ctyp][2], base)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cj.CosmoHipError) as e:
        cj.Handle(0)
    assert e.value.code == 2                                  # COSMO_HIP_ERR_HIP: fail loudly, never compute on the CPU
    model = cj.Model()
    model.set(np.eye(2), np.zeros(2), np.eye(2), np.zeros(2), [cj.Nonnegatives(2)])
    with pytest.raises(cj.CosmoHipError):
        cj.optimize(model)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "cosmo.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f


def test_default_params_match_reference_settings():
    p = cj._ffi.Params()
    cj.load_library().cosmo_hip_default_params(ctypes.byref(p))
    st = O.Settings()                                         # src/settings.jl:101-139
    assert (p.sigma, p.alpha, p.rho, p.eps_abs, p.eps_rel) == (st.sigma, st.alpha, st.rho, st.eps_abs, st.eps_rel)
    assert (p.max_iter, p.check_termination, p.check_infeasibility, p.adaptive_rho_interval) == (5000, 25, 40, 40)
    assert (p.rho_min, p.rho_max, p.rho_tol, p.rho_eq_over_rho_ineq, p.adaptive_rho_tolerance) == (1e-6, 1e6, 1e-4, 1e3, 5.0)
    assert p.cosmo_infty_min_scaling == 1e20 * 1e-4 and p.tol_constant == 1.0 and p.tol_exponent == 1.5


def test_assemble_sorts_merges_and_negates():
    # src/interface.jl:30-77,411-484 ; test/UnitTests/interface.jl:53
    rng = np.random.default_rng(0)
    n = 4
    cs = [cj.Constraint(rng.standard_normal((3, n)), rng.standard_normal(3), cj.SecondOrderCone),
          cj.Constraint(rng.standard_normal((2, n)), rng.standard_normal(2), cj.Nonnegatives),
          cj.Constraint(rng.standard_normal((1, n)), rng.standard_normal(1), cj.ZeroSet),
          cj.Constraint(rng.standard_normal((2, n)), rng.standard_normal(2), cj.Box([0.0, 0], [1.0, 1])),
          cj.Constraint(rng.standard_normal((2, n)), rng.standard_normal(2), cj.Nonnegatives),
          cj.Constraint(rng.standard_normal((3, n)), rng.standard_normal(3), cj.PsdConeTriangle)]
    model = cj.Model()
    cj.assemble(model, np.eye(n), np.zeros(n), cs)
    assert [type(K).__name__ for K in model.sets] == ["ZeroSet", "Nonnegatives", "Box", "SecondOrderCone", "PsdConeTriangle"]
    assert [K.dim for K in model.sets] == [1, 4, 2, 3, 3]
    Ao, bo, cones = O.assemble([O.Constraint(c.A, c.b, util.oracle_cones([c.convex_set])[0]) for c in cs])
    assert np.array_equal(model.A.toarray(), Ao.toarray()) and np.array_equal(model.b, bo)
    assert np.array_equal(model.A.toarray()[:1], -cs[2].A.toarray())
    with pytest.raises(ValueError):
        cj.Constraint(np.eye(3), np.zeros(2), cj.Nonnegatives)
    with pytest.raises(ValueError):
        cj.Box([1.0], [0.0])
    with pytest.raises(ValueError):
        cj.PsdCone(5)
    with pytest.raises(RuntimeError):
        cj.optimize(cj.Model())


def test_host_scale_ruiz_equals_oracle():
    rng = np.random.default_rng(2)
    prob = util.random_qp(rng, 30, 3, 20, 25, soc_dims=(4, 5), psd_tri_dims=(3,))
    st = cj.Settings()
    P = prob["P"].copy(); A = prob["A"].copy(); q = prob["q"].copy(); b = prob["b"].copy()
    model = cj.Model(); model.set(P, q, A, b, prob["sets"], st)
    sm = cj.model.scale_ruiz(model.P, model.q, model.A, model.b, model.sets, st)
    ws = O.Workspace(prob["P"], prob["q"], prob["A"], prob["b"], util.oracle_cones(prob["sets"]), O.Settings())
    for a, b_ in ((sm.D, ws.sm.D), (sm.E, ws.sm.E), (model.q, ws.q), (model.b, ws.b), (model.P.toarray(), ws.P.toarray()),
                  (model.A.toarray(), ws.A.toarray())):
        assert np.array_equal(a, b_)
    assert sm.c == ws.sm.c
    box_m = [K for K in model.sets if K.kind == cj._ffi.BOX][0]; box_o = [c for c in ws.cones if c.kind == O.BOX][0]
    assert np.array_equal(box_m.l, box_o.l) and np.array_equal(box_m.u, box_o.u)


def test_generators_are_deterministic_and_shaped():
    p1 = cj.problems.sparse_box_qp(n=2000, m=4000, nnz=40000); p2 = cj.problems.sparse_box_qp(n=2000, m=4000, nnz=40000)
    assert (p1["A"] != p2["A"]).nnz == 0 and np.array_equal(p1["q"], p2["q"])
    assert p1["A"].shape == (4000, 2000) and abs(p1["A"].nnz - 40000) < 400
    K = p1["sets"][0]
    assert 0.07 < np.mean(K.l == K.u) < 0.13 and 0.03 < np.mean(K.l < -1e29) < 0.07
    p = cj.problems.socp()
    assert p["A"].shape == (1000, 500) and len(p["sets"]) == 50 and all(K.dim == 20 for K in p["sets"])
    p = cj.problems.closest_correlation(d=30)
    assert p["A"].shape == (30 + 465, 465) and [K.dim for K in p["sets"]] == [30, 465]
    x = cj.problems.svec(p["C"]); assert np.allclose(cj.problems.smat(x), p["C"])
    p = cj.problems.chordal_sdp(ncliques=12, dmin=4, dmax=9, sep_min=1, sep_max=3, n_total=400, n_zero=10, n_nonneg=20)
    assert p["A"].shape[1] == 400 and p["A"].shape[0] == sum(K.dim for K in p["sets"])
    # feasible by construction: oracle solves it
    res = O.solve(p["P"], p["q"], p["A"], p["b"], util.oracle_cones(p["sets"]), O.Settings(max_iter=3000))
    assert res.status == "Solved"


def test_constraint_constructors_mirror_the_reference():
    """test/UnitTests/constraints.jl:41-88 on the Python mirror of COSMO.Constraint (src/constraint.jl:47-108)."""
    rng = np.random.default_rng(1872381)
    # constructors: integers / scalars / row and column vectors / sparse inputs all end up as float64 CSC + vector
    for A, b, shape in ((4, 2, (1, 1)), (4.0, 2.0, (1, 1)), (np.array([[1.0, 2, 3, 4]]), 1.0, (1, 4)), (np.array([1.0, 2, 3, 4]), np.array([4.0, 3, 2, 1]), (4, 1)),
                        (sp.random(10, 2, 0.4, random_state=1), sp.csc_matrix(rng.random((10, 1))), (10, 2)), (rng.random((10, 10)), rng.random((10, 1)), (10, 10))):
        c = cj.Constraint(A, b, cj.ZeroSet)
        assert sp.issparse(c.A) and c.A.dtype == np.float64 and c.A.shape == shape and c.b.shape == (shape[0],) and c.convex_set.dim == shape[0]
    # indices: the constraint acts on a slice of a longer decision vector (:47-55; 0-based half-open range here)
    A = rng.random((3, 3)); b = rng.random(3)
    cs = cj.Constraint(A, b, cj.ZeroSet, 10, range(2, 5))
    assert cs.A.shape == (3, 10) and np.array_equal(cs.A[:, 2:5].toarray(), A) and cs.A[:, :2].nnz == 0 and cs.A[:, 5:].nnz == 0
    with pytest.raises(ValueError):
        cj.Constraint(A, b, cj.ZeroSet, 4, range(2, 5))                       # dim < stop (DomainError in the reference)
    with pytest.raises(ValueError):
        cj.Constraint(A, rng.random(4), cj.ZeroSet)                           # DimensionMismatch
    with pytest.raises(ValueError):
        cj.Constraint(A, b, cj.ZeroSet(4))                                    # DimensionMismatch with the set
    with pytest.raises(TypeError):
        cj.Constraint(np.eye(3), np.zeros(3), cj.PowerCone)                   # ArgumentCones need an object (:97)
    # PsdConeTriangle given as a type: real or complex is deduced from the dimension (:101-106)
    assert type(cj.Constraint(np.eye(6), np.zeros(6), cj.PsdConeTriangle).convex_set) is cj.PsdConeTriangle
    assert type(cj.Constraint(np.eye(9), np.zeros(9), cj.PsdConeTriangle).convex_set) is cj.ComplexPsdConeTriangle
    assert type(cj.Constraint(np.eye(1), np.zeros(1), cj.PsdConeTriangle).convex_set) is cj.PsdConeTriangle


def test_merge_constraints_mirrors_the_reference():
    # constraints.jl:57-88: ZeroSet and Nonnegatives constraints are stacked into one (merge_constraints!, src/constraint.jl:127-160)
    rng = np.random.default_rng(3)
    for K in (cj.ZeroSet, cj.Nonnegatives):
        A1, b1, A2, b2 = rng.random((10, 10)), rng.random(10), rng.random((10, 10)), rng.random(10)
        model = cj.Model()
        cj.assemble(model, np.eye(10), np.zeros(10), [cj.Constraint(A1, b1, K), cj.Constraint(A2, b2, K)])
        assert len(model.sets) == 1 and type(model.sets[0]) is K and model.sets[0].dim == 20
        assert np.array_equal(model.A.toarray(), -np.vstack([A1, A2])) and np.array_equal(model.b, np.concatenate([b1, b2]))


def test_interface_set_and_assemble_mirror_the_reference():
    """test/UnitTests/interface.jl:16-101 and model.jl:40-62 on the Python mirror of set! / assemble! / empty_model!
    (src/interface.jl:30-77, 218-250): dimension checks and the scalar / vector / matrix / sparse input combinations."""
    P = np.array([[4.0, 1], [1, 2]]); q = np.array([1.0, 1]); A = np.array([[1.0, 1], [1, 0], [0, 1]])
    l = np.array([1.0, 0, 0]); u = np.array([1.0, 0.7, 0.7])
    Aa = np.vstack([A, -A]); b = np.concatenate([u, -l]); sets = [cj.Nonnegatives(3), cj.Nonnegatives(3)]
    m = cj.Model(); m.set(P, q, Aa, b, sets)
    assert (m.n, m.m) == (2, 6)
    for bad in ((P, np.ones(3), Aa, b), (np.zeros((1, 1)), q, Aa, b), (P, q, np.array([[1.0, 2], [1, 2]]), b), (P, q, Aa, np.array([1.0, 2]))):
        with pytest.raises(ValueError):                                       # DimensionMismatch (interface.jl:33-36)
            cj.Model().set(*bad, sets)
    rng = np.random.default_rng(41)
    P = rng.random((2, 2)); q = rng.random(2); A1 = rng.random((1, 2))
    con = cj.Constraint(A1, 1.0, cj.Nonnegatives)
    m = cj.Model(); cj.assemble(m, P, q, con)
    assert np.array_equal(m.P.toarray(), P) and np.array_equal(m.q, q) and np.array_equal(m.A.toarray(), -A1) and np.array_equal(m.b, [1.0])
    m.empty(); cj.assemble(m, sp.csc_matrix(P), sp.csc_matrix(q.reshape(-1, 1)), con)   # sparse inputs (interface.jl:55-58)
    assert np.array_equal(m.P.toarray(), P) and np.array_equal(m.q, q)
    con5 = cj.Constraint(rng.random((5, 1)), rng.random(5), cj.Nonnegatives)
    for Pin, qin in ((1.0, [1.0]), ([1.0], 1.0), (1.0, 1.0), ([1.0], np.array([[1.0]])), (4, 2)):   # number / vector / matrix mixes (:64-86)
        m = cj.Model(); cj.assemble(m, Pin, qin, con5)
        assert m.P.shape == (1, 1) and m.q.shape == (1,) and m.P[0, 0] == float(np.ravel(Pin)[0]) and m.q[0] == float(np.ravel(qin)[0])
    with pytest.raises(ValueError):                                           # constraint with the wrong number of columns (:88-95)
        cj.assemble(cj.Model(), sp.identity(2, format="csc"), rng.random(2), [cj.Constraint(1.0, 0.0, cj.Nonnegatives)])
    # model.jl:40-62: scalar and 10 x 10 problems assemble
    scalar_c = cj.Constraint(1.0, 2.0, cj.ZeroSet)
    constr = cj.Constraint(np.eye(10), rng.random(10), cj.ZeroSet)
    for Pin, qin, c in ((4, 2, scalar_c), (4.0, 2.0, scalar_c), (rng.random((10, 10)), rng.random((10, 1)), constr),
                        (sp.random(10, 10, 0.4, random_state=2), rng.random((10, 1)), constr)):
        assert cj.assemble(cj.Model(), Pin, qin, [c]) is None


def test_cosmo_python_interface_helpers():
    """test/UnitTests/interface_python.jl:9-33: the SCS-style cone dictionary and the raw-CSC set! of the cosmo-python interface."""
    cone = {"f": 2, "l": 3, "q": [3, 4], "s": [3, 6, 10], "ep": 2, "ed": 1, "p": [0.3, -0.4], "b": 2}
    sets = cj.convex_sets_from_dict(cone, -np.array([0.3, 0.6]), np.array([0.2, 0.9]))
    assert isinstance(sets[2], cj.SecondOrderCone) and sets[2].dim == 3            # :12-13 (1-based 3)
    assert isinstance(sets[5], cj.PsdConeTriangle)                                  # :14
    assert type(sets[10]) is cj.PowerCone and sets[10].alpha == 0.3                 # :15-16
    assert type(sets[11]) is cj.DualPowerCone and sets[11].alpha == 0.4
    assert isinstance(sets[12], cj.Box) and len(sets) == 13                         # :17
    assert [type(K).__name__ for K in sets[:2]] == ["ZeroSet", "Nonnegatives"] and type(sets[9]) is cj.DualExponentialCone
    P = sp.csc_matrix(np.array([[4.0, 1], [1, 2]])); A1 = np.array([[1.0, 1], [1, 0], [0, 1]]); A = sp.csc_matrix(np.vstack([A1, -A1]))
    b = np.concatenate([[1, 0.7, 0.7], -np.array([1.0, 0, 0])])
    model = cj.Model()
    cj.set_csc(model, P.indices, P.indptr, P.data, np.array([1.0, 1]), A.indices, A.indptr, A.data, b, {"l": 6}, None, None, 6, 2)
    assert (model.n, model.m) == (2, 6) and type(model.sets[0]) is cj.Nonnegatives and model.sets[0].dim == 6
    assert np.array_equal(model.A.toarray(), A.toarray()) and np.array_equal(model.P.toarray(), P.toarray())


def test_headers_are_plain_c():
    """The drop-in boundary is a C ABI: both headers must compile as strict C99 (no C++ constructs, no torch / HIP types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    for name, defs in (("cosmo_hip.h", []), ("cosmo_hip.h", ["-DCOSMO_HIP_REAL_FLOAT"]), ("cosmo_chordal.h", [])):      # both element types of the ABI
        path = os.path.join(ROOT, "include", name)
        r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c"] + defs + [path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        txt = open(path).read()
        assert "torch" not in txt and "hipStream" not in txt and "#include <hip" not in txt


def test_plain_c_client_links_and_runs():
    """tests/c_client/abi_example.c compiled with gcc against both shared libraries: defaults readable from C, and without a GPU
    cosmo_hip_create fails loudly instead of falling back to a CPU path."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(cj._ffi.LIB_PATH):
        pytest.skip("gcc or the library not available")
    libdir = os.path.dirname(cj._ffi.LIB_PATH)
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "abi_example")
        r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_client", "abi_example.c"),
                            "-o", exe, "-L", libdir, "-lcosmo_hip", "-lcosmo_chordal", "-Wl,-rpath," + libdir], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert "version=1004 alpha=1.6 max_iter=5000 check_termination=25 accel_mem=15 merge=2 obj_true_is_nan=1" in out.stdout
        import torch
        if not torch.cuda.is_available():
            assert "create_rc=2" in out.stdout                                # COSMO_HIP_ERR_HIP: no device, no fallback
        # the same client against the Float32 instantiation: -DCOSMO_HIP_REAL_FLOAT before the header, libcosmo_hip_f32.so at link time
        exe32 = os.path.join(td, "abi_example_f32")
        r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-DCOSMO_HIP_REAL_FLOAT", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "c_client", "abi_example.c"), "-o", exe32, "-L", libdir, "-lcosmo_hip_f32", "-lcosmo_chordal",
                            "-Wl,-rpath," + libdir], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out32 = subprocess.run([exe32], capture_output=True, text=True, timeout=120)
        assert out32.returncode == 0 and "version=1004 alpha=1.6 max_iter=5000" in out32.stdout, out32.stderr


def test_c_client_that_solves_compiles_warning_free_against_both_libraries():
    """tests/c_client/solve_simple_qp.c (the reference's simple QP through the header; run on the GPU by tests/test_gpu_c_client.py) builds with
    gcc -std=c99 -Wall -Werror against both libraries, and without a GPU it fails at cosmo_hip_create with COSMO_HIP_ERR_HIP (exit code 10 + 2)."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(cj._ffi.LIB_PATH):
        pytest.skip("gcc or the library not available")
    libdir = os.path.dirname(cj._ffi.LIB_PATH)
    import torch
    with tempfile.TemporaryDirectory() as td:
        for flags, lib in (([], "-lcosmo_hip"), (["-DCOSMO_HIP_REAL_FLOAT"], "-lcosmo_hip_f32")):
            exe = os.path.join(td, "solve" + lib[2:])
            r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror"] + flags + ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_client", "solve_simple_qp.c"),
                                "-o", exe, "-L", libdir, lib, "-lm", "-Wl,-rpath," + libdir], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            if not torch.cuda.is_available():
                out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
                assert out.returncode == 12 and "cosmo_hip_create" in out.stderr, (out.returncode, out.stderr)


def test_scripts_compile():
    """bench.py, __graft_entry__.py and every helper under tools/ at least parse (they only run on a GPU box)."""
    import glob
    import py_compile
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + \
        sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.py")))
    assert len(files) >= 10
    for f in files:
        py_compile.compile(f, doraise=True, cfile=os.path.join(os.environ.get("TMPDIR", "/tmp"), "cosmo_pyc_" + os.path.basename(f) + "c"))


def test_committed_bench_line_follows_the_contract():
    """The newest bench line kept under profiles/ (written by `python bench.py` on the MI355X) carries every field of the bench contract,
    the roofline and cpu_baseline objects, and the extra workloads -- a guard against the JSON drifting while bench.py is edited on a
    GPU-less host."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_all_v*.json")), key=lambda f: (f.split("_bench_all_v")[0], int(f.split("_bench_all_v")[1].split(".")[0])))
    assert files, "no committed bench line"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None                 # BASELINE.md publishes no number for this metric
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f64" and "workload" in d["config"] and "model" not in d["config"]
    if files[-1].split(os.sep)[-1] >= "r04":                                 # the driver's parsed copy cut round 3's longer string in the middle of a word
        assert len(d["config"]["workload"]) < 120 and all(len(e["config"]["workload"]) < 120 for e in d.get("extra", {}).values())
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3             # value = steps / elapsed, ms_per_step = elapsed / steps
    r05 = files[-1].split(os.sep)[-1] >= "r05"
    # round 5: the headline is config 5, the workload north_star's targets are stated on; configs 2 / 3 / 4 are the extras
    assert d["config"]["workload"].startswith("cfg5" if r05 else "cfg2")
    assert set(d.get("extra", {})) == ({"cfg2", "cfg3", "cfg4"} if r05 else {"cfg3", "cfg4", "cfg5"})
    for name, e in d["extra"].items():
        assert "error" not in e and e["value"] > 0 and "roofline" in e and "cpu_baseline" in e, name
    if r05:
        assert d["scaling"] == "strong" and list(d)[-1] == "summary" and set(d["summary"]) >= {"cfg5", "cfg2", "cfg3", "cfg4"}
        assert all(d["summary"][k]["value"] == (d if k == "cfg5" else d["extra"][k])["value"] for k in ("cfg5", "cfg2", "cfg3", "cfg4"))
        # the CPU path north_star names ("the reference Julia/QDLDL CPU path"): the direct-KKT leg, next to the CG leg
        dk = cb["direct_kkt"]
        assert dk["value"] > cb["value"] and dk["nnz_L"] > 0 and dk["factor_s"] > 0 and dk["solve_ms"] > 0 and dk["cores"] == 1
        assert abs(d["config"]["gpu_over_cpu_direct_kkt"] - d["value"] / dk["value"]) < 0.05 * d["config"]["gpu_over_cpu_direct_kkt"]
        assert d["extra"]["cfg2"]["cpu_baseline"]["direct_kkt"]["feasible"] is False


def test_optimize_batch_routing_knows_what_the_batch_kernels_take():
    """Host logic of model._solve_shard_on_device: a list goes to cosmo_hip_batch directly only if it has ONE structure the persistent kernels take
    (csrc/batch.hip: CG kinds, cones of batch mode with PSD side <= 64, a fixed rho interval); everything else goes through the batch group, where
    refused classes get one handle per problem (csrc/batch_group.hip)."""
    import scipy.sparse as sp
    from cosmo_jl_amd import model as M

    def mk(sets, m, **st):
        md = cj.Model()
        md.set(sp.identity(3, format="csc"), np.zeros(3), sp.random(m, 3, density=0.5, format="csc", random_state=1), np.zeros(m), sets, cj.Settings(**st))
        return md
    assert M._batch_kernels_take(mk([cj.ZeroSet(2), cj.Nonnegatives(3), cj.SecondOrderCone(4)], 9))
    assert M._batch_kernels_take(mk([cj.PsdConeTriangle(64 * 65 // 2)], 64 * 65 // 2))
    assert M._batch_kernels_take(mk([cj.PsdCone(64 * 64)], 64 * 64))
    assert not M._batch_kernels_take(mk([cj.PsdConeTriangle(65 * 66 // 2)], 65 * 66 // 2))
    assert not M._batch_kernels_take(mk([cj.PsdCone(65 * 65)], 65 * 65))
    assert M._batch_kernels_take(mk([cj.ExponentialCone(), cj.PowerCone(0.3)], 6))
    assert not M._batch_kernels_take(mk([cj.Nonnegatives(3)], 3, kkt_solver=cj.MINRESIndirectKKTSolver))
    assert not M._batch_kernels_take(mk([cj.Nonnegatives(3)], 3, kkt_solver=cj.with_options(cj.IndirectReducedKKTSolverMINRES)))
    assert M._batch_kernels_take(mk([cj.Nonnegatives(3)], 3, kkt_solver=cj.with_options(cj.CGIndirectKKTSolver, tol_constant=2.0)))
    assert not M._batch_kernels_take(mk([cj.Nonnegatives(3)], 3, adaptive_rho_interval=0))
    assert M._batch_kernels_take(mk([cj.Nonnegatives(3)], 3, adaptive_rho_interval=0, adaptive_rho=False))
    a, b = mk([cj.Nonnegatives(3)], 3), mk([cj.Nonnegatives(4)], 4)
    assert M._structure_key(a) != M._structure_key(b) and M._structure_key(a) == M._structure_key(mk([cj.Nonnegatives(3)], 3))
