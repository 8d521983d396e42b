"""A C-language client SOLVES through include/cosmo_hip.h on the GPU: tests/c_client/solve_simple_qp.c is compiled with
`gcc -std=c99 -Wall -Werror -I include` against libcosmo_hip.so and (with -DCOSMO_HIP_REAL_FLOAT) libcosmo_hip_f32.so and run; it drives
create -> set_problem (1-based Int64 CSC, Julia's SparseMatrixCSC arrays) -> set_cones -> scale_ruiz -> set_params -> set_iterates -> optimize ->
get_iterates on the reference's simple QP and must reproduce the reference's own known answer (/root/reference/test/UnitTests/simple.jl:22-31,45-47:
status :Solved, x = [0.3, 0.7], obj_val = 1.88, atol 1e-3) -- the struct layouts and the argument order as a C compiler sees them, not through the
generated ctypes mirror."""
import os
import re
import shutil
import subprocess

import pytest

import cosmo_jl_amd as cj

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flavour", ["float64", "float32"])
def test_c_client_solves_the_reference_simple_qp(flavour, tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    libdir = os.path.dirname(cj._ffi.LIB_PATH)
    flags, lib = ([], "-lcosmo_hip") if flavour == "float64" else (["-DCOSMO_HIP_REAL_FLOAT"], "-lcosmo_hip_f32")
    exe = str(tmp_path / ("solve_" + flavour))
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror"] + flags + ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_client", "solve_simple_qp.c"),
                        "-o", exe, "-L", libdir, lib, "-lm", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    m = re.search(r"status=(\d+) iter=(\d+) x0=(\S+) x1=(\S+) obj=(\S+) cost=(\S+) rho_updates=(\d+) kkt=\"(.*)\"", out.stdout)
    assert m, out.stdout
    status, it, x0, x1, obj, cost = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))
    assert status == 1                                                      # COSMO_HIP_SOLVED (:45)
    assert 0 < it <= 5000 and it % 25 == 0                                  # stops at a termination check (check_termination = 25)
    assert abs(x0 - 0.3) <= 1e-3 and abs(x1 - 0.7) <= 1e-3                  # :46, atol = 1e-3
    assert abs(obj - 1.88) <= 1e-3 and abs(cost - 1.88) <= 1e-3             # :47
    assert m.group(8).startswith("cg: literal recurrence")                  # kkt_kind CG from cosmo_hip_default_params
