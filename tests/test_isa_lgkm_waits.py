"""Build-time ISA check behind the hand-scheduled LDS row loops of the register batch kernel (csrc/batch.hip: row_pipe3 / row_sliced / rowT_sliced).

Those loops issue `ds_read_*` through `asm volatile` and wait with COUNTED `s_waitcnt lgkmcnt(N)`, N > 0.  LDS reads return in order, but scalar memory loads
(`s_load_*`, `s_buffer_load_*`) share the lgkm counter and return OUT of order: if the compiler ever places an s_load between an asm ds_read and its counted wait,
the wait can be satisfied by the s_load while the LDS read is still in flight -- a stale operand -- and the compiler's own wait-count insertion does not see the
counters of inline asm.  The bit-identity tests check the ISA of TODAY's toolchain; this test checks the property itself on the code objects of both libraries:
in every device function, no `s_waitcnt lgkmcnt(N > 0)` is reached (in program order) while a scalar load issued since the last `lgkmcnt(0)` may be outstanding.
(Correct compiler-generated code satisfies the same rule, so the scan runs over all kernels, not only k_batch_admm_reg.)"""
import glob
import os
import re
import shutil
import subprocess

import pytest

import cosmo_jl_amd as cj

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SMEM = ("s_load", "s_buffer_load", "s_scratch_load", "s_memtime", "s_memrealtime", "s_atc_probe", "s_dcache")


def _scan(asm_path):
    viol, counted, fn, smem, last = [], 0, None, False, None
    with open(asm_path) as f:
        for ln in f:
            m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
            if m:
                fn, smem = m.group(1), False
                continue
            t = ln.strip().split("//")[0].strip()
            if not t:
                continue
            op = t.split()[0]
            if op.startswith(SMEM):
                smem, last = True, t
            elif op == "s_waitcnt":
                m2 = re.search(r"lgkmcnt\((\d+)\)", t)
                if m2:
                    if int(m2.group(1)) == 0:
                        smem = False
                    else:
                        counted += 1
                        if smem:
                            viol.append((fn, t, last))
    return viol, counted


@pytest.mark.parametrize("which", ["float64", "float32"])
def test_no_counted_lgkm_wait_with_a_scalar_load_in_flight(which, tmp_path):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not available")
    lib = cj._ffi.LIB_PATH if which == "float64" else cj._ffi.LIB_PATH_F32
    work = tmp_path / "co"
    work.mkdir()
    local = str(work / os.path.basename(lib))
    shutil.copy(lib, local)                                            # (--offloading writes the extracted bundles NEXT TO its input)
    subprocess.run([OBJDUMP, "--offloading", local], cwd=str(work), capture_output=True, text=True, check=True)
    cos = sorted(glob.glob(str(work / "*hipv4-amdgcn-amd-amdhsa--gfx950")))
    assert len(cos) >= 10, cos                                         # one code object per translation unit
    total_counted, reg_kernels = 0, 0
    for co in cos:
        out = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=True).stdout
        asm = co + ".s"
        with open(asm, "w") as f:
            f.write(out)
        reg_kernels += len(set(re.findall(r"<(_Z\d*k_batch_admm_reg[^>]*)>:", out)))
        viol, counted = _scan(asm)
        total_counted += counted
        assert not viol, viol[:3]
    assert total_counted > 50                                          # the scan saw counted waits at all (the hand-scheduled loops alone have dozens)
    assert reg_kernels >= 4                                            # ... and the instantiations of the register batch kernel
