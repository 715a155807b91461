"""Static audit of the device code in libmvgx_hip.so (no GPU needed): the projections of the cascade hashing stage must be separate
rounded products and sums (DESIGN.md 3.7 "Contraction": the toolchain's __fmul_rn / __fadd_rn are plain operators that hipcc's default
contraction fused into v_pk_fma_f32 in rounds 2-3). The GPU tier checks the arithmetic itself (tests/test_rounded_ops_gpu.py); this test
catches the same regression where the driver only builds."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from openmvg_amd import build

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_cascade_hash_kernel_has_no_fused_multiply_add():
    lib = build.build_hip()
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        found = False
        for co in glob.glob(so + ".*gfx950*"):
            syms = subprocess.run([OBJDUMP, "-t", co], check=True, capture_output=True, text=True).stdout
            names = [ln.split()[-1] for ln in syms.splitlines() if "cascade_hash_kernel" in ln and " F " in ln]
            for name in names:
                asm = subprocess.run([OBJDUMP, "-d", f"--disassemble-symbols={name}", co], check=True, capture_output=True, text=True).stdout
                ops = re.findall(r"\b(v_(?:pk_)?(?:fma|fmac|mul|add)_f32)\w*", asm)
                if not ops:
                    continue
                found = True
                fused = [o for o in ops if "fma" in o]
                assert not fused, f"{len(fused)} fused multiply-adds in {name}"
                assert sum(1 for o in ops if "mul" in o) >= 128   # (the 128 x (128 + 60) projections are there, as products)
        assert found, "cascade_hash_kernel not found in the gfx950 code objects"
