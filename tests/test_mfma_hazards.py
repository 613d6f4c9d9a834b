"""The two-stage scan kernels issue their MFMAs from inline asm, where hipcc pads no hazards.  The
bf16-row kernels run WITHOUT the s_nop pad in front of each MFMA (their operands never come from a VALU
write just before); this test compiles csrc/aura_knn.hip to assembly (no GPU needed) and checks that
property on the generated code, so a compiler or source change that breaks it fails the CPU suite."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_no_valu_write_right_before_an_mfma(tmp_path):
    import check_mfma_hazards as chk
    asm = tmp_path / "aura_knn.s"
    src = os.path.join(ROOT, "aura_snn_rag_amd", "csrc", "aura_knn.hip")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
                           "-Wno-unused-function", "--cuda-device-only", "-S", src, "-o", str(asm)],
                          stderr=subprocess.DEVNULL)
    n, bad = chk.check(str(asm))
    assert n > 1000, "scan kernels not found in the generated code"
    assert not bad, f"{len(bad)} MFMA(s) read a register a VALU instruction wrote less than two wait states before: {bad[:3]}"


def test_checker_flags_a_violation(tmp_path):
    import check_mfma_hazards as chk
    p = tmp_path / "x.s"
    p.write_text("coarse_scan_kernel_demo:\n\tv_cvt_pk_bf16_f32 v12, v8, v9\n"
                 "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[12:15], a[4:7], a[0:3]\n"
                 "\tv_add_f32 v20, v1, v2\n\ts_nop 1\n\tv_mfma_f32_16x16x32_bf16 a[0:3], v[20:23], a[4:7], a[0:3]\n")
    n, bad = chk.check(str(p))
    assert n == 2 and len(bad) == 1 and "v_cvt_pk" in bad[0][2]
