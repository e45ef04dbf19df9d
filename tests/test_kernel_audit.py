"""The attention backward issues its MFMAs through inline asm (csrc/attention_emu_bwd4.hip: the operand register FILE is chosen per
statement); hipcc pads no hazard around an asm statement, so every build is audited: no VALU write of an MFMA operand within two
instructions ahead of the MFMA, every non-MFMA reader of an accumulator at least two MFMAs behind the chain's last product, no spills,
no scratch.  hipcc cross-compiles for gfx950 without a GPU (about ten seconds)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_attention_backward_asm_mfma_hazard_audit(tmp_path):
    src = os.path.join(ROOT, "hoisdf_amd", "csrc", "attention_emu_bwd4.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-nonnull", "-c", src, "-o", str(tmp_path / "b4.o"),
           "-save-temps=obj"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert asm, os.listdir(tmp_path)
    path = str(tmp_path / asm[0])
    text = open(path).read()
    kernels = re.findall(r"\.name:\s+(_ZN6hoisdf20emu_attn_bwd4_kernel\S+)", text)
    assert len(set(kernels)) == 4, kernels                       # <DROP, CHAIN> x 2 x 2
    for m in re.finditer(r"\.name:\s+_ZN6hoisdf20emu_attn_bwd4_kernel.*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
        assert int(m.group(1)) == 0
    for m in re.finditer(r"\.name:\s+_ZN6hoisdf20emu_attn_bwd4_kernel.*?\.private_segment_fixed_size:\s+(\d+)", text, re.S):
        assert int(m.group(1)) == 0
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_mfma.py"), path, "bwd4_kernel"], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout[-3000:]
    assert a.stdout.count("MFMAs 120 findings 0") == 4, a.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_f16x2_attention_backward_asm_mfma_hazard_audit(tmp_path):
    """the same audit for csrc/attention_emu_bwd4h.hip (76 MFMAs per query tile)"""
    src = os.path.join(ROOT, "hoisdf_amd", "csrc", "attention_emu_bwd4h.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-nonnull", "-c", src, "-o", str(tmp_path / "b4h.o"),
           "-save-temps=obj"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert asm, os.listdir(tmp_path)
    path = str(tmp_path / asm[0])
    text = open(path).read()
    for m in re.finditer(r"\.name:\s+_ZN6hoisdf21emu_attn_bwd4h_kernel.*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
        assert int(m.group(1)) == 0
    for m in re.finditer(r"\.name:\s+_ZN6hoisdf21emu_attn_bwd4h_kernel.*?\.private_segment_fixed_size:\s+(\d+)", text, re.S):
        assert int(m.group(1)) == 0
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_mfma.py"), path, "bwd4h_kernel"], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout[-3000:]
    assert a.stdout.count("MFMAs 76 findings 0") == 2, a.stdout


def test_generated_schedules_are_current():
    """the committed *_phase.inc files are what their generators print"""
    # (kc2_phase.inc holds several variants of tools/gen/kc2_phase.py <variant>: not a one-to-one print)
    for gen, inc in (("attn_bwd4_phase.py", "attn_bwd4_phase.inc"), ("attn_fwd2_phase.py", "attn_fwd2_phase.inc"), ("dw2_phase.py", "dw2_phase.inc"),
                     ("h2_phase.py", "h2_phase.inc"), ("dw2h_phase.py", "dw2h_phase.inc"), ("attn_bwd4h_phase.py", "attn_bwd4h_phase.inc")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", gen)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        have = open(os.path.join(ROOT, "hoisdf_amd", "csrc", inc)).read()
        assert out.stdout.strip() == have.strip(), inc
