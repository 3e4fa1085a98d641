"""Register budgets that multi-stream throughput depends on: the big-tile GEMM shares CUs with resident
recurrence workgroups only while 2 x GEMM + 1 x recurrence waves fit a SIMD's 512 VGPRs.  Checked on the compiler's own
metadata (hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "music-transcription_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _vgprs(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True, timeout=600)
    txt = out.read_text()
    v = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.set (\S+)\.num_vgpr, (\d+)", txt)}
    a = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.set (\S+)\.num_agpr, (\d+)", txt)}
    # gfx950 has ONE register file per SIMD: a wave's allocation is its VGPRs (rounded up to the accumulator offset's granule of 4)
    # plus its AGPRs
    return {k: (n + 3) // 4 * 4 + a.get(k, 0) for k, n in v.items()}


def _alloc(n):
    return (n + 7) // 8 * 8


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm256x_and_plain_recurrence_share_a_simd(tmp_path):
    g = {k: v for k, v in _vgprs("gemm.hip", tmp_path).items() if "gemm256x_kernel" in k}
    # the one-group-per-workgroup recurrence (what a batch of 32 launches); the variants that interleave several batch groups
    # (NG = 2..4) spend registers on requests in flight instead and are not held to this budget
    r = {k: v for k, v in _vgprs("lstm.hip", tmp_path).items() if "lstm_rec_kernelILi8ELb0ELb0ELi1E" in k}
    assert g and r
    gemm, rec = max(g.values()), max(r.values())
    assert 2 * _alloc(gemm) + _alloc(rec) <= 512, (gemm, rec)
