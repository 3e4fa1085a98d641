"""Register budgets that multi-stream throughput depends on, checked on the compiler's own metadata (hipcc
cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "music-transcription_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _alloc(n):
    return (n + 7) // 8 * 8


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_three_recurrence_workgroups_fit_a_cu(tmp_path):
    """A plain recurrence workgroup is five waves (4 compute + the gx loader).  At 128 registers a SIMD holds 4 waves, a CU 16:
    three workgroups per CU -- 43 CUs per launch at H = 512, which is what leaves the rest of the chip to the GEMMs of the other
    forwards in flight (csrc/residency.hip, DESIGN.md section 4: capped at two per CU the default schedule loses 15 %).  Every
    inference variant at H = 512 (NG = 1..4 interleaved batch groups, gx as f32 or f16) stays within 128 registers, and none of
    them spills."""
    out = tmp_path / "lstm.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(CSRC, "lstm.hip"), "-o", str(out)], check=True, capture_output=True, timeout=900)
    txt = out.read_text()
    v = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.set (\S+)\.num_vgpr, (\d+)", txt)}
    a = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.set (\S+)\.num_agpr, (\d+)", txt)}
    scratch = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.set (\S+)\.private_seg_size, (\d+)", txt)}
    rec = [k for k in v if "lstm_rec_kernelILi8ELb0ELb0E" in k]                       # NKSW = 8 (H = 512), inference, no fused projection
    assert len(rec) == 8, rec                                                         # NG = 1..4 x {f32, f16} gx
    for k in rec:
        alloc = _alloc((v[k] + 3) // 4 * 4 + a.get(k, 0))
        assert 4 * alloc <= 512, (k, v[k], a.get(k, 0))
        assert scratch.get(k, 0) == 0, (k, scratch.get(k))
