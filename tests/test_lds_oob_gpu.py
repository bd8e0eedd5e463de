"""The hardware behaviour the STACK walk's depth limit leans on: an LDS access beyond a workgroup's allocation is
dropped (writes) or answered with 0 (reads), so that in a tree deeper than the caller declared the pushes below the last stack row --
the stacks are the LAST region of the workgroup's LDS -- cannot reach another ray's or another workgroup's data.  The probe is built
and run here, on whatever GPU runs the suite; tests/test_parity_gpu.py::test_tree_deeper_than_declared_is_refused covers the kernel."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not installed")
def test_lds_accesses_beyond_the_allocation_are_dropped(tmp_path):
    exe = str(tmp_path / "lds_oob_probe")
    src = os.path.join(ROOT, "tools", "experiments", "lds_oob_probe.hip")
    subprocess.run([HIPCC, "-O2", "--offload-arch=gfx950", "-Wno-unused-value", "-o", exe, src], check=True, timeout=600, capture_output=True)
    out = subprocess.run([exe], check=True, timeout=120, capture_output=True, text=True).stdout
    m = re.search(r"out-of-range writes: (\d+); out-of-range reads that returned non-zero: (\d+)", out)
    assert m, out
    changed, nonzero = int(m.group(1)), int(m.group(2))
    assert changed == 0, out
    # 4096 workgroups x 50 rounds: the only out-of-range reads that may see data are those inside the 1280-byte granule the
    # 1024-byte allocation is rounded up to (offsets 1024 .. 1279: 64 threads)
    assert nonzero <= 4096 * 50 * 64, out


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not installed")
def test_rcp_plus_one_newton_step_is_the_ieee_reciprocal(tmp_path):
    """Bit-exact parity of the pick-up's RN(1 / Dr) (svo_trace_fn.h: recip_rn) rests on gfx950's v_rcp_f32 followed by one Newton step
    being the correctly rounded reciprocal for every f32 of a clean direction's range: all 85 binades x 2^24 values, on the GPU that
    runs the suite (this was a manual tool)."""
    exe = str(tmp_path / "rcptest")
    src = os.path.join(ROOT, "tools", "rcptest_gpu.hip")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-o", exe, src], check=True, timeout=600, capture_output=True)
    res = subprocess.run([exe], timeout=300, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    m = re.search(r"= (\d+) values, (\d+) mismatches", res.stdout)
    assert m and int(m.group(1)) == 85 << 24 and int(m.group(2)) == 0, res.stdout
