"""ISA regression guard for the round-3 hazard: hipcc left a uniform-branch TARGET that starts with the VALU
consumer of an MFMA result without the wait states that result needs; the single-row-tile instantiations of the batched scans
returned garbage.  The device listing of every MFMA kernel file is generated here (hipcc cross-compiles without a GPU) and
tools/mfma_hazard_scan.py walks it: along no path may a VALU instruction touch an MFMA's destination registers fewer than 7 issue
slots after the MFMA (the shortest legitimate distance in the library is 8; the broken instantiations had 1 and 2)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "morphik-core_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math",
        "-I" + os.path.join(ROOT, "include"), "-S", "--offload-device-only"]
SCAN = ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"]  # csrc/Makefile: SCAN_FLAGS
FILES = {"mv_fp8": SCAN, "mv_batch": SCAN, "mv_binary": SCAN, "mv_maxsim": SCAN, "mv_fde": [], "mv_fde_batch": []}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_valu_consumer_sits_right_behind_an_mfma_on_any_branch(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_hazard_scan as hz

    def listing(name):
        out = str(tmp_path / (name + ".s"))
        subprocess.run([HIPCC] + BASE + FILES[name] + [os.path.join(CSRC, name + ".hip"), "-o", out], check=True, capture_output=True, timeout=900)
        return out

    with ThreadPoolExecutor(max_workers=len(FILES)) as pool:
        paths = list(pool.map(listing, FILES))
    seen = set()
    for p in paths:
        kernels = hz.parse(p)
        assert kernels, p
        for fn, ins in kernels.items():
            for ws, mfma, consumer in hz.scan(ins):
                seen.add(mfma.split()[0])
                assert ws >= 7, f"{os.path.basename(p)} {fn}: `{consumer}` reads the result of `{mfma}` {ws} issue slots behind it"
    # the scan saw the kernels it is meant to see
    assert {"v_mfma_f32_16x16x32_bf16", "v_mfma_scale_f32_16x16x128_f8f6f4"} <= seen
    shutil.rmtree(tmp_path, ignore_errors=True)
