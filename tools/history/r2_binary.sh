#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 600 -k "binary" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8) > $OUT/pytest_binary.log 2>&1
tail -4 $OUT/pytest_binary.log
python - <<'PY'
import sys, json, numpy as np
sys.path.insert(0, '.')
from morphik_core_amd import _lib as L
from morphik_core_amd.index import MvIndex, synth_rows
out = {}
for n in (200_000, 1_000_000):
    ix = MvIndex(capacity_pages=n, stride_rows=1024, with_float=False, with_binary=True)
    ix.fill_synthetic(1234, 0, n)
    qs = [synth_rows(4321, j, 32) for j in range(8)]
    for r in range(60):
        ix.query(qs[r % 8], 10, mode="binary")
    for rnd in range(3):
        for v in (4, 6):
            ix.set_option(L.MV_OPT_BINARY_VARIANT, v)
            ts = []
            for r in range(30):
                _s, _i, st = ix.query(qs[r % 8], 10, mode="binary", want_stats=True)
                if r >= 5: ts.append(st.score_kernel_ms)
            ms = float(np.median(ts))
            out[f"n{n}_v{v}_r{rnd}"] = round(n * 16384 / ms / 1e6, 1)
    ix.close()
print(json.dumps(out))
PY
