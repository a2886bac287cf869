#!/bin/bash
# round 5, call 3: dynamic chunk claiming in the LDS-DMA FDE scan -- parity, chunk-size sweep (back-to-back launches), request path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fde_scan_ldsdma or fde_coarse_scan_and_pipeline" > gpurun_out/r5c_fde_scan_tests.log 2>&1
echo "fde tests rc=$?" ; tail -3 gpurun_out/r5c_fde_scan_tests.log
: > gpurun_out/r5c_fde_scan_sweep.jsonl
for bpc in 2 1; do for ppw in 4 8 16 32 64; do
MV_FDE_SCAN_PPW=$ppw MV_FDE_SCAN_BLOCKS_PER_CU=$bpc timeout 120 python - >> gpurun_out/r5c_fde_scan_sweep.jsonl 2>>gpurun_out/r5c_sweep.err <<PY
import json, os, sys
sys.path.insert(0, ".")
from morphik_core_amd.index import calibrate
b = int(25.6e9)
out = {"ppw": $ppw, "blocks_per_cu": $bpc}
for what in ("fde_scan_ldsdma", "fde_scan_ldsdma_static", "fde_scan_regs", "read_ldsdma"):
    calibrate(what, b, 2)
    out[what] = round(calibrate(what, b, 8), 1)
print(json.dumps(out))
PY
done; done
cat gpurun_out/r5c_fde_scan_sweep.jsonl
timeout 300 python tools/fde_scan_probe.py 1250000 "default shape" > gpurun_out/r5c_fde_scan_probe.jsonl 2>gpurun_out/r5c_probe.err
cat gpurun_out/r5c_fde_scan_probe.jsonl
