#!/usr/bin/env python3
"""Workload for the effective-clock probe: the batched scan at B=16 and the MFMA calibration loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd.index import MvIndex, synth_rows, calibrate
ix = MvIndex(capacity_pages=100_000, stride_rows=1024)
ix.fill_synthetic(1234, 0, 100_000)
qs = [synth_rows(4321, j, 32) for j in range(16)]
for _ in range(4):
    ix.query_batch(qs, 10)
q = qs[0]
for _ in range(4):
    ix.query(q, 10)
print("mfma_cal", calibrate("mfma_bf16", 0, 3))
