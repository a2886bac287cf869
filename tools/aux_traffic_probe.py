#!/usr/bin/env python3
"""Workload for the PMC traffic passes over the OTHER scans (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per run): a known-size read
for the gfx950 FETCH_SIZE correction, then the e4m3 scan, the sign-bit scan, the FDE coarse scan and the batched FDE coarse pass (32 requests)
over one index, three launches each.  tools/aux_traffic.py turns the two summaries into bytes per page against the algorithmic figure.
   python tools/aux_traffic_probe.py [pages=400000]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd import _lib
from morphik_core_amd.index import MvIndex, synth_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
L = _lib.lib()
v = C.c_double()
assert L.mv_calibrate_read_bw(0, 4 << 30, 3, C.byref(v)) == 0  # read_bw_kernel over a known 4 GiB: the FETCH_SIZE correction
ix = MvIndex(capacity_pages=n, stride_rows=1024, with_float=False, with_binary=True, with_fde=True, with_fp8=True)
ix.fill_synthetic(1234, 0, n)
q = synth_rows(4321, 0, 32)
qs = [synth_rows(4321, j, 32) for j in range(32)]
for _ in range(3):
    ix.query(q, 10, mode="float_fp8")
    ix.query(q, 10, mode="binary")
    ix.query(q, 10, mode="fde")
    ix.query_batch(qs, 10, mode="fde")
print("done", n)
