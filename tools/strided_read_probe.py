#!/usr/bin/env python3
"""What the batched FDE coarse pass's ACCESS PATTERN alone sustains (mv_calibrate, no LDS / barriers / MFMA): a [rows][20 480 B] matrix
read tile by tile with 512 B (the pass's K chunk) / 1 / 2 / 4 KiB per row and step, whole rows, and the contiguous nt stream.
   python tools/strided_read_probe.py [GiB=24]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd import _lib

L = _lib.lib()
L.mv_calibrate.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 24
out = {"bytes": gib << 30}
for name, what in (("contiguous_nt_16KiB_pieces", 1), ("strided_512B", 5), ("strided_1KiB", 6), ("strided_2KiB", 7), ("strided_4KiB", 8),
                   ("ldsdma_strided_128B_x8rows", 10), ("ldsdma_strided_512B", 11), ("ldsdma_strided_1KiB", 12), ("ldsdma_strided_2KiB", 13), ("ldsdma_contiguous_float_scan_transport", 3)):
    best = []
    for r in range(3):
        v = C.c_double()
        rc = L.mv_calibrate(0, what, gib << 30, 4, C.byref(v))
        assert rc == 0, L.mv_last_error()
        best.append(v.value)
    out[name] = {"GBps": round(max(best), 1), "frac_8TBps": round(max(best) / 8000, 4), "runs": [round(b, 1) for b in best]}
print(json.dumps(out))
