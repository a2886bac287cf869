#!/usr/bin/env python3
"""Which SHAPE of work lets the nt LDS-DMA ring stream fastest?  (csrc/mv_synth.hip: stream_probe_kernel.)  One process, interleaved
rounds, median GB/s per shape over the same 25.6 GB:  own = who owns a unit (0 workgroup / tiles interleaved, 1 workgroup / contiguous
quarters, 2 one wave), sched = 0 fresh workgroups, 1 persistent static, 2 persistent + claimed; ct = 4 KiB tiles per unit."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd.index import calibrate  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 25.6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
b = int(gb * 1e9)
SHAPES2 = [  # second sweep (round 5, call 6): rows of 20 KiB, row quarters, the query-slice load
    ("float-scan shape: wg/interleaved, fresh, 256 KiB", 0, 0, 64, 2, 0),
    ("wg/interleaved, fresh, 20 KiB (one FDE row)", 0, 0, 5, 2, 0),
    ("wg/interleaved, fresh, 20 KiB + 40 KiB query slice per wg", 0, 0, 5, 2, 1),
    ("wg/interleaved, fresh, 40 KiB (2 rows)", 0, 0, 10, 2, 0),
    ("wg/interleaved, fresh, 60 KiB (3 rows)", 0, 0, 15, 2, 0),
    ("wg/interleaved, fresh, 64 KiB", 0, 0, 16, 2, 0),
    ("wg/interleaved, fresh, 128 KiB", 0, 0, 32, 2, 0),
    ("wg/interleaved, fresh, 512 KiB", 0, 0, 128, 2, 0),
    ("row quarters, fresh, 1 row", 3, 0, 1, 2, 0),
    ("row quarters, fresh, 1 row + query slice", 3, 0, 1, 2, 1),
    ("row quarters, fresh, 4 rows", 3, 0, 4, 2, 0),
    ("row quarters, fresh, 16 rows + query slice", 3, 0, 16, 2, 1),
    ("row quarters, persistent claimed, 16 rows", 3, 2, 16, 2, 0),
    ("row quarters, persistent static, 64 rows", 3, 1, 64, 2, 0),
]
SHAPES3 = [  # third sweep: shapes within reach of a 16 KiB sign-bit page (4 tiles of 4 KiB)
    ("float-scan shape: wg/interleaved, fresh, 256 KiB", 0, 0, 64, 2, 0),
    ("sign-bit scan today: four wave-owned 16 KiB pages per fresh workgroup", 2, 0, 4, 2, 0),
    ("one 16 KiB page per fresh workgroup, one 4 KiB tile per wave", 0, 0, 4, 2, 0),
    ("four pages (64 KiB) per fresh workgroup, tiles interleaved", 0, 0, 16, 2, 0),
    ("four pages per fresh workgroup, contiguous quarters (= one page per wave, one unit)", 1, 0, 16, 2, 0),
    ("sixteen pages (256 KiB) per fresh workgroup, tiles interleaved", 0, 0, 64, 2, 0),
    ("wave-owned 64 KiB (four pages per wave), fresh", 2, 0, 16, 2, 0),
    ("wave-owned 16 KiB, persistent claimed", 2, 2, 4, 2, 0),
    ("e4m3 page: 128 KiB per fresh workgroup, tiles interleaved", 0, 0, 32, 2, 0),
]
SHAPES = [  # (label, own, sched, ct, blocks_per_cu)
    ("float-scan shape: wg/interleaved, fresh, 256 KiB", 0, 0, 64, 2),
    ("wg/interleaved, fresh, 20 KiB (one FDE row)", 0, 0, 5, 2),
    ("wg/interleaved, fresh, 80 KiB (4 rows)", 0, 0, 20, 2),
    ("wg/interleaved, fresh, 320 KiB (16 rows)", 0, 0, 80, 2),
    ("wg/interleaved, fresh, 1.25 MiB (64 rows)", 0, 0, 320, 2),
    ("wg/interleaved, fresh, 5 MiB (256 rows)", 0, 0, 1280, 2),
    ("wg/interleaved, persistent static, 1.25 MiB", 0, 1, 320, 2),
    ("wg/interleaved, persistent claimed, 1.25 MiB", 0, 2, 320, 2),
    ("wg/interleaved, persistent claimed, 320 KiB", 0, 2, 80, 2),
    ("wg/contiguous quarters, fresh, 1.25 MiB", 1, 0, 320, 2),
    ("wave-owned, fresh (4 units per wg), 320 KiB per wave", 2, 0, 80, 2),
    ("wave-owned, persistent claimed, 320 KiB per wave (FDE scan r5)", 2, 2, 80, 2),
    ("wave-owned, persistent claimed, 160 KiB per wave", 2, 2, 40, 2),
    ("wave-owned, persistent static, 1.25 MiB per wave", 2, 1, 320, 2),
]
if os.environ.get("MV_PROBE_SWEEP") == "2":
    SHAPES = SHAPES2
if os.environ.get("MV_PROBE_SWEEP") == "3":
    SHAPES = SHAPES3
SHAPES = [t if len(t) == 6 else t + (0,) for t in SHAPES]
got = {s[0]: [] for s in SHAPES}
ref = []
for r in range(rounds + 1):
    for label, own, sched, ct, bpc, qload in SHAPES:
        os.environ.update(MV_PROBE_OWN=str(own), MV_PROBE_SCHED=str(sched), MV_PROBE_CT=str(ct), MV_PROBE_BPC=str(bpc), MV_PROBE_QLOAD=str(qload))
        v = calibrate("stream_probe", b, 8)
        if r:
            got[label].append(v)
    v = calibrate("read_ldsdma", b, 8)
    if r:
        ref.append(v)
out = {"GB": gb, "rounds": rounds, "read_ldsdma (float scan kernel, stream only)": round(float(np.median(ref)), 1)}
for label, *_ in SHAPES:
    out[label] = {"median": round(float(np.median(got[label])), 1), "min": round(float(np.min(got[label])), 1), "max": round(float(np.max(got[label])), 1)}
print(json.dumps(out, indent=1))
