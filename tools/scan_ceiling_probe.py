#!/usr/bin/env python3
"""Back-to-back launch rates (GB/s) of the transports and FDE scan forms over the SAME byte count, interleaved rounds in one
process, median of the rounds (single runs scatter +-2 % on this pool).  python tools/scan_ceiling_probe.py [GB] [rounds]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd.index import calibrate  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 25.6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
whats = ("read_ldsdma", "fde_scan_rowq", "fde_scan_regs", "read_nt")
b = int(gb * 1e9)
got = {w: [] for w in whats}
for w in whats:
    calibrate(w, b, 2)
for r in range(rounds):
    for w in whats:
        got[w].append(calibrate(w, b, 8))
out = {"GB": gb, "rounds": rounds, "rows_per_workgroup": os.environ.get("MV_FDE_SCAN_RU")}
for w in whats:
    out[w] = {"median": round(float(np.median(got[w])), 1), "min": round(float(np.min(got[w])), 1), "max": round(float(np.max(got[w])), 1)}
print(json.dumps(out))
