#!/usr/bin/env python3
"""round 5, call 2: where is the ceiling of a 3.7 ms scan?  Transport calibrations (back-to-back launches) at several sizes,
the FDE scan kernels back to back, and the request-path probe at larger corpora."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd.index import calibrate  # noqa: E402

out = {}
for gb in (8, 25.6, 64, 160):
    b = int(gb * 1e9)
    ent = {}
    for what in ("read_ldsdma", "read_nt", "fde_scan_regs", "fde_scan_ldsdma"):
        calibrate(what, b, 2)
        ent[what] = round(calibrate(what, b, 8), 1)
    out[f"{gb}GB"] = ent
    print(gb, ent, file=sys.stderr, flush=True)
print(json.dumps(out))
