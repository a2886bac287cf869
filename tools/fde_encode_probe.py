#!/usr/bin/env python3
"""Corpus build with the FDE slab under both document-encode kernels (MV_OPT_FDE_ENCODE_VARIANT 3 = bf16-pipe AMS / 1 = f32 pipe):
the workload the PMC passes of tools/r3s2_encode_pmc.sh profile.   python tools/fde_encode_probe.py [pages=20000]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib
    from morphik_core_amd.index import MvIndex

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
    out = {"pages": n}
    for variant, name in ((4, "two_pass_onehot"), (3, "bf16_pipe_ams"), (1, "f32_pipe"), (-1, "no_fde")):
        for rep in range(2):
            ix = MvIndex(capacity_pages=n, stride_rows=1024, with_float=True, with_fde=variant >= 0)
            if variant >= 0:
                ix.set_option(_lib.MV_OPT_FDE_ENCODE_VARIANT, variant)
            t0 = time.perf_counter()
            ix.fill_synthetic(1234, 0, n)
            out[name + "_fill_s"] = round(time.perf_counter() - t0, 4)
            ix.close()
    for name in ("two_pass_onehot", "bf16_pipe_ams", "f32_pipe"):
        out[name + "_us_per_page"] = round((out[name + "_fill_s"] - out["no_fde_fill_s"]) / n * 1e6, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
