#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summaries (tools/rocprof_summary.py output, separate passes over
tools/variant_bench.py) into the pmc_traffic record bench.py reads: HBM bytes per page of the float scan kernel, with the
gfx950 FETCH_SIZE correction calibrated on the known-size read of the same pass, the kernel symbol and the sha256 of the
libmvmaxsim.so that was profiled (bench.py ignores the record when the loaded library differs)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(fetch_json, write_json, pages, out):
    pages = int(pages)
    f = json.load(open(fetch_json))["counters"]
    w = json.load(open(write_json))["counters"]
    scan = [k for k in f if "maxsim_ldsdma_kernel" in k and "true, false, true" not in k]  # not the stream-only calibration variant
    cal = [k for k in f if "read_bw_kernel" in k]
    assert scan and cal, (list(f)[:8])
    kern = max(scan, key=lambda k: f[k]["FETCH_SIZE"]["sum"])
    fetch_kib = f[kern]["FETCH_SIZE"]["avg"]
    write_kib = w[kern]["WRITE_SIZE"]["avg"] if kern in w else 0.0
    known = 4 << 30  # variant_bench's calibrate_read_bw(4 GiB, ...)
    ratio = f[cal[0]]["FETCH_SIZE"]["avg"] * 1024.0 / known
    corr = 1.0 / ratio  # the guide: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; measured here
    rd = fetch_kib * 1024.0 * corr
    wr = write_kib * 1024.0
    alg = pages * 1024 * 256
    h = hashlib.sha256(open(os.path.join(ROOT, "morphik-core_amd", "libmvmaxsim.so"), "rb").read()).hexdigest()
    sys.path.insert(0, ROOT)
    import bench  # the sources' hash, by bench.py's own rule (the built file embeds its build directory)

    src = bench.src_sha256()
    rec = {
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/variant_bench.py --pages {pages} --variants 6 --rounds 3, MI355X",
        "kernel": kern, "lib_sha256": h, "src_sha256": src, "pages_per_launch": pages, "algorithmic_bytes_per_launch": alg,
        "FETCH_SIZE_KiB_avg": fetch_kib, "WRITE_SIZE_KiB_avg": write_kib,
        "calibration": {"kernel": cal[0], "known_bytes": known, "FETCH_SIZE_KiB_avg": f[cal[0]]["FETCH_SIZE"]["avg"], "reported_over_known": ratio},
        "gfx950_fetch_correction": corr,
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
        "hbm_bytes_per_page": (rd + wr) / pages, "traffic_over_algorithmic": (rd + wr) / alg,
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("kernel", "hbm_bytes_per_page", "traffic_over_algorithmic", "gfx950_fetch_correction")}))


if __name__ == "__main__":
    main(*sys.argv[1:5])
