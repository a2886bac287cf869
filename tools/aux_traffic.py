#!/usr/bin/env python3
"""HBM bytes per page of the e4m3 / sign-bit / FDE scans and of the batched FDE pass from the two PMC summaries of
tools/aux_traffic_probe.py (FETCH_SIZE x the correction measured on the known 4 GiB read of the same pass, + WRITE_SIZE).
   python tools/aux_traffic.py fetch.json write.json pages out.json"""
import json
import sys


def main(fetch_json, write_json, pages, out):
    pages = int(pages)
    f = json.load(open(fetch_json))["counters"]
    w = json.load(open(write_json))["counters"]
    cal = [k for k in f if "read_bw_kernel" in k]
    corr = (4 << 30) / (f[cal[0]]["FETCH_SIZE"]["avg"] * 1024.0)
    want = {"maxsim_fp8_pair_kernel": ("e4m3 scan", 1024 * 128), "maxsim_fp8_kernel": ("e4m3 scan, one page per workgroup (candidate lists)", 1024 * 128), "maxsim_binary_mfma2_kernel": ("sign-bit scan", 1024 * 16),
            "fde_scan_rowq_kernel": ("FDE coarse scan (row quarters, default since round 5)", 20480), "fde_scan_kernel": ("FDE coarse scan (register form)", 20480), "fde_scan_batch2_kernel": ("batched FDE coarse pass, 32 requests", 20480)}
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/aux_traffic_probe.py, MI355X", "pages": pages,
           "gfx950_fetch_correction": corr, "kernels": {}}
    for key, (name, alg) in want.items():
        ks = [k for k in f if key in k]
        if not ks:
            continue
        k = max(ks, key=lambda x: f[x]["FETCH_SIZE"]["sum"])
        rd = f[k]["FETCH_SIZE"]["avg"] * 1024.0 * corr
        wr = w[k]["WRITE_SIZE"]["avg"] * 1024.0 if k in w else 0.0
        rec["kernels"][name] = {"kernel": k[:120], "launches": f[k]["FETCH_SIZE"]["dispatches"], "algorithmic_bytes_per_page": alg,
                                "hbm_read_bytes_per_page": round(rd / pages, 1), "hbm_write_bytes_per_page": round(wr / pages, 2),
                                "traffic_over_algorithmic": round((rd + wr) / (pages * alg), 5)}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec["kernels"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
