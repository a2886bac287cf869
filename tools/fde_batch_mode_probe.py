#!/usr/bin/env python3
"""The batched FDE coarse pass is bimodal ACROSS PROCESSES (profiles/r5/fde_batch_pass_bimodal_across_processes_r5p.json: 3 of 12 fresh
processes ~4 % faster, same code, same virtual layout).  This driver looks for the cause:

  1. placement: N fresh processes per setting of MV_BSCORE_STRIDE_PAD -- the distance between two requests' score vectors
     ([32][capacity] floats; every 64-page tile writes 256 B into each of the 32 rows, `capacity * 4` bytes apart: at 1 250 000 pages the
     rows are 5 000 000 B apart).  pad = 0 (the shipped layout), 64 (+256 B), 1056 (+4 224 B), 32800 (+128 KiB + 128 B).
  2. counters: fresh processes of the shipped layout under `rocprofv3 --pmc` (one counter set per process: address translation / the L2's
     external-agent queues), each process reporting its own times -> the counters of a fast and of a slow process side by side.

  python tools/fde_batch_mode_probe.py [pages=1250000] [procs_per_setting=6] [pmc_procs_per_set=4] [pads=0,64,1056,32800 | none] [sets=all | name,name]
One JSON document on stdout."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "fde_batch_scan_probe.py")
PMC_SETS = {  # <= 4 counters of one block per pass (the TCC / TCP blocks expose four counter registers per instance)
    "translation": "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum",
    "l2_read_side": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum",
    "l2_write_side": "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
    "l2_hit_miss": "TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum TCC_REQ_sum",
    # GRBM_GUI_ACTIVE / 8 XCDs / (kernel time) = the shader clock the launch ran at; the SQ counters are in its cycles
    "clock_and_issue": "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA",
    # instruction fetch: the code object lands at another address in every process -- do the hot loop's lines miss the instruction cache more
    # often in a slow process?  (GRBM_GUI_ACTIVE rides along: the launch's cycle count is what tells fast from slow)
    "instruction_cache": "GRBM_GUI_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ",
}


def run_probe(pages, env_extra, prefix=None, cwd=None):
    env = dict(os.environ, **env_extra)
    cmd = (prefix or []) + [sys.executable, PROBE, str(pages)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=cwd)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode or not lines:
        return {"error": (p.stderr or p.stdout)[-400:]}
    d = json.loads(lines[-1])
    return {"B16_ms": d["B16"]["coarse_ms"], "B32_ms": d["B32"]["coarse_ms"], "single_ms": d["single"]["coarse_ms"]}


def counters_of(d, match):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if match in row.get("Kernel_Name", ""):
                c = row["Counter_Name"]
                a = agg.setdefault(c, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return {c: round(v[1] / max(v[0], 1), 1) for c, v in agg.items()}


def main():
    pages = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    pmc_procs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    pads = [] if (len(sys.argv) > 4 and sys.argv[4] == "none") else [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,64,1056,32800").split(",")]
    sets = None if len(sys.argv) <= 5 or sys.argv[5] == "all" else sys.argv[5].split(",")
    out = {"pages": pages, "what": "batched FDE coarse pass (32 / 16 requests per read of the FDE slab), stats.coarse_ms medians of fresh processes",
           "placement": {}, "counters": {}}
    for pad in pads:
        runs = [run_probe(pages, {"MV_BSCORE_STRIDE_PAD": str(pad)}) for _ in range(procs)]
        out["placement"][f"score_stride_pad_{pad}_elements"] = runs
        print(f"pad {pad}: " + " ".join(f"{r.get('B16_ms')}/{r.get('B32_ms')}" for r in runs), file=sys.stderr, flush=True)
    for name, cset in PMC_SETS.items():
        if sets is not None and name not in sets:
            continue
        recs = []
        for i in range(pmc_procs):
            d = f"/tmp/fde_mode_pmc_{name}_{i}"
            shutil.rmtree(d, ignore_errors=True)
            r = run_probe(pages, {"MV_BSCORE_STRIDE_PAD": "0"}, prefix=["rocprofv3", "--pmc"] + cset.split() + ["--output-format", "csv", "-d", d, "--"], cwd="/tmp")
            r["batch16_kernel_counters_avg_per_launch"] = counters_of(d, "fde_scan_batch2_kernel<1")  # one query tile: 16 requests
            r["batch32_kernel_counters_avg_per_launch"] = counters_of(d, "fde_scan_batch2_kernel<2")  # two query tiles: 32 requests
            r["single_query_kernel_counters_avg_per_launch"] = counters_of(d, "fde_scan_rowq_kernel")
            shutil.rmtree(d, ignore_errors=True)
            recs.append(r)
            print(f"pmc {name} #{i}: {r.get('B16_ms')}/{r.get('B32_ms')}", file=sys.stderr, flush=True)
        out["counters"][name] = recs
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
