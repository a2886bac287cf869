#!/usr/bin/env python3
"""The batched FDE coarse pass changes its time in steps of 1-7 % -- between processes, between index builds of one process, and (r6 record
fde_batch_pass_mode_changes_with_reallocation_in_one_process_r6.json, `last_build_B32_ms_repeated`) over a few seconds with NOTHING re-allocated.
The shader clock is the same in both modes (cycles / ms of GRBM_GUI_ACTIVE); this probe watches the clocks the counters do not show: it runs the
pass back to back for `seconds` while a thread samples the device's sysfs power-management files (pp_dpm_sclk / mclk / fclk / socclk, the hwmon
power and temperatures) every 50 ms, then lines each measurement up with the samples taken during it.  Phase 2 asks the SMU for a fixed
performance level (`rocm-smi --setperflevel high`, restored to auto afterwards; skipped when the box refuses) and repeats.

  python tools/fde_batch_clock_probe.py [pages=1250000] [seconds=20]
One JSON document on stdout."""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sysfs_device():
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            return d
    return None


def read(p):
    try:
        with open(p) as f:
            return f.read()
    except OSError:
        return None


def active_level(txt):
    """pp_dpm_* lists the levels, the active one carries a '*':  '1: 1400Mhz *'"""
    if not txt:
        return None
    for ln in txt.splitlines():
        if "*" in ln:
            m = re.search(r"(\d+)\s*[Mm][Hh][Zz]", ln)
            return int(m.group(1)) if m else None
    return None


class Sampler(threading.Thread):
    def __init__(self, dev, period=0.05):
        super().__init__(daemon=True)
        self.dev, self.period, self.stop_flag, self.samples = dev, period, False, []
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))) if dev else []
        self.hw = hw[0] if hw else None
        self.hw_files = []
        if self.hw:
            for f in sorted(os.listdir(self.hw)):
                if re.match(r"(power\d+_(average|input)|temp\d+_input|freq\d+_input)$", f):
                    self.hw_files.append(f)

    def run(self):
        while not self.stop_flag:
            s = {"t": time.perf_counter()}
            if self.dev:
                for k in ("sclk", "mclk", "fclk", "socclk"):
                    s[k] = active_level(read(os.path.join(self.dev, f"pp_dpm_{k}")))
            for f in self.hw_files:
                v = read(os.path.join(self.hw, f))
                try:
                    s[f] = int(v)
                except (TypeError, ValueError):
                    s[f] = None
            self.samples.append(s)
            time.sleep(self.period)


def run_phase(ix, qs, seconds, dev):
    import torch

    smp = Sampler(dev)
    smp.start()
    rows = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        ts = []
        for _ in range(9):
            _r, st = ix.query_batch(qs, 10, mode="fde", want_stats=True)
            ts.append(st.coarse_ms)
        torch.cuda.synchronize()
        rows.append({"t0": t0, "t1": time.perf_counter(), "B32_ms": round(float(np.median(ts)), 4)})
    smp.stop_flag = True
    smp.join()
    keys = [k for k in (smp.samples[0].keys() if smp.samples else []) if k != "t"]
    for r in rows:
        inside = [s for s in smp.samples if r["t0"] <= s["t"] <= r["t1"]]
        for k in keys:
            vals = [s[k] for s in inside if s.get(k) is not None]
            if vals:
                r[k] = round(float(np.mean(vals)), 1)
        r["t0"] = round(r["t0"] - rows[0]["t0"], 3) if r is not rows[0] else 0.0
        del r["t1"]
    # correlation of every sampled quantity with the pass's time
    corr = {}
    y = np.array([r["B32_ms"] for r in rows])
    for k in keys:
        x = np.array([r.get(k, np.nan) for r in rows], dtype=float)
        ok = ~np.isnan(x)
        if ok.sum() > 4 and np.std(x[ok]) > 0 and np.std(y[ok]) > 0:
            corr[k] = round(float(np.corrcoef(x[ok], y[ok])[0, 1]), 3)
    return {"n_samples": len(smp.samples), "sampled": keys, "correlation_with_B32_ms": corr, "B32_ms_min_median_max": [float(y.min()), float(np.median(y)), float(y.max())], "rows": rows}


def sh(cmd):
    try:
        p = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
        return {"rc": p.returncode, "out": p.stdout[-1500:]}
    except Exception as e:  # noqa: BLE001
        return {"rc": -1, "out": repr(e)}


def main():
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    dev = sysfs_device()
    res = {"pages": n, "sysfs_device": dev, "pp_files": sorted(os.path.basename(p) for p in glob.glob(os.path.join(dev, "pp_*"))) if dev else []}
    if dev:
        res["levels"] = {k: read(os.path.join(dev, f"pp_dpm_{k}")) for k in ("sclk", "mclk", "fclk", "socclk")}
        res["power_dpm_force_performance_level"] = read(os.path.join(dev, "power_dpm_force_performance_level"))
    res["rocm_smi_clocks_idle"] = sh("rocm-smi --showclocks --showperflevel --showpower 2>&1 | grep -v '^=\\|^$' | head -30")
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True)
    ix.fill_synthetic(1234, 0, n)
    for _ in range(6):
        ix.query_batch(qs, 10, mode="fde")
    res["auto"] = run_phase(ix, qs, seconds, dev)
    print("auto:", res["auto"]["B32_ms_min_median_max"], res["auto"]["correlation_with_B32_ms"], file=sys.stderr, flush=True)
    res["setperflevel_high"] = sh("rocm-smi --setperflevel high 2>&1 | tail -5")
    if dev:
        res["level_after_set"] = read(os.path.join(dev, "power_dpm_force_performance_level"))
    try:
        res["high"] = run_phase(ix, qs, seconds, dev)
        print("high:", res["high"]["B32_ms_min_median_max"], res["high"]["correlation_with_B32_ms"], file=sys.stderr, flush=True)
    finally:
        res["setperflevel_auto"] = sh("rocm-smi --setperflevel auto 2>&1 | tail -3")
    ix.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
