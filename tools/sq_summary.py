#!/usr/bin/env python3
"""Two counter-only rocprofv3 passes (tools/measure.sh: SQ1 / SQ2 sets, condensed by rocprof_summary.py) -> one record per kernel
whose name contains <match>: per-launch averages + the derived fractions the DESIGN tables quote.

  python tools/sq_summary.py set1.raw.json set2.raw.json <kernel-name-substring> out.json [duration_ms_per_launch]

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is per
SE (x 32 on this part) ; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE counts shader-clock cycles
of the launch (x 8 XCDs in the sum rocprofv3 reports)."""
import json
import sys


def main():
    a, b, match, out = sys.argv[1:5]
    dur_ms = float(sys.argv[5]) if len(sys.argv) > 5 else None
    res = {"kernels": {}}
    raw = {}
    for f in (a, b):
        for k, cs in json.load(open(f))["counters"].items():
            if match in k:
                raw.setdefault(k, {}).update({c: v["avg"] for c, v in cs.items()})
    for k, c in raw.items():
        d = {}
        wc = c.get("SQ_WAVE_CYCLES")
        gui = c.get("GRBM_GUI_ACTIVE")
        if wc:
            d["wave_cycles_waiting_any_frac"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 4)
            d["wave_cycles_waiting_for_an_instruction_to_issue_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
            d["wave_cycles_issuing_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4)
            if "SQ_WAIT_INST_LDS" in c:
                d["wave_cycles_waiting_on_lds_frac"] = round(c["SQ_WAIT_INST_LDS"] / wc, 4)
        if gui:
            cyc = gui / 8.0  # per-XCD active cycles of one launch
            d["launch_cycles"] = round(cyc)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                d["matrix_pipe_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4)
            if "SQ_ACTIVE_INST_VALU" in c:
                d["vector_pipe_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0 / cyc, 4)
            if wc:
                d["waves_resident_per_simd_avg"] = round(wc * 4 / 1024.0 / cyc, 2)
            if dur_ms:
                d["effective_clock_GHz"] = round(cyc / (dur_ms * 1e-3) / 1e9, 3)
        if "SQ_INSTS_MFMA" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["SQ_INSTS_MFMA"]:
            d["matrix_busy_cycles_per_mfma"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"], 2)
        res["kernels"][k] = dict({kk: round(v, 1) for kk, v in sorted(c.items())}, derived=d)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:5000])


if __name__ == "__main__":
    main()
