"""Round-4 probe: can one box pin the 328 GB exact tier of a 1.25 M-page shard, how long does it take to create / fill,
and what does the PCIe read of the rerank cost?  Writes gpurun_out/r4_host_tier_probe.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from morphik_core_amd import _lib as L, synth
from morphik_core_amd.index import MvIndex, synth_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
res = {"pages": n}
res["meminfo"] = {ln.split(":")[0]: ln.split()[1] for ln in open("/proc/meminfo") if ln.split(":")[0] in ("MemTotal", "MemAvailable", "HugePages_Total")}
res["cpus"] = os.cpu_count()
t0 = time.time()
ix = MvIndex(capacity_pages=n, stride_rows=1024, device=0, with_float=False, with_binary=False, with_fde=True, with_fp8=True, with_host_exact=True)
res["create_s"] = round(time.time() - t0, 1)
print("create", res["create_s"], flush=True)
t0 = time.time()
ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=1024)
res["fill_s"] = round(time.time() - t0, 1)
print("fill", res["fill_s"], flush=True)
qs = [synth_rows(synth.SEED_QUERIES, i, 32, device=0) for i in range(8)]
for nn in (75, 128, 1000):
    ix.set_option(L.MV_OPT_RERANK_N, min(nn, 1024))
    ts = []
    for r in range(6):
        s, i, st = ix.query(qs[r % 8], 10, mode="fp8_then_float", want_stats=True)
        ts.append((st.total_device_ms, st.rerank_ms, st.select_ms))
    m = np.median(np.array(ts[2:]), axis=0)
    res[f"fp8_then_float_n{nn}"] = {"device_ms": float(m[0]), "rerank_ms": float(m[1]), "select_ms": float(m[2]), "pcie_GBps": nn * 262144 / m[1] / 1e6}
print(json.dumps(res))
json.dump(res, open("gpurun_out/r4_host_tier_probe.json", "w"), indent=1)
t0 = time.time(); ix.close(); print("close", time.time() - t0)
