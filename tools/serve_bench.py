#!/usr/bin/env python3
"""Measurement at the PLUGIN boundary: QPS and latency of `await store.query_similar(...)` under concurrent asyncio
clients, next to the device time of the same requests (the reference logs per-query totals at exactly this call:
core/vector_store/fast_multivector_store.py:513-605).

  python tools/serve_bench.py --mode fde_then_float --pages 200000 --clients 1,8,32,128 --seconds 2
  python tools/serve_bench.py --mode float --pages 1000000 --clients 1,8,32 --seconds 3
  python tools/serve_bench.py --null-index ...      # no GPU: the store's own Python cost per request (development aid)

For every client count and coalescer setting (off / adaptive) it reports requests/s, p50 / p99 latency of the coroutine and
the device time the library reported for those requests; `direct_batch` is the request rate of mv_query_topk_batch called
back to back with 32 requests per pass (what the store could reach with no Python in the way).  Prints one JSON object.
"""
import argparse
import asyncio
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class NullIndex:
    """Answers instantly with fixed pages: isolates the store's own per-request cost (no scoring happens here)."""

    delay_s = 0.0          # simulated device time of one call (sleeps with the GIL released, serialised like mv_index's query mutex)
    delay_per_query_s = 0.0

    def __init__(self, **kw):
        import threading

        self.n = 0
        self.device = 0
        self._mutex = threading.Lock()

    def _device(self, nq):
        d = self.delay_s + self.delay_per_query_s * nq
        if d > 0:
            with self._mutex:
                t_end = time.perf_counter() + d
                time.sleep(max(d - 1e-4, 0))
                while time.perf_counter() < t_end:
                    pass

    def __len__(self):
        return self.n

    def fill_synthetic(self, seed, first, n, n_rows=None, pages_per_doc=1):
        self.n += n

    def set_option(self, *a):
        pass

    def query(self, q, k, mode="float", allow=None, want_stats=False):
        from morphik_core_amd.index import QueryStats

        if not getattr(self, "_in_batch", False):
            self._device(1)
        s = np.linspace(1.0, 0.5, k, dtype=np.float32)
        i = np.arange(k, dtype=np.int64)
        return (s, i, QueryStats()) if want_stats else (s, i)

    def query_batch(self, queries, k, mode="float", allow=None, want_stats=False, allows=None, n_docs=0):
        self._device(len(queries))
        s = np.linspace(1.0, 0.5, k, dtype=np.float32)
        i = np.arange(k, dtype=np.int64)
        return [(s, i) for _q in queries]

    def close(self):
        pass


def build_store(a):
    from morphik_core_amd import synth
    from morphik_core_amd.store import MI355XFastMultiVectorStore, MI355XMultiVectorStore

    stride = ((a.patches + 15) // 16) * 16
    kw = dict(capacity_pages=a.pages, stride_rows=stride, device=0, batch_window_ms=0.0, max_batch=32)
    if a.null_index:
        kw["index_factory"] = lambda **k: NullIndex()
    if a.mode == "fde_then_float":
        st = MI355XFastMultiVectorStore(**kw)
    else:
        st = MI355XMultiVectorStore(mode=a.mode, **kw)
    assert st.initialize()
    st.collect_device_time = True
    st.adopt_synthetic_corpus(synth.SEED_CORPUS, a.pages, n_rows=a.patches, pages_per_doc=4)
    return st


async def run_clients(st, queries, n_clients, seconds, k, doc_ids):
    lat, dev = [], []
    t_end = time.perf_counter() + seconds

    async def client(ci):
        j = ci
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            hits = await st.query_similar(queries[j % len(queries)], k, doc_ids=doc_ids)
            lat.append(time.perf_counter() - t0)
            d = st.last_query_timing.get("device_ms")
            if d is not None:
                dev.append(d / max(st.last_query_timing.get("batched_queries", 1), 1))
            assert len(hits) == k
            j += n_clients

    t0 = time.perf_counter()
    await asyncio.gather(*[client(c) for c in range(n_clients)])
    wall = time.perf_counter() - t0
    la = np.array(lat) * 1e3
    return {"clients": n_clients, "requests": len(lat), "requests_per_s": round(len(lat) / wall, 1), "p50_ms": round(float(np.percentile(la, 50)), 4),
            "p99_ms": round(float(np.percentile(la, 99)), 4), "mean_ms": round(float(la.mean()), 4),
            "device_ms_per_request_p50": (round(float(np.percentile(dev, 50)), 4) if dev else None)}


def measure(a):
    """a: namespace with mode, pages, patches, clients ("1,8,.."), seconds, k, null_index.  -> result dict."""
    from morphik_core_amd import synth

    st = build_store(a)
    if a.null_index:
        queries = [np.random.default_rng(j).standard_normal((32, 128)).astype(np.float32) for j in range(16)]
    else:
        from morphik_core_amd.index import synth_rows

        queries = [synth_rows(synth.SEED_QUERIES, j, 32) for j in range(16)]
    res = {"mode": a.mode, "pages": a.pages, "patches": a.patches, "k": a.k, "null_index": a.null_index, "runs": []}
    # device-side yardsticks through the index itself
    if not a.null_index:
        ix = st._index
        for _ in range(5):
            ix.query(queries[0], a.k, mode=a.mode)
        d1 = [ix.query(queries[j % 16], a.k, mode=a.mode, want_stats=True)[2].total_device_ms for j in range(20)]
        t0 = time.perf_counter()
        for j in range(50):
            ix.query(queries[j % 16], a.k, mode=a.mode)
        res["direct_single"] = {"device_ms_p50": round(float(np.median(d1)), 4), "wall_ms_per_request": round((time.perf_counter() - t0) / 50 * 1e3, 4)}
        if a.mode in ("float", "fde_then_float"):
            nb = 32 if a.mode == "fde_then_float" else 16
            bq = [queries[j % 16] for j in range(nb)]
            for _ in range(3):
                ix.query_batch(bq, a.k, mode=a.mode)
            t0 = time.perf_counter()
            reps = 20 if a.mode == "fde_then_float" else 5
            for _ in range(reps):
                ix.query_batch(bq, a.k, mode=a.mode)
            res["direct_batch"] = {"requests_per_pass": nb, "requests_per_s": round(nb * reps / (time.perf_counter() - t0), 1)}
    for window, label in ((0.0, "coalescer_off"), (-1.0, "coalescer_adaptive")):
        if label == "coalescer_adaptive" and a.mode not in ("float", "fde_then_float"):
            continue
        st.batch_window_s = window / 1e3
        for nc in [int(x) for x in str(a.clients).split(",")]:
            st.coalesced_batches.clear()
            r = asyncio.run(run_clients(st, queries, nc, a.seconds, a.k, None))
            r["coalescer"] = label
            if st.coalesced_batches:
                r["mean_batch"] = round(float(np.mean(st.coalesced_batches)), 2)
            res["runs"].append(r)
            # plain text on stderr: a JSON-shaped progress line is what the driver mistook for the bench record in round 4
            print("[serving] %s clients=%d: %.0f req/s, p50 %.3f ms, p99 %.3f ms%s" % (label, nc, r["requests_per_s"], r["p50_ms"], r["p99_ms"],
                  (", mean batch %.1f" % r["mean_batch"]) if "mean_batch" in r else ""), file=sys.stderr, flush=True)
    st.close()
    one = [r for r in res["runs"] if r["clients"] == 1 and r["coalescer"] == "coalescer_off"]
    if one and "direct_single" in res:
        res["lone_request_overhead_over_device_ms"] = round(one[0]["p50_ms"] - res["direct_single"]["device_ms_p50"], 4)
    if "direct_batch" in res:
        best = max(r["requests_per_s"] for r in res["runs"])
        res["best_store_requests_per_s"] = best
        res["best_store_vs_direct_batch"] = round(best / res["direct_batch"]["requests_per_s"], 4)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="fde_then_float", choices=["fde_then_float", "float", "float_fp8", "binary"])
    ap.add_argument("--pages", type=int, default=200_000)
    ap.add_argument("--patches", type=int, default=1024)
    ap.add_argument("--clients", default="1,8,32,128")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--null-index", action="store_true")
    ap.add_argument("--null-delay-ms", type=float, default=0.0, help="--null-index: simulated device time per call (fixed part)")
    ap.add_argument("--null-delay-per-query-us", type=float, default=0.0, help="--null-index: simulated device time per query of a call")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    NullIndex.delay_s = a.null_delay_ms / 1e3
    NullIndex.delay_per_query_s = a.null_delay_per_query_us / 1e6
    res = measure(a)
    js = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(js)
    print(js)


if __name__ == "__main__":
    main()
