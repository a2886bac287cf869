"""Batched fp8 scan with 2 / 16 requests against the single-request scan at 0.26 - 1.25 M pages: the reproducer of the MTW == 1
wait-state hazard.  Prints one line per (size, batch shape, request)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd import _lib, synth
from morphik_core_amd.index import MvIndex, synth_rows
stride = 1024
for N in (262144, 1_000_000, 1_048_576, 1_048_577, 1_250_000):
    ix = MvIndex(capacity_pages=N, stride_rows=stride, with_float=False, with_fp8=True, with_fde=False)
    ix.fill_synthetic(synth.SEED_CORPUS, 0, N)
    pq = [synth_rows(synth.SEED_QUERIES, qi, 32) for qi in range(2)]
    pspec = synth.planted_spec(pq, N, stride)
    synth.plant_neighbours_any(ix, pspec, synth.SEED_CORPUS, stride)
    sc = [ix.score_all(q, mode="float_fp8") for q in pq]
    for label, qs in (("pq0,pq1", pq), ("pq0,pq0", [pq[0], pq[0]]), ("pq0 x16", [pq[0]] * 16)):
        res = ix.query_batch(qs, 1000, mode="float_fp8")
        for b, (s, i) in enumerate(res[:2]):
            ref = sc[0] if (label != "pq0,pq1" or b == 0) else sc[1]
            wi = np.lexsort((np.arange(ref.size), -ref.astype(np.float64)))[:1000]  # score desc, id asc
            bad = np.nonzero(np.abs(s - ref[i]) > 1e-5 * np.abs(ref[i]))[0]
            print(N, label, b, "ids_equal", i.tolist() == wi.tolist(), "top3", i[:3].tolist(), wi[:3].tolist(), "score mismatches", len(bad),
                  "max|ds|", float(np.abs(s - ref[i]).max()), "id range", int(i.min()), int(i.max()), flush=True)
    ix.close()
