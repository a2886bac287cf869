#!/usr/bin/env python3
"""bench.py -- MaxSim pages scored / sec on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[2] -- pre-embedded pages x 1024 patches x 128-d
bf16, MaxSim-only, one 32-token query per step, exact top-10.  Total corpus = 1 M pages
(262 144 B each = 262 GB), row-sharded over the N ranks ("strong" scaling: the corpus is fixed, the
per-GPU shard shrinks); if a rank's shard does not fit its HBM the corpus is cut to what fits and
the JSON says so.  The corpus is synthetic, generated ON the device by the counter-based
generator; 10 planted neighbours per query give an exact, unique top-10.

A "step" = one query: every rank scans its resident shard with the fused HIP MaxSim kernel,
selects its local top-10 on the device, and (N>1) one RCCL all-gather of 10 (score,id) pairs per
rank merges them.  Inputs are resident in HBM when the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--pages P] [--patches 1024] [--qtokens 32]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher in the environment starts the N ranks itself (it re-executes
this file under torch.distributed.run on 127.0.0.1 and relays rank 0's JSON line).

Other workloads (`--workload`): fp8 (configs[4] shape), binary (MultiVectorStore's sign-bit max_sim), fde_fp8
(configs[3] shard shape: FDE coarse top-1000 -> exact fp8 rerank), embed (configs[1]: ColPali-v1.2 architecture embeds
1 k synthetic pages -> device ingest -> MaxSim top-10).

Rank 0's LAST stdout line is the result record (contract in the task statement) with `roofline` and `cpu_baseline`, kept under
3 KB (HEADLINE_MAX_BYTES).  Everything else -- the cut fields of those two objects and `aux_paths` (the secondary kernels and
pipelines, measured by a child process after the headline is complete) -- is one EARLIER stdout line without a `metric` key,
also written to gpurun_out/bench_aux.json.  stderr carries plain-text progress only.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A GPU fault in a side measurement must cost that measurement, not the run: the ROCm runtime otherwise writes a GPU core dump (the size of
# the resident HBM) into the working directory -- round 6 lost a whole bench record to "No space left on device" that way.
os.environ.setdefault("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1")

PAGE_ROW_BYTES = 256  # 128 x bf16
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 (same guide)
N_QUERIES = 16
N_PLANTED = 10
K = 10


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def flush_c_stdio():
    """RCCL prints its version banner through C stdio: into a pipe that is block-buffered, so the lines surface when the process exits --
    BEHIND the result line (seen in round 5: `Librccl path : ...` was the last stdout line of a 1-rank RCCL run).  Flushing the C streams
    right after the communicator is up, and again before the record is printed, keeps the headline the last line."""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def timed_runs(fn, min_runs=5, budget_s=12.0, max_runs=7):
    """Warm-up call, then >= min_runs timed calls (more while the budget lasts).  -> list of seconds."""
    fn()
    times = []
    t_end = time.time() + budget_s
    while len(times) < min_runs or (time.time() < t_end and len(times) < max_runs):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return times


CPU_SWEEP_SCALE = 1.0  # --cpu-baseline-quick shortens every timing budget of cpu_baseline (tests of the LINE, not of the host)


def cpu_baseline(sample_pages_u16, q_u16):
    """Reference CPU path timed on this box's host cores, on a bounded sample of the same workload (BASELINE.md section 3:
    >= 20 000 pages, >= 5 repeats, median).  The reference's float MaxSim (fast_multivector_store.py:553-555 ->
    score_multi_vector) in its two CPU formulations, each at the thread count that serves it best on THIS box -- a 32-column
    skinny GEMM does not scale to 256 BLAS threads, so "all cores" is not automatically the fastest the host does:
      (i)   numpy sgemm -> max -> sum, BLAS threads swept over {8, 32, 64, 128, all}
      (ii)  the same arithmetic chunk-parallel: a pool of T workers over page chunks, ONE BLAS thread each, T swept
      (iii) torch einsum over page batches of 128 (the reference's own expression), torch threads swept
    fp32 on upcast bf16 data (the reference upcasts at load, fast_multivector_store.py:736,774).  The sweep runs on a 4096-page
    slice; the winner of every family is then timed on the whole sample (median of >= 5).  The fastest is the baseline of record."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as orc  # CPU checker / baseline only

    import torch
    from threadpoolctl import threadpool_limits

    cores = os.cpu_count() or 1
    q = orc.bf16_to_f32(q_u16)
    pages = orc.bf16_to_f32(sample_pages_u16)  # upcast outside the timed region, like the reference's load step
    n = pages.shape[0]
    sweep_n = min(n, 4096)
    counts = sorted({t for t in (8, 32, 64, 128, cores) if t <= cores})

    def numpy_blas(t, m):
        def run():
            with threadpool_limits(limits=t, user_api="blas"):
                return orc.maxsim_float_np(q, pages[:m])
        return run

    def numpy_chunks(t, m):
        step = max(16, min(128, -(-m // (4 * t))))
        spans = [(s0, min(s0 + step, m)) for s0 in range(0, m, step)]
        pool = ThreadPoolExecutor(max_workers=t)

        def run():
            with threadpool_limits(limits=1, user_api="blas"):
                return np.concatenate(list(pool.map(lambda ab: orc.maxsim_float_np(q, pages[ab[0] : ab[1]], chunk=64), spans)))
        return run, pool

    def torch_einsum(t, m):
        def run():
            torch.set_num_threads(t)
            return orc.maxsim_float_torch(q, pages[:m])
        return run

    def rate(fn, m, min_runs, budget):
        if CPU_SWEEP_SCALE < 1.0:
            min_runs, budget = min(min_runs, 2), budget * CPU_SWEEP_SCALE
        times = timed_runs(fn, min_runs=min_runs, budget_s=budget, max_runs=max(min_runs, 7))
        return m / float(np.median(times)), len(times)

    want = orc.maxsim_float_np(q, pages[:256])
    sweep = {"numpy_sgemm_blas_threads": {}, "numpy_chunk_parallel_workers": {}, "torch_einsum_threads": {}}
    for t in counts:
        sweep["numpy_sgemm_blas_threads"][t] = rate(numpy_blas(t, sweep_n), sweep_n, 2, 1.5)[0]
        fn, pool = numpy_chunks(t, sweep_n)
        assert np.allclose(fn()[:256], want, rtol=1e-6)
        sweep["numpy_chunk_parallel_workers"][t] = rate(fn, sweep_n, 2, 1.5)[0]
        pool.shutdown()
        sweep["torch_einsum_threads"][t] = rate(torch_einsum(t, min(sweep_n, 1024)), min(sweep_n, 1024), 2, 1.0)[0]
    best_t = {fam: max(v, key=v.get) for fam, v in sweep.items()}
    res, used = {}, {}
    res["numpy_sgemm"], used["numpy_sgemm"] = rate(numpy_blas(best_t["numpy_sgemm_blas_threads"], n), n, 5, 8.0)
    fn, pool = numpy_chunks(best_t["numpy_chunk_parallel_workers"], n)
    res["numpy_chunk_parallel"], used["numpy_chunk_parallel"] = rate(fn, n, 5, 8.0)
    pool.shutdown()
    n_torch = min(n, 4096)
    res["torch_einsum"], used["torch_einsum"] = rate(torch_einsum(best_t["torch_einsum_threads"], n_torch), n_torch, 5, 6.0)
    torch.set_num_threads(cores)
    best = max(res, key=res.get)
    threads = {"numpy_sgemm": best_t["numpy_sgemm_blas_threads"], "numpy_chunk_parallel": best_t["numpy_chunk_parallel_workers"],
               "torch_einsum": best_t["torch_einsum_threads"]}
    page_bytes_f32 = pages.shape[1] * 128 * 4
    numa = None
    try:
        numa = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        pass
    return {
        "value": round(res[best], 1),
        "unit": "pages/s",
        "cores": int(threads[best]),
        "cores_available": cores,
        "kind": "port",
        "formulation": best,
        "achieved_GBps_fp32_pages": round(res[best] * page_bytes_f32 / 1e9, 2),
        "pages_per_s_by_formulation": {k_: round(v, 1) for k_, v in res.items()},
        "threads_by_formulation": threads,
        "thread_sweep_pages_per_s_on_4096_pages": {fam: {str(t): round(v, 1) for t, v in d.items()} for fam, d in sweep.items()},
        "numa_nodes": numa,
        "numa_note": "the sample is first-touched by one thread (one node); workers on the other socket read it across the link -- as a single-process "
                     "reference deployment would",
        "sample": ("QUICK (a tenth of the timing budgets: not a record) " if CPU_SWEEP_SCALE < 1.0 else "") + f"{n} pages x {pages.shape[1]} patches x 128-d fp32 (upcast bf16), Q={q.shape[0]}; thread counts swept on {sweep_n} pages, every family's best "
                  f"timed on the whole sample (median of {min(used.values())}+ runs; torch_einsum on {n_torch} pages); best = {best} with {threads[best]} threads",
    }


def cpu_baseline_binary(bits_sample, q_bits):
    """CPU restatement of SQL max_sim (core/vector_store/multi_vector_store.py:285-313) on the host cores, bounded
    sample: (i) 64-bit popcounts in numpy (one thread), (ii) the +-1 identity as an all-core sgemm -> max -> sum
    (0.5 Q + sum_q max_d (s_q . s_d) / 256).  Postgres itself is not available on the box."""
    from oracle import oracle as orc  # CPU checker / baseline only

    n = bits_sample.shape[0]
    res = {}

    def pm1():
        qf = np.unpackbits(q_bits, axis=1).astype(np.float32) * 2 - 1
        out = np.empty(n, np.float64)
        for s0 in range(0, n, 256):
            pf = np.unpackbits(bits_sample[s0 : s0 + 256], axis=2).astype(np.float32) * 2 - 1  # expansion is part of the work
            sim = (pf.reshape(-1, 128) @ qf.T).reshape(pf.shape[0], pf.shape[1], -1)
            out[s0 : s0 + 256] = 0.5 * qf.shape[0] + sim.max(axis=1).sum(axis=1) / 256.0
        return out

    ref = orc.maxsim_binary_popcount_np(bits_sample[:64], q_bits)
    assert np.array_equal(ref, orc.maxsim_binary_np(bits_sample[:64], q_bits)) and np.allclose(ref, pm1()[:64])
    for name, fn in (("numpy_popcount_1thread", lambda: orc.maxsim_binary_popcount_np(bits_sample, q_bits)), ("pm1_sgemm_allcores", pm1)):
        times = timed_runs(fn, min_runs=5, budget_s=8.0)
        res[name] = n / float(np.median(times))
    best = max(res, key=res.get)
    return {"value": round(res[best], 1), "unit": "pages/s", "cores": 1 if best.endswith("1thread") else (os.cpu_count() or 1), "kind": "port",
            "sample": f"{n} pages x {bits_sample.shape[1]} patches x BIT(128), Q={q_bits.shape[0]}, median of >=5 runs; "
                      + "; ".join(f"{k}={v:.0f}" for k, v in res.items()) + f" pages/s; best={best}"}


def lib_sha256():
    import morphik_core_amd as mca

    h = hashlib.sha256()
    with open(mca.library_path(), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def src_sha256():
    """sha256 over the library's SOURCES (csrc/*.hip, csrc/*.h, csrc/Makefile, include/mvmaxsim.h; names + bytes, sorted).  The built file
    embeds its build directory (__FILE__ in the error messages), so the same sources built in another checkout hash differently: a PMC
    record is also accepted for the sources it was taken on."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "morphik-core_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "morphik-core_amd", "csrc", "*.h"))
                   + [os.path.join(ROOT, "morphik-core_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "mvmaxsim.h")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _try(f):
    try:
        return f()
    except Exception:  # noqa: BLE001
        return None


def pmc_traffic(n_local, patches):
    """HBM traffic of the scan kernel from rocprofv3 PMC counters (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 fetch
    correction calibrated on a known byte count in the same pass), collected by tools/pmc_traffic.sh and committed under
    profiles/.  Only a record taken on THIS library build counts: the file names the kernel symbol and the sha256 of the
    libmvmaxsim.so it profiled; anything else -> null (a kernel change must not inherit an old measurement)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "pmc_traffic*.json"), recursive=True), key=os.path.getmtime)
    sha = lib_sha256()
    try:
        ssha = src_sha256()
    except OSError:
        ssha = None
    for f in reversed(files):
        try:
            rec = json.load(open(f))
            same_build = rec.get("lib_sha256") == sha or (ssha is not None and rec.get("src_sha256") == ssha)  # this very file, or these very sources
            if same_build and "maxsim_ldsdma_kernel" in rec.get("kernel", ""):
                return int(round(float(rec["hbm_bytes_per_page"]) * n_local * patches / 1024.0)), os.path.relpath(f, ROOT), rec.get("kernel")
        except Exception:  # noqa: BLE001
            continue
    return None, None, None


N_HARD_Q = 64      # hard-negative queries (64 near-tied pages each)
N_TOPIC_Q = 64     # clustered-corpus queries (one per topic, graded relevance)
N_RANDOM_Q = 16    # queries against the unstructured background only


def recall_sets(args, n_pages, device, headline_spec):
    """The query / page sets every lossy path is scored on (>= 64 queries per structured corpus), all written into the FIRST
    n_pages pages of the corpus (the pages the shard-shaped indexes of aux_paths hold too): planted (3x margin), hard negatives
    (64 near-tied pages per query), clustered topics (graded relevance), and queries with no structure at all.  Pages of the
    sets are disjoint from each other and from the headline's planted pages; everything is a pure function of
    (seeds, n_pages, patches)."""
    from morphik_core_amd import synth
    from morphik_core_amd.index import synth_rows

    taken = {p for (_q, _r, p, _a, _b) in headline_spec}
    pq = [synth_rows(synth.SEED_QUERIES, 3000 + j, args.qtokens, device=device) for j in range(N_QUERIES)]
    pspec = [t for t in synth.planted_spec(pq, n_pages, args.patches, n_ranks=N_PLANTED, seed=synth.SEED_PLANTED + 1) if t[2] not in taken]
    taken |= {t[2] for t in pspec}
    hq = [synth_rows(synth.SEED_QUERIES, 1000 + j, args.qtokens, device=device) for j in range(N_HARD_Q)]
    hspec = [t for t in synth.hard_spec(hq, n_pages, args.patches) if t[2] not in taken]
    taken |= {t[2] for t in hspec}
    cq, cspec = synth.clustered_spec(N_TOPIC_Q, n_pages, args.patches, q_tokens=args.qtokens, exclude=taken)
    rq = [synth_rows(synth.SEED_QUERIES, 2000 + j, args.qtokens, device=device) for j in range(N_RANDOM_Q)]
    return {"planted": {"queries": pq, "spec": pspec},
            "hard_negatives": {"queries": hq, "spec": hspec},
            "clustered_topics": {"queries": cq, "spec": cspec},
            "unstructured": {"queries": rq, "spec": []}}


def first_pages_bitmap(n_allowed, n_total):
    """Doc bitmap allowing pages [0, n_allowed) of a corpus whose doc ordinal == page (pages_per_doc 1); None = everything."""
    if n_allowed >= n_total:
        return None
    bm = np.zeros((n_total + 31) // 32, np.uint32)
    bm[: n_allowed // 32] = 0xFFFFFFFF
    if n_allowed % 32:
        bm[n_allowed // 32] = (1 << (n_allowed % 32)) - 1
    return bm


def exact_truth(ix, queries, k=K, allow=None):
    """Exact bf16 top-k of every query over the (allowed pages of the) index (the parity-checked float scan, 16 queries per slab
    pass) + the relative margin between rank k and rank k+1."""
    tops, gaps = [], []
    for g0 in range(0, len(queries), 16):
        for s, i in ix.query_batch(queries[g0 : g0 + 16], k + 1, allow=allow):
            tops.append(i[:k].tolist())
            gaps.append(float((s[k - 1] - s[k]) / abs(s[k - 1])) if len(s) > k else float("nan"))
    return tops, gaps


def timed_mode(ix, qs, mode, n_timed=9, warm_s=0.25, k=K):
    """HIP-event times of one query mode after a time-based warm-up -> (median QueryStats fields as dict)."""
    t_end = time.perf_counter() + warm_s
    i = 0
    while time.perf_counter() < t_end:
        ix.query(qs[i % len(qs)], k, mode=mode)
        i += 1
    rows = []
    for r in range(n_timed):
        _s, _i, st = ix.query(qs[r % len(qs)], k, mode=mode, want_stats=True)
        rows.append((st.score_kernel_ms, st.total_device_ms, st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms))
    m = np.median(np.array(rows), axis=0)
    return dict(zip(("score_kernel_ms", "total_device_ms", "encode_ms", "coarse_ms", "select_ms", "rerank_ms", "topk_ms"), (float(x) for x in m)))


def scan_entry(n, bytes_per_page, ms):
    return {"kernel_ms": round(ms, 4), "pages_per_s": round(n / ms * 1e3, 1), "GBps": round(n * bytes_per_page / ms / 1e6, 1),
            "frac_hbm_8TBps": round(n * bytes_per_page / ms / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_page": bytes_per_page}


def recall_of(ix, sets, truths, gaps, modes, allow_unstructured=None):
    """recall@10 vs the exact bf16 top-10 for every (set, mode); `modes` = [(name, callable(q, allow) -> ids)].
    + recall as a function of the rank-10 / rank-11 margin over the hard-negative and clustered queries together."""
    from morphik_core_amd import synth

    out = {}
    per_query = {name: ([], []) for name, _f in modes}
    for sname, st_ in sets.items():
        ent = {"queries": len(st_["queries"]), "median_rel_gap_rank10_rank11": float(np.median(gaps[sname]))}
        al = allow_unstructured if sname == "unstructured" else None
        for name, f in modes:
            hits = [synth.recall_at_k(f(q, al), t) for q, t in zip(st_["queries"], truths[sname])]
            ent[name] = round(float(np.mean(hits)), 4)
            if sname in ("hard_negatives", "clustered_topics"):
                per_query[name][0].extend(gaps[sname])
                per_query[name][1].extend(hits)
        out[sname] = ent
    out["by_margin_hard_and_clustered"] = {name: synth.margin_bins(g, h) for name, (g, h) in per_query.items()}
    return out


def container_memory_GB():
    """Memory charged to this container right now and its limit (cgroup v2 / v1), GB -- next to the pinned tier it carries."""
    out = {}
    for key, paths in (("current", ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes")),
                       ("limit", ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"))):
        for pth in paths:
            try:
                v = open(pth).read().strip()
                out[key] = round(int(v) / 1e9, 1) if v.isdigit() else v
                break
            except OSError:
                continue
    return out


def exact_shard_pages(args, stride, device):
    """Pages of the shard-shaped index that carries its exact tier outside the slabs (aux_paths.exact_shard).  configs[3] / [4]
    name 1.25 M pages per GPU = 328 GB of exact rows; the process may pin only what its memory cgroup allows
    (mv_host_pin_budget_bytes(): limit - usage - headroom; the MI355X pool's containers run with memory.max = 300 GiB on a
    3 TiB host, and pinning past it gets the container killed, not an error).
    --exact-shard-split 1 (default): MV_WITH_EXACT_SPLIT -- the exact rows of the leading pages fill the HBM the e4m3 + FDE +
    sign-bit slabs leave free, the rest is pinned; the page count is cut (if at all) so that the pinned part stays under
    args.exact_shard_split_pin_frac of the budget.  0: the whole tier pinned, args.exact_shard_pin_frac of the budget.
    -> (pages, pin budget, split)"""
    import torch

    from morphik_core_amd import _lib

    if args.exact_shard_pages <= 0:
        return 0, 0, False
    budget = int(_lib.lib().mv_host_pin_budget_bytes())
    page_b = stride * 256
    if not args.exact_shard_split:
        fit = int(args.exact_shard_pin_frac * budget // page_b)
        return max(min(args.exact_shard_pages, fit), 0), budget, False
    _free, total_b = torch.cuda.mem_get_info(device)
    hbm = total_b - (3 << 30)  # what is free once the headline's bf16 slab is gone (context, torch's own blocks)
    slab_b = stride * 128 + stride * 16 + 20480 + 10240 + 5120 + 80 + 33 * 4  # e4m3 + sign bits + FDE (+ its e4m3 and fp4 copies) + metadata + the batched score vectors
    n = int(min(args.exact_shard_pages, (hbm - (14 << 30)) // slab_b))
    while n > 0:
        in_hbm = max(0, (hbm - n * slab_b - (13 << 30)) // page_b)  # the library keeps 12 GiB back (MV_EXACT_HBM_RESERVE_BYTES)
        if (n - min(n, in_hbm)) * page_b <= args.exact_shard_split_pin_frac * budget:
            break
        n -= 10_000
    return max(n, 0), budget, True


def full_shard(args, device, qs):
    """BASELINE configs[3] / [4] at their per-GPU shard SIZE: ONE index of args.full_shard_pages pages (10 M / 8 GPUs = 1.25 M)
    holding the e4m3, FDE and sign-bit slabs (no bf16 slab: 328 GB would not fit) -- the rates of the three full-corpus scans, of
    the batched e4m3 scan and of the batched FDE pipeline, which scale with the page count.  (The exact tier of such a shard is
    328 GB of pinned host RAM, more than this container may pin: the exact pipelines and every recall figure are measured on
    aux_paths.exact_shard, the largest shard whose exact tier does fit.)"""
    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex

    import torch

    stride = ((args.patches + 15) // 16) * 16
    per_page = stride * 128 + stride * 16 + 20480 + 10240 + 24 + 32 * 4  # slabs (+ the FDE slab's e4m3 copy) + metadata + the batched score vectors
    free_b, _tot = torch.cuda.mem_get_info(device)
    n = int(min(args.full_shard_pages, (free_b - (8 << 30)) // per_page))
    res = {"pages": n, "slabs": "e4m3 + FDE(10240 bf16) + sign bits, no bf16 slab, no exact tier", "resident_GB": round(n * per_page / 1e9, 1),
           "exact_tier_needed_GB": round(n * stride * 256 / 1e9, 1),
           "note": "kernel-only HIP-event times (median of 9 after 0.25 s of warm-up queries); unstructured corpus (timing only: recall and the exact "
                   "pipelines are on aux_paths.exact_shard)"}
    t0 = time.time()
    ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, with_float=False, with_binary=True, with_fde=True, with_fp8=True, with_fde_e4m3=True, with_fde_fp4=True)
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 0)  # the bf16 FDE slab first (the figures of rounds 2-5); its e4m3 copy below
    res["create_s"] = round(time.time() - t0, 1)
    t0 = time.time()
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=args.patches)
    res["fill_s"] = round(time.time() - t0, 1)
    log(f"[full shard] {n} pages generated in {res['fill_s']} s")
    # ---- the three full-corpus scans
    for key, mode, bpp in (("fp8_scan", "float_fp8", args.patches * 128), ("sign_bit_scan", "binary", args.patches * 16)):
        t = timed_mode(ix, qs, mode)
        res[key] = scan_entry(n, bpp, t["score_kernel_ms"])
    t = timed_mode(ix, qs, "fde")
    res["fde_coarse_scan"] = dict(scan_entry(n, 20480, t["coarse_ms"]), query_encode_ms=round(t["encode_ms"], 4))
    # ---- B requests per pass over the e4m3 slab (maxsim_batch_fp8_kernel): two-term queries (the scan's scores) and the
    # single-term coarse form (MV_OPT_BATCH_VARIANT 7: half the matrix work)
    res["batched_fp8_scan"] = {}
    for bv, B, name in ((0, 4, "B4_two_term"), (0, 16, "B16_two_term"), (7, 16, "B16_single_term")):
        ix.set_option(L.MV_OPT_BATCH_VARIANT, bv)
        ms = []
        for r in range(5):
            _res_b, st = ix.query_batch(qs[:B], K, mode="float_fp8", want_stats=True)
            if r >= 2:
                ms.append(st.score_kernel_ms)
        m = float(np.median(ms))
        useful = 2.0 * B * args.qtokens * args.patches * 128 * n / m / 1e9
        res["batched_fp8_scan"][name] = {"kernel_ms": round(m, 4), "query_pages_per_s": round(B * n / m * 1e3, 1), "useful_TFLOPs": round(useful, 1),
                                         "issued_fp8_TFLOPs": round(useful * (1 if bv == 7 else 2), 1),
                                         "frac_fp8_mfma_5000TF_issued": round(useful * (1 if bv == 7 else 2) / 5000.0, 4),
                                         "GBps": round(n * args.patches * 128 / m / 1e6, 1), "frac_hbm_8TBps": round(n * args.patches * 128 / m / 1e6 / HBM_PEAK_GBPS, 4)}
    ix.set_option(L.MV_OPT_BATCH_VARIANT, -1)
    # ---- FDE -> top-n -> rerank on the e4m3 slab (what an index WITHOUT an exact tier does), one request and 32 per slab pass
    res["fde_then_fp8_rerank"] = {cn_key: ent for cn_key, ent in fde_pipeline_timings(ix, qs, n, (75, 1000)).items()}
    # ---- the batched pass's time follows the FDE slab's allocation: up to three other allocations tried, the fastest kept
    t0 = time.time()
    before, after, moves = ix.fde_placement_trial(3)
    trial = {"pass_ms_32_requests_before": round(before, 4), "after": round(after, 4), "moves": moves, "wall_s": round(time.time() - t0, 2)}
    if moves:
        b32 = fde_pipeline_timings(ix, qs, n, (75,))["coarse75"]["batch_of_32"]
        trial["coarse75_batch_of_32_after"] = {k: b32[k] for k in ("device_ms_per_batch", "requests_per_s", "coarse_pass_frac_hbm_8TBps", "same_ids_as_single_query")}
    res["fde_placement_trial"] = trial
    # ---- round 6: the coarse stage on the e4m3 COPY of the FDE slab (MV_WITH_FDE_E4M3: 10 240 B per page instead of 20 480)
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 1)
    ix.query_batch([qs[i % len(qs)] for i in range(32)], K, mode="fde")  # (allocates the batch workspace's e4m3 part: the trial times the pass the index would run)
    t0 = time.time()
    before, after, moves = ix.fde_placement_trial(3)  # the pass's time follows the allocation of THIS slab too
    res["fde_placement_trial_e4m3"] = {"pass_ms_32_requests_before": round(before, 4), "after": round(after, 4), "moves": moves, "wall_s": round(time.time() - t0, 2)}
    t = timed_mode(ix, qs, "fde")
    res["fde_coarse_scan_e4m3"] = dict(scan_entry(n, 10240, t["coarse_ms"]), query_encode_ms=round(t["encode_ms"], 4))
    e8 = fde_pipeline_timings(ix, qs, n, (75, 1000), bytes_per_page=10240)
    res["fde_e4m3_then_fp8_rerank"] = e8
    # ---- closing session of round 6: the coarse stage of a SINGLE request on the FP4 copy (MV_WITH_FDE_FP4: 5 120 B per page)
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 2)
    t = timed_mode(ix, qs, "fde")
    res["fde_coarse_scan_fp4"] = dict(scan_entry(n, 5120, t["coarse_ms"]), query_encode_ms=round(t["encode_ms"], 4), request_device_ms=round(t["total_device_ms"], 4))
    res["fde_fp4_then_fp8_rerank"] = fde_pipeline_timings(ix, qs, n, (75,), bytes_per_page=5120)  # 32 requests per pass: both MFMA operands FP4, 12 chunks read for 10
    ix.close()
    return res


def fde_pipeline_timings(ix, qs, n, coarse_ns, bytes_per_page=20480):
    """Device times of MV_MODE_FDE_THEN_FLOAT on `ix` for every coarse list length: one request (stage split) and 32 requests per
    pass over the FDE slab (mv_query_topk_batch)."""
    from morphik_core_amd import _lib as L

    out = {}
    bq = [qs[i % len(qs)] for i in range(32)]
    for cn in coarse_ns:
        ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
        t = timed_mode(ix, qs, "fde_then_float", n_timed=12)
        dev, stg, out_b = [], [], None
        for r in range(8):
            out_b, st = ix.query_batch(bq, K, mode="fde_then_float", want_stats=True)
            if r >= 3:
                dev.append(st.total_device_ms)
                stg.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms))
        d, sb = float(np.median(dev)), np.median(np.array(stg), axis=0)
        same = float(np.mean([out_b[i][1].tolist() == ix.query(bq[i], K, mode="fde_then_float")[1].tolist() for i in range(0, 32, 4)]))
        out[f"coarse{cn}"] = {
            "one_request": {"device_ms": round(t["total_device_ms"], 4), "pages_searched_per_s": round(n / t["total_device_ms"] * 1e3, 1),
                            "stage_ms": {k: round(t[k], 4) for k in ("encode_ms", "coarse_ms", "select_ms", "rerank_ms", "topk_ms")},
                            "coarse_scan_GBps": round(n * bytes_per_page / t["coarse_ms"] / 1e6, 1),
                            "coarse_scan_frac_hbm_8TBps": round(n * bytes_per_page / t["coarse_ms"] / 1e6 / HBM_PEAK_GBPS, 4)},
            "batch_of_32": {"device_ms_per_batch": round(d, 4), "device_us_per_request": round(d * 1e3 / 32, 2), "requests_per_s": round(32 / d * 1e3, 1),
                            "throughput_vs_one_request_per_pass": round(t["total_device_ms"] * 32 / d, 2),
                            "stage_ms": {k: round(float(v), 4) for k, v in zip(("encode_32_queries", "coarse_gemm_one_slab_pass", "select", "rerank", "topk"), sb)},
                            "coarse_pass_GBps": round(n * bytes_per_page / float(sb[1]) / 1e6, 1),
                            "coarse_pass_frac_hbm_8TBps": round(n * bytes_per_page / float(sb[1]) / 1e6 / HBM_PEAK_GBPS, 4),
                            "same_ids_as_single_query": same}}
    return out


def exact_shard(args, device, n, n_truth, budget, split, sets, truths, gaps):
    """BASELINE configs[3] / [4] at their per-GPU shard SHAPE with the EXACT rerank the reference does (fp32 MaxSim on fp32 pages,
    fast_multivector_store.py:553-556): e4m3 + FDE + sign-bit slabs in HBM (no bf16 slab), the exact bf16 rows of every page in
    PINNED HOST memory (n x 256 KiB) -- split: of the leading pages in the HBM the slabs leave free (MV_WITH_EXACT_SPLIT), which
    is what lets the FULL 1.25 M-page shard run in a 300 GiB container (exact_shard_pages()).  Its first n_truth pages are
    those of the headline corpus -- the pages the recall sets were written into and the exact bf16 truth was taken on (with
    a doc filter over those pages) before the bf16 slab was freed; the recall queries carry the same doc filter when the
    shard is larger than that corpus.
    Measured: the FDE pipeline with its exact rerank (coarse top-75: straight out of host RAM; coarse top-1000: e4m3 pruning to
    128, then host RAM) and fp8_then_float -- device ms for one request and for a batch, the PCIe rate of the exact stage, recall@10
    of every lossy and every exact path against the bf16 truth on the four corpora, and the largest relative error of the
    returned scores against the FLOAT oracle on the rows the tier holds."""
    from oracle import oracle as orc  # checker only (outside every timed region)

    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex

    stride = ((args.patches + 15) // 16) * 16
    res = {"pages": n, "pages_of_a_full_shard": args.exact_shard_pages,
           "slabs": "e4m3 + FDE(10240 bf16) + sign bits in HBM; exact bf16 rows " + ("split: leading pages in the free HBM, the rest in pinned host RAM" if split else "in pinned host RAM"),
           "exact_tier_GB": round(n * stride * 256 / 1e9, 1), "pin_budget_GB": round(budget / 1e9, 1),
           "pin_budget_note": "mv_host_pin_budget_bytes(): memory cgroup limit - usage - headroom (the pool's containers: memory.max 300 GiB on a 3 TiB host; "
                              "a 1.25 M-page shard's 328 GB tier cannot be pinned whole here -- two boxes were lost finding that out)",
           "note": "kernel-only HIP-event times (median after 0.25 s of warm-up queries); recall@10 against the exact bf16 top-10 of the same pages computed by the "
                   "float scan (doc filter over the first n pages) before the bf16 slab was freed"}
    t0 = time.time()
    ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, with_float=False, with_binary=True, with_fde=True, with_fp8=True, with_host_exact=True,
                 with_exact_split=split, with_fde_e4m3=True, with_fde_fp4=True)
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 0)  # every figure below reads the bf16 FDE slab unless its name says e4m3
    res["create_and_pin_s"] = round(time.time() - t0, 1)
    in_hbm = ix.exact_hbm_pages
    res["exact_tier_pages_in_hbm"] = in_hbm
    res["exact_tier_in_hbm_GB"] = round(in_hbm * stride * 256 / 1e9, 1)
    res["pinned_host_exact_tier_GB"] = round((n - in_hbm) * stride * 256 / 1e9, 1)
    res["pinned_share_of_budget"] = round((n - in_hbm) * stride * 256 / max(budget, 1), 3)
    allow_truth = first_pages_bitmap(n_truth, n)  # None when the shard is no larger than the corpus the truth was taken on
    res["recall_doc_filter"] = None if allow_truth is None else f"first {n_truth} pages (the corpus of the exact bf16 truth)"
    t0 = time.time()
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=args.patches)
    res["fill_s"] = round(time.time() - t0, 1)
    res["container_memory_GB_with_the_tier_resident"] = container_memory_GB()
    t0 = time.time()
    planted_pages = 0
    for name, st_ in sets.items():  # "_headline_spec": the headline's own planted pages that fall inside these n pages
        planted_pages += synth.plant_neighbours_any(ix, st_ if name.startswith("_") else st_["spec"], synth.SEED_CORPUS, args.patches, 0, n)
    res["planted_pages"] = planted_pages
    res["plant_s"] = round(time.time() - t0, 1)
    log(f"[exact shard] {n} pages ({res['pinned_host_exact_tier_GB']} GB pinned in {res['create_and_pin_s']} s) generated in {res['fill_s']} s, "
        f"{planted_pages} structured pages written in {res['plant_s']} s")
    rsets = {k_: v for k_, v in sets.items() if not k_.startswith("_")}
    qs = rsets["planted"]["queries"]
    # ---- configs[3]: FDE coarse -> exact rerank out of the pinned-host tier
    base_fde = timed_mode(ix, qs, "fde")
    res["fde_then_exact_rerank"] = fde_pipeline_timings(ix, qs, n, (75, 1000))
    for cn in (75, 1000):
        ent = res["fde_then_exact_rerank"][f"coarse{cn}"]
        n_mid, tier = ix.rerank_plan(cn, K, args.qtokens)
        host_share = (n - in_hbm) / n  # candidates are spread over the shard: this share of the exact reads crosses PCIe
        read = (n_mid or cn) * args.patches * 256 * host_share
        ent["rerank_plan"] = {"tier": tier, "e4m3_pruning_to": n_mid, "exact_pages_read": (n_mid or cn), "expected_share_over_pcie": round(host_share, 3)}
        one = ent["one_request"]
        one["added_ms_over_coarse_scan_and_encode"] = round(one["device_ms"] - base_fde["total_device_ms"], 4)
        one["exact_stage_GBps_over_pcie_upper_bound"] = round(read / max(one["stage_ms"]["rerank_ms"], 1e-6) / 1e6, 1)
    # ---- the same pipeline with the coarse stage on the e4m3 copy of the FDE slab (round 6)
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 1)
    res["fde_e4m3_coarse_then_exact_rerank"] = fde_pipeline_timings(ix, qs, n, (75,), bytes_per_page=10240)
    # ---- ... and on the fp4 copy (single requests: the conversion scan of mv_fde4.hip; batches: the FP4 x FP4 MFMA form of the batched pass)
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 2)
    ix.set_option(L.MV_OPT_FDE_COARSE_N, 75)
    res["fde_fp4_coarse_then_exact_rerank"] = fde_pipeline_timings(ix, qs, n, (75,), bytes_per_page=5120)
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 0)
    # ---- configs[4]: e4m3 scan of every page -> top-128 -> exact re-score out of the pinned-host tier
    base = timed_mode(ix, qs, "float_fp8")
    t = timed_mode(ix, qs, "fp8_then_float")
    res["fp8_then_float_n128"] = {
        "rerank_n": 128, "device_ms": round(t["total_device_ms"], 4), "fp8_scan_alone_device_ms": round(base["total_device_ms"], 4),
        "added_ms_over_fp8_scan": round(t["total_device_ms"] - base["total_device_ms"], 4), "rerank_ms": round(t["rerank_ms"], 4),
        "rerank_GBps_over_pcie": round(128 * args.patches * 256 * (n - in_hbm) / n / max(t["rerank_ms"], 1e-6) / 1e6, 1),
        "pages_searched_per_s": round(n / t["total_device_ms"] * 1e3, 1)}
    dev = []
    for r in range(5):
        _o, st = ix.query_batch(qs[:16], K, mode="fp8_then_float", want_stats=True)
        if r >= 2:
            dev.append(st.total_device_ms)
    d = float(np.median(dev))
    res["fp8_then_float_n128"]["batch_of_16"] = {"device_ms_per_batch": round(d, 4), "device_us_per_request": round(d * 1e3 / 16, 1),
                                                  "query_pages_per_s": round(16 * n / d * 1e3, 1)}
    # ---- returned scores vs the FLOAT oracle on the tier's own rows (the bar: 1e-3 relative; the fp8 rerank misses it by 3-10x)
    worst = {}
    for name, mode, cn in (("fde_top75_then_exact", "fde_then_float", 75), ("fde_top1000_then_exact", "fde_then_float", 1000), ("fp8_then_float_n128", "fp8_then_float", None)):
        if cn:
            ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
        w = 0.0
        for q in qs[:4]:
            s, i = ix.query(q, K, mode=mode)
            want = np.array([orc.maxsim_bf16(q, ix.read_pages(int(p), 1)[0, : args.patches]) for p in i], np.float32)
            w = max(w, float(np.max(np.abs(s - want) / np.maximum(np.abs(want), 1e-6))))
        worst[name] = w
    res["max_rel_score_err_vs_float_oracle"] = worst

    def ids_of(mode, k=K, cn=None):
        def f(q, al):
            if cn is not None:
                ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
            return ix.query(q, k, mode=mode, allow=al)[1].tolist()
        return f

    def on_e4m3(f, slab=1):  # the same path with the coarse stage on the e4m3 (1) / fp4 (2) copy of the FDE slab
        def g(q, al):
            ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, slab)
            try:
                return f(q, al)
            finally:
                ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 0)
        return g

    modes = [("fp8_scan", ids_of("float_fp8")), ("fp8_then_float_n128", ids_of("fp8_then_float")), ("sign_bit_scan", ids_of("binary")),
             ("fde_top75_then_exact", ids_of("fde_then_float", cn=75)), ("fde_top1000_then_exact", ids_of("fde_then_float", cn=1000)),
             ("fde_coarse_recall_at_75", ids_of("fde", k=75)), ("fde_coarse_recall_at_1000", ids_of("fde", k=1000)),
             ("fde_e4m3_top75_then_exact", on_e4m3(ids_of("fde_then_float", cn=75))), ("fde_e4m3_coarse_recall_at_75", on_e4m3(ids_of("fde", k=75))),
             ("fde_fp4_top75_then_exact", on_e4m3(ids_of("fde_then_float", cn=75), 2)), ("fde_fp4_top1000_then_exact", on_e4m3(ids_of("fde_then_float", cn=1000), 2))]
    t0 = time.time()
    res["recall_at_10_vs_exact_bf16"] = recall_of(ix, rsets, truths, gaps, modes, allow_truth)
    # the e4m3 rerank an index WITHOUT an exact tier falls back to (MV_OPT_EXACT_TIER 2), on the same candidates: what the exact tier buys
    ix.set_option(L.MV_OPT_EXACT_TIER, 2)
    res["recall_at_10_vs_exact_bf16_with_the_e4m3_rerank_instead"] = recall_of(
        ix, {k_: rsets[k_] for k_ in ("hard_negatives", "clustered_topics")}, truths, gaps,
        [("fde_top75_then_fp8", ids_of("fde_then_float", cn=75)), ("fde_top1000_then_fp8", ids_of("fde_then_float", cn=1000))], allow_truth)
    ix.set_option(L.MV_OPT_EXACT_TIER, 0)
    res["recall_s"] = round(time.time() - t0, 1)
    ix.set_option(L.MV_OPT_FDE_COARSE_N, 1000)
    if split and in_hbm < n:
        try:  # placement of the split tier: the pages the reranks read most move into its HBM part (mv_index_exact_tier_rebalance)
            res["hot_pages_in_hbm"] = hot_pages_block(ix, rsets["clustered_topics"]["queries"], allow_truth)
        except Exception as e:  # noqa: BLE001
            res["hot_pages_in_hbm"] = {"error": repr(e)}
    ix.close()
    if split and args.exact_shard_lean:
        try:
            res["lean_fde_plus_split_exact_tier"] = exact_shard_lean(args, device, n, n_truth, budget, sets, truths, gaps)
        except Exception as e:  # noqa: BLE001
            res["lean_fde_plus_split_exact_tier"] = {"error": repr(e)}
    return res


def hot_pages_block(ix, cq, allow):
    """The split exact tier keeps the LEADING pages in HBM; mv_index_exact_tier_rebalance keeps the HOT ones there (read counts per page
    from every rerank).  Clustered-topics queries (overlapping candidate sets: the case a serving corpus presents) at coarse top-1000
    (e4m3 pruning to 128 exact reads per request), 32 requests per pass: device time of a batch and the share of the exact reads that
    crossed PCIe, before and after ONE rebalance trained on the first 32 queries; the other 32 queries of the same topics were never
    seen by the counters.  Answers must not change."""
    train, test = cq[:32], cq[32:64]

    def batch(qs32):
        dev, out = [], None
        for r in range(6):
            out, st = ix.query_batch(qs32, K, mode="fde_then_float", want_stats=True, allow=allow)
            if r >= 2:
                dev.append(st.total_device_ms)
        return float(np.median(dev)), [o[1].tolist() for o in out], [o[0].tolist() for o in out]

    def share(qs32):
        h0 = ix.exact_tier_hits()
        ix.query_batch(qs32, K, mode="fde_then_float", allow=allow)
        h1 = ix.exact_tier_hits()
        hb, ho = h1[0] - h0[0], h1[1] - h0[1]
        return round(ho / max(hb + ho, 1), 4)

    ix.rebalance_exact_tier(max_moves=1)  # (clears the counters the earlier measurements left; one swap at most)
    res = {"queries": "clustered_topics: 32 train + 32 unseen of the same topics; coarse top-1000 -> e4m3 pruning -> 128 exact reads per request, 32 requests per pass"}
    res["share_of_exact_reads_over_pcie_before"] = {"train": share(train), "unseen": share(test)}
    ix.rebalance_exact_tier(max_moves=1)
    b_ms, b_ids, b_sc = batch(train)
    t0 = time.time()
    moved = ix.rebalance_exact_tier()
    res["pages_moved_into_hbm"] = moved
    res["rebalance_s"] = round(time.time() - t0, 2)
    a_ms, a_ids, a_sc = batch(train)
    u_ms, _ui, _us = batch(test)
    ix.rebalance_exact_tier(max_moves=1)
    res["share_of_exact_reads_over_pcie_after"] = {"train": share(train), "unseen": share(test)}
    res["batch_of_32_device_ms"] = {"train_before": round(b_ms, 4), "train_after": round(a_ms, 4), "unseen_after": round(u_ms, 4)}
    res["same_ids_and_scores_as_before"] = a_ids == b_ids and a_sc == b_sc
    return res


def exact_shard_lean(args, device, n, n_truth, budget, sets, truths, gaps):
    """configs[3]'s shard WITHOUT the e4m3 slab (store option prune_slab=False, provider mi355x_fast_split_exact_lean): the FDE slab
    and the exact rows only.  In the FDE pipeline the e4m3 slab only serves the pruning stage behind lists longer than
    MV_OPT_RERANK_N, which the reference's own candidate rule (min(10 k, 75)) never reaches -- and its 164 GB hold the exact rows of
    another 625 k pages: ~80 % of a 1.25 M-page shard's exact tier then sits in HBM, and a batch's rerank stops being a PCIe mover."""
    from oracle import oracle as orc  # checker only (outside every timed region)

    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex

    stride = ((args.patches + 15) // 16) * 16
    t0 = time.time()
    ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, with_float=False, with_fde=True, with_host_exact=True, with_exact_split=True)
    in_hbm = ix.exact_hbm_pages
    res = {"pages": n, "slabs": "FDE(10240 bf16) in HBM; exact bf16 rows split between the rest of the HBM and pinned host RAM; no e4m3 / sign-bit slab",
           "exact_tier_pages_in_hbm": in_hbm, "exact_tier_in_hbm_GB": round(in_hbm * stride * 256 / 1e9, 1),
           "pinned_host_exact_tier_GB": round((n - in_hbm) * stride * 256 / 1e9, 1), "share_of_exact_reads_over_pcie": round((n - in_hbm) / n, 3),
           "create_and_pin_s": round(time.time() - t0, 1)}
    t0 = time.time()
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=args.patches)
    for name, st_ in sets.items():
        synth.plant_neighbours_any(ix, st_ if name.startswith("_") else st_["spec"], synth.SEED_CORPUS, args.patches, 0, n)
    res["fill_and_plant_s"] = round(time.time() - t0, 1)
    rsets = {k_: v for k_, v in sets.items() if not k_.startswith("_")}
    qs = rsets["planted"]["queries"]
    res["fde_then_exact_rerank"] = fde_pipeline_timings(ix, qs, n, (75, 1000))
    for cn in (75, 1000):
        n_mid, tier = ix.rerank_plan(cn, K, args.qtokens)
        res["fde_then_exact_rerank"][f"coarse{cn}"]["rerank_plan"] = {"tier": tier, "e4m3_pruning_to": n_mid, "exact_pages_read": (n_mid or cn)}
    ix.set_option(L.MV_OPT_FDE_COARSE_N, 75)
    w = 0.0
    for q in qs[:4]:
        s, i = ix.query(q, K, mode="fde_then_float")
        want = np.array([orc.maxsim_bf16(q, ix.read_pages(int(p), 1)[0, : args.patches]) for p in i], np.float32)
        w = max(w, float(np.max(np.abs(s - want) / np.maximum(np.abs(want), 1e-6))))
    res["max_rel_score_err_vs_float_oracle"] = w

    def ids_of(cn):
        def f(q, al):
            ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
            return ix.query(q, K, mode="fde_then_float", allow=al)[1].tolist()
        return f

    keep = {k_: rsets[k_] for k_ in ("planted", "hard_negatives", "clustered_topics")}
    rec = recall_of(ix, keep, truths, gaps, [("fde_top75_then_exact", ids_of(75)), ("fde_top1000_then_exact", ids_of(1000))], first_pages_bitmap(n_truth, n))
    res["recall_at_10_vs_exact_bf16"] = {k_: {m: v[m] for m in ("fde_top75_then_exact", "fde_top1000_then_exact")} for k_, v in rec.items() if k_ in keep}
    ix.close()
    return res


def two_tier(args, device):
    """fp8 scan -> top-n -> exact bf16 re-score from the exact tier (MV_MODE_FP8_THEN_FLOAT), on one index holding the e4m3
    slab, the bf16 slab (the truth, and the HBM form of the exact tier) and the PINNED-HOST exact tier the rerank kernel
    reads over PCIe (what a shard without room for a bf16 slab uses): recall of the fp8 scan alone next to the two-tier
    result, and what the second tier adds to the device time."""
    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex, synth_rows

    n = args.aux_pages
    stride = ((args.patches + 15) // 16) * 16
    res = {"pages": n, "host_tier_GB": round(n * stride * 256 / 1e9, 1)}
    t0 = time.time()
    ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, with_float=True, with_fp8=True, with_host_exact=True)
    res["create_with_pinned_host_tier_s"] = round(time.time() - t0, 1)
    t0 = time.time()
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=args.patches)
    res["fill_s"] = round(time.time() - t0, 1)
    sets = recall_sets(args, n, device, [])
    t0 = time.time()
    for st_ in sets.values():
        synth.plant_neighbours_any(ix, st_["spec"], synth.SEED_CORPUS, args.patches, 0, n)
    res["plant_s"] = round(time.time() - t0, 1)
    truths, gaps = {}, {}
    for sname, st_ in sets.items():
        truths[sname], gaps[sname] = exact_truth(ix, st_["queries"])
    qs = sets["hard_negatives"]["queries"]
    base = timed_mode(ix, qs, "float_fp8")
    res["fp8_scan_alone_device_ms"] = round(base["total_device_ms"], 4)

    def ids_of(mode):
        return lambda q, al: ix.query(q, K, mode=mode, allow=al)[1].tolist()

    res["by_rerank_n"] = {}
    for nn in (64, 128, 256):
        ix.set_option(L.MV_OPT_RERANK_N, nn)
        ent = {}
        for tier, code in (("hbm_bf16_slab", 0), ("pinned_host_over_pcie", 1)):
            ix.set_option(L.MV_OPT_EXACT_TIER, code)
            t = timed_mode(ix, qs, "fp8_then_float")
            ent[tier] = {"device_ms": round(t["total_device_ms"], 4), "added_ms_over_fp8_scan": round(t["total_device_ms"] - base["total_device_ms"], 4),
                         "rerank_ms": round(t["rerank_ms"], 4), "select_ms": round(t["select_ms"], 4),
                         "rerank_GBps": round(nn * args.patches * 256 / max(t["rerank_ms"], 1e-6) / 1e6, 1)}
        res["by_rerank_n"][f"n{nn}"] = ent
    ix.set_option(L.MV_OPT_RERANK_N, 128)
    # the same for a BATCH of 16 requests per pass of the batched fp8 scan, every list re-scored exactly in one launch
    from morphik_core_amd import synth as _synth

    res["batch_of_16"] = {}
    hq16, ht16 = qs[:16], truths["hard_negatives"][:16]
    for bv, name in ((0, "two_term_first_stage"), (7, "single_term_first_stage")):
        ix.set_option(L.MV_OPT_BATCH_VARIANT, bv)
        for tier, code in (("hbm_bf16_slab", 0), ("pinned_host_over_pcie", 1)):
            ix.set_option(L.MV_OPT_EXACT_TIER, code)
            dev, out_b = [], None
            for r in range(5):
                out_b, st = ix.query_batch(hq16, K, mode="fp8_then_float", want_stats=True)
                if r >= 2:
                    dev.append(st.total_device_ms)
            d = float(np.median(dev))
            res["batch_of_16"][f"{name}_{tier}"] = {
                "device_ms_per_batch": round(d, 4), "device_us_per_request": round(d * 1e3 / 16, 1), "query_pages_per_s": round(16 * n / d * 1e3, 1),
                "recall_at_10_hard_negatives": round(float(np.mean([_synth.recall_at_k(out_b[j][1].tolist(), ht16[j]) for j in range(16)])), 4)}
    ix.set_option(L.MV_OPT_BATCH_VARIANT, -1)
    ix.set_option(L.MV_OPT_EXACT_TIER, 1)  # recall through the host tier (same rows: same answers as the HBM tier)
    res["recall_at_10_vs_exact_bf16"] = recall_of(ix, sets, truths, gaps, [("fp8_scan", ids_of("float_fp8")), ("fp8_then_float_n128", ids_of("fp8_then_float"))])
    ix.close()
    return res


def fde_encode_block(args, device):
    """Roofline entry of the FDE DOCUMENT encode (fde.generate_document_encoding, fast_multivector_store.py:447-449): corpus build of
    the same pages with and without the FDE slab; the difference is the encode, which reads every page's bf16 rows (262 144 B; twice in
    the two-pass form) and writes 20 480 B.  Default since round 4 (MV_OPT_FDE_ENCODE_VARIANT 4): pass 1 fde_hash_kernel -- the SimHash
    sketches as k-ordered fp32 fmaf chains on the f32 matrix path (v_mfma_f32_16x16x4_f32: 128 x 112 MACs per row as issued, 7 column
    tiles) -> one partition byte per (row, repetition); pass 2 fde_project_kernel -- the AMS projection ({0, +1, -1} matrix against rows
    that are already bf16: exact products) on the bf16 matrix path (v_mfma_f32_16x16x32_bf16) and the bucket sums as a ONE-HOT matrix
    product on the f32 path (16 rows x 32 partitions x 16 columns per repetition and tile).  The round-3 one-pass kernel (variant 3:
    bucket sums through LDS float atomics) and the round-2 kernel (variant 1: everything on the f32 path, operands from LDS) are timed
    beside it."""
    from morphik_core_amd import _lib, synth
    from morphik_core_amd.index import MvIndex

    n = min(args.aux_pages, 100_000)
    stride = ((args.patches + 15) // 16) * 16
    t = {}
    for key, fde, variant in (("gen", False, None), ("two_pass", True, 4), ("f32_pipe", True, 1)):
        for warm in (True, False):
            ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, with_float=True, with_fde=fde)
            if variant is not None:
                ix.set_option(_lib.MV_OPT_FDE_ENCODE_VARIANT, variant)
            t0 = time.perf_counter()
            ix.fill_synthetic(synth.SEED_CORPUS, 0, min(n, 2000) if warm else n, n_rows=args.patches)  # warm-up: tables, clocks
            if not warm:
                t[key] = time.perf_counter() - t0
            ix.close()
    enc = max(t["two_pass"] - t["gen"], 1e-9)
    enc_f32 = max(t["f32_pipe"] - t["gen"], 1e-9)
    us, us_f32 = enc / n * 1e6, enc_f32 / n * 1e6
    simhash = 2.0 * args.patches * 128 * 112    # f32 MFMA flops per page as issued (7 column tiles of 16 hashes)
    ams = 2.0 * args.patches * 128 * (20 * 16)  # AMS flops per page as issued (one 16-column tile per repetition)
    onehot = 2.0 * args.patches * 32 * 16 * 20  # one-hot bucket sums as issued: rows x 32 partitions x 16 columns per repetition
    useful = 2.0 * args.patches * 128 * (20 * 5) + args.patches * 128 * 20  # SimHash MACs + one signed add per (dim, repetition)
    f32_issued = simhash + onehot
    return {"pages": n, "us_per_page": round(us, 3), "pages_per_s": round(n / enc, 1), "page_input_GBps": round(2 * args.patches * 256 / us / 1e3, 1),
            "frac_hbm_8TBps": round(2 * args.patches * 256 / us / 1e3 / HBM_PEAK_GBPS, 4),
            "f32_mfma_TFLOPs_as_issued": round(f32_issued / us / 1e6, 1), "frac_f32_mfma_155TF": round(f32_issued / us / 1e6 / 155.0, 4),
            "bf16_mfma_TFLOPs_as_issued": round(ams / us / 1e6, 1),
            "matrix_pipe_time_frac_est": round((f32_issued / 155e12 + ams / 2500e12) / (us * 1e-6), 4),
            "useful_TFLOPs": round(useful / us / 1e6, 1),
            "bound": "the f32 matrix pipe: SQ counters (profiles/r4/pmc_fde_encode_kernels_r4.json) show it 51 % busy in the hash pass and 47 % in the "
                     "projection pass, waves waiting on LDS 0.4 % / 0.0 % of their cycles (the round-3 kernel: matrix pipes 11 % busy, 17 % of the "
                     "wave cycles behind LDS atomics) -- DESIGN.md 3.9",
            "round2_kernel_f32_pipe_only": {"us_per_page": round(us_f32, 3), "pages_per_s": round(n / enc_f32, 1),
                                            "f32_mfma_TFLOPs_as_issued": round((simhash + ams) / us_f32 / 1e6, 1),
                                            "frac_f32_mfma_155TF": round((simhash + ams) / us_f32 / 1e6 / 155.0, 4)},
            "corpus_generation_us_per_page": round(t["gen"] / n * 1e6, 3)}


def fp32_split_block(args, device):
    """The fp32-faithful tier (MV_WITH_FLOAT_LO): pages kept as split-bf16 pairs hi + lo -- the reference's 4 bytes per element
    (fp32 `.npy` pages scored in fp32: fast_multivector_store.py:676-681, :736, :774, :553-555) -- scored on the bf16 MFMA as
    qhi.phi + qlo.phi + qhi.plo.  Reports (i) the score error against a numpy fp32 restatement on fp32 pages and an fp32 query,
    (ii) the rate of the three-term full scan against the HBM roof at ITS bytes (2 x 262 144 B per page), (iii) the hi-only scan
    with the fp32 query's lo chain riding along (the always-on query split on a plain bf16 index costs this), (iv) the cascade
    hi-only scan -> split-bf16 re-score of the best 128."""
    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex

    import torch

    stride = ((args.patches + 15) // 16) * 16
    free_b, _tot = torch.cuda.mem_get_info(device)
    n = int(min(args.aux_pages, (free_b - (8 << 30)) // (2 * stride * 256 + 64)))
    n_f32 = 256
    rng = np.random.default_rng(20260930)

    def unit(rows):
        x = rng.standard_normal((rows, 128)).astype(np.float32)
        return x / np.linalg.norm(x, axis=-1, keepdims=True)

    ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, with_float_lo=True)
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n - n_f32, n_rows=args.patches)
    pages = [unit(args.patches) for _ in range(n_f32)]
    first = ix.add(pages)
    qs = [unit(args.qtokens) for _ in range(4)]
    res = {"pages": n, "fp32_sample_pages": n_f32, "bytes_per_page_hi_plus_lo": 2 * args.patches * 256,
           "note": "fp32 unit-row pages and queries (not bf16-representable); reference = numpy fp32 (pages @ q.T).max(0).sum(); kernel-only HIP-event times"}
    want = np.stack([np.array([(p @ q.T).max(0).sum(dtype=np.float32) for p in pages], np.float32) for q in qs])
    cand = np.arange(first, first + n_f32)
    got = np.stack([ix.score_candidates(q, cand) for q in qs])
    res["max_rel_score_err_vs_fp32_reference"] = float(np.max(np.abs(got - want) / np.abs(want)))
    plain = MvIndex(capacity_pages=n_f32, stride_rows=stride, device=device)  # what rounding the pages to bf16 costs (the fp32 query still exact)
    plain.add(pages)
    got_hi = np.stack([plain.score_candidates(q, np.arange(n_f32)) for q in qs])
    plain.close()
    res["max_rel_score_err_hi_only_pages"] = float(np.max(np.abs(got_hi - want) / np.abs(want)))
    for lo_scan, key, bpp in ((1, "scan_hi_plus_lo", 2 * args.patches * 256), (0, "scan_hi_only_fp32_query", args.patches * 256), (2, "cascade_hi_scan_then_rescore_128", args.patches * 256)):
        ix.set_option(L.MV_OPT_FLOAT_LO_SCAN, lo_scan)
        t = timed_mode(ix, qs, "float")
        res[key] = dict(scan_entry(n, bpp, t["score_kernel_ms"] if lo_scan != 2 else t["coarse_ms"]), total_device_ms=round(t["total_device_ms"], 4))
        if lo_scan == 2:
            res[key]["rerank_ms"] = round(t["rerank_ms"], 4)
    # a batch of fp32 requests in cascade mode: ONE batched hi scan per group + the one-launch split rerank of all lists (== the single requests)
    ix.set_option(L.MV_OPT_FLOAT_LO_SCAN, 2)
    bqs = [unit(args.qtokens) for _ in range(16)]
    singles = [ix.query(q, K) for q in bqs]
    ms = []
    for r in range(5):
        got_b, st_b = ix.query_batch(bqs, K, want_stats=True)
        if r >= 2:
            ms.append(st_b.total_device_ms)
    res["cascade_batch_of_16"] = {"device_ms_per_batch": round(float(np.median(ms)), 4), "sixteen_single_requests_device_ms": round(16 * res["cascade_hi_scan_then_rescore_128"]["total_device_ms"], 4),
                                  "same_ids_and_scores_as_the_single_requests": all(a[1].tolist() == b[1].tolist() and np.array_equal(a[0], b[0]) for a, b in zip(got_b, singles))}
    # the same queries rounded to bf16 (RNE): the one-term kernel on the hi slab -- what the query's lo chain adds to an HBM-bound scan
    bq = [((q.view(np.uint32) + 0x7FFF + ((q.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16) for q in qs]
    ix.set_option(L.MV_OPT_FLOAT_LO_SCAN, 0)
    t = timed_mode(ix, bq, "float")
    res["scan_hi_only_bf16_query"] = scan_entry(n, args.patches * 256, t["score_kernel_ms"])
    ix.close()
    return res


def aux_summary(out, aux):
    """<= ~700 bytes of the secondary kernels' figures for the driver-parsed line (configs[3] / [4] and the batched scan would
    otherwise only exist in the detail record above it).  Missing measurements are left out, never invented."""
    def g(*path):
        x = aux
        for k in path:
            if not isinstance(x, dict) or k not in x:
                return None
            x = x[k]
        return x

    rec = {
        "fp8_scan_frac": g("full_shard", "fp8_scan", "frac_hbm_8TBps"),
        "sign_bit_frac": g("full_shard", "sign_bit_scan", "frac_hbm_8TBps"),
        "fde_scan_frac": g("full_shard", "fde_coarse_scan", "frac_hbm_8TBps"),
        "fde_batch32_frac": g("full_shard", "fde_then_fp8_rerank", "coarse75", "batch_of_32", "coarse_pass_frac_hbm_8TBps"),
        "fde_batch32_frac_placed": g("full_shard", "fde_placement_trial", "coarse75_batch_of_32_after", "coarse_pass_frac_hbm_8TBps"),
        "batched_bf16_B16_PF": (lambda v: None if v is None else round(v / 1000.0, 3))(g("batched_float", "B16", "TFLOPs")),
        "batched_bf16_B16_frac_2500TF": g("batched_float", "B16", "frac_mfma_bf16_2500TF"),
        "fde_request_ms": g("exact_shard", "fde_then_exact_rerank", "coarse75", "one_request", "device_ms"),
        "fde8_scan_frac": g("full_shard", "fde_coarse_scan_e4m3", "frac_hbm_8TBps"),
        "fde8_request_ms": g("exact_shard", "fde_e4m3_coarse_then_exact_rerank", "coarse75", "one_request", "device_ms"),
        "fde8_batch32_ms": g("full_shard", "fde_e4m3_then_fp8_rerank", "coarse75", "batch_of_32", "device_ms_per_batch"),
        "fde8_75_recall_hard": g("exact_shard", "recall_at_10_vs_exact_bf16", "hard_negatives", "fde_e4m3_top75_then_exact"),
        "fde4_scan_frac": g("full_shard", "fde_coarse_scan_fp4", "frac_hbm_8TBps"),
        "fde4_request_ms": g("exact_shard", "fde_fp4_coarse_then_exact_rerank", "coarse75", "one_request", "device_ms"),
        "fde4_batch32_ms": g("full_shard", "fde_fp4_then_fp8_rerank", "coarse75", "batch_of_32", "device_ms_per_batch"),
        "fde4_75_recall_hard": g("exact_shard", "recall_at_10_vs_exact_bf16", "hard_negatives", "fde_fp4_top75_then_exact"),
        "fde_batch32_exact_ms": g("exact_shard", "fde_then_exact_rerank", "coarse1000", "batch_of_32", "device_ms_per_batch"),
        "hot_pages_batch32_ms": [g("exact_shard", "hot_pages_in_hbm", "batch_of_32_device_ms", "train_before"), g("exact_shard", "hot_pages_in_hbm", "batch_of_32_device_ms", "train_after"),
                                 g("exact_shard", "hot_pages_in_hbm", "batch_of_32_device_ms", "unseen_after")] if g("exact_shard", "hot_pages_in_hbm", "batch_of_32_device_ms") else None,
        "fp8_recall_hard": g("exact_shard", "recall_at_10_vs_exact_bf16", "hard_negatives", "fp8_scan"),
        "fp8_then_float_recall": g("exact_shard", "recall_at_10_vs_exact_bf16", "hard_negatives", "fp8_then_float_n128"),
        "fde75_recall_hard": g("exact_shard", "recall_at_10_vs_exact_bf16", "hard_negatives", "fde_top75_then_exact"),
        "fp32_split_max_rel_err": g("fp32_split_bf16", "max_rel_score_err_vs_fp32_reference"),
        "fp32_hi_lo_scan_frac": g("fp32_split_bf16", "scan_hi_plus_lo", "frac_hbm_8TBps"),
        "fp32_batch16_ms": [g("fp32_split_bf16", "cascade_batch_of_16", "device_ms_per_batch"), g("fp32_split_bf16", "cascade_batch_of_16", "sixteen_single_requests_device_ms")] if g("fp32_split_bf16", "cascade_batch_of_16") else None,
        "ragged_packed_valid_frac": g("ragged_corpus", "packed", "frac_hbm_8TBps_valid_bytes"),
        "ragged_fixed_valid_frac": g("ragged_corpus", "fixed_stride", "frac_hbm_8TBps_valid_bytes"),
        "ragged_capacity_gain": g("ragged_corpus", "capacity_gain_packed_over_fixed"),
        "q16_q64_frac": [g("query_length_sweep", "Q16", "frac_hbm_8TBps"), g("query_length_sweep", "Q64", "frac_hbm_8TBps")] if g("query_length_sweep", "Q16") else None,
        "aux_s": g("aux_child_seconds"),
    }
    rec = {k: v for k, v in rec.items() if v is not None}
    if "aux_child_error" in aux:
        rec["error"] = str(aux["aux_child_error"])[:80]
    return rec


def ragged_block(args, device):
    """A ColQwen2.5-like corpus (the reference's real encoder emits a different token count per page: colpali_embedding_model.py:47-52):
    n_rows(page) uniform in 550..1024, on BOTH layouts -- fixed stride_rows slots and MV_LAYOUT_PACKED (whole 16-row tiles back to back).
    The scan reads only a page's valid tiles either way, so the rate is quoted on VALID bytes; what the packed layout buys is HBM:
    pages resident per GB."""
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex, synth_ragged_rows

    import torch

    stride = ((args.patches + 15) // 16) * 16
    lo_rows, hi_rows = min(550, args.patches), args.patches
    free_b, _tot = torch.cuda.mem_get_info(device)
    n = int(min(args.ragged_pages, (free_b - (8 << 30)) // (stride * 256 + 64)))
    rows = np.array([synth_ragged_rows(synth.SEED_CORPUS, u, lo_rows, hi_rows) for u in range(n)], np.int64)
    valid_bytes = int(rows.sum()) * 256
    slot_rows = int(((rows + 15) // 16 * 16).sum())
    qs = [synth_rows_dev(synth.SEED_QUERIES, 2000 + j, args.qtokens, device) for j in range(4)]
    res = {"pages": n, "rows_per_page": f"uniform {lo_rows}..{hi_rows} (mean {rows.mean():.1f})", "valid_GB": round(valid_bytes / 1e9, 2)}
    ref = None
    for name, kw in (("fixed_stride", {}), ("packed", {"packed": True, "capacity_rows": slot_rows})):
        ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, **kw)
        ix.fill_synthetic_ragged(synth.SEED_CORPUS, 0, n, lo_rows, hi_rows)
        t = timed_mode(ix, qs, "float")
        top = [ix.query(q, K) for q in qs]
        if ref is None:
            ref = top
        ms = t["score_kernel_ms"]
        slab_bytes = ix.capacity_rows * 256
        res[name] = {"kernel_ms": round(ms, 4), "pages_per_s": round(n / ms * 1e3, 1), "valid_GBps": round(valid_bytes / ms / 1e6, 1),
                     "frac_hbm_8TBps_valid_bytes": round(valid_bytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "slab_GB": round(slab_bytes / 1e9, 2),
                     "pages_per_262GB_slab": int(262.144e9 / (slab_bytes / n)),
                     "same_top10_as_fixed_stride": all(a[1].tolist() == b[1].tolist() and a[0].tolist() == b[0].tolist() for a, b in zip(top, ref))}
        ix.close()
    res["capacity_gain_packed_over_fixed"] = round(res["packed"]["pages_per_262GB_slab"] / res["fixed_stride"]["pages_per_262GB_slab"], 3)
    return res


def synth_rows_dev(seed, unit, n_rows, device):
    from morphik_core_amd.index import synth_rows

    return synth_rows(seed, unit, n_rows, device=device)


def serving_block(args):
    """QPS / latency of `await store.query_similar(...)` (the plugin boundary, where the reference logs its per-query totals:
    fast_multivector_store.py:513-605) under 1 / 8 / 32 / 128 concurrent asyncio clients, coalescer off / adaptive, next
    to the device time of the same requests -- tools/serve_bench.py on an MI355XFastMultiVectorStore of args.aux_pages pages."""
    import types

    from tools import serve_bench

    a = types.SimpleNamespace(mode="fde_then_float", pages=args.aux_pages, patches=args.patches, clients="1,8,32,128", seconds=args.aux_serve_seconds, k=K, null_index=False)
    return serve_bench.measure(a)


def batched_float_block(ix, queries, n, args, mfma_cal):
    """Batched MFMA scan (B x 32 tokens per slab pass) on the headline bf16 index."""
    from morphik_core_amd import synth

    out = {}
    for B in (4, 16):
        ms = []
        for r in range(6):
            res_b, st = ix.query_batch(queries[:B], K, want_stats=True)
            if r >= 2:
                ms.append(st.score_kernel_ms)
        m = float(np.median(ms))
        tf = 2.0 * B * args.qtokens * args.patches * 128 * n / m / 1e9
        out[f"B{B}"] = {"pages": n, "kernel_ms": round(m, 4), "query_pages_per_s": round(B * n / m * 1e3, 1), "TFLOPs": round(tf, 1),
                        "frac_mfma_bf16_2500TF": round(tf / MFMA_BF16_PEAK_TF, 4), "GBps": round(n * args.patches * 256 / m / 1e6, 1),
                        "frac_hbm_8TBps": round(n * args.patches * 256 / m / 1e6 / HBM_PEAK_GBPS, 4)}
    return out


def embed_workload(args, pages, quick=False):
    """BASELINE configs[1]: ColPali-v1.2 architecture embeds synthetic pages -> device ingest -> MaxSim top-10."""
    from tools.bench_embed import run as run_embed

    return run_embed(pages=pages, batch=32, feed=16, preset="colpali-v1.2", queries=4 if quick else 8)


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks here (one per GPU, rendezvous on
    127.0.0.1) by re-executing this file under torch.distributed.run, and relay rank 0's JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"bench.py: launching {args.gpus} ranks: {' '.join(cmd[1:8])} ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in lines:  # rank 0's detail record, then its headline (last)
        print(ln, flush=True)
    return p.returncode if (p.returncode or lines) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=1_000_000, help="total corpus pages across all ranks")
    ap.add_argument("--patches", type=int, default=1024)
    ap.add_argument("--qtokens", type=int, default=32)
    ap.add_argument("--variant", type=int, default=-1, help="float kernel variant (-1 = library default)")
    ap.add_argument("--cpu-sample-pages", type=int, default=20480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="a tenth of every CPU timing budget (tests of the result line; not for a record)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--workload", choices=["float", "fp8", "binary", "fde_fp8", "embed"], default="float",
                    help="float = BASELINE configs[2] (the headline, default); fp8 = e4m3 slab (configs[4]); binary = sign-bit "
                         "max_sim (MultiVectorStore); fde_fp8 = FDE coarse top-1000 -> exact fp8 rerank (configs[3] shard shape); "
                         "embed = configs[1] (ColPali-v1.2 architecture, 1 k pages -> top-10)")
    ap.add_argument("--no-aux", action="store_true", help="skip the secondary kernels' quick measurements (aux_paths)")
    ap.add_argument("--aux-pages", type=int, default=200_000, help="pages of the two-tier (fp8 -> exact bf16 from the pinned-host tier) index in aux_paths (0 = skip)")
    ap.add_argument("--exact-shard-pages", type=int, default=1_250_000,
                    help="pages of the shard-shaped index WITH its exact tier in pinned host RAM (aux_paths.exact_shard; n x 256 KiB = 328 GB at 1.25 M "
                         "pages), cut to what the process may pin (see --exact-shard-pin-frac); 0 = skip")
    ap.add_argument("--exact-shard-pin-frac", type=float, default=0.7,
                    help="share of mv_host_pin_budget_bytes() (memory cgroup limit - usage - headroom) the exact shard's pinned tier may take (--exact-shard-split 0)")
    ap.add_argument("--exact-shard-split", type=int, default=1,
                    help="1: MV_WITH_EXACT_SPLIT -- the exact rows of the leading pages in the HBM the slabs leave free, the rest pinned (the full 1.25 M pages fit); 0: all pinned")
    ap.add_argument("--exact-shard-lean", type=int, default=1,
                    help="1: after the exact shard, the same shard WITHOUT the e4m3 / sign-bit slabs (FDE slab + split exact tier only: most exact rows in HBM)")
    ap.add_argument("--exact-shard-split-pin-frac", type=float, default=0.84,
                    help="share of the pin budget the PINNED part of a split exact tier may take (the page count is cut to keep it)")
    ap.add_argument("--full-shard-pages", type=int, default=1_250_000,
                    help="pages of the e4m3 + FDE + sign-bit index in aux_paths.full_shard (BASELINE configs[3]/[4] per-GPU shard of 10 M pages / 8 GPUs; 0 = skip)")
    ap.add_argument("--ragged-pages", type=int, default=300_000, help="pages of the ColQwen-like ragged corpus (550..1024 rows per page) scanned on the fixed-stride and the packed layout (aux_paths.ragged_corpus; 0 = skip)")
    ap.add_argument("--aux-embed-pages", type=int, default=1000, help="pages of the full-size encoder run inside aux_paths (configs[1] names 1 k pages; 0 = skip)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend for N>1 (nccl == RCCL; gloo + MV_BENCH_SINGLE_DEVICE=1 lets N ranks share one GPU to "
                         "exercise the multi-rank path on a 1-GPU box -- a functional check, not a measurement)")
    ap.add_argument("--aux-serve-seconds", type=float, default=0.8, help="seconds per (clients, coalescer) cell of aux_paths.serving")
    ap.add_argument("--aux-timeout", type=int, default=900, help="limit (s) of the child process that measures aux_paths")
    ap.add_argument("--aux-child", default=None, help=argparse.SUPPRESS)  # internal: state file of the aux child (run_aux_child)
    args = ap.parse_args()
    if args.cpu_baseline_quick:
        global CPU_SWEEP_SCALE
        CPU_SWEEP_SCALE = 0.1

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the product path is HIP-only (no CPU fallback)")
    single_device = os.environ.get("MV_BENCH_SINGLE_DEVICE") == "1"
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.aux_child:
        aux_child(args, local_rank)
        return

    if args.workload == "embed":
        if world != 1:
            sys.exit("bench.py --workload embed: the encoder scales as independent replicas; run it with --gpus 1")
        pages = args.pages if args.pages != 1_000_000 else 1000
        r = embed_workload(args, pages)
        print(json.dumps({
            "metric": "pages embedded + ingested per second (ColPali-v1.2 architecture, bf16) -> MaxSim top-10", "value": r["embed_pages_per_s"],
            "unit": "pages/s", "n_gpus": 1, "steps": pages, "warmup": 32, "ms_per_step": round(1e3 / r["embed_pages_per_s"], 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic page images; random-init weights of the ColPali-v1.2 architecture (no checkpoint in this environment)",
            "config": {"workload": r["workload"], "pages_total": pages, "rows_per_page": r["rows_per_page"], "params": r["params"], "k": 10,
                       "parallelism": "replicas only"},
            "roofline": {"bound": "mfma", "kernel": "hipBLASLt GEMMs of the PyTorch-ROCm forward (plumbing, not a hand-written kernel)",
                         "achieved": r["embed_tflops_est"], "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(r["embed_tflops_est"] / MFMA_BF16_PEAK_TF, 4), "traffic": None},
            "cpu_baseline": None, "detail": r}), flush=True)
        return

    # MV_BENCH_FORCE_DIST=1 runs the collective path even with one rank (RCCL smoke test on a 1-GPU box)
    dist_on = world > 1 or os.environ.get("MV_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collective payloads live

    from morphik_core_amd import _lib, sharded, synth
    from morphik_core_amd.index import MvIndex, calibrate, synth_rows

    stride = ((args.patches + 15) // 16) * 16
    WL = {
        "float": dict(mode="float", flags=dict(with_float=True), row_bytes=256, dtype="bf16", resident=stride * 256),
        "fp8": dict(mode="float_fp8", flags=dict(with_float=False, with_fp8=True), row_bytes=128, dtype="fp8_e4m3", resident=stride * 128 + 4),
        "binary": dict(mode="binary", flags=dict(with_float=False, with_binary=True), row_bytes=16, dtype="u1 (sign bits)", resident=stride * 16),
        "fde_fp8": dict(mode="fde_then_float", flags=dict(with_float=False, with_fp8=True, with_fde=True), row_bytes=None, dtype="bf16 FDE + fp8_e4m3",
                        resident=stride * 128 + 20480 + 8),
    }[args.workload]
    page_bytes = WL["resident"]

    # ---- size the shard to the HBM that is actually free
    free_b, total_b = torch.cuda.mem_get_info(dev)
    reserve = 6 << 30
    if single_device:
        # the ranks share one GPU: an equal share of its TOTAL memory (what is free right now depends on which ranks allocated first)
        free_b = total_b // world
        reserve = 2 << 30
    fit = max(int((free_b - reserve) // (page_bytes + 64)), 1)
    if args.scaling == "strong":
        n_total = args.pages
        lo, hi = sharded.shard_range(n_total, rank, world)
        if hi - lo > fit:
            n_total = fit * world
    else:
        n_total = min(args.pages, fit) * world
    if dist_on:  # agree on the smallest feasible corpus
        t = torch.tensor([n_total], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        n_total = int(t.item())
        flush_c_stdio()  # every rank: the collective library's start-up banner leaves its buffer NOW
    lo, hi = sharded.shard_range(n_total, rank, world)
    n_local = hi - lo
    log(f"[rank {rank}] HBM free {free_b/2**30:.1f} GiB of {total_b/2**30:.1f}; corpus {n_total} pages, shard [{lo},{hi}) = {n_local*page_bytes/1e9:.1f} GB")

    measured_peak = measured_nt = measured_mfma = None  # taken right after the timed run, on the warm GPU (see below)

    t0 = time.time()
    ix = MvIndex(capacity_pages=n_local, stride_rows=stride, device=local_rank, id_base=lo, **WL["flags"])
    if args.variant >= 0:
        ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, args.variant)
    if args.workload == "fde_fp8":
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 1000)
    ix.fill_synthetic(synth.SEED_CORPUS, lo, n_local, n_rows=args.patches)
    queries = [synth_rows(synth.SEED_QUERIES, qi, args.qtokens, device=local_rank) for qi in range(N_QUERIES)]
    spec = synth.planted_spec(queries, n_total, args.patches, n_ranks=N_PLANTED)
    if args.workload == "float":
        synth.plant_neighbours(ix, spec, lo, hi)
    else:
        synth.plant_neighbours_any(ix, spec, synth.SEED_CORPUS, args.patches, lo, hi)
    # the query / page sets the lossy paths are scored on (aux_paths): written into the bf16 corpus now, so the exact bf16
    # truth can be taken from THIS slab before it is freed for the configs[3] / [4] shard
    want_full = world == 1 and args.workload == "float" and not args.no_aux and (args.full_shard_pages > 0 or args.exact_shard_pages > 0)
    rsets = None
    n_exact = exact_budget = 0
    n_truth = n_total
    if want_full:
        t1 = time.time()
        n_exact, exact_budget, exact_split = exact_shard_pages(args, stride, local_rank)
        if n_exact < 50_000:
            n_exact = 0  # nothing worth measuring fits: the recall sets then span the whole corpus, as the lossy paths of full_shard need them
        else:
            n_truth = min(n_total, n_exact)  # every structured page lies inside the pages the exact shard will hold
        rsets = recall_sets(args, n_truth, local_rank, spec)
        for name, st_ in rsets.items():
            synth.plant_neighbours(ix, st_["spec"], lo, hi)
        log(f"[rank {rank}] recall sets: {sum(len(v['spec']) for v in rsets.values())} structured pages written in {time.time()-t1:.1f}s")
    MODE = WL["mode"]
    torch.cuda.synchronize()
    log(f"[rank {rank}] corpus generated + planted in {time.time()-t0:.1f}s")

    stats = []
    gpu_topk = sharded.make_gpu_local_topk(ix, dev, MODE, collect_stats=stats)
    # one rank's step: RCCL -- scan enqueued behind torch's stream, ONE all-gather of a 16-byte-aligned {ids, scores} block, one
    # library merge launch, no host wait in between; gloo (ranks sharing one GPU) -- the same step through host memory
    fast_searcher = sharded.GpuShardedSearcher(ix, dev, MODE, collect_stats=stats) if (dist_on and args.backend == "nccl") else None
    host_searcher = sharded.HostShardedSearcher(ix, MODE, collect_stats=stats) if (dist_on and args.backend == "gloo") else None
    # config 4 across ranks: GLOBAL coarse top-1000, owners rerank (the same candidate set as one big index); over RCCL the
    # device-resident stages (no host copy of any intermediate), over gloo the host-driven cross-check
    two_stage = None
    if dist_on and args.workload == "fde_fp8":
        two_stage = sharded.GpuTwoStageSearcher(ix, dev) if args.backend == "nccl" else sharded.make_gpu_two_stage(ix, cdev)

    def step(i):
        q = queries[i % N_QUERIES]
        if not dist_on:
            s, ids, st = ix.query(q, K, mode=MODE, want_stats=True)
            stats.append(st)
            return s, ids
        if two_stage is not None:
            return two_stage.query(q, K, coarse_n=1000)
        if fast_searcher is not None:
            return fast_searcher.query(q, K)
        return host_searcher.query(q, K)

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if fast_searcher is not None:
        fast_searcher.flush()  # (its timings trail the queries by one)
    stats.clear()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if fast_searcher is not None:
        fast_searcher.flush()
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed
    local_only_ms = None
    if dist_on:  # the same steps WITHOUT the exchange (scan + local top-k only): what the collective + merge add per step
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if host_searcher is not None:
                ix.query(queries[i % N_QUERIES], K, mode=MODE)  # the gloo step's local part: results through the pinned host buffers
            else:
                gpu_topk(queries[i % N_QUERIES], K)
        fence()
        t = torch.tensor([(time.perf_counter() - t0) / args.steps * 1e3], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        local_only_ms = float(t.item())
    exchange_ms = None
    if dist_on and two_stage is None:  # ... and the exchange BY ITSELF: the all-gather of the k pairs + the merge, no scan in front of it
        ex = fast_searcher if fast_searcher is not None else host_searcher
        for i in range(3):
            ex.exchange_only(K)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ex.exchange_only(K)
        fence()
        t = torch.tensor([(time.perf_counter() - t0) / args.steps * 1e3], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange_ms = float(t.item())

    # ---- roofline of the dominant kernel (the page scan), from HIP events recorded in the timed region
    kms = np.array([s.score_kernel_ms for s in stats[: args.steps] if s is not None and s.score_kernel_ms > 0])  # the timed steps' launches
    if args.workload == "fde_fp8":  # coarse scan of every FDE vector + exact rerank of 1000 candidates
        bytes_per_launch = n_local * 20480 + min(1000, n_local) * args.patches * 128
    else:
        bytes_per_launch = n_local * args.patches * WL["row_bytes"]  # algorithmic: every valid patch row read once
    my_kms = float(kms.mean()) if kms.size else 0.0
    per_rank_kms = [my_kms]
    if dist_on:  # report the slowest rank's kernel, and every rank's
        t = torch.tensor([my_kms], dtype=torch.float64, device=cdev)
        allk = torch.empty(world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allk, t)
        per_rank_kms = [float(x) for x in allk.cpu().tolist()]
    k_ms = max(per_rank_kms)
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    # ---- parity inside the bench (outside the timed region): recall@10 and sampled oracle scores
    recall = []
    for qi in range(N_QUERIES):
        s, ids = step(qi) if dist_on else ix.query(queries[qi], K, mode=MODE)
        if dist_on:
            torch.cuda.synchronize()
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        ids = [p for p in ids if p >= 0]
        planted = [p for (qq, r, p, _, _) in spec if qq == qi]
        recall.append(synth.recall_at_k(ids, planted))
    recall10 = float(np.mean(recall))

    out = None
    if rank == 0:
        traffic, traffic_src, traffic_kernel = pmc_traffic(n_local, args.patches) if args.workload == "float" else (None, None, None)
        roofline = {
            "bound": "hbm",
            "kernel": {"float": "maxsim_ldsdma_kernel (bf16 page scan, variant %s)" % (args.variant if args.variant >= 0 else "default: nt LDS-DMA"),
                       "fp8": "maxsim_fp8_pair_kernel (e4m3 page scan, two pages per workgroup, MX-scaled MFMA)", "binary": "maxsim_binary_mfma2_kernel (sign-bit scan, FP4 MFMA)",
                       "fde_fp8": "fde_scan_kernel + top-1000 + maxsim_fp8_kernel rerank (whole device span)"}[args.workload],
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "measured_read_peak": None if measured_peak is None else round(measured_peak, 1),
            "measured_read_peak_kind": "the scan's own nt LDS-DMA ring with the arithmetic removed (MV_CAL_READ_LDSDMA)",
            "frac_of_measured_peak": None if not measured_peak else round(achieved / measured_peak, 4),
            "measured_plain_nt_read": None if measured_nt is None else round(measured_nt, 1),
            "measured_mfma_bf16_tflops": None if not measured_mfma else round(measured_mfma, 1),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_kernel": traffic_kernel,
            "lib_sha256": lib_sha256()[:16],
            "src_sha256": _try(lambda: src_sha256()[:16]),  # the traffic record is accepted for this built file or for these sources (another checkout embeds another path)
            "bytes_per_launch": bytes_per_launch,
            "kernel_ms_avg": round(k_ms, 4),
            "kernel_ms_per_rank": [round(x, 4) for x in per_rank_kms],
            "launches_timed": int(kms.size),
            "mfma_tflops_achieved": round(2.0 * args.qtokens * args.patches * 128 * n_local / (k_ms * 1e-3) / 1e12, 2) if k_ms > 0 else 0.0,
        }
        cpu = None
        max_rel = None
        if not args.no_cpu_baseline and args.workload == "float":  # rank 0 of any world size: its own shard is the sample
            from oracle import oracle as orc  # checker / baseline only

            ns = min(args.cpu_sample_pages, n_local)
            sample = ix.read_pages(0, ns)[:, : args.patches]  # the device's own bytes (planted rows included)
            q0 = queries[0]
            cpu = cpu_baseline(sample, q0)
            want = orc.maxsim_float_np(orc.bf16_to_f32(q0), orc.bf16_to_f32(sample))
            full = ix.score_all(q0)
            max_rel = float(np.max(np.abs(full[:ns] - want) / np.maximum(np.abs(want), 1e-6)))
            # the device generator must equal the oracle's across the WHOLE slab (catches partial fills),
            # and the scan must agree with the oracle on those far-apart pages too
            planted_pages = {p for (_, _, p, _, _) in spec}
            if rsets:
                planted_pages |= {t[2] for st_ in rsets.values() for t in st_["spec"]}
            probe = [p for p in np.unique(np.linspace(0, n_local - 1, 24).astype(np.int64)).tolist() if (lo + p) not in planted_pages]
            gen_ok = True
            for p in probe:
                dev_page = ix.read_pages(p, 1)[0, : args.patches]
                cpu_page = orc.synth_rows(synth.SEED_CORPUS, lo + p, 0, args.patches)
                gen_ok = gen_ok and bool(np.array_equal(dev_page, cpu_page))
                w = orc.maxsim_float_np(orc.bf16_to_f32(q0), orc.bf16_to_f32(cpu_page)[None])[0]
                max_rel = max(max_rel, float(abs(full[p] - w) / max(abs(w), 1e-6)))
            if not gen_ok:
                sys.exit("bench.py: device-generated corpus differs from the oracle generator")
        if not args.no_cpu_baseline and args.workload == "binary":
            from oracle import oracle as orc  # checker / baseline only

            ns = min(args.cpu_sample_pages, n_local, 2048)
            pg = [orc.synth_rows(synth.SEED_CORPUS, lo + p, 0, args.patches) for p in range(ns)]  # unplanted sample
            bits = np.stack([orc.sign_pack(orc.bf16_to_f32(x)) for x in pg])
            qb = orc.sign_pack(orc.bf16_to_f32(queries[0]))
            cpu = cpu_baseline_binary(bits, qb)
            full = ix.score_all(queries[0], mode="binary")
            planted_pages = {p for (_, _, p, _, _) in spec}
            keep = [p for p in range(ns) if (lo + p) not in planted_pages]
            want = orc.maxsim_binary_popcount_np(bits[keep], qb)
            if not np.array_equal(full[keep].astype(np.float64), want):
                sys.exit("bench.py: sign-bit scores differ from the CPU restatement of SQL max_sim")
            max_rel = 0.0
        out = {
            "metric": {"float": "MaxSim pages scored/sec (exact top-10, 1 query of %d tokens per step)",
                       "fp8": "MaxSim pages scored/sec on the fp8 (e4m3) slab (exact top-10 of the quantised corpus, 1 query of %d tokens per step)",
                       "binary": "sign-bit max_sim pages scored/sec (SQL max_sim semantics, top-10, 1 query of %d tokens per step)",
                       "fde_fp8": "pages searched/sec: FDE coarse scan -> top-1000 -> exact fp8 MaxSim rerank -> top-10 (1 query of %d tokens per step)"}[args.workload]
            % args.qtokens,
            "value": round(value, 1),
            "unit": "pages/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,  # the reference publishes no number for this metric
            "speedup_vs_cpu_baseline": (round(value / cpu["value"], 1) if (cpu and cpu.get("value")) else None),
            "dtype": WL["dtype"],
            "data": "synthetic (on-device counter-based generator, L2-normalised bf16 rows, planted neighbours)",
            "config": {
                "workload": {"float": "BASELINE configs[2]: %d pre-embedded pages x %d patches x 128-d bf16, MaxSim-only, corpus row-sharded over %d GPU(s)",
                             "fp8": "BASELINE configs[4] shape: %d pre-embedded pages x %d patches x 128-d fp8 e4m3 (quantised from the bf16 corpus), MaxSim-only, row-sharded over %d GPU(s)",
                             "binary": "MultiVectorStore shape: %d pages x %d patches x BIT(128), sign-bit max_sim, row-sharded over %d GPU(s)",
                             "fde_fp8": "BASELINE configs[3] shard shape: %d pages x %d patches, FDE(10240) coarse top-1000 -> exact fp8 rerank, row-sharded over %d GPU(s)"}[args.workload]
                % (n_total, args.patches, world),
                "pages_total": n_total,
                "pages_per_gpu": n_local,
                "patches": args.patches,
                "dim": 128,
                "query_tokens": args.qtokens,
                "k": K,
                "requested_pages": args.pages,
                "parallelism": "row-shard x%d + all-gather top-k" % world,
                "collective_backend": (args.backend if dist_on else None),
                "rccl_ranks": (dist.get_world_size() if (dist_on and args.backend == "nccl") else 0),
                "local_scan_and_topk_ms_per_step": None if local_only_ms is None else round(local_only_ms, 4),
                # the exchange timed by itself (all-gather of k pairs per rank + merge, K back-to-back rounds); the difference of the two
                # loops beside it also carries the ranks' waiting for each other (large when they share one GPU)
                "collective_and_merge_ms_per_step": None if exchange_ms is None else round(exchange_ms, 4),
                "step_minus_local_ms_per_step": None if local_only_ms is None else round(ms_per_step - local_only_ms, 4),
            },
            "recall_at_10": recall10,
            "max_rel_score_err_vs_oracle": max_rel,
            "generator_matches_oracle": (True if max_rel is not None else None),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    aux = {}
    if out is not None and world == 1 and args.workload == "fde_fp8" and not args.no_aux:
        # the same shard serving 32 concurrent requests: mv_query_topk_batch, one FDE-slab pass for all of them
        bq = [queries[i % N_QUERIES] for i in range(32)]
        dev_ms, stages, res_b = [], [], None
        for r in range(8):
            res_b, st = ix.query_batch(bq, K, mode=MODE, want_stats=True)
            if r >= 3:
                dev_ms.append(st.total_device_ms)
                stages.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms))
        d, sb = float(np.median(dev_ms)), np.median(np.array(stages), axis=0)
        rb = [synth.recall_at_k([p for p in res_b[i][1].tolist() if p >= 0], [p for (qq, _r, p, _a, _b) in spec if qq == i % N_QUERIES]) for i in range(32)]
        aux["batched_32_requests"] = {
            "device_ms_per_batch": round(d, 4), "device_us_per_request": round(d * 1e3 / 32, 2), "requests_per_s": round(32 / d * 1e3, 1),
            "pages_searched_per_s": round(32 * n_local / d * 1e3, 1), "throughput_vs_one_request_per_step": round(32 * ms_per_step / d, 2),
            "stage_ms": {k: round(float(v), 4) for k, v in zip(("encode_32_queries", "coarse_gemm_one_slab_pass", "select", "rerank_fp8", "topk"), sb)},
            "coarse_pass_GBps": round(n_local * 20480 / float(sb[1]) / 1e6, 1), "recall_at_10": float(np.mean(rb)),
        }
    truths = gaps = None
    if out is not None and want_full:
        try:  # exact bf16 truth of every recall set + the batched MFMA scan, on the headline slab before it is freed
            t1 = time.time()
            truths, gaps = {}, {}
            truth_allow = first_pages_bitmap(n_truth, n_total)  # the pages the exact shard holds (doc ordinal == page)
            for name, st_ in rsets.items():
                truths[name], gaps[name] = exact_truth(ix, st_["queries"], allow=truth_allow)
            for j, top in enumerate(truths["hard_negatives"]):  # the near-tied set must hold the exact top-10
                assert set(top) <= set(synth.hard_pages_of(rsets["hard_negatives"]["spec"], j)), "hard set does not hold the exact top-10"
            for qi, t in enumerate(truths["planted"]):
                pl = [p for (qq, _r, p, _a, _b) in rsets["planted"]["spec"] if qq == qi]
                assert t[: len(pl)] == pl, "planted truth"
            aux["truth"] = {"source": "exact bf16 float scan over the first %d pages of the %d-page corpus (doc filter; batched form, 16 queries per slab pass), top-11" % (n_truth, n_total),
                            "seconds": round(time.time() - t1, 1),
                            "median_rel_gap_rank10_rank11": {k_: float(np.median(v)) for k_, v in gaps.items()}}
            aux["batched_float"] = batched_float_block(ix, queries, n_local, args, None)
            # the reference's queries run ~15-40 rows -- the default kernel at Q = 16 / 32 / 64 on the headline slab
            sweep = {}
            for qt in (16, 32, 64):
                qs_q = [synth_rows(synth.SEED_QUERIES, 1000 + qt + j, qt, device=local_rank) for j in range(4)]
                t = timed_mode(ix, qs_q, "float", n_timed=7, warm_s=0.15)
                sweep[f"Q{qt}"] = scan_entry(n_local, args.patches * PAGE_ROW_BYTES, t["score_kernel_ms"])
            aux["query_length_sweep"] = sweep
            for B in (4, 16):
                res_b = ix.query_batch(queries[:B], K)
                aux["batched_float"][f"B{B}"]["recall_at_10"] = float(np.mean([synth.recall_at_k(res_b[qi][1].tolist(), [p for (qq, _r, p, _a, _b) in spec if qq == qi])
                                                                               for qi in range(B)]))  # the headline's own planted top-10
        except Exception as e:  # noqa: BLE001 -- the headline number must survive a failure of the side measurements
            aux["truth_error"] = repr(e)
            truths = None
    ix.close()
    if out is not None and world == 1:
        # measured denominators, same process, GPU still warm from the timed run (the 262 GB slab had to go first): the scan's
        # own transport with the arithmetic removed, plain nt loads, register-only MFMA chains
        # (a launch must be as long as the scan's own: a 2 ms launch over 16 GiB loses ~2.5 % to ramp-up and tail)
        free_now, _tot = torch.cuda.mem_get_info(dev)
        cal_bytes = int(min(max(free_now - (12 << 30), 8 << 30), 160 << 30))
        calibrate("read_ldsdma", cal_bytes, 2, device=local_rank)
        measured_peak = calibrate("read_ldsdma", cal_bytes, 8, device=local_rank)
        measured_nt = calibrate("read_nt", min(cal_bytes, 64 << 30), 5, device=local_rank)
        measured_mfma = calibrate("mfma_bf16", 0, 5, device=local_rank)
        measured_mfma32 = calibrate("mfma_bf16_32x32", 0, 5, device=local_rank)
        log(f"[rank 0] calibration: nt LDS-DMA ring without arithmetic {measured_peak:.0f} GB/s (plain nt loads {measured_nt:.0f}); bf16 MFMA 16x16x32 {measured_mfma:.0f} / 32x32x16 {measured_mfma32:.0f} TFLOP/s")
        rf = out["roofline"]
        rf["measured_read_peak"] = round(measured_peak, 1)
        rf["frac_of_measured_peak"] = round(rf["achieved"] / measured_peak, 4)
        rf["measured_plain_nt_read"] = round(measured_nt, 1)
        rf["measured_mfma_bf16_tflops"] = round(measured_mfma, 1)
        rf["measured_mfma_bf16_32x32x16_tflops"] = round(measured_mfma32, 1)
        rf["measured_mfma_note"] = ("register-only MFMA chains, pseudo-random operands of embedding magnitude, ~4 ms launches (sustained clock): "
                                    "a proxy for what the matrix pipe sustains on this box, not a strict ceiling")
        rf["measured_read_peak_note"] = "a proxy (the scan's own transport without arithmetic), not a strict ceiling: boxes read 0.99-1.02 of it"
        for ent in aux.get("batched_float", {}).values():
            ent["frac_of_measured_mfma_16x16x32"] = round(ent["TFLOPs"] / measured_mfma, 4)
            ent["frac_of_measured_mfma_32x32x16"] = round(ent["TFLOPs"] / measured_mfma32, 4)
    if out is not None:
        # the record of the timed region is complete: keep it on disk before any side measurement runs
        write_record("bench_headline.json", split_headline(out)[0])
        if world == 1 and not args.no_aux and args.workload == "float":
            # ... and on stdout: if anything below took the process down, this line is the result (emit() prints the final record --
            # the same fields plus aux_summary -- as the LAST line)
            flush_c_stdio()
            print(json.dumps(fit_headline(*split_headline(out))), flush=True)
    if out is not None and world == 1 and not args.no_aux and args.workload == "float":
        # the secondary paths (configs[1] / [3] / [4] shapes, serving, encoder) run in a CHILD process with every slab of this one
        # freed: a crash, an out-of-memory kill or a hang there costs the aux record, never the headline line printed below
        state = {"n_total": n_total, "n_truth": n_truth, "n_exact": n_exact, "exact_budget": exact_budget,
                 "exact_split": bool(exact_split) if want_full and n_exact else False, "truths": truths, "gaps": gaps, "have_sets": rsets is not None}
        del rsets, queries
        torch.cuda.empty_cache()
        aux.update(run_aux_child(args, state))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
        flush_c_stdio()
        if out is not None and world > 1:
            time.sleep(1.5)  # the other ranks are tearing down too: whatever their runtimes still print reaches the shared stdout before the record
    if out is not None:
        emit(out, aux)


HEADLINE_ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "bytes_per_launch", "kernel_ms_avg",
                          "kernel_ms_per_rank", "launches_timed", "measured_read_peak", "frac_of_measured_peak")
HEADLINE_CPU_KEYS = ("value", "unit", "cores", "cores_available", "kind", "formulation", "sample")
HEADLINE_MAX_BYTES = 3000


def split_headline(out):
    """-> (headline, detail): the compact result record (every contract key, `roofline` and `cpu_baseline` cut to the fields a
    reader needs to recompute them) and what was cut.  The headline must stay under HEADLINE_MAX_BYTES serialised: round 4's
    one-line record grew to 23-28 KB and fell out of the driver's capture window."""
    head = {k: v for k, v in out.items() if k not in ("roofline", "cpu_baseline")}
    detail = {}
    for key, keep in (("roofline", HEADLINE_ROOFLINE_KEYS), ("cpu_baseline", HEADLINE_CPU_KEYS)):
        full = out.get(key)
        if isinstance(full, dict):
            head[key] = {k: full[k] for k in keep if k in full}
            rest = {k: v for k, v in full.items() if k not in keep}
            if rest:
                detail[key] = rest
        else:
            head[key] = full
    if len(json.dumps(head)) >= HEADLINE_MAX_BYTES and isinstance(head.get("cpu_baseline"), dict):  # the free-text fields go first
        detail.setdefault("cpu_baseline", {})["sample"] = head["cpu_baseline"].pop("sample", None)
    return head, detail


def out_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d


def write_record(name, doc):
    """Best effort: gpurun_out/ is scratch that travels back from the GPU box; a read-only tree must not cost the result line."""
    try:
        with open(os.path.join(out_dir(), name), "w") as f:
            json.dump(doc, f, indent=1)
        return os.path.join("gpurun_out", name)
    except OSError:
        return None


def emit(out, aux):
    """stdout of rank 0: (1) the detail record -- `bench_detail` + `aux_paths`, no `metric` key, any length -- then (2) the
    headline as the LAST line, < HEADLINE_MAX_BYTES.  Nothing JSON-shaped goes to stderr."""
    head, detail = split_headline(out)
    summ = aux_summary(out, aux) if aux else {}
    if summ:
        head["aux_summary"] = summ
    doc = {"bench_detail": detail, "aux_paths": aux}
    head["aux_file"] = os.path.join("gpurun_out", "bench_aux.json")
    head = fit_headline(head, detail)  # (moves what it cuts into `detail`, i.e. into doc)
    head["aux_file"] = write_record("bench_aux.json", doc)
    write_record("bench_headline.json", head)  # the line printed LAST (the fallback written before the aux child is overwritten)
    flush_c_stdio()
    if detail or aux:
        try:
            print(json.dumps(doc), flush=True)
        except OSError as e:  # a full disk behind a redirected stdout: the (short) headline below may still fit
            log(f"bench.py: could not print the detail record: {e!r}")
    print(json.dumps(head), flush=True)


# what leaves the headline, in this order, until it fits HEADLINE_MAX_BYTES (each lands in the detail record instead): free text first,
# then notes a reader can do without, then whole optional objects.  The contract keys, roofline.{bound,achieved,peak,unit,frac,traffic}
# and cpu_baseline.{value,unit,cores,kind} are never cut: a headline is ALWAYS printed (the old emit() exited without a record).
HEADLINE_CUT_ORDER = (("cpu_baseline", "sample"), ("roofline", "traffic_source"), ("cpu_baseline", "formulation"), ("roofline", "kernel"),
                      ("roofline", "kernel_ms_per_rank"), ("config", "collective_backend"), ("config", "requested_pages"), ("data", None),
                      ("config", "workload"), ("aux_summary", None))


def fit_headline(head, detail):
    for key, sub in HEADLINE_CUT_ORDER:
        if len(json.dumps(head)) < HEADLINE_MAX_BYTES:
            break
        if sub is None:
            if key in head and key not in ("metric", "value", "unit"):
                if key == "data":
                    detail["data"], head["data"] = head["data"], str(head["data"])[:40]
                else:
                    detail[key] = head.pop(key)
        elif isinstance(head.get(key), dict) and sub in head[key]:
            v = head[key].pop(sub)
            if key == "config" and sub == "workload":
                head[key][sub] = str(v)[:60]
            detail.setdefault(key, {})[sub] = v
    return head


def run_aux_child(args, state):
    """Re-execute this file with --aux-child <state file>: the child rebuilds the (deterministic) query / page sets, takes the
    exact truths from the state file and measures the secondary paths; its one stdout line is the aux dict."""
    import tempfile

    try:  # nothing on this path may cost the headline: an unwritable tree falls back to the temp directory, any other failure is recorded
        try:
            path = os.path.join(out_dir(), "bench_aux_state.json")
            with open(path, "w") as f:
                json.dump(state, f)
        except OSError:
            fd, path = tempfile.mkstemp(prefix="bench_aux_state_", suffix=".json")
            with os.fdopen(fd, "w") as f:
                json.dump(state, f)
    except Exception as e:  # noqa: BLE001
        return {"aux_child_error": f"could not write the state file: {e!r}"}
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--aux-child", path]
    t0 = time.time()
    def no_core():  # (and no host core file either)
        import resource

        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))

    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, timeout=args.aux_timeout, preexec_fn=no_core,
                           env=dict(os.environ, HSA_DISABLE_COREDUMP_ON_EXCEPTION="1"))
    except subprocess.TimeoutExpired:
        return {"aux_child_error": f"timed out after {args.aux_timeout} s"}
    except OSError as e:
        return {"aux_child_error": f"could not start the child: {e!r}"}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return {"aux_child_error": f"no record (exit code {p.returncode})"}
    try:
        res = json.loads(lines[-1])
    except ValueError as e:
        return {"aux_child_error": repr(e)}
    res["aux_child_seconds"] = round(time.time() - t0, 1)
    if p.returncode:
        res["aux_child_exit_code"] = p.returncode
    return res


def aux_child(args, local_rank):
    """The secondary paths of the default run (BASELINE configs[1] / [3] / [4] shapes, the plugin boundary under load), one
    try-block each; prints ONE JSON line (the aux dict)."""
    import torch

    from morphik_core_amd import synth
    from morphik_core_amd.index import synth_rows

    st = json.load(open(args.aux_child))
    stride = ((args.patches + 15) // 16) * 16
    n_total, n_truth, n_exact = st["n_total"], st["n_truth"], st["n_exact"]
    truths, gaps = st["truths"], st["gaps"]
    aux = {}
    rsets = spec = None
    if st["have_sets"]:
        queries = [synth_rows(synth.SEED_QUERIES, qi, args.qtokens, device=local_rank) for qi in range(N_QUERIES)]
        spec = synth.planted_spec(queries, n_total, args.patches, n_ranks=N_PLANTED)
        rsets = recall_sets(args, n_truth, local_rank, spec)
        del queries
    if rsets is not None and args.full_shard_pages > 0:
        try:  # BASELINE configs[3] / [4] at their per-GPU shard SIZE: the scan rates
            aux["full_shard"] = full_shard(args, local_rank, rsets["planted"]["queries"])
        except Exception as e:  # noqa: BLE001
            aux["full_shard"] = {"error": repr(e)}
    if truths is not None and n_exact > 0:
        try:  # ... and at their shard SHAPE with the exact tier in pinned host RAM: the exact pipelines and every recall figure
            aux["exact_shard"] = exact_shard(args, local_rank, n_exact, n_truth, st["exact_budget"], st["exact_split"], dict(rsets, _headline_spec=spec), truths, gaps)
        except Exception as e:  # noqa: BLE001
            aux["exact_shard"] = {"error": repr(e)}
    if args.aux_pages > 0:
        try:  # fp8 scan -> exact bf16 re-score from the exact tier (HBM / pinned host)
            aux["fp8_then_float"] = two_tier(args, local_rank)
        except Exception as e:  # noqa: BLE001
            aux["fp8_then_float"] = {"error": repr(e)}
        try:  # roofline entry of the FDE document encode
            aux["fde_document_encode"] = fde_encode_block(args, local_rank)
        except Exception as e:  # noqa: BLE001
            aux["fde_document_encode"] = {"error": repr(e)}
        try:  # the same path measured where the reference measures it: at the store's coroutine
            aux["serving"] = serving_block(args)
        except Exception as e:  # noqa: BLE001
            aux["serving"] = {"error": repr(e)}
    if args.ragged_pages > 0:
        try:  # a ColQwen-like ragged corpus on the fixed-stride and the packed layout
            aux["ragged_corpus"] = ragged_block(args, local_rank)
        except Exception as e:  # noqa: BLE001
            aux["ragged_corpus"] = {"error": repr(e)}
    if args.aux_pages > 0:
        try:  # the fp32-faithful tier: split-bf16 pages + query against a numpy fp32 restatement, and what its scans cost
            aux["fp32_split_bf16"] = fp32_split_block(args, local_rank)
        except Exception as e:  # noqa: BLE001
            aux["fp32_split_bf16"] = {"error": repr(e)}
    if args.aux_embed_pages > 0:
        try:  # configs[1] at full model size, short: encoder -> device ingest -> top-10
            r = embed_workload(args, args.aux_embed_pages, quick=True)
            aux["embed_colpali_v1_2"] = {k: r[k] for k in ("workload", "params", "rows_per_page", "embed_pages_per_s", "embed_model_only_pages_per_s",
                                                           "embed_tflops_est", "store_device_path_pages_per_s", "query_embed_ms_med",
                                                           "query_maxsim_top10_ms_med", "model_batch", "chunks_per_call", "fused_encoder_ops", "tuned_gemm_selections", "dtype", "data")}
        except Exception as e:  # noqa: BLE001
            aux["embed_colpali_v1_2"] = {"error": repr(e)}
    torch.cuda.synchronize()
    print(json.dumps(aux), flush=True)

if __name__ == "__main__":
    main()
