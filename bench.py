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

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
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

PAGE_ROW_BYTES = 256  # 128 x bf16
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 (same guide)
N_QUERIES = 16
N_PLANTED = 10
K = 10


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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


def cpu_baseline(sample_pages_u16, q_u16):
    """Reference CPU path timed on this box's host cores, on a bounded sample of the same workload (BASELINE.md section
    3: >= 20 000 pages, >= 5 repeats, median).  Two formulations of the reference's float MaxSim
    (fast_multivector_store.py:553-555 -> score_multi_vector): (i) numpy sgemm -> max -> sum over all cores,
    (ii) torch einsum over page batches of 128 (the reference's own expression; ~20x slower, smaller sample).
    fp32 on upcast bf16 data (the reference upcasts at load, fast_multivector_store.py:736,774).  The faster one is the
    baseline of record."""
    from oracle import oracle as orc  # CPU checker / baseline only

    import torch

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    q = orc.bf16_to_f32(q_u16)
    pages = orc.bf16_to_f32(sample_pages_u16)  # upcast outside the timed region, like the reference's load step
    n = pages.shape[0]
    n_torch = min(n, 2048)
    res, used = {}, {}
    for name, fn, m, budget in (("numpy_sgemm", lambda: orc.maxsim_float_np(q, pages), n, 14.0),
                                ("torch_einsum", lambda: orc.maxsim_float_torch(q, pages[:n_torch]), n_torch, 8.0)):
        times = timed_runs(fn, min_runs=5, budget_s=budget)
        res[name] = m / float(np.median(times))
        used[name] = (m, len(times))
    best = max(res, key=res.get)
    page_bytes_f32 = pages.shape[1] * 128 * 4
    return {
        "value": round(res[best], 1),
        "unit": "pages/s",
        "cores": cores,
        "kind": "port",
        "achieved_GBps_fp32_pages": round(res[best] * page_bytes_f32 / 1e9, 2),
        "sample": f"{pages.shape[1]} patches x 128-d fp32 (upcast bf16), Q={q.shape[0]}; numpy_sgemm={res['numpy_sgemm']:.0f} pages/s on {used['numpy_sgemm'][0]} pages "
                  f"(median of {used['numpy_sgemm'][1]}), torch_einsum={res['torch_einsum']:.0f} pages/s on {used['torch_einsum'][0]} pages "
                  f"(median of {used['torch_einsum'][1]}); best={best}",
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


def pmc_traffic(n_local, patches):
    """HBM traffic of the scan kernel from rocprofv3 PMC counters (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 fetch
    correction calibrated on a known byte count in the same pass), collected by tools/pmc_traffic.sh and committed under
    profiles/.  Only a record taken on THIS library build counts: the file names the kernel symbol and the sha256 of the
    libmvmaxsim.so it profiled; anything else -> null (a kernel change must not inherit an old measurement)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "pmc_traffic*.json"), recursive=True), key=os.path.getmtime)
    sha = lib_sha256()
    for f in reversed(files):
        try:
            rec = json.load(open(f))
            if rec.get("lib_sha256") == sha and "maxsim_ldsdma_kernel" in rec.get("kernel", ""):
                return int(round(float(rec["hbm_bytes_per_page"]) * n_local * patches / 1024.0)), os.path.relpath(f, ROOT), rec.get("kernel")
        except Exception:  # noqa: BLE001
            continue
    return None, None, None


def recall_block(ix, qs, truths, K, coarse_ns=(75, 1000)):
    """recall@10 against the exact bf16 top-10 for every lossy path of an all-slab index, + FDE coarse recall."""
    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth

    out = {}
    for mode in ("float_fp8", "binary"):
        out[mode] = float(np.mean([synth.recall_at_k(ix.query(q, K, mode=mode)[1].tolist(), t) for q, t in zip(qs, truths)]))
    for cn in coarse_ns:
        ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
        out[f"fde_top{cn}_then_float"] = float(np.mean([synth.recall_at_k(ix.query(q, K, mode="fde_then_float")[1].tolist(), t) for q, t in zip(qs, truths)]))
    out["fde_coarse_recall_at_1000"] = float(np.mean([synth.recall_at_k(ix.query(q, 1000, mode="fde")[1].tolist(), t) for q, t in zip(qs, truths)]))
    return out


def aux_paths(args, device, mfma_peak=None):
    """Quick, separately sized measurements of the other hot-path kernels (same HIP-event method, a smaller corpus with
    EVERY slab): sign-bit MaxSim (SQL max_sim semantics), fp8 slab, FDE coarse scan, FDE -> rerank, the batched-query MFMA
    form, and recall@10 of the lossy paths against the exact bf16 top-10 on hard negatives and on a corpus with no
    planted structure at all.  Reported next to the headline number, never mixed into `value`.  None of them launches the
    headline scan kernel inside a timed region of the main workload."""
    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex, synth_rows

    n = args.aux_pages
    stride = ((args.patches + 15) // 16) * 16
    NH, NR = 8, 4  # hard-negative queries, unplanted queries
    qs = [synth_rows(synth.SEED_QUERIES, qi, args.qtokens, device=device) for qi in range(N_QUERIES + NH + NR)]
    spec = synth.planted_spec(qs[:N_QUERIES], n, args.patches, n_ranks=N_PLANTED)
    planted = {qi: [p for (qq, _r, p, _a, _b) in spec if qq == qi] for qi in range(N_QUERIES)}
    taken = {p for (_q, _r, p, _a, _b) in spec}
    hq = qs[N_QUERIES : N_QUERIES + NH]
    hspec = [t for t in synth.hard_spec(hq, n, args.patches) if t[2] not in taken]
    res = {"pages": n, "note": "kernel-only HIP-event times, median of 15 after 0.25 s of warm-up queries, on one index holding bf16 + e4m3 + sign-bit + FDE slabs"}
    ix = MvIndex(capacity_pages=n, stride_rows=stride, device=device, with_float=True, with_binary=True, with_fde=True, with_fp8=True)
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=args.patches)
    synth.plant_neighbours_any(ix, spec, synth.SEED_CORPUS, args.patches)
    synth.plant_neighbours_any(ix, hspec, synth.SEED_CORPUS, args.patches)
    per_page = {"binary": args.patches * 16, "float_fp8": args.patches * 128, "fde": 10240 * 2}
    WARM, TIMED = 10, 15

    def warm(mode, seconds=0.25):  # the clocks need ~100 ms of load to settle after an idle spell: warm up by TIME, not by count
        t_end = time.perf_counter() + seconds
        i = 0
        while time.perf_counter() < t_end:
            ix.query(qs[i % N_QUERIES], K, mode=mode)
            i += 1

    for mode in ("binary", "float_fp8", "fde"):
        ms, coarse = [], []
        warm(mode)
        for r in range(TIMED):
            _s, _i, st = ix.query(qs[r % N_QUERIES], K, mode=mode, want_stats=True)
            ms.append(st.score_kernel_ms)
            coarse.append(st.coarse_ms)
        m = float(np.median(coarse)) if mode == "fde" else float(np.median(ms))  # FDE: the slab scan alone (the query encode is its own stage)
        ent = {"kernel_ms": round(m, 4), "pages_per_s": round(n / m * 1e3, 1), "GBps": round(n * per_page[mode] / m / 1e6, 1),
               "frac_hbm_8TBps": round(n * per_page[mode] / m / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_page": per_page[mode]}
        if mode == "fde":
            ent["span_with_query_encode_ms"] = round(float(np.median(ms)), 4)
        res[mode] = ent
    # FDE coarse top-1000 -> exact rerank (configs[3] pipeline), all in stream order on the device
    ix.set_option(L.MV_OPT_FDE_COARSE_N, 1000)
    ms, stg = [], []
    warm("fde_then_float")
    for r in range(WARM + N_QUERIES):
        _s, ids, st = ix.query(qs[r % N_QUERIES], K, mode="fde_then_float", want_stats=True)
        if r >= WARM:
            ms.append(st.total_device_ms)
            stg.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms))
    stg = np.median(np.array(stg), axis=0)
    res["fde_top1000_then_float"] = {"device_ms": round(float(np.median(ms)), 4), "pages_per_s": round(n / float(np.median(ms)) * 1e3, 1),
                                     "stage_ms": {k: round(float(v), 4) for k, v in zip(("encode_query", "coarse_scan", "select_top1000", "rerank_1000", "topk"), stg)},
                                     "overhead_over_coarse_scan_ms": round(float(np.median(ms)) - float(stg[1]), 4)}
    # ---- the same pipeline for a BATCH of requests (mv_query_topk_batch): one FDE-slab pass per 32 queries
    bq = [qs[i % N_QUERIES] for i in range(32)]
    res["fde_batched_32_queries"] = {}
    for cn, key in ((1000, "coarse1000_then_float"), (75, "coarse75_then_float_reference_rule")):
        ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
        one = []
        for r in range(12):
            _s, _i, st = ix.query(qs[r % N_QUERIES], K, mode="fde_then_float", want_stats=True)
            if r >= 4:
                one.append(st.total_device_ms)
        dev, stg_b, out_b = [], [], None
        for r in range(9):
            out_b, st = ix.query_batch(bq, K, mode="fde_then_float", want_stats=True)
            if r >= 3:
                dev.append(st.total_device_ms)
                stg_b.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms))
        d, sb = float(np.median(dev)), np.median(np.array(stg_b), axis=0)
        res["fde_batched_32_queries"][key] = {
            "device_ms_per_batch": round(d, 4), "device_us_per_query": round(d * 1e3 / 32, 2), "queries_per_s": round(32 / d * 1e3, 1),
            "single_query_device_us": round(float(np.median(one)) * 1e3, 2), "throughput_vs_query_by_query": round(float(np.median(one)) * 32 / d, 2),
            "stage_ms": {k: round(float(v), 4) for k, v in zip(("encode_32_queries", "coarse_gemm_one_slab_pass", "select", "rerank", "topk"), sb)},
            "coarse_pass_GBps": round(n * per_page["fde"] / float(sb[1]) / 1e6, 1),
            "recall_at_10": float(np.mean([synth.recall_at_k(out_b[i][1].tolist(), planted[i % N_QUERIES]) for i in range(32)])),
            "same_ids_as_single_query": float(np.mean([out_b[i][1].tolist() == ix.query(bq[i], K, mode="fde_then_float")[1].tolist() for i in range(0, 32, 4)])),
        }
    ix.set_option(L.MV_OPT_FDE_COARSE_N, 1000)
    # ---- recall@10 of the lossy paths vs the exact bf16 top-10
    easy = recall_block(ix, qs[:N_QUERIES], [planted[qi] for qi in range(N_QUERIES)], K)
    truths, hardness = [], []
    for j, q in enumerate(hq):
        pages = synth.hard_pages_of(hspec, j)
        exact = ix.score_candidates(q, pages, pad_to=0)  # exact bf16 MaxSim (parity-checked kernel) of the hard set
        top, info = synth.exact_truth_from_scores(pages, exact, K)
        truths.append(top)
        hardness.append(info)
        full = ix.query(q, K, mode="float")[1].tolist()
        assert full == top, "the hard set must hold the exact top-10"
    hard = recall_block(ix, hq, truths, K)
    rq = qs[N_QUERIES + NH :]
    rtruth = [ix.query(q, K, mode="float")[1].tolist() for q in rq]
    rnd = recall_block(ix, rq, rtruth, K)
    res["recall_at_10_vs_exact_bf16"] = {
        "planted_3x_margin": easy,
        "hard_negatives": dict(hard, queries=NH, pages_per_query=synth.N_HARD,
                               median_rel_gap_rank10_rank11=float(np.median([h["gap_10_11"] for h in hardness])),
                               min_distractors_within_2pct_of_rank10=int(min(h["within_2pct"] for h in hardness))),
        "unplanted_random_corpus": dict(rnd, queries=NR, note="no planted structure: the top-10 of 200 k random pages are separated by ~1e-3 relative"),
    }
    # ---- batched form (B x 32 tokens per slab pass)
    res["batched_float"] = {}
    for B in (4, 16):
        ms = []
        for r in range(7):
            out, st = ix.query_batch(qs[:B], K, want_stats=True)
            if r >= 2:
                ms.append(st.score_kernel_ms)
        m = float(np.median(ms))
        tf = 2.0 * B * args.qtokens * args.patches * 128 * n / m / 1e9
        res["batched_float"][f"B{B}"] = {"kernel_ms": round(m, 4), "query_pages_per_s": round(B * n / m * 1e3, 1), "TFLOPs": round(tf, 1),
                                         "frac_mfma_bf16_2500TF": round(tf / MFMA_BF16_PEAK_TF, 4),
                                         "frac_of_measured_mfma_peak": None if not mfma_peak else round(tf / mfma_peak, 4),
                                         "GBps": round(n * args.patches * 256 / m / 1e6, 1),
                                         "recall_at_10": float(np.mean([synth.recall_at_k(out[qi][1].tolist(), planted[qi]) for qi in range(B)]))}
    ix.close()
    return res


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
    if lines:
        print(lines[-1], flush=True)
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
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--workload", choices=["float", "fp8", "binary", "fde_fp8", "embed"], default="float",
                    help="float = BASELINE configs[2] (the headline, default); fp8 = e4m3 slab (configs[4]); binary = sign-bit "
                         "max_sim (MultiVectorStore); fde_fp8 = FDE coarse top-1000 -> exact fp8 rerank (configs[3] shard shape); "
                         "embed = configs[1] (ColPali-v1.2 architecture, 1 k pages -> top-10)")
    ap.add_argument("--no-aux", action="store_true", help="skip the secondary kernels' quick measurements (aux_paths)")
    ap.add_argument("--aux-pages", type=int, default=200_000)
    ap.add_argument("--aux-embed-pages", type=int, default=96, help="pages of the full-size encoder run inside aux_paths (0 = skip)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend for N>1 (nccl == RCCL; gloo + MV_BENCH_SINGLE_DEVICE=1 lets N ranks share one GPU to "
                         "exercise the multi-rank path on a 1-GPU box -- a functional check, not a measurement)")
    args = ap.parse_args()

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
        free_b = free_b // world  # the ranks share one GPU
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
    MODE = WL["mode"]
    torch.cuda.synchronize()
    log(f"[rank {rank}] corpus generated + planted in {time.time()-t0:.1f}s")

    stats = []
    gpu_topk = sharded.make_gpu_local_topk(ix, dev, MODE, collect_stats=stats)
    if args.backend == "nccl":
        local_topk = gpu_topk
    else:
        def local_topk(q, k):  # gloo: the k (score, id) pairs go through host memory
            s, i = gpu_topk(q, k)
            return s.cpu(), i.cpu()
    searcher = sharded.ShardedSearcher(local_topk)
    fast_searcher = sharded.GpuShardedSearcher(ix, dev, MODE, collect_stats=stats) if (dist_on and args.backend == "nccl") else None
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
        if fast_searcher is not None:  # RCCL: 2 collectives + one library merge launch, nothing else on the host
            return fast_searcher.query(q, K)
        s, ids = searcher.query(q, K, compact=False)  # padded (-inf, -1) tail: no host sync inside the timed loop
        return s, ids

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    stats.clear()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed

    # ---- roofline of the dominant kernel (the page scan), from HIP events recorded in the timed region
    kms = np.array([s.score_kernel_ms for s in stats if s is not None and s.score_kernel_ms > 0])
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
                       "fp8": "maxsim_fp8_kernel (e4m3 page scan, MX-scaled MFMA)", "binary": "maxsim_binary_mfma2_kernel (sign-bit scan, FP4 MFMA)",
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
            "bytes_per_launch": bytes_per_launch,
            "kernel_ms_avg": round(k_ms, 4),
            "kernel_ms_per_rank": [round(x, 4) for x in per_rank_kms],
            "launches_timed": int(kms.size),
            "mfma_tflops_achieved": round(2.0 * args.qtokens * args.patches * 128 * n_local / (k_ms * 1e-3) / 1e12, 2) if k_ms > 0 else 0.0,
        }
        cpu = None
        max_rel = None
        if world == 1 and not args.no_cpu_baseline and args.workload == "float":
            from oracle import oracle as orc  # checker / baseline only

            ns = min(args.cpu_sample_pages, n_local)
            sample = ix.read_pages(0, ns)[:, : args.patches]
            q0 = queries[0]
            cpu = cpu_baseline(sample, q0)
            want = orc.maxsim_float_np(orc.bf16_to_f32(q0), orc.bf16_to_f32(sample))
            full = ix.score_all(q0)
            max_rel = float(np.max(np.abs(full[:ns] - want) / np.maximum(np.abs(want), 1e-6)))
            # the device generator must equal the oracle's across the WHOLE slab (catches partial fills),
            # and the scan must agree with the oracle on those far-apart pages too
            planted_pages = {p for (_, _, p, _, _) in spec}
            probe = [p for p in np.unique(np.linspace(0, n_local - 1, 24).astype(np.int64)).tolist() if p not in planted_pages]
            gen_ok = True
            for p in probe:
                dev_page = ix.read_pages(p, 1)[0, : args.patches]
                cpu_page = orc.synth_rows(synth.SEED_CORPUS, lo + p, 0, args.patches)
                gen_ok = gen_ok and bool(np.array_equal(dev_page, cpu_page))
                w = orc.maxsim_float_np(orc.bf16_to_f32(q0), orc.bf16_to_f32(cpu_page)[None])[0]
                max_rel = max(max_rel, float(abs(full[p] - w) / max(abs(w), 1e-6)))
            if not gen_ok:
                sys.exit("bench.py: device-generated corpus differs from the oracle generator")
        if world == 1 and not args.no_cpu_baseline and args.workload == "binary":
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
            "vs_baseline": None,
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
            },
            "recall_at_10": recall10,
            "max_rel_score_err_vs_oracle": max_rel,
            "generator_matches_oracle": (True if max_rel is not None else None),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
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
        out["batched_32_requests"] = {
            "device_ms_per_batch": round(d, 4), "device_us_per_request": round(d * 1e3 / 32, 2), "requests_per_s": round(32 / d * 1e3, 1),
            "pages_searched_per_s": round(32 * n_local / d * 1e3, 1), "throughput_vs_one_request_per_step": round(32 * ms_per_step / d, 2),
            "stage_ms": {k: round(float(v), 4) for k, v in zip(("encode_32_queries", "coarse_gemm_one_slab_pass", "select", "rerank_fp8", "topk"), sb)},
            "coarse_pass_GBps": round(n_local * 20480 / float(sb[1]) / 1e6, 1), "recall_at_10": float(np.mean(rb)),
        }
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
        log(f"[rank 0] calibration: nt LDS-DMA ring without arithmetic {measured_peak:.0f} GB/s (plain nt loads {measured_nt:.0f}); bf16 MFMA {measured_mfma:.0f} TFLOP/s")
        rf = out["roofline"]
        rf["measured_read_peak"] = round(measured_peak, 1)
        rf["frac_of_measured_peak"] = round(rf["achieved"] / measured_peak, 4)
        rf["measured_plain_nt_read"] = round(measured_nt, 1)
        rf["measured_mfma_bf16_tflops"] = round(measured_mfma, 1)
    if out is not None and world == 1 and not args.no_aux:
        try:
            out["aux_paths"] = aux_paths(args, local_rank, measured_mfma)
        except Exception as e:  # the headline number must survive a failure of the side measurements
            out["aux_paths"] = {"error": repr(e)}
        if args.aux_embed_pages > 0:
            try:  # configs[1] at full model size, short: encoder -> device ingest -> top-10
                r = embed_workload(args, args.aux_embed_pages, quick=True)
                out["aux_paths"]["embed_colpali_v1_2"] = {k: r[k] for k in ("workload", "params", "rows_per_page", "embed_pages_per_s", "embed_model_only_pages_per_s",
                                                                             "embed_tflops_est", "store_device_path_pages_per_s", "query_embed_ms_med",
                                                                             "query_maxsim_top10_ms_med", "model_batch", "dtype", "data")}
            except Exception as e:  # noqa: BLE001
                out["aux_paths"]["embed_colpali_v1_2"] = {"error": repr(e)}
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
