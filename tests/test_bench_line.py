"""bench.py's result line: the LAST stdout line, compact (< 3 KB), complete.

Round 4's record grew to 23-28 KB (aux_paths inside the one line) and the driver parsed no result line at all.  The CPU test
cuts a recorded round-4 line with the bench's own split_headline(); the GPU test runs the default-aux path (child process and
all) at a reduced size and checks what the driver's parser will see."""
import importlib.util
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _check_headline(line):
    assert len(line) < 3000, len(line)
    d = json.loads(line)
    for k in CONTRACT_KEYS:
        assert k in d, k
    rf, cpu = d["roofline"], d["cpu_baseline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert "workload" in d["config"] and "model" not in d["config"] and "aux_paths" not in d
    return d


def test_recorded_round4_line_cut_to_a_compact_headline():
    b = _bench_module()
    rec = json.load(open(os.path.join(ROOT, "profiles", "r4", "bench_1gpu_1M_pages_r4f.json")))
    assert len(json.dumps(rec)) > 20000  # what the driver could not parse
    aux = rec.pop("aux_paths")
    head, detail = b.split_headline(rec)
    d = _check_headline(json.dumps(head))
    assert d["value"] == rec["value"] and d["roofline"]["frac"] == rec["roofline"]["frac"] and d["cpu_baseline"]["value"] == rec["cpu_baseline"]["value"]
    # nothing is lost: what the headline drops is in the detail record
    assert set(detail["roofline"]) | set(head["roofline"]) == set(rec["roofline"])
    assert set(detail["cpu_baseline"]) | set(head["cpu_baseline"]) == set(rec["cpu_baseline"])
    assert aux  # (goes to the earlier stdout line / gpurun_out/bench_aux.json)


def test_emit_prints_detail_first_and_headline_last(capsys, tmp_path, monkeypatch):
    b = _bench_module()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    rec = json.load(open(os.path.join(ROOT, "profiles", "r4", "bench_1gpu_1M_pages_r4f.json")))
    aux = rec.pop("aux_paths")
    b.emit(rec, aux)
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 2 and not any(ln.startswith("{") for ln in cap.err.splitlines())
    first = json.loads(lines[0])
    assert "metric" not in first and first["aux_paths"] == aux and "roofline" in first["bench_detail"]
    d = _check_headline(lines[1])
    assert d["aux_file"] == os.path.join("gpurun_out", "bench_aux.json")
    assert json.load(open(tmp_path / "gpurun_out" / "bench_aux.json"))["aux_paths"] == aux
    # the secondary kernels' figures ride in the headline: every value is the aux record's own
    summ = d["aux_summary"]
    assert summ and len(json.dumps(summ)) <= 900
    if "full_shard" in aux and "fp8_scan" in aux["full_shard"]:
        assert summ["fp8_scan_frac"] == aux["full_shard"]["fp8_scan"]["frac_hbm_8TBps"]


def test_emit_always_prints_a_headline_even_when_fields_must_be_cut(capsys, tmp_path, monkeypatch):
    """emit() used to sys.exit when the line reached the limit -- after the whole measurement, leaving NO record.  Now
    free-text and optional fields move to the detail record step by step; the contract keys and the roofline / cpu_baseline
    numbers always print."""
    b = _bench_module()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    rec = json.load(open(os.path.join(ROOT, "profiles", "r4", "bench_1gpu_1M_pages_r4f.json")))
    aux = rec.pop("aux_paths")
    rec["data"] = "synthetic " + "x" * 1500
    rec["config"]["workload"] = "BASELINE configs[2] " + "y" * 1500
    rec["cpu_baseline"]["sample"] = "z" * 1500
    b.emit(rec, aux)
    lines = capsys.readouterr().out.splitlines()
    assert len(lines) == 2 and len(lines[1]) < 3000
    d = json.loads(lines[1])
    for k in CONTRACT_KEYS:
        assert k in d, k
    assert d["value"] == rec["value"] and d["roofline"]["frac"] == rec["roofline"]["frac"] and d["cpu_baseline"]["value"] == rec["cpu_baseline"]["value"]
    assert d["config"]["workload"].startswith("BASELINE configs[2]") and d["data"].startswith("synthetic")
    detail = json.loads(lines[0])["bench_detail"]
    assert detail["cpu_baseline"]["sample"] == "z" * 1500 and detail["config"]["workload"].endswith("y" * 100)


def test_serving_progress_lines_are_not_json():
    src = open(os.path.join(ROOT, "tools", "serve_bench.py")).read()
    assert "print(json.dumps(r), file=sys.stderr" not in src


@pytest.mark.gpu
def test_default_aux_run_last_line_is_the_compact_headline():
    """The driver's command shape (`python bench.py --steps K --warmup W`, aux ON) at a reduced corpus: the last stdout line is
    the headline, short, with roofline.frac and cpu_baseline.value; the timed region fits the wall clock of the run; the aux
    record is an earlier line and a file."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--pages", "50000", "--cpu-sample-pages", "512", "--cpu-baseline-quick",
           "--full-shard-pages", "50000", "--exact-shard-pages", "50000", "--aux-pages", "8000", "--aux-embed-pages", "0", "--aux-serve-seconds", "0.15", "--ragged-pages", "6000"]
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200, cwd=ROOT)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert not any(ln.startswith("{") for ln in p.stderr.splitlines()), "JSON-shaped line on stderr"
    d = _check_headline(out_lines[-1])
    assert len(out_lines[-1]) < 4096
    assert d["steps"] == 8 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["recall_at_10"] == 1.0
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["vs_baseline"] is None
    assert d["ms_per_step"] * d["steps"] / 1e3 <= wall
    assert abs(d["value"] - d["config"]["pages_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    earlier = [json.loads(ln) for ln in out_lines[:-1] if ln.startswith("{")]
    # (1) the crash-fallback headline printed BEFORE the aux child started (same timed-region fields, no aux_summary), (2) the detail record
    assert len(earlier) == 2 and earlier[0]["value"] == d["value"] and "aux_summary" not in earlier[0] and "metric" not in earlier[1]
    _check_headline(json.dumps(earlier[0]))
    aux = earlier[1]["aux_paths"]
    assert "aux_child_error" not in aux, aux
    for key in ("truth", "batched_float", "query_length_sweep", "full_shard", "exact_shard", "fp8_then_float", "fde_document_encode", "serving", "fp32_split_bf16", "ragged_corpus"):
        assert key in aux and "error" not in aux[key], (key, aux.get(key))
    # the secondary kernels' figures travel in the driver-parsed line itself
    summ = d["aux_summary"]
    for key in ("fp8_scan_frac", "sign_bit_frac", "fde_scan_frac", "fde_batch32_frac", "batched_bf16_B16_PF", "fde_request_ms", "fp8_recall_hard",
                "fp8_then_float_recall", "fp32_split_max_rel_err", "fp32_hi_lo_scan_frac"):
        assert key in summ, (key, summ)
    assert len(json.dumps(summ)) <= 900 and summ["fp32_split_max_rel_err"] < 1e-4
    rg = aux["ragged_corpus"]
    assert rg["packed"]["same_top10_as_fixed_stride"] and rg["capacity_gain_packed_over_fixed"] > 1.2 and summ["ragged_packed_valid_frac"] == rg["packed"]["frac_hbm_8TBps_valid_bytes"]
    assert set(aux["query_length_sweep"]) == {"Q16", "Q32", "Q64"}
    assert summ["fp8_scan_frac"] == aux["full_shard"]["fp8_scan"]["frac_hbm_8TBps"]
    assert json.load(open(os.path.join(ROOT, "gpurun_out", "bench_aux.json")))["aux_paths"].keys() == aux.keys()


@pytest.mark.gpu
def test_rccl_rank_under_the_drivers_launcher_keeps_the_headline_last():
    """The driver's N > 1 command shape (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`) with the one rank this box has and a real RCCL communicator (MV_BENCH_FORCE_DIST=1).  RCCL
    prints a version banner through C stdio, which a pipe block-buffers until the process exits: in round 5 `Librccl path : ...` was the
    last stdout line of such a run.  bench.py flushes the C streams once the communicator is up and again before it prints: the LAST
    non-empty stdout line must be the headline, with the exchange timed by itself."""
    import socket

    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, MV_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--pages", "60000", "--cpu-sample-pages", "512",
           "--cpu-baseline-quick", "--no-aux"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    d = _check_headline(out_lines[-1])
    cfg = d["config"]
    assert cfg["collective_backend"] == "nccl" and cfg["rccl_ranks"] == 1 and d["recall_at_10"] == 1.0
    assert 0 < cfg["collective_and_merge_ms_per_step"] < 1.0 and d["roofline"]["launches_timed"] == 10


def test_a_pmc_traffic_record_exists_for_the_sources_in_the_tree():
    """`roofline.traffic` is only reported from a PMC record taken on THIS library (bench.pmc_traffic matches the record's
    `src_sha256` / `lib_sha256`).  A kernel edit after the last `tools/measure.sh pmc_traffic` silently turns the figure into null
    (it happened at the end of round 6): this test names the state.  Skipped, not failed, on a mismatch -- an edited kernel is a
    legitimate state of the tree until it is re-measured -- so the skip reason is the to-do."""
    b = _bench_module()
    traffic, src, kernel = b.pmc_traffic(1_000_000, 1024)
    if traffic is None:
        pytest.skip("no profiles/**/pmc_traffic*.json matches src_sha256 %s...: re-run `tools/measure.sh pmc_traffic` on the MI355X and commit "
                    "gpurun_out/<round>/pmc_traffic_<round>.json under profiles/<round>/" % b.src_sha256()[:12])
    assert src.startswith("profiles/") and "maxsim_ldsdma_kernel" in kernel
    assert 1.0 <= traffic / (1_000_000 * 1024 * 256) < 1.01  # HBM bytes per launch against the algorithmic 262 144 B per page
