#!/bin/bash
# tools/measure.sh <profile> [args...] -- ONE entry point for every measurement whose record lives under profiles/rN/ (run on the
# MI355X box:  gpurun --timeout T -- 'bash tools/measure.sh <profile>').  Outputs land in gpurun_out/<round>/ under the NAME they
# are committed with in profiles/<round>/ (copy them there; gpurun_out/ is scratch).  ROUND defaults to r6.
#
#   suite            smoke() + the whole `pytest -m gpu` suite                       -> pytest_gpu_full.log, smoke.log
#   bench            the driver's command (python bench.py --steps 20 --warmup 5)    -> bench_1gpu_1M_pages_{stdout,stderr}.txt, bench_headline.json, bench_detail_and_aux.json
#   bench_rocprof    the same (no aux, no CPU leg) under rocprofv3 --kernel-trace --stats -> rocprofv3_kernel_stats_bench_1M.csv, bench_under_rocprofv3.json
#   pmc_traffic      FETCH_SIZE / WRITE_SIZE passes (separate runs) of the headline kernel -> pmc_traffic_<round>.json (hash-tied to the library; also copied to profiles/<round>/)
#   pmc_traffic_aux [pages]  the same two passes over the secondary scans (e4m3, sign-bit, FDE)   -> pmc_traffic_aux_scans_<round>.json
#   batch_sq [pages] SQ counters (two counter-only passes) of the batched bf16 scan at B = 16 -> pmc_sq_batched_bf16_B16.json, batch_scan_probe.jsonl
#   fde_batch_sq [pages]  SQ / TCC counters of the batched FDE coarse pass, one process per placement -> pmc_fde_batch_modes.json (tools/fde_batch_mode_probe.py)
#   fde_e4m3_sq [pages]  SQ counters of the FDE scans on both slabs (single-query and batched)  -> pmc_sq_fde_scans_bf16_and_e4m3.json
#   binary_sq [pages] SQ counters of the sign-bit scan                                -> pmc_sq_sign_bit_scan.json
#   probe <script.py> [args...]   any tools/*.py probe, stdout -> <script>.jsonl
#   pmc <name> <counters...> -- <cmd...>   a counter-only rocprofv3 pass of any command -> pmc_<name>.json
set -u
ROUND=${ROUND:-r6}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
what=${1:-suite}; shift || true
# the library travels prebuilt with the snapshot; an incremental make (hipcc is in the image) guarantees it matches the sources that came with it
(cd $R && python -c "import morphik_core_amd as m; m.build_library()" > $OUT/build.log 2>&1) || { echo "build failed"; tail -20 $OUT/build.log; exit 3; }

pmc_pass() {  # pmc_pass <name> "<counters>" <cmd...>
  local name=$1 set=$2; shift 2
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 900 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$name -- "$@" > $OUT/pmc_$name.log 2>&1)
  python $R/tools/rocprof_summary.py /tmp/pmc_$name $OUT/pmc_$name.raw.json > /dev/null 2>&1
}
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"
SQ2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"

case $what in
  suite)
    (cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log)
    t0=$(date +%s)
    (cd $R && timeout 1500 python -m pytest tests -q -m gpu --durations=8 "$@" > $OUT/pytest_gpu_full.log 2>&1)
    echo "gpu suite rc=$? wall=$(( $(date +%s) - t0 )) s"
    grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^$" $OUT/pytest_gpu_full.log | tail -14 ;;
  bench)
    t0=$(date +%s)
    (cd $R && timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $OUT/bench_1gpu_1M_pages_stdout.txt 2> $OUT/bench_1gpu_1M_pages_stderr.txt)
    echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s; stdout line lengths: $(awk '{printf "%d ", length($0)}' $OUT/bench_1gpu_1M_pages_stdout.txt); JSON-shaped stderr lines: $(grep -c '^{' $OUT/bench_1gpu_1M_pages_stderr.txt)"
    cp $R/gpurun_out/bench_headline.json $OUT/bench_headline.json 2>/dev/null; cp $R/gpurun_out/bench_aux.json $OUT/bench_detail_and_aux.json 2>/dev/null
    tail -n 1 $OUT/bench_1gpu_1M_pages_stdout.txt; tail -n 25 $OUT/bench_1gpu_1M_pages_stderr.txt ;;
  bench_rocprof)
    rm -rf /tmp/tr_bench
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_bench -- python $R/bench.py --steps 20 --warmup 5 --no-aux --no-cpu-baseline "$@" > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err)
    python $R/tools/rocprof_summary.py /tmp/tr_bench $OUT/rocprofv3_kernel_trace_summary_bench_1M.json > /dev/null 2>&1
    f=$(find /tmp/tr_bench -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_bench_1M.csv; cut -c1-220 $f | head -4
    tail -n 1 $OUT/bench_under_rocprofv3.json | cut -c1-400 ;;
  pmc_traffic)
    for C in FETCH_SIZE WRITE_SIZE; do
      pmc_pass traffic_$C $C python $R/tools/variant_bench.py --pages 50000 --variants 6 --rounds 3 --no-batch
    done
    python $R/tools/pmc_traffic.py $OUT/pmc_traffic_FETCH_SIZE.raw.json $OUT/pmc_traffic_WRITE_SIZE.raw.json 50000 $OUT/pmc_traffic_$ROUND.json && mkdir -p $R/profiles/$ROUND && cp $OUT/pmc_traffic_$ROUND.json $R/profiles/$ROUND/pmc_traffic_$ROUND.json ;;
  pmc_traffic_aux)
    pages=${1:-400000}
    for C in FETCH_SIZE WRITE_SIZE; do
      pmc_pass aux_traffic_$C $C python $R/tools/aux_traffic_probe.py $pages
    done
    python $R/tools/aux_traffic.py $OUT/pmc_aux_traffic_FETCH_SIZE.raw.json $OUT/pmc_aux_traffic_WRITE_SIZE.raw.json $pages $OUT/pmc_traffic_aux_scans_$ROUND.json | head -30 ;;
  batch_sq)
    pages=${1:-200000}
    (cd $R && python tools/batch_scan_probe.py $pages ${2:-0:16,0:4} 5 > $OUT/batch_scan_probe.jsonl 2> $OUT/batch_scan_probe.err; cat $OUT/batch_scan_probe.jsonl)
    pmc_pass batch_set1 "$SQ1" python $R/tools/batch_scan_probe.py $pages ${2:-0:16} 3
    pmc_pass batch_set2 "$SQ2" python $R/tools/batch_scan_probe.py $pages ${2:-0:16} 3
    python $R/tools/sq_summary.py $OUT/pmc_batch_set1.raw.json $OUT/pmc_batch_set2.raw.json maxsim_batch $OUT/pmc_sq_batched_bf16_B16.json ;;
  binary_sq)
    pages=${1:-1000000}
    pmc_pass binary_set1 "$SQ1" python $R/tools/binary_probe.py $pages 4
    pmc_pass binary_set2 "$SQ2" python $R/tools/binary_probe.py $pages 4
    python $R/tools/sq_summary.py $OUT/pmc_binary_set1.raw.json $OUT/pmc_binary_set2.raw.json maxsim_binary $OUT/pmc_sq_sign_bit_scan.json ;;
  fde_e4m3_sq)
    pages=${1:-1250000}
    pmc_pass fde8_set1 "$SQ1" python $R/tools/fde_e4m3_probe.py $pages 1
    pmc_pass fde8_set2 "$SQ2" python $R/tools/fde_e4m3_probe.py $pages 1
    python $R/tools/sq_summary.py $OUT/pmc_fde8_set1.raw.json $OUT/pmc_fde8_set2.raw.json fde_scan $OUT/pmc_sq_fde_scans_bf16_and_e4m3.json ;;
  probe)
    s=$1; shift
    (cd $R && python tools/$s "$@" > $OUT/${s%.py}.jsonl 2> $OUT/${s%.py}.err; cat $OUT/${s%.py}.jsonl | cut -c1-600) ;;
  pmc)
    name=$1; shift; set=""
    while [ $# -gt 0 ] && [ "$1" != "--" ]; do set="$set $1"; shift; done; shift
    pmc_pass $name "$set" "$@"; python -c "import json,sys; d=json.load(open('$OUT/pmc_$name.raw.json'))['counters']; print(json.dumps({k[:80]: {c: round(v['avg'],1) for c,v in cs.items()} for k,cs in d.items()}, indent=1)[:4000])" ;;
  *) echo "unknown profile $what"; exit 2 ;;
esac
