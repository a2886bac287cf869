#!/bin/bash
# PMC passes (counters only) over the batched FDE coarse scan: HBM traffic (self-calibrated on the single-query scan of
# the same process: 20 480 B per page, known) and SQ activity.  Summaries -> gpurun_out/pmc_fde_batch_*.json
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmcfb_$i
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcfb_$i -- python $R/tools/fde_batch_scan_probe.py 200000 > $OUT/pmc_fde_batch_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcfb_$i $OUT/pmc_fde_batch_$i.json > /dev/null 2>&1
  tail -1 $OUT/pmc_fde_batch_$i.log | cut -c1-200
done
