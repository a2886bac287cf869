#!/bin/bash
# rocprofv3 PMC passes (counters only, no trace domains) over the aux kernel set; summaries -> gpurun_out/pmc_*.json
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/rocprof_counters_list.txt 2>&1 || true
CMD="python $GRAFT_REPO_ROOT/tools/variant_bench.py --pages 100000 --variants 3 --rounds 2 --aux"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_$i -- $CMD > $OUT/pmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/pmc_$i $OUT/pmc_$i.json > /dev/null 2>&1
  tail -3 $OUT/pmc_$i.log
done
