#!/bin/bash
# Round-3 batched-scan iteration: parity of the batched forms (incl. the transposed 32x32x16 variants 5 / 6), the variant
# sweep with both MFMA calibrations, and (arg "pmc") SQ counters of the B = 16 forms.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 600 -k "batched or long_query or uniform_corpus" 2>&1 | tail -15) > $OUT/pytest_batch.log 2>&1
tail -6 $OUT/pytest_batch.log
(cd $R && timeout 600 python tools/variant_bench.py --pages 200000 --variants 6 --rounds 4 --batch-only --out gpurun_out/variants_batch_r3.json 2>&1 | grep "^batch\|^mfma") | tee $OUT/variants_batch.log
if [ "${1:-}" = "pmc" ]; then
cd /tmp
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmcb_$i
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmcb_$i -- python $R/tools/variant_bench.py --pages 100000 --variants 6 --rounds 2 --batch-only --batch-variants "${2:-0:16,5:16,6:16,1:16}" > $OUT/pmcb_$i.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pmcb_$i $OUT/rocprofv3_pmc_sq_set${i}_batch_100k_r3.json > /dev/null 2>&1
done
fi
