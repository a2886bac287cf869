#!/bin/bash
# Round-end dress rehearsal: smoke, full GPU suite, then the measurement call (tools/r2_bench.sh).
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
bash tools/r2_bench.sh
