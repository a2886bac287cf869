#!/bin/bash
# One gpurun call: parity tests, then kernel A/B numbers.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -80) > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
(timeout 600 python tools/variant_bench.py --pages 200000 --variants 3,6,2,7 --rounds 5 --aux --out gpurun_out/variants.json 2>&1 | tail -20) > gpurun_out/variants.log 2>&1
grep -v '^ *"' gpurun_out/variants.log | grep -v "^[{}]" | head -20
