#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -60) > gpurun_out/pytest_gpu.log 2>&1
tail -8 gpurun_out/pytest_gpu.log
timeout 1200 python tools/bench_embed.py --pages 1000 --out gpurun_out/bench_embed_cfg2.json 2> gpurun_out/bench_embed.err | cut -c1-1500; tail -5 gpurun_out/bench_embed.err
