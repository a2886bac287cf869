#!/bin/bash
# rocprofv3 kernel stats of the encoder forward at batch 32 (random-init ColPali-v1.2 architecture)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/tr_embed
MV_EMBED_PROBE_BATCHES=32 timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_embed -- python $R/tools/embed_batch_probe.py > $OUT/embed_probe_prof.log 2>&1
f=$(find /tmp/tr_embed -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprofv3_kernel_stats_embed_b32.csv; cut -c1-150 $f | head -25
tail -3 $OUT/embed_probe_prof.log
