set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_stage or fde" 2>&1 | tail -5
# 2 ranks sharing the one GPU over gloo: functional run of the two-stage sharded pipeline
MV_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --workload fde_fp8 --pages 100000 --steps 5 --warmup 2 --no-aux 2> gpurun_out/ts2.err | grep '^{' > gpurun_out/ts2.json; tail -3 gpurun_out/ts2.err; cut -c1-600 gpurun_out/ts2.json
timeout 600 python bench.py --workload fde_fp8 --pages 200000 --steps 5 --warmup 2 --no-aux 2> gpurun_out/ts1.err | grep '^{' > gpurun_out/ts1.json; cut -c1-600 gpurun_out/ts1.json
