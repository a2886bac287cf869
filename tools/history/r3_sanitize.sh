#!/bin/bash
# Sanitizer builds of the host side (csrc/Makefile: tsan / asan) under the concurrency tests of the GPU suite: reader threads on
# one index, ingest beside queries, the request coalescer, mv_comm (shard fan-out, exchange, batched two-stage).  Device code
# is unchanged.  Logs -> gpurun_out/sanitize_{tsan,asan}.log; reports are counted at the end.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
make -C $R/morphik-core_amd/csrc -s -j8 tsan asan 2>&1 | tail -3
TESTS="tests/test_gpu_parity.py::test_concurrent_threads_on_one_index_get_their_own_answers tests/test_gpu_parity.py::test_ingest_runs_beside_queries_append_only_publish tests/test_gpu_store.py::test_request_coalescing_on_the_real_index tests/test_gpu_store.py::test_request_coalescing_on_the_fast_store_rides_the_batched_fde_pipeline tests/test_gpu_sharded.py::test_shard_comm_equals_single_index_all_modes tests/test_gpu_sharded.py::test_sharded_index_batch_runs_every_shard_batched_and_merges_exactly tests/test_gpu_sharded.py::test_comm_batched_two_stage_equals_single_index_batched_pipeline"
for SAN in tsan asan; do
  RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.$SAN-x86_64.so)
  if [ $SAN = tsan ]; then
    OPTS="TSAN_OPTIONS=report_signal_unsafe=0:history_size=4:halt_on_error=0:second_deadlock_stack=1:suppressions=$R/tools/tsan.supp"
  else
    OPTS="ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0"
  fi
  (cd $R && env $OPTS LD_PRELOAD=$RT MVMAXSIM_LIB=$R/morphik-core_amd/libmvmaxsim_$SAN.so timeout 1200 python -m pytest $TESTS -x -q --timeout 1100 -p no:cacheprovider > $OUT/sanitize_$SAN.log 2>&1)
  echo "$SAN: exit $? ; reports: $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer' $OUT/sanitize_$SAN.log) ; in libmvmaxsim frames: $(grep -c 'libmvmaxsim_' $OUT/sanitize_$SAN.log)"
  tail -4 $OUT/sanitize_$SAN.log
done
