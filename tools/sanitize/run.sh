#!/bin/bash
# Host-side sanitizer runs WITHOUT a GPU: libmvmaxsim_{tsan,asan}.so (csrc/Makefile) + the host-only HIP stub + host_stress.cpp.
#   bash tools/sanitize/run.sh [iterations]      -> profiles/r6/sanitize_{tsan,asan}_host_8dev.log (PROFILE_DIR)
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
ITERS=${1:-200}
CLANG=/opt/rocm/lib/llvm/bin/clang
CLANGXX=/opt/rocm/lib/llvm/bin/clang++
P=${PROFILE_DIR:-$R/profiles/r6}
B=$R/tools/sanitize/build; mkdir -p $B $P
make -C $R/morphik-core_amd/csrc -s -j8 tsan asan || exit 1
for SAN in thread address; do
  S=$([ $SAN = thread ] && echo tsan || echo asan)
  mkdir -p $B/$S
  $CLANG -O1 -g -fPIC -shared -fsanitize=$SAN -shared-libsan -Wl,--version-script=$R/tools/sanitize/hip_stub.map -Wl,-soname,libamdhip64.so.7 \
      -o $B/$S/libamdhip64.so.7 $R/tools/sanitize/hip_stub.c -lpthread || exit 1
  # the RCCL stand-in mv_comm dlopens as librccl.so.1 (grouped all-gather = memcpy across the stub's devices, with rank / device checks)
  $CLANG -O1 -g -fPIC -shared -fsanitize=$SAN -shared-libsan -Wl,-soname,librccl.so.1 -o $B/$S/librccl.so.1 $R/tools/sanitize/rccl_stub.c \
      -L$B/$S -l:libamdhip64.so.7 -lpthread || exit 1
  $CLANGXX -O1 -g -std=c++17 -fsanitize=$SAN -shared-libsan -I$R/include -o $B/$S/host_stress $R/tools/sanitize/host_stress.cpp \
      $R/morphik-core_amd/libmvmaxsim_$S.so -L$B/$S -l:libamdhip64.so.7 -lpthread -ldl -Wl,-rpath,$R/morphik-core_amd || exit 1
  RT=$(dirname $($CLANG -print-file-name=libclang_rt.$S-x86_64.so))
  LOG=$P/sanitize_${S}_host_8dev.log
  if [ $S = tsan ]; then OPT="TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1:history_size=4"; else OPT="ASAN_OPTIONS=detect_leaks=1:halt_on_error=0"; fi
  ( cd /tmp && env $OPT LD_LIBRARY_PATH=$B/$S:$RT timeout 1200 $B/$S/host_stress $ITERS ) > $LOG 2>&1
  echo "$S: exit $? ; reports: $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|ERROR: LeakSanitizer\|AFFINITY VIOLATION\|RCCL STUB VIOLATION' $LOG)"; tail -2 $LOG
  # the checker itself must bite: a deliberate cross-device memset has to abort
  ( cd /tmp && env $OPT LD_LIBRARY_PATH=$B/$S:$RT $B/$S/host_stress violation ) > $B/$S/violation.log 2>&1
  if grep -q "HIP STUB AFFINITY VIOLATION: hipMemset" $B/$S/violation.log; then echo "$S: affinity self-test: the deliberate violation aborted as it must" | tee -a $LOG; else echo "$S: affinity self-test FAILED" | tee -a $LOG; fi
done
