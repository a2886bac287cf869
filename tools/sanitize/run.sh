#!/bin/bash
# Host-side sanitizer runs WITHOUT a GPU: libmvmaxsim_{tsan,asan}.so (csrc/Makefile) + the host-only HIP stub + host_stress.cpp.
#   bash tools/sanitize/run.sh [iterations]      -> profiles/r3/sanitize_{tsan,asan}_host.log
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
ITERS=${1:-200}
CLANG=/opt/rocm/lib/llvm/bin/clang
CLANGXX=/opt/rocm/lib/llvm/bin/clang++
B=$R/tools/sanitize/build; mkdir -p $B $R/profiles/r3
make -C $R/morphik-core_amd/csrc -s -j8 tsan asan || exit 1
for SAN in thread address; do
  S=$([ $SAN = thread ] && echo tsan || echo asan)
  mkdir -p $B/$S
  $CLANG -O1 -g -fPIC -shared -fsanitize=$SAN -shared-libsan -Wl,--version-script=$R/tools/sanitize/hip_stub.map -Wl,-soname,libamdhip64.so.7 \
      -o $B/$S/libamdhip64.so.7 $R/tools/sanitize/hip_stub.c || exit 1
  $CLANGXX -O1 -g -std=c++17 -fsanitize=$SAN -shared-libsan -I$R/include -o $B/$S/host_stress $R/tools/sanitize/host_stress.cpp \
      $R/morphik-core_amd/libmvmaxsim_$S.so -L$B/$S -l:libamdhip64.so.7 -lpthread -Wl,-rpath,$R/morphik-core_amd || exit 1
  RT=$(dirname $($CLANG -print-file-name=libclang_rt.$S-x86_64.so))
  LOG=$R/profiles/r3/sanitize_${S}_host.log
  if [ $S = tsan ]; then OPT="TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1:history_size=4"; else OPT="ASAN_OPTIONS=detect_leaks=1:halt_on_error=0"; fi
  ( cd /tmp && env $OPT LD_LIBRARY_PATH=$B/$S:$RT timeout 1200 $B/$S/host_stress $ITERS ) > $LOG 2>&1
  echo "$S: exit $? ; reports: $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|ERROR: LeakSanitizer' $LOG)"; tail -2 $LOG
done
