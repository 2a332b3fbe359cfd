#!/bin/bash
# Sanitizer pass over the C-ABI shim (no GPU needed): builds the library with the HOST code instrumented (device code untouched, -O1),
# once with AddressSanitizer + UndefinedBehaviorSanitizer and once with ThreadSanitizer, links tools/abi_sanitize_driver.cpp against each
# and runs it.  ~3 minutes per flavour on 8 cores.  Output: what the driver and the sanitizers print; exit code 0 = clean.
#   bash tools/sanitize_abi.sh [asan|tsan|both]       (tests/test_sanitizer_cpu.py runs it when UMV_TEST_SANITIZE=1)
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
WHAT=${1:-both}
SRCS="host_error elementwise pack gemm gemm_w4 gemm_fp8mfma attention attention_prefill vision"
build_and_run() {
  local name=$1 flags=$2 D=unimedvl_amd/lib/san_$1
  mkdir -p $D
  for f in $SRCS; do
    extra=""; [ $f = attention_prefill ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -fno-omit-frame-pointer $flags -fno-gpu-sanitize $extra \
      -Wno-unused-result -Wno-unused-value -c unimedvl_amd/csrc/$f.hip -o $D/$f.o 2> $D/$f.log &
  done
  wait
  objs=""; for f in $SRCS; do [ -f $D/$f.o ] || { echo "compile failed: $f"; cat $D/$f.log | tail -20; exit 1; }; objs="$objs $D/$f.o"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC $flags -o $D/libunimedvl_hip_$name.so $objs
  $HIPCC -x c++ -O1 -g -std=c++17 $flags -fno-omit-frame-pointer tools/abi_sanitize_driver.cpp -L$D -lunimedvl_hip_$name -Wl,-rpath,$PWD/$D -lpthread -o $D/driver
  echo "== $name: running tools/abi_sanitize_driver.cpp against $D/libunimedvl_hip_$name.so"
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 TSAN_OPTIONS=halt_on_error=1 $D/driver 8 50
  echo "== $name: clean (exit 0)"
}
[ $WHAT = asan ] || [ $WHAT = both ] && build_and_run asan "-fsanitize=address,undefined"
[ $WHAT = tsan ] || [ $WHAT = both ] && build_and_run tsan "-fsanitize=thread"
exit 0
