#!/bin/bash
# Host-side sanitizer run (GPU AddressSanitizer is not available on this pool): builds libcurious_hip with
# -fsanitize=address,undefined for the HOST code only (-fno-gpu-sanitize; the device code is compiled as usual) and runs
# the CPU tests of the C ABI -- argument validation, descriptor / layout arithmetic, symbol table -- against it.
#   bash tools/sanitize_cpu.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/curious_asan; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O1 -g -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -x hip -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
for s in api.cpp her_sample.hip store.hip normalizer.hip optim.hip ipc.hip actor.hip env.hip mlp.hip; do
  /opt/rocm/bin/hipcc $FLAGS -DCURIOUS_BUILD_DIGEST=\"sanitizer-build\" -c $ROOT/curious_amd/csrc/$s -o $OUT/${s%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $OUT/libcurious_hip_asan.so $OUT/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $ROOT
CURIOUS_LIB=$OUT/libcurious_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest tests/test_lib_cpu.py -q -k "argument_validation or refuses or host_descriptor or store_slots or transposed_copy or round3_entry or round4_entry or round5_entry" "$@"
