#!/bin/bash
# The CPU-emulated kernel build under AddressSanitizer: every kernel source compiled with -fsanitize=address against tests/emu/hip/hip_runtime.h
# (device buffers are host allocations there, so an out-of-bounds global or LDS access of a kernel is a heap / global redzone hit), then the
# emulated test files run against it (H2HIP_EMU_LIB).  Takes ~1 h on 4 workers; run from the repo root:  bash tools/emu_asan.sh [pytest args]
# What it cannot see: races between the lanes of a workgroup (fibers run one at a time) — those need the GPU (tools/soak.py, tools/fuzz_shapes.py).
set -eu
OUT=${H2HIP_ASAN_DIR:-/tmp/emu_asan}
CXX=/opt/rocm/lib/llvm/bin/clang++
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p $OUT
for s in halo2-lib_amd/csrc/*.hip; do
    $CXX -x c++ -std=c++17 -O1 -g -fPIC -fsanitize=address -fno-omit-frame-pointer -I tests/emu -Wno-unused-value -Wno-pass-failed -c $s -o $OUT/$(basename $s .hip).o &
done
wait
$CXX -shared -fPIC -fsanitize=address -shared-libasan -o $OUT/libh2hip_emu_asan.so $OUT/*.o -lpthread
H2HIP_EMU_LIB=$OUT/libh2hip_emu_asan.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    python -m pytest tests/test_emu_kernels.py tests/test_plonk_prover.py tests/test_prover_flow.py tests/test_virtual_region.py tests/test_sharded_single_rank.py \
    -q -m "not gpu" -n 4 "$@"
