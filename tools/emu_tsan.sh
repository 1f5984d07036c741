#!/bin/bash
# A data-race check for the kernels without a GPU: the CPU-emulated build with every emulated GPU thread registered as a ThreadSanitizer *fiber*
# (tests/emu/hip/hip_runtime.h, H2_EMU_TSAN).  Fiber switches carry no happens-before edge; __syncthreads (block), cross-lane operations and
# H2_WAVE_SYNC (wave) and kernel boundaries do.  Two lanes that touch the same LDS / global word with no such edge between them — a race on the
# GPU, invisible to the plain emulation because its fibers run one at a time — become ThreadSanitizer reports (validated on the r03 two-level
# sort's LDS race: 40 reports in msm_csort_fine_kernel with the bug put back, none without).  Limits: blocks that ran one after the other on
# the same emulator worker are ordered (inter-block races are only seen across workers); ~15x slower than the plain emulation.
#   bash tools/emu_tsan.sh [pytest args, default: tests/test_emu_kernels.py]        reports: $OUT/report.<pid>
set -eu
OUT=${H2HIP_TSAN_DIR:-/tmp/emu_tsan}
CXX=/opt/rocm/lib/llvm/bin/clang++
TSAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
mkdir -p $OUT
for s in halo2-lib_amd/csrc/*.hip; do
    $CXX -x c++ -std=c++17 -O1 -g -fPIC -fsanitize=thread -DH2_EMU_TSAN -fno-omit-frame-pointer -I tests/emu -Wno-unused-value -Wno-pass-failed -c $s -o $OUT/$(basename $s .hip).o &
done
wait
$CXX -shared -fPIC -fsanitize=thread -shared-libsan -o $OUT/libh2hip_emu_tsan.so $OUT/*.o -lpthread
rm -f $OUT/report.*
[ $# -gt 0 ] || set -- tests/test_emu_kernels.py
rc=0
H2HIP_EMU_LIB=$OUT/libh2hip_emu_tsan.so LD_PRELOAD=$TSAN TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 log_path=$OUT/report suppressions=$PWD/tests/emu/tsan.supp" \
    python -m pytest -q -s -m "not gpu" -n 3 -p no:cacheprovider "$@" || rc=$?
echo "ThreadSanitizer reports: $(cat $OUT/report.* 2>/dev/null | grep -c 'WARNING: ThreadSanitizer' || true)  (pytest rc $rc; 66 = reports were written)"
cat $OUT/report.* 2>/dev/null | grep -A3 "WARNING: ThreadSanitizer" | grep "#0" | sed 's/(.*//' | sort | uniq -c | sort -rn | head -40
