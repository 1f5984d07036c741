#!/bin/bash
# r05 GPU call 24: passes against coalescing once more on warm clocks — ntt_min_col_bits 2 (three passes of 6-7 bits at 2^19 / 2^20) / 1 / 0 (two passes of 9-10 bits, 32-byte row segments)
set -u
O=$PWD/gpurun_out/r05c24; mkdir -p $O
timeout 600 python tools/ntt_r04.py ntt_min_col_bits=2:1:0 > $O/ntt_mcb.log 2>&1; grep -v amdgpu $O/ntt_mcb.log
