#!/bin/bash
# r06 GPU call 3: which part of the new LDS layout costs the small transforms 3-4 % — planes without a swizzle (2), the matrix swizzle (1), a two-shift swizzle (3), r05's layout (0)
set -u
O=$PWD/gpurun_out/r06c03; mkdir -p $O
timeout 900 python tools/ntt_r04.py ntt_lds_planes=0:1:2:3:0:1:2:3 > $O/ntt_times_variants.log 2>&1; cat $O/ntt_times_variants.log
