#!/bin/bash
# r06 GPU call 5: fewer, wider passes? ntt_min_col_bits 2 (default: 3 passes at 2^19..2^24) against 1 and 0 (2 passes up to 2^20 / 2^18..)
set -u
O=$PWD/gpurun_out/r06c05; mkdir -p $O
timeout 900 python tools/ntt_r04.py ntt_min_col_bits=2:1:0:2:1:0 > $O/ntt_min_col_bits.log 2>&1; cat $O/ntt_min_col_bits.log
