#!/bin/bash
# r05 GPU call 17: several proofs in flight (one host thread + context + key each): throughput against the single proof's latency
set -u
O=$PWD/gpurun_out/r05c17; mkdir -p $O
timeout 600 python tools/two_in_flight.py 19 1 1 1 18 20 3 > $O/k19.log 2>&1; tail -12 $O/k19.log
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/two_in_flight.py 19 1 1 1 18 20 3 > $O/k19_q8.log 2>&1; tail -11 $O/k19_q8.log
timeout 600 python tools/two_in_flight.py 17 4 1 1 16 20 3 > $O/k17.log 2>&1; tail -11 $O/k17.log
