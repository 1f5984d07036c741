#!/bin/bash
# r04 GPU call 10: hardware-queue aliasing — GPU_MAX_HW_QUEUES x plonk_tail_overlap
mkdir -p gpurun_out/c10; O=gpurun_out/c10
for q in 4 8 12; do
  for ov in 1 2; do
    GPU_MAX_HW_QUEUES=$q timeout 200 python tools/prove_time.py 19 1 1 1 0 18 14 --param=plonk_tail_overlap=$ov 2>&1 | grep "create_proof rep" | tail -8 | awk -v q=$q -v ov=$ov '{s+=$4; n++; if(min==""||$4<min)min=$4} END {printf "queues=%s overlap=%s mean %.3f min %.3f ms\n", q, ov, s/n, min}' >> $O/q.log
  done
done
GPU_MAX_HW_QUEUES=8 timeout 200 python tools/prove_time.py 19 1 1 1 0 18 8 --param=plonk_tail_overlap=2 2>&1 | tail -14 >> $O/stages_q8_ov2.log
cat $O/q.log; cat $O/stages_q8_ov2.log
