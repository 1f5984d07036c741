#!/bin/bash
# r04 GPU call 7: accumulation loop A/B (emptiness flag + prefetched offsets), timing and PMC
mkdir -p gpurun_out/c7; O=gpurun_out/c7
timeout 300 python tools/msm_r03.py 19,20 msm_accum_flag=0:1:0:1 > $O/msm_ab.log 2>&1
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 4 --ab=msm_accum_flag 2>&1 | grep "create_proof min" > $O/prove19.log
H2_PARAMS="msm_accum_flag=0" timeout 300 bash tools/accum_pmc.sh > $O/pmc_flag0.log 2>&1
H2_PARAMS="msm_accum_flag=1" timeout 300 bash tools/accum_pmc.sh > $O/pmc_flag1.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm" > $O/pytest.log 2>&1
cat $O/msm_ab.log | cut -c1-260; cat $O/prove19.log; echo flag0; cat $O/pmc_flag0.log; echo flag1; cat $O/pmc_flag1.log; tail -2 $O/pytest.log
