#!/bin/bash
# r04 GPU call 13: the newly sharded stages on the GPU (one rank forced over RCCL / callback; two gloo ranks sharing the GPU at k = 19 and k = 21 with
# every stage sharded), MSM lanes at k = 20 / 21
mkdir -p gpurun_out/c13; O=gpurun_out/c13
timeout 600 python -m pytest tests/test_sharded_single_rank.py tests/test_reference_shapes_golden.py -m gpu -q -x > $O/pytest.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --share-device --dist-backend gloo --k 19 --steps 3 --warmup 1 --no-replay --no-sweep --no-pmc-traffic --shard-ntt-columns on 2> $O/bench_2rank_k19.err | tail -1 > $O/bench_2rank_shared_gpu_gloo_k19.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --share-device --dist-backend gloo --k 21 --steps 3 --warmup 1 --no-replay --no-sweep --no-pmc-traffic --shard-ntt-columns on 2> $O/bench_2rank_k21.err | tail -1 > $O/bench_2rank_shared_gpu_gloo_k21.json
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 4 --ab=msm_lanes:1,2 2>&1 | grep "create_proof min" > $O/lanes_k21.log
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 4 --ab=msm_lanes:1,3 2>&1 | grep "create_proof min" >> $O/lanes_k21.log
timeout 300 python tools/prove_time.py 20 1 1 1 0 19 4 --ab=msm_lanes:1,3 2>&1 | grep "create_proof min" > $O/lanes_k20.log
tail -3 $O/pytest.log; for f in bench_2rank_shared_gpu_gloo_k19 bench_2rank_shared_gpu_gloo_k21; do head -c 1500 $O/$f.json; echo; done; tail -3 $O/bench_2rank_k21.err; cat $O/lanes_k21.log $O/lanes_k20.log
