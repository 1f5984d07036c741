#!/bin/bash
# r06 GPU call 17 (run inline): the 40th golden shape — bench_msm.config:13, k = 23 with 6 + 1 advice columns, digest written by the oracle in its
# low-memory mode — proved on the GPU, plus the k = 23 2 + 1-column shape and the one-rank RCCL transport test with the all-to-all
timeout 1500 python -m pytest tests/test_reference_shapes_golden.py tests/test_sharded_single_rank.py -m gpu -x -q -k "msm-L13 or msm-L8 or rccl_transport" --durations=3 2>&1 | tail -12
