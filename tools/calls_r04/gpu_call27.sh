#!/bin/bash
# r04 GPU call 27: after the round's scheduling changes (tail hook before the reduction, permutation inside the commitment batch, third lane context,
# zero-fill on lane 0): random shapes byte-equal to the oracle prover, repeated proofs byte-identical
mkdir -p gpurun_out/c27; O=gpurun_out/c27
timeout 400 python tools/fuzz_shapes.py 240 4 > $O/fuzz_small.log 2>&1
timeout 300 python tools/fuzz_shapes.py 150 5 13 16 > $O/fuzz_mid.log 2>&1
timeout 400 python tools/soak.py 300 > $O/soak.log 2>&1
tail -3 $O/fuzz_small.log; tail -3 $O/fuzz_mid.log; tail -4 $O/soak.log
