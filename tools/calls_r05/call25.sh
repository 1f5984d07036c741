#!/bin/bash
# r05 GPU call 25 (last): the whole -m gpu suite + smoke on the round's final sources, a long fuzz with the knobs (incl. the round's last switches), the bench line
set -u
O=$PWD/gpurun_out/r05c25; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
H2HIP_FUZZ_KNOBS=1 timeout 500 python tools/fuzz_shapes.py 400 21 > $O/fuzz_knobs.log 2>&1; tail -1 $O/fuzz_knobs.log
H2HIP_FUZZ_KNOBS=1 timeout 400 python tools/fuzz_shapes.py 300 22 13 17 > $O/fuzz_knobs_mid.log 2>&1; tail -1 $O/fuzz_knobs_mid.log
( time timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json ) 2> $O/time.log; tail -3 $O/time.log; head -c 400 $O/bench.json; echo
