mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_ring.log 2>&1
tail -3 gpurun_out/pytest_ring.log
for shape in "19 1 1 1 0 18" "15 17 3 1 0 14" "17 25 3 1 0 16"; do
  python tools/prove_time.py $shape 6 2>&1 | grep -E "rep [3-5]|multiopen" | sed "s/^/[$shape] /" >> gpurun_out/ring_ab.log
done
python tools/soak.py 100 > gpurun_out/soak_ring.log 2>&1; tail -2 gpurun_out/soak_ring.log
