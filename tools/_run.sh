mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py tests/test_host_mirror.py -x -q -m gpu > gpurun_out/pytest_batch4.log 2>&1
tail -3 gpurun_out/pytest_batch4.log
python tools/config_sweep.py all 4 > gpurun_out/config_sweep_v5.md 2> gpurun_out/config_sweep_v5.err
