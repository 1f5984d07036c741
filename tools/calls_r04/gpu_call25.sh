#!/bin/bash
mkdir -p gpurun_out/c25
timeout 300 python tools/rng_ab.py 19 > gpurun_out/c25/rng_ab.log 2>&1
tail -12 gpurun_out/c25/rng_ab.log
