#!/bin/bash
# Scene load phases of the 30M-triangle stress scene: one host thread against the default thread count.
set -e
mkdir -p gpurun_out
RT_HOST_THREADS=1 timeout -k 10 600 python bench.py --scene stress --steps 1 --warmup 0 --spp-per-step 16 > gpurun_out/load_1thread.json 2> gpurun_out/load_1thread.log
timeout -k 10 400 python bench.py --scene stress --steps 1 --warmup 0 --spp-per-step 16 > gpurun_out/load_default.json 2> gpurun_out/load_default.log
grep "scene stress" gpurun_out/load_1thread.log gpurun_out/load_default.log
