#!/bin/bash
# Kernel-trace + stats profile of the benchmark (run on the GPU box via gpurun). Usage: tools/prof_trace.sh <tag> [bench args]
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
echo exit $?
find $out -name '*stats*' | head
