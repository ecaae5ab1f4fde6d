#!/bin/bash
# One PMC pass of the benchmark (counters in their own run, no trace domains). Usage: tools/prof_pmc.sh <tag> "<counters>" [bench args]
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 400 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
echo exit $?
