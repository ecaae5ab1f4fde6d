#!/bin/bash
# The benchmark through the reference's own entry point: bin/main <scene> <outdir> <batch_size> on a tracked scene file
# (benchmarks/scenes/), timed the way upstream times a frame (csrc/worker.cpp:67-96: clear .. post-process, "done in N ms").
# Usage: tools/bench_cli.sh <luyu|formula1|formula1_glass|shuttle|stress> [batch_size=16] [outdir=gpurun_out/cli]
# RT_GPUS=N splits the samples over N GPUs of the node.  Model files are generated on first use.
set -e
name=${1:-formula1}; batch=${2:-16}; out=${3:-gpurun_out/cli}
cd "$(dirname "$0")/.."
[ -f benchmarks/assets/$name/${name}_obj0.obj ] || python3 tools/make_bench_scenes.py --assets $name >&2
mkdir -p $out
RT_SEED=${RT_SEED:-0x5EED0001} bin/main benchmarks/scenes/scene_$name.txt $out $batch > $out/$name.log 2>&1
grep -E "done in|Image saved|contexts on" $out/$name.log
ms=$(sed -n 's/.*done in \([0-9]*\)ms.*/\1/p' $out/$name.log | tail -1)
echo "{\"scene\": \"$name\", \"batch_size\": $batch, \"gpus\": ${RT_GPUS:-1}, \"seconds_per_frame\": $(python3 -c "print($ms/1000)")}" | tee $out/$name.json
