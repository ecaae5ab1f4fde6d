#!/bin/bash
# bench.py with every raytrace_amd/_ab_*.so swapped in for librtcore.so in turn, two rounds, on one box.
set -e
cd "$(dirname "$0")/.."
cp raytrace_amd/librtcore.so /tmp/_keep.so
for round in 1 2; do
  for lib in raytrace_amd/_ab_*.so; do
    cp "$lib" raytrace_amd/librtcore.so
    v=$(timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_share']['extend_ms'], d['kernel_share']['shade_ms'])")
    echo "$(basename $lib) $v"
  done
done
cp /tmp/_keep.so raytrace_amd/librtcore.so
