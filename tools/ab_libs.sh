#!/bin/bash
# Interleaved A/B of library builds on one box: tools/ab_libs.sh "<bench args>" lib1.so lib2.so ... (rounds: $ROUNDS, default 2)
args=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    RT_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $args > gpurun_out/ab/$name.$r.json 2> gpurun_out/ab/$name.$r.err || { echo "$name FAILED"; tail -3 gpurun_out/ab/$name.$r.err; exit 1; }
    python3 -c "
import json;d=json.load(open('gpurun_out/ab/$name.$r.json'));k=d['kernel_share'];print('$name round $r:',d['value'],'Mrays/s extend',round(k['extend_ms'],1),'shade',round(k['shade_ms'],1),'wall',round(k['wall_ms'],1))"
  done
done
