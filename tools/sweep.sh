#!/bin/bash
# Runs bench.py over a list of "name:args" variants in one GPU job; prints value/ms per variant.
for v in "$@"; do
  name=${v%%:*}; args=${v#*:}
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 3 --warmup 1 $args > gpurun_out/sweep_$name.json 2> gpurun_out/sweep_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/sweep_$name.err; exit 1; }
  python - "$name" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/sweep_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "Mrays/s", d["ms_per_step"], "ms/step", d["seconds_per_frame"], "s/frame")
PY
done
