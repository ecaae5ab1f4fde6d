#!/bin/bash
# Two bench processes at once on one GPU, each with half the wavefront pool: do the tails of one fill with the other's work?
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 6 --warmup 2 --wf-log2 24 > gpurun_out/two_a.json 2> gpurun_out/two_a.err &
pa=$!
timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 6 --warmup 2 --wf-log2 24 > gpurun_out/two_b.json 2> gpurun_out/two_b.err &
pb=$!
wait $pa; ra=$?
wait $pb; rb=$?
echo "exit $ra $rb"
python - <<'PY'
import json
t=0
for f in ("gpurun_out/two_a.json","gpurun_out/two_b.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"]); t+=d["value"]
print("sum", t)
PY
timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 3 --warmup 1 --wf-log2 24 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single, pool 2^24:', d['value'])"
