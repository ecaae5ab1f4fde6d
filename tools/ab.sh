#!/bin/bash
# A/B on one box: bench.py with raytrace_amd/librtcore.so (new) and with raytrace_amd/_ab_old.so swapped in (old),
# interleaved, so that box-to-box differences cancel.  Arguments are passed to bench.py.
set -e
cd "$(dirname "$0")/.."
cp raytrace_amd/librtcore.so /tmp/_new.so
run() {
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
}
for round in 1 2 3; do
  cp /tmp/_new.so raytrace_amd/librtcore.so; echo "new $(run "$@")"
  cp raytrace_amd/_ab_old.so raytrace_amd/librtcore.so; echo "old $(run "$@")"
done
cp /tmp/_new.so raytrace_amd/librtcore.so
