#!/bin/bash
# Three more counter passes on the vector L1 of the closest-hit kernel (latency of its L2 reads, its stall reasons,
# address translation).  Usage: tools/profile_l1.sh <tag> [bench args]; results under gpurun_out/<tag>/.
set -e
tag=$1; shift
one="--share-primary 0"
case " $* " in *" --share-primary "*) one="";; esac
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
pmc() {   # <name> <counters>
    rm -rf $out/pmc_$1; mkdir -p $out/pmc_$1
    timeout -s KILL 500 rocprofv3 --pmc $2 --output-format csv -d $out/pmc_$1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-device-tree $one "${@:3}" > $out/pmc_$1/bench.json 2> $out/pmc_$1/bench.err
    python3 tools/pmc_summary.py $out/pmc_$1 > $out/pmc_$1.txt
    grep -A8 "wf_extend_kernel<false, true>" $out/pmc_$1.txt | cut -c1-100
}
pmc l1lat "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "$@"
pmc l1tlb "GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "$@"
pmc l1stall "GRBM_GUI_ACTIVE TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" "$@"
