#!/bin/bash
# Everything the committed profile summaries come from, in one GPU job: the default bench, a kernel trace, and the counter
# passes (each counter set in a run of its own, no trace domains).  Usage: tools/profile_all.sh <tag> [bench args]
# Results under gpurun_out/<tag>/; tools/make_traffic.py <tag> <round> turns them into profiles/<round>/.
set -e
tag=$1; shift
# bench.py by itself times the frame twice (every ray walked = its value; shared first walk beside it).  The trace and the
# counter passes must see one kind of launch only: the headline's, unless the arguments choose.
one="--share-primary 0"
case " $* " in *" --share-primary "*) one="";; esac
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf $out      # counter files of an earlier run under the same tag would be summed in
mkdir -p $out
timeout -k 10 400 python3 bench.py "$@" > $out/bench_default.json 2> $out/bench_default.log
mkdir -p $out/trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $one "$@" > $out/bench_under_rocprof.json 2> $out/trace/bench.err
cp "$(find $out/trace -name '*kernel_stats.csv' | head -1)" $out/bench_kernel_stats.csv
pmc() {   # <name> <counters>
    mkdir -p $out/pmc_$1
    timeout -s KILL 500 rocprofv3 --pmc $2 --output-format csv -d $out/pmc_$1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $one "${@:3}" > $out/pmc_$1/bench.json 2> $out/pmc_$1/bench.err
    python3 tools/pmc_summary.py $out/pmc_$1 > $out/pmc_$1.txt
}
pmc fetch "FETCH_SIZE" "$@"
pmc write "WRITE_SIZE" "$@"
pmc sq "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$@"
pmc stall "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS" "$@"
pmc stall2 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "$@"
pmc l2 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$@"
rm -rf $out/trace/*/  # the raw per-process trace directories are large; the stats file was copied out
cut -c1-200 $out/bench_default.json
head -4 $out/bench_kernel_stats.csv | cut -c1-160
