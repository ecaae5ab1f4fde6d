#!/bin/bash
# Everything the committed profile summaries come from, in one GPU job: the default bench, a kernel trace, and three
# counter passes (each in a run of its own).  Results under gpurun_out/; copy the summaries to profiles/<round>/.
set -e
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log
tools/prof_trace.sh final
tools/prof_pmc.sh fetch FETCH_SIZE
tools/prof_pmc.sh write WRITE_SIZE
tools/prof_pmc.sh sq "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
python3 tools/pmc_summary.py gpurun_out/pmc_fetch > gpurun_out/pmc_fetch_size.txt
python3 tools/pmc_summary.py gpurun_out/pmc_write > gpurun_out/pmc_write_size.txt
python3 tools/pmc_summary.py gpurun_out/pmc_sq > gpurun_out/pmc_sq_valu.txt
cp "$(find gpurun_out/prof_final -name '*kernel_stats.csv' | head -1)" gpurun_out/bench_kernel_stats.csv
cp gpurun_out/prof_final/bench.json gpurun_out/bench_under_rocprof.json
cut -c1-160 gpurun_out/bench_default.json
head -4 gpurun_out/bench_kernel_stats.csv | cut -c1-160
