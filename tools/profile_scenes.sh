#!/bin/bash
# Kernel traces of the other benchmark scenes and the three counter passes of the 30 M-triangle scene, in one GPU job.
# Results under gpurun_out/; copy the summaries to profiles/<round>/.
set -e
mkdir -p gpurun_out
for sc in formula1_glass shuttle luyu; do
  tools/prof_trace.sh $sc --scene $sc
  cp "$(find gpurun_out/prof_$sc -name '*kernel_stats.csv' | head -1)" gpurun_out/${sc}_kernel_stats.csv
done
S="--scene stress --width 3840 --height 2160"
tools/prof_pmc.sh stress_fetch FETCH_SIZE $S
tools/prof_pmc.sh stress_write WRITE_SIZE $S
tools/prof_pmc.sh stress_sq "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $S
python3 tools/pmc_summary.py gpurun_out/pmc_stress_fetch > gpurun_out/stress_pmc_fetch_size.txt
python3 tools/pmc_summary.py gpurun_out/pmc_stress_write > gpurun_out/stress_pmc_write_size.txt
python3 tools/pmc_summary.py gpurun_out/pmc_stress_sq > gpurun_out/stress_pmc_sq.txt
cp gpurun_out/pmc_stress_sq/bench.json gpurun_out/stress_bench_under_pmc.json
grep -h -A12 "wf_extend_kernel<false, true>" gpurun_out/stress_pmc_fetch_size.txt gpurun_out/stress_pmc_write_size.txt gpurun_out/stress_pmc_sq.txt | cut -c1-100
