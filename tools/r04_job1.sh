#!/bin/bash
# Round 4, GPU call 1: probes (memory-pipeline order of a CU; in-launch merge tail), GEMM grid A/B, TA counters of the
# decode projections, the encode-only step's kernel breakdown, and a first bench line on this round's tree.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_job1; mkdir -p $O
step() {  # step <name> <timeout> <cmd...>: a step that is killed at its limit ends the job (no GPU step after a hang)
  local name=$1 lim=$2; shift 2
  echo "== $name" | tee -a $O/steps.log
  timeout -k 10 $lim "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "$name rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping after $name" | tee -a $O/steps.log; exit 1; fi
}
step probe_mem_order 120 tools/bin/probe_mem_order
step probe_handoff_32x1024 120 tools/bin/probe_handoff 256 2 256
step probe_handoff_8x4096 120 tools/bin/probe_handoff 64 4 512
step probe_handoff_16x2048 120 tools/bin/probe_handoff 128 2 512
step ab_gemm_grid 300 python3 tools/ab_gemm_grid.py 256 224 192 128 0
for grp in "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum" "TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum"; do
  tag=ta_$(echo $grp | cut -d' ' -f1)
  step pmc_$tag 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -- python3 tools/pmc_gemm.py
done
python3 tools/prof_summary.py counters skinny_gemm_kernel $O/r04_pmc_gemm_ta.json $(ls -d $O/ta_*/ 2>/dev/null) > $O/ta_summary.log 2>&1
rm -rf $O/ta_*/
step prof_encode 500 bash tools/prof_encode.sh
cp gpurun_out/prof_encode/kernel_stats.csv $O/r04_encode_kernel_stats.csv 2>/dev/null
step bench 900 python3 bench.py
tail -1 $O/bench.log > $O/r04_bench_line_base.json
cat $O/steps.log
