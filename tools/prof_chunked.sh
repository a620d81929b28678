#!/bin/bash
# Where a mixed step of BASELINE config 3 (chunked prefill, budget 64) spends its GPU time: rocprofv3 kernel stats of
# tools/bench_chunked_prefill.py on a short run.  Run on the GPU box.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_chunked; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/bench_chunked_prefill.py --num-prompts 48 --output-len 256 > $O/run.log 2> $O/run.err
python3 tools/prof_summary.py stats $O/stats $O/r03_chunked_prefill_kernel_stats.csv | head -14
rm -rf $O/stats; tail -2 $O/run.log
