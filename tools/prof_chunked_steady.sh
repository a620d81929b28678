#!/bin/bash
# Kernel mix of BASELINE config 3 in its STEADY state (32 decodes + a 32-token chunk per step): rocprofv3 kernel stats of
# tools/bench_chunked_prefill.py with enough prompts that the drain is a small share.  Run on the GPU box.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_chunked_steady; rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/bench_chunked_prefill.py --num-prompts 256 > $O/run.log 2> $O/run.err
python3 tools/prof_summary.py stats $O/stats $O/r04_chunked_prefill_steady_kernel_stats.csv | head -16
rm -rf $O/stats; tail -2 $O/run.log
