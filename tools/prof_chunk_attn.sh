#!/bin/bash
# kernel durations of the partitioned prompt launches (tools/bench_chunk_attn.py --only-shipped) under rocprofv3
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_chunk; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run -- python3 tools/bench_chunk_attn.py --only-shipped --iters 50 > $O/out.txt 2> $O/err.txt
python3 tools/prof_summary.py stats $O/run $O/kernel_stats.csv > /dev/null
head -12 $O/kernel_stats.csv | cut -c1-160
