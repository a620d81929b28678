#!/bin/bash
# Where an encode-only step (BASELINE config 4, bge-m3 shapes) spends its GPU time: rocprofv3 kernel stats of tools/bench_encode.py
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_encode; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/bench_encode.py --num-prompts 256 > $O/run.log 2> $O/run.err
tail -3 $O/run.log
python3 tools/prof_summary.py stats $O/stats $O/kernel_stats.csv > /dev/null
head -25 $O/kernel_stats.csv | cut -c1-180
