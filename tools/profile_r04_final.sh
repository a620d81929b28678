#!/bin/bash
# Round 4 records: the driver's own command (--steps 20 --warmup 5: five 20-step regions, the median reported), the
# default bench line (headline + BASELINE configs 3 / 4 / 5 in other_settings), then the headline region under rocprofv3
# (kernel stats).  Run on the GPU box.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r04_final; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_line_driver_cmd.json 2> $O/bench_driver.err; echo "bench (driver's command) rc=$?"
timeout -k 10 900 python3 bench.py > $O/r04_bench_line.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 300 $O/r04_bench_line.json; echo
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --skip-cpu-baseline --skip-ops-baseline --skip-other-configs > $O/bench_under_rocprof.json 2> $O/bench_stats.err
python3 tools/prof_summary.py stats $O/bench_stats $O/r04_bench_kernel_stats.csv > /dev/null; rm -rf $O/bench_stats
head -14 $O/r04_bench_kernel_stats.csv | cut -c1-170
# the encode-only step (BASELINE config 4) under rocprofv3: where its time goes with K / V read in place
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encode_stats -- python3 tools/bench_encode.py --num-prompts 256 > $O/encode_under_rocprof.log 2> $O/encode_stats.err
python3 tools/prof_summary.py stats $O/encode_stats $O/r04_encode_kernel_stats.csv > /dev/null; rm -rf $O/encode_stats
head -8 $O/r04_encode_kernel_stats.csv | cut -c1-170
