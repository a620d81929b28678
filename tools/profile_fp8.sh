ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02_fp8; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --quantization fp8 --kv-cache-dtype fp8 --skip-cpu-baseline --skip-ops-baseline --also-on-the-fly 0 > $O/bench_line.json 2> $O/bench.err
python3 tools/prof_summary.py stats $O/bench $O/r02_bench_fp8_kernel_stats.csv
rm -rf $O/bench
