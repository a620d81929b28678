# Round-2 profile passes (run on the GPU box; summaries land in gpurun_out/prof_r02/, the ones worth keeping are
# copied to profiles/).  Counter passes are separate runs with --kernel-trace only, as the pool requires.
set -o pipefail
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --skip-cpu-baseline --skip-ops-baseline > $O/bench_line.json 2> $O/bench.err
python3 tools/prof_summary.py stats $O/bench $O/r02_bench_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cache -- python3 tools/bench_cache.py > $O/cache.log 2> $O/cache.err
python3 tools/prof_summary.py stats $O/cache $O/r02_cache_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn -- python3 tools/bench_attn.py > $O/attn.log 2> $O/attn.err
python3 tools/prof_summary.py stats $O/attn $O/r02_attn_microbench_kernel_stats.csv > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_cache_fetch -- python3 tools/bench_cache.py --iters 24 --ncaches 8 > /dev/null 2> $O/pmc1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_cache_write -- python3 tools/bench_cache.py --iters 24 --ncaches 8 > /dev/null 2> $O/pmc2.err
python3 tools/prof_summary.py pmc $O/pmc_cache_fetch $O/pmc_cache_write reshape_and_cache_tile_kernel 67174400 $O/r02_pmc_cache.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_fetch -- python3 tools/bench_attn.py --iters 50 > /dev/null 2> $O/pmc3.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_write -- python3 tools/bench_attn.py --iters 50 > /dev/null 2> $O/pmc4.err
python3 tools/prof_summary.py pmc $O/pmc_attn_fetch $O/pmc_attn_write paged_attn_mfma_kernel 134750336 $O/r02_pmc_attn.json
rm -rf $O/bench $O/cache $O/attn $O/pmc_cache_fetch $O/pmc_cache_write $O/pmc_attn_fetch $O/pmc_attn_write
ls -la $O
