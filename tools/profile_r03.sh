#!/bin/bash
# Round 3 records: the default bench line (with BASELINE configs 3 / 4 / 5 beside the headline), kernel stats of the
# same command under rocprofv3, the ROPE instantiation of the attention kernel alone, paged_attention_v2 in the
# reference-partition mode, SQ counter passes of the fp8 and 16-bit attention launches.  Run on the GPU box.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r03; mkdir -p $O
python3 bench.py > $O/r03_bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/r03_bench_line.json
# the attention kernel alone: plain, ROPE instantiation, reference partitions (attn_splits = -1: 2 x 512 + reduce)
python3 tools/bench_attn.py --rope --iters 256 > $O/r03_attn_microbench.txt 2>&1
LVLLM_ATTN_SPLITS=-1 python3 tools/bench_attn.py --iters 256 > $O/r03_attn_microbench_ref_partitions.txt 2>&1
python3 tools/bench_attn.py --kv fp8 --rope --iters 256 > $O/r03_attn_microbench_fp8.txt 2>&1
cat $O/r03_attn_microbench*.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn_stats -- python3 tools/bench_attn.py --rope --iters 256 > /dev/null 2> $O/attn_stats.err
python3 tools/prof_summary.py stats $O/attn_stats $O/r03_attn_rope_kernel_stats.csv > /dev/null; rm -rf $O/attn_stats
LVLLM_ATTN_SPLITS=-1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn_stats -- python3 tools/bench_attn.py --iters 256 > /dev/null 2> $O/attn_stats2.err
python3 tools/prof_summary.py stats $O/attn_stats $O/r03_attn_ref_partitions_kernel_stats.csv > /dev/null; rm -rf $O/attn_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn_stats -- python3 tools/bench_attn.py --kv fp8 --iters 256 > /dev/null 2> $O/attn_stats3.err
python3 tools/prof_summary.py stats $O/attn_stats $O/r03_attn_fp8_kernel_stats.csv > /dev/null; rm -rf $O/attn_stats
head -5 $O/r03_attn_*kernel_stats.csv
