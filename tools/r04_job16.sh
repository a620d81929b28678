#!/bin/bash
# config 3 (steady state dominated: 256 prompts) against the workgroups per projection launch at two steps in flight
set -o pipefail
ulimit -c 0
OUT=gpurun_out/r04_job16; mkdir -p $OUT
for r in 1 2; do
for w in 128 96 160 192 256; do
  echo -n "gemm_workgroups $w round $r: " >> $OUT/sweep.txt
  timeout -k 10 200 python tools/bench_chunked_prefill.py --num-prompts 256 --gemm-workgroups $w 2>/dev/null | grep Throughput >> $OUT/sweep.txt || exit 1
done
done
cat $OUT/sweep.txt
