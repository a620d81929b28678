#!/bin/bash
# decode attention prologue: kernel arguments in one batch + the first block numbers beside seq_len (LVLLM_ATTN_EARLY_ARGS): parity, then A/B
set -o pipefail
ulimit -c 0
OUT=gpurun_out/r04_job20; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fp8_kv.py tests/test_golden_gpu.py tests/test_paged_attention_class_gpu.py -x -q > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
for round in 1 2; do
  for v in early1 early0; do
    cp variants/$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
    for kv in auto fp8; do
      for shape in "--bs 32 --seq 1024" "--bs 64 --seq 2048 --ncaches 6" "--bs 8 --seq 4096"; do
        echo "== $v kv=$kv $shape (round $round)" >> $OUT/ab.txt
        timeout -k 10 120 python tools/bench_attn.py --rope --kv $kv --iters 256 --contiguous --block-pad 1024 $shape 2>&1 | grep -E "^ *(v1|v2)" | cut -c1-150 >> $OUT/ab.txt
      done
    done
  done
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
cat $OUT/ab.txt
