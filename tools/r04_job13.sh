#!/bin/bash
# dense twin with 4 waves per workgroup (two workgroups per CU) against 8: parity, the call alone, the encode-only step
set -o pipefail
ulimit -c 0
OUT=gpurun_out/r04_job13; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_varlen_gpu.py -x -q > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
for r in 1 2; do
for w in 4 8; do
  echo "== varlen_dense_waves $w round $r" >> $OUT/attn_ab.txt
  for shape in "--seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64 --encoder" \
               "--seqs 8 --qlen 2048 --heads 16 --kv-heads 16 --head-size 64 --encoder" \
               "--seqs 128 --qlen 128 --heads 16 --kv-heads 16 --head-size 64 --encoder" \
               "--seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64"; do
    LVLLM_VARLEN_DENSE_WAVES=$w timeout -k 10 120 python tools/bench_prefill.py --dense $shape --iters 50 2>/dev/null | grep "hip prefill" >> $OUT/attn_ab.txt || exit 1
  done
done
done
cat $OUT/attn_ab.txt
for r in 1 2; do
for w in 4 8; do
  echo "== varlen_dense_waves $w round $r" >> $OUT/encode_long.txt
  LVLLM_VARLEN_DENSE_WAVES=$w timeout -k 10 300 python tools/bench_encode.py --num-prompts 4096 2>/dev/null | grep -v "^\[" >> $OUT/encode_long.txt || exit 1
done
done
cat $OUT/encode_long.txt
