#!/bin/bash
# dense twin (K / V rows read in place) against the pack pass + paged body: the attention call alone, then the encode-only step's profile
set -o pipefail
ulimit -c 0
OUT=gpurun_out/r04_job10; mkdir -p $OUT
for r in 1 2; do
for d in 1 0; do
  echo "== varlen_dense $d round $r" >> $OUT/attn_ab.txt
  for shape in "--seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64 --encoder" \
               "--seqs 8 --qlen 2048 --heads 16 --kv-heads 16 --head-size 64 --encoder" \
               "--seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64" \
               "--seqs 8 --qlen 1024 --heads 32 --kv-heads 8 --head-size 128" \
               "--seqs 8 --qlen 1024 --heads 32 --kv-heads 8 --head-size 128 --encoder"; do
    LVLLM_VARLEN_DENSE=$d timeout -k 10 120 python tools/bench_prefill.py --dense $shape --iters 50 2>/dev/null | grep "hip prefill" >> $OUT/attn_ab.txt || exit 1
  done
done
done
cat $OUT/attn_ab.txt
for d in 1 0; do
  echo "== varlen_dense $d" >> $OUT/encode_long.txt
  LVLLM_VARLEN_DENSE=$d timeout -k 10 300 python tools/bench_encode.py --num-prompts 4096 2>/dev/null | grep -v "^\[" >> $OUT/encode_long.txt || exit 1
done
cat $OUT/encode_long.txt
