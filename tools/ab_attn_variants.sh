#!/bin/bash
# Decode attention: builds of liblvllm_hip.so under variants/<name>/ alternating on one box (tools/bench_attn.py, fp8 and
# 16-bit caches, the metric's shape and two others).  usage: tools/ab_attn_variants.sh <name> <name> ...   (GPU box)
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out/ab_attn_variants.txt
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
: > $O
for round in 1 2; do
  for v in "$@"; do
    cp variants/$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
    for kv in fp8 auto; do
      for shape in "--bs 32 --seq 1024" "--bs 64 --seq 2048 --ncaches 6" "--bs 8 --seq 4096" "--bs 32 --seq 1000"; do
        echo "== $v kv=$kv $shape (round $round)" >> $O
        timeout -k 10 120 python tools/bench_attn.py --kv $kv --iters 256 $shape 2>&1 | grep -E "^v2" >> $O
      done
    done
  done
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
grep -A1 "seq 1024" $O | grep -v "^--"
