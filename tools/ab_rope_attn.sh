#!/bin/bash
# The in-step attention launch (rotation + cache write + attention): builds under variants/<name>/ alternating on one
# box, over consecutive blocks padded apart (what the engine's caches look like).  usage: tools/ab_rope_attn.sh <name> ...
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out/ab_rope_attn.txt
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
: > $O
for round in 1 2; do
  for v in "$@"; do
    cp variants/$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
    for kv in auto fp8; do
      for shape in "--bs 32 --seq 1024" "--bs 64 --seq 2048 --ncaches 6" "--bs 8 --seq 4096"; do
        echo "== $v kv=$kv $shape (round $round)" >> $O
        timeout -k 10 120 python tools/bench_attn.py --rope --kv $kv --iters 256 --contiguous --block-pad 1024 $shape 2>&1 | grep -E "rope" | cut -c1-170 >> $O
      done
    done
  done
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
cat $O
