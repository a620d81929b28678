#!/bin/bash
# fp8-cache decode attention: register sets per wave (variants/kv8n{2,3,4}: -DLVLLM_ATTN_NBUF_KV8) x waves per
# workgroup (LVLLM_ATTN_WAVES_FP8 = 8 | 16), alternating on one box; the bf16 kernel beside it.  Run on the GPU box.
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out/ab_fp8_attn.txt
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
: > $O
for round in 1 2; do
  for v in kv8n2 kv8n3 kv8n4; do
    cp variants/$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
    for w in 8 16; do
      for shape in "--bs 32 --seq 1024" "--bs 64 --seq 2048 --ncaches 6" "--bs 8 --seq 1024"; do
        echo "== $v waves=$w $shape (round $round)" >> $O
        LVLLM_ATTN_WAVES_FP8=$w timeout -k 10 120 python tools/bench_attn.py --kv fp8 --iters 256 $shape 2>&1 | grep -E "^v2" >> $O
      done
    done
  done
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
echo "== bf16 (default build)" >> $O
timeout -k 10 120 python tools/bench_attn.py --iters 256 2>&1 | grep -E "^v2" >> $O
cat $O
