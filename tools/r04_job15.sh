#!/bin/bash
# the 32x32 prefill body with two tiles per trip (a tile's LDS stage a compile-time constant): parity of every prefill path, then A/B
set -o pipefail
ulimit -c 0
OUT=gpurun_out/r04_job15; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_varlen_gpu.py tests/test_prefill_mfma32_gpu.py tests/test_prefill_gpu.py tests/test_prefill_chunk_gpu.py -x -q > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
for r in 1 2; do
for v in pairs1 pairs0; do
  cp variants/pf32_$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
  for shape in "--dense --encoder --seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64" \
               "--dense --encoder --seqs 128 --qlen 128 --heads 16 --kv-heads 16 --head-size 64" \
               "--dense --encoder --seqs 8 --qlen 2048 --heads 16 --kv-heads 16 --head-size 64" \
               "--qlen 4096" "--qlen 16384" "--seqs 8 --qlen 1024" "--seqs 4 --ctx 8192 --qlen 512"; do
    echo -n "$v r$r | " >> $OUT/ab.txt
    timeout -k 10 120 python tools/bench_prefill.py $shape --iters 40 2>/dev/null | grep "hip prefill" | sed 's/hip prefill: //' >> $OUT/ab.txt || { cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so; exit 1; }
  done
done
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
cat $OUT/ab.txt
