#!/bin/bash
# Round 4, GPU call 7: what the fused rope + cache + attention launch pays over the plain one, by part (diagnosis builds,
# WRONG results): bs 32 x 1 024, 16-bit and fp8 caches, alternating on one box.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_job7; mkdir -p $O
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
: > $O/rope_diag.txt
for round in 1 2; do
  for v in base ropediag1 ropediag8 ropediag2 ropediag4 ropediag15; do
    cp variants/$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
    for kv in auto fp8; do
      echo -n "$v kv=$kv round $round: " >> $O/rope_diag.txt
      timeout -k 10 120 python3 tools/bench_attn.py --rope --kv $kv --iters 256 --contiguous --block-pad 1024 --bs 32 --seq 1024 2>&1 | grep -E "^v2:|rope" | sed 's/median.*trains of 32: //; s/ per launch.*//' | tr '\n' ' ' >> $O/rope_diag.txt
      echo >> $O/rope_diag.txt
    done
  done
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
cat $O/rope_diag.txt
