#!/bin/bash
# Round 4, GPU call 2: the whole GPU suite on this round's tree, the encode-only step with its MLP in row blocks,
# per-workgroup timelines of the 32x32-MFMA prefill body at the encoder's shape (32 x 512 tokens, 16 heads of 64) and at
# one 4 096-token prompt.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_job2; mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  echo "== $name" | tee -a $O/steps.log
  timeout -k 10 $lim "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "$name rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping after $name" | tee -a $O/steps.log; exit 1; fi
}
step pytest_gpu 1000 python3 -m pytest tests -x -q -m gpu
tail -3 $O/pytest_gpu.log
for blk in 0 8192 4096 2048 0; do
  step encode_blk$blk 200 python3 tools/bench_encode.py --mlp-block $blk
done
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
cp variants/pf32_stamps2/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
LVLLM_PREFILL32_WG_FILE=$O/wg_encoder.txt step timeline_encoder 120 python3 tools/bench_prefill.py --dense --encoder --seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64 --iters 5
LVLLM_PREFILL32_WG_FILE=$O/wg_4096.txt step timeline_4096 120 python3 tools/bench_prefill.py --qlen 4096 --iters 5
LVLLM_PREFILL32_WG_FILE=$O/wg_8x1024.txt step timeline_8x1024 120 python3 tools/bench_prefill.py --seqs 8 --qlen 1024 --iters 5
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
for f in encoder 4096 8x1024; do python3 tools/wg_timeline_prefill32.py $O/wg_$f.txt > $O/timeline_$f.txt 2>&1; done
step prefill_base_encoder 120 python3 tools/bench_prefill.py --dense --encoder --seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64
cat $O/steps.log
