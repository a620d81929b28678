#!/bin/bash
# A/B of two builds of liblvllm_hip.so (variants/attn_old, variants/attn_new) through bench.py, alternating on one
# box (run on the GPU box).  Extra bench.py flags: BENCH_FLAGS="--kv-cache-dtype fp8" tools/ab_attn_hoist.sh
cd "$(dirname "$0")/.."
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
for v in attn_old attn_new attn_old attn_new attn_old attn_new; do
  cp variants/$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
  python bench.py --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline $BENCH_FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], 'roofline', d['roofline']['avg_launch_us'], d['roofline']['frac'], 'in flight 3:', d['other_settings']['max_num_on_the_fly=3']['value'])"
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
