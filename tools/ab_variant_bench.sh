#!/bin/bash
# bench.py with the shipped library against variants/<name>/liblvllm_hip.so, alternating on one box (diagnosis builds
# included: the token counts are checked, the tokens are not).  usage: tools/ab_variant_bench.sh <name> [rounds] [flags]
ulimit -c 0
cd "$(dirname "$0")/.."
v=$1; n=${2:-2}; shift; shift
O=gpurun_out/ab_variant_$v.txt; : > $O
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
F="--skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline --skip-other-configs --steps 128 --warmup 16 $*"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'], 'in flight 3:', d['other_settings']['max_num_on_the_fly=3']['value'], 'proj', {k: v['us'] for k, v in d.get('roofline_projections', {}).get('per_shape', {}).items()})"; }
for i in $(seq $n); do
  cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
  python bench.py $F 2>/dev/null | line base >> $O
  cp variants/$v/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
  python bench.py $F 2>/dev/null | line $v >> $O
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
cat $O
