#!/bin/bash
# Prices the [add + norm -> projection] fusion before anything is built: alternates on one box
#   base      bench.py as shipped
#   skip      the add + norm launch behind o_proj skipped (wrong results): the most a fusion could give
#   fused     skip + the SwiGLU projection normalising its activations in its prologue (variants/fakenorm)
# tokens/s at two (value) and three steps in flight.  Run on the GPU box.
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out/ab_fake_norm.txt
: > $O
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
F="--skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline --skip-other-configs --steps 128 --warmup 16"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'], 'in flight 3:', d['other_settings']['max_num_on_the_fly=3']['value'])"; }
for round in 1 2 3; do
  cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
  python bench.py $F 2>/dev/null | line base >> $O
  python tools/diag_fake_norm.py --skip post $F 2>/dev/null | line skip >> $O
  cp variants/fakenorm/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
  python tools/diag_fake_norm.py --skip post $F 2>/dev/null | line fused >> $O
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
cat $O
