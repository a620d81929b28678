#!/bin/bash
# BASELINE config 5 (fp8 W8A8 projections + fp8 KV cache), alternating on one box:
#   once    activations quantised once by their producer (shipped)
#   each    every projection quantises its own activations (round 2)
#   rope    once + rope / cache write / attention in one launch over the fp8 cache
# tokens/s at two (value) and three steps in flight.  Run on the GPU box.
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out/ab_fp8.txt
: > $O
F="--quantization fp8 --kv-cache-dtype fp8 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline --skip-other-configs --steps 128 --warmup 16"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'], 'in flight 3:', d['other_settings']['max_num_on_the_fly=3']['value'], 'attn', d['roofline']['avg_launch_us'], 'proj', d.get('roofline_projections', {}).get('per_shape'))"; }
for round in ${ROUNDS:-1 2}; do
  python bench.py $F 2>/dev/null | line once >> $O
  python bench.py $F --no-fp8-activations-once 2>/dev/null | line each >> $O
  python bench.py $F --no-rope-in-attention-fp8 2>/dev/null | line norope >> $O
done
cat $O
