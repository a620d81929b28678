#!/bin/bash
# KV blocks padded apart (CacheConfig.block_pad_bytes) against the reference's dense layout: the attention microbench
# over block tables as a block manager hands them out at prefill (--contiguous) and bench.py's headline region,
# alternating on one box.
ulimit -c 0
cd "$(dirname "$0")/.."
O=gpurun_out/r03_ab_block_pad.txt
: > $O
echo "== tools/bench_attn.py, bs 32 x seq 1024, bf16 cache (trains of 32 launches, 10 caches)" >> $O
for opts in "" "--contiguous" "--contiguous --block-pad 128" "--contiguous --block-pad 256" "--contiguous --block-pad 512" \
            "--contiguous --block-pad 1024" "--contiguous --block-pad 2048" "--contiguous --block-pad 4096" "--block-pad 1024"; do
  echo "-- ${opts:-random block tables, dense blocks}" >> $O
  timeout -k 10 120 python tools/bench_attn.py --iters 256 $opts 2>&1 | grep -E "^v2" >> $O
done
echo "== bs 64 x seq 2048" >> $O
for opts in "" "--contiguous" "--contiguous --block-pad 1024"; do
  echo "-- ${opts:-random block tables, dense blocks}" >> $O
  timeout -k 10 120 python tools/bench_attn.py --iters 128 --bs 64 --seq 2048 --ncaches 6 $opts 2>&1 | grep -E "^v2" >> $O
done
echo "== fp8 cache, bs 32 x seq 1024" >> $O
for opts in "" "--contiguous" "--contiguous --block-pad 1024"; do
  echo "-- ${opts:-random block tables, dense blocks}" >> $O
  timeout -k 10 120 python tools/bench_attn.py --kv fp8 --iters 256 $opts 2>&1 | grep -E "^v2" >> $O
done
echo "== bench.py --skip-other-configs (value, ms per step, roofline of the attention leg)" >> $O
for round in 1 2; do
  for pad in 0 1024; do
    echo "-- block_pad_bytes $pad (round $round)" >> $O
    timeout -k 10 400 python bench.py --skip-other-configs --kv-block-pad-bytes $pad 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print(d['value'], 'tokens/s', d['ms_per_step'], 'ms/step; attention', r['avg_launch_us'], 'us per launch =', r['frac'], 'of 8 TB/s; three in flight:', d['other_settings']['max_num_on_the_fly=3']['value'])" >> $O
  done
done
cat $O
