#!/bin/bash
# the driver's command against the burst length (model steps chained on the device per engine step): 20 steps = 4 x 5 (default: the largest divisor <= 8), 2 x 10, 5 x 4, 10 x 2
ulimit -c 0
O=gpurun_out/r04_job18; rm -rf $O; mkdir -p $O
for r in 1 2; do
for k in 8 10 4 2 20; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --num-scheduler-steps $k --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline --skip-other-configs > $O/k$k.json 2> $O/k$k.err || { echo "k $k failed"; tail -3 $O/k$k.err; continue; }
  python -c "
import json; d=json.loads(open('$O/k$k.json').read().strip().splitlines()[-1])
print('num_scheduler_steps $k round $r: %8.1f tok/s  regions %s  burst %s' % (d['value'], ' '.join('%d' % v for v in d['timed_regions']['tokens_per_s']), d['config'].get('num_scheduler_steps')))" | tee -a $O/summary.txt
done
done
