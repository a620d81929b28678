#!/bin/bash
# the driver's command four times on one box: spread of the median-of-five headline, of the attention legs and of configs 3 / 4 / 5
ulimit -c 0
O=gpurun_out/r04_repeat; rm -rf $O; mkdir -p $O
for i in 1 2 3 4; do
  timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu-baseline --skip-ops-baseline > $O/run$i.json 2> $O/run$i.err || { echo "run $i did not finish"; exit 1; }
  python -c "
import json; d=json.loads(open('$O/run$i.json').read().strip().splitlines()[-1])
o=d.get('other_settings',{})
r=d['roofline']
print('run $i  %8.1f tok/s (regions %s)  plain %.2f us = %.3f  in-step %.2f us = %.3f  prefill %.3f' % (d['value'], ' '.join('%d' % v for v in d['timed_regions']['tokens_per_s']), r['avg_launch_us'], r['frac'], r['in_step']['avg_launch_us'], r['in_step']['frac'], d['roofline_prefill']['frac']))
print('        ' + '  '.join('%s: %s' % (k, v['value']) for k, v in o.items()))
c5=o['config5_fp8_weights_fp8_kv']['roofline_attention']
print('        fp8 attention %.2f us = %.3f  in-step %.2f us = %.3f   config 3 short run %s' % (c5['avg_launch_us'], c5['frac'], c5['in_step']['avg_launch_us'], c5['in_step']['frac'], o['config3_chunked_prefill']['short_run']['value']))" | tee -a $O/summary.txt
done
