# The default bench line N times on one box: run-to-run spread of `value` (and of the three-in-flight region).
ulimit -c 0
N=${1:-4}; O=gpurun_out/${TAG:-r03}_repeat; mkdir -p $O
for i in $(seq 1 $N); do
  timeout -k 10 300 python bench.py --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline --skip-other-configs > $O/run$i.json 2> $O/run$i.err || { echo "run $i did not finish"; exit 1; }
  python -c "
import json; d=json.loads(open('$O/run$i.json').read().strip().splitlines()[-1])
o=d.get('other_settings',{})
print('run $i  %8.1f tok/s  %.4f ms/step  attention %.2f us  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], '  '.join('%s: %s' % (k, v['value']) for k, v in o.items())))"
done
