ulimit -c 0
for i in 1 2; do
for cfg in "" "--o-proj-partials-min-rows 1" "--o-proj-partials-min-rows 1 --gemm-partials-ksplit 2" "--o-proj-partials-min-rows 1 --gemm-partials-ksplit 4" "--gemm-partials-ksplit 8" "--o-proj-partials-min-rows 1 --gemm-partials-ksplit 8"; do
python bench.py --skip-cpu-baseline --skip-ops-baseline --steps 96 --warmup 24 $cfg | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$cfg', d['value'], d['other_settings']['max_num_on_the_fly=3']['value'])"
done; done
