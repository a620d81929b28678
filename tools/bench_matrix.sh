# The bench line of the tree and its A/B settings, one box (numbers of different boxes differ by +-3 %).
ulimit -c 0
O=gpurun_out/r02_matrix; mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name: bench.py did not finish (exit $?)"; return 0; }; python -c "
import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
o=d.get('other_settings',{})
print('%-24s %8.1f tok/s  %.4f ms/step  (%d in flight)   %s' % ('$name', d['value'], d['ms_per_step'], d['config']['max_num_on_the_fly'], '  '.join('%s: %s' % (k.replace('max_num_on_the_fly=', 'in flight '), v['value']) for k, v in o.items())))"; }
run default
run also_4 --also-on-the-fly 4 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run on_the_fly_3 --on-the-fly 3 --also-on-the-fly 0 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run single_step --num-scheduler-steps 1 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run no_rope_in_attention --no-rope-in-attention --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run no_fusion --no-fusion --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run sync --scheduling sync --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run fp8_kv --kv-cache-dtype fp8 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run fp8_w8a8 --quantization fp8 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run fp8_both --quantization fp8 --kv-cache-dtype fp8 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run bs64 --batch-size 64 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
run ctx4096 --context 4096 --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
timeout -k 10 300 python tools/bench_chunked_prefill.py 2>&1 | grep -v amdgpu | tail -1
# last: hipBLASLt under multi-stream graph capture has hung before (library GEMM path); bounded
run library_gemm --library-gemm --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline
