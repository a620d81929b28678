ulimit -c 0
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
for cfg in "2 128" "3 128" "3 96" "4 128" "4 64" "2 160" "3 160"; do
  set -- $cfg
  LVLLM_ENGINE_GEMM_WGS=$2 python bench.py --on-the-fly $1 --num-scheduler-steps 8 --steps 96 --warmup 24 --skip-cpu-baseline --skip-ops-baseline > gpurun_out/r02_sweep_$1_$2.log 2> gpurun_out/r02_sweep_$1_$2.err
  python -c "import json; d=json.loads(open('gpurun_out/r02_sweep_$1_$2.log').read().strip().splitlines()[-1]); print('on_the_fly $1 wgs $2:', d['value'], d['ms_per_step'])"
done
