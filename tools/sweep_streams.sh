# tokens/s of bench.py by steps in flight x GEMM workgroups per launch (multi-step decode, k = 8)
ulimit -c 0
for cfg in ${SWEEP:-"2 128" "3 96" "4 64" "6 48" "8 32"}; do
  set -- $cfg
  python bench.py --gemm-workgroups $2 --on-the-fly $1 --num-scheduler-steps 8 --steps 96 --warmup 24 --skip-cpu-baseline --skip-ops-baseline ${EXTRA:-} > gpurun_out/r02_sweep_$1_$2.log 2> gpurun_out/r02_sweep_$1_$2.err
  python -c "import json; d=json.loads(open('gpurun_out/r02_sweep_$1_$2.log').read().strip().splitlines()[-1]); print('on_the_fly $1 wgs $2:', d['value'], d['ms_per_step'])"
done
