#!/bin/bash
# Round 4, GPU call 3: the whole GPU suite; BASELINE config 5 (fp8 weights + fp8 KV cache) against the workgroups per
# decode GEMM at two steps in flight.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_job3; mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  echo "== $name" | tee -a $O/steps.log
  timeout -k 10 $lim "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "$name rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping after $name" | tee -a $O/steps.log; exit 1; fi
}
step pytest_gpu 1100 python3 -m pytest tests -q -m gpu
tail -4 $O/pytest_gpu.log
COMMON="--skip-cpu-baseline --skip-ops-baseline --skip-other-configs --skip-prefill-roofline --also-on-the-fly 0"
for rnd in 1 2; do
  for g in 128 96 64 160; do
    step fp8_g${g}_r$rnd 300 python3 bench.py --quantization fp8 --kv-cache-dtype fp8 --gemm-workgroups $g $COMMON
    python3 - <<PY >> $O/fp8_sweep.txt
import json
d=json.loads(open("$O/fp8_g${g}_r$rnd.log").read().strip().splitlines()[-1])
print("fp8+fp8kv gemm_workgroups $g round $rnd:", d["value"], "tok/s", d["ms_per_step"], "ms/step", {k:v["us"] for k,v in d["roofline_projections"]["per_shape"].items()})
PY
  done
done
cat $O/fp8_sweep.txt
cat $O/steps.log | grep rc=
