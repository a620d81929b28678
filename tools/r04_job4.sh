#!/bin/bash
# Round 4, GPU call 4: the split-K fold of the QKV projection into the rope + cache-write launch -- parity, then config 3
# (chunked prefill at the reference's benchmark shape) with and without it, alternating.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_job4; mkdir -p $O
step() {
  local name=$1 lim=$2; shift 2
  echo "== $name" | tee -a $O/steps.log
  timeout -k 10 $lim "$@" > $O/$name.log 2> $O/$name.err; local rc=$?
  echo "$name rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping after $name" | tee -a $O/steps.log; exit 1; fi
}
step pytest_fold 600 python3 -m pytest tests/test_skinny_gemm_gpu.py tests/test_engine_gpu.py tests/test_kv_sizing.py -q -m gpu -k "splitk or slabs or fused_decode or mixed or kv_sizing or profile_run"
tail -3 $O/pytest_fold.log
for rnd in 1 2 3; do
  step c3_fold_r$rnd 400 python3 tools/bench_chunked_prefill.py
  step c3_nofold_r$rnd 400 python3 tools/bench_chunked_prefill.py --no-qkv-reduce-in-rope
done
grep -h "tokens/s\|requests/s" $O/c3_*.log | head -20
cat $O/steps.log | grep rc=
