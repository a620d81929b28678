#!/bin/bash
# library GEMMs of the encode-only step: torch's TunableOp picks among the hipBLASLt / rocBLAS solutions per shape; does the pick beat the default heuristic?
set -o pipefail
ulimit -c 0
OUT=gpurun_out/r04_job11; mkdir -p $OUT
export PYTORCH_TUNABLEOP_FILENAME=$PWD/$OUT/tunable_encode.csv
echo "== tuning run" > $OUT/log.txt
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 \
  timeout -k 10 900 python tools/bench_encode.py --num-prompts 256 > $OUT/tune.out 2> $OUT/tune.err || { tail -5 $OUT/tune.err; exit 1; }
grep -v "^\[" $OUT/tune.out | tail -3
ls -la $OUT; cat $OUT/tunable_encode*.csv | cut -c1-200
for r in 1 2; do
  echo "== default heuristic, round $r" >> $OUT/log.txt
  timeout -k 10 300 python tools/bench_encode.py --num-prompts 4096 2>/dev/null | grep -v "^\[" >> $OUT/log.txt || exit 1
  echo "== recorded picks, round $r" >> $OUT/log.txt
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 timeout -k 10 300 python tools/bench_encode.py --num-prompts 4096 2>/dev/null | grep -v "^\[" >> $OUT/log.txt || exit 1
done
cat $OUT/log.txt
