#!/bin/bash
# dense twin of the 32x32 prefill body (K / V rows read in place): parity, then the encode-only step with and without
set -o pipefail
ulimit -c 0
OUT=gpurun_out/r04_job9; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_varlen_gpu.py tests/test_prefill_mfma32_gpu.py -x -q > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -2 $OUT/tests.txt
for r in 1 2; do
  for d in 1 0; do
    echo "== varlen_dense $d round $r" >> $OUT/encode_ab.txt
    LVLLM_VARLEN_DENSE=$d timeout -k 10 300 python tools/bench_encode.py >> $OUT/encode_ab.txt 2>&1 || exit 1
  done
done
grep -v "^\[" $OUT/encode_ab.txt | tail -40
