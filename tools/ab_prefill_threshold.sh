#!/bin/bash
# Old body (prefill_mfma.h) against the 32x32-MFMA body (prefill_mfma32.h) over chunk lengths and cached contexts:
# picks the default of the tuning key prefill_mfma32_min_query.  Run on the GPU box.
cd "$(dirname "$0")/.."
for shape in "1 0 64" "1 0 128" "1 0 256" "1 0 512" "1 0 1024" "1 0 2048" "1 4096 64" "1 4096 128" "1 4096 256" "1 4096 512" "1 4096 1024" \
             "8 0 128" "8 0 256" "8 0 512" "8 0 1024" "8 2048 128" "8 2048 512" "32 1024 64" "32 1024 128" "4 8192 512" "2 0 8192"; do
  set -- $shape
  for m in 0 1; do
    echo -n "seqs $1 ctx $2 qlen $3 mfma32 $m: "
    timeout -k 10 120 python tools/bench_prefill.py --seqs $1 --ctx $2 --qlen $3 --mfma32-min-query $m 2>/dev/null | sed 's/.*median \([0-9.]*\) us.*-> \([0-9.]*\) TFLOP.*/\1 us \2 TF/'
  done
done
