#!/bin/bash
# HBM traffic of the decode attention launch over caches laid out the way the engine's are since round 3 -- consecutive
# blocks per sequence, 1 KiB further apart than their size: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate
# passes (kernel trace only), 2*FETCH_SIZE + WRITE_SIZE per launch against the algorithmic bytes.  Run on the GPU box.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_attn_padded; rm -rf $O; mkdir -p $O
for kv in auto fp8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/attn_${kv}_$c -- python3 tools/bench_attn.py --kv $kv --iters 64 --contiguous --block-pad 1024 > $O/attn_${kv}_$c.log 2>&1 || echo "attn $kv $c failed"
  done
  algo=$([ $kv = fp8 ] && echo 67641472 || echo 134750336)
  python3 tools/prof_summary.py pmc $O/attn_${kv}_FETCH_SIZE $O/attn_${kv}_WRITE_SIZE paged_attn_mfma_kernel $algo $O/r03_pmc_attn_padded_${kv}.json > /dev/null || echo "attn summary $kv failed"
  rm -rf $O/attn_${kv}_FETCH_SIZE $O/attn_${kv}_WRITE_SIZE
done
cat $O/r03_pmc_attn_padded_*.json
