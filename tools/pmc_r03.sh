#!/bin/bash
# Round 3 counter records (separate --pmc passes, --kernel-trace only): HBM traffic of the decode projections (16-bit
# and the W8A8 launches with fp8 activations in) and of the attention launch (16-bit and fp8 cache); SQ counters of both
# attention launches as shipped.  Run on the GPU box.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_r03; mkdir -p $O
for mode in "" "--w8"; do
  tag=gemm${mode:+_w8a8}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${tag}_$c -- python3 tools/pmc_gemm.py $mode > $O/${tag}_$c.log 2>&1 || echo "$tag $c failed"
  done
  python3 tools/pmc_gemm_summary.py $(find $O/${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
      $(find $O/${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/r03_pmc_${tag}.json $mode > /dev/null || echo "summary $tag failed"
  rm -rf $O/${tag}_FETCH_SIZE $O/${tag}_WRITE_SIZE
done
for kv in auto fp8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/attn_${kv}_$c -- python3 tools/bench_attn.py --kv $kv --iters 64 > $O/attn_${kv}_$c.log 2>&1 || echo "attn $kv $c failed"
  done
  algo=$([ $kv = fp8 ] && echo 67641472 || echo 134750336)
  python3 tools/prof_summary.py pmc $O/attn_${kv}_FETCH_SIZE $O/attn_${kv}_WRITE_SIZE paged_attn_mfma_kernel $algo $O/r03_pmc_attn_${kv}.json > /dev/null || echo "attn summary $kv failed"
  rm -rf $O/attn_${kv}_FETCH_SIZE $O/attn_${kv}_WRITE_SIZE
done
cat $O/r03_pmc_*.json | grep -E "traffic_over|kernel\"" 
tools/pmc_attn.sh > $O/pmc_attn_sq.log 2>&1; tail -3 $O/pmc_attn_sq.log
