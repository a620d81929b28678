#!/bin/bash
# SQ / TA / TCP counter passes and kernel stats of the decode attention kernel at the metric's shape (bs 32, seq 1024),
# fp8 cache and 16-bit cache side by side.  One --pmc pass per counter group, --kernel-trace only.  Run on the GPU box.
#   KV=fp8|auto  (default: both)
set -o pipefail
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_attn
mkdir -p $O
for kv in ${KV:-fp8 auto}; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
             "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do  # (a TA_* group aborted rocprofv3 on this pool, round 3: left out)
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${kv}_p$i -- python3 tools/bench_attn.py --kv $kv --iters 64 > $O/${kv}_p$i.log 2> $O/${kv}_p$i.err || echo "pass $kv $i failed: $grp"
  done
  python3 tools/prof_summary.py counters paged_attn_mfma_kernel $O/pmc_attn_${kv}.json $(ls -d $O/${kv}_p*/) > /dev/null
  rm -rf $O/${kv}_p*
  timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${kv}_stats -- python3 tools/bench_attn.py --kv $kv --iters 256 > $O/${kv}_stats.log 2> $O/${kv}_stats.err
  python3 tools/prof_summary.py stats $O/${kv}_stats $O/attn_${kv}_kernel_stats.csv > /dev/null
  rm -rf $O/${kv}_stats
done
cat $O/pmc_attn_*.json | head -150
