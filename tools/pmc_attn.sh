#!/bin/bash
# SQ / TA / TCP counter passes and kernel stats of the decode attention kernel at the metric's shape (bs 32, seq 1024),
# fp8 cache and 16-bit cache side by side.  One --pmc pass per counter group, --kernel-trace only.  Run on the GPU box.
#   KV=fp8|auto  (default: both)
set -o pipefail
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_attn
mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1 || true
for kv in ${KV:-fp8 auto}; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
             "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_LDS" \
             "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES TCP_TCC_NC_READ_REQ_sum" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${kv}_p$i -- python3 tools/bench_attn.py --kv $kv --iters 64 > $O/${kv}_p$i.log 2> $O/${kv}_p$i.err || echo "pass $kv $i failed: $grp"
  done
  python3 tools/prof_summary.py counters paged_attn_mfma_kernel $O/pmc_attn_${kv}.json $O/${kv}_p[0-9]* > /dev/null
  rm -rf $O/${kv}_p[0-9]*
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${kv}_stats -- python3 tools/bench_attn.py --kv $kv --iters 256 > $O/${kv}_stats.log 2> $O/${kv}_stats.err
  python3 tools/prof_summary.py stats $O/${kv}_stats $O/attn_${kv}_kernel_stats.csv > /dev/null
  rm -rf $O/${kv}_stats
done
cat $O/pmc_attn_*.json | head -150
