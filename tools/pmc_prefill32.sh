# SQ counter passes over the 32x32-MFMA prefill body at 1 x 16384 (run on the GPU box): LDS bank conflicts, LDS and
# MFMA busy cycles, what the waves wait for.  One --pmc pass per counter group, --kernel-trace only.
set -o pipefail
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_pf32
mkdir -p $O
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -- python3 tools/bench_prefill.py --qlen ${QLEN:-16384} --iters 8 --mfma32-min-query 1 > $O/p$i.log 2> $O/p$i.err || echo "pass $i failed"
done
python3 tools/prof_summary.py counters paged_prefill_mfma32_kernel $O/pmc_prefill32.json $O/p[0-9] > /dev/null
cat $O/pmc_prefill32.json
rm -rf $O/p[0-9]
# kernel time under the profiler, both headline shapes (the stats csv is what profiles/ keeps)
for q in 4096 16384; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$q -- python3 tools/bench_prefill.py --qlen $q --iters 20 > $O/stats_$q.log 2> $O/stats_$q.err
  python3 tools/prof_summary.py stats $O/stats_$q $O/r02_prefill32_${q}_kernel_stats.csv > /dev/null
  rm -rf $O/stats_$q
done
grep "hip prefill" $O/stats_*.log
