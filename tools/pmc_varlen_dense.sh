# SQ counter passes over the dense twin of the 32x32 prefill body at bge-m3's launch (32 x 512 tokens, 16 heads of 64,
# bidirectional): LDS bank conflicts of the row-major images, LDS / MFMA / vector busy cycles.  One --pmc pass per counter
# group, --kernel-trace only.  Run on the GPU box.
set -o pipefail
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_varlen_dense
rm -rf $O; mkdir -p $O
for mode in dense pack; do
  i=0
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
    i=$((i+1))
    if [ $mode = dense ]; then export LVLLM_VARLEN_DENSE=1; else export LVLLM_VARLEN_DENSE=0; fi
    LVLLM_VARLEN_DENSE_WAVES=8 timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/${mode}_p$i -- python3 tools/bench_prefill.py --dense --encoder --seqs 32 --qlen 512 --heads 16 --kv-heads 16 --head-size 64 --iters 8 > $O/${mode}_p$i.log 2> $O/${mode}_p$i.err || echo "pass $mode $i failed"
  done
  python3 tools/prof_summary.py counters paged_prefill_mfma32_kernel $O/pmc_${mode}.json $O/${mode}_p[0-9] > /dev/null
  echo "== $mode"; cat $O/pmc_${mode}.json; echo
  rm -rf $O/${mode}_p[0-9]
done
