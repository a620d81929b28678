# A/B of prefill-kernel build variants on the GPU box: variants/liblvllm_hip_prefill_<name>.so are swapped in
# for light-vllm_amd/lib/liblvllm_hip.so one at a time (the box's copy of the tree is scratch).
ulimit -c 0
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so light-vllm_amd/lib/liblvllm_hip.so; else cp variants/liblvllm_hip_prefill_$v.so light-vllm_amd/lib/liblvllm_hip.so; fi
  echo "=== $v"
  python tools/bench_prefill.py --qlen 4096 2>&1 | grep "hip prefill"
  python tools/bench_prefill.py --qlen 16384 --iters 10 2>&1 | grep "hip prefill"
  python tools/bench_prefill.py --seqs 8 --qlen 1024 2>&1 | grep "hip prefill"
done
cp /tmp/base.so light-vllm_amd/lib/liblvllm_hip.so
