#!/bin/bash
# A/B of the 32x32-MFMA prefill body (prefill_mfma32.h): builds variant libraries with -D flags (in the dev
# container: tools/ab_prefill32.sh build) and times them on the GPU box (tools/ab_prefill32.sh run).
# Diagnosis builds (LVLLM_PREFILL32_DIAG) compute WRONG results; they only price a part of the loop.
cd "$(dirname "$0")/.."
VARIANTS=${VARIANTS:-"base: pp:-DLVLLM_PREFILL32_PINGPONG=1 ppprio:-DLVLLM_PREFILL32_PINGPONG=1;-DLVLLM_PREFILL32_PRIO=1 nolds:-DLVLLM_PREFILL32_DIAG=16 nocopy:-DLVLLM_PREFILL32_DIAG=1 noexp:-DLVLLM_PREFILL32_DIAG=2 mfmaonly:-DLVLLM_PREFILL32_DIAG=31"}
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    name=${v%%:*}; flags=${v#*:}
    tools/build_variant.sh pf32_$name prefill_attention.hip ${flags//;/ } > /dev/null || exit 1
    mkdir -p variants/pf32_$name && cp build/pf32_$name/liblvllm_hip.so variants/pf32_$name/
  done
  exit 0
fi
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
for v in $VARIANTS; do
  name=${v%%:*}
  cp variants/pf32_$name/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
  for shape in "--qlen 4096" "--qlen 16384" "--seqs 8 --qlen 1024"; do
    echo -n "$name | "
    timeout -k 10 120 python tools/bench_prefill.py $shape --mfma32-min-query 1 | sed 's/hip prefill: //'
  done
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
