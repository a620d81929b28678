#!/bin/bash
# Diagnosis builds of the weight-streaming GEMM: build/trace_e<N>/liblvllm_hip.so with
# -DLVLLM_GEMM_TRACE -DLVLLM_GEMM_EXP=<N>  (0 = shipped kernel, 2 = no weight stream; the variants 1 = no
# activation loads and 3 = activations first of profiles/r01_tuning.md belonged to the fragment-order
# activation path that the trace replaced; EXTRA=-DLVLLM_GEMM_XORDER=1 requests the weights before X);
# tools/trace_gemm.py reads the timestamps.
set -e
cd "$(dirname "$0")/.."
objs=$(ls build/obj/*.o | grep -v skinny_gemm)
for e in "$@"; do
  d=build/trace_e$e; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLVLLM_GEMM_TRACE \
    -DLVLLM_GEMM_EXP=$e $EXTRA -c light-vllm_amd/csrc/skinny_gemm.hip -o $d/skinny_gemm.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblvllm_hip.so $objs $d/skinny_gemm.o
done
