#!/bin/bash
# Timeline build: the instrumented kernels (csrc/trace.h) compiled with -DLVLLM_TRACE and linked with
# the cached objects of everything else -> build/trace_all/liblvllm_hip.so (copied over
# light-vllm_amd/lib/ on the GPU box before tools/trace_step.py runs).
set -e
cd "$(dirname "$0")/.."
d=build/trace_all; mkdir -p $d
files="skinny_gemm attention_bf16 layernorm pos_encoding"
pids=""
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLVLLM_TRACE \
    -c light-vllm_amd/csrc/$f.hip -o $d/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
objs=""
for o in build/obj/*.o; do
  b=$(basename $o .o)
  case " $files " in *" $b "*) objs="$objs $d/$b.o";; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblvllm_hip.so $objs
echo $d/liblvllm_hip.so
