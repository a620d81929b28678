#!/bin/bash
# Alternate build of one kernel file with extra -D flags, linked with the cached objects of the rest:
#   tools/build_variant.sh <name> <file.hip> <flags...>   ->  build/<name>/liblvllm_hip.so
# (copied over light-vllm_amd/lib/ on the GPU box for an A/B run)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
stem=$(basename $src .hip)
d=build/$name; mkdir -p $d
objs=$(ls build/obj/*.o | grep -v "/$stem.o")
extra=""
[ "$stem" = "prefill_attention" ] && extra="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra "$@" \
  -c light-vllm_amd/csrc/$src -o $d/$stem.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/liblvllm_hip.so $objs $d/$stem.o
echo $d/liblvllm_hip.so
