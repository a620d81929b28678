#!/bin/bash
# Builds attention-kernel variants on the GPU box and times them back to back (one process per
# variant; rank variants only by large differences, confirm close ones interleaved).
#   usage: tools/tune_attn.sh "<flags variant 1>" "<flags variant 2>" ...
set -e
cd "$(dirname "$0")/.."
for flags in "$@"; do
  echo "=== variant: $flags"
  rm -f build/obj/attention*.o
  LVLLM_EXTRA_HIPCC_FLAGS="-DLVLLM_ATTN_TUNE_ONLY $flags" python -c "
import sys; sys.path.insert(0,'light-vllm_amd')
import build; build.build_kernels()"
  python tools/bench_attn.py --iters 300 ${BENCH_ARGS} 2>&1 | grep -E "^v[12]"
done
rm -f build/obj/attention*.o
