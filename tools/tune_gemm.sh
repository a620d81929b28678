#!/bin/bash
# usage: tools/tune_gemm.sh "<flags variant 1>" "<flags variant 2>" ...
set -e
cd "$(dirname "$0")/.."
for flags in "$@"; do
  echo "=== variant: $flags"
  rm -f build/obj/skinny_gemm.o
  LVLLM_EXTRA_HIPCC_FLAGS="$flags" python -c "
import sys; sys.path.insert(0,'light-vllm_amd')
import build; build.build_kernels()"
  python tools/bench_gemm.py 2>&1 | grep -E "N="
done
rm -f build/obj/skinny_gemm.o
