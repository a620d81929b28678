#!/bin/bash
# Round 4, GPU call 8: the new token's rows one rotation pair per lane (fused rope + cache + attention launch): parity, A/B.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_job8; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_skinny_gemm_gpu.py tests/test_fp8_kv.py tests/test_padded_cache_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "rope or engine or fused or padded or burst or multi_step or mixed or tokens" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/ab_rope_attn.sh tilestash base > /dev/null 2>&1; cp gpurun_out/ab_rope_attn.txt $O/ab_rope_attn.txt
sed 's/median.*trains of 32: //; s/ per launch.*//' $O/ab_rope_attn.txt | paste - - 
