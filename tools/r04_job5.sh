#!/bin/bash
# Round 4, GPU call 5: per-workgroup timeline of decode steps across the two streams (instrumented build).
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_job5; mkdir -p $O
cp light-vllm_amd/lib/liblvllm_hip.so /tmp/liblvllm_hip.orig.so
cp variants/trace_all/liblvllm_hip.so light-vllm_amd/lib/liblvllm_hip.so
for fly in 2 3 1; do
  timeout -k 10 300 python3 tools/trace_step.py --on-the-fly $fly --steps 16 --out $O/trace_fly$fly.npz > $O/trace_fly$fly.log 2> $O/trace_fly$fly.err; echo "trace fly $fly rc=$?"
  timeout 120 python3 tools/analyze_trace.py $O/trace_fly$fly.npz 260 > $O/analysis_fly$fly.txt 2>&1
  rm -f $O/trace_fly$fly.npz
done
cp /tmp/liblvllm_hip.orig.so light-vllm_amd/lib/liblvllm_hip.so
tail -40 $O/analysis_fly2.txt
