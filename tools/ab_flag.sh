#!/bin/bash
# Alternates `python bench.py` with and without extra flags on one box: tools/ab_flag.sh 5 --no-rope-in-attention
cd "$(dirname "$0")/.."
n=$1; shift
for i in $(seq $n); do
  for f in "" "$*"; do
    python bench.py --skip-cpu-baseline --skip-ops-baseline --skip-prefill-roofline --steps 128 $f 2>/dev/null > /tmp/ab_flag.json
    python - "$f" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_flag.json").read().strip().splitlines()[-1])
print("flags [%s]" % sys.argv[1], d["value"], d["ms_per_step"], "in flight 3:", d["other_settings"]["max_num_on_the_fly=3"]["value"])
PY
  done
done
