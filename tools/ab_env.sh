#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_env.sh VAR [steps]  ->  ms/step of bench.py with VAR unset / VAR=0, twice each, interleaved
V=$1; S=${2:-8}
for r in 1 2; do
  for val in unset 0; do
    if [ $val = unset ]; then out=$(python3 bench.py --steps $S --warmup 3 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1)
    else out=$(env $V=0 python3 bench.py --steps $S --warmup 3 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1); fi
    echo "$V=$val: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"])')"
  done
done
