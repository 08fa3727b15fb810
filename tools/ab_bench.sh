#!/bin/bash
# Same-box A/B of two builds of libmsk144hip.so (boxes of the pool differ by +-4 %, so only this counts):
#   tools/ab_bench.sh <other.so> [rounds]   -> alternates `bench.py --no-cpu-baseline` between the tree's library and <other.so>
OTHER=$1
N=${2:-3}
for i in $(seq 1 $N); do
  for lib in "" "$OTHER"; do
    MSK144HIP_LIBRARY=$lib python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-tree}'.split('/')[-1], round(d['ms_per_step'],3), d['stage_ms'])"
  done
done
