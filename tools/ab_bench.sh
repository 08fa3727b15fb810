#!/bin/bash
# Same-box A/B of builds of libmsk144hip.so (boxes of the pool differ by +-4 %, so only this counts):
#   tools/ab_bench.sh <other.so> [rounds]            -> alternates bench.py between the tree's library and <other.so>
#   tools/ab_bench.sh "<a.so> <b.so> ..." [rounds]   -> the tree's library and every listed one, round-robin
OTHERS=$1
N=${2:-3}
for i in $(seq 1 $N); do
  for lib in "" $OTHERS; do
    MSK144HIP_LIBRARY=$lib python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-tree}'.split('/')[-1], round(d['ms_per_step'],3), d['stage_ms'])"
  done
done
