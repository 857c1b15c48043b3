#!/bin/bash
# Dev tool (GPU box): the default bench line with two builds of the library on the SAME box, alternating (box-to-box spread is 3-5 %, a kernel change is often 1 %).
#   tools/ab_bench.sh <prev .so> [bench args]      -- the current build is retrieval-fuse_amd/rfuse/librfuse_hip.so
PREV=$1; shift
for i in 1 2 3; do
  for tag in prev cur; do
    if [ $tag = prev ]; then export RFUSE_LIB=$PREV; else unset RFUSE_LIB; fi
    python bench.py --no-extras --no-cpu-baseline --steps 60 --repeats 2 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1), [round(x,3) for x in d['blocks']['ms_per_step']])"
  done
done
