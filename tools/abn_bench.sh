#!/bin/bash
# Dev tool (GPU box): the default bench line with several builds of the library on the SAME box, alternating (box-to-box spread is 3-7 %).
#   tools/abn_bench.sh "<tag>=<lib.so> <tag>=<lib.so> ..." [bench args]     -- lib "cur" = the in-tree build
LIBS=$1; shift
for i in 1 2 3; do
  for kv in $LIBS; do
    tag=${kv%%=*}; lib=${kv#*=}
    if [ "$lib" = cur ]; then unset RFUSE_LIB; else export RFUSE_LIB=$lib; fi
    python bench.py --no-extras --no-cpu-baseline --steps 60 --repeats 2 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1), [round(x,3) for x in d['blocks']['ms_per_step']], 'unpipelined', round(d['unpipelined']['ms_per_step'],3))"
  done
done
