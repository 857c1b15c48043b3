for cfg in "--config C4 --batch 16" "--config C1 --batch 16" "--config C3 --db 1000000" "--config C5 --batch 16"; do
  echo "== $cfg"
  for i in 1 2; do
  for kv in base=tools/_haz/librfuse_base.so cur=cur; do
    tag=${kv%%=*}; lib=${kv#*=}
    if [ "$lib" = cur ]; then unset RFUSE_LIB; else export RFUSE_LIB=$lib; fi
    python bench.py --no-extras --no-cpu-baseline --steps 40 --repeats 1 $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],1), [round(x,3) for x in d['blocks']['ms_per_step']], 'unpipelined', round(d['unpipelined']['ms_per_step'],3))"
  done; done
done
