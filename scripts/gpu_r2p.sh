B="python bench.py --reads 20000000 --genome 10000000 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; print('$name', round(j['ms_per_step'],2), 'k1', round(r['k1_scatter_ms_per_step'],2), 'k2', round(r['k2_count_ms_per_step'],2), 'parts', r['partitions'])"; }
run base X=1
run cfg3_p19 PG_K2CFG=3 PG_LOG2_PARTS=19
run cfg3_p18 PG_K2CFG=3 PG_LOG2_PARTS=18
run cfg3_p20 PG_K2CFG=3 PG_LOG2_PARTS=20
PG_DBG=2 PG_K2CFG=3 PG_LOG2_PARTS=19 $B 2>&1 | grep "K2 phase" | head -12
