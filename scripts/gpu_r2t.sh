B="python bench.py --reads 20000000 --genome 10000000 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
for v in 0 8 0 8; do PG_DBG=$v $B 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; print('dbg=$v', round(j['ms_per_step'],2), 'k1', round(r['k1_scatter_ms_per_step'],2), 'k2', round(r['k2_count_ms_per_step'],2), j['config']['distinct_kmers'])"; done
