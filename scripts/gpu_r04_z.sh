# round 4, really the last seconds: the same library at K = 127
O=gpurun_out/r4z; mkdir -p $O
timeout 40 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --kmer 127 > $O/bench_k127.json 2> $O/bench_k127.err; echo "bench rc=$?"
python -c "
import json
l=[x for x in open('$O/bench_k127.json') if x.startswith('{')][-1]; j=json.loads(l); r=j['roofline']
print('k127 pass', round(j['ms_per_step'],1), 'k2', round(r['k2_count_ms_per_step'],1), 'frac', round(r['frac'],3), 'ok', j['conservation']['ok'])"
