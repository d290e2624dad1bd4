# round 4, the budget's last seconds: the one-K kernels with the default switches as constants (a third of the spilled scalar registers gone): parity of the counting kernels, one bench
O=gpurun_out/r4y; mkdir -p $O
timeout 60 python -m pytest tests -m gpu -x -q -k "count_matches_oracle" > $O/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_sub.log
timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-k127 > $O/bench_k63.json 2> $O/bench_k63.err; echo "bench rc=$?"
python -c "
import json
l=[x for x in open('$O/bench_k63.json') if x.startswith('{')][-1]; j=json.loads(l); r=j['roofline']
print('k63 pass', round(j['ms_per_step'],1), 'k2', round(r['k2_count_ms_per_step'],1), 'frac', round(r['frac'],3), 'ok', j['conservation']['ok'])"
