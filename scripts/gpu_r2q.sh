mkdir -p gpurun_out/r2q
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "count_matches or ragged or cli_matches or route or sharded_pass1 or corner" > gpurun_out/r2q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2q/pytest.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r2q/pytest.log | tail -3
B="python bench.py --reads 20000000 --genome 10000000 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for i in 1 2; do $B 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; print('b20', round(j['ms_per_step'],2), 'k1', round(r['k1_scatter_ms_per_step'],2), 'k2', round(r['k2_count_ms_per_step'],2), 'distinct', j['config']['distinct_kmers'])"; done
python bench.py --kmer 31 --read-len 100 --genome 4600000 --reads 10000000 --err 0.005 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; print('k31', round(j['ms_per_step'],2), 'k1', round(r['k1_scatter_ms_per_step'],2), 'k2', round(r['k2_count_ms_per_step'],2), 'distinct', j['config']['distinct_kmers'])"
