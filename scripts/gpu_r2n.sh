mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "count_matches or ragged or cli_matches or route" > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2n/pytest.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r2n/pytest.log | tail -3
B="python bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
$B > gpurun_out/r2n/b20.log 2>&1
python - <<PY
import json
for f in ("b20",):
    l=[x for x in open(f"gpurun_out/r2n/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
    print(f, round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),2), "k2", round(r.get("k2_count_ms_per_step",0),1))
PY
