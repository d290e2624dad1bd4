mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "count_matches or growth or ragged or cli_matches or full_size or sharded_pass1 or corner or option_semantics" > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r2l/pytest.log | tail -3
B="python bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
$B > gpurun_out/r2l/b20.log 2>&1
PG_DBG=2 $B 2>&1 | grep "K2 phase" | head -10
timeout 1200 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/r2l/bench200.log 2> gpurun_out/r2l/bench200.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("b20","bench200"):
    try:
        l=[x for x in open(f"gpurun_out/r2l/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
        print(f, round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3), "distinct", j["config"]["distinct_kmers"])
    except Exception as e: print(f, "ERR", e)
PY
