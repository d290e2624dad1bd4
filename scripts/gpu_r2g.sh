mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "count_matches or growth or ragged or cli_matches or full_size or sharded_pass1 or sort_records or corner" > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g/pytest.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r2g/pytest.log | tail -5
B="python bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
$B > gpurun_out/r2g/b20.log 2>&1
PG_DBG=2 $B 2>&1 | grep "K2 phase" | head -10
PG_SORT_VERBOSE=1 timeout 1200 python bench.py --no-cpu-baseline > gpurun_out/r2g/bench_default.log 2> gpurun_out/r2g/bench_default.err; echo "bench rc=$?"
grep "\[sort\]" gpurun_out/r2g/bench_default.err
python - <<PY
import json
for f in ("b20","bench_default"):
    try:
        l=[x for x in open(f"gpurun_out/r2g/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{}); w=j.get("whole_command",{})
        print(f, round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3))
        print("   hand_over", j.get("pass1_hand_over"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r2g/bench_default.err
