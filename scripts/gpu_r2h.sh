mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "count_matches or growth or ragged or cli_matches or full_size or sharded_pass1 or corner" > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r2h/pytest.log | tail -5
B="python bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2h/$name.log 2>&1; python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/r2h/$name.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
    print("$name", round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "distinct", j["config"]["distinct_kmers"])
except Exception as e: print("$name", "ERR", e, open("gpurun_out/r2h/$name.log").read()[-600:])
PY
}
run dedupe X=1
run nodedupe PG_DBG=4
PG_DBG=2 $B 2>&1 | grep "K2 phase" | head -10
# low coverage: same reads over a 1 Gb genome (hardly any copies)
python bench.py --reads 20000000 --genome 1000000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2h/lowcov.log 2>&1
PG_DBG=4 python bench.py --reads 20000000 --genome 1000000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2h/lowcov_nodedupe.log 2>&1
timeout 1200 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > gpurun_out/r2h/bench200.log 2> gpurun_out/r2h/bench200.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("lowcov","lowcov_nodedupe","bench200"):
    try:
        l=[x for x in open(f"gpurun_out/r2h/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
        print(f, round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3), "distinct", j["config"]["distinct_kmers"])
    except Exception as e: print(f, "ERR", e)
PY
