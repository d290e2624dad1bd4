mkdir -p gpurun_out/r2b
B="python bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2b/$name.log 2>&1; python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/r2b/$name.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
    print("$name", round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "parts", r.get("partitions"), "distinct", j["config"]["distinct_kmers"])
except Exception as e: print("$name", "ERR", e, open("gpurun_out/r2b/$name.log").read()[-600:])
PY
}
run base X=1
run cfg2_p20 PG_K2CFG=2 PG_LOG2_PARTS=20
run cfg2_p19 PG_K2CFG=2 PG_LOG2_PARTS=19
run cfg1_p19 PG_K2CFG=1 PG_LOG2_PARTS=19
run cfg1_p18 PG_K2CFG=1 PG_LOG2_PARTS=18
run cfg0_p17 PG_LOG2_PARTS=17
run cfg0_p19 PG_LOG2_PARTS=19
PG_DBG=2 PG_K2CFG=2 PG_LOG2_PARTS=20 $B 2>&1 | grep "K2 phase" | head -10
PG_DBG=3 $B 2>&1 | grep "K2 phase" | head -10
