mkdir -p gpurun_out/r2m
B="python bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2m/$name.log 2>&1; python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/r2m/$name.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
    print("$name", round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),2), "k2", round(r.get("k2_count_ms_per_step",0),1), "parts", r.get("partitions"), "distinct", j["config"]["distinct_kmers"])
except Exception as e: print("$name", "ERR", e, open("gpurun_out/r2m/$name.log").read()[-300:])
PY
}
run base X=1
run k1_nostore PG_K1DBG=1
run k1_noslot PG_K1DBG=2
run k1_R16 PG_K1_R=16
run k1_R24 PG_K1_R=24
run cfg1_p19 PG_K2CFG=1 PG_LOG2_PARTS=19
run cfg0_p19 PG_LOG2_PARTS=19
