mkdir -p gpurun_out/ab
for v in 8 2 4 16 32 8; do
  PG_K2_WG_PER_CU=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/ab/c_$v.log 2>gpurun_out/ab/c_$v.err; echo "rc=$?"
  python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/ab/c_$v.log") if x.startswith("{")][-1]; j=json.loads(l); r=j["roofline"]
    print("wg/cu=$v", round(j["ms_per_step"],1), "k1", round(r["k1_scatter_ms_per_step"],1), "k2", round(r["k2_count_ms_per_step"],1), "frac", round(r["frac"],3))
except Exception as e: print("ERR", e)
PY
done
