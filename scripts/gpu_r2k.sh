mkdir -p gpurun_out/r2k
timeout 1200 python bench.py --kmer 127 --no-cpu-baseline --whole-reads 0 --no-extras > gpurun_out/r2k/bench_k127.log 2> gpurun_out/r2k/bench_k127.err; echo "k127 rc=$?"
PG_DBG=2 python bench.py --kmer 127 --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "K2 phase" | head -10
PG_LOG2_PARTS=21 timeout 1200 python bench.py --kmer 127 --no-cpu-baseline --whole-reads 0 --no-extras --steps 2 > gpurun_out/r2k/bench_k127_p21.log 2>&1
PG_LOG2_PARTS=22 timeout 1200 python bench.py --kmer 127 --no-cpu-baseline --whole-reads 0 --no-extras --steps 2 > gpurun_out/r2k/bench_k127_p22.log 2>&1
python - <<PY
import json
for f in ("bench_k127","bench_k127_p21","bench_k127_p22"):
    try:
        l=[x for x in open(f"gpurun_out/r2k/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
        print(f, "value", round(j["value"]/1e6,1), "M reads/s", round(j["ms_per_step"],1), "ms k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "parts", r.get("partitions"), "rec/read", r.get("records_per_read"))
    except Exception as e: print(f, "ERR", e)
PY
