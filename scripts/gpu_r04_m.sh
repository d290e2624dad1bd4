# round 4, call 13: the search for exact copies switched off by every workgroup that finds few (first 4096 records, PG_K2_DEDUPE_PCT, default 70 %): K = 127 and K = 63 at the
# headline coverage, K = 63 at half the coverage (the review's configs[3]-per-GPU shape) and at 1/10 (genome 1 Gb), always-on (100) beside it; parity of the K = 127 cases
O=gpurun_out/r4m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or (round3_switches and t6k_k127)" > $O/pytest_sub.log 2>&1; echo "pytest subset rc=$?"; tail -2 $O/pytest_sub.log
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_adaptive PG_NOP=1
run k127_always PG_K2_DEDUPE_PCT=100
run k127_pct85 PG_K2_DEDUPE_PCT=85
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
run k63_adaptive PG_NOP=1
run k63_always PG_K2_DEDUPE_PCT=100
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127 --genome 200000000 --err 0.0005"
run k63_cov150_adaptive PG_NOP=1
run k63_cov150_always PG_K2_DEDUPE_PCT=100
run k63_cov150_never PG_K2_OPT=25
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127 --genome 1000000000 --reads 100000000"
run k63_cov15_adaptive PG_NOP=1
run k63_cov15_always PG_K2_DEDUPE_PCT=100
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
