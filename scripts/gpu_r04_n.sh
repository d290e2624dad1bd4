# round 4, call 14: hash tags in the table of the copy search (PG_K2_OPT bit 5 = without), the two-word flavour's new threshold at 15x, K2's phases at K = 127 without the search
O=gpurun_out/r4n; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
run k63_tags PG_NOP=1
run k63_notags PG_K2_OPT=41
run k63_tags_again PG_NOP=1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127 --genome 1000000000 --reads 100000000"
run k63_cov15_adaptive82 PG_NOP=1
run k63_cov15_always PG_K2_DEDUPE_PCT=100
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_default PG_NOP=1
run k127_always_tags PG_K2_DEDUPE_PCT=100
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
for k in 127 63; do
PG_DBG=2 timeout 300 python bench.py --kmer $k --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-k127 2>&1 | grep "K2 phase" | tail -12 > $O/k2_phase_cycles_20M_k$k.txt
echo "== K = $k"; cat $O/k2_phase_cycles_20M_k$k.txt
done
