# round 4, call 12: K2 without the search for exact copies of a record (PG_K2_OPT bit 4), K = 127 (few copies: a record is most of a read) and K = 63 (half the records are copies)
O=gpurun_out/r4l; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_nodedupe PG_K2_OPT=21
run k127_nodedupe_parts22 PG_K2_OPT=21 PG_PARTS_SHIFT=1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
run k63_base PG_NOP=1
run k63_nodedupe PG_K2_OPT=25
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
