# round 4, call 10: the 127-mer flavour's K2 with four-word keys (one claim a key) in a 2048-slot set, windows of 192 records, partitions of
# ~4 k occurrences -- against round 4's shape (1024 slots, windows of 512, partitions of ~2 k) with the same keys; parity first
O=gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or (round3_switches and t6k_k127) or device_pass2" > $O/pytest_k127.log 2>&1; echo "pytest k127 subset rc=$?"; tail -3 $O/pytest_k127.log
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_new PG_NOP=1
run k127_new_parts22 PG_PARTS_SHIFT=1
run k127_new_parts20 PG_PARTS_SHIFT=-1
run k127_oldshape PG_K2CFG=4 PG_PARTS_SHIFT=1
run k127_oldshape_parts21 PG_K2CFG=4
run k127_new_pre60 PG_K2_PRESPLIT_PCT=60
run k127_new_pre90 PG_K2_PRESPLIT_PCT=90
run k127_new_nopresplit PG_K2_OPT=1
run k127_new_wg1 PG_K2_WG_PER_CU=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
PG_DBG=2 timeout 300 python bench.py --kmer 127 --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "K2 phase" | tail -12 > $O/k2_phase_cycles_20M_k127.txt
cat $O/k2_phase_cycles_20M_k127.txt
