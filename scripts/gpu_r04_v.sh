# round 4, call 22 (the budget's last minutes): the one-K kernels with the record geometry as constants (17 scalar-register spills fewer at K = 63) against the committed library, parity first
O=gpurun_out/r4v; mkdir -p $O
timeout 200 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or (round3_switches and general-kernels)" > $O/pytest_sub.log 2>&1; echo "pytest subset rc=$?"; tail -1 $O/pytest_sub.log
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
A=$PWD/soapdenovo2_amd/ab
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
run k63_const PG_NOP=1
run k63_head SOAPDENOVO2_AMD_LIB=$A/lib_head.so
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_const PG_NOP=1
run k127_head SOAPDENOVO2_AMD_LIB=$A/lib_head.so
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
