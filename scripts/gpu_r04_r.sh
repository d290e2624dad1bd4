# round 4, call 18: K2 at K = 63 on ONE box, the library as it was at four commits of the day (boxes differ by more than the changes): 4f4618c (the round's first final run),
# e0cbc06 (static first tiles), b0576da (adaptive copy search + hash tags), HEAD (window through registers, nothing waited for at a partition's top)
O=gpurun_out/r4r; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
A=$PWD/soapdenovo2_amd/ab
for rep in 1 2; do
run head_$rep PG_NOP=1
run c4f4618c_$rep SOAPDENOVO2_AMD_LIB=$A/lib_4f4618c.so
run ce0cbc06_$rep SOAPDENOVO2_AMD_LIB=$A/lib_e0cbc06.so
run cb0576da_$rep SOAPDENOVO2_AMD_LIB=$A/lib_b0576da.so
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
