# round 4, call 20: K2 at K = 63 on one box: the copy search's counting and the static tile rounds compiled out (a0s0), the same with the window by global_load_lds, without hash
# tags; the round's first final library beside them
O=gpurun_out/r4t; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
A=$PWD/soapdenovo2_amd/ab
run old4f4618c SOAPDENOVO2_AMD_LIB=$A/lib_4f4618c.so
run a0s0 SOAPDENOVO2_AMD_LIB=$A/lib_a0s0.so
run a0s0_dma SOAPDENOVO2_AMD_LIB=$A/lib_a0s0_dma.so
run a0s0_notags SOAPDENOVO2_AMD_LIB=$A/lib_a0s0.so PG_K2_OPT=33
run a0s0_dma_notags SOAPDENOVO2_AMD_LIB=$A/lib_a0s0_dma.so PG_K2_OPT=33
run a0s0_again SOAPDENOVO2_AMD_LIB=$A/lib_a0s0.so
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
