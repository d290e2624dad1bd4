# round 4, call 19: K2 at K = 63 on one box -- which of the day's changes cost the two-word flavour 3 %: the copy search's counting compiled out (adapt0), with that: no static tile
# rounds (opt 1), no hash tags (opt 33), the window by global_load_lds (dma), start bits for 127 k-mers a record (the first kernels' LDS layout); the round's first final library beside them
O=gpurun_out/r4s; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
A=$PWD/soapdenovo2_amd/ab
run head PG_NOP=1
run old4f4618c SOAPDENOVO2_AMD_LIB=$A/lib_4f4618c.so
run adapt0 SOAPDENOVO2_AMD_LIB=$A/lib_adapt0.so
run adapt0_opt1 SOAPDENOVO2_AMD_LIB=$A/lib_adapt0.so PG_K2_OPT=1
run adapt0_opt33 SOAPDENOVO2_AMD_LIB=$A/lib_adapt0.so PG_K2_OPT=33
run adapt0_dma SOAPDENOVO2_AMD_LIB=$A/lib_adapt0_dma.so
run adapt0_dma_opt1 SOAPDENOVO2_AMD_LIB=$A/lib_adapt0_dma.so PG_K2_OPT=1
run adapt0_nmax127 SOAPDENOVO2_AMD_LIB=$A/lib_adapt0_nmax127.so
run adapt0_nmax127_opt1 SOAPDENOVO2_AMD_LIB=$A/lib_adapt0_nmax127.so PG_K2_OPT=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
