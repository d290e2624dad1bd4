# round 4, call 15: K2 with the next window asked for into registers at the emit's start and written to LDS in front of the export stores (no global_load_lds: the compiler
# guards every LDS write behind one with s_waitcnt vmcnt(0)), nothing looked at or reloaded at a partition's top, the 127-mer put's reads as LDS reads -- against the
# global_load_lds build (-DPG_K2_DMA=1); parity first
O=gpurun_out/r4o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or round3_switches or device_pass2" > $O/pytest_sub.log 2>&1; echo "pytest subset rc=$?"; tail -2 $O/pytest_sub.log
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
AB=$PWD/soapdenovo2_amd/ab/libsoapdenovo2_amd_dma.so
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
run k63_regs PG_NOP=1
run k63_dma SOAPDENOVO2_AMD_LIB=$AB
run k63_regs_again PG_NOP=1
run k63_noprefetch PG_DBG=16 PG_K2_KS=0
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_regs PG_NOP=1
run k127_dma SOAPDENOVO2_AMD_LIB=$AB
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127 --kmer 31 --reads 100000000 --read-len 100 --genome 46000000"
run k31_regs PG_NOP=1
run k31_dma SOAPDENOVO2_AMD_LIB=$AB
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
for k in 63 127; do
PG_DBG=2 timeout 300 python bench.py --kmer $k --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-k127 2>&1 | grep "K2 phase" | tail -12 > $O/k2_phase_cycles_20M_k$k.txt
echo "== K = $k"; cat $O/k2_phase_cycles_20M_k$k.txt
done
