# round 3, call 26: K1 with the window length at compile time (phase B's loops unrolled, loads with immediate offsets): parity, A/B
mkdir -p gpurun_out/r3z
timeout 1500 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or growth_and_batch or ragged or route_then or skm_route or full_size or cli_matches_reference_files or sharded_pass1 or sharded_matches" > gpurun_out/r3z/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3z/pytest.log | tail -5
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3z/$tag.log 2> gpurun_out/r3z/$tag.err; echo "$tag rc=$?"; }
run general PG_K1_W=0
run w48 PG_NONE=1
run general_b PG_K1_W=0
run w48_b PG_NONE=1
PG_K1_W=0 timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3z/k127_general.log 2> gpurun_out/r3z/k127_general.err; echo "k127 general rc=$?"
timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3z/k127_w112.log 2> gpurun_out/r3z/k127_w112.err; echo "k127 w112 rc=$?"
timeout 600 python bench.py --kmer 31 --reads 10000000 --read-len 100 --genome 4600000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3z/k31_w16.log 2> gpurun_out/r3z/k31_w16.err; echo "k31 w16 rc=$?"
PG_K1_W=0 timeout 600 python bench.py --kmer 31 --reads 10000000 --read-len 100 --genome 4600000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3z/k31_general.log 2> gpurun_out/r3z/k31_general.err; echo "k31 general rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3z/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        pass
PY
