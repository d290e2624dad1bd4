# round 3, call 20: K1's records stored by four lanes a record out of an LDS staging area -- parity cases, time, WRITE_SIZE; 48- and 64-byte slots
mkdir -p gpurun_out/r3t
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or growth_and_batch or ragged or route_then or skm_route or full_size or sharded_pass1 or cli_matches_reference_files" > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3t/pytest.log | tail -5
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3t/$tag.log 2> gpurun_out/r3t/$tag.err; echo "$tag rc=$?"; }
run direct PG_K1_STAGE=0
run staged PG_NONE=1
run staged_s8 PG_REC_STRIDE=8
timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3t/k127.log 2> gpurun_out/r3t/k127.err; echo "k127 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3t/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3t/pmc_write.log 2>&1
PG_REC_STRIDE=8 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3t/pmc_write_s8 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3t/pmc_write_s8.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/r3t/pmc_write gpurun_out/r3t/pmc_write.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r3t/pmc_write_s8 gpurun_out/r3t/pmc_write_s8.json > /dev/null 2>&1
find gpurun_out/r3t -name "*.db" -delete; find gpurun_out/r3t -name "*counter_collection.csv" -delete; find gpurun_out/r3t -name "*agent_info.csv" -delete
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3t/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        pass
for n in ("pmc_write", "pmc_write_s8"):
    try:
        j = json.load(open(f"gpurun_out/r3t/{n}.json"))
        for k, v in j.items():
            if "skm_" in k: print(n, k[:45], {a: round(b * 1024 / 600e6, 1) for a, b in v.items() if a.endswith("SIZE")}, "B/read")
    except Exception as e: print(n, "ERR", e)
PY
