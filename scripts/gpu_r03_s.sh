# round 3, call 19: the whole GPU suite; 64-byte record slots (PG_REC_STRIDE=8) -- time and the WRITE_SIZE counter; layout lanes 2 / 8 once more
mkdir -p gpurun_out/r3s
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3s/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3s/pytest.log | tail -5
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3s/$tag.log 2> gpurun_out/r3s/$tag.err; echo "$tag rc=$?"; }
run base PG_NONE=1
run stride8 PG_REC_STRIDE=8
cd /tmp && export TMPDIR=/tmp
PG_REC_STRIDE=8 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3s/pmc_write_s8 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3s/pmc_write_s8.log 2>&1
PG_REC_STRIDE=8 timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3s/pmc_fetch_s8 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3s/pmc_fetch_s8.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/r3s/pmc_write_s8 gpurun_out/r3s/pmc_write_s8.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r3s/pmc_fetch_s8 gpurun_out/r3s/pmc_fetch_s8.json > /dev/null 2>&1
find gpurun_out/r3s -name "*.db" -delete; find gpurun_out/r3s -name "*counter_collection.csv" -delete; find gpurun_out/r3s -name "*agent_info.csv" -delete
D=/tmp/pgbig60
Bc="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $Bc --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3s/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $Bc --expect profiles/r03_ref_60M_K63.json --tag _l2 > gpurun_out/r3s/a.log 2>&1; echo "big60 2 lanes rc=$?"
timeout 900 python scripts/big_cli_check.py $Bc --expect profiles/r03_ref_60M_K63.json --tag _s8 --env PG_REC_STRIDE=8 > gpurun_out/r3s/b.log 2>&1; echo "big60 stride 8 rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3s/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3s/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3s/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        pass
for n in ("pmc_write_s8", "pmc_fetch_s8"):
    try:
        j = json.load(open(f"gpurun_out/r3s/{n}.json"))
        for k, v in j.items():
            if "skm_" in k: print(n, k[:45], {a: round(b * 1024 / 600e6, 1) for a, b in v.items() if a.endswith("SIZE")}, "B/read")
    except Exception as e: print(n, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r3s/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
PY
grep -h "growable sets on device\|^reader:\|cli\] layout" gpurun_out/r3s/big*/stderr*.txt | head -12
