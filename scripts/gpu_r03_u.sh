# round 3, call 21: K1 tile sizes (phase A runs R * 9 tasks on 256 lanes: 32 reads = 288 tasks = a second pass for 32 lanes); what K1 writes without its record stores
mkdir -p gpurun_out/r3u
R=$GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3u/$tag.log 2> gpurun_out/r3u/$tag.err; echo "$tag rc=$?"; }
run r32 PG_NONE=1
run r28 PG_K1_R=28
run r56 PG_K1_R=56
run r24 PG_K1_R=24
run r28_k127 PG_K1_R=28 PG_NONE=1
cd /tmp && export TMPDIR=/tmp
PG_K1DBG=1 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3u/pmc_write_nostore -- python $R/bench.py --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3u/pmc_write_nostore.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3u/pmc_write_20M -- python $R/bench.py --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3u/pmc_write_20M.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/r3u/pmc_write_nostore gpurun_out/r3u/pmc_write_nostore.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r3u/pmc_write_20M gpurun_out/r3u/pmc_write_20M.json > /dev/null 2>&1
find gpurun_out/r3u -name "*.db" -delete; find gpurun_out/r3u -name "*counter_collection.csv" -delete; find gpurun_out/r3u -name "*agent_info.csv" -delete
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3u/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        pass
for n in ("pmc_write_nostore", "pmc_write_20M"):
    try:
        j = json.load(open(f"gpurun_out/r3u/{n}.json"))
        for k, v in j.items():
            if "skm_scatter" in k: print(n, k[:45], {a: round(b * 1024 / 60e6, 1) for a, b in v.items() if a.endswith("SIZE")}, "B/read", v.get("launches"))
    except Exception as e: print(n, "ERR", e)
PY
