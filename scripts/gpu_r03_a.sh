# round 3, call 1: the GPU tests (K6 layout through the -a fixtures), the 60 M-read command against the reference's md5s
# (profiles/r03_ref_60M_K63.json) with a kernel trace of the graph stages, the same with -a 16 (K6), a bench line without the CPU leg
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3a/pytest.log
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _plain > gpurun_out/r3a/big60_plain.log 2>&1; echo "big60 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _trace --rocprof "--kernel-trace --stats" > gpurun_out/r3a/big60_trace.log 2>&1; echo "big60 trace rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --unverified --tag _a16 > gpurun_out/r3a/big60_a16.log 2>&1; echo "big60 -a 16 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --unverified --tag _a16host --env SOAPDENOVO2_AMD_LAYOUT=host > gpurun_out/r3a/big60_a16host.log 2>&1; echo "big60 -a 16 host layout rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _fetch --rocprof "--pmc FETCH_SIZE" > gpurun_out/r3a/big60_fetch.log 2>&1; echo "big60 fetch rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _write --rocprof "--pmc WRITE_SIZE" > gpurun_out/r3a/big60_write.log 2>&1; echo "big60 write rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3a/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3a/big60/ 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.log 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
for f in $(find $D -name "*kernel_stats.csv"); do cp $f gpurun_out/r3a/graph_kernel_stats_60M.csv; done
python scripts/pmc_summary.py $D/prof_fetch gpurun_out/r3a/pmc_fetch.json > /dev/null 2>&1
python scripts/pmc_summary.py $D/prof_write gpurun_out/r3a/pmc_write.json > /dev/null 2>&1
find gpurun_out/r3a -name "*.db" -delete; find gpurun_out/r3a -name "*counter_collection.csv" -delete; find gpurun_out/r3a -name "*kernel_trace.csv" -delete; find gpurun_out/r3a -name "*agent_info.csv" -delete
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3a/big60/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"), j.get("md5", {}).get("edge"))
    for l in j["log"]:
        if l.startswith("replay set") or l.startswith("grow "): continue
        print("    ", l)
try:
    l = [x for x in open("gpurun_out/r3a/bench.log") if x.startswith("{")][-1]; j = json.loads(l); r = j.get("roofline", {})
    print("bench", round(j["ms_per_step"], 1), "k1", round(r.get("k1_scatter_ms_per_step", 0), 1), "k2", round(r.get("k2_count_ms_per_step", 0), 1), "frac", round(r.get("frac", 0), 3))
    print("   conservation", j.get("conservation"))
except Exception as e: print("bench ERR", e)
PY
head -40 gpurun_out/r3a/graph_kernel_stats_60M.csv
tail -5 gpurun_out/r3a/bench.err
