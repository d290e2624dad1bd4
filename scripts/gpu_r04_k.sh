# round 4, call 11: (1) start-up of the 60 M-read command, finer trace: first GPU process on the box, a second run, a run with the context made BEFORE the batch
# buffers (PG_CTX_THREAD=0), a run right behind a process that allocated and released 200 GB; (2) K2 at K = 63: the k-mer's own two words claimed by one
# compare-and-swap against words of 63 bits claimed one by one (two builds of the library), a wave's first tiles static (PG_K2_OPT bit 3); the same bit at K = 127
O=gpurun_out/r4k; mkdir -p $O
D=/tmp/pgbig60
C="--reads 60000000 --a-gb 16 --out $D --keep-fastq --expect profiles/r03_ref_60M_K63_a16.json --env PG_STARTUP_TRACE=1"
timeout 600 python scripts/big_cli_check.py $C --tag _1 > $O/run1.log 2>&1; echo "run 1 rc=$?"
timeout 600 python scripts/big_cli_check.py $C --tag _2 > $O/run2.log 2>&1; echo "run 2 rc=$?"
timeout 600 python scripts/big_cli_check.py $C --env PG_CTX_THREAD=0 --tag _3serial > $O/run3.log 2>&1; echo "run 3 rc=$?"
soapdenovo2_amd/bin/startup_probe 50 > $O/probe_200GB.txt 2>&1
timeout 600 python scripts/big_cli_check.py $C --tag _4behind > $O/run4.log 2>&1; echo "run 4 rc=$?"
sleep 8
timeout 600 python scripts/big_cli_check.py $C --tag _5settled > $O/run5.log 2>&1; echo "run 5 rc=$?"
mkdir -p $O/big60; cp $D/result*.json $D/stderr*.txt $O/big60/ 2>/dev/null; rm -rf $D
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big60/result_*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
for f in sorted(glob.glob("$O/big60/stderr_*.txt")):
    print(f); print("".join(l for l in open(f) if "[ctx]" in l or "[cli]   " in l))
PY
cat $O/probe_200GB.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "count_matches_oracle" > $O/pytest_count.log 2>&1; echo "pytest count rc=$?"; tail -2 $O/pytest_count.log
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
run k63_raw PG_NOP=1
run k63_keys63 SOAPDENOVO2_AMD_LIB=$PWD/soapdenovo2_amd/ab/libsoapdenovo2_amd_keys63.so
run k63_raw_static PG_K2_OPT=9
run k63_keys63_static PG_K2_OPT=9 SOAPDENOVO2_AMD_LIB=$PWD/soapdenovo2_amd/ab/libsoapdenovo2_amd_keys63.so
run k63_raw_again PG_NOP=1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_static PG_K2_OPT=13
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
