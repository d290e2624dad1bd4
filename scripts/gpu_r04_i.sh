# round 4, call 9: (1) the start-up of the 60 M-read command step by step (PG_STARTUP_TRACE), first run right behind the generator and two more;
# (2) K2 as two workgroups of 512 lanes a CU (half the set, half the window, twice the partitions) against one of 1024, K = 63 and K = 127
O=gpurun_out/r4i; mkdir -p $O
D=/tmp/pgbig60
C="--reads 60000000 --a-gb 16 --out $D --keep-fastq --expect profiles/r03_ref_60M_K63_a16.json --env PG_STARTUP_TRACE=1"
for t in 1 2 3; do timeout 600 python scripts/big_cli_check.py $C --tag _$t > $O/run$t.log 2>&1; echo "run $t rc=$?"; done
mkdir -p $O/big60; cp $D/result*.json $D/stderr*.txt $O/big60/ 2>/dev/null; rm -rf $D
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big60/result_*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
for f in sorted(glob.glob("$O/big60/stderr_*.txt")):
    print(f); print("".join(l for l in open(f) if "[ctx]" in l or "[cli]   " in l))
PY
timeout 900 python -m pytest tests -m gpu -x -q -k "round3_switches and two-workgroups" > $O/pytest_cfg1.log 2>&1; echo "pytest cfg1 rc=$?"; tail -3 $O/pytest_cfg1.log
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-k127"
run k63_base PG_NOP=1
run k63_vt1 PG_K2_VT=1
run k63_cfg1 PG_K2CFG=1 PG_PARTS_SHIFT=1
run k63_cfg1_wg4 PG_K2CFG=1 PG_PARTS_SHIFT=1 PG_K2_WG_PER_CU=4
run k63_cfg1_sameparts PG_K2CFG=1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_vt1 PG_K2_VT=1
run k127_cfg1 PG_K2CFG=1 PG_PARTS_SHIFT=1
run k127_cfg1_wg4 PG_K2CFG=1 PG_PARTS_SHIFT=1 PG_K2_WG_PER_CU=4
run k127_cfg1_sameparts PG_K2CFG=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
