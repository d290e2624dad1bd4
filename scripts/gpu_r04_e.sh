# round 4, call 5: the GPU suite with the edge builder's waypoints; the 60 M-read K = 127 command again (23.5 s of serial edge walks before); kernel stats of the
# 60 M-read K = 63 command (listings through workgroup batches, pre-arcs counted at read-out); more partitions than 2^21 for K = 63 (2^22 ids, 60 / 70 % in use);
# the 127-mer presplit threshold
O=gpurun_out/r4e; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest.log | tail -8
D=/tmp/pgbig127
timeout 900 python scripts/big_cli_check.py --reads 60000000 --kmer 127 --out $D --keep-fastq --expect profiles/r04_ref_60M_K127.json > $O/k127_60M_cli.log 2>&1; echo "k127 60M cli rc=$?"
timeout 900 python scripts/big_cli_check.py --reads 60000000 --kmer 127 --out $D --keep-fastq --expect profiles/r04_ref_60M_K127.json --tag _nowp --env SOAPDENOVO2_AMD_EB_WAYPOINTS=0 > $O/k127_60M_cli_nowp.log 2>&1; echo "k127 60M cli, no waypoints rc=$?"
mkdir -p $O/k127; cp $D/result*.json $D/stderr*.txt $O/k127/ 2>/dev/null; rm -rf $D
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/k127/result*.json")):
    j = json.load(open(f)); print(f, 'rc', j['rc'], 'wall', j['wall_s'], 'identical', j.get('identical_to_reference')); print('   ', [l for l in j['log'] if '[cli]' in l and ' at ' not in l][-6:], [l for l in j['log'] if 'edges' in l][-3:])
PY
D=/tmp/pgbig60
C="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $C --expect profiles/r03_ref_60M_K63.json --tag _warm > $O/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $C --expect profiles/r03_ref_60M_K63.json --tag _stats --rocprof "--kernel-trace --stats" > $O/stats.log 2>&1; echo "big60 stats rc=$?"
timeout 900 python scripts/big_cli_check.py $C --expect profiles/r03_ref_60M_K63.json --tag _wp512 --env SOAPDENOVO2_AMD_EB_WAYPOINTS=512 > $O/wp.log 2>&1; echo "big60 waypoints forced rc=$?"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16 --env SOAPDENOVO2_AMD_DEVICES=0,0,0 > $O/s3.log 2>&1; echo "big60 -a 16 three ranks rc=$?"
rm -rf $D/reads.fq
mkdir -p $O/big60; cp $D/result*.json $D/stderr*.txt $O/big60/ 2>/dev/null
f=$(find $D/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/big60/kernel_stats_60M.csv && head -12 $f | cut -c1-150
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big60/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
    print("   ", [l for l in j["log"] if "[cli]" in l and "rank" not in l and " at " not in l][-7:])
PY
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
run base PG_NOP=1
run p22_eff60 PG_PARTS_SHIFT=1 PG_PARTS_EFF_PCT=60
run p22_eff70 PG_PARTS_SHIFT=1 PG_PARTS_EFF_PCT=70
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_pre45 PG_K2_PRESPLIT_PCT=45
run k127_pre65 PG_K2_PRESPLIT_PCT=65
run k127_nopre PG_K2_OPT=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
