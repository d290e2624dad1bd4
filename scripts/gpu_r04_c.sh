# round 4, call 3: is the slow device context of the driver's round-3 bench run (2.5 - 4 s) a fresh box's first large allocation?  (the 60 M-read command FIRST, on
# a GPU nobody has touched; then again); the GPU suite; K2 with the two stalls gone (the compiler's early use of the prefetched record count, the export ticket
# waited for on the spot); K = 127 partition counts and window size; the sharded command with marks, in another order
O=gpurun_out/r4c; mkdir -p $O
R=$GRAFT_REPO_ROOT
D=/tmp/pgbig60
C="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _first > $O/first.log 2>&1; echo "big60 -a 16 FIRST thing on the box rc=$?"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _second > $O/second.log 2>&1; echo "big60 -a 16 second rc=$?"
timeout 900 python scripts/big_cli_check.py $C --expect profiles/r03_ref_60M_K63.json --tag _third > $O/third.log 2>&1; echo "big60 growable third rc=$?"
grep -h "at .*s: " $D/stderr_first.txt $D/stderr_second.txt $D/stderr_third.txt | grep "sized\|context\|pinned\|files read"
timeout 900 python scripts/big_cli_check.py $C --expect profiles/r03_ref_60M_K63.json --tag _sh2 --env SOAPDENOVO2_AMD_DEVICES=0,0 > $O/s2.log 2>&1; echo "big60 two ranks rc=$?"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16_serial --env SOAPDENOVO2_AMD_DEVICES=0,0,0 --env PG_PIPE_SERIAL=1 > $O/s3s.log 2>&1; echo "big60 -a 16 three ranks, no overlap rc=$?"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16 --env SOAPDENOVO2_AMD_DEVICES=0,0,0 > $O/s3.log 2>&1; echo "big60 -a 16 three ranks rc=$?"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16_serial2 --env SOAPDENOVO2_AMD_DEVICES=0,0,0 --env PG_PIPE_SERIAL=1 > $O/s3s2.log 2>&1; echo "big60 -a 16 three ranks, no overlap, again rc=$?"
rm -rf $D/reads.fq
mkdir -p $O/big60; cp $D/result*.json $D/stderr*.txt $O/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big60/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
    print("   ", [l for l in j["log"] if "[cli]" in l and "rank" not in l and " at " not in l][-7:])
PY
for t in sh2 sh3a16_serial sh3a16 sh3a16_serial2; do echo $t; grep -h "at .*s: " $D/stderr_$t.txt | head -8; done
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest.log | tail -8
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
run base PG_NOP=1
run parts_m1 PG_PARTS_SHIFT=-1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_parts_m1 PG_PARTS_SHIFT=-1
run k127_parts_p1 PG_PARTS_SHIFT=1
run k127_win256 PG_K2_WIN=256
run k127_win256_parts_m1 PG_K2_WIN=256 PG_PARTS_SHIFT=-1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 31 --reads 10000000 --read-len 100 --genome 4600000"
run k31_10M PG_NOP=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"], "valu", r.get("valu_issue_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
PG_DBG=2 timeout 300 python bench.py --kmer 63 --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "K2 phase" | tail -12 > $O/k2_phase_cycles_20M_k63.txt
cat $O/k2_phase_cycles_20M_k63.txt
