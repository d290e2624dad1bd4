# round 4, call 1: the whole GPU suite on this round's fixes (BAM pairing state, handed-back layout blocks, rocPRIM, reciprocal modulus, batched pass-2 probes),
# A/Bs of two K2 changes (PG_K2_VT=0: occurrences dealt 64 at a time with start bits; PG_K2_OPT=1: live slots listed in any order), the pass-2 block sizes, and the
# 127-mer command against the reference's own run at 20 M reads
O=gpurun_out/r4a; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest.log | tail -5
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
run base PG_NOP=1
run vt0 PG_K2_VT=0
run opt1 PG_K2_OPT=1
run vt0_opt1 PG_K2_VT=0 PG_K2_OPT=1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_vt0_opt1 PG_K2_VT=0 PG_K2_OPT=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"], "distinct", j["config"]["distinct_kmers"])
    except Exception as e:
        print(f, "ERR", e)
PY
# the 127-mer command at 20 M reads against the reference's own files
D=/tmp/pgbig127
timeout 900 python scripts/big_cli_check.py --reads 20000000 --kmer 127 --out $D --expect profiles/r04_ref_20M_K127.json > $O/k127_cli.log 2>&1; echo "k127 20M cli rc=$?"
mkdir -p $O/k127; cp $D/result*.json $D/stderr*.txt $O/k127/ 2>/dev/null; rm -rf $D
# pass 2: probes in flight per lane
D=/tmp/pgbig60
C="--reads 60000000 --out $D --keep-fastq --expect profiles/r03_ref_60M_K63.json"
timeout 900 python scripts/big_cli_check.py $C --tag _warm > $O/w.log 2>&1; echo "big60 warm rc=$?"
for bl in 1 4 8; do
  timeout 900 python scripts/big_cli_check.py $C --tag _p2b$bl --env SOAPDENOVO2_AMD_P2_BLOCK=$bl --rocprof "--kernel-trace --stats" > $O/p2b$bl.log 2>&1; echo "big60 p2 block $bl rc=$?"
done
timeout 900 python scripts/big_cli_check.py --reads 60000000 --out $D --keep-fastq --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16 --env SOAPDENOVO2_AMD_DEVICES=0,0,0 > $O/s3.log 2>&1; echo "big60 -a 16 three ranks rc=$?"
rm -rf $D/reads.fq
mkdir -p $O/big60; cp $D/result*.json $D/stderr*.txt $O/big60/ 2>/dev/null
for bl in 1 4 8; do f=$(find $D/prof_p2b$bl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/big60/kernel_stats_p2b$bl.csv && grep -E "p2_thread|eb_walk|eb_list|tip_walk" $f | cut -c1-200; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big60/result*.json") + glob.glob("$O/k127/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
    print("   ", [l for l in j["log"] if "[cli]" in l][-8:])
PY
