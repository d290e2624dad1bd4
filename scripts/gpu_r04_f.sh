# round 4, call 6: the growable layout with one read-back a round / blind rounds for the small sizes (60 M-read command, A/B); presplit thresholds; the GPU suite;
# the default bench line and the kernel stats of the same command (the round's final numbers)
O=gpurun_out/r4f; mkdir -p $O
R=$GRAFT_REPO_ROOT
D=/tmp/pgbig60
C="--reads 60000000 --out $D --keep-fastq --expect profiles/r03_ref_60M_K63.json"
timeout 900 python scripts/big_cli_check.py $C --tag _warm > $O/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $C --tag _blind > $O/b.log 2>&1; echo "big60 growable (blind rounds) rc=$?"
timeout 900 python scripts/big_cli_check.py $C --tag _noblind --env PG_RH_BLIND_MAX=0 > $O/nb.log 2>&1; echo "big60 growable (every round read back) rc=$?"
timeout 900 python scripts/big_cli_check.py $C --tag _blind2 > $O/b2.log 2>&1; echo "big60 growable (blind rounds) again rc=$?"
timeout 900 python scripts/big_cli_check.py $C --tag _lanes4 --env SOAPDENOVO2_AMD_LAYOUT_LANES=4 > $O/l4.log 2>&1; echo "big60 growable, four sets side by side rc=$?"
timeout 900 python scripts/big_cli_check.py $C --tag _sh2 --env SOAPDENOVO2_AMD_DEVICES=0,0 > $O/s2.log 2>&1; echo "big60 two ranks rc=$?"
rm -rf $D/reads.fq
mkdir -p $O/big60; cp $D/result*.json $D/stderr*.txt $O/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big60/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"), [l for l in j["log"] if "rebuilding the k-mer set layout" in l or "growable sets on device" in l][:3])
PY
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
run k127_base PG_NOP=1
run k127_pre75 PG_K2_PRESPLIT_PCT=75
run k127_pre90 PG_K2_PRESPLIT_PCT=90
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run base PG_NOP=1
run pre90 PG_K2_OPT=5 PG_K2_PRESPLIT_PCT=90
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 1700 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
try:
    l = [x for x in open("$O/bench_default.log") if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
    print("default: pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"], "valu", r.get("valu_issue_frac"))
    print("  k127", {k: j["k127"].get(k) for k in ("ms_per_pass", "k2_count_ms", "roofline_frac_k2")}, j["k127"]["conservation"]["ok"])
    for k in ("whole_command", "whole_command_60M_a16", "whole_command_60M", "whole_command_k127_20M", "whole_command_200M_a40"):
        b = j.get(k) or {}
        print("  ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "files_identical_to_reference", "reference_wall_s", "skipped", "rc")})
except Exception as e:
    print("default bench ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/prof_stats.log 2>&1
cd $R
for f in $(find $O/prof_stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_200M.csv; head -4 $f | cut -c1-200; done
grep "^{" $O/prof_stats.log | tail -1 > $O/bench_under_stats.json
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
