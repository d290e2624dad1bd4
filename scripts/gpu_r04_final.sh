# round 4, final call: the 200 M-read command on its own with the start-up marks (twice: right behind the generator's 63 GB of writes, and again), the GPU suite,
# smoke, the default bench line, kernel stats of the same command
O=gpurun_out/r4g; mkdir -p $O
R=$GRAFT_REPO_ROOT
D=/tmp/pgbig200
C="--reads 200000000 --a-gb 40 --out $D --keep-fastq --expect profiles/r04_ref_200M_K63_a40.json"
timeout 1200 python scripts/big_cli_check.py $C --tag _first > $O/first.log 2>&1; echo "200M -a 40 first rc=$?"
timeout 1200 python scripts/big_cli_check.py $C --tag _second > $O/second.log 2>&1; echo "200M -a 40 second rc=$?"
rm -rf $D/reads.fq
mkdir -p $O/big200; cp $D/result*.json $D/stderr*.txt $O/big200/ 2>/dev/null; rm -rf $D
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big200/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    print("   ", [l for l in j["log"] if "[cli]" in l][:14])
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
        print("  ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "files_identical_to_reference", "skipped", "rc", "stages_s")})
except Exception as e:
    print("default bench ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/prof_stats.log 2>&1
cd $R
for f in $(find $O/prof_stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_200M.csv; head -4 $f | cut -c1-200; done
grep "^{" $O/prof_stats.log | tail -1 > $O/bench_under_stats.json
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
