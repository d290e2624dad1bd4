# round 4, last call: the 60 M-read command under kernel stats (pass 2's probes as global loads), the GPU suite, smoke, the default bench line, kernel stats of the same
# command, the SQ counters of K2 at the benchmarked coverage (K = 63 and K = 127)
O=gpurun_out/r4p; mkdir -p $O
R=$GRAFT_REPO_ROOT
D=/tmp/pgbig60
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/scripts/big_cli_check.py --reads 60000000 --a-gb 16 --out $D --expect $R/profiles/r03_ref_60M_K63_a16.json --rocprof "--kernel-trace --stats --output-format csv" > $R/$O/big60_stats.log 2>&1; echo "big60 under stats rc=$?"
cd $R
for f in $(find $D -name "*kernel_stats.csv" 2>/dev/null); do cp $f $O/graph_kernels_60M.csv; done
cp $D/result*.json $O/ 2>/dev/null; rm -rf $D
grep -E "p2_thread|eb_walk|eb_list|be_append|K6" $O/graph_kernels_60M.csv | cut -c1-160 | head -8
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
        print("  ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "device_context_steps_s", "files_identical_to_reference", "skipped", "rc", "stages_s")})
except Exception as e:
    print("default bench ERR", e)
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/prof_stats.log 2>&1
for k in 63 127; do
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/pmc_sq_k$k -- python $R/bench.py --kmer $k --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-k127 > $R/$O/pmc_sq_k$k.log 2>&1
done
cd $R
for f in $(find $O/prof_stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_200M.csv; head -4 $f | cut -c1-200; done
grep "^{" $O/prof_stats.log | tail -1 > $O/bench_under_stats.json
for k in 63 127; do python scripts/pmc_summary.py $O/pmc_sq_k$k $O/pmc_sq_k$k.json > $O/pmc_sq_k$k.txt 2>&1; done
python - <<PY
import json
for k in (63, 127):
    try:
        j = json.load(open("$O/pmc_sq_k%d.json" % k))
        for name, v in j.items():
            if "skm_count" in name or "skm_scatter_seg" in name: print(k, name[:60], {a: round(b) for a, b in v.items()})
    except Exception as e: print("pmc", k, "ERR", e)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
