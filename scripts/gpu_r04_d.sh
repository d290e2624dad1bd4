# round 4, call 4: the GPU suite; partitions between two powers of two (PG_PARTS_EFF_PCT); K2 at half the coverage; the default bench line with the new start-up
# (k-mer estimate by the read geometry, export array allocated beside pass 1) and the 200 M-read / K = 127 command legs; kernel stats and FETCH / WRITE counters of the
# new kernels at the benchmarked size
O=gpurun_out/r4d; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest.log | tail -8
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
run base PG_NOP=1
run eff90 PG_PARTS_EFF_PCT=90
run eff80 PG_PARTS_EFF_PCT=80
run eff70 PG_PARTS_EFF_PCT=70
run eff60 PG_PARTS_EFF_PCT=60
run cfg3 PG_K2CFG=3
run opt5 PG_K2_OPT=5
run opt5_eff80 PG_K2_OPT=5 PG_PARTS_EFF_PCT=80
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_eff85 PG_PARTS_EFF_PCT=85
run k127_eff70 PG_PARTS_EFF_PCT=70
run k127_opt5 PG_K2_OPT=5
run k127_opt5_eff85 PG_K2_OPT=5 PG_PARTS_EFF_PCT=85
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --genome 200000000 --err 0.0005"
run cov150 PG_NOP=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"], "distinct", j["config"]["distinct_kmers"], "recs/read", round(r["records_per_read"], 2))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 1700 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
try:
    l = [x for x in open("$O/bench_default.log") if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
    print("default: pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"], "valu", r.get("valu_issue_frac"))
    print("  k127", {k: j["k127"].get(k) for k in ("ms_per_pass", "k2_count_ms", "roofline_frac_k2")}, j["k127"]["conservation"]["ok"])
    for k in ("whole_command", "whole_command_60M_a16", "whole_command_60M", "whole_command_k127_20M", "whole_command_200M_a40"):
        b = j.get(k) or {}
        print("  ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "files_identical_to_reference", "reference_wall_s", "skipped", "rc", "stages_s")})
    print("  cpu", j.get("cpu_baseline"))
    print("  hand_over", j.get("pass1_hand_over"))
except Exception as e:
    print("default bench ERR", e)
PY
tail -3 $O/bench_default.err
if [ -f profiles/r04_ref_60M_K127.json ]; then
  D=/tmp/pgbig127
  timeout 900 python scripts/big_cli_check.py --reads 60000000 --kmer 127 --out $D --expect profiles/r04_ref_60M_K127.json > $O/k127_60M_cli.log 2>&1; echo "k127 60M cli rc=$?"
  mkdir -p $O/k127; cp $D/result*.json $D/stderr*.txt $O/k127/ 2>/dev/null; rm -rf $D
  python -c "
import json; j = json.load(open('$O/k127/result.json')); print('k127 60M: rc', j['rc'], 'wall', j['wall_s'], 'identical', j.get('identical_to_reference')); print([l for l in j['log'] if '[cli]' in l][-8:])"
fi
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_stats -- $B > $R/$O/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- $B > $R/$O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- $B > $R/$O/pmc_write.log 2>&1
cd $R
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_fetch.json > /dev/null 2>&1
python scripts/pmc_summary.py $O/pmc_write $O/pmc_write.json > /dev/null 2>&1
for f in $(find $O/prof_stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_200M.csv; head -4 $f | cut -c1-200; done
grep "^{" $O/prof_stats.log | tail -1 > $O/bench_under_stats.json
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<PY
import json
for n in ("fetch", "write"):
    try:
        j = json.load(open("$O/pmc_%s.json" % n))
        for name, v in j.items():
            if "skm_" in name: print(n, name[:50], v)
    except Exception as e: print(n, "ERR", e)
PY
