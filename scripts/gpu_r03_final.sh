# round 3, final call: the whole GPU suite, smoke, the default bench line, kernel stats + PMC passes of the same command
mkdir -p gpurun_out/r3f
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3f/pytest.log | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3f/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3f/smoke.log
timeout 1800 python bench.py > gpurun_out/r3f/bench_default.log 2> gpurun_out/r3f/bench_default.err; echo "bench default rc=$?"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/prof_stats -- $B > $R/gpurun_out/r3f/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3f/pmc_fetch -- $B > $R/gpurun_out/r3f/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3f/pmc_write -- $B > $R/gpurun_out/r3f/pmc_write.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/r3f/pmc_fetch gpurun_out/r3f/pmc_fetch.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r3f/pmc_write gpurun_out/r3f/pmc_write.json > /dev/null 2>&1
for f in $(find gpurun_out/r3f/prof_stats -name "*kernel_stats.csv"); do cp $f gpurun_out/r3f/kernel_stats_200M.csv; head -4 $f | cut -c1-220; done
grep "^{" gpurun_out/r3f/prof_stats.log | tail -1 > gpurun_out/r3f/bench_under_stats.json
find gpurun_out/r3f -name "*.db" -delete; find gpurun_out/r3f -name "*counter_collection.csv" -delete; find gpurun_out/r3f -name "*kernel_trace.csv" -delete; find gpurun_out/r3f -name "*agent_info.csv" -delete
python - <<PY
import json
for f in ("bench_default.log", "bench_under_stats.json"):
    try:
        l = [x for x in open("gpurun_out/r3f/" + f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f, "pass", round(j["ms_per_step"], 1), "value", round(j["value"] / 1e6, 1), "M reads/s; k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
        if "whole_command" in j:
            w = j["whole_command"]; print("   whole", {k: w.get(k) for k in ("reads", "wall_s", "reference_wall_s", "files_identical_to_reference")})
            for k in ("whole_command_60M_a16", "whole_command_60M"):
                b = j.get(k) or {}
                print("   ", k, {q: b.get(q) for q in ("wall_s", "layout_s", "layout_on_device", "files_identical_to_reference", "reference_wall_s", "stages_s", "tips")})
            print("   cpu", j.get("cpu_baseline"))
            print("   hand_over", j.get("pass1_hand_over"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r3f/bench_default.err
