# round 2, second half (K2 with packed counters, global_load_lds window prefetch, fewer barriers): the default bench line,
# kernel stats + PMC at 200 M reads, SQ counters and phase timers at 20 M
mkdir -p gpurun_out/r2f
R=$GRAFT_REPO_ROOT
timeout 1500 python bench.py > gpurun_out/r2f/bench_default.log 2> gpurun_out/r2f/bench_default.err; echo "bench rc=$?"
PG_DBG=2 timeout 300 python bench.py --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "K2 phase" | tail -12 > gpurun_out/r2f/k2_phase_cycles_20M.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2f/prof_stats -- $B > $R/gpurun_out/r2f/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2f/pmc_fetch -- $B > $R/gpurun_out/r2f/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2f/pmc_write -- $B > $R/gpurun_out/r2f/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/r2f/pmc_sq -- python $R/bench.py --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2f/pmc_sq.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/r2f/pmc_fetch gpurun_out/r2f/pmc_fetch.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r2f/pmc_write gpurun_out/r2f/pmc_write.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r2f/pmc_sq gpurun_out/r2f/pmc_sq.json > gpurun_out/r2f/pmc_sq.txt 2>&1
for f in $(find gpurun_out/r2f/prof_stats -name "*kernel_stats.csv"); do cp $f gpurun_out/r2f/kernel_stats_200M.csv; head -6 $f; done
find gpurun_out/r2f -name "*.db" -delete; find gpurun_out/r2f -name "*counter_collection.csv" -delete; find gpurun_out/r2f -name "*kernel_trace.csv" -delete; find gpurun_out/r2f -name "*agent_info.csv" -delete
python - <<PY
import json
for f in ("bench_default",):
    try:
        l=[x for x in open(f"gpurun_out/r2f/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{}); w=j.get("whole_command",{})
        print(f, round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3))
        print("   hand_over", j.get("pass1_hand_over"))
        print("   whole", {k:w.get(k) for k in ("reads","wall_s","reads_per_sec","stages_s","reference_wall_s","files_identical_to_reference","distinct_kmers")})
        print("   cpu", j.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r2f/bench_default.err
cat gpurun_out/r2f/k2_phase_cycles_20M.txt
cat gpurun_out/r2f/pmc_fetch.json gpurun_out/r2f/pmc_write.json 2>/dev/null | head -40
head -30 gpurun_out/r2f/pmc_sq.txt
