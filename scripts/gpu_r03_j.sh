# round 3, call 10: evidence for the round's kernels -- kernel stats + PMC (separate FETCH / WRITE passes) of the default bench at 200 M reads,
# SQ counters and K2 phase timers at 20 M (K = 63 and K = 127), kernel trace of the 60 M-read command with growable sets
mkdir -p gpurun_out/r3j
R=$GRAFT_REPO_ROOT
for k in 63 127; do
PG_DBG=2 timeout 300 python bench.py --kmer $k --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "K2 phase" | tail -12 > gpurun_out/r3j/k2_phase_cycles_20M_k$k.txt
done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3j/prof_stats -- $B > $R/gpurun_out/r3j/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3j/pmc_fetch -- $B > $R/gpurun_out/r3j/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3j/pmc_write -- $B > $R/gpurun_out/r3j/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/r3j/pmc_sq -- python $R/bench.py --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3j/pmc_sq.log 2>&1
cd $R
python scripts/pmc_summary.py gpurun_out/r3j/pmc_fetch gpurun_out/r3j/pmc_fetch.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r3j/pmc_write gpurun_out/r3j/pmc_write.json > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r3j/pmc_sq gpurun_out/r3j/pmc_sq.json > gpurun_out/r3j/pmc_sq.txt 2>&1
for f in $(find gpurun_out/r3j/prof_stats -name "*kernel_stats.csv"); do cp $f gpurun_out/r3j/kernel_stats_200M.csv; head -6 $f; done
grep "^{" gpurun_out/r3j/prof_stats.log | tail -1 > gpurun_out/r3j/bench_under_stats.json
D=/tmp/pgbig60
timeout 900 python scripts/big_cli_check.py --reads 60000000 --out $D --expect profiles/r03_ref_60M_K63.json --tag _prof --rocprof "--kernel-trace --stats" > gpurun_out/r3j/big60_prof.log 2>&1; echo "big60 under rocprof rc=$?"
for f in $(find $D -name "*kernel_stats.csv"); do cp $f gpurun_out/r3j/kernel_stats_60M_growable.csv; head -25 $f; done
cp $D/result_prof.json gpurun_out/r3j/ 2>/dev/null
find gpurun_out/r3j -name "*.db" -delete; find gpurun_out/r3j -name "*counter_collection.csv" -delete; find gpurun_out/r3j -name "*kernel_trace.csv" -delete; find gpurun_out/r3j -name "*agent_info.csv" -delete
cat gpurun_out/r3j/k2_phase_cycles_20M_k63.txt gpurun_out/r3j/k2_phase_cycles_20M_k127.txt
cat gpurun_out/r3j/pmc_fetch.json gpurun_out/r3j/pmc_write.json 2>/dev/null | head -60
head -30 gpurun_out/r3j/pmc_sq.txt
python - <<PY
import json
j=json.loads(open("gpurun_out/r3j/bench_under_stats.json").read()); r=j["roofline"]
print("under stats: pass", round(j["ms_per_step"],1), "k1", round(r["k1_scatter_ms_per_step"],1), "k2", round(r["k2_count_ms_per_step"],1), "frac", round(r["frac"],3), "both", round(r["pass1_both_kernels_frac"],3))
PY
