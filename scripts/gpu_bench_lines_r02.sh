mkdir -p gpurun_out/r2j
R=$GRAFT_REPO_ROOT
timeout 1200 python bench.py --kmer 127 --no-cpu-baseline --whole-reads 0 > gpurun_out/r2j/bench_k127.log 2> gpurun_out/r2j/bench_k127.err; echo "k127 rc=$?"
timeout 1200 python bench.py --kmer 31 --read-len 100 --genome 4600000 --reads 10000000 --err 0.005 --no-cpu-baseline > gpurun_out/r2j/bench_k31.log 2> gpurun_out/r2j/bench_k31.err; echo "k31 rc=$?"
timeout 1500 python bench.py > gpurun_out/r2j/bench_default.log 2> gpurun_out/r2j/bench_default.err; echo "default rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2j/prof_stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2j/prof_stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2j/prof_stats127 -- python $R/bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2j/prof_stats127.log 2>&1
cd $R
for d in prof_stats prof_stats127; do for f in $(find gpurun_out/r2j/$d -name "*kernel_stats.csv"); do cp $f gpurun_out/r2j/${d}_kernel_stats.csv; head -3 $f | cut -c1-160; done; done
find gpurun_out/r2j -name "*.db" -delete; find gpurun_out/r2j -name "*kernel_trace.csv" -delete; find gpurun_out/r2j -name "*agent_info.csv" -delete
python - <<PY
import json
for f in ("bench_k127","bench_k31","bench_default"):
    try:
        l=[x for x in open(f"gpurun_out/r2j/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{}); w=j.get("whole_command",{})
        print(f, "value", round(j["value"]/1e6,1), "M reads/s", round(j["ms_per_step"],1), "ms k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3), "distinct", j["config"]["distinct_kmers"], "kernel", r.get("kernel"))
        print("   hand_over", j.get("pass1_hand_over"))
        if w: print("   whole", {k:w.get(k) for k in ("reads","wall_s","reads_per_sec","stages_s","reference_wall_s","files_identical_to_reference","distinct_kmers")})
    except Exception as e: print(f, "ERR", e)
PY
tail -2 gpurun_out/r2j/bench_k127.err gpurun_out/r2j/bench_k31.err gpurun_out/r2j/bench_default.err
