# end of round 2: CLI tests, the whole command at 60 M reads (twice: the first process on a fresh box pays the HIP start-up),
# the bench lines (default with the reference beside it, K = 127, K = 31 x 100 bp, whole command at 10 M reads)
mkdir -p gpurun_out/r2z
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "cli or linked or twice" > gpurun_out/r2z/pytest_cli.log 2>&1; echo "pytest rc=$?"; grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r2z/pytest_cli.log | tail -2
timeout 1500 python scripts/big_cli_check.py --out gpurun_out/r2z/big60 --reads 60000000 --read-len 150 --genome 100000000 --err 0.001 --kmer 63 --single --variant PG_GROW_VERBOSE=0 > gpurun_out/r2z/big60.json 2> gpurun_out/r2z/big60.err; echo "rc=$?"
rm -rf gpurun_out/r2z/big60/reads.fq
timeout 1500 python bench.py > gpurun_out/r2z/bench_default.log 2> gpurun_out/r2z/bench_default.err; echo "default rc=$?"
timeout 1200 python bench.py --kmer 127 --no-cpu-baseline --whole-reads 0 > gpurun_out/r2z/bench_k127.log 2> gpurun_out/r2z/bench_k127.err; echo "k127 rc=$?"
timeout 1200 python bench.py --kmer 31 --read-len 100 --genome 4600000 --reads 10000000 --err 0.005 --no-cpu-baseline > gpurun_out/r2z/bench_k31.log 2> gpurun_out/r2z/bench_k31.err; echo "k31 rc=$?"
python - <<PY
import json
j=json.load(open("gpurun_out/r2z/big60.json"))
for k,v in j.items():
    if isinstance(v,dict) and "wall_s" in v: print(k, round(v["wall_s"],2), v.get("md5",{}).get("edge"))
for f in ("bench_default","bench_k127","bench_k31"):
    try:
        l=[x for x in open(f"gpurun_out/r2z/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{}); w=j.get("whole_command",{})
        print(f, "value", round(j["value"]/1e6,1), "M reads/s", round(j["ms_per_step"],1), "ms k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3))
        print("   hand_over", j.get("pass1_hand_over"))
        if w: print("   whole", {k:w.get(k) for k in ("reads","wall_s","reads_per_sec","stages_s","reference_wall_s","files_identical_to_reference","distinct_kmers")})
        if j.get("cpu_baseline"): print("   cpu", j.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e)
PY
grep "\[cli\]\|reader:\|Time spent\|released\|uploaded\|tip scan: 7\|edges:" gpurun_out/r2z/stderr_variant1.txt | tail -30
