# end of round 2: the full GPU suite, persistent workgroups per CU once more, the default bench line, the whole command at 60 M
# reads (twice: the first process on a fresh box pays the HIP start-up)
mkdir -p gpurun_out/r2z
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2z/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r2z/pytest.log | tail -2
for v in 1 3 2; do
  PG_K2_WG_PER_CU=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
j=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=j['roofline']
print('wg/cu=$v', round(j['ms_per_step'],1), 'k1', round(r['k1_scatter_ms_per_step'],1), 'k2', round(r['k2_count_ms_per_step'],1), 'frac', round(r['frac'],3))"
done
timeout 1500 python bench.py > gpurun_out/r2z/bench_default.log 2> gpurun_out/r2z/bench_default.err; echo "default rc=$?"
timeout 1500 python scripts/big_cli_check.py --out gpurun_out/r2z/big60 --reads 60000000 --read-len 150 --genome 100000000 --err 0.001 --kmer 63 --single --variant PG_GROW_VERBOSE=0 > gpurun_out/r2z/big60.json 2> gpurun_out/r2z/big60.err; echo "rc=$?"
rm -rf gpurun_out/r2z/big60/reads.fq
python - <<PY
import json
j=json.load(open("gpurun_out/r2z/big60.json"))
for k,v in j.items():
    if isinstance(v,dict) and "wall_s" in v: print(k, round(v["wall_s"],2), v.get("md5",{}).get("edge"))
for f in ("bench_default",):
    try:
        l=[x for x in open(f"gpurun_out/r2z/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{}); w=j.get("whole_command",{})
        print(f, "value", round(j["value"]/1e6,1), "M reads/s", round(j["ms_per_step"],1), "ms k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3))
        print("   hand_over", j.get("pass1_hand_over"))
        if w: print("   whole", {k:w.get(k) for k in ("reads","wall_s","reads_per_sec","stages_s","reference_wall_s","files_identical_to_reference","distinct_kmers")})
        if j.get("cpu_baseline"): print("   cpu", j.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e)
PY
grep "\[cli\]\|reader:\|Time spent\|released\|uploaded\|tip scan: 7\|edges:" gpurun_out/r2z/stderr_variant1.txt | tail -30
