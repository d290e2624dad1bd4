# final validation of the round: full GPU suite, the default bench line, bench --gpus 2 sharing one GPU over gloo (plumbing of
# the N > 1 path), 2^24 partitions on a small input
mkdir -p gpurun_out/r2v
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2v/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2v/pytest.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r2v/pytest.log | tail -4
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --reads 4000000 --genome 4000000 --comm gloo --share-gpu > gpurun_out/r2v/bench_2ranks_gloo.log 2> gpurun_out/r2v/bench_2ranks_gloo.err; echo "2rank rc=$?"
PG_LOG2_PARTS=24 timeout 600 python bench.py --reads 8000000 --genome 4000000 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/r2v/bench_p24.log 2>&1; echo "p24 rc=$?"; tail -c 300 gpurun_out/r2v/bench_p24.log
timeout 1500 python bench.py > gpurun_out/r2v/bench_default.log 2> gpurun_out/r2v/bench_default.err; echo "default rc=$?"
python - <<PY
import json
for f in ("bench_default",):
    try:
        l=[x for x in open(f"gpurun_out/r2v/{f}.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{}); w=j.get("whole_command",{})
        print(f, "value", round(j["value"]/1e6,1), "M reads/s", round(j["ms_per_step"],1), "ms k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "both", round(r.get("pass1_both_kernels_frac",0),3))
        print("   hand_over", j.get("pass1_hand_over"))
        print("   whole", {k:w.get(k) for k in ("reads","wall_s","reads_per_sec","stages_s","reference_wall_s","files_identical_to_reference","distinct_kmers")})
        print("   cpu", j.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r2v/bench_default.err
