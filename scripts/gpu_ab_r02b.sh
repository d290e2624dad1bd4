# K2 with the emit of a partition overlapped with the prepare of the next: parity first (under short timeouts), then A/B
# (PG_DBG=16 = no overlap) on one box
mkdir -p gpurun_out/ab
timeout 300 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "count_matches or growth or ragged or route" > gpurun_out/ab/pytest1.log 2>&1; echo "pytest1 rc=$?"; grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/ab/pytest1.log | tail -5
if grep -q "passed" gpurun_out/ab/pytest1.log && ! grep -q "failed" gpurun_out/ab/pytest1.log; then
for v in 16 0 16 0; do
  PG_DBG=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/ab/b_$v.log 2>gpurun_out/ab/b_$v.err; echo "rc=$?"
  python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/ab/b_$v.log") if x.startswith("{")][-1]; j=json.loads(l); r=j["roofline"]
    print("dbg=$v", round(j["ms_per_step"],1), "k1", round(r["k1_scatter_ms_per_step"],1), "k2", round(r["k2_count_ms_per_step"],1), "frac", round(r["frac"],3))
except Exception as e: print("ERR", e)
PY
done
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/ab/pytest2.log 2>&1; echo "pytest2 rc=$?"; grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/ab/pytest2.log | tail -5
fi
