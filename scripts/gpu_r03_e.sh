# round 3, call 5: K1 with the chunk hand-out split over 1024 counters; record stride / chunk size / direct chunks on top; all GPU tests
mkdir -p gpurun_out/r3e
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3e/$tag.log 2> gpurun_out/r3e/$tag.err; echo "$tag rc=$?"; }
run base PG_NONE=1
run stride8 PG_REC_STRIDE=8
run rpc64 PG_RPC=64
run rpc32 PG_RPC=32
run rpc32_stride8 PG_RPC=32 PG_REC_STRIDE=8
run direct4 PG_DIRECT_CHUNKS=4
run direct4_stride8 PG_DIRECT_CHUNKS=4 PG_REC_STRIDE=8
run direct16_rpc32 PG_RPC=32 PG_REC_STRIDE=8 PG_DIRECT_CHUNKS=16
run r16 PG_K1_R=16
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3e/pytest.log | tail -2
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3e/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "ok", j["conservation"]["ok"], "distinct", j["config"]["distinct_kmers"])
    except Exception as e:
        pass
PY
