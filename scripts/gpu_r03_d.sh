# round 3, call 4: K1 variants (64-byte record slots, direct-addressed first chunks, tile size) on the default bench workload
mkdir -p gpurun_out/r3d
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3d/$tag.log 2> gpurun_out/r3d/$tag.err; echo "$tag rc=$?"; }
run base PG_NONE=1
run stride8 PG_REC_STRIDE=8
run rpc32 PG_RPC=32
run rpc32_stride8 PG_RPC=32 PG_REC_STRIDE=8
run direct16 PG_RPC=32 PG_REC_STRIDE=8 PG_DIRECT_CHUNKS=16
run direct16_s6 PG_RPC=32 PG_DIRECT_CHUNKS=16
run direct24 PG_RPC=32 PG_REC_STRIDE=8 PG_DIRECT_CHUNKS=24
run r16 PG_K1_R=16
run r24 PG_K1_R=24
run r16_direct16 PG_K1_R=16 PG_RPC=32 PG_REC_STRIDE=8 PG_DIRECT_CHUNKS=16
# parity with the toggles on (small partitions: two direct chunks of 32 records)
PG_RPC=32 PG_REC_STRIDE=8 PG_DIRECT_CHUNKS=2 timeout 900 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or growth_and_batch or ragged or cli_matches_reference_files or sharded_pass1 or last_put" > gpurun_out/r3d/pytest_toggles.log 2>&1; echo "pytest toggles rc=$?"; grep -E "passed|failed" gpurun_out/r3d/pytest_toggles.log | tail -2
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3d/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "ok", j["conservation"]["ok"], "distinct", j["config"]["distinct_kmers"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r3d/*.err | tail -30
