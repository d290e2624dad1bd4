mkdir -p gpurun_out/r2x
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "cli or pass2 or linked" > gpurun_out/r2x/pytest_cli.log 2>&1; echo "pytest rc=$?"; grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r2x/pytest_cli.log | tail -2
timeout 1500 python scripts/big_cli_check.py --out gpurun_out/r2x/big60 --reads 60000000 --read-len 150 --genome 100000000 --err 0.001 --kmer 63 --single --variant PG_GROW_VERBOSE=0 > gpurun_out/r2x/big60.json 2> gpurun_out/r2x/big60.err; echo "rc=$?"
rm -rf gpurun_out/r2x/big60/reads.fq
python - <<PY
import json
j=json.load(open("gpurun_out/r2x/big60.json"))
for k,v in j.items():
    if isinstance(v,dict) and "wall_s" in v: print(k, round(v["wall_s"],2), v.get("md5",{}))
PY
grep "\[cli\]\|reader:\|Time spent\|uploaded\|tip scan: 7\|edges:" gpurun_out/r2x/stderr_variant1.txt | tail -24
