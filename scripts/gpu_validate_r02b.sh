# second-half validation: the full GPU suite, then the whole command at 60 M reads twice (the first process on a fresh box
# pays 2-4 s of HIP start-up that the second does not)
mkdir -p gpurun_out/r2w
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2w/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r2w/pytest.log | tail -4
timeout 1500 python scripts/big_cli_check.py --out gpurun_out/r2w/big60 --reads 60000000 --read-len 150 --genome 100000000 --err 0.001 --kmer 63 --single --variant PG_GROW_VERBOSE=0 > gpurun_out/r2w/big60.json 2> gpurun_out/r2w/big60.err; echo "rc=$?"
rm -rf gpurun_out/r2w/big60/reads.fq
python - <<PY
import json
j=json.load(open("gpurun_out/r2w/big60.json"))
for k,v in j.items():
    if isinstance(v,dict) and "wall_s" in v: print(k, round(v["wall_s"],2), v.get("md5",{}).get("edge"))
PY
grep "\[cli\]\|memory at\|reader:\|Time spent\|vertex writer\|uploaded\|tip scan\|edges:" gpurun_out/r2w/stderr_variant1.txt | tail -40
