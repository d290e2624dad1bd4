mkdir -p gpurun_out/r2o
timeout 1500 python scripts/big_cli_check.py --out gpurun_out/r2o/big60 --reads 60000000 --read-len 150 --genome 100000000 --err 0.001 --kmer 63 --single > gpurun_out/r2o/big60.log 2>&1; echo "big60 rc=$?"
python - <<PY
import json
try:
    j=json.load(open("gpurun_out/r2o/big60/result.json")); p=j["partitions"]; print("60M wall", p["wall_s"]); print("\n".join(p["log"][:60])); print(p.get("md5"))
except Exception as e: print("ERR", e); print(open("gpurun_out/r2o/big60.log").read()[-1500:])
PY
