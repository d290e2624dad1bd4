# whole command at 60 M reads x 150 bp (genome 100 Mb, K = 63) + the CLI tests that cover the reader
mkdir -p gpurun_out/r2w
timeout 1500 python scripts/big_cli_check.py --out gpurun_out/r2w/big60 --reads 60000000 --read-len 150 --genome 100000000 --err 0.001 --kmer 63 --single --variant PG_GROW_VERBOSE=0 > gpurun_out/r2w/big60.json 2> gpurun_out/r2w/big60.err; echo "rc=$?"
python - <<PY
import json
j=json.load(open("gpurun_out/r2w/big60.json"))
for k,v in j.items():
    if isinstance(v,dict) and "wall_s" in v:
        print(k, round(v["wall_s"],2), v.get("md5",{}).get("edge"))
        for l in v["log"]:
            if l.startswith("grow ") or l.startswith("replay set"): continue
            print("   ", l)
PY
