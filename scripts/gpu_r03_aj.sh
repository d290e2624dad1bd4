# round 3, call 39: the vertex list sized for one slot in 32 (count and come again if there are more): golden cases, the 200 M-read command
mkdir -p gpurun_out/r3aj
timeout 600 python -m pytest tests -m gpu -x -q -k "cli_matches_reference_files or device_pass2_matches or sharded_matches" > gpurun_out/r3aj/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3aj/pytest.log | tail -3
D2=/tmp/pgbig200
timeout 600 python scripts/big_cli_check.py --reads 200000000 --out $D2 --expect profiles/r03_hostreplay_200M_K63.json --tag _a0 > gpurun_out/r3aj/a200.log 2>&1; echo "big200 rc=$?"
grep -h "vertex list\|Time spent on constructing edges\|cli\] layout" $D2/stderr_a0.txt
python - <<PY
import json
j = json.load(open("/tmp/pgbig200/result_a0.json")); print("wall", j["wall_s"], "identical to the earlier run", j.get("identical_to_earlier_run"))
PY
cp $D2/result_a0.json gpurun_out/r3aj/
