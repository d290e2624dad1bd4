# round 4, the last call: parity of the counting kernels and the K = 127 / toggled commands, then the default bench line of the final library
O=gpurun_out/r4u; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or round3_switches or device_pass2 or sharded_pass1" > $O/pytest_sub.log 2>&1; echo "pytest subset rc=$?"; tail -2 $O/pytest_sub.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
try:
    l = [x for x in open("$O/bench_default.log") if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
    print("default: pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"], "valu", r.get("valu_issue_frac"))
    print("  k127", {k: j["k127"].get(k) for k in ("ms_per_pass", "k2_count_ms", "roofline_frac_k2")}, j["k127"]["conservation"]["ok"])
    for k in ("whole_command", "whole_command_60M_a16", "whole_command_60M", "whole_command_k127_20M", "whole_command_200M_a40"):
        b = j.get(k) or {}
        print("  ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "device_context_steps_s", "files_identical_to_reference", "skipped", "rc")})
except Exception as e:
    print("default bench ERR", e)
PY
