# round 3, call 29: the default bench line with the executable's runs in front of the timed pass
mkdir -p gpurun_out/r3ab
timeout 1800 python bench.py > gpurun_out/r3ab/bench_default.log 2> gpurun_out/r3ab/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
l = [x for x in open("gpurun_out/r3ab/bench_default.log") if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
print("pass", round(j["ms_per_step"], 1), "value", round(j["value"] / 1e6, 1), "M reads/s; k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
w = j["whole_command"]; print("   whole", {k: w.get(k) for k in ("reads", "wall_s", "reference_wall_s", "files_identical_to_reference", "stages_s")})
for k in ("whole_command_60M_a16", "whole_command_60M"):
    b = j.get(k) or {}
    print("   ", k, {q: b.get(q) for q in ("wall_s", "layout_s", "layout_on_device", "files_identical_to_reference", "stages_s")})
print("   cpu", j.get("cpu_baseline"))
PY
tail -3 gpurun_out/r3ab/bench_default.err
