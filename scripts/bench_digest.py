#!/usr/bin/env python3
"""The numbers of one bench.py line that a GPU call's tail should show: python scripts/bench_digest.py <file with the JSON line>"""
import json, sys


def main():
    try:
        line = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
        j = json.loads(line)
    except Exception as e:  # noqa: BLE001
        print("   no bench line:", e)
        return
    r = j.get("roofline", {})
    print("   pass", round(j["ms_per_step"], 2), "ms; K1", r.get("k1_scatter_ms_per_step"), "K2", r.get("k2_count_ms_per_step"), "frac", round(r.get("frac", 0), 4),
          "traffic", r.get("traffic"), "ok", (j.get("conservation") or {}).get("ok"), "workload:", (j.get("config") or {}).get("workload"))
    for k, b in j.items():
        if isinstance(b, dict) and k.startswith("k") and k[1:].isdigit():
            print("  ", k, {q: b.get(q) for q in ("ms_per_pass", "k1_scatter_ms", "k2_count_ms", "roofline_frac_k2")}, (b.get("conservation") or {}).get("ok"))
        if isinstance(b, dict) and k.startswith("whole_command"):
            print("  ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "files_identical_to_reference", "skipped", "rc")}, b.get("stages_s"), b.get("arena"))
    if "cpu_baseline" in j:
        print("   cpu_baseline", j["cpu_baseline"])


if __name__ == "__main__":
    main()
