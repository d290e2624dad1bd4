#!/usr/bin/env python3
"""What the compiler made of a kernel, without a GPU: hipcc -S of one .hip file for gfx950, then per kernel the number of instructions by loop depth,
the scalar registers spilled to vector lanes (and the v_readlane / v_writelane moves that go with them, by loop depth), scratch and flat memory
operations, and the s_waitcnt vmcnt waits -- the things that cost K2 between 2 and 5 % each in round 4 (DESIGN.md 3.2 "one box", "four more").

    python scripts/isa_stats.py soapdenovo2_amd/csrc/partition_kernels.hip --match skm_count_kernel -D PG_MEASURE=1
    python scripts/isa_stats.py soapdenovo2_amd/csrc/graph_kernels.hip --match p2_thread --waits

--match keeps the kernels whose (mangled) name contains the text; --waits lists every vmcnt wait, scratch and flat operation with the instruction
behind it (a wait in front of an LDS write behind a global_load_lds, a reload from scratch at the top of a loop, a probe that became a flat load).
"""
import argparse, os, re, subprocess, sys, tempfile


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--match", default="")
    ap.add_argument("-D", action="append", default=[], help="NAME=VALUE for the compiler (repeatable)")
    ap.add_argument("--waits", action="store_true")
    ap.add_argument("--keep", default="", help="write the assembly here")
    a = ap.parse_args()
    out = a.keep or os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out, a.source] + ["-D" + d for d in a.D]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-2000:])
    t = open(out).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)", t):
        blk = t[m.start():m.start() + 1200]
        g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
        lds = re.findall(r"\.group_segment_fixed_size: (\d+)", t[max(0, m.start() - 4000):m.start()])
        meta[m.group(1)] = {"sgpr": g("sgpr_count"), "sgpr_spill": g("sgpr_spill_count"), "vgpr": g("vgpr_count"), "vgpr_spill": g("vgpr_spill_count"), "lds": lds[-1] if lds else "?"}
    for m in re.finditer(r"^(_Z\S+):\s*; @", t, re.M):
        name = m.group(1)
        if a.match not in name or name not in meta:
            continue
        body = t[m.start():t.find(".Lfunc_end", m.start())].split("\n")
        depth, instr, lane, notes = 0, {}, {}, []
        for n, l in enumerate(body):
            d = re.search(r"Loop Header: Depth=(\d+)", l) or re.search(r"in Loop: Header=\S+ Depth=(\d+)", l)
            if d:
                depth = int(d.group(1))
            if not l.startswith("\t") or l.strip().startswith((".", ";")):
                continue
            instr[depth] = instr.get(depth, 0) + 1
            if "v_readlane" in l or "v_writelane" in l:
                lane[depth] = lane.get(depth, 0) + 1
            if a.waits and ("vmcnt" in l or "scratch_" in l or "flat_" in l):
                nxt = next((x.strip() for x in body[n + 1:n + 4] if x.startswith("\t") and not x.strip().startswith((".", ";"))), "")
                notes.append(f"      depth {depth}: {l.strip()[:70]:70s} | {nxt[:60]}")
        k = meta[name]
        print(f"{name[:110]}\n   instructions {sum(instr.values())} by loop depth {dict(sorted(instr.items()))}\n   VGPRs {k['vgpr']} (spilled {k['vgpr_spill']}), SGPRs {k['sgpr']} (spilled {k['sgpr_spill']}), "
              f"lane moves {sum(lane.values())} by loop depth {dict(sorted(lane.items()))}, LDS {k['lds']} B\n   scratch ops {sum('scratch_' in l for l in body)}, flat ops "
              f"{sum(('flat_load' in l or 'flat_store' in l or 'flat_atomic' in l) for l in body)}, vmcnt waits {sum('vmcnt' in l for l in body)}")
        print("\n".join(notes))


if __name__ == "__main__":
    main()
