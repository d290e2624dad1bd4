#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the summaries of separate rocprofv3 --pmc passes (scripts/gpu_run.sh `pmc:` steps -> pmc<i>_summary.json):

    python scripts/pmc_traffic.py --reads 200000000 --fetch F.json --write W.json [--sq S.json --sq-reads N] [--sq127 S127.json --sq127-reads N] --source "profiles/r05_final_*"

Bytes = counter x 1024 (FETCH_SIZE / WRITE_SIZE count kilobytes), per read of the measured run.  FETCH_SIZE of K2 is doubled as MI355X_MICROARCH.md
prescribes for 16-byte-a-lane streamed reads on gfx950 (its record reads); K1's loads (8 bytes a lane) and WRITE_SIZE are uncalibrated there and
stay as counted.  The file carries a hash of the kernels' sources (bench.library_source_sha): bench.py prints `traffic` only for that library."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_rows(path, name):
    j = json.load(open(path))
    rows = [v for k, v in j.items() if k.startswith("pg::" + name) or name in k]
    out = {}
    for r in rows:
        for c, x in r.items():
            out[c] = out.get(c, 0) + x
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, required=True, help="reads of the FETCH / WRITE runs x the passes each made")
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq", default="")
    ap.add_argument("--sq-reads", type=int, default=0)
    ap.add_argument("--sq127", default="")
    ap.add_argument("--sq127-reads", type=int, default=0)
    ap.add_argument("--source", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    a = ap.parse_args()
    import bench
    res = {"source": a.source, "library_sha": bench.library_source_sha(), "reads_in_measured_run": a.reads,
           "note": "bytes = counter x 1024 per read; K2's FETCH_SIZE x 2 (gfx950: 16-byte-a-lane streamed reads are tallied at half, MI355X_MICROARCH.md HBM section); "
                   "K1's fetch and both kernels' WRITE_SIZE as counted (uncalibrated access widths)"}
    for kn, fetch_factor in (("skm_scatter_seg_kernel", 1.0), ("skm_count_kernel", 2.0)):
        f, w = kernel_rows(a.fetch, kn).get("FETCH_SIZE", 0), kernel_rows(a.write, kn).get("WRITE_SIZE", 0)
        res[kn + "_fetch_bytes_per_read_raw"] = f * 1024 / a.reads
        res[kn + "_fetch_bytes_per_read"] = f * 1024 * fetch_factor / a.reads
        res[kn + "_write_bytes_per_read"] = w * 1024 / a.reads
        res[kn + "_bytes_per_read"] = (f * fetch_factor + w) * 1024 / a.reads
    for key, path, reads in (("", a.sq, a.sq_reads), ("_K127", a.sq127, a.sq127_reads)):
        if not path:
            continue
        r = kernel_rows(path, "skm_count_kernel")
        res["skm_count_kernel_valu_insts_per_read" + key] = r.get("SQ_INSTS_VALU", 0) / reads
        wc = r.get("SQ_WAVE_CYCLES", 0)
        res["skm_count_kernel_sq" + key] = {"reads": reads, "SQ_WAIT_ANY_over_WAVE_CYCLES": r.get("SQ_WAIT_ANY", 0) / wc if wc else None,
                                            "LDS_BANK_CONFLICT_over_ACTIVE_INST_LDS": r.get("SQ_LDS_BANK_CONFLICT", 0) / r["SQ_ACTIVE_INST_LDS"] if r.get("SQ_ACTIVE_INST_LDS") else None,
                                            "raw": r}
        r1 = kernel_rows(path, "skm_scatter_seg_kernel")
        if r1:
            res["skm_scatter_seg_kernel_sq" + key] = {"reads": reads, "valu_insts_per_read": r1.get("SQ_INSTS_VALU", 0) / reads,
                                                      "LDS_BANK_CONFLICT_over_ACTIVE_INST_LDS": r1.get("SQ_LDS_BANK_CONFLICT", 0) / r1["SQ_ACTIVE_INST_LDS"] if r1.get("SQ_ACTIVE_INST_LDS") else None, "raw": r1}
    res["sq_source"] = a.source
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
