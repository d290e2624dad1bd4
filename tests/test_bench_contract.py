"""bench.py's host-side helpers, without a GPU: the workload label follows the arguments, and the HBM counters bench.py prints as
`roofline.traffic` (profiles/pmc_traffic.json) belong to the pass-1 kernels that are in the tree -- a change to those sources without a new
PMC pass would silently turn `traffic` into null in the driver's run."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(mod)          # (bench.py runs nothing at import: its work is under main())
    finally:
        sys.argv = argv
    return mod


def test_workload_label_names_a_baseline_config_only_when_the_arguments_are_it():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "configs" in base and len(base["configs"]) >= 3
    assert "configs[2]" in b.workload_label(200_000_000, 150, 100_000_000, 0.001, 63, 8, 1)
    assert "configs[1]" in b.workload_label(10_000_000, 100, 4_600_000, 0.005, 31, 8, 1)
    for other in ((200_000_000, 150, 100_000_000, 0.001, 31, 8, 1), (20_000_000, 150, 100_000_000, 0.001, 63, 8, 1), (200_000_000, 150, 100_000_000, 0.001, 63, 8, 2)):
        label = b.workload_label(*other)
        assert "not one of BASELINE.json's configs" in label and f"K={other[4]}" in label and str(other[0]) in label


def test_committed_hbm_counters_belong_to_the_committed_pass1_kernels():
    b = _bench()
    tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert tj["library_sha"] == b.library_source_sha(), (
        "partition_kernels.hip or one of its headers changed after profiles/pmc_traffic.json was taken: re-run the pmc:FETCH_SIZE / pmc:WRITE_SIZE steps of "
        "scripts/gpu_run.sh and scripts/pmc_traffic.py, or bench.py prints roofline.traffic = null")


def test_every_whole_command_leg_has_its_expectation_in_profiles():
    """bench.py's big legs compare the executable's files with md5s of the REFERENCE's own runs, committed under profiles/; a leg whose file is missing is skipped
    without a word in the line.  Every file bench.py names must be there, carry the five md5s and the reference's wall time, and say what it was run on."""
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    names = sorted(set(re.findall(r'\("whole_command_[A-Za-z0-9_]+", "(r0\d_ref_[A-Za-z0-9_]+\.json)"\)', src)))
    assert len(names) >= 7, names
    for f in names:
        j = json.load(open(os.path.join(ROOT, "profiles", f)))
        assert set(j["md5"]) >= {"kmerFreq", "preGraphBasic", "vertex", "preArc", "edge"}, f
        assert j.get("reference_wall_s", 0) > 100 and {"reads", "read_len", "genome", "err", "seed", "kmer", "sets", "a_gb"} <= set(j["workload"]), f
