import gzip
import hashlib
import json
import os
import subprocess
import sys

# (numpy asks for transparent huge pages for large arrays; on the build container's VM a first touch of 2 GB takes 8 s in 4 KB pages and 80 s in
#  2 MB ones -- the -a pools of the host twins are gigabytes of zeroes)
os.environ.setdefault("NUMPY_MADVISE_HUGEPAGE", "0")
if not os.path.exists("/dev/kfd"):                          # (no GPU = the build container: the host twins' k-mer sets in 4 KB pages too, csrc/host_graph.cpp: HugeArray)
    os.environ.setdefault("SOAPDENOVO2_AMD_THP", "0")
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _ensure_built():
    from soapdenovo2_amd import api
    if not os.path.exists(api.LIB_PATH) or not os.path.exists(api.binary(False)):
        api.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-f", "oracle/Makefile", "oracle/liboracle.so"], cwd=ROOT)


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


@pytest.fixture(scope="session")
def golden():
    return json.load(open(os.path.join(GOLDEN, "cases.json")))


def case_tag(name, run):
    P, D, a, m = run
    return f"{name}_p{P}_d{D}_a{a}_{'127' if m else '63'}"


def host_runs(case):
    """The runs of a golden case the CPU twins of the graph stages go through: -p up to 8.  The runs with -p 16 / 37 / 64 / 255 (round 6) are held against the
    reference by the oracle (tests/test_oracle_golden.py) and by the device path and the executable (-m gpu); through the host twins they only repeat the same
    code with more, emptier sets -- and took the CPU suite from 20 to 32 minutes (a -a pool of 37 sets is 15 GB of host memory)."""
    return [r for r in case["runs"] if r[0] <= 8]


def md5_file(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def md5_gz_text(path):
    return hashlib.md5(gzip.open(path, "rb").read()).hexdigest()


def case_codes(case):
    from soapdenovo2_amd import synth
    return synth.reads_codes_model(case.get("model", "uniform"), case["G"], case["N"], case["L"], case["err"], case["seed"], case["K"])


def case_config(case, outdir, name):
    """FASTQ + config of a golden case, as tests/golden/make_golden.py wrote them for the reference."""
    from soapdenovo2_amd import synth
    return synth.make_case(outdir, name, case["G"], case["N"], case["L"], case["err"], case["seed"], model=case.get("model", "uniform"), K=case["K"],
                           min_len=case.get("min_len", 0))


def case_lens(case):
    """Read lengths of a golden case: None when every read has case["L"] bases, else the trimmed lengths (synth.ragged_lens)."""
    from soapdenovo2_amd import synth
    return synth.ragged_lens(case["N"], case["min_len"], case["L"], case["seed"]) if case.get("min_len") else None


def oracle_records(codes, K, P, D=0, mer127=False, a_gb=0, prefix=None):
    """Pass 1 through the oracle, returned in the product's record format (key words, cnt, set<<56|ord)."""
    from oracle_binding import Oracle
    o = Oracle(K, P=P, D=D, a_gb=a_gb, mer127=mer127, max_read_len=codes.shape[1])
    o.add_reads(codes)
    o.finish_count(prefix if prefix else os.devnull[:-4] + "null")
    nd = o.nodes()
    nw = o.NW
    rec = np.zeros((len(nd["A"]), nw + 2), dtype=np.uint64)
    rec[:, :nw] = nd["keys"]
    rec[:, nw] = nd["A"].astype(np.uint64) | (nd["B"].astype(np.uint64) << np.uint64(32))
    rec[:, nw + 1] = (nd["set"].astype(np.uint64) << np.uint64(56)) | nd["ord"]
    last = np.array(o.set_last_put(), dtype=np.uint64)
    K_eff = o.K
    o.close()
    return rec, last, K_eff
