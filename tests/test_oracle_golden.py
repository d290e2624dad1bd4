"""The CPU restatement (oracle/) against the golden files produced by the real reference (tests/golden)."""
import os

import pytest

from conftest import GOLDEN, case_codes, case_lens, case_tag, md5_file
from oracle_binding import run_oracle

SMALL = ["t6k_k31", "t8k_k63", "t6k_k127", "t5k_k24", "d8k_k127", "r8k_k127", "d8k_k63"]


def _cases(golden, names):
    for name in names:
        c = golden["cases"][name]
        for run in c["runs"]:
            yield name, c, run


# (g*: trimmed reads, lengths uniform in [min_len, L]; the runs of t6k_k31 / t8k_k63 / t6k_k127 include -p 16, 37, 64, 255 and -d with -a)
@pytest.mark.parametrize("name", SMALL + ["m60k_k63", "g120k_k63", "g40k_k127", "g60k_k31", "x500_k63", "x400_k127", "y300_k63"])
def test_oracle_matches_reference_digests(golden, tmp_path, name):
    c = golden["cases"][name]
    codes = case_codes(c)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        run_oracle(codes, c["K"], P, pre, D=D, a_gb=a, mer127=bool(m), lens=case_lens(c))
        want = golden["md5"][t]
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "edge"):
            assert md5_file(f"{pre}.{ext}") == want[ext], (t, ext)


def test_golden_files_match_their_digests(golden):
    import gzip, hashlib
    for f in os.listdir(GOLDEN):
        if f.endswith(".py") or f == "cases.json":
            continue
        t, ext = f.split(".", 1)
        data = gzip.open(os.path.join(GOLDEN, f), "rb").read() if ext == "edge.gz" else open(os.path.join(GOLDEN, f), "rb").read()
        assert hashlib.md5(data).hexdigest() == golden["md5"][t][ext.replace(".gz", "")], f
