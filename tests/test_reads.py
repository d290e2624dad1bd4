"""Ingestion stage of the product (config parser + chunked reader + record parsers, pg_host_read_all) on CPU:
the reads it delivers, pushed through the oracle, must reproduce what the real reference made of the same files
(golden md5 + the reference's own counters), including its corner cases."""
import pytest

from conftest import md5_file
from oracle_binding import run_oracle
from soapdenovo2_amd import api, synth


@pytest.mark.parametrize("name", synth.QUIRK_CASES)
def test_reader_corner_cases(golden, tmp_path, name):
    cfg = synth.make_quirk_case(str(tmp_path), name)
    K, P = 31, 3
    codes, lens, n_records, mrl = api.host_read_all(cfg, K)
    q = golden["quirks"][name]
    assert n_records == q["reads_processed"]                           # "N read(s) processed" of the reference
    assert int((lens - K + 1).sum()) == q["kmers"]                     # "kmer(s) in reads"
    pre = str(tmp_path / "o")
    run_oracle(codes, K, P, pre, lens=lens, max_read_len=mrl)
    want = golden["md5"][name]
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "edge"):
        assert md5_file(f"{pre}.{ext}") == want[ext], (name, ext)


def test_reader_plain_fastq_and_fasta(tmp_path):
    codes = synth.reads_codes(5000, 3000, 75, 0.01, 5)
    for fmt in ("fastq", "fasta"):
        cfg = synth.make_case(str(tmp_path), "c_" + fmt, 5000, 3000, 75, 0.01, 5, fmt=fmt)
        got, lens, n, mrl = api.host_read_all(cfg, 25)
        assert n == 3000 and mrl == 75 and (lens == 75).all()
        assert (got == codes).all()


def test_config_errors_exit_like_the_reference(tmp_path):
    import subprocess, sys
    bad = tmp_path / "bad.cfg"
    bad.write_text("max_rd_len=100\navg_ins=200\nq=/nonexistent\n")
    code = ("import sys; sys.path.insert(0, %r); from soapdenovo2_amd import api; api.host_read_all(%r, 31)" % (str(api.ROOT), str(bad)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "no [LIB] in file" in r.stderr
    missing = tmp_path / "m.cfg"
    missing.write_text("max_rd_len=100\n[LIB]\navg_ins=200\nq=/nonexistent.fq\n")
    code = ("import sys; sys.path.insert(0, %r); from soapdenovo2_amd import api; api.host_read_all(%r, 31)" % (str(api.ROOT), str(missing)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "Cannot open /nonexistent.fq" in r.stderr


@pytest.mark.parametrize("reader", ["map", "copy"])
@pytest.mark.parametrize("threads,window", [(3, 2), (5, 1), (2, 64)])
def test_parallel_reader_delivers_the_same_reads(tmp_path, threads, window, reader):
    """Single files are parsed by all host threads (windows of 32 KiB chunks, cuts and buffers in parallel): same reads,
    same order, same record count as the sequential chunk emulation -- on the corner cases (N x 32768 file, ragged reads,
    truncation, two libs, lower case / N / '.') and on plain FASTQ / FASTA, with tiny windows to cross window borders.
    Both for the mapped file (parsed where the page cache holds it) and for windows copied into buffers (what a pipe gets)."""
    import os
    import numpy as np
    cfgs = [(synth.make_quirk_case(str(tmp_path), name), 31) for name in synth.QUIRK_CASES]
    for fmt in ("fastq", "fasta"):
        cfgs.append((synth.make_case(str(tmp_path), "p_" + fmt, 30000, 4000, 90, 0.01, 9, fmt=fmt), 25))
    knobs = {"SOAPDENOVO2_AMD_PARSE_PARALLEL_MIN": "0", "SOAPDENOVO2_AMD_PARSE_THREADS": str(threads),
             "SOAPDENOVO2_AMD_PARSE_WINDOW": str(window), "SOAPDENOVO2_AMD_READER": reader}
    for cfg, K in cfgs:
        os.environ["SOAPDENOVO2_AMD_PARSE_THREADS"] = "1"
        try:
            want = api.host_read_all(cfg, K)
        finally:
            del os.environ["SOAPDENOVO2_AMD_PARSE_THREADS"]
        os.environ.update(knobs)
        try:
            got = api.host_read_all(cfg, K)
        finally:
            for k in knobs:
                del os.environ[k]
        assert got[2] == want[2] and got[3] == want[3], cfg
        assert np.array_equal(got[1], want[1]), cfg
        assert np.array_equal(got[0], want[0]), cfg


def test_parallel_mate_reader_stops_with_file_two(tmp_path):
    """q1 longer than q2: the reference stops when file 2 is used up; the multi-threaded mate reader must do the same."""
    import os
    import numpy as np
    a = synth.reads_codes(20000, 1500, 80, 0.004, 21)
    b = synth.reads_codes(20000, 1100, 80, 0.004, 22)
    p = lambda f: os.path.abspath(os.path.join(str(tmp_path), f))
    synth.write_fastq(p("m_1.fq"), a, name_prefix="a")
    synth.write_fastq(p("m_2.fq"), b, name_prefix="b")
    cfg = p("m.cfg")
    open(cfg, "w").write(f"max_rd_len=80\n[LIB]\navg_ins=300\nasm_flags=3\nq1={p('m_1.fq')}\nq2={p('m_2.fq')}\n")
    os.environ["SOAPDENOVO2_AMD_PARSE_THREADS"] = "1"
    try:
        want = api.host_read_all(cfg, 31)
    finally:
        del os.environ["SOAPDENOVO2_AMD_PARSE_THREADS"]
    knobs = {"SOAPDENOVO2_AMD_PARSE_PARALLEL_MIN": "0", "SOAPDENOVO2_AMD_PARSE_THREADS": "4", "SOAPDENOVO2_AMD_PARSE_WINDOW": "2"}
    os.environ.update(knobs)
    try:
        got = api.host_read_all(cfg, 31)
    finally:
        for k in knobs:
            del os.environ[k]
    assert want[2] == 2200 and got[2] == want[2]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[0][0::2], a[:1100]) and np.array_equal(got[0][1::2], b)


def test_simd_fastq_parse_equals_the_scalar_definition(tmp_path):
    """The AVX2 form of the usual FASTQ record (sequence line converted while its end is looked for) hands every record it does
    not fully understand to the scalar forms: same reads with SOAPDENOVO2_AMD_PARSE_SIMD=0, and the same as the sequential chunk
    emulation (always scalar) -- on records with lower case, N, '.', digits, CR LF, reads longer than max_rd_len, reads shorter
    than a vector, '+' lines that repeat the name and quality lines that begin with '@'."""
    import os
    import numpy as np
    rng = np.random.default_rng(5)
    lines = []
    for i in range(60000):
        kind = int(rng.integers(0, 12))
        n = int(rng.integers(1, 40)) if kind == 0 else int(rng.integers(60, 181))
        seq = "".join("ACGT"[c] for c in rng.integers(0, 4, size=n))
        if kind == 1:
            seq = seq.lower()
        elif kind == 2:
            seq = seq[: n // 2] + "N" + seq[n // 2 + 1:]
        elif kind == 3:
            seq = seq[: n // 3] + "." + seq[n // 3 + 1:]
        elif kind == 4:
            seq = seq[: n // 3] + "7" + seq[n // 3 + 1:]
        eol = "\r\n" if kind == 5 else "\n"
        plus = "+" + ("r%d" % i if kind == 6 else "")
        qual = ("@" if kind == 7 else "I") + "I" * (n - 1)
        lines.append("@r%d some text%s%s%s%s%s%s%s" % (i, eol, seq, eol, plus, eol, qual, eol))
    fq = tmp_path / "w.fq"
    fq.write_text("".join(lines), newline="")
    cfg = tmp_path / "w.cfg"
    cfg.write_text(f"max_rd_len=150\n[LIB]\navg_ins=200\nasm_flags=3\nq={fq}\n")
    knobs = {"SOAPDENOVO2_AMD_PARSE_PARALLEL_MIN": "0", "SOAPDENOVO2_AMD_PARSE_THREADS": "3", "SOAPDENOVO2_AMD_PARSE_WINDOW": "7"}
    got = {}
    for tag, extra in (("seq", {"SOAPDENOVO2_AMD_PARSE_THREADS": "1"}), ("scalar", dict(knobs, SOAPDENOVO2_AMD_PARSE_SIMD="0")), ("simd", knobs)):
        os.environ.update(extra)
        try:
            got[tag] = api.host_read_all(str(cfg), 31)
        finally:
            for k in extra:
                del os.environ[k]
    assert got["seq"][2] == 60000
    for tag in ("scalar", "simd"):
        assert got[tag][2] == got["seq"][2] and got[tag][3] == got["seq"][3], tag
        assert np.array_equal(got[tag][1], got["seq"][1]), tag
        assert np.array_equal(got[tag][0], got["seq"][0]), tag


def test_simd_fasta_parse_equals_the_scalar_definition(tmp_path):
    """The same for FASTA (a '>' line, the sequence on one line): lower case, N, '.', digits, CR LF, long and
    very short reads.  (A sequence broken over two lines and a '>' inside a name (the 32 KiB chunks are cut at the last '>') are
    errors to the reference -- "invalid data left in buffer" -- and to this reader.)"""
    import os
    import numpy as np
    rng = np.random.default_rng(6)
    lines = []
    n_rec = 0
    for i in range(50000):
        kind = int(rng.integers(0, 12))
        n = int(rng.integers(1, 40)) if kind == 0 else int(rng.integers(60, 181))
        seq = "".join("ACGT"[c] for c in rng.integers(0, 4, size=n))
        if kind == 1:
            seq = seq.lower()
        elif kind == 2:
            seq = seq[: n // 2] + "N" + seq[n // 2 + 1:]
        elif kind == 3:
            seq = seq[: n // 3] + "." + seq[n // 3 + 1:]
        elif kind == 4:
            seq = seq[: n // 3] + "7" + seq[n // 3 + 1:]
        eol = "\r\n" if kind == 5 else "\n"
        name = ">r%d" % i + (" some text" if kind == 6 else "")
        lines.append(name + eol + seq + eol)
        n_rec += 1
    fa = tmp_path / "w.fa"
    fa.write_text("".join(lines), newline="")
    cfg = tmp_path / "w.cfg"
    cfg.write_text(f"max_rd_len=150\n[LIB]\navg_ins=200\nasm_flags=3\nf={fa}\n")
    knobs = {"SOAPDENOVO2_AMD_PARSE_PARALLEL_MIN": "0", "SOAPDENOVO2_AMD_PARSE_THREADS": "3", "SOAPDENOVO2_AMD_PARSE_WINDOW": "5"}
    got = {}
    for tag, extra in (("seq", {"SOAPDENOVO2_AMD_PARSE_THREADS": "1"}), ("scalar", dict(knobs, SOAPDENOVO2_AMD_PARSE_SIMD="0")), ("simd", knobs)):
        os.environ.update(extra)
        try:
            got[tag] = api.host_read_all(str(cfg), 31)
        finally:
            for k in extra:
                del os.environ[k]
    assert got["seq"][2] == n_rec
    for tag in ("scalar", "simd"):
        assert got[tag][2] == got["seq"][2] and got[tag][3] == got["seq"][3], tag
        assert np.array_equal(got[tag][1], got["seq"][1]), tag
        assert np.array_equal(got[tag][0], got["seq"][0]), tag
