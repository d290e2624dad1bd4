"""Pass 2 of pregraph on the host (read -> edge threading, pre-arcs: pg_host_graph_begin / add_reads / finish) on CPU:
fed with the oracle's pass-1 records and the same reads it must reproduce the reference's .preArc byte for byte
(plus the three graph files again, now written through the handle API)."""
import numpy as np
import pytest

from conftest import case_codes, case_tag, md5_file, md5_gz_text, oracle_records, host_runs
from soapdenovo2_amd import api


@pytest.mark.parametrize("name", ["t6k_k31", "t8k_k63", "t6k_k127", "t5k_k24", "m60k_k63", "d8k_k127", "r8k_k127", "d8k_k63"])
def test_prearc_matches_reference(golden, tmp_path, name):
    c = golden["cases"][name]
    codes = case_codes(c)
    for run in host_runs(c):
        P, D, a, m = run
        t = case_tag(name, run)
        rec, last, K = oracle_records(codes, c["K"], P, D=D, mer127=bool(m), a_gb=a, prefix=str(tmp_path / ("o_" + t)))
        pre = str(tmp_path / t)
        nv, ne, na = api.host_pregraph_files(rec, last, codes, None, K, P, pre, mer127=bool(m), cut_single=(D == 0), a_gb=a,
                                             max_read_len=c["L"], batches=3, resolve_repeats=True)
        want = golden["md5"][t]
        assert md5_file(pre + ".preArc") == want["preArc"], t
        assert md5_file(pre + ".path") == want["path"], t                  # the -R files (prlRead2path.c:478-543, 435-449)
        assert md5_file(pre + ".markOnEdge") == want["markOnEdge"], t
        assert md5_file(pre + ".vertex") == want["vertex"], t
        assert md5_file(pre + ".preGraphBasic") == want["preGraphBasic"], t
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t
        assert na > 0 or D > 0                                             # (-d leaves an error-free 40 kb genome as a single chain)


def test_prearc_on_reader_corner_cases(golden, tmp_path):
    """Ragged reads, truncated reads, mate files: pass 2 sees exactly the reads pass 1 saw."""
    from soapdenovo2_amd import synth
    from oracle_binding import Oracle
    K, P = 31, 3
    for name in synth.QUIRK_CASES:
        cfg = synth.make_quirk_case(str(tmp_path), name)
        codes, lens, _, mrl = api.host_read_all(cfg, K)
        # (BAM: the reference puts its pairing state back at every end of file, readseq1by1.c:584-587, so a second pass over the
        #  inputs -- rq_bam_odd leaves a first mate dangling at the end of its first file -- sees the reads of the first)
        codes2, lens2 = codes, lens
        if name.startswith("rq_bam"):
            assert api.host_bam_state() == -3
            codes2, lens2, _, _ = api.host_read_all(cfg, K)
            assert api.host_bam_state() == -3 and (lens2 == lens).all() and (codes2 == codes).all()
        o = Oracle(K, P=P, max_read_len=mrl)
        o.add_reads(codes, lens=lens)
        o.finish_count(str(tmp_path / ("o_" + name)))
        nd = o.nodes()
        rec = np.zeros((len(nd["A"]), 4), dtype=np.uint64)
        rec[:, :2] = nd["keys"]
        rec[:, 2] = nd["A"].astype(np.uint64) | (nd["B"].astype(np.uint64) << np.uint64(32))
        rec[:, 3] = (nd["set"].astype(np.uint64) << np.uint64(56)) | nd["ord"]
        last = np.array(o.set_last_put(), dtype=np.uint64)
        o.close()
        pre = str(tmp_path / name)
        api.host_pregraph_files(rec, last, codes2, lens2, K, P, pre, max_read_len=mrl)
        assert md5_file(pre + ".preArc") == golden["md5"][name]["preArc"], name
        # the same reads handed over 2-bit packed (pg_host_graph_add_packed: what the executable keeps from pass 1)
        api.host_pregraph_files(rec, last, codes2, lens2, K, P, pre + "_pk", max_read_len=mrl, packed=True, batches=3)
        assert md5_file(pre + "_pk.preArc") == golden["md5"][name]["preArc"], name


def test_contig_stage_accepts_our_files(golden, tmp_path):
    """P3: the reference's unchanged `contig` stage on the files written by the host stages gives the same contigs as on
    the reference's own pregraph output (needs oracle/_ref, which also travels to the GPU box)."""
    import os, subprocess
    from conftest import ROOT
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built")
    for name, run in (("t6k_k31", (8, 0, 0, 0)), ("t8k_k63", (2, 0, 0, 0))):
        c = golden["cases"][name]
        codes = case_codes(c)
        P, D, a, m = run
        t = case_tag(name, run)
        rec, last, K = oracle_records(codes, c["K"], P, prefix=str(tmp_path / ("o_" + t)))
        pre = str(tmp_path / t)
        api.host_pregraph_files(rec, last, codes, None, K, P, pre, max_read_len=c["L"], resolve_repeats=True)
        out = subprocess.run([ref, "contig", "-g", pre], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-500:]
        assert md5_file(pre + ".contig") == golden["md5"][t]["contig"], t
        out = subprocess.run([ref, "contig", "-g", pre, "-R"], capture_output=True, text=True)   # consumes .path / .markOnEdge
        assert out.returncode == 0, out.stderr[-500:]
        assert md5_file(pre + ".contig") == golden["md5"][t]["contigR"], t


@pytest.mark.parametrize("n_threads", [1, 3, 16])
def test_host_stages_do_not_depend_on_thread_count(golden, tmp_path, n_threads):
    """Tips, edges and pre-arcs are built by all host threads (speculative walks, ordered commits): the files must not
    change with the number of threads, nor with the sequential edge builder (PG_SERIAL_EDGES=1)."""
    import os
    name, run = "m60k_k63", (8, 0, 0, 0)
    c = golden["cases"][name]
    codes = case_codes(c)
    P, D, a, m = run
    t = case_tag(name, run)
    rec, last, K = oracle_records(codes, c["K"], P, prefix=str(tmp_path / ("o_" + t)))
    want = golden["md5"][t]
    for serial in ("0", "1") if n_threads == 3 else ("0",):
        os.environ["PG_SERIAL_EDGES"] = serial
        try:
            pre = str(tmp_path / (t + "_" + serial))
            api.host_pregraph_files(rec, last, codes, None, K, P, pre, max_read_len=c["L"], batches=2, n_threads=n_threads,
                                    resolve_repeats=True)
        finally:
            del os.environ["PG_SERIAL_EDGES"]
        for ext in ("preArc", "vertex", "preGraphBasic", "path", "markOnEdge"):
            assert md5_file(pre + "." + ext) == want[ext], (ext, n_threads, serial)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], (n_threads, serial)


@pytest.mark.parametrize("name", ["t6k_k31", "t6k_k127", "m60k_k63"])
def test_streamed_records_give_the_same_graph(golden, tmp_path, name):
    """pg_graph_begin_streamed: records handed over in replay order through a fetch callback (what the executable does
    with the device-sorted records) -- same layout, same files, incl. the -a pools and the trailing-duplicate growth."""
    c = golden["cases"][name]
    codes = case_codes(c)
    for run in host_runs(c):
        P, D, a, m = run
        t = case_tag(name, run)
        rec, last, K = oracle_records(codes, c["K"], P, D=D, mer127=bool(m), a_gb=a, prefix=str(tmp_path / ("o_" + t)))
        pre = str(tmp_path / t)
        api.host_pregraph_files(rec, last, codes, None, K, P, pre, mer127=bool(m), cut_single=(D == 0), a_gb=a, max_read_len=c["L"],
                                streamed=True)
        want = golden["md5"][t]
        assert md5_file(pre + ".vertex") == want["vertex"], t
        assert md5_file(pre + ".preArc") == want["preArc"], t
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


def test_streamed_replay_more_sets_than_threads(golden, tmp_path):
    """pg_graph_begin_streamed with fewer workers than sets: a worker re-uses its chunk buffer from set to set, and a later
    set may hold more records than the one the buffer was first sized for (the per-set counts differ by ~sqrt(N/P))."""
    name, run = "t6k_k31", [7, 0, 0, 0]
    c = golden["cases"][name]
    codes = case_codes(c)
    P, D, a, m = run
    t = case_tag(name, run)
    rec, last, K = oracle_records(codes, c["K"], P, prefix=str(tmp_path / ("o_" + t)))
    counts = np.bincount((rec[:, 3] >> np.uint64(56)).astype(int), minlength=P)
    assert (np.diff(counts) > 0).any()                       # some later set is larger than an earlier one
    want = golden["md5"][t]
    for nt in (1, 2, 3):
        pre = str(tmp_path / f"s{nt}")
        api.host_pregraph_files(rec, last, codes, None, K, P, pre, max_read_len=c["L"], streamed=True, n_threads=nt)
        assert md5_file(pre + ".vertex") == want["vertex"]
        assert md5_file(pre + ".preArc") == want["preArc"]
        assert md5_gz_text(pre + ".edge.gz") == want["edge"]
