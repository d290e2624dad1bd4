"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI against the CPU oracle on the
same seeded inputs and against the golden files produced by the real reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, case_codes, case_config, case_lens, case_tag, md5_file, md5_gz_text, oracle_records

pytestmark = pytest.mark.gpu


def _sorted(rec, nw):
    order = np.lexsort([rec[:, i] for i in range(nw - 1, -1, -1)])
    return rec[order]


ENGINES = [1, 2]       # 1 = DRAM-resident set, 2 = super-k-mer partitions counted in LDS (default)


def _gpu_records(codes, K, P, mer127=False, D=0, log2_slots=20, batches=1, engine=0):
    import torch
    from soapdenovo2_amd import api
    n, L = codes.shape
    kc = api.KmerCounter(K, n_sets=P, mer127=mer127, log2_slots=log2_slots, engine=engine)
    kpr = L - K + 1
    bounds = np.linspace(0, n, batches + 1).astype(int)
    order = list(range(batches))
    if batches > 1:
        order = order[::-1]                 # batches in reverse: the first-occurrence ordinal must not care
    for b in order:
        lo, hi = bounds[b], bounds[b + 1]
        if hi == lo:
            continue
        packed = torch.from_numpy(api.pack_reads_uniform(codes[lo:hi]).view(np.int64)).cuda()
        kc.count_uniform(packed, hi - lo, L, ord_base=int(lo) * kpr)
    hist, last = kc.finalize(D)
    rec = kc.export()
    info = kc.table_info()
    kc.close()
    return rec, hist, last, info


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name,P,m,D", [("t6k_k31", 8, False, 0), ("t6k_k31", 7, False, 1), ("t8k_k63", 2, False, 0),
                                        ("t8k_k63", 5, True, 0), ("t6k_k127", 3, True, 0), ("t5k_k24", 8, False, 0),
                                        ("d8k_k127", 3, True, 0), ("d8k_k63", 5, False, 0),
                                        # -p beyond 8: the set id is hash % thrd_num in an unsigned char (prlHashReads.c:66-126)
                                        ("t6k_k31", 16, False, 0), ("t6k_k31", 255, False, 1), ("t8k_k63", 37, False, 0),
                                        ("t8k_k63", 64, True, 0), ("t6k_k127", 255, True, 0)])
def test_count_matches_oracle(golden, tmp_path, name, P, m, D, engine):
    c = golden["cases"][name]
    codes = case_codes(c)
    want, last_want, K = oracle_records(codes, c["K"], P, D=D, mer127=m, prefix=str(tmp_path / "o"))
    got, hist, last, _ = _gpu_records(codes, K, P, mer127=m, D=D, engine=engine)
    nw = 4 if m else 2
    assert got.shape == want.shape
    assert (_sorted(got, nw) == _sorted(want, nw)).all()         # keys, counters+flags, set id, first ordinal: bit-exact
    assert (last == last_want).all()
    freq = [int(x) for x in open(str(tmp_path / "o.kmerFreq")).read().split()]
    assert [int(x) for x in hist[1:]] == freq


@pytest.mark.parametrize("engine", ENGINES)
def test_growth_and_batch_order(golden, tmp_path, engine):
    """Tiny initial sizing (the set / the record pool and export array grow several times) and batches submitted in
    reverse order."""
    c = golden["cases"]["t6k_k31"]
    codes = case_codes(c)
    want, last_want, K = oracle_records(codes, c["K"], 8, prefix=str(tmp_path / "o"))
    got, hist, last, info = _gpu_records(codes, K, 8, log2_slots=10, batches=5, engine=engine)
    assert info[0] > 1024
    assert (_sorted(got, 2) == _sorted(want, 2)).all()
    assert (last == last_want).all()


@pytest.mark.parametrize("engine", ENGINES)
def test_ragged_batch(tmp_path, engine):
    """Reads of different lengths (K+1 .. 150) in one batch through the prefix-sum path."""
    import torch
    from soapdenovo2_amd import api, synth
    from oracle_binding import Oracle
    K, P = 41, 4
    rng = np.random.default_rng(17)
    base = synth.reads_codes(20000, 3000, 150, 0.004, 99)
    lens = rng.integers(K + 1, 151, size=3000)
    lens[:5] = K + 1
    reads = [base[i, : lens[i]].copy() for i in range(3000)]
    o = Oracle(K, P=P, max_read_len=150)
    o.add_reads(base, lens=lens)
    o.finish_count(str(tmp_path / "o"))
    nd = o.nodes()
    want = np.zeros((len(nd["A"]), 4), dtype=np.uint64)
    want[:, :2] = nd["keys"]
    want[:, 2] = nd["A"].astype(np.uint64) | (nd["B"].astype(np.uint64) << np.uint64(32))
    want[:, 3] = (nd["set"].astype(np.uint64) << np.uint64(56)) | nd["ord"]
    o.close()
    words, off, kb = api.pack_reads_ragged(reads, K)
    kc = api.KmerCounter(K, n_sets=P, log2_slots=18, engine=engine)
    kc.count_ragged(torch.from_numpy(words.view(np.int64)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(),
                    torch.from_numpy(kb.view(np.int64)).cuda(), len(reads), int(kb[-1]))
    kc.finalize(0)
    got = kc.export()
    kc.close()
    assert (_sorted(got, 2) == _sorted(want, 2)).all()


@pytest.mark.parametrize("bound", ["exact", "loose", "none"])
@pytest.mark.parametrize("name,m127", [("g120k_k63", False), ("g40k_k127", True), ("g60k_k31", False)])
def test_ragged_tiles_match_oracle(golden, tmp_path, name, m127, bound):
    """Trimmed reads -- every batch ragged -- through the TILED cutter (round 6: rows for the batch's longest read, a read's own length
    masks its segments) in several batches: keys, counters, flags, set ids and first ordinals bit-exact against the oracle at 40 - 120 k
    reads.  With the longest read given exactly, given loosely, and not given (the device finds it)."""
    import torch
    from soapdenovo2_amd import api
    from oracle_binding import Oracle
    c = golden["cases"][name]
    codes, lens = case_codes(c), case_lens(c)
    K, P = c["K"], 5
    o = Oracle(K, P=P, mer127=m127, max_read_len=c["L"])
    o.add_reads(codes, lens=lens)
    o.finish_count(str(tmp_path / "o"))
    nd = o.nodes()
    nw = o.NW
    want = np.zeros((len(nd["A"]), nw + 2), dtype=np.uint64)
    want[:, :nw] = nd["keys"]
    want[:, nw] = nd["A"].astype(np.uint64) | (nd["B"].astype(np.uint64) << np.uint64(32))
    want[:, nw + 1] = (nd["set"].astype(np.uint64) << np.uint64(56)) | nd["ord"]
    o.close()
    kc = api.KmerCounter(K, n_sets=P, mer127=m127, log2_slots=22)
    n = codes.shape[0]
    bounds = [0, 1, 25, n // 3, n // 3 + 24, n]                    # a batch of one read, one of a single full tile, large ones
    ord_base = 0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        reads = [codes[i, :lens[i]] for i in range(lo, hi)]
        words, off, kb = api.pack_reads_ragged(reads, K)
        kc.set_read_len_bound({"exact": int(lens[lo:hi].max()), "loose": c["L"] + 37, "none": 0}[bound])
        kc.count_ragged(torch.from_numpy(words.view(np.int64)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(),
                        torch.from_numpy(kb.view(np.int64)).cuda(), len(reads), int(kb[-1]), ord_base=ord_base)
        ord_base += int(kb[-1])
    kc.finalize(0)
    got = kc.export()
    kc.close()
    assert got.shape == want.shape
    assert (_sorted(got, nw) == _sorted(want, nw)).all()


def test_ragged_read_longer_than_the_bound_fails_loudly(golden):
    """A ragged batch with a read longer than pg_set_read_len_bound said: the pass fails in pg_finalize, nothing is cut wrongly."""
    import torch
    from soapdenovo2_amd import api
    c = golden["cases"]["g60k_k31"]
    codes, lens = case_codes(c)[:500], case_lens(c)[:500]
    K = c["K"]
    reads = [codes[i, :lens[i]] for i in range(500)]
    words, off, kb = api.pack_reads_ragged(reads, K)
    kc = api.KmerCounter(K, n_sets=3, log2_slots=18)
    kc.set_read_len_bound(int(lens.max()) - 1)
    kc.count_ragged(torch.from_numpy(words.view(np.int64)).cuda(), torch.from_numpy(off.view(np.int64)).cuda(),
                    torch.from_numpy(kb.view(np.int64)).cuda(), len(reads), int(kb[-1]))
    with pytest.raises(api.PgError, match="longer than the bound"):
        kc.finalize(0)
    kc.close()


def test_route_then_count_equals_fused(golden, tmp_path):
    """The multi-GPU data path on one GPU: extract + route to 2 owners, insert each owner's records into its own
    set; the union must equal the fused single-set result, and owners must partition the reference sets."""
    import torch
    from soapdenovo2_amd import api
    c = golden["cases"]["t8k_k63"]
    codes = case_codes(c)
    K, P, L, n = c["K"], 8, c["L"], codes.shape[0]
    want, last_want, _ = oracle_records(codes, K, P, prefix=str(tmp_path / "o"))
    packed = torch.from_numpy(api.pack_reads_uniform(codes).view(np.int64)).cuda()
    router = api.KmerCounter(K, n_sets=P, log2_slots=10, engine=1)
    owners = 2
    counts = router.route_count(packed, n, L, owners)
    off = torch.zeros(owners + 1, dtype=torch.int64, device="cuda")
    off[1:] = torch.cumsum(counts, 0)
    total = int(off[-1])
    assert total == n * (L - K + 1)
    out = torch.empty(total * 3, dtype=torch.int64, device="cuda")
    router.route_scatter(packed, n, L, 0, owners, off, out)
    torch.cuda.synchronize()
    parts, lasts = [], []
    for o in range(owners):
        lo, hi = int(off[o]), int(off[o + 1])
        kc = api.KmerCounter(K, n_sets=P, log2_slots=18, engine=1)
        kc.count_records(out[lo * 3: hi * 3], hi - lo)
        _, last = kc.finalize(0)
        rec = kc.export()
        assert ((rec[:, 3] >> np.uint64(56)) % owners == o).all()
        parts.append(rec)
        lasts.append(last)
        kc.close()
    router.close()
    got = np.concatenate(parts)
    assert (_sorted(got, 2) == _sorted(want, 2)).all()
    assert (np.maximum(lasts[0], lasts[1]) == last_want).all()


def test_skm_route_then_ingest_equals_local(golden, tmp_path):
    """The multi-GPU data path of the partition engine on one GPU: cut + route to 3 owners, ingest each owner's records
    into its own context; owners must partition the partitions and the union must equal the oracle."""
    import torch
    from soapdenovo2_amd import api
    c = golden["cases"]["t8k_k63"]
    codes = case_codes(c)
    K, P, L, n = c["K"], 8, c["L"], codes.shape[0]
    want, last_want, _ = oracle_records(codes, K, P, prefix=str(tmp_path / "o"))
    packed = torch.from_numpy(api.pack_reads_uniform(codes).view(np.int64)).cuda()
    owners = 3
    router = api.KmerCounter(K, n_sets=P, log2_slots=18, engine=2)
    recs, parts, counts = router.skm_route(packed, n, L, 0, owners, 40000)
    torch.cuda.synchronize()
    cn = counts.tolist()
    assert n <= sum(cn) <= n * (L - K + 1) and max(cn) < 40000
    got, lasts = [], []
    for o in range(owners):
        assert bool((parts[o, :cn[o]] % owners == o).all())
        kc = api.KmerCounter(K, n_sets=P, log2_slots=18, engine=2)
        kc.skm_ingest(recs[o, :cn[o]].contiguous(), parts[o, :cn[o]].contiguous(), cn[o])
        _, last = kc.finalize(0)
        got.append(kc.export())
        lasts.append(last)
        kc.close()
    router.close()
    allrec = np.concatenate(got)
    assert (_sorted(allrec, 2) == _sorted(want, 2)).all()
    assert (np.maximum.reduce(lasts) == last_want).all()


PARALLEL_PARSE = {"SOAPDENOVO2_AMD_PARSE_PARALLEL_MIN": "0", "SOAPDENOVO2_AMD_PARSE_THREADS": "4", "SOAPDENOVO2_AMD_PARSE_WINDOW": "8"}


def _run_cli(cfg, K, prefix, P, D, a, m, engine=None, R=False, extra_env=None):
    from soapdenovo2_amd import api
    env = dict(os.environ)
    if extra_env:
        env.update(extra_env)
    if engine:
        env["PG_ENGINE"] = str(engine)
    args = ["-s", cfg, "-K", str(K), "-o", prefix, "-p", str(P)]
    if D:
        args += ["-d", str(D)]
    if a:
        args += ["-a", str(a)]
    if R:
        args += ["-R"]
    rc = subprocess.run([api.binary(bool(m)), "pregraph"] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    assert rc.returncode == 0, rc.stderr[-2000:]
    return rc.stderr


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name", ["t6k_k31", "t8k_k63", "t6k_k127", "t5k_k24", "m100k_k31", "m60k_k63", "d8k_k127", "r8k_k127", "d8k_k63",
                                  "g120k_k63", "g40k_k127", "g60k_k31", "x500_k63", "x400_k127", "y300_k63"])
def test_cli_matches_reference_files(golden, tmp_path, name, engine):
    """`SOAPdenovo-63mer|127mer pregraph -s cfg -K k -o pfx -p n [-d -a]` end to end against the reference's files."""
    from soapdenovo2_amd import synth
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        # engine 2 also takes the multi-threaded reader (packed runs straight into the pinned batches)
        _run_cli(cfg, c["K"], pre, P, D, a, m, engine=engine, R=(engine == 2), extra_env=PARALLEL_PARSE if engine == 2 else None)
        want = golden["md5"][t]
        if engine == 2:                                                # -R: the read paths and the per-edge marker counts
            assert md5_file(pre + ".path") == want["path"], t
            assert md5_file(pre + ".markOnEdge") == want["markOnEdge"], t
        assert md5_file(pre + ".kmerFreq") == want["kmerFreq"], t
        assert md5_file(pre + ".preGraphBasic") == want["preGraphBasic"], t
        assert md5_file(pre + ".vertex") == want["vertex"], t
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t
        assert md5_file(pre + ".preArc") == want["preArc"], t
    # P3: the reference's unchanged contig stage consumes the files of the last run and produces the same contigs
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-127mer" if m else "SOAPdenovo-63mer")
    if engine == 2 and os.path.exists(ref):
        out = subprocess.run([ref, "contig", "-g", pre], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-500:]
        assert md5_file(pre + ".contig") == golden["md5"][t]["contig"], t


@pytest.mark.parametrize("form,says", [("SOAPDENOVO2_AMD_P2_SORT", "sorted by their smallest hashed 16-mer"),
                                       ("SOAPDENOVO2_AMD_P2_PARTITIONED", "through the partition engine"),
                                       ("SOAPDENOVO2_AMD_P2_LOOK", "pass 2: lookup table of")])
@pytest.mark.parametrize("name", ["t6k_k31", "t8k_k63", "t6k_k127", "d8k_k63"])
def test_cli_pass2_opt_in_forms_write_the_same_files(golden, tmp_path, name, form, says):
    """Pass 2's three opt-in forms of the lookups (round 6: reads in genome order; every distinct k-mer of a partition looked up once through the
    partition engine; a line-aligned lookup table beside the sets -- graph_kernels.hip, profiles/r06_p2_lookup_ab.json) are not the default, but they
    ship: each must say that it ran and leave the reference's five files (prlRead2path.c:159-248,388-403: a pre-arc is a multiplicity and a first
    meeting, whatever order the reads were threaded in and wherever the node words came from)."""
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        log = _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env=dict(PARALLEL_PARSE, PG_HOST_VERBOSE="1", **{form: "1"}))
        assert says in log, (t, form)
        want = golden["md5"][t]
        assert md5_file(pre + ".kmerFreq") == want["kmerFreq"], t
        assert md5_file(pre + ".preGraphBasic") == want["preGraphBasic"], t
        assert md5_file(pre + ".vertex") == want["vertex"], t
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t
        assert md5_file(pre + ".preArc") == want["preArc"], t


@pytest.mark.parametrize("period", [1, 4, 32])
@pytest.mark.parametrize("name", ["t6k_k31", "t8k_k63", "t6k_k127", "m100k_k31", "m60k_k63", "d8k_k127", "r8k_k127", "d8k_k63"])
def test_cli_edges_through_waypoints(golden, tmp_path, name, period):
    """The edge builder's jumps (graph_kernels.hip: EbWay): with every linear node (period 1), every 4th or every 32nd a waypoint, the
    chains of the golden cases -- a few dozen to a few thousand nodes, SNP bubbles with length-1 edges, repeats, both flavours -- are walked
    by jumps and written by the segment lanes: same edge text, same ids, same tags (the pre-arcs of pass 2 read them), same -R files."""
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        log = _run_cli(cfg, c["K"], pre, P, D, a, m, R=True, extra_env=dict(PARALLEL_PARSE, PG_HOST_VERBOSE="1", SOAPDENOVO2_AMD_EB_WAYPOINTS=str(period)))
        assert "waypoint(s)" in log, log[-1500:]
        want = golden["md5"][t]
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc", "path", "markOnEdge"):
            assert md5_file(pre + "." + ext) == want[ext], (t, ext)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


def test_cli_static_pools_under_load(golden, tmp_path):
    """-a pools at a realistic load: the smallest pool is 16.7 M slots a set (prlHashReads.c:372-390), so the other -a fixtures never
    collide.  1.5 M reads x 100 bp over a 12 Mb genome, K = 31, -p 2 -a 1: two sets of 33.5 M slots at about 52 % -- probe clusters of
    dozens of keys, first-come-first-served linear probing in first-occurrence order as the device layout (dev_graph.hpp: layout_static,
    K6Sweep) has to rebuild it, the cluster that runs round the end of a table.  The reference's md5s are in the golden file; one rank
    and three ranks (the sets on two of them)."""
    import re
    name = "l1500k_k31"
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    run = c["runs"][0]
    P, D, a, m = run
    t = case_tag(name, run)
    want = golden["md5"][t]
    for tag, env in (("one", dict(PARALLEL_PARSE, PG_HOST_VERBOSE="1")), ("three", dict(PARALLEL_PARSE, PG_HOST_VERBOSE="1", SOAPDENOVO2_AMD_DEVICES="0,0,0"))):
        pre = str(tmp_path / (t + "_" + tag))
        log = _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env=env)
        assert "K6 on device" in log, log[-2000:]                          # the layout was made on the device, not replayed
        nodes = int(re.search(r"(\d+) node\(s\) allocated", log).group(1))
        assert 0.40 < nodes / (2 * 2 * 0xFFFFFF) < 0.70, nodes             # (the load the case was made for)
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
            assert md5_file(pre + "." + ext) == want[ext], (tag, ext)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], tag


def test_cli_fasta_and_reference_binary(golden, tmp_path):
    """FASTA input (f=) gives the same files as FASTQ (q=); and, when the reference binary travelled with the
    snapshot (oracle/_ref), a direct byte comparison on a fresh seed that has no golden file."""
    from soapdenovo2_amd import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
    G, N, L, err, seed, K, P = 60000, 20000, 120, 0.004, 4242, 45, 6
    cfq = synth.make_case(str(tmp_path), "fq", G, N, L, err, seed, fmt="fastq")
    cfa = synth.make_case(str(tmp_path), "fa", G, N, L, err, seed, fmt="fasta")
    _run_cli(cfq, K, str(tmp_path / "a"), P, 0, 0, 0)
    _run_cli(cfa, K, str(tmp_path / "b"), P, 0, 0, 0)
    for ext in ("kmerFreq", "preGraphBasic", "vertex"):
        assert md5_file(str(tmp_path / ("a." + ext))) == md5_file(str(tmp_path / ("b." + ext)))
    assert md5_gz_text(str(tmp_path / "a.edge.gz")) == md5_gz_text(str(tmp_path / "b.edge.gz"))
    if os.path.exists(ref):
        subprocess.run([ref, "pregraph", "-s", cfq, "-K", str(K), "-o", str(tmp_path / "r"), "-p", str(P)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
            assert md5_file(str(tmp_path / ("a." + ext))) == md5_file(str(tmp_path / ("r." + ext))), ext
        assert md5_gz_text(str(tmp_path / "a.edge.gz")) == md5_gz_text(str(tmp_path / "r.edge.gz"))


def test_full_size_properties():
    """Size-independent properties at a bench-like size (2 M reads x 150 bp, K = 63): every occurrence is
    accounted for, the histogram covers every stored k-mer, counting the same reads twice doubles nothing but
    the saturating counters, and a second pass leaves first-occurrence ordinals untouched."""
    import torch
    from soapdenovo2_amd import api, synth
    K, L, n = 63, 150, 2_000_000
    codes = synth.reads_codes(2_000_000, n, L, 0.001, 31)
    packed = torch.from_numpy(api.pack_reads_uniform(codes).view(np.int64)).cuda()
    kc = api.KmerCounter(K, n_sets=8, log2_slots=26)
    kc.count_uniform(packed, n, L, 0)
    hist, last = kc.finalize(0)
    d1 = kc.distinct()
    r1 = kc.export()
    assert int(hist.sum()) == d1 == r1.shape[0]
    assert int(last.max()) == n * (L - K + 1)                      # the very last occurrence went somewhere
    cov = (r1[:, 2] & np.uint64(0xFFFFFFFF)) >> np.uint64(24)
    # total coverage (unsaturated part) equals the number of occurrences
    unsat = cov < 255
    assert int(cov[unsat].sum()) + int((~unsat).sum()) * 255 <= n * (L - K + 1)
    kc.reset()
    kc.count_uniform(packed, n, L, 0)
    kc.count_uniform(packed, n, L, n * (L - K + 1))                # same reads again, later ordinals
    kc.finalize(0)
    assert kc.distinct() == d1
    r2 = kc.export()
    kc.close()
    s1, s2 = _sorted(r1, 2), _sorted(r2, 2)
    assert (s1[:, :2] == s2[:, :2]).all()
    assert (s1[:, 3] == s2[:, 3]).all()                            # first ordinals and set ids unchanged
    cov2 = (s2[:, 2] & np.uint64(0xFFFFFFFF)) >> np.uint64(24)
    cov1 = (s1[:, 2] & np.uint64(0xFFFFFFFF)) >> np.uint64(24)
    assert (cov2 == np.minimum(2 * cov1, 255)).all()


def test_cli_reader_corner_cases(golden, tmp_path):
    """The executable on the reader corner cases (N x 32768 file, truncation / lib order / reverse_seq, ragged reads,
    mate files) against the reference's files."""
    from soapdenovo2_amd import synth
    for name in synth.QUIRK_CASES:
        cfg = synth.make_quirk_case(str(tmp_path), name)
        pre = str(tmp_path / ("cli_" + name))
        # multi-threaded reader, and the records downloaded whole instead of streamed into the layout replay
        _run_cli(cfg, 31, pre + "_par", 3, 0, 0, 0, extra_env=dict(PARALLEL_PARSE, SOAPDENOVO2_AMD_STREAM_RECORDS="0"))
        for ext in ("kmerFreq", "vertex", "preArc"):
            assert md5_file(pre + "_par." + ext) == golden["md5"][name][ext], (name, ext)
        log = _run_cli(cfg, 31, pre, 3, 0, 0, 0)
        q = golden["quirks"][name]
        assert f"{q['reads_processed']} read(s) processed" in log
        assert f"{q['nodes']} node(s) allocated, {q['kmers']} kmer(s) in reads" in log
        want = golden["md5"][name]
        assert md5_file(pre + ".kmerFreq") == want["kmerFreq"], name
        assert md5_file(pre + ".preGraphBasic") == want["preGraphBasic"], name
        assert md5_file(pre + ".vertex") == want["vertex"], name
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], name
        assert md5_file(pre + ".preArc") == want["preArc"], name


@pytest.mark.parametrize("name", ["t6k_k31", "t8k_k63", "t6k_k127", "t5k_k24", "m60k_k63", "d8k_k127", "r8k_k127", "d8k_k63"])
def test_device_pass2_matches_reference(golden, tmp_path, name):
    """pg_graph_use_device: read -> edge threading and the pre-arc table on the GPU (graph_kernels.hip), fed with the
    oracle's pass-1 records: .preArc, and with -R .path / .markOnEdge, byte for byte as the reference wrote them."""
    from conftest import case_codes, oracle_records
    from soapdenovo2_amd import api
    c = golden["cases"][name]
    codes = case_codes(c)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        rec, last, K = oracle_records(codes, c["K"], P, D=D, mer127=bool(m), a_gb=a, prefix=str(tmp_path / ("o_" + t)))
        want = golden["md5"][t]
        for reps in (False, True):
            pre = str(tmp_path / (t + ("_R" if reps else "")))
            # without -R: host edges + device pass 2; with -R: the edges come from the device too (pg_graph_begin)
            nv, ne, na = api.host_pregraph_files(rec, last, codes, None, K, P, pre, mer127=bool(m), cut_single=(D == 0), a_gb=a,
                                                 max_read_len=c["L"], batches=3, resolve_repeats=reps, device=0, packed=reps,
                                                 device_edges=reps)
            assert md5_file(pre + ".preArc") == want["preArc"], (t, reps)
            assert md5_file(pre + ".vertex") == want["vertex"], (t, reps)
            assert md5_file(pre + ".preGraphBasic") == want["preGraphBasic"], (t, reps)
            assert md5_gz_text(pre + ".edge.gz") == want["edge"], (t, reps)
            if reps:
                assert md5_file(pre + ".path") == want["path"], t
                assert md5_file(pre + ".markOnEdge") == want["markOnEdge"], t


def test_device_pass2_on_reader_corner_cases(golden, tmp_path):
    """Ragged / truncated / mate-file reads through the device pass 2 (both hand-over formats)."""
    from soapdenovo2_amd import api, synth
    from oracle_binding import Oracle
    K, P = 31, 3
    for name in synth.QUIRK_CASES:
        cfg = synth.make_quirk_case(str(tmp_path), name)
        codes, lens, _, mrl = api.host_read_all(cfg, K)
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
        for packed in (False, True):
            pre = str(tmp_path / (name + ("_pk" if packed else "")))
            api.host_pregraph_files(rec, last, codes, lens, K, P, pre, max_read_len=mrl, device=0, packed=packed, batches=2,
                                    device_edges=packed)
            assert md5_file(pre + ".preArc") == golden["md5"][name]["preArc"], (name, packed)
            assert md5_gz_text(pre + ".edge.gz") == golden["md5"][name]["edge"], (name, packed)


@pytest.mark.parametrize("K,m,wide", [(31, 0, False), (127, 1, False), (31, 0, True), (127, 1, True)])
def test_sort_records_is_replay_order(K, m, wide, monkeypatch):
    """pg_sort_records: exported records ordered by (set, first ordinal) -- the same multiset, sorted by the last word.
    wide: the flavour for more than 2^32 - 1 records (64-bit permutation indices), forced on a small input."""
    import torch
    from soapdenovo2_amd import api, synth
    if wide:
        monkeypatch.setenv("PG_SORT_WIDE", "1")
    L, n = 150, 40000
    codes = synth.reads_codes(50000, n, L, 0.004, 77)
    packed = torch.from_numpy(api.pack_reads_uniform(codes).view(np.int64)).cuda()
    kc = api.KmerCounter(K, n_sets=7, mer127=bool(m), log2_slots=22)
    kc.count_uniform(packed, n, L, 0)
    kc.finalize(0)
    a = kc.export()
    b = kc.export(sort=True)
    assert a.shape == b.shape
    tags = b[:, -1]
    assert (tags[1:] > tags[:-1]).all()                        # distinct ordinals within a set, sets ascending
    order = np.argsort(a[:, -1], kind="stable")
    assert np.array_equal(a[order], b)


def test_cli_falls_back_to_the_global_set_on_a_skewed_partition(golden, tmp_path):
    """With 16 partitions forced, a partition outgrows its chunk list: the executable must notice, count again with the
    global-set engine (another HIP path, not the CPU) and still write the reference's files."""
    from soapdenovo2_amd import synth
    name = "m100k_k31"
    c = golden["cases"][name]
    cfg = synth.make_case(str(tmp_path), name, c["G"], c["N"], c["L"], c["err"], c["seed"])
    run = c["runs"][0]
    P, D, a, m = run
    t = case_tag(name, run)
    pre = str(tmp_path / t)
    log = _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env={"PG_LOG2_PARTS": "4"})
    assert "counting again with the global k-mer set" in log
    want = golden["md5"][t]
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        assert md5_file(pre + "." + ext) == want[ext], ext
    assert md5_gz_text(pre + ".edge.gz") == want["edge"]


@pytest.mark.parametrize("opts", [["-K", "30"], ["-K", "11"], ["-K", "65"], ["-K", "25", "-p", "1"], ["-K", "31", "-d", "2", "-R", "-p", "5"],
                                  ["-K", "41", "-a", "1", "-p", "3"]])
def test_cli_option_semantics_against_the_reference_binary(tmp_path, opts):
    """Option handling that leaks into the bytes (even K -> K + 1, K < 13 -> 13, K > 63 -> 63, -p 1, -d with -R, -a):
    the executable and the reference binary (oracle/_ref travels with the snapshot) on the same small input."""
    from soapdenovo2_amd import api, synth
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built")
    cfg = synth.make_case(str(tmp_path), "opt", 25000, 5000, 90, 0.006, 1234)
    outs = {}
    for tag, binary in (("amd", api.binary(False)), ("ref", ref)):
        pre = str(tmp_path / tag)
        r = subprocess.run([binary, "pregraph", "-s", cfg, "-o", pre] + opts, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        outs[tag] = pre
    exts = ["kmerFreq", "preGraphBasic", "vertex", "preArc"] + (["path", "markOnEdge"] if "-R" in opts else [])
    for ext in exts:
        assert md5_file(outs["amd"] + "." + ext) == md5_file(outs["ref"] + "." + ext), (opts, ext)
    assert md5_gz_text(outs["amd"] + ".edge.gz") == md5_gz_text(outs["ref"] + ".edge.gz"), opts


@pytest.mark.parametrize("opts", [["-K", "31"], ["-K", "76"], ["-K", "129"], ["-K", "13", "-p", "2"], ["-K", "63", "-d", "1", "-R", "-p", "3"]])
def test_cli_option_semantics_of_the_127mer_flavour_against_the_reference_binary(tmp_path, opts):
    """The same for SOAPdenovo-127mer (four-word k-mers whatever K is): a small K in the wide flavour, even K -> K + 1, K > 127 -> 127 (on reads of
    150 bases: with K beyond max_rd_len the reference asks for a negative number of bytes and dies -- no behaviour to match), K = 13, -d with -R."""
    from soapdenovo2_amd import api, synth
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-127mer")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built")
    cfg = synth.make_case(str(tmp_path), "opt", 25000, 5000 if int(opts[1]) < 100 else 3000, 90 if int(opts[1]) < 100 else 150, 0.006, 1234)
    outs = {}
    for tag, binary in (("amd", api.binary(True)), ("ref", ref)):
        pre = str(tmp_path / tag)
        r = subprocess.run([binary, "pregraph", "-s", cfg, "-o", pre] + opts, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        outs[tag] = pre
    exts = ["kmerFreq", "preGraphBasic", "vertex", "preArc"] + (["path", "markOnEdge"] if "-R" in opts else [])
    for ext in exts:
        assert md5_file(outs["amd"] + "." + ext) == md5_file(outs["ref"] + "." + ext), (opts, ext)
    assert md5_gz_text(outs["amd"] + ".edge.gz") == md5_gz_text(outs["ref"] + ".edge.gz"), opts


@pytest.mark.parametrize("kind", ["empty", "all_short"])
def test_cli_degenerate_inputs_against_the_reference_binary(tmp_path, kind):
    """No read at all / no read of K + 1 bases: both binaries must finish and write the same (empty) graph."""
    from soapdenovo2_amd import api, synth
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built")
    fq = str(tmp_path / "d.fq")
    if kind == "empty":
        open(fq, "w").close()
    else:
        synth.write_fastq(fq, synth.reads_codes(5000, 300, 25, 0.0, 3))
    cfg = str(tmp_path / "d.cfg")
    open(cfg, "w").write(f"max_rd_len=100\n[LIB]\navg_ins=200\nasm_flags=3\nq={fq}\n")
    for tag, binary in (("amd", api.binary(False)), ("ref", ref)):
        r = subprocess.run([binary, "pregraph", "-s", cfg, "-K", "31", "-o", str(tmp_path / tag), "-p", "2"], stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, (tag, r.stderr[-1500:])
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        assert md5_file(str(tmp_path / ("amd." + ext))) == md5_file(str(tmp_path / ("ref." + ext))), ext
    assert md5_gz_text(str(tmp_path / "amd.edge.gz")) == md5_gz_text(str(tmp_path / "ref.edge.gz"))


# ---- multi-GPU pass 1 through product code (exchange.hip), ranks sharing the one GPU of the test box --------------------
def _sharded_records(codes, K, P, n_ranks, mer127=False, batches=5, transport=-1, rccl_world1=False):
    """n_ranks ranks (host threads) on cuda:0: batches are dealt to the ranks in turn, a round = one batch per rank (the last
    round leaves ranks empty-handed), every rank runs pg_count_reads_sharded per round and pg_finalize at the end."""
    import threading
    import torch
    from soapdenovo2_amd import api
    n, L = codes.shape
    kpr = L - K + 1
    if rccl_world1:
        comms = [api.Comm.rccl(1, 0, 0, api.Comm.unique_id())]
    else:
        comms = api.Comm.local([0] * n_ranks, transport)
    kcs = [api.KmerCounter(K, n_sets=P, mer127=mer127, log2_slots=18, engine=2) for _ in range(n_ranks)]
    bounds = np.linspace(0, n, batches + 1).astype(int)
    rounds = (batches + n_ranks - 1) // n_ranks
    packed = [torch.from_numpy(api.pack_reads_uniform(codes[bounds[b]:bounds[b + 1]]).view(np.int64)).cuda() for b in range(batches)]
    out, errs = [None] * n_ranks, []

    def rank_main(r):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rd in range(rounds):
                    b = rd * n_ranks + r
                    if b < batches:
                        kcs[r].count_sharded(comms[r], packed[b], int(bounds[b + 1] - bounds[b]), L, int(bounds[b]) * kpr)
                    else:
                        kcs[r].count_sharded(comms[r], None, 0, L, 0)
                hist, last = kcs[r].finalize(0)
                d_h = torch.from_numpy(hist.view(np.int64)).cuda()
                comms[r].allreduce_u64(d_h, C_void(st))
                st.synchronize()
                out[r] = (kcs[r].export(), d_h.cpu().numpy().view(np.uint64), last, comms[r].stats(), comms[r].transport)
        except Exception as e:                                   # a dead rank would leave the others at a barrier
            import sys
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(17)

    import ctypes
    C_void = lambda s: ctypes.c_void_p(s.cuda_stream)
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for kc in kcs:
        kc.close()
    for c in comms:
        c.close()
    return out


@pytest.mark.parametrize("name,P,m,n_ranks", [("m60k_k63", 8, False, 2), ("m60k_k63", 8, False, 3), ("t8k_k63", 5, True, 3),
                                               ("t6k_k127", 3, True, 2), ("t6k_k31", 7, False, 3)])
def test_sharded_pass1_ranks_on_one_gpu(golden, tmp_path, name, P, m, n_ranks):
    c = golden["cases"][name]
    codes = case_codes(c)
    want, last_want, K = oracle_records(codes, c["K"], P, mer127=m, prefix=str(tmp_path / "o"))
    out = _sharded_records(codes, K, P, n_ranks, mer127=m)
    nw = 4 if m else 2
    got = np.concatenate([o[0] for o in out])
    assert got.shape == want.shape                                                   # no k-mer counted on two ranks
    assert (_sorted(got, nw) == _sorted(want, nw)).all()
    assert (np.maximum.reduce([o[2] for o in out]) == last_want).all()
    freq = [int(x) for x in open(str(tmp_path / "o.kmerFreq")).read().split()]
    for o in out:
        assert [int(x) for x in o[1][1:]] == freq                                    # every rank holds the all-reduced histogram
        assert o[4] == "p2p" and o[3]["rounds"] == (5 + n_ranks - 1) // n_ranks
    assert sum(o[3]["sent_records"] for o in out) == sum(o[3]["recv_records"] for o in out) > 0
    assert all(len(o[0]) > 0 for o in out)


@pytest.mark.gpu
@pytest.mark.parametrize("K,reads,L,log2_slots", [(127, 20_000_000, 150, 28), (127, 60_000_000, 150, 30), (63, 60_000_000, 150, 30), (63, 200_000_000, 150, 31), (31, 10_000_000, 100, 28),
                                                  (25, 3_000_000, 80, 26)])
def test_context_sized_like_the_command_legs_can_be_created(K, reads, L, log2_slots):
    """pg_create_sized with the k-mer estimate call_pregraph derives for the bench's command legs (allocation only, nothing is counted): the
    record pool must satisfy the engine's own floor -- the computed-address region + one spare chunk a partition -- at every geometry.  (Round
    5: a pool trimmed by too much passed every golden case and failed `pregraph -K 127` at 20 M reads with "record pool too small".)"""
    from soapdenovo2_amd import api
    L_ = api.lib()
    n_kmers = reads * (L - K + 1)
    before = api.arena_stats(0)["in_use"]
    h = L_.pg_create_sized(0, K, 1 if K > 63 else 0, 8, log2_slots, 2, n_kmers)
    assert h, L_.pg_last_error().decode()
    st = (C.c_uint64 * 8)()
    assert L_.pg_stats(h, C.cast(st, C.POINTER(C.c_uint64))) == 0
    assert st[0] == 2 and st[5] > st[6]                                      # engine 2; more pool chunks than partitions
    L_.pg_destroy(h)
    a = api.arena_stats(0)
    assert a["active"] in (0, 1) and a["in_use"] == before                    # everything went back to the arena (and, unpinned and empty, to the driver)


@pytest.mark.gpu
def test_device_arena_gives_memory_back_when_empty_and_keeps_it_while_pinned():
    """csrc/arena.hpp through the C ABI: a context's blocks come out of the arena (in use > 0, physical memory mapped); pg_create pins the arena for
    the process unless the caller holds a pin (round 6), pg_device_arena_unpin gives that up and an empty unpinned arena gives its physical memory
    back to the driver; an explicitly pinned arena keeps it across contexts (what call_pregraph does for the length of a command) until the unpin."""
    from soapdenovo2_amd import api
    L_ = api.lib()
    if not api.arena_stats(0)["active"] and os.environ.get("SOAPDENOVO2_AMD_ARENA") == "0":
        pytest.skip("the arena is switched off")
    import gc
    gc.collect()                                                              # (contexts of earlier tests that were left to the collector)
    base = api.arena_stats(0)
    alone = base["in_use"] == 0                                               # nothing else of this process holds a block
    h = L_.pg_create_sized(0, 31, 0, 8, 24, 2, 70_000_000)
    assert h, L_.pg_last_error().decode()
    a = api.arena_stats(0)
    assert a["active"] == 1 and a["in_use"] > base["in_use"] and a["mapped"] > 0 and a["reserved"] >= a["mapped"]     # (the export array's thread may still be backing its block)
    L_.pg_destroy(h)
    b = api.arena_stats(0)
    assert b["in_use"] == base["in_use"]
    # the API path: the first pg_create without a pin of the caller's pins the arena for the PROCESS (a caller that cycles contexts does not pay an
    # unmap + a fresh reservation + new pieces per cycle, nor retire an address range each time): the pieces stay, the next context is cut from them
    assert b["mapped"] > 0
    made = b["pieces_created"]
    h = L_.pg_create_sized(0, 31, 0, 8, 24, 2, 70_000_000)
    assert h and api.arena_stats(0)["pieces_created"] == made
    L_.pg_destroy(h)
    L_.pg_device_arena_unpin(0)                                               # ... until the caller gives the pin up
    if alone:
        assert api.arena_stats(0)["mapped"] == 0                              # empty and unpinned: back to the driver
    with api.arena_pinned(0):
        h = L_.pg_create_sized(0, 31, 0, 8, 24, 2, 70_000_000)
        assert h
        L_.pg_destroy(h)
        c = api.arena_stats(0)
        assert c["in_use"] == base["in_use"] and c["mapped"] > 0              # pinned: the pieces stay
        made = c["pieces_created"]
        h = L_.pg_create_sized(0, 31, 0, 8, 24, 2, 70_000_000)
        assert h and api.arena_stats(0)["pieces_created"] == made              # ... and the next context is cut from them: nothing new from the driver
        L_.pg_destroy(h)
    if alone:
        assert api.arena_stats(0)["mapped"] == 0


def test_sharded_pass1_rccl_single_rank(golden, tmp_path):
    """The RCCL transport on the one GPU there is: a world of one rank -- librccl is loaded, a communicator is made from a
    unique id, the (self) exchange and the all-reduce run through it."""
    c = golden["cases"]["t8k_k63"]
    codes = case_codes(c)
    want, last_want, K = oracle_records(codes, c["K"], 4, prefix=str(tmp_path / "o"))
    out = _sharded_records(codes, K, 4, 1, rccl_world1=True)
    assert out[0][4] == "rccl"
    assert (_sorted(out[0][0], 2) == _sorted(want, 2)).all()
    assert (out[0][2] == last_want).all()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
@pytest.mark.parametrize("name", ["m60k_k63", "t6k_k127", "t5k_k24", "d8k_k127", "t6k_k31", "g120k_k63", "x500_k63"])   # (t6k_k31: -p 1 and 2, fewer sets than ranks; g120k_k63: trimmed reads)
def test_cli_sharded_matches_reference_files(golden, tmp_path, name, devices):
    """`pregraph` with pass 1 sharded over several ranks (SOAPDENOVO2_AMD_DEVICES, here all on GPU 0): the five files (and
    the -R pair) equal the reference's byte for byte, as with one rank."""
    from soapdenovo2_amd import synth
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        env = dict(PARALLEL_PARSE, SOAPDENOVO2_AMD_DEVICES=devices, PG_HOST_VERBOSE="1", SOAPDENOVO2_AMD_BATCH_READS="7000")
        log = _run_cli(cfg, c["K"], pre, P, D, a, m, R=True, extra_env=env)
        n_ranks = len(devices.split(','))
        assert f"pass 1 on {n_ranks} rank(s)" in log
        # after the regroup every rank holds the k-mers of its own sets and no more (set s -> rank s mod n_ranks): nothing is gathered
        import re
        held = [(int(m.group(1)), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"rank (\d+) \(device \d+\): \d+ distinct k-mers after pass 1, (\d+) of (\d+) after the regroup", log)]
        assert len(held) == n_ranks and sum(h[1] for h in held) == held[0][2] > 0
        share = -(-P // n_ranks) / P                                       # the largest number of sets a rank owns / all sets
        assert max(h[1] for h in held) <= share * held[0][2] * 1.25 + 64, held
        # the graph stages' COMPUTE is sharded too: every rank is a lane of the graph -- pass 2 deals its read batches to the lanes in
        # turn (each threads about 1 / n_ranks of the reads against the peer-mapped sets, own pre-arc table, merged at the end) and the
        # scans over a set's slots run on the lane that owns the set
        lanes = [(int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))) for m in
                 re.finditer(r"graph lane (\d+) \(device \d+\): pass 2 threaded (\d+) read\(s\) in (\d+) batch\(es\), \d+ distinct pre-arc\(s\); (\d+) per-set scan\(s\) ran here; (\d+) tip / edge walk", log)]
        assert [l[0] for l in lanes] == list(range(n_ranks)), log[-3000:]
        n_reads = sum(l[1] for l in lanes)
        assert n_reads == c["N"]
        if c["N"] >= n_ranks * 7000:                                       # (at least a batch of 7000 reads for every lane)
            assert all(l[2] >= 1 for l in lanes), lanes
        assert max(l[1] for l in lanes) <= n_reads / n_ranks + 2 * 7000, lanes
        owners = min(P, n_ranks)
        assert sum(1 for l in lanes if l[3] > 0) == owners, lanes              # a lane that owns a set scanned it; a lane that owns none scanned nothing
        # the WALKS of the tip and edge stages (a lane a candidate, crossing sets by nature) are dealt to the lanes in equal shares
        walks = [l[4] for l in lanes]
        assert min(walks) > 0 and max(walks) <= sum(walks) / n_ranks * 1.05 + 64, lanes
        # ... and pass 2's LOOKUPS go to the sets' owners (the routed form: keys out, node words back, the owner probes its own HBM)
        routed = [(int(m.group(1)), int(m.group(2)), int(m.group(3))) for m in
                  re.finditer(r"pass 2 routed, lane of device \d+: asked (\d+) lookup\(s\), (\d+) of them of other lanes, in \d+ round\(s\); answered (\d+) from its own sets; 0 probes of peer-mapped sets", log)]
        assert len(routed) == n_ranks, log[-3000:]
        assert sum(r[0] for r in routed) == sum(r[2] for r in routed) > 0
        assert sum(1 for r in routed if r[2] > 0) == owners, routed            # exactly the owners answered
        want = golden["md5"][t]
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc", "path", "markOnEdge"):
            assert md5_file(pre + "." + ext) == want[ext], (t, ext)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["1", "0"], ids=["routed", "peer-probes"])
@pytest.mark.parametrize("name", ["m60k_k63", "t6k_k127", "t5k_k24", "g40k_k127"])
def test_cli_sharded_pass2_lookups_routed_to_the_owners(golden, tmp_path, name, route):
    """Pass 2 of a sharded run without -R, so that the routed form runs in full rounds (a batch on every lane, all lanes cutting,
    answering and threading at once): every k-mer of every read of K + 1 bases and more is looked up exactly once, by the lane that owns
    its set; SOAPDENOVO2_AMD_P2_ROUTE=0 is the A/B form (every lane probes the peer-mapped sets itself).  Same files either way.
    (prlRead2path.c:159-248: every read one worker, every set one owner.)"""
    import re
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        env = dict(PARALLEL_PARSE, SOAPDENOVO2_AMD_DEVICES="0,0,0", PG_HOST_VERBOSE="1", SOAPDENOVO2_AMD_BATCH_READS="5000", SOAPDENOVO2_AMD_P2_ROUTE=route)
        log = _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env=env)
        if route == "1":
            routed = [(int(x.group(1)), int(x.group(2)), int(x.group(3)), int(x.group(4))) for x in
                      re.finditer(r"pass 2 routed, lane of device \d+: asked (\d+) lookup\(s\), (\d+) of them of other lanes, in (\d+) round\(s\); answered (\d+) from its own sets; 0 probes of peer-mapped sets", log)]
            assert len(routed) == 3, log[-3000:]
            assert sum(r[0] for r in routed) == sum(r[3] for r in routed) > 0
            Ke = c["K"] + (1 - c["K"] % 2)                                  # (pregraph.c:71-97: an even K becomes K + 1)
            if c.get("min_len"):                                           # trimmed reads: the sum over the reads of (len - K + 1)
                assert sum(r[0] for r in routed) == int((case_lens(c).astype(np.int64) - Ke + 1).sum()), routed
            else:                                                          # uniform reads: N x (L - K + 1) lookups in all
                assert sum(r[0] for r in routed) == c["N"] * (c["L"] - Ke + 1), routed
            assert "bytes a read crossed between lanes" in log
        else:
            assert "pass 2 direct: every lane probed the peer-mapped sets itself" in log and "pass 2 routed" not in log
        want = golden["md5"][t]
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
            assert md5_file(pre + "." + ext) == want[ext], (t, ext)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


@pytest.mark.parametrize("name,run_i", [("m60k_k63", 0), ("t6k_k127", 1)])
def test_cli_sharded_one_rank_unsuited_replays_on_the_host(golden, tmp_path, name, run_i):
    """One rank reports its sets unsuited for the device layout while the others have already laid theirs out INSIDE the record
    pools they took over (the regrouped records lie in the tail of the same allocation): the taken blocks go back on offer, none is
    freed, and the host replay reads every rank's records (graph_kernels.hip p2_layout_rank*, host_graph.cpp
    layout_on_ranks).  Growable sets and -a pools."""
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    run = c["runs"][run_i]
    P, D, a, m = run
    t = case_tag(name, run)
    pre = str(tmp_path / t)
    env = dict(PARALLEL_PARSE, SOAPDENOVO2_AMD_DEVICES="0,0,0", PG_HOST_VERBOSE="1", SOAPDENOVO2_AMD_BATCH_READS="7000", SOAPDENOVO2_AMD_TEST_UNSUITED_RANK="1")
    log = _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env=env)
    assert "replay set" in log                                             # the host replay ran
    want = golden["md5"][t]
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        assert md5_file(pre + "." + ext) == want[ext], (t, ext)
    assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


def test_cli_sharded_repeats_the_cut_when_an_owner_region_overflows(golden, tmp_path):
    """Send regions that start far too small (PG_ROUTE_CAP): every round's cut overflows, the ranks agree on a larger capacity and
    cut again -- the files do not change."""
    name = "m60k_k63"
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    P, D, a, m = c["runs"][0]
    t = case_tag(name, c["runs"][0])
    pre = str(tmp_path / t)
    env = dict(PARALLEL_PARSE, SOAPDENOVO2_AMD_DEVICES="0,0,0", PG_HOST_VERBOSE="1", SOAPDENOVO2_AMD_BATCH_READS="7000", PG_ROUTE_CAP="50")
    _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env=env)
    want = golden["md5"][t]
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        assert md5_file(pre + "." + ext) == want[ext], (t, ext)
    assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


def test_cli_sharded_reader_corner_cases(golden, tmp_path):
    """Ragged / truncated / mate-file inputs through the sharded pass 1 (the one-lane-per-read cutter with routing)."""
    from soapdenovo2_amd import synth
    for name in synth.QUIRK_CASES:
        cfg = synth.make_quirk_case(str(tmp_path), name)
        pre = str(tmp_path / (name + "_sh"))
        _run_cli(cfg, 31, pre, 3, 0, 0, 0, extra_env={"SOAPDENOVO2_AMD_DEVICES": "0,0,0", "SOAPDENOVO2_AMD_BATCH_READS": "300"})
        want = golden["md5"][name]
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
            assert md5_file(pre + "." + ext) == want[ext], (name, ext)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], name


def _threshold_reads(K=31, target=793):
    """Reads (all K + 1 bases) with exactly `target` distinct k-mers in one set (1031 slots * 0.77f -> max 793), then pure
    duplicates: the reference grows the set on the duplicate (newhash.c:477 tests before it probes)."""
    from oracle_binding import Oracle
    rng = np.random.default_rng(9)
    o = Oracle(K, P=1, max_read_len=K + 2)
    reads = []
    while o.L.oracle_node_count(o.h) + 2 <= target:
        r = rng.integers(0, 4, size=(1, K + 1), dtype=np.uint8)
        o.add_reads(r)
        reads.append(r[0])
    while o.L.oracle_node_count(o.h) < target:
        r = np.concatenate([reads[0][1:], rng.integers(0, 4, size=1, dtype=np.uint8)])[None, :]
        o.add_reads(r)
        reads.append(r[0])
    assert o.L.oracle_node_count(o.h) == target
    o.close()
    return reads


@pytest.mark.parametrize("tail", ["duplicate", "none"])
def test_cli_last_put_on_demand(tmp_path, tail):
    """The per-set last put is computed only when a set ends exactly at a growth threshold (pg_host_last_put_matters): with
    793 distinct k-mers in the single set a trailing duplicate read grows the reference's table and changes .vertex order;
    without it the table stays.  Both against the reference binary, with one rank and with three."""
    from soapdenovo2_amd import api, synth
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built")
    reads = _threshold_reads()
    if tail == "duplicate":
        reads = reads + [reads[0]]
    fa = str(tmp_path / "t.fa")
    synth.write_fasta(fa, np.stack(reads))
    cfg = str(tmp_path / "t.cfg")
    open(cfg, "w").write(f"max_rd_len=100\n[LIB]\navg_ins=200\nasm_flags=3\nf={fa}\n")
    outs = {}
    for tag, binary, env in (("amd", api.binary(False), {}), ("amd3", api.binary(False), {"SOAPDENOVO2_AMD_DEVICES": "0,0,0", "SOAPDENOVO2_AMD_BATCH_READS": "100"}),
                             ("ref", ref, {})):
        pre = str(tmp_path / tag)
        r = subprocess.run([binary, "pregraph", "-s", cfg, "-K", "31", "-o", pre, "-p", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        outs[tag] = pre
    for tag in ("amd", "amd3"):
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
            assert md5_file(outs[tag] + "." + ext) == md5_file(outs["ref"] + "." + ext), (tag, tail, ext)
        assert md5_gz_text(outs[tag] + ".edge.gz") == md5_gz_text(outs["ref"] + ".edge.gz"), (tag, tail)


def test_set_counts_and_last_put_calls(golden, tmp_path):
    import torch
    from soapdenovo2_amd import api
    c = golden["cases"]["t8k_k63"]
    codes = case_codes(c)
    P = 5
    want, last_want, K = oracle_records(codes, c["K"], P, prefix=str(tmp_path / "o"))
    kc = api.KmerCounter(K, n_sets=P, log2_slots=18, engine=2)
    packed = torch.from_numpy(api.pack_reads_uniform(codes).view(np.int64)).cuda()
    kc.count_uniform(packed, codes.shape[0], codes.shape[1], 0)
    kc.finalize(0, want_last_put=False)
    counts = kc.set_counts()
    assert (counts == np.bincount((want[:, 3] >> np.uint64(56)).astype(int), minlength=P)).all()
    assert (kc.last_put() == last_want).all()
    kc.close()
    assert api.lib().pg_host_last_put_matters(np.array([793], dtype=np.uint64).ctypes.data, 1, 0, 0) == 1
    assert api.lib().pg_host_last_put_matters(np.array([792], dtype=np.uint64).ctypes.data, 1, 0, 0) == 0
    assert api.lib().pg_host_last_put_matters(np.array([793], dtype=np.uint64).ctypes.data, 1, 1, 0) == 0


# ---- the drop-in, proven: the reference's own executable with its pregraph stage replaced by the library ------------------
@pytest.mark.parametrize("flavour,K", [("63", 31), ("127", 75)])
def test_linked_into_the_reference_all_pipeline(tmp_path, flavour, K):
    """oracle/_ref/SOAPdenovo-*mer-amd = the reference's objects minus pregraph.o prlHashReads.o cutTipPreGraph.o node2edge.o
    output_pregraph.o prlRead2path.o, linked against libsoapdenovo2_amd.so (oracle/Makefile.ref, target `linkin`).  Its
    `all` pipeline calls call_pregraph in-process (standardPregraph/main.c:341) and then runs the reference's unchanged
    contig / map / scaff stages on what the GPU wrote: every output must equal the all-reference run's."""
    from soapdenovo2_amd import synth
    ref = os.path.join(ROOT, "oracle", "_ref", f"SOAPdenovo-{flavour}mer")
    amd = ref + "-amd"
    if not (os.path.exists(ref) and os.path.exists(amd)):
        pytest.skip("reference / link-in binaries not built")
    cfg = synth.make_case(str(tmp_path), "all", 30000, 6000, 100, 0.005, 20260926)
    outs = {}
    for tag, binary in (("ref", ref), ("amd", amd)):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([binary, "all", "-s", cfg, "-K", str(K), "-o", str(d / "o"), "-p", "4", "-R"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                           text=True, cwd=str(d))
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        outs[tag] = str(d / "o")
    assert "HIP device" in r.stderr                                  # the pregraph stage of the second run was ours
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc", "path", "markOnEdge", "contig", "ContigIndex", "Arc", "updated.edge", "scafSeq", "scaf"):
        assert md5_file(outs["amd"] + "." + ext) == md5_file(outs["ref"] + "." + ext), ext
    assert md5_gz_text(outs["amd"] + ".edge.gz") == md5_gz_text(outs["ref"] + ".edge.gz")


def test_call_pregraph_twice_in_one_process(golden, tmp_path):
    """The boundary is a function, not a process (the reference's pipeline() calls it in-process and carries on): two calls in
    one process -- getopt state reset, no device or host state carried over -- then one of the 127-mer flavour."""
    code = r'''
import sys
sys.path.insert(0, sys.argv[1])
from soapdenovo2_amd import api
cfg, out = sys.argv[2], sys.argv[3]
for i, (m, K) in enumerate(((False, 31), (False, 31), (True, 31))):
    rc = api.call_pregraph(["-s", cfg, "-K", str(K), "-o", f"{out}/r{i}", "-p", "8", "-R"], mer127=m, in_process=True)
    assert rc == 0
print("three calls done")
'''
    c = golden["cases"]["t6k_k31"]
    cfg = case_config(c, str(tmp_path), "t6k_k31")
    r = subprocess.run([os.sys.executable, "-c", code, ROOT, cfg, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "three calls done" in r.stdout, r.stderr[-1500:]
    want = golden["md5"]["t6k_k31_p8_d0_a0_63"]
    for i in (0, 1):
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc", "path", "markOnEdge"):
            assert md5_file(str(tmp_path / f"r{i}.{ext}")) == want[ext], (i, ext)
        assert md5_gz_text(str(tmp_path / f"r{i}.edge.gz")) == want["edge"], i
    assert md5_file(str(tmp_path / "r2.kmerFreq")) == want["kmerFreq"]          # same counts through the four-word path


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["device", "host"])
@pytest.mark.parametrize("name", ["t8k_k63", "m60k_k63", "d8k_k127", "m100k_k31"])
def test_cli_layout_on_the_device_and_on_the_host(golden, tmp_path, name, layout):
    """The k-mer sets' layout is made on the device by default -- growable sets (-a 0) through the fixed point over the in-place
    rehashes' insertion times (dev_rehash.hpp), -a pools through one sweep (dev_graph.hpp); SOAPDENOVO2_AMD_LAYOUT=host keeps the
    sequential host replay.  Same files either way."""
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        env = dict(PARALLEL_PARSE, PG_HOST_VERBOSE="1")
        if layout == "host":
            env["SOAPDENOVO2_AMD_LAYOUT"] = "host"
        log = _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env=env)
        on_device = "k-mer set layout on the device" in log
        assert on_device == (layout == "device"), log[-3000:]
        if layout == "device" and not a:
            assert "growable sets on device" in log
        want = golden["md5"][t]
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
            assert md5_file(pre + "." + ext) == want[ext], (t, ext)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


@pytest.mark.gpu
@pytest.mark.parametrize("toggle", [{"SOAPDENOVO2_AMD_KEEP_ON_HOST": "1"}, {"SOAPDENOVO2_AMD_EDGE_FILE_INLINE": "1"},
                                    {"SOAPDENOVO2_AMD_PARSE_SIMD": "0", "SOAPDENOVO2_AMD_READER": "map"}, {"SOAPDENOVO2_AMD_EB_WAYPOINTS": "0"}],
                         ids=["reads-kept-on-host", "edge-file-inline", "scalar-mapped-reader", "no-waypoints"])
@pytest.mark.parametrize("name", ["m60k_k63", "t6k_k127"])
def test_cli_switches_do_not_change_the_files(golden, tmp_path, name, toggle):
    """The product's remaining either/or switches (csrc/env.hpp: user switches and test hooks) do not change the files: the reads of pass 1
    kept on the device against the host store, <o>.edge.gz written beside pass 2 against before it, the AVX2 record and the copied windows of
    the reader against the scalar record on the mapped file, chains walked through waypoints against node by node.  (The kernels' tuning
    knobs of rounds 2 - 4 -- share tables, workgroup shapes, block probes -- are not switches any more: the product build has one form of
    each kernel, and a -DPG_MEASURE build reads the geometry knobs that are left.)"""
    c = golden["cases"][name]
    cfg = case_config(c, str(tmp_path), name)
    for run in c["runs"]:
        P, D, a, m = run
        t = case_tag(name, run)
        pre = str(tmp_path / t)
        _run_cli(cfg, c["K"], pre, P, D, a, m, extra_env=dict(PARALLEL_PARSE, **toggle))
        want = golden["md5"][t]
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
            assert md5_file(pre + "." + ext) == want[ext], (t, ext, toggle)
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], (t, toggle)


def test_bench_two_ranks_share_the_gpu_over_gloo(tmp_path):
    """The N > 1 bench path end to end without an 8-GPU node: `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --share-gpu
    --comm gloo` -- two PROCESSES, one rank each, both on this GPU; the library's own pipelined pg_count_reads_sharded (cut, flags,
    records of round i travelling while batch i + 1 is cut, appends) over the host-staged transport (pg_comm_create_host + gloo
    all-to-all), every occurrence conserved across the ranks, and the line carries the exchange's time, volume and the owner rule."""
    import json, socket, sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--comm", "gloo", "--reads", "2000000", "--batch-reads", "500000", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["conservation"]["ok"], j["conservation"]
    assert j["conservation"]["kmer_occurrences_in"] == 2 * 2000000 * 88 == j["conservation"]["sum_of_coverage_histogram"]
    assert j["exchange_ms"] > 0 and j["bytes_sent_per_rank"] > 0
    ex = j["exchange"]
    assert ex["transport"] == "host" and ex["rounds_per_step"] == 4 and ex["host_waits_per_round"] <= 1.5, ex      # one host wait a round (two in the first)
    assert "minimizer partition mod 2" in j["config"]["parallelism"] and "reference set id mod 2" in j["config"]["parallelism"]


def test_suite_subset_on_poisoned_arena_blocks():
    """Every device allocation of the library is a recycled arena block that holds whatever its last user left (a fresh hipMalloc hands out zeroes in
    practice): PG_ARENA_POISON=1 fills every block with 0xA5 as it is cut, so code that counts on zeroes fails every time instead of once in a
    while.  The counting kernels, the executable on uniform and trimmed reads (growable sets and -a pools, -R), the sharded run and the
    device layouts, again, in a process of their own with the hook on."""
    import sys
    sel = ("(test_count_matches_oracle and (t6k_k127 or t8k_k63-2-False or t6k_k31-16)) or (test_cli_matches_reference_files and (t6k_k31-2 or t8k_k63-2 or g60k_k31-2 or g40k_k127-2))"
           " or (test_cli_sharded_matches_reference_files and t5k_k24) or test_cli_layout_on_the_device_and_on_the_host or test_ragged_tiles_match_oracle and g60k and exact"
           " or test_cli_reader_corner_cases")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_pregraph.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", "-k", sel],
                       capture_output=True, text=True, env=dict(os.environ, PG_ARENA_POISON="1"), cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 15, tail
