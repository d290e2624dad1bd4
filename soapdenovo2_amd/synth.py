"""Deterministic synthetic read sets.

Uniform random genome over ACGT, fixed-length reads with uniform start, strand flipped with p=0.5,
independent substitution errors.  The same generator feeds the fixtures under tests/golden/, the
parity tests and (via :func:`reads_codes`) bench.py, so every number in the repository can be
regenerated from (genome_len, n_reads, read_len, err, seed).

Base codes follow the reference (standardPregraph/inc/def.h:39-42): A=0 C=1 T=2 G=3, complement = code ^ 2.
"""
from __future__ import annotations

import os
import numpy as np

_ASCII = np.frombuffer(b"ACTG", dtype=np.uint8)  # code -> letter


def reads_codes(genome_len: int, n_reads: int, read_len: int, err: float, seed: int) -> np.ndarray:
    """Return an (n_reads, read_len) uint8 array of base codes (A0 C1 T2 G3)."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    starts = rng.integers(0, genome_len - read_len, size=n_reads, dtype=np.int64)
    flip = rng.random(n_reads) < 0.5
    idx = starts[:, None] + np.arange(read_len, dtype=np.int64)[None, :]
    reads = genome[idx]
    # reverse complement of the flipped reads
    rc = (reads[:, ::-1] ^ 2).astype(np.uint8)
    reads = np.where(flip[:, None], rc, reads)
    if err > 0:
        mask = rng.random(reads.shape) < err
        shift = rng.integers(1, 4, size=reads.shape, dtype=np.uint8)
        reads = np.where(mask, (reads + shift) & 3, reads).astype(np.uint8)
    return np.ascontiguousarray(reads)


def genome_model(model: str, genome_len: int, seed: int, K: int):
    """Haplotypes to draw reads from.  "uniform": one random sequence.  "diploid": a second haplotype that differs by
    pairs of SNPs K + 2 apart (two branch nodes joined by a single (K+1)-mer: the length-1 edges of node2edge.c:481-542).
    "repeat": one haplotype with segments longer than K copied to other places (branching inside long shared stretches)."""
    rng = np.random.default_rng(seed + 7919)
    a = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    if model == "uniform":
        return [a]
    if model == "diploid":
        b = a.copy()
        step = max(8 * K, 1500)
        for p in range(2 * K, genome_len - 4 * K, step):
            for q in (p, p + K + 2):
                b[q] = (b[q] + 1 + rng.integers(0, 3)) & 3
        return [a, b]
    if model == "repeat":
        seg = 2 * K + 40
        for i in range(6):
            src = int(rng.integers(0, genome_len - seg))
            dst = int(rng.integers(0, genome_len - seg))
            a[dst:dst + seg] = a[src:src + seg]
            if i % 2:                                   # an inverted copy: the reverse strand of the same stretch
                a[dst:dst + seg] = (a[src:src + seg][::-1] ^ 2)
        return [a]
    raise ValueError(model)


def reads_codes_model(model: str, genome_len: int, n_reads: int, read_len: int, err: float, seed: int, K: int) -> np.ndarray:
    """reads_codes over the haplotypes of genome_model (each read picks a haplotype at random)."""
    if model == "uniform":
        return reads_codes(genome_len, n_reads, read_len, err, seed)
    haps = genome_model(model, genome_len, seed, K)
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, genome_len - read_len, size=n_reads, dtype=np.int64)
    which = rng.integers(0, len(haps), size=n_reads)
    flip = rng.random(n_reads) < 0.5
    idx = starts[:, None] + np.arange(read_len, dtype=np.int64)[None, :]
    reads = np.stack(haps)[which[:, None], idx]
    rc = (reads[:, ::-1] ^ 2).astype(np.uint8)
    reads = np.where(flip[:, None], rc, reads)
    if err > 0:
        mask = rng.random(reads.shape) < err
        shift = rng.integers(1, 4, size=reads.shape, dtype=np.uint8)
        reads = np.where(mask, (reads + shift) & 3, reads).astype(np.uint8)
    return np.ascontiguousarray(reads.astype(np.uint8))


def write_fastq(path: str, codes: np.ndarray, name_prefix: str = "r") -> None:
    """Write reads as single-end FASTQ (`@r<i>`, quality 'I' x L).

    The reference loses the last record when the file size is an exact multiple of 32768
    (standardPregraph/prlHashReads.c:873-877), so such a size is avoided by lengthening the last name.
    """
    n, L = codes.shape
    qual = b"I" * L
    chunks = []
    for i in range(n):
        chunks.append(b"@" + name_prefix.encode() + str(i).encode() + b"\n"
                      + _ASCII[codes[i]].tobytes() + b"\n+\n" + qual + b"\n")
    blob = b"".join(chunks)
    if len(blob) % 32768 == 0:
        last = chunks[-1]
        chunks[-1] = last.replace(b"\n", b"x\n", 1)
        blob = b"".join(chunks)
    with open(path, "wb") as f:
        f.write(blob)


def write_fastq_fast(path: str, codes: np.ndarray) -> None:
    """write_fastq for millions of reads: fixed-width names (`@r000000123`), one numpy block, no Python loop."""
    n, L = codes.shape
    names = np.char.zfill(np.arange(n).astype("U9"), 9).astype("S9").view(np.uint8).reshape(n, 9)
    rec = np.empty((n, 2 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:11] = names; rec[:, 11] = 10
    rec[:, 12:12 + L] = _ASCII[codes]
    rec[:, 12 + L] = 10; rec[:, 13 + L] = ord("+"); rec[:, 14 + L] = 10
    rec[:, 15 + L:15 + 2 * L] = ord("I"); rec[:, 15 + 2 * L] = 10
    blob = rec.tobytes()
    if len(blob) % 32768 == 0:                 # the reference drops the tail of such a file (prlHashReads.c:873-877): avoid the size
        blob = blob[:-1] + b" \n"
    with open(path, "wb") as f:
        f.write(blob)


def gpu_reads_codes(genome_len: int, n_reads: int, read_len: int, err: float, seed: int, device: int = 0, chunk: int = 2_000_000) -> np.ndarray:
    """The read model of reads_codes drawn with torch on a GPU (numpy takes minutes at 10 M reads); a different random
    stream, so the reads differ from reads_codes' for the same seed."""
    import torch
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev, generator=g)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        starts = torch.randint(0, genome_len - read_len, (n,), device=dev, generator=g, dtype=torch.int64)
        reads = genome[starts[:, None] + ar[None, :]]
        flip = torch.rand(n, device=dev, generator=g) < 0.5
        reads = torch.where(flip[:, None], torch.flip(reads, dims=[1]) ^ 2, reads)
        if err > 0:
            mask = torch.rand(reads.shape, device=dev, generator=g) < err
            shift = torch.randint(1, 4, reads.shape, device=dev, generator=g, dtype=torch.uint8)
            reads = torch.where(mask, (reads + shift) & 3, reads)
        out[lo:lo + n] = reads.cpu().numpy()
    return out


def write_fasta(path: str, codes: np.ndarray, name_prefix: str = "r") -> None:
    n, L = codes.shape
    chunks = [b">" + name_prefix.encode() + str(i).encode() + b"\n" + _ASCII[codes[i]].tobytes() + b"\n"
              for i in range(n)]
    blob = b"".join(chunks)
    if len(blob) % 32768 == 0:
        chunks[-1] = chunks[-1].replace(b"\n", b"x\n", 1)
        blob = b"".join(chunks)
    with open(path, "wb") as f:
        f.write(blob)


def write_config(path: str, fastq: str, max_rd_len: int, key: str = "q", avg_ins: int = 200) -> None:
    """Minimal SOAPdenovo2 library config (reference README.md:40-70)."""
    with open(path, "w") as f:
        f.write(f"max_rd_len={max_rd_len}\n[LIB]\navg_ins={avg_ins}\nreverse_seq=0\nasm_flags=3\n"
                f"rank=1\n{key}={os.path.abspath(fastq)}\n")


def ragged_lens(n_reads: int, min_len: int, read_len: int, seed: int) -> np.ndarray:
    """Read lengths of a trimmed library: uniform in [min_len, read_len] (its own random stream, so the bases are those of
    the untrimmed case)."""
    return np.random.default_rng(seed + 7919).integers(min_len, read_len + 1, size=n_reads).astype(np.int32)


def make_case(outdir: str, name: str, genome_len: int, n_reads: int, read_len: int, err: float,
              seed: int, fmt: str = "fastq", model: str = "uniform", K: int = 0, min_len: int = 0) -> str:
    """Generate <outdir>/<name>.fq (or .fa) + <name>.cfg; return the config path.  min_len: reads trimmed to ragged_lens."""
    os.makedirs(outdir, exist_ok=True)
    codes = reads_codes_model(model, genome_len, n_reads, read_len, err, seed, K)
    if min_len:
        lens = ragged_lens(n_reads, min_len, read_len, seed)
        data = os.path.join(outdir, name + ".fq")
        blob = b"".join(_fastq_blob([codes[i, :lens[i]] for i in range(n_reads)], [b"r%d" % i for i in range(n_reads)]))
        if len(blob) % 32768 == 0:                 # (the reference would drop the tail, prlHashReads.c:873-877)
            blob = blob[:-1] + b" \n"
        with open(data, "wb") as f:
            f.write(blob)
        cfg = os.path.join(outdir, name + ".cfg")
        write_config(cfg, data, read_len, "q")
        return cfg
    if fmt == "fastq":
        data = os.path.join(outdir, name + ".fq")
        write_fastq(data, codes)
        key = "q"
    else:
        data = os.path.join(outdir, name + ".fa")
        write_fasta(data, codes)
        key = "f"
    cfg = os.path.join(outdir, name + ".cfg")
    write_config(cfg, data, read_len, key)
    return cfg


# ---------------------------------------------------------------------------------------------------------
# Reader corner cases (SURVEY.md section 7 "bit-exactness quirks"): small inputs that exercise the reference's
# chunked reader and record parsers.  Deterministic; used by tests/golden/make_golden.py and the tests.
# ---------------------------------------------------------------------------------------------------------
def _fastq_blob(codes_list, names, lower_every=0, n_every=0, dot_every=0):
    chunks = []
    for i, (c, nm) in enumerate(zip(codes_list, names)):
        seq = bytearray(_ASCII[np.asarray(c, dtype=np.uint8)].tobytes())
        if n_every and i % n_every == 1 and len(seq) > 5:
            seq[3] = ord("N")
        if dot_every and i % dot_every == 2 and len(seq) > 7:
            seq[6] = ord(".")
        if lower_every and i % lower_every == 0:
            seq = bytearray(bytes(seq).lower())
        chunks.append(b"@" + nm + b"\n" + bytes(seq) + b"\n+\n" + b"I" * len(seq) + b"\n")
    return chunks


def make_quirk_case(outdir: str, name: str) -> str:
    """Write the input files + config of one reader corner case; returns the config path."""
    os.makedirs(outdir, exist_ok=True)
    p = lambda f: os.path.abspath(os.path.join(outdir, f))
    cfg = p(name + ".cfg")
    if name == "rq_32k":
        # file size an exact multiple of 32768: the reference parses its last full buffer twice and drops the tail
        codes = reads_codes(20000, 700, 100, 0.005, 101)
        names = [b"r%d" % i for i in range(700)]
        size = sum(len(n) for n in names) + 700 * (1 + 1 + 100 + 1 + 2 + 100 + 1)
        deficit = (-size) % 32768
        per, extra = divmod(deficit, 700)
        names = [n + b"x" * (per + (1 if i < extra else 0)) for i, n in enumerate(names)]
        blob = b"".join(_fastq_blob(list(codes), names))
        assert len(blob) % 32768 == 0
        open(p(name + ".fq"), "wb").write(blob)
        open(cfg, "w").write(f"max_rd_len=100\n[LIB]\navg_ins=200\nasm_flags=3\nq={p(name + '.fq')}\n")
    elif name == "rq_trunc":
        # two libs visited in avg_ins order (the second in the file comes first), max_rd_len cut, rd_len_cutoff,
        # reverse_seq, FASTA + FASTQ in one lib, lower case / N / '.' bases, a lib with asm_flags=2 (skipped)
        a = reads_codes(15000, 900, 100, 0.004, 102)
        b = reads_codes(15000, 600, 100, 0.004, 103)
        c = reads_codes(15000, 300, 100, 0.004, 104)
        open(p(name + "_a.fq"), "wb").write(b"".join(_fastq_blob(list(a), [b"a%d some text" % i for i in range(900)], lower_every=7, n_every=11, dot_every=13)))
        fa = b"".join(b">b%d\n" % i + _ASCII[b[i]].tobytes() + b"\n" for i in range(600))
        open(p(name + "_b.fa"), "wb").write(fa)
        open(p(name + "_c.fq"), "wb").write(b"".join(_fastq_blob(list(c), [b"c%d" % i for i in range(300)])))
        open(cfg, "w").write(
            f"max_rd_len=80\n[LIB]\navg_ins=500\nreverse_seq=1\nasm_flags=3\nrank=2\nq={p(name + '_a.fq')}\n"
            f"[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=1\nrd_len_cutoff=60\nq={p(name + '_c.fq')}\nf={p(name + '_b.fa')}\n"
            f"[LIB]\navg_ins=300\nasm_flags=2\nq={p(name + '_a.fq')}\n")
    elif name == "rq_ragged":
        # read lengths 20..120 (some shorter than K + 1 = 32 and dropped), records straddling 32 KiB chunk boundaries
        rng = np.random.default_rng(105)
        base = reads_codes(30000, 2500, 120, 0.004, 106)
        lens = rng.integers(20, 121, size=2500)
        open(p(name + ".fq"), "wb").write(b"".join(_fastq_blob([base[i, :lens[i]] for i in range(2500)], [b"read_%d/1" % i for i in range(2500)])))
        open(cfg, "w").write(f"max_rd_len=120\n[LIB]\navg_ins=200\nasm_flags=3\nq={p(name + '.fq')}\n")
    elif name == "rq_pair":
        # mate files: reads alternate file 1 / file 2 (q1/q2 and f1/f2 in one lib)
        a = reads_codes(20000, 800, 90, 0.004, 107)
        b = reads_codes(20000, 800, 90, 0.004, 108)
        c = reads_codes(20000, 400, 90, 0.004, 109)
        d = reads_codes(20000, 400, 90, 0.004, 110)
        open(p(name + "_1.fq"), "wb").write(b"".join(_fastq_blob(list(a), [b"p%d/1" % i for i in range(800)])))
        open(p(name + "_2.fq"), "wb").write(b"".join(_fastq_blob(list(b), [b"p%d/2" % i for i in range(800)])))
        open(p(name + "_1.fa"), "wb").write(b"".join(b">s%d/1\n" % i + _ASCII[c[i]].tobytes() + b"\n" for i in range(400)))
        open(p(name + "_2.fa"), "wb").write(b"".join(b">s%d/2\n" % i + _ASCII[d[i]].tobytes() + b"\n" for i in range(400)))
        open(cfg, "w").write(f"max_rd_len=90\n[LIB]\navg_ins=300\nreverse_seq=0\nasm_flags=3\n"
                             f"q1={p(name + '_1.fq')}\nq2={p(name + '_2.fq')}\nf1={p(name + '_1.fa')}\nf2={p(name + '_2.fa')}\n")
    elif name == "rq_gz":
        # gzip-compressed inputs: the reference reads them through popen("gzip -dc ...") (readseq1by1.c:676-714)
        import gzip
        a = reads_codes(20000, 1500, 100, 0.004, 111)
        b = reads_codes(20000, 700, 100, 0.004, 112)
        with gzip.GzipFile(p(name + ".fq.gz"), "wb", mtime=0) as f:
            f.write(b"".join(_fastq_blob(list(a), [b"z%d" % i for i in range(1500)])))
        with gzip.GzipFile(p(name + ".fa.gz"), "wb", mtime=0) as f:
            f.write(b"".join(b">g%d\n" % i + _ASCII[b[i]].tobytes() + b"\n" for i in range(700)))
        open(cfg, "w").write(f"max_rd_len=100\n[LIB]\navg_ins=200\nasm_flags=3\nq={p(name + '.fq.gz')}\nf={p(name + '.fa.gz')}\n")
    elif name == "rq_p":
        # p=: both mates of a pair in one FASTA file, one after the other (lib.c:130-506 type 3), next to a plain q= file
        a = reads_codes(20000, 1200, 90, 0.004, 113)
        b = reads_codes(20000, 500, 90, 0.004, 114)
        open(p(name + "_pairs.fa"), "wb").write(b"".join(b">m%d/%d\n" % (i // 2, 1 + i % 2) + _ASCII[a[i]].tobytes() + b"\n" for i in range(1200)))
        open(p(name + ".fq"), "wb").write(b"".join(_fastq_blob(list(b), [b"s%d" % i for i in range(500)])))
        open(cfg, "w").write(f"max_rd_len=90\n[LIB]\navg_ins=250\nreverse_seq=0\nasm_flags=3\np={p(name + '_pairs.fa')}\nq={p(name + '.fq')}\n")
    elif name == "rq_tie":
        # three libs, two of them with the same avg_ins: the reference orders libs with qsort (lib.c:505), whose order
        # among equals is the C library's business (glibc's merge sort keeps the file order); a third lib sorts in front
        a = reads_codes(20000, 700, 100, 0.004, 115)
        b = reads_codes(20000, 500, 100, 0.004, 116)
        c = reads_codes(20000, 300, 100, 0.004, 117)
        for tag, codes in (("a", a), ("b", b), ("c", c)):
            open(p(f"{name}_{tag}.fq"), "wb").write(b"".join(_fastq_blob(list(codes), [b"%s%d" % (tag.encode(), i) for i in range(len(codes))])))
        open(cfg, "w").write(
            f"max_rd_len=100\n[LIB]\navg_ins=300\nasm_flags=3\nrank=1\nq={p(name + '_a.fq')}\n"
            f"[LIB]\navg_ins=300\nasm_flags=3\nrank=2\nq={p(name + '_b.fq')}\n"
            f"[LIB]\navg_ins=200\nasm_flags=3\nrank=3\nq={p(name + '_c.fq')}\n")
    elif name == "rq_bam":
        # b=: unaligned BAM (readseq1by1.c:449-592).  Lib 1 (asm_flags=3): records pair up two by two and a pair with a
        # QC-fail mate (flag 0x200) is taken back; lib 2 (asm_flags=1): QC-fail records are skipped one by one.  IUPAC codes,
        # '=' and an empty sequence in between; reverse_seq on the second lib.
        a = reads_codes(20000, 1400, 100, 0.004, 118)
        b = reads_codes(20000, 602, 100, 0.004, 119)
        def records(codes, qc_every, weird_every):
            out = []
            for i, c in enumerate(codes):
                seq = _ASCII[c].tobytes().decode()
                if weird_every and i % weird_every == 3:
                    seq = seq[:10] + "N" + seq[11:20] + "M" + seq[21:30] + "=" + seq[31:]
                if weird_every and i % (weird_every * 7) == 5:
                    seq = ""
                flag = 0x4 | 0x1 | (0x40 if i % 2 == 0 else 0x80)
                if qc_every and (i % qc_every) in (2, 7):
                    flag |= 0x200
                out.append((b"pair%d/%d" % (i // 2, 1 + i % 2), flag, seq))
            return out
        write_bam(p(name + "_1.bam"), records(a, 37, 53))
        write_bam(p(name + "_2.bam"), records(b, 11, 0))
        open(cfg, "w").write(f"max_rd_len=90\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nb={p(name + '_1.bam')}\n"
                             f"[LIB]\navg_ins=300\nreverse_seq=1\nasm_flags=1\nb={p(name + '_2.bam')}\n")
    elif name == "rq_bam_odd":
        # two BAM files in one lib (asm_flags=3), the first with an ODD number of records: its last record is a first mate
        # without a second when the file ends, where the reference puts the pairing state back (readseq1by1.c:584-587), so the
        # second file pairs from its first record on -- and its second record is a QC-fail.  Also: runs of pairs taken back whose
        # first mates are too short to be kept (each removes one earlier kept read, prlHashReads.c:414-426), reads of mixed lengths.
        a = reads_codes(20000, 701, 100, 0.004, 120)
        b = reads_codes(20000, 904, 100, 0.004, 121)
        def records(codes, qc, short):
            out = []
            for i, c in enumerate(codes):
                seq = _ASCII[c].tobytes().decode()
                if i in short:
                    seq = seq[:20]                          # shorter than K + 1 = 32: not kept
                flag = 0x4 | 0x1 | (0x40 if i % 2 == 0 else 0x80)
                if i in qc:
                    flag |= 0x200
                out.append((b"pair%d/%d" % (i // 2, 1 + i % 2), flag, seq))
            return out
        qa = {9, 40, 41, 77, 300, 523}
        qb = {1, 101, 103, 105, 401, 640, 641, 642, 900}
        sb = {100, 102, 104, 400}                          # three pairs in a row: short first mate + QC-fail second mate
        write_bam(p(name + "_1.bam"), records(a, qa, set()))
        write_bam(p(name + "_2.bam"), records(b, qb, sb))
        open(cfg, "w").write(f"max_rd_len=100\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nb={p(name + '_1.bam')}\nb={p(name + '_2.bam')}\n")
    else:
        raise ValueError(name)
    return cfg


def write_bam(path: str, records) -> None:
    """A minimal unaligned BAM file: records = [(name bytes, flag, sequence str)], qualities 0xFF, no references.  BGZF blocks of
    <= 60000 bytes with the 'BC' extra field and the empty end-of-file block (SAM/BAM specification, sections 4.1, 4.2)."""
    import struct, zlib
    nt16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    text = b"@HD\tVN:1.0\tSO:unsorted\n"
    body = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0))
    for name, flag, seq in records:
        l = len(seq)
        packed = bytearray((l + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i // 2] |= nt16[ch] << (4 if i % 2 == 0 else 0)
        rec = struct.pack("<iiBBHHHiiii", -1, -1, len(name) + 1, 0, 4680, 0, flag, l, -1, -1, 0) + name + b"\0" + bytes(packed) + b"\xff" * l
        body += struct.pack("<i", len(rec)) + rec
    out = bytearray()
    def block(data):
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = comp.compress(bytes(data)) + comp.flush()
        bsize = len(raw) + 25
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + raw
                + struct.pack("<II", zlib.crc32(bytes(data)) & 0xFFFFFFFF, len(data)))
    for lo in range(0, len(body), 60000):
        out += block(body[lo:lo + 60000])
    out += block(b"")
    with open(path, "wb") as f:
        f.write(out)


QUIRK_CASES = ["rq_32k", "rq_trunc", "rq_ragged", "rq_pair", "rq_gz", "rq_p", "rq_tie", "rq_bam", "rq_bam_odd"]
