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


def make_case(outdir: str, name: str, genome_len: int, n_reads: int, read_len: int, err: float,
              seed: int, fmt: str = "fastq") -> str:
    """Generate <outdir>/<name>.fq (or .fa) + <name>.cfg; return the config path."""
    os.makedirs(outdir, exist_ok=True)
    codes = reads_codes(genome_len, n_reads, read_len, err, seed)
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
