"""The executable memory plan (csrc/host_plan.cpp: pg_host_plan_memory -- the sizing functions the command and the partition engine allocate by,
cmd_plan.hpp / e2_plan.hpp, plus the later stages' blocks) against the device arena's MEASURED peaks of the six whole-command legs
(PG_ARENA_TRACE=1 on an MI355X, profiles/r06_arena_trace_*.txt, gpurun_out/r6k), and what it says about BASELINE.json's configs[3] and configs[4],
which have never met hardware.  No GPU."""
import re
import os

import pytest

from conftest import ROOT
from soapdenovo2_amd import api

GB = 1e9
# leg: (the trace kept under profiles/, the plan's arguments)
LEGS = {
    "200M_a40": dict(reads_total=200_000_000, read_len=150, K=63, a_gb=40),
    "60M_a0": dict(reads_total=60_000_000, read_len=150, K=63, a_gb=0),
    "60M_a16": dict(reads_total=60_000_000, read_len=150, K=63, a_gb=16),
    "20M_K127": dict(reads_total=20_000_000, read_len=150, K=127, a_gb=0),
    "20M_ragged": dict(reads_total=20_000_000, read_len=150, K=63, a_gb=16, fastq_bytes=5_320_155_450),     # (trimmed reads: the file is smaller than 20 M x 316 B)
    "10M_k31": dict(reads_total=10_000_000, read_len=100, K=31, a_gb=0),
}


def _measured(leg):
    t = open(os.path.join(ROOT, "profiles", f"r06_arena_trace_{leg}.txt")).read()
    peak = float(re.search(r"peak in use ([0-9.]+) GB", t).group(1))
    distinct = int(re.search(r"(\d+) node\(s\) allocated", t).group(1))
    pool = max(float(m.group(1)) for m in re.finditer(r"cut +([0-9.]+) GB", t.split("pinned batch buffers")[0] + t.split("pinned batch buffers")[1].split("\n")[1]))
    return peak, distinct, pool


@pytest.mark.parametrize("leg", sorted(LEGS))
def test_plan_reproduces_the_measured_peak(leg):
    """Within 5 % of the arena's peak in use, for -a pools and growable sets, both flavours, uniform and trimmed reads; the record pool itself
    (the first large block of the trace) to the megabyte -- it is the same function that allocated it."""
    peak, distinct, pool = _measured(leg)
    p = api.plan_memory(distinct_total=distinct, **LEGS[leg])
    assert abs(p["peak"] / GB - peak) <= 0.05 * peak, (leg, p["peak"] / GB, peak, p)
    assert abs(p["record_pool"] / GB - pool) < 0.01, (leg, p["record_pool"] / GB, pool)
    assert p["fits"] == 1 and p["peak"] == max(p[k] for k in ("stage1_pass1_count", "stage2_hand_over", "stage3_layout", "stage4_graph_pass2"))


def test_export_array_is_made_for_the_estimate_not_for_a_power_of_two():
    """Rounds 2 - 5: 0.7 x 2^32 records = 96 GB for configs[2]'s 1.15 G distinct k-mers (37 GB).  Round 6: one record per 8 occurrences of the
    estimate, cut back to the true count behind the counting pass."""
    p = api.plan_memory(distinct_total=1_146_737_909, **LEGS["200M_a40"])
    assert 70 * GB < p["export_allocated"] < 76 * GB and abs(p["export_after_count"] - 1_146_737_909 * 32) < 1e6
    assert not p["counts_twice"]
    # an estimate that is too small is said so: the K = 127 leg has one distinct k-mer per 3 occurrences, not per 8
    peak, distinct, _ = _measured("20M_K127")
    assert api.plan_memory(distinct_total=distinct, **LEGS["20M_K127"])["counts_twice"]


def test_a_rank_stores_the_partitions_it_owns_and_no_others():
    """configs[3]: 264 G occurrences cut into 2^25 partition ids; a rank of eight stores 2^22 of them (id mod 8 == rank, at id / 8): cursors,
    chunk table and the chunks at computed addresses follow that.  With every rank sized for all ids -- rounds 2 - 5 -- the spare chunk a
    partition alone was 2^24 x 6 KB = 103 GB a rank."""
    p = api.plan_memory(reads_total=3_000_000_000, read_len=150, distinct_total=18_800_000_000, K=63, n_sets=64, n_ranks=8)
    assert p["log2_partition_ids"] == 25 and p["log2_partitions_stored"] == 22
    assert p["tables"] < 2 * GB
    one = api.plan_memory(reads_total=375_000_000, read_len=150, distinct_total=2_350_000_000, K=63, n_sets=8, n_ranks=1)
    assert p["record_pool"] < 1.1 * one["record_pool"]                         # a rank's pool is that of a one-GPU job of its share


def test_pass_2s_pre_arc_table_is_planned_for_the_graphs_edges_and_bounded():
    """A lane's reads meet any edge of the graph, so its pre-arc table is made for ALL edge ids (cmd_plan.hpp: cmd_prearc_entries, 32 bytes an entry, eight
    entries an edge id) -- a rule that would ask a rank of configs[3] for 137 GB; it stops at an eighth of the device while two entries an edge remain."""
    big = api.plan_memory(reads_total=3_000_000_000, read_len=150, distinct_total=18_800_000_000, K=63, n_sets=64, n_ranks=8)
    sets_only = big["kmer_sets"]
    assert 30 * GB < big["stage4_graph_pass2"] - sets_only < 45 * GB            # the table (34 GB) + the edge lists
    small = api.plan_memory(reads_total=200_000_000, read_len=150, distinct_total=1_146_737_909, K=63, n_sets=8, a_gb=40, n_ranks=1)
    assert 8 * GB < small["stage4_graph_pass2"] - small["kmer_sets"] - small["reads_kept"] < 12 * GB   # 2^28 entries x 32 B = 8.6 GB + lists
    assert big["stage4_graph_pass2"] < big["peak"] and small["stage4_graph_pass2"] < small["peak"]


@pytest.mark.parametrize("K,distinct", [(63, 18_800_000_000), (127, 13_800_000_000)])
def test_configs_3_and_4_fit_a_288_GB_GPU_with_enough_sets(K, distinct):
    """BASELINE.json configs[3] / configs[4]: 3 G x 150 bp on 8 MI355X (distinct k-mers: the genome's 3 G + about 35 / 24 error k-mers an
    erroneous base at err 0.001).  The plan's verdict: pass 1 + count fits at any -p (the record pool and the export array are per rank); the
    LAYOUT stage needs its arrays for one whole set of the rank -- at -p 8 that is one set of 2.4 G keys a rank and does not fit beside the pool
    and the records, at -p 64 (eight sets a rank, what a 64-thread host would pass anyway) it does."""
    few = api.plan_memory(reads_total=3_000_000_000, read_len=150, distinct_total=distinct, K=K, n_sets=8, n_ranks=8)
    many = api.plan_memory(reads_total=3_000_000_000, read_len=150, distinct_total=distinct, K=K, n_sets=64, n_ranks=8)
    assert few["stage1_pass1_count"] == many["stage1_pass1_count"] < 0.97 * 288 * GB
    assert few["fits"] == 0 and few["peak_stage"] == 3
    assert many["fits"] == 1 and many["peak"] < 0.97 * 288 * GB, many
