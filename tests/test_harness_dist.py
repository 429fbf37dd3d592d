"""Host-side mirror of the reference caller (nextpolish_amd/nextpolish1.py) and the multi-GPU sharding,
the latter through a world_size-2 gloo group on CPU (no GPU compute: only who-polishes-what)."""
import os
import subprocess
import sys

import pytest

from nextpolish_amd import nextpolish1 as np1
from conftest import ROOT


def test_output_naming_rule():
    # reference: source/lib/nextpolish1.py:228
    assert np1.output_name("ctg1", 1) == "ctg1_np1"
    assert np1.output_name("ctg1_np1", 2) == "ctg1_np12"
    assert np1.output_name("a_b", 1) == "a_b_np1"
    assert np1.output_name("x_np12", 1) == "x_np121"


def test_parse_num_unit():
    assert np1.parse_num_unit("150k") == 150000 and np1.parse_num_unit("2m") == 2000000 and np1.parse_num_unit(17) == 17


def test_block_file_and_resume(tmp_path):
    blc = tmp_path / "g.blc"
    blc.write_text("ctgA_np1 0\nctgB_np1 1\nctgC_np1 0\nctgD_np1 0\n")
    out = tmp_path / "part.fa"
    # a finished record, then one cut off in the middle of its sequence line
    out.write_text(">ctgA_np12 8\nACGTACGT\n>ctgC_np12 8\nACG")
    done = set()
    pos = np1.read_polished_seqs(str(out), done)
    assert done == {"ctgA"} and pos == len(">ctgA_np12 8\nACGTACGT\n")
    todo = np1.read_unpolished_seqs(str(blc), "0", done)
    assert todo == ["ctgC_np1", "ctgD_np1"]
    fa = tmp_path / "g.fa"
    fa.write_text(">c1 x\nAC\n>c2\nGG\n")
    assert np1.read_unpolished_seqs(str(fa), "all", set()) == ["c1", "c2"]
    assert np1.fasta_lengths(str(fa)) == {"c1": 2, "c2": 2}


def test_batch_planning_partitions_contigs():
    lens = {"a": 400, "b": 300, "c": 500, "d": 50, "e": 2000, "f": 10}
    names = list(lens)
    b = np1.plan_batches(names, lens, 800)
    assert b == [["a", "b"], ["c", "d"], ["e"], ["f"]]


def test_deal_is_longest_first_and_resume_stable():
    """the deal must be a function of the block's full list, not of what a rank still has to do."""
    lens = {"A": 100, "B": 90, "C": 80, "D": 10, "E": 1000}
    names = ["A", "B", "C", "D", "E"]
    owner = np1.deal_contigs(names, lens, 2)
    assert owner["E"] == 0 and {owner[n] for n in "ABCD"} == {1}          # one giant contig does not drag others with it
    # rank 0 finished A-equivalent work and restarts: its share is its old share minus what is in its own output
    lens = {n: 100 for n in "ABCD"}
    full = np1.rank_share(list("ABCD"), lens, 2, 0, set(), True)
    again = np1.rank_share(list("ABCD"), lens, 2, 0, {full[0]}, True)
    assert again == full[1:]
    other = np1.rank_share(list("ABCD"), lens, 2, 1, set(), True)
    assert sorted(full + other) == list("ABCD") and not set(full) & set(other)
    loads = [sum(lens[n] for n in sh) for sh in (full, other)]
    assert max(loads) - min(loads) <= 100


WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from nextpolish_amd import nextpolish1 as np1
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
lens = {"c%%02d" %% i: 1000 + 137 * i for i in range(23)}
names = sorted(lens)
lens["c07"] = 40000          # one long contig: longest-first keeps the ranks within one contig of each other
mine = [n for b in np1.plan_batches(np1.rank_share(names, lens, world, rank, set(), True), lens, 3000) for n in b]
gathered = [None] * world
dist.all_gather_object(gathered, (mine, sum(lens[n] for n in mine)))
if rank == 0:
    json.dump(gathered, open(sys.argv[1], "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % {"root": ROOT})
    out = tmp_path / "o.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(out)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    import json
    got = json.load(open(str(out)))
    shards, loads = [g[0] for g in got], [g[1] for g in got]
    assert abs(loads[0] - loads[1]) <= 40000 and min(loads) > 0.4 * max(loads)
    assert len(shards) == 2 and shards[0] and shards[1]
    assert not set(shards[0]) & set(shards[1])
    assert sorted(shards[0] + shards[1]) == ["c%02d" % i for i in range(23)]


def _fasta_records(path):
    recs = {}
    lines = open(path).read().strip().split("\n")
    for k in range(0, len(lines), 2):
        recs[lines[k].split()[0][1:]] = lines[k + 1]
    return recs


@pytest.mark.gpu
@pytest.mark.parametrize("task", [1, 2])
def test_gpu_two_ranks_cat_of_parts_equals_one_rank(task, tmp_path):
    """`--world 2 --rank r` of the short-read caller on ONE GPU (both ranks on device 0, the way a node runs one rank per GPU): the
    concatenation of the two parts -- what the workflow's `cat` makes of them (source/nextPolish:231-234) -- holds exactly the records
    of a one-rank run (the resume side of the deal is covered on the CPU: test_deal_is_longest_first_and_resume_stable)."""
    from nextpolish_amd import _native as nat
    st = nat.Stream.synth([40000, 9000, 30000, 45000, 2000, 70000, 12000], depth=30, seed=4242, with_qual=1, draft_lower=0.01)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    exe = [sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py"), "-g", fa, "-s", bam, "-t", str(task), "--batch_bp", "60000"]
    one = str(tmp_path / "one.fa")
    p = subprocess.run(exe + ["-o", one], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    parts = [str(tmp_path / ("part%d.fa" % r)) for r in range(2)]
    procs = [subprocess.Popen(exe + ["-o", parts[r], "--world", "2", "--rank", str(r), "--device", "0"], stderr=subprocess.PIPE, text=True) for r in range(2)]
    for q in procs:
        assert q.wait(timeout=600) == 0, q.stderr.read()
    merged = {}
    for part in parts:
        recs = _fasta_records(part)
        assert recs and not set(recs) & set(merged)
        merged.update(recs)
    assert merged == _fasta_records(one)


@pytest.mark.gpu
def test_gpu_two_ranks_of_the_long_read_caller(tmp_path):
    """nextpolish2.py --world 2 --rank r on one GPU: cat of the parts == the one-rank output (records in completion order)"""
    import np2_cases
    cid, kw, rt = np2_cases.CASES[0]
    kw = dict(kw, contig_lens=(9000, 5000, 7000, 3000))
    fa, fofn, _contigs = np2_cases.materialise(kw, str(tmp_path))
    exe = [sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish2.py"), "-g", fa, "-l", fofn, "-r", {1: "ont", 2: "clr", 3: "hifi"}[rt], "-p", "1", "-sp"]
    one = str(tmp_path / "one.fa")
    p = subprocess.run(exe + ["-o", one], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    parts = [str(tmp_path / ("part%d.fa" % r)) for r in range(2)]
    procs = [subprocess.Popen(exe + ["-o", parts[r], "--world", "2", "--rank", str(r)], env=dict(os.environ, NP2_DEVICE="0"), stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    for q in procs:
        assert q.wait(timeout=600) == 0, q.stderr.read()
    merged = {}
    for part in parts:
        recs = _fasta_records(part)
        assert not set(recs) & set(merged)
        merged.update(recs)
    assert merged == _fasta_records(one) and len(merged) == 4
