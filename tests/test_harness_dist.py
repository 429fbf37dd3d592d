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
    flat = [n for w in range(3) for bb in np1.shard_batches(b, 3, w) for n in bb]
    assert sorted(flat) == sorted(names)


WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from nextpolish_amd import nextpolish1 as np1
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
lens = {"c%%02d" %% i: 1000 + 137 * i for i in range(23)}
names = sorted(lens)
mine = [n for b in np1.shard_batches(np1.plan_batches(names, lens, 3000), world, rank) for n in b]
gathered = [None] * world
dist.all_gather_object(gathered, mine)
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
    shards = json.load(open(str(out)))
    assert len(shards) == 2 and shards[0] and shards[1]
    assert not set(shards[0]) & set(shards[1])
    assert sorted(shards[0] + shards[1]) == ["c%02d" % i for i in range(23)]
