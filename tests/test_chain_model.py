"""BASELINE config 5 as one run on the CPU (tests/tools/check_config5_chain.py with engine = "model"): the host models of both launch sequences chained in
the reference's `task = best` order, every step's contigs against the compiled reference's md5s (tests/golden/config5_chain_golden.json).  The GPU
form of the same chain is tests/test_gpu_chain.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "tools"))
import check_config5_chain as chain  # noqa: E402


def test_host_models_chained_like_task_best_equal_the_reference_at_every_step(tmp_path):
    r = chain.run("quick", workdir=str(tmp_path), engine="model")
    assert len(r["steps"]) == len(chain.STEPS) and r["identical"], r
