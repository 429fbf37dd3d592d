#!/usr/bin/env python
"""Adversarial fuzz of task 3 (snp_phase): tests/snpphase_gen.adversarial_case (odd CIGAR shapes and letters, thin patchy coverage) through
the compiled reference, the oracle and the host model.  usage: np1_phase_fuzz.py FIRST LAST   (needs oracle/_ref; CPU only)
Prints one line per disagreement and a summary: bad_or = oracle != reference where both answer, bad_mo = model != oracle,
refcrash = the reference crashed or hung (> 30 s), refcrash_def = it did so on an input the oracle calls defined."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_binding as mb  # noqa: E402
import oracle_binding as ob  # noqa: E402
import snpphase_gen  # noqa: E402
from conftest import parse_cli_fasta, ref_binary  # noqa: E402
from nextpolish_amd import _native as nat  # noqa: E402

a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp(prefix="np1phasefz_")
st = dict(tot=0, bad_or=0, bad_mo=0, refcrash=0, und=0, refcrash_def=0)
for seed in range(a, b):
    ctgs, sr, lr = snpphase_gen.adversarial_case(seed)
    s, l = nat.Stream.from_reads(ctgs, sr), nat.Stream.from_reads(ctgs, lr)
    fa, sb, lb = d + "/s.fa", d + "/sr.bam", d + "/lr.bam"
    s.write_files(fa, sb)
    l.write_files(d + "/l.fa", lb)
    try:
        p = subprocess.run([ref_binary(), "snpphase", fa, sb, lb], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=30)
        ref = parse_cli_fasta(p.stdout.decode()) if p.returncode == 0 else None
    except subprocess.TimeoutExpired:
        ref = None
    cfgp = nat.lib().config_init(fa.encode(), sb.encode(), lb.encode())
    ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    s2, l2 = nat.Stream.load(fa, sb, with_qual=True), nat.Stream.load(fa, lb, with_qual=True)
    ors = [ob.snp_phase(s2, l2, i, ocfg) for i in range(len(ctgs))]
    try:
        ms = mb.snp_phase(s2, l2, cfgp.contents)
    except ValueError:
        ms = None
    nat.lib().config_destory(cfgp)
    st["tot"] += 1
    if ref is None:
        st["refcrash"] += 1
        if all(o is not None for o in ors):
            st["refcrash_def"] += 1
            print("reference crashed, oracle defined:", seed)
        continue
    und = any(o is None for o in ors)
    st["und"] += und
    for i, (n, _) in enumerate(ctgs):
        if ors[i] is not None and ors[i] != ref.get(n):
            st["bad_or"] += 1
            print("oracle != reference:", seed, n)
    if (ms is not None) if und else (ms != ors):
        st["bad_mo"] += 1
        print("model != oracle:", seed)
print(st)
