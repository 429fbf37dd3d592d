#!/bin/bash
# Long-read path, one worker: wall clocks of the host and device stages (NP2_TIMING) of the second of two calls on a synthetic window.
# usage (GPU box, repo root): tests/tools/np2_timing.sh [contig Mb] [host threads]
cd "$(dirname "$0")/../.."
R=$PWD
MB=${1:-5}
TH=${2:-4}
python tests/tools/np2_make_case.py /tmp/np2case $MB 20 > /dev/null
cat > /tmp/np2case/run.py <<PY
import sys, time, os
sys.path.insert(0, "$R/tests")
import ref2_binding as rb
G = rb.bind("$R/nextpolish_amd/lib/nextpolish2.so")
for k in range(3):
    t0 = time.time(); c0 = os.times()
    rb.polish(G, "/tmp/np2case/g.fa", "/tmp/np2case/bam.fofn", read_type=1)
    c1 = os.times()
    sys.stderr.write("=== call %d: %.3f s wall, %.3f s CPU (user %.3f sys %.3f)\n" % (k, time.time() - t0, c1[0] + c1[1] - c0[0] - c0[1], c1[0] - c0[0], c1[1] - c0[1]))
PY
NP2_TIMING=1 NP_HOST_THREADS=$TH NP_IO_THREADS=$TH python /tmp/np2case/run.py 2>&1 >/dev/null | awk '/=== call 1/{f=1} f' | head -150
