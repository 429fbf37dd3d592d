#!/bin/bash
# Round 5, third GPU step: the pseudo-seed kernels after the reconvergence fix -- the probe, the long-read tests, the 5 Mb window's kernel stats
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/r5
mkdir -p $O
timeout 200 tests/tools/r5_poa_probe.sh 2>&1 | cut -c1-260
timeout 300 python -m pytest tests/test_gpu_np2.py -x -q -m gpu -p no:cacheprovider > $O/np2_tests.log 2>&1; echo "np2 tests rc=$? $(tail -1 $O/np2_tests.log)"
python tests/tools/np2_make_case.py /tmp/np2case 5 20 > /dev/null
cat > /tmp/np2case/run.py <<PY
import sys
sys.path.insert(0, "$R/tests")
import ref2_binding as rb
G = rb.bind("$R/nextpolish_amd/lib/nextpolish2.so")
for _ in range(2):
    rb.polish(G, "/tmp/np2case/g.fa", "/tmp/np2case/bam.fofn", read_type=1)
PY
NP2_POA_DEBUG=1 NP2_POA_CHECK=1 NP2_TIMING=1 NP_HOST_THREADS=8 NP_IO_THREADS=8 timeout 150 python /tmp/np2case/run.py 2>&1 | grep -a "np2 poa" | tail -4
( export NP_HOST_THREADS=8 NP_IO_THREADS=8; cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python /tmp/np2case/run.py > /dev/null 2>&1 )
python tools/rocprof_summary.py stats $O/ks/ks_results.db > $O/np2_kernel_stats_5mb.txt 2>&1; rm -rf $O/ks
head -16 $O/np2_kernel_stats_5mb.txt | cut -c1-120
awk 'NR>1 && $1 !~ /^#/ {s+=$(NF-2)} END {print "sum of kernel time (us, 2 windows):", s}' $O/np2_kernel_stats_5mb.txt
