#!/bin/bash
# Round 5, second GPU step: the two-class pseudo-seed kernel (parity: the long-read GPU tests, a fuzz slice with every device pseudo-seed checked against
# the host version; time: rocprofv3 kernel stats of the 5 Mb window), the two-rank tile test, smoke with the drop-in calls, the probe's new variants.
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/r5
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_np2.py -x -q -m gpu -p no:cacheprovider > $O/np2_tests.log 2>&1; echo "np2 tests rc=$? $(tail -1 $O/np2_tests.log)"
NP2_POA_CHECK=1 timeout 400 python tests/tools/np2_fuzz_gpu.py 300 316 > $O/np2_fuzz_poa.log 2>&1; echo "fuzz rc=$? $(tail -2 $O/np2_fuzz_poa.log | tr '\n' ' ')"
python tests/tools/np2_make_case.py /tmp/np2case 5 20 > /dev/null
cat > /tmp/np2case/run.py <<PY
import sys
sys.path.insert(0, "$R/tests")
import ref2_binding as rb
G = rb.bind("$R/nextpolish_amd/lib/nextpolish2.so")
for _ in range(2):
    rb.polish(G, "/tmp/np2case/g.fa", "/tmp/np2case/bam.fofn", read_type=1)
PY
( export NP_HOST_THREADS=8 NP_IO_THREADS=8; cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python /tmp/np2case/run.py > /dev/null 2>&1 )
python tools/rocprof_summary.py stats $O/ks/ks_results.db > $O/np2_kernel_stats_5mb.txt 2>&1; rm -rf $O/ks
head -14 $O/np2_kernel_stats_5mb.txt | cut -c1-120
NP2_TIMING=1 NP_HOST_THREADS=8 NP_IO_THREADS=8 timeout 120 python /tmp/np2case/run.py 2>&1 | grep -a "np2 poa" | tail -3
NP2_POA_BIG_ONLY=1 NP2_TIMING=1 NP_HOST_THREADS=8 timeout 120 python /tmp/np2case/run.py 2>&1 | grep -a -E "np2 poa|pseudo" | tail -3
timeout 600 python -m pytest tests/test_gpu_tiling.py -x -q -m gpu -p no:cacheprovider > $O/tiling_tests.log 2>&1; echo "tiling tests rc=$? $(tail -1 $O/tiling_tests.log)"
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -6 $O/smoke.log
for m in 32 34 64 66 98; do timeout 120 gpurun_in/r5_probe2 $m; done > $O/probe2_more.log 2>&1; grep -c "exit 0" $O/probe2_more.log; grep -a "KILLED\|fault" $O/probe2_more.log | head
