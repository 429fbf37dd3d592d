#!/bin/bash
# Long-read path, one worker, 5 Mb / 20x synthetic window polished twice: rocprofv3 kernel stats and the two PMC passes, per kernel.
# usage (GPU box, repo root): tests/tools/np2_prof.sh <out dir> [contig Mb]
set -x
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/${1:-gpurun_out/np2prof}
MB=${2:-5}
mkdir -p $O
python tests/tools/np2_make_case.py /tmp/np2case $MB 20
cat > /tmp/np2case/run.py <<PY
import sys
sys.path.insert(0, "$R/tests")
import ref2_binding as rb
G = rb.bind("$R/nextpolish_amd/lib/nextpolish2.so")
for _ in range(2):
    rb.polish(G, "/tmp/np2case/g.fa", "/tmp/np2case/bam.fofn", read_type=1)
PY
export NP_HOST_THREADS=8 NP_IO_THREADS=8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python /tmp/np2case/run.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- python /tmp/np2case/run.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- python /tmp/np2case/run.py > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py stats $O/ks/ks_results.db > $O/np2_kernel_stats.txt
python tools/rocprof_summary.py pmc $O/pf/pf_results.db > $O/np2_pmc_fetch.txt
python tools/rocprof_summary.py pmc $O/pw/pw_results.db > $O/np2_pmc_write.txt
rm -rf $O/ks $O/pf $O/pw
