#!/bin/bash
# Final round-3 evidence after the upload changes: the default bench line and the one-lane rocprofv3 kernel stats of the same workload
# (the PMC passes and the long-read window's stats of r3_profiles.sh are unchanged by them).  Run on the GPU box from the repo root.
set -x
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/r3f
mkdir -p $O
python bench.py > $O/r3_c5_bench.json 2> $O/bench_c5.err
tail -c 600 $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python $R/bench.py --lanes 1 --no-pmc --no-lgs --no-phase --no-cpu-baseline --no-e2e --steps 2 --resident-passes 2 > $O/r3_c5_bench_one_lane.json 2>> $O/bench_c5.err
cd $R
python tools/rocprof_summary.py stats $O/ks/ks_results.db > $O/r3_c5_kernel_stats.txt
rm -rf $O/ks
ls -la $O
python -c "
import json
d=json.load(open('$O/r3_c5_bench.json'))
print('value', d['value'], 'resident', d['resident']['mbp_s'], 'frac', d['roofline']['frac'], 'h2d', d['config']['h2d_gb_per_s_rank0'], d['config'].get('h2d_bytes_per_draft_bp'), 'parity', d['parity']['identical'])
print('e2e', json.dumps(d.get('e2e_from_files'))[:400])
print('lgs', d['lgs'].get('value'), d['lgs']['roofline'].get('kernel'))
print('phase', d['snp_phase'].get('value'), json.dumps(d['snp_phase'].get('e2e_from_files'))[:300])
"
