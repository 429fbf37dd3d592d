#!/bin/bash
# round 6: A/B of the lane decoders after a change (tests/tools/r6_inflate_ab.py), the ingest tests through the LDS decoder, and the from-files leg of a 100 Mb bench
tag=${1:-x}
mkdir -p gpurun_out/r6
{
timeout 600 python3 tests/tools/r6_inflate_ab.py 2 16 lanes,lds85,lds75,lds65
timeout 600 python3 tests/tools/r6_inflate_ab.py 1 16 lanes,lds85,lds75,lds65
timeout 900 python3 tests/tools/r6_inflate_ab.py 2 64 lanes,lds85,lds75,lds65
NP1_INFLATE=lds85 timeout 900 python3 -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -5
for m in lanes lds85 lds65; do
  echo "== bench c3_100mb_30x, NP1_INFLATE=$m"
  NP1_INFLATE=$m timeout 900 python3 bench.py --workload c3_100mb_30x --no-lgs --no-phase --no-pmc --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); e=d.get('e2e_from_files',{})
        print(json.dumps({k:e.get(k) for k in ('mbp_s','warm_mbp_s','parity','roofline')}))
"
done
} > gpurun_out/r6/inflate_step_$tag.txt 2>&1
tail -60 gpurun_out/r6/inflate_step_$tag.txt | cut -c1-400
