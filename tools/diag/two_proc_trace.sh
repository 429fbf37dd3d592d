#!/bin/bash
# two concurrent workers under rocprofv3 --kernel-trace; prints the longest kernels and the longest gaps of each
export TMPDIR=/tmp
CASE=$1
for i in 1 2; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$i -o t -- python tools/np2_prof_case.py $CASE 1 3 > /tmp/tr$i.log 2>&1 &
done
wait
cat /tmp/tr1.log | grep "per call"; cat /tmp/tr2.log | grep "per call"
python - <<'PY'
import csv, glob
for i in (1, 2):
    f = glob.glob('/tmp/tr%d/**/*kernel_trace.csv' % i, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in rows)
    t0 = ev[0][0]
    print("== process %d: %d kernels over %.2f s" % (i, len(ev), (ev[-1][1] - t0) / 1e9))
    for s, e, n in sorted(ev, key=lambda x: x[0] - x[1])[:6]:
        print("   long kernel %8.1f ms at t=%.3f s  %s" % ((e - s) / 1e6, (s - t0) / 1e9, n))
    gaps = sorted(((ev[k + 1][0] - ev[k][1]), ev[k][1], ev[k][2], ev[k + 1][2]) for k in range(len(ev) - 1))[-6:]
    for g, at, a, b in reversed(gaps):
        print("   gap %8.1f ms after t=%.3f s  %s -> %s" % (g / 1e6, (at - t0) / 1e9, a, b))
PY
