"""The long-read leg of bench.py alone (24 workers x 2 host threads on the one GPU, no roofline sub-run, no reference): the quick A/B of a
host-side change of the long-read feed.  usage: r4_lgs_quick.py [workers] [calls]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 24
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
os.environ.setdefault("NP2_WORKER_THREADS", "2")
r = bench.lgs_leg(1, 0, workers, 5.0, calls, False, False)      # (rank 1: the roofline sub-run belongs to rank 0)
if "error" not in r:
    r["mbp_s"] = round(r["bp"] / 1e6 / r["seconds"], 2)
print(json.dumps(r))
