"""Stage timing of ONE worker while P-1 other workers polish the same case on the same GPU: np2_multi_diag.py <dir> P"""
import os, sys, subprocess, time
here = os.path.dirname(os.path.abspath(__file__))
case, P = sys.argv[1], int(sys.argv[2])
bg = [subprocess.Popen([sys.executable, os.path.join(here, "np2_prof_case.py"), case, "1", "6"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(P - 1)]
time.sleep(3.0)
p = subprocess.run([sys.executable, os.path.join(here, "np2_stage_time.py"), case, "1", "8"], capture_output=True, text=True)
print(p.stdout[-3500:])
for b in bg:
    b.wait()
