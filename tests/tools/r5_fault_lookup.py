"""Looks the address of the runtime's "Memory access fault by GPU ... on address 0x..." line up in the NP_ALLOCLOG files of the run
(np_devalloc.h): which device / pinned / registered range of which process held it, whether it was live or already released at the end of
the log, and what lay next to it.  usage: r5_fault_lookup.py <suite log> <alloc log>...   """
import re
import sys


def main():
    log = open(sys.argv[1], errors="replace").read()
    m = re.findall(r"Memory access fault by GPU[^\n]*on address (0x[0-9a-fA-F]+)[^\n]*", log)
    for line in re.findall(r"Memory access fault by GPU[^\n]*", log):
        print(line)
    if not m:
        print("no fault line in", sys.argv[1])
        return
    addr = int(m[-1], 16)
    for path in sys.argv[2:]:
        live, dead = {}, []
        n = 0
        last_t = "?"
        for ln in open(path, errors="replace"):
            f = ln.split()
            if len(f) != 5:
                continue
            t, tid, op, p, nbytes = f
            p, nbytes = int(p, 16), int(nbytes)
            n += 1
            last_t = t
            if op in ("D+", "H+", "R+"):
                live[p] = (t, tid, op, nbytes)
            elif op in ("D-", "H-", "R-"):
                if p in live:
                    a = live.pop(p)
                    dead.append((p, a, t, tid))
        hits = []
        for p, (t, tid, op, nb) in live.items():
            if p - (1 << 21) <= addr < p + nb + (1 << 21):
                hits.append(("LIVE", op, p, nb, t, tid, "", ""))
        for p, (t, tid, op, nb), t1, tid1 in dead[-4000:]:
            if p - (1 << 21) <= addr < p + nb + (1 << 21):
                hits.append(("RELEASED", op, p, nb, t, tid, t1, tid1))
        if hits or n > 200:
            print("%s: %d lines, %d live ranges at the end (t=%s)" % (path, n, len(live), last_t))
        for h in hits:
            st, op, p, nb, t, tid, t1, tid1 = h
            where = "INSIDE (+%d of %d)" % (addr - p, nb) if p <= addr < p + nb else ("%d bytes BEFORE its start" % (p - addr) if addr < p else "%d bytes PAST its end" % (addr - p - nb))
            print("   %-8s %s %#x + %d  allocated t=%s tid=%s%s : fault address is %s" % (st, op, p, nb, t, tid, (" released t=%s tid=%s" % (t1, tid1)) if t1 else "", where))


if __name__ == "__main__":
    main()
