"""What a previous, possibly interrupted run of one of the callers left in its output file.

Both reference callers restart by re-reading their own output (source/lib/nextpolish1.py:163-179 for the short-read tasks,
source/lib/nextpolish2.py:116-137 for the long-read one): every contig whose records are complete is skipped, the contig that was being
written when the run ended is dropped and written again from the offset of its first record.  The two differ only in how a FASTA header
maps to a contig -- "<name>_np<..>" is one record per contig, "<name>_s<k>" is piece k of a contig that may have several -- so there is one
walker here and two header rules.
"""


def scan_output(path, header_rule):
    """Walks the FASTA `path`.  header_rule(token) -> (contig name, True when this record is the FIRST of its contig); token is the header's
    first word without '>'.  Returns (names seen, name of the contig the file ends in or None, byte offset at which that contig starts --
    the point to truncate the file at)."""
    seen = set()
    tail_name, tail_at, at = None, 0, 0
    with open(path, "rb") as f:
        for raw in f:
            if raw[:1] == b">":
                words = raw.split(None, 1)
                tail_name, opens = header_rule(words[0][1:].decode() if words else "")
                if opens:
                    tail_at = at
                seen.add(tail_name)
            at += len(raw)
    return seen, tail_name, tail_at


def polished_record(token):
    """nextpolish1 output: '<contig>_np<round><task> <length>' -- one record per contig."""
    return token.split("_np", 1)[0], True


def corrected_piece(token):
    """nextpolish2 output: '<contig>' or '<contig>_s<k> ...' -- piece 0 (or an unsplit contig) opens the contig."""
    name, mark, rest = token.partition("_s")
    return name, (not mark) or rest.split("_s", 1)[0] == "0"


def finished_contigs(path, header_rule, done):
    """Adds the contigs `path` holds to the set `done`, except the one the file ends in (it may be cut off); returns the truncation offset."""
    seen, tail_name, tail_at = scan_output(path, header_rule)
    done |= seen
    if tail_name:      # (a header that is just ">" names nothing: like the reference, nothing is taken back for it)
        done.discard(tail_name)
    return tail_at


def fit_workers(window, workers, available_bytes, cpus):
    """The reference's -a adjustment of the long-read caller (nextpolish2.py:67-79) as a function: a window costs the host about 1536 bytes
    per base, so `budget` bases fit in the memory that is free; workers are capped by the logical CPUs, a window below 5 Mb (or a
    workers x window product beyond the budget) falls back to 5 Mb, and the workers are cut to what the budget then holds.
    Returns (window, workers)."""
    budget = available_bytes / 1536.0
    workers = min(workers, cpus)
    if window < 5000000 or workers * window > budget:
        window = 5000000
    return window, min(workers, int(budget / window))
