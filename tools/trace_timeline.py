"""Timeline of one iteration of the device-resident adaptive loop from a rocprofv3 kernel_trace.csv: kernels in start order
with their durations and the idle gap before each (start - end of the previous kernel), averaged over the iterations found.

    python tools/trace_timeline.py kernel_trace.csv [anchor-substring = k_select]
"""
import collections
import csv
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("mbar::", "")
    return n[:44]


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_select"
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))]
    rows.sort()
    # split into iterations at the anchor kernel; keep iterations whose kernel sequence is the most common one
    its, cur = [], []
    prev_end = None
    for s, e, n in rows:
        cur.append((n, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end is not None else 0.0))
        prev_end = e
        if anchor in n:
            its.append(cur)
            cur = []
    seqs = collections.Counter(tuple(n for n, _, _ in it) for it in its)
    # (a bench run holds several loops -- config 3's, config 5's, ...: one block of tables per launch sequence, the most frequent four)
    for seq, cnt in seqs.most_common(4):
        if cnt < 3:
            continue
        report(its, seq)


def report(its, seq):
    # (iterations past convergence are no-ops of a few microseconds: keep those whose longest kernel is a real sweep)
    sel = [it for it in its if tuple(n for n, _, _ in it) == seq and max(d for _, d, _ in it) > 100.0]
    if not sel:
        sel = [it for it in its if tuple(n for n, _, _ in it) == seq]
    # iterations of one launch sequence can still be of different kinds (the last iteration of a solve runs its plain sweep and
    # leaves the fused one idle): one table per LONGEST kernel
    kinds = collections.OrderedDict()
    for it in sel:
        kinds.setdefault(max(it, key=lambda e: e[1])[0], []).append(it)
    for longest, group in kinds.items():
        print(f"{len(group)} iterations whose longest kernel is {longest} (of {len(its)} with any sequence); microseconds, mean over those iterations")
        tot_d = tot_g = 0.0
        for i, n in enumerate(seq):
            d = sum(it[i][1] for it in group) / len(group)
            g = sum(it[i][2] for it in group) / len(group)
            tot_d += d
            tot_g += g
            print(f"  gap {g:8.1f}   {n:44s} {d:9.1f}")
        print(f"  sum of gaps {tot_g:.1f} us, sum of kernels {tot_d:.1f} us, iteration {tot_d + tot_g:.1f} us")


if __name__ == "__main__":
    main()
