"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV (development aid).
usage: step_timeline.py kernel_trace.csv [anchor substring, default 'gather_kernel'] [step index from the end, default 2]
Steps are cut at the anchor kernel's launches (the first kernel of a step's forward).  Prints, for one step, every kernel
with its start offset, duration and queue, and the idle gaps of the union of all queues."""
import csv
import sys


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "gather_kernel"
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if anchor in r[2]]
    # a step has several anchor launches; group launches closer than 1.5 ms
    starts = []
    for i in cuts:
        if not starts or rows[i][0] - rows[starts[-1]][0] > 2_500_000:
            starts.append(i)
    if len(starts) < back + 1:
        print("not enough steps", len(starts))
        return
    lo, hi = starts[-back - 1], starts[-back]
    t0 = rows[lo][0]
    busy_end = t0
    print(f"step: {(rows[hi][0] - t0) / 1e6:.3f} ms, {hi - lo} kernels")
    for s, e, name, q, st in rows[lo:hi]:
        gap = s - busy_end
        flag = f"   <-- idle {gap / 1e3:.0f} us" if gap > 15_000 else ""
        short = name.split("(")[0][-70:]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{q} s{st}  {short}{flag}")
        busy_end = max(busy_end, e)


if __name__ == "__main__":
    main()
