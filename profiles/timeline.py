"""Timeline of ONE hipGraph replay out of a rocprofv3 --kernel-trace rocpd database (or its -f csv *_kernel_trace.csv): start offset, duration and queue of
every kernel between two consecutive launches of an anchor kernel (default: the first kernel of the step).
    python profiles/timeline.py <results.db> [anchor-substring] [which-occurrence] [steps]
(steps > 1: that many consecutive anchor intervals in one listing -- the boundaries between steps of one captured graph and
between two graphs.)
"""
import csv
import sqlite3
import sys


def main():
    anchor = sys.argv[2] if len(sys.argv) > 2 else "rezero_rows"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -5
    if sys.argv[1].endswith(".csv"):                   # rocprofv3 --kernel-trace -f csv: *_kernel_trace.csv
        qcol = "Queue_Id"
        with open(sys.argv[1]) as fh:
            rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""))
                           for r in csv.DictReader(fh)), key=lambda r: r[1])
    else:
        db = sqlite3.connect(sys.argv[1])
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
        rows = db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    a, b = marks[which], marks[which + steps]
    t0 = rows[a][1]
    print("# %s: %d kernels, %.1f us from the first start to the next step's first start" %
          ("one step" if steps == 1 else "%d steps" % steps, b - a, (rows[b][1] - t0) / 1e3))
    print("%-60s %9s %9s %9s  %s" % ("kernel", "start_us", "dur_us", "end_us", "queue"))
    for r in rows[a:b]:
        name = r[0].split("(")[0].replace("void ", "")[:60]
        print("%-60s %9.1f %9.1f %9.1f  %s" % (name, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3,
                                               r[3] if qcol else ""))


if __name__ == "__main__":
    main()
