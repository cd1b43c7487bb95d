"""Average duration of ONE kernel over a slice of its launches in a rocprofv3 --kernel-trace rocpd database.
    python profiles/kernel_slice.py <results.db> <name-substring> <first-launch> <count>
bench.py launches the FM forward kernel in several phases (eager warm-up while the graphs are captured, the replayed warm-up
and timed steps, then eager launches that measure it as it runs in the step / warm / alone); `rocprofv3 --stats` averages all
of them.  This prints the average of the replayed launches only -- the number bench.py's `roofline.kernel_ms` (the kernel as
it runs in the step) has to agree with -- next to the overall one."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    name, first, count = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    d = [(r[2] - r[1]) / 1e3 for r in rows if name in r[0]]
    part = d[first:first + count]
    print("%s: %d launches, overall average %.2f us" % (name, len(d), sum(d) / max(len(d), 1)))
    print("launches %d..%d (the replayed warm-up and timed steps): average %.2f us, min %.2f, max %.2f" %
          (first, first + len(part) - 1, sum(part) / max(len(part), 1), min(part), max(part)))
    tail = d[first + count:]
    for k in range(0, len(tail), 10):
        seg = tail[k:k + 10]
        print("launches %d..%d (eager, after the timed loop): average %.2f us" %
              (first + count + k, first + count + k + len(seg) - 1, sum(seg) / len(seg)))


if __name__ == "__main__":
    main()
