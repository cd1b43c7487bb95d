"""Print the per-kernel summary of a rocprofv3 rocpd database (the `top_kernels` view):
    python profiles/topk.py gpurun_out/prof/bench_results.db [n_steps_for_per_step_column]
The optional us/step column divides a kernel's TOTAL time by the step count: it is only meaningful for kernels that run
once per step in every phase of the bench.  bench.py's post-timing pass (eager steps behind ~1 GiB ballast fills, there to
put HIP events around the dominant kernel) adds launches of its own, so the column is left EMPTY for the fill / copy
kernels of that pass ("FillFunctor 800 us per step"); use avg_us for everything else.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("%-72s %7s %12s %10s %6s%s" % ("kernel", "calls", "total_us", "avg_us", "%", "  us/step" if steps else ""))
    for name, calls, total, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 72:
            short = short[:69] + "..."
        ballast = ("FillFunctor" in name or "fillBuffer" in name or "copyBuffer" in name)
        extra = ("  %8s" % "-" if ballast else "  %8.1f" % (total / steps)) if steps else ""   # the view reports microseconds
        print("%-72s %7d %12.1f %10.2f %6.2f%s" % (short, calls, total, avg, pct, extra))


if __name__ == "__main__":
    main()
