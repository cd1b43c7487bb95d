"""Sum a PMC counter per kernel from a rocprofv3 rocpd database (one --pmc counter per run):
    python profiles/pmc.py <results.db> <COUNTER> [kernel-substring]
prints kernel, dispatches, average counter value per dispatch."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    want = sys.argv[2]
    sub = sys.argv[3] if len(sys.argv) > 3 else ""
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [t for t in tables if t.startswith("counters_collection")] or [t for t in tables if "pmc" in t.lower()]
    if not view:
        print("no counter table in", tables)
        return
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % view[0])]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "name" in c and "kernel" not in c][0]
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else cols[0]
    q = "select %s, count(distinct %s), sum(%s) from %s where %s = ? group by %s" % (kcol, dcol, vcol, view[0], ccol, kcol)
    for name, n, total in db.execute(q, (want,)):
        if sub in name:
            print("%-70s dispatches %5d   %s per dispatch %14.1f" % (name.split("(")[0][:70], n, want, total / n))


if __name__ == "__main__":
    main()
