"""Where the wavefronts of each kernel spend their cycles, from a rocprofv3 rocpd database collected with
    --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
(MI355X_MICROARCH.md: WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall (MFMA RAW / pipe busy),
ACTIVE_INST_ANY = issuing; the three are disjoint and sum to WAVE_CYCLES).
    python profiles/sq_stalls.py <results.db> [kernel-substring ...]"""
import sqlite3
import sys

WANT = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT",
        "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES"]


def main():
    db = sqlite3.connect(sys.argv[1])
    subs = sys.argv[2:]
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [t for t in tables if t.startswith("counters_collection")] or [t for t in tables if "pmc" in t.lower()]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % view[0])]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "name" in c and "kernel" not in c][0]
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else cols[0]
    q = "select %s, %s, count(distinct %s), sum(%s) from %s group by %s, %s" % (kcol, ccol, dcol, vcol, view[0], kcol, ccol)
    per = {}
    for name, counter, n, total in db.execute(q):
        per.setdefault(name, {})[counter] = (n, total / max(n, 1))
    print("%-56s %5s %8s %8s %8s %8s %8s %8s" % ("kernel", "calls", "parked", "issue-st", "issuing", "lds-st", "bankconf", "mfma/wave"))
    for name, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0))[1]):
        if subs and not any(s in name for s in subs):
            continue
        wc = c.get("SQ_WAVE_CYCLES", (0, 0.0))
        if wc[1] <= 0:
            continue
        g = lambda k: c.get(k, (0, 0.0))[1]                                                   # noqa: E731
        print("%-56s %5d %8.3f %8.3f %8.3f %8.3f %8.3f %8.3f" % (
            name.split("(")[0].replace("void ", "")[:56], wc[0], g("SQ_WAIT_ANY") / wc[1], g("SQ_WAIT_INST_ANY") / wc[1],
            g("SQ_ACTIVE_INST_ANY") / wc[1], g("SQ_WAIT_INST_LDS") / wc[1], g("SQ_LDS_BANK_CONFLICT") / wc[1],
            g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * wc[1])))


if __name__ == "__main__":
    main()
