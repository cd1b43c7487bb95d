"""MFMA utilisation per kernel from a rocprofv3 rocpd database collected with
    --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
(one pass, --kernel-trace only):   python profiles/mfma_util.py <results.db> [kernel-substring ...]
Prints, per kernel (averages per dispatch): the raw counters and two ratios --
  busy  = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES       (share of the cycles some SQ is busy in which an MFMA pipe is)
  pipe  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): MFMA-pipe busy cycles (64 per v_mfma_f32_32x32x2_f32)
          over all SIMD cycles of the launch -- GRBM_GUI_ACTIVE arrives summed over the 8 XCDs, 1024 = 256 CUs x 4 SIMDs.
          This is the MFMA utilisation; roofline.frac = pipe x (useful FLOPs / issued MFMA FLOPs);
  GFLOP = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 / 1e9: FLOPs the matrix pipes actually executed per dispatch (padding and
          masked tile halves included)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    subs = sys.argv[2:]
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
    q = "select %s, %s, count(distinct %s), sum(%s) from %s group by %s, %s" % (kcol, ccol, dcol, vcol, view[0], kcol, ccol)
    per = {}
    for name, counter, n, total in db.execute(q):
        per.setdefault(name, {})[counter] = (n, total / max(n, 1))
    print("%-64s %6s %14s %14s %14s %14s %8s %7s" % ("kernel", "calls", "MFMA_BUSY", "SQ_BUSY", "MFMA_MOPS_F32", "GUI_ACTIVE",
                                                        "GFLOP", "pipe"))
    for name, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]):
        if subs and not any(s in name for s in subs):
            continue
        mb = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))
        if mb[1] <= 0:
            continue
        sb = c.get("SQ_BUSY_CYCLES", (0, 0.0))[1]
        mo = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", (0, 0.0))[1]
        ga = c.get("GRBM_GUI_ACTIVE", (0, 0.0))[1]
        print("%-64s %6d %14.0f %14.0f %14.0f %14.0f %8.1f %7.3f" % (
            name.split("(")[0].replace("void ", "")[:64], mb[0], mb[1], sb, mo, ga, mo * 512 / 1e9,
            mb[1] / (ga / 8.0 * 1024.0) if ga else 0.0))


if __name__ == "__main__":
    main()
