cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/pf -o b -- python /root/repo/bench.py --config deepfm --steps 6 --warmup 2 --no-cpu-baseline > /tmp/log_f 2>&1)
python - <<'P' > gpurun_out/fills.txt 2>&1
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/pf/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
gc = [c for c in cols if "grid" in c.lower() or "workgroup" in c.lower()]
rows = db.execute("select name, start, end, %s from kernels order by start" % ", ".join(gc)).fetchall()
last = None
for i, r in enumerate(rows):
    if "FillFunctor" in r[0] and (r[2] - r[1]) > 20000:
        prev = rows[i - 1][0].split("(")[0][-50:] if i else ""
        nxt = rows[i + 1][0].split("(")[0][-50:] if i + 1 < len(rows) else ""
        print("%.1f us  grid %s   after %s   before %s" % ((r[2] - r[1]) / 1e3, r[3:], prev, nxt))
P
