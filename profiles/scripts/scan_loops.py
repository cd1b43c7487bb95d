"""Inner loops that wait for memory more than once per iteration, from the ISA of every kernel:
    python profiles/scripts/scan_loops.py            (compiles recbox_amd/csrc/*.hip to /tmp/asm/*.s first)
For each inner loop of at most 600 lines: the number of `s_waitcnt vmcnt(0)` and of global loads in it.  A streaming kernel
whose loop shows several full waits per iteration makes that many dependent trips to memory per row; this is how
LayerNorm's gamma / beta loads inside its loop over rows were found (round 3: forward 113 -> 76 us once they were read in front
of the loop).  Loops that issue all their loads first and then wait (several vmcnt(0) in a row behind the last load) are
false positives: read the listing."""
import glob
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.makedirs("/tmp/asm", exist_ok=True)
procs = []
for src in sorted(glob.glob(os.path.join(root, "recbox_amd", "csrc", "*.hip"))):
    out = "/tmp/asm/%s.s" % os.path.basename(src)[:-4]
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "recbox_amd", "csrc"),
                                   "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
found = []
for f in sorted(glob.glob("/tmp/asm/*.s")):
    cur, name = None, None
    for ln in open(f):
        m = re.match(r"^(_ZN3rbx\S+):", ln)
        if m:
            name = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", ln)
        if m:
            cur = [m.group(1), name, 0, 0, 0]
            found.append((cur, os.path.basename(f)))
            continue
        if cur is not None:
            cur[4] += 1
            if re.search(r"s_waitcnt.*vmcnt\(0\)", ln):
                cur[2] += 1
            if "global_load" in ln or "buffer_load" in ln:
                cur[3] += 1
            if re.search(r"s_cbranch\S+\s+" + re.escape(cur[0]) + r"\b", ln) or ln.startswith("_ZN") or "s_endpgm" in ln:
                cur = None
rows = sorted(((c[2], c[3], c[4], c[1], c[0], f) for c, f in found if c[2] >= 2 and c[4] < 600), reverse=True)
limit = int(sys.argv[1]) if len(sys.argv) > 1 else 40
names = {}
for w, g, n, k, lab, f in rows[:limit]:
    if k not in names:
        names[k] = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%2d vmcnt(0) %3d loads %4d lines  %-70s %s %s" % (w, g, n, names[k][-70:], f, lab))
