"""Per-kernel register / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python profiles/scripts/kres.py recbox_amd/csrc/rbx_fm_fused.hip [name-regex]"""
import re, subprocess, sys
src = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude",
                      "-Irecbox_amd/csrc", "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"],
                     stderr=subprocess.PIPE, text=True).stderr
for b in re.split(r"remark: Function Name: ", out)[1:]:
    name = subprocess.run(["c++filt", b.split()[0]], stdout=subprocess.PIPE, text=True).stdout.strip()
    if pat and not pat.search(name):
        continue
    g = lambda k: re.search(k + r": (\S+)", b).group(1)
    print("%-90s VGPR %3s AGPR %3s SGPR %3s scratch %3s LDS %6s occ %s" % (
        name.split("(")[0][-90:], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
