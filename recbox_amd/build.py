"""Build librecbox_hip.so in-tree for gfx950 with hipcc (no cmake, no JIT cache).

    python -m recbox_amd.build [--force] [--jobs N]

Every ``csrc/*.hip`` is compiled to an object (in parallel) and linked into
``recbox_amd/lib/librecbox_hip.so``.  Objects are rebuilt when the source, a
header, or the flags are newer.  hipcc cross-compiles without a GPU, so this is
also the CPU-side "does it build" check run by ``__graft_entry__.build()``.
"""
import argparse
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librecbox_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; librecbox_hip.so cannot be built")


def _stamp(deps):
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for d in sorted(deps):
        h.update(d.encode())
        h.update(str(os.path.getmtime(d)).encode())
    return h.hexdigest()


def _compile(hipcc, src, headers, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp([src] + headers)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file):
        with open(stamp_file) as fh:
            if fh.read() == stamp:
                return obj, False
    cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, proc.stdout))
    if proc.stdout.strip():
        sys.stderr.write(proc.stdout)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(force=False, jobs=None, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    if not sources:
        raise RuntimeError("no HIP sources under %s" % CSRC)
    jobs = jobs or min(len(sources), os.cpu_count() or 4)
    objs, rebuilt = [], False
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        for obj, did in pool.map(lambda s: _compile(hipcc, s, headers, force), sources):
            objs.append(obj)
            rebuilt = rebuilt or did
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise RuntimeError("link failed:\n%s" % proc.stdout)
        if verbose:
            print("built %s (%d objects)" % (os.path.relpath(LIB, ROOT), len(objs)))
    elif verbose:
        print("%s is up to date" % os.path.relpath(LIB, ROOT))
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs)
