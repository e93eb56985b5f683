"""Builds libvx355.so (HIP, gfx950 only) in-tree with hipcc.

`python -m velox_amd.build` or `build_lib()`. Objects go to velox_amd/csrc/build/,
the shared library to velox_amd/libvx355.so (git-ignored, travels with gpurun).
hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libvx355.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-ffp-contract=off"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "vx355.h"))
    return hs


def _digest(paths):
    """Content hash of the inputs of one object file (mtimes do not survive the copy to a GPU box)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in sorted(paths):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, deps):
    """True when 'target' is missing or was not built from the current contents of 'deps'
    (recorded next to it in target + '.stamp')."""
    stamp = target + ".stamp"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    try:
        return open(stamp).read().strip() != _digest(deps)
    except OSError:
        return True


def _write_stamp(target, deps):
    with open(target + ".stamp", "w") as f:
        f.write(_digest(deps))


def build_lib(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = _headers()
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
        for cmd in jobs:
            _write_stamp(cmd[-1], [cmd[-3]] + headers)
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lhiprtc", "-ldl"])
        _write_stamp(LIB, objs)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
