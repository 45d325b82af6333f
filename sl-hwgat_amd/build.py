"""Build recipe for libhwgat_hip.so (gfx950 only, in-tree, no JIT cache).

`python sl-hwgat_amd/build.py` or `build()` from `__graft_entry__`.  hipcc
cross-compiles without a GPU; the built .so travels to the GPU box with the
repo snapshot (it is git-ignored but not gpurun-ignored).

Every csrc/*.hip is compiled to its own object (in parallel, cached by the digest of
that file + the shared headers + the flags under sl-hwgat_amd/build/) and the objects
are linked into the one shared library; `libhwgat_hip.so.stamp` carries the digest of
ALL sources, which `_lib.lib()` checks before loading.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libhwgat_hip.so")
STAMP = LIB + ".stamp"
HEADER = os.path.join(os.path.dirname(HERE), "include", "hwgat_hip.h")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]
if os.environ.get("HWGAT_LAB") == "1":       # kernel-lab build: compiles the A/B environment switches in (common.h)
    FLAGS.append("-DHWGAT_LAB")
# per-source extra flags (none at present)
EXTRA = {}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [HEADER]


def _hash(paths):
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(EXTRA.items()))).encode())
    for f in paths:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())
            h.update(fh.read())
    return h.hexdigest()


def _digest():
    return _hash(sources() + _headers())


def _compile(src, hipcc, verbose):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    dig = _hash([src] + _headers())
    if os.path.exists(obj) and os.path.exists(obj + ".stamp") and open(obj + ".stamp").read() == dig:
        return obj, None
    cmd = [hipcc] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        return obj, res.stdout + res.stderr
    if verbose:
        sys.stderr.write(res.stderr)
    with open(obj + ".stamp", "w") as fh:
        fh.write(dig)
    return obj, None


def build(force: bool = False, verbose: bool = False) -> str:
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, hipcc, verbose), sources()))
    errors = [e for _, e in results if e]
    if errors:
        sys.stderr.write("\n".join(errors))
        raise RuntimeError("hipcc failed building libhwgat_hip.so")
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [o for o, _ in results],
                         capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("linking libhwgat_hip.so failed")
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
