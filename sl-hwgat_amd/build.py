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
# kernel-lab build (`python sl-hwgat_amd/build.py --lab`): a SECOND library, libhwgat_hip_lab.so, with the A/B environment
# switches compiled in (-DHWGAT_LAB, csrc/common.h).  Only tools/*_lab.py load it; the product and the tests never do.
LAB_LIB = os.path.join(HERE, "libhwgat_hip_lab.so")
# per-source extra flags (none at present)
EXTRA = {}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [HEADER]


def _hash(paths, flags=None):
    h = hashlib.sha256((" ".join(flags or FLAGS) + repr(sorted(EXTRA.items()))).encode())
    for f in paths:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())
            h.update(fh.read())
    return h.hexdigest()


def _digest():
    return _hash(sources() + _headers())


def _compile(src, hipcc, verbose, objdir=None, flags=None):
    objdir, flags = objdir or OBJ, flags or FLAGS
    obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
    dig = _hash([src] + _headers(), flags)
    if os.path.exists(obj) and os.path.exists(obj + ".stamp") and open(obj + ".stamp").read() == dig:
        return obj, None
    cmd = [hipcc] + flags + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
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


def build(force: bool = False, verbose: bool = False, lab: bool = False) -> str:
    lib, stamp = (LAB_LIB, LAB_LIB + ".stamp") if lab else (LIB, STAMP)
    flags = FLAGS + ["-DHWGAT_LAB"] if lab else FLAGS
    objdir = OBJ + "_lab" if lab else OBJ
    dig = _hash(sources() + _headers(), flags)
    if not force and os.path.exists(lib) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == dig:
                return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(objdir, exist_ok=True)
    if force:
        for f in os.listdir(objdir):
            os.remove(os.path.join(objdir, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, hipcc, verbose, objdir, flags), sources()))
    errors = [e for _, e in results if e]
    if errors:
        sys.stderr.write("\n".join(errors))
        raise RuntimeError(f"hipcc failed building {os.path.basename(lib)}")
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for o, _ in results],
                         capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"linking {os.path.basename(lib)} failed")
    with open(stamp, "w") as fh:
        fh.write(dig)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, lab="--lab" in sys.argv))
