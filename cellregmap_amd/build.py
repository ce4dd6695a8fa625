"""Builds libcrm_hip.so (hand-written HIP kernels + C-ABI) for gfx950, in-tree.

    python -m cellregmap_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the
repository snapshot.  One object per source, rebuilt only when the source (or a header)
is newer than the object.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libcrm_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}",
            "-Wall", "-Wno-unused-function"]
LDLIBS = []  # no vendor BLAS / solver: every kernel of the path is in csrc/


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_mtime)):
        return obj, False
    cmd = [HIPCC, *CXXFLAGS, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr = _headers_mtime()
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr), _sources()))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(LIB):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs, *LDLIBS]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[cellregmap_amd.build] linked {LIB}")
    elif verbose:
        print(f"[cellregmap_amd.build] {LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
