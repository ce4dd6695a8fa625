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
LDLIBS = ["-ldl"]  # no vendor BLAS / solver: every kernel of the path is in csrc/ (dl: optional roctx ranges)


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


def build_asan(verbose=True):
    """Host side only (no device code), with AddressSanitizer: cellregmap_amd/_build/libcrm_hip_asan.so.  GPU
    AddressSanitizer is not available on this pool; this build serves the host-only hooks (tests/test_asan_cpu.py):
    everything that runs without a GPU -- argument checks, the divide-and-conquer planning, error paths."""
    out_dir = os.path.join(OBJ, "asan")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(OBJ, "libcrm_hip_asan.so")
    flags = ["--offload-host-only", "-fsanitize=address", "-fno-omit-frame-pointer", "-O1", "-g", "-std=c++17", "-fPIC"]

    def one(src):
        obj = os.path.join(out_dir, src.replace(".hip", ".o"))
        r = subprocess.run([HIPCC, *flags, "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (asan) failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, _sources()))
    # the host objects refer to their (absent) device images: give every one an empty offload bundle
    nm = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True).stdout
    syms = sorted({ln.split()[-1] for ln in nm.splitlines() if "__hip_fatbin_" in ln})
    stub = os.path.join(out_dir, "no_device_images.c")
    with open(stub, "w") as fh:
        fh.write("/* generated: empty clang offload bundles (magic + zero entries) for the host-only build */\n")
        for sym in syms:
            fh.write(f'const char {sym}[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";\n')
    stub_o = stub.replace(".c", ".o")
    subprocess.check_call(["gcc", "-fPIC", "-c", stub, "-o", stub_o])
    r = subprocess.run([HIPCC, "--offload-host-only", "-fsanitize=address", "-shared", "-fPIC", "-o", lib, *objs, stub_o, *LDLIBS],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link (asan) failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[cellregmap_amd.build] linked {lib}")
    return lib


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan()
    else:
        build(force="--force" in sys.argv)
