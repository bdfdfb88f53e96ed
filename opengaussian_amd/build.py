"""Build libogs_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

``python -m opengaussian_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a
GPU.  Objects are rebuilt only when a source / header is newer; the .so stays in
``opengaussian_amd/lib/`` so it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(CSRC, "obj")
LIB = os.path.join(LIBDIR, "libogs_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]
# per-file extra flags: the forward preprocess must match the oracle's operation order bit-for-bit
EXTRA = {
    "preprocess_fwd.hip": ["-ffp-contract=off"],
    # k-means argmin: same sum-of-squares operation order as oracle/kmeans_oracle.py (HBM-bound, FMA buys nothing)
    "kmeans.hip": ["-ffp-contract=off"],
    "adam.hip": ["-ffp-contract=off"],
}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(os.path.dirname(HERE), "include")
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, objdir: str = OBJDIR, defines=()) -> str:
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(spath), _newest_header()):
        return obj
    cmd = [hipcc(), *COMMON, *EXTRA.get(src, []), *defines, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[opengaussian_amd.build] {LIB} ({os.path.getsize(LIB)} bytes)")
    return LIB


def build_variant(name: str, defines, verbose: bool = True, only=()) -> str:
    """A-B timing of kernel variants on ONE box: the same sources compiled with extra -D switches into
    lib/variants/libogs_hip_<name>.so (git-ignored like every .so, travels with the gpurun snapshot); a bench run picks it
    up through OGS_LIB_PATH.  Never loaded by the product unless that variable names it."""
    vdir = os.path.join(LIBDIR, "variants")
    odir = os.path.join(OBJDIR, "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    srcs = sources()
    build(verbose=False)        # the default objects: sources not named in `only` are linked from there
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, True, odir, tuple(defines)) if (not only or s in only)
                           else os.path.join(OBJDIR, s.replace(".hip", ".o")), srcs))
    lib = os.path.join(vdir, f"libogs_hip_{name}.so")
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", lib, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[opengaussian_amd.build] variant {name}: {lib} ({' '.join(defines)})")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")],
                      only=tuple(a for a in sys.argv[i + 2:] if a.endswith(".hip")))
    else:
        build(force="--force" in sys.argv)
