"""Build libgeopolars_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m geopolars_amd.build [--force] [--verbose]

Objects go to geopolars_amd/csrc/build/, the shared library to geopolars_amd/libgeopolars_hip.so
(git-ignored, but it travels with the tree to the GPU box).  -ffp-contract=off is part of the
numerical contract: kernels must round exactly like the CPU semantics they restate
(see csrc/gpk_device.h); the only fused operations are explicit fma() calls.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libgeopolars_hip.so")
ARCH = "gfx950"

SOURCES = [
    "gpk_runtime.hip",
    "gpk_unary.hip",
    "gpk_ringstream.hip",
    "gpk_join.hip",
    "gpk_pipflow.hip",
    "gpk_pipindex.hip",
    "gpk_rowwise.hip",
    "gpk_hull.hip",
    "gpk_wkb.cpp",
    "gpk_arrow.cpp",
    "gpk_wkb_device.hip",
    "gpk_wkb_encode.hip",
    "gpk_take.hip",
    "gpk_structural.hip",
    "gpk_lineal_ops.hip",
    "gpk_comm.hip",
]

FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-result",
]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libgeopolars_hip.so cannot be built (there is no CPU fallback)")


def _deps_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for fn in os.listdir(root):
            if fn.endswith((".h", ".hip", ".cpp")):
                m = max(m, os.path.getmtime(os.path.join(root, fn)))
    return m


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    lang = ["-x", "hip"] if src.endswith(".hip") else []
    cmd = [_hipcc(), *FLAGS, *lang, "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build_variant(name: str, defines: list[str], only: list[str] | None = None) -> str:
    """A/B builds for kernel tuning: same sources with extra -D flags -> geopolars_amd/variants/<name>.so
    (select at run time with GPK_LIB_PATH).  `only`: the sources the flags affect; the others are linked from the
    objects of the in-tree build (run build() first)."""
    vdir = os.path.join(HERE, "variants")
    odir = os.path.join(OBJ, name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    objs = []
    for src in SOURCES:
        if only is not None and src not in only:
            objs.append(os.path.join(OBJ, os.path.splitext(src)[0] + ".o"))
            continue
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        cmd = [_hipcc(), *FLAGS, *[f"-D{d}" for d in defines], *lang, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        objs.append(obj)
    out = os.path.join(vdir, f"{name}.so")
    r = subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out, *objs], capture_output=True, text=True)
    shutil.rmtree(odir, ignore_errors=True)  # objects of a variant are not reused; keep the snapshot sent to the GPU box small
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
