"""In-tree build of libisac_hip.so for gfx950 (hipcc cross-compiles without a GPU).

`python -m` is awkward for a package whose name starts with a digit, so this module is
also runnable as a script:  python 5g_..._amd/_build.py [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libisac_hip.so")
SOURCES = ["capi.hip", "echo.hip", "rdm.hip", "music.hip", "cdl.hip", "cdl_os.hip", "cqi.hip", "los.hip"]
HEADERS = ["isac_common.hpp", "fft_lds.hpp", "echo_dev.hpp", os.path.join("..", "..", "include", "isac.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result", "-ffp-contract=on"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and _newer(obj, deps):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link libisac_hip.so in-tree."""
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs,
               "-Wl,-soname,libisac_hip.so", "-Wl,--no-undefined"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
