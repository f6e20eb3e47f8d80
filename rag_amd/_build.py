"""Builds rag_amd/lib/libmi355rag.so from rag_amd/csrc/*.hip with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "..", "build", "obj")
LIB = os.path.join(LIBDIR, "libmi355rag.so")

SOURCES = ["capi.hip", "multi.hip", "ingest.hip", "search_exact.hip", "search_mfma.hip", "encoder.hip", "encoder_ffn3.hip"]
HEADERS = ["common.h", "kernels.h", "encoder_ffn3.h", os.path.join("..", "..", "include", "mi355rag.h")]

# -ffp-contract=off: the canonical score is an explicit fmaf chain; nothing else may be fused.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
         # the SGPR-base LDS-DMA helpers write M0 inside their asm and list it as clobbered (it is a reserved register)
         "-Wno-inline-asm"]


# encoder_ffn3.hip names its accumulator tile a[0:191] literally in inline asm: hipcc must not park spilled vector
# registers in accumulator registers of its own choosing (it has none left to choose: see the kernel's header)
FILE_FLAGS = {"encoder_ffn3.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *FLAGS, *FILE_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
