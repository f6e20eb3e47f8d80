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

SOURCES = ["capi.hip", "multi.hip", "ingest.hip", "search_exact.hip", "search_mfma.hip", "encoder.hip", "encoder_ffn3.hip", "encoder_qkv_attn.hip", "wordpiece.hip"]
HEADERS = ["common.h", "kernels.h", "encoder_ffn3.h", "encoder_qkv_attn.h", os.path.join("..", "..", "include", "mi355rag.h")]

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


def audit_ffn3(workdir: str | None = None, extra_flags=()) -> None:
    """encoder_ffn3.hip keeps its output tile in accumulator registers a[0:191] that its inline asm names literally; that is
    sound only while hipcc itself touches no accumulator register and spills nothing.  Compiles the file to assembly with
    the library's flags and raises if the kernel breaks that contract (the first build of the kernel faulted on the GPU:
    hipcc had parked LDS addresses in a0, a1, ...)."""
    import re
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        out = os.path.join(td, "ffn3.s")
        subprocess.check_call([hipcc, *FLAGS, *FILE_FLAGS["encoder_ffn3.hip"], *extra_flags, "-S", "--cuda-device-only",
                               os.path.join(CSRC, "encoder_ffn3.hip"), "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
    body = text[text.index("_Z11ffn3_kernel"):]
    inside, bad = False, []
    for ln in body.splitlines():
        if "#ASMSTART" in ln:
            inside = True
        elif "#ASMEND" in ln:
            inside = False
        elif not inside and (re.search(r"\bv_accvgpr", ln) or re.search(r"\ba\[?\d", ln.split(";")[0])):
            bad.append(ln.strip())
    if bad:
        raise RuntimeError(f"encoder_ffn3.hip: hipcc uses accumulator registers of its own ({len(bad)} instructions, e.g. {bad[:3]})")
    kd = body[body.index(".amdhsa_kernel"):]
    if int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", kd).group(1)) != 0 or "scratch_" in body:
        raise RuntimeError("encoder_ffn3.hip: the kernel uses scratch memory")
    accum = int(re.search(r"\.amdhsa_accum_offset (\d+)", kd).group(1))
    nfree = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", kd).group(1))
    if nfree - accum < 192 or nfree > 512:
        raise RuntimeError(f"encoder_ffn3.hip: accumulator allocation {accum}..{nfree} does not hold a[0:191]")
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)
    if not spills or any(int(x) for x in spills):
        raise RuntimeError(f"encoder_ffn3.hip: vector register spills {spills}")


# Files whose kernels (names containing the substring) must compile without a single spilled register or byte of scratch:
# the scan kernels sit at 250-256 VGPRs with two waves per SIMD, and one more live value makes hipcc spill ~200 registers --
# the kernel stays correct and gets 2.3 x slower at 512 queries (a scratch reload in the loop waits for vmcnt(0), which drains
# the LDS-DMA prefetch).  Measured the hard way in round 3.
NO_SPILL = {"search_mfma.hip": "mfma_scan_kernel", "encoder_qkv_attn.hip": "qkv_attn_kernel"}


def audit_spills(remarks: str, name_part: str):
    """Parses hipcc -Rpass-analysis=kernel-resource-usage output; returns [(kernel, spilled VGPRs, scratch bytes)] of the
    offenders among the kernels whose mangled name contains `name_part` (raises if none was reported at all)."""
    import re
    blocks = re.split(r"remark: Function Name: ", remarks)[1:]
    seen, bad = 0, []
    for b in blocks:
        name = b.split()[0]
        if name_part not in name:
            continue
        seen += 1
        sp = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        sc = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        if sp or sc:
            bad.append((name, sp, sc))
    if not seen:
        raise RuntimeError(f"no resource report for kernels named *{name_part}* (hipcc output format changed?)")
    return bad


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
        src = next((os.path.basename(a) for a in cmd if a.endswith(".hip")), None)
        if src not in NO_SPILL:
            subprocess.check_call(cmd)
            return
        # the compiler's own resource report of every kernel in the file, taken from the build that is linked
        r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)
        bad = audit_spills(r.stderr, NO_SPILL[src])
        if bad:
            obj = cmd[cmd.index("-o") + 1]
            if os.path.exists(obj):
                os.remove(obj)                                         # never link such a build
            raise RuntimeError(f"{src}: kernels with register spills or scratch memory: {bad}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(run, jobs))
        if any("encoder_ffn3.hip" in " ".join(j) for j in jobs):
            try:
                audit_ffn3()
            except Exception:
                os.remove(os.path.join(OBJDIR, "encoder_ffn3.o"))      # never link a build that breaks the contract
                raise
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
