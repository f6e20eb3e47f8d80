"""CPU: build-time audit of encoder_ffn3.hip.  The kernel's [384 x 32] fp32 output tile lives in accumulator registers
a[0:191] that its inline asm names literally; hipcc never sees them.  That is only sound while hipcc itself uses NO
accumulator register and spills nothing (cdna_hip_programming.md section 5.7 item 4): the first build of this kernel faulted
on the GPU because hipcc had parked LDS addresses in a0, a1, ... .  This test compiles the file to assembly with the
library's own flags and checks exactly that."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hipcc_keeps_out_of_the_accumulator_tile(tmp_path):
    from rag_amd._build import CSRC, FILE_FLAGS, FLAGS
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = str(tmp_path / "ffn3.s")
    subprocess.check_call([hipcc, *FLAGS, *FILE_FLAGS["encoder_ffn3.hip"], "-S", "--cuda-device-only",
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
    assert not bad, bad[:5]
    meta = text[text.rindex("ffn3_kernel"):] if ".vgpr_spill_count" not in body else body
    kd = body[body.index(".amdhsa_kernel"):]
    assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", kd).group(1)) == 0
    accum = int(re.search(r"\.amdhsa_accum_offset (\d+)", kd).group(1))
    nfree = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", kd).group(1))
    assert nfree - accum >= 192 and nfree <= 512, (accum, nfree)
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)
    assert spills and all(int(x) == 0 for x in spills), spills
    assert "scratch_" not in body
