"""CPU: build-time audit of encoder_ffn3.hip.  The kernel's [384 x 32] fp32 output tile lives in accumulator registers
a[0:191] that its inline asm names literally; hipcc never sees them.  That is only sound while hipcc itself uses NO
accumulator register and spills nothing (cdna_hip_programming.md section 5.7 item 4): the first build of this kernel faulted
on the GPU because hipcc had parked LDS addresses in a0, a1, ... .  rag_amd._build.audit_ffn3 compiles the file to assembly
with the library's own flags and checks exactly that; the build runs it too and refuses to link a failing object."""


def test_hipcc_keeps_out_of_the_accumulator_tile(tmp_path):
    from rag_amd._build import audit_ffn3
    audit_ffn3(str(tmp_path))
