"""CPU: build-time audit of encoder_ffn3.hip.  The kernel's [384 x 32] fp32 output tile lives in accumulator registers
a[0:191] that its inline asm names literally; hipcc never sees them.  That is only sound while hipcc itself uses NO
accumulator register and spills nothing (cdna_hip_programming.md section 5.7 item 4): the first build of this kernel faulted
on the GPU because hipcc had parked LDS addresses in a0, a1, ... .  rag_amd._build.audit_ffn3 compiles the file to assembly
with the library's own flags and checks exactly that; the build runs it too and refuses to link a failing object."""


def test_hipcc_keeps_out_of_the_accumulator_tile(tmp_path):
    from rag_amd._build import audit_ffn3
    audit_ffn3(str(tmp_path))


def test_spill_audit_reads_the_compiler_report():
    """rag_amd/_build.py refuses to link search_mfma.hip when any scan kernel spills (build-time; here only the parser)."""
    from rag_amd._build import audit_spills
    ok = ("x.hip:1:1: remark: Function Name: _Z16mfma_scan_kernelILi3EE [-Rpass]\n    VGPRs: 254 [-R]\n    ScratchSize [bytes/lane]: 0 [-R]\n"
          "    VGPRs Spill: 0 [-R]\nx.hip:9:1: remark: Function Name: _Z5other [-R]\n    ScratchSize [bytes/lane]: 64 [-R]\n    VGPRs Spill: 3 [-R]\n")
    assert audit_spills(ok, "mfma_scan_kernel") == []
    bad = ok.replace("VGPRs Spill: 0", "VGPRs Spill: 170").replace("ScratchSize [bytes/lane]: 0", "ScratchSize [bytes/lane]: 268")
    assert audit_spills(bad, "mfma_scan_kernel") == [("_Z16mfma_scan_kernelILi3EE", 170, 268)]
    import pytest
    with pytest.raises(RuntimeError):
        audit_spills("nothing here", "mfma_scan_kernel")
