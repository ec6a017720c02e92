"""CPU-side check of the emitted gfx950 ISA of the conv kernels (no GPU needed: hipcc cross-compiles).

The round-1 "stale accumulator" bug was a ROTATED MFMA (vDst != SrcC) whose result was read by the epilogue before it
had landed: such results are not hardware-interlocked and hipcc's wait states for them are too few
(profiles/r2_mfma_hazard.md).  conv.hip closes every accumulator chain with in-place terminator MFMAs inside one asm
statement (mfma_result_guard); this test proves on the real listing that the guard sits where it must."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_rotated_mfma_result_is_read_early(tmp_path):
    import isa_check_mfma
    from coivo_amd import build
    asm = build.emit_asm("conv.hip", str(tmp_path / "conv.s"))
    nmfma, nrot, bad = isa_check_mfma.check(asm, 16, 12)
    assert nmfma > 5000, nmfma                      # every instantiation was seen
    assert not bad, "\n".join("%s: %s %s <- SrcC %s %s after %d wait states" % b[:6] for b in bad[:20])
    text = open(asm).read()
    # accumulators live in VGPRs (-mllvm -amdgpu-mfma-vgpr-form): an AGPR accumulator would be copied out in FRONT of
    # the guard's asm statement, i.e. read unguarded
    assert "v_accvgpr" not in text and " a[" not in text
    # the guard is present in every kernel that has MFMAs
    import re
    kernels = re.split(r"\n(?=_Z\S+:)", text)
    with_mfma = [k for k in kernels if "v_mfma" in k and k.startswith("_Z")]
    assert len(with_mfma) >= 60
    for k in with_mfma:
        m = re.search(r";;#ASMSTART\s+s_nop 1\s+v_mfma\S+ (v\[\d+:\d+\]), (\S+), \2, \1", k)
        assert m, k.split(":")[0]
