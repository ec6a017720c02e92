"""CPU-side check of the emitted gfx950 ISA of the conv kernels (no GPU needed: hipcc cross-compiles).

The round-1 "stale accumulator" bug was a ROTATED MFMA (vDst != SrcC) whose result was read by the epilogue before it
had landed: such results are not hardware-interlocked and hipcc's wait states for them are too few
(profiles/r2_mfma_hazard.md).  conv.hip / wgrad.hip close every accumulator chain with in-place terminator MFMAs inside
one asm statement (conv_common.h mfma_result_guard); this test proves on the real listings that the guard sits where it
must and that no rotated MFMA result is read early anywhere."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("src,min_mfma,min_kernels,agpr_free", [("conv.hip", 5000, 60, True), ("wgrad.hip", 1500, 30, False),
                                                                 ("bwd16.hip", 20, 1, True), ("fwd16.hip", 10, 1, True),
                                                                 ("conv_rt.hip", 500, 4, True), ("wgrad_rt.hip", 15, 1, True)])
def test_no_rotated_mfma_result_is_read_early(tmp_path, src, min_mfma, min_kernels, agpr_free):
    import isa_check_mfma
    from coivo_amd import build
    asm = build.emit_asm(src, str(tmp_path / (src + ".s")))
    nmfma, nrot, bad = isa_check_mfma.check(asm, 16, 12)
    assert nmfma > min_mfma, nmfma                  # every instantiation was seen
    assert not bad, "\n".join("%s: %s %s <- SrcC %s %s after %d wait states" % b[:6] for b in bad[:20])
    text = open(asm).read()
    if agpr_free:
        # conv.hip: accumulators live in VGPRs (-mllvm -amdgpu-mfma-vgpr-form) -- its guard ties them with "+v", and an
        # AGPR accumulator would be copied out in FRONT of the guard's asm statement, i.e. read unguarded
        assert "v_accvgpr" not in text and " a[" not in text
    # the guard is present in every kernel that has MFMAs: an asm statement opening with s_nop 1 and in-place MFMAs
    # (vDst == SrcC, zero A/B operand)
    kernels = [k for k in re.split(r"\n(?=_Z\S+:)", text) if k.startswith("_Z") and "v_mfma" in k]
    assert len(kernels) >= min_kernels
    for k in kernels:
        m = re.search(r";;#ASMSTART\s+s_nop 1\s+v_mfma\S+ ([va]\[\d+:\d+\]), (\S+), \2, \1", k)
        assert m, k.split(":")[0]
    if not agpr_free:
        # wgrad.hip keeps AGPR accumulators; hipcc may shuffle some of them through VGPRs in front of the guard (operand
        # assignment of the asm statement).  That is a read of an MFMA result: safe for the in-place kind (hipcc's tables,
        # and for f32 the hardware interlock), and for a rotated MFMA only after the wait states `check` demanded above --
        # the 1024-thread team kernels (128 registers per lane) do contain rotated MFMAs, so make sure they were looked at
        assert nrot < nmfma


@pytest.mark.parametrize("src,pat", [("fwd16.hip", "k_fwd16_head"), ("bwd16.hip", "k_bwd16"), ("conv_rt.hip", "k_conv_rt")])
def test_hand_pipelined_kernels_keep_their_operand_reads_ahead_of_the_mfmas(src, pat):
    """Round 5: hipcc had sunk every LDS read of k_fwd16_head / k_bwd16 / the first k_conv_rt to just in front of the MFMA that needs it
    ("ds_read, s_waitcnt lgkmcnt(0), v_mfma": 25 / 22 exposed LDS round trips per tile).  Their operand sets
    are now requested one step ahead and pinned with sched_barrier; this holds the emitted ISA to it: no MFMA directly behind a FULL
    drain of the LDS queue that follows a read (counted waits with younger reads in flight are what the pipeline looks like)."""
    import isa_sunk_reads
    res = isa_sunk_reads.count(src, pat)
    assert res, (src, pat)
    for name, n_mfma, sunk in res:
        assert n_mfma >= 20, (name, n_mfma)
        assert sunk == 0, f"{name}: {sunk} of {n_mfma} MFMAs wait for an LDS read issued right in front of them"
