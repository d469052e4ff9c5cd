"""The shipped binary's register allocation is a measured property of this tree, and it has regressed silently before:
a few bytes of scratch cost the per-pixel kernel 15-80 % (LABNOTES), and 31 scalar registers spilled to vector-register
lanes cost it 5 % (round 3: late kernel arguments).  CPU-side: read the kernel descriptors' metadata out of
oat_amd/lib/liboatgpu.so with llvm-readelf and hold the per-pixel kernel's instantiations to what was measured."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard_check as H  # noqa: E402

READELF = os.path.join(os.path.dirname(H.OBJDUMP), "llvm-readelf")


def kernel_metadata(lib):
    """-> {kernel name: {key: int}} from the AMDGPU metadata notes of every gfx950 code object in lib."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in H.code_objects(lib, tmp):
            txt = subprocess.run([READELF, "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                out[name.group(1)] = {k: int(v) for k, v in re.findall(
                    r"\.(private_segment_fixed_size|sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)", blk)}
    return out


@pytest.mark.skipif(not (os.path.exists(H.OBJDUMP) and os.path.exists(READELF)), reason="llvm tools of ROCm not found")
def test_per_pixel_kernel_register_allocation():
    lib = os.path.join(ROOT, "oat_amd", "lib", "liboatgpu.so")
    assert os.path.exists(lib), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    md = kernel_metadata(lib)
    k1 = {n: m for n, m in md.items() if "k_mog_fused" in n}
    assert len(k1) >= 27, sorted(k1)                              # CH x AUDIT x NTLD x NF x FROZEN x WG as instantiated
    for n, m in k1.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (n, m)     # no scratch, anywhere
        assert m["group_segment_fixed_size"] == 0, (n, m)                                      # K1 uses no LDS
    # template arguments in the mangled name: ILi<CH>ELb<AUDIT>ELb<NTLD>ELi<NF>ELb<FROZEN>ELi<WG>E
    def inst(ch, audit, ntld, nf, frozen):
        """the 256-thread instantiation; its one-wave-a-workgroup twin (r05; product instantiations only) must allocate alike"""
        tag = f"ILi{ch}ELb{int(audit)}ELb{int(ntld)}ELi{nf}ELb{int(frozen)}E"
        hit = [m for n, m in k1.items() if tag + "Li256E" in n]
        assert len(hit) == 1, tag
        twin = [m for n, m in k1.items() if tag + "Li64E" in n]
        assert len(twin) == (0 if audit else 1), tag
        for t in twin:                        # the same occupancy tier (64 registers = 8 waves a SIMD, 72 = 7), no more spills
            assert (t["vgpr_count"] <= 64) == (hit[0]["vgpr_count"] <= 64) and t["vgpr_count"] <= 72 and \
                ((t["sgpr_count"] + 6 <= 96) == (hit[0]["sgpr_count"] + 6 <= 96)) and \
                t["sgpr_spill_count"] <= hit[0]["sgpr_spill_count"] + 2, (tag, t, hit[0])
        return hit[0]
    for frozen in (False, True):
        two = inst(3, False, False, 2, frozen)                    # the product kernel of the pipelined path
        assert two["vgpr_count"] <= 64, two                       # 8 waves a SIMD
        assert two["sgpr_spill_count"] <= 6, two                  # 31 before the late arguments
        one = inst(3, False, False, 1, frozen)
        assert one["vgpr_count"] <= 64 and one["sgpr_spill_count"] <= 14, one
    dense = inst(3, False, True, 2, False)                        # the streaming-load instantiation: 7 waves a SIMD
    assert dense["vgpr_count"] <= 72 and dense["sgpr_spill_count"] <= 10, dense
    # the ONE-frame instantiations (r04): what a caller that collects every frame before the next one runs, and SURVEY 8d's
    # literal 205 B/px launch.  Everyday model: 8 waves a SIMD is what it lives on (64 vector registers, at most 96 scalar
    # registers incl. the 6 the hardware adds: compiled for 7 waves it took 99 and lost the eighth wave -- 76 -> 81 us at 4K).
    for frozen in (False, True):
        one = inst(3, False, False, 1, frozen)
        assert one["vgpr_count"] <= 64 and one["sgpr_count"] + 6 <= 96, one
    assert inst(3, False, False, 1, False)["sgpr_spill_count"] <= 6
    # dense model, one frame a launch: no scalar spill any more (13 in round 3), still 8 waves a SIMD in hardware
    dense1 = inst(3, False, True, 1, False)
    assert dense1["vgpr_count"] <= 64 and dense1["sgpr_spill_count"] == 0 and dense1["sgpr_count"] + 6 <= 96, dense1
    for ch in (1,):                                               # GREY one-frame instantiations: 8 waves, no scratch (checked above)
        assert inst(ch, False, False, 1, False)["vgpr_count"] <= 64 and inst(ch, False, True, 1, False)["vgpr_count"] <= 64


@pytest.mark.skipif(not (os.path.exists(H.OBJDUMP) and os.path.exists(READELF)), reason="llvm tools of ROCm not found")
def test_no_kernel_of_the_library_uses_scratch():
    md = kernel_metadata(os.path.join(ROOT, "oat_amd", "lib", "liboatgpu.so"))
    assert len(md) >= 30
    assert not {n: m for n, m in md.items() if m.get("private_segment_fixed_size", 0) or m.get("vgpr_spill_count", 0)}
