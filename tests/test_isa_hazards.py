"""The shipped binary is checked for the gfx950 wide-store data hazard (tools/isa_hazard_check.py): no store of more
than 64 bits may be followed within two wait states by an instruction that writes one of its data VGPRs.  CPU-side:
it disassembles oat_amd/lib/liboatgpu.so with llvm-objdump."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard_check as H  # noqa: E402


def test_checker_sees_the_sequences_of_round_2():
    store = "buffer_store_dwordx4 v[0:3], v22, s[84:87], s6 offen"
    tail = [(0x100 + 4 * i, "s_mov_b32 s0, 0") for i in range(4)]
    for gap, expect in ((0, 1), (1, 1), (2, 0)):
        body = [(0x0, store)] + [(0x8 + 4 * i, "s_and_b64 s[0:1], s[0:1], s[2:3]") for i in range(gap)]
        body += [(0x40, "v_mov_b32_e32 v1, 16")] + tail
        assert len(H.check_function("f", body)) == expect, gap
    # s_nop 1 = two wait states; a write of another register is no hazard; a branch target is followed
    assert not H.check_function("f", [(0, store), (8, "s_nop 1"), (12, "v_mov_b32_e32 v1, 16")] + tail)
    assert not H.check_function("f", [(0, store), (8, "v_mov_b32_e32 v4, 16")] + tail)
    body = [(0, store), (8, "s_cbranch_execz 2"), (12, "s_nop 1"), (16, "s_nop 0"), (20, "v_swap_b32 v9, v3")] + tail
    assert len(H.check_function("f", body)) == 1
    # global stores carry their data in the second operand
    assert H.check_function("f", [(0, "global_store_dwordx4 v0, v[4:7], s[2:3]"), (8, "s_nop 0"), (12, "v_add_u32_e32 v5, 1, v9")] + tail)
    assert not H.check_function("f", [(0, "global_store_dwordx4 v0, v[4:7], s[2:3]"), (8, "v_add_u32_e32 v0, 1, v9")] + tail)
    assert not H.check_function("f", [(0, "buffer_store_dwordx2 v[0:1], v22, s[84:87], s6 offen"), (8, "v_mov_b32_e32 v1, 16")] + tail)
    #  data in accumulator registers is data too, and is not confused with the VGPR of the same number;
    # an indirect jump inside the window cannot be followed and counts as a hazard
    astore = "buffer_store_dwordx4 a[0:3], v22, s[84:87], s6 offen"
    assert H.check_function("f", [(0, astore), (8, "v_accvgpr_write_b32 a1, v9")] + tail)
    assert not H.check_function("f", [(0, astore), (8, "v_mov_b32_e32 v1, 16")] + tail)
    assert not H.check_function("f", [(0, store), (8, "v_accvgpr_write_b32 a1, v9")] + tail)
    assert H.check_function("f", [(0, store), (8, "s_setpc_b64 s[4:5]")] + tail)


@pytest.mark.skipif(not os.path.exists(H.OBJDUMP), reason="llvm-objdump of ROCm not found")
def test_product_library_has_no_wide_store_hazard():
    lib = os.path.join(ROOT, "oat_amd", "lib", "liboatgpu.so")
    assert os.path.exists(lib), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    bad, n_stores, n_funcs = H.check([lib])
    assert n_stores > 20 and n_funcs > 20          # the disassembly was really read
    assert not bad, bad
