"""The build-time ISA guard of the layer-synchronous renderer (nerf_atlas_amd/build.py: check_isa; tools/check_isa.py):
render_ls_kernel must not contain compiler-formed packed fp32 arithmetic (DESIGN 3b "Reproducibility")."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_built_render_ls_units_have_no_packed_fp32():
    from nerf_atlas_amd import build as B
    B.build(verbose=False)  # no-op when up to date; rebuilds (and checks) the render_ls units otherwise
    listings = B.isa_listings()
    assert {n for n, _ in listings} == {"render_ls_bf16.o", "render_ls_bf16x3.o", "render_ls_f16.o", "render_ls_f16x.o"}
    for name, path in listings:
        bad, seen = B.check_isa(path)
        assert len(seen) >= (1 if "f16x" in name else 4), (name, seen)  # MODEL 0..3 of the unit's precision (f16x: MODEL 0)
        assert not bad, (name, {k: v[:3] for k, v in bad.items()})


def test_check_isa_flags_an_offending_listing(tmp_path):
    from nerf_atlas_amd import build as B
    lst = tmp_path / "fake.s"
    lst.write_text(
        "_ZN2na2ls16render_ls_kernelILi1ELi0EEEvNS0_4ArgsE:\n"
        "\tv_mul_f32_e32 v1, v2, v3\n"
        "\tv_pk_mul_f32 v[4:5], v[6:7], v[8:9] op_sel_hi:[1,0]\n"
        ".Lfunc_end0:\n"
        "_ZN2na5otherEv:\n"
        "\tv_pk_add_f32 v[4:5], v[6:7], v[8:9]\n"
        ".Lfunc_end1:\n")
    bad, seen = B.check_isa(str(lst))
    assert seen == ["_ZN2na2ls16render_ls_kernelILi1ELi0EEEvNS0_4ArgsE"]
    assert list(bad.values()) == [[(3, "v_pk_mul_f32")]]


def test_built_f16x_unit_has_no_overlapping_fp6_conversion():
    """v_cvt_scalef32_2xpk16_fp6_f32 writes its destination while it still reads its operands (tools/hw/cvt_fp6_overlap.hip);
    the compiler does not know: every instance of the shipped build keeps scale and source tails out of the destination"""
    from nerf_atlas_amd import build as B
    B.build(verbose=False)
    scanned = 0
    for name, path in B.isa_listings():
        bad, n = B.check_cvt_overlap(path)
        assert not bad, (name, bad[:3])
        scanned += n
    assert scanned >= 30, scanned  # (the f16x unit holds 34 of them: 8 activation stores x 2 planes x 2 blocks + the packs)


def test_check_cvt_overlap_flags_the_cases_the_hardware_gets_wrong(tmp_path):
    from nerf_atlas_amd import build as B
    lst = tmp_path / "fake.s"
    lst.write_text(
        "\tv_cvt_scalef32_2xpk16_fp6_f32 v[96:101], v[64:79], v[80:95], v102\n"    # disjoint: fine
        "\tv_cvt_scalef32_2xpk16_fp6_f32 v[16:21], v[16:31], v[0:15], v59\n"      # destination = head of src0: fine
        "\tv_cvt_scalef32_2xpk16_fp6_f32 v[0:5], v[32:47], v[0:15], v23\n"        # destination = head of src1: fine
        "\tv_cvt_scalef32_2xpk16_fp6_f32 v[0:5], v[32:47], v[48:63], v1\n"        # scale inside the destination (the max-ilp pack)
        "\tv_cvt_scalef32_2xpk16_fp6_f32 v[38:43], v[48:63], v[32:47], v157\n"    # destination inside src1, not on its head
        "\tv_cvt_scalef32_2xpk16_bf6_f32 v[90:95], v[64:79], v[80:95], v102\n")   # destination = tail of src1
    bad, n = B.check_cvt_overlap(str(lst))
    assert n == 6
    assert [b[0] for b in bad] == [4, 5, 6], bad
