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
