"""The build-time ISA guard of the layer-synchronous renderer (nerf_atlas_amd/build.py: check_isa; tools/check_isa.py):
render_ls_kernel must not contain compiler-formed packed fp32 arithmetic (DESIGN 3b "Reproducibility")."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_built_render_ls_units_have_no_packed_fp32():
    from nerf_atlas_amd import build as B
    B.build(verbose=False, stress=True)  # no-op when up to date; rebuilds (and checks) the render_ls units otherwise
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
    B.build(verbose=False, stress=True)
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


def test_built_units_keep_inline_asm_behind_mfma_and_trans_results():
    """The compiler's hazard recogniser does not see the operands of inline assembly: an asm v_max3_f32 / v_fma_mix_f32 / fp6
    conversion scheduled right behind the MFMA or the transcendental op that produces its operand reads stale registers
    (tools/hw/mfma_use_hazard.hip, trans_use_hazard.hip; round 4: last-bit run-to-run differences of the mip renderer).  Every
    listing of the shipped and of the lag-3 stress build is scanned at build time; here once more, with the counts."""
    from nerf_atlas_amd import build as B
    B.build(verbose=False, stress=True)
    paths = [p for _, p in B.hazard_listings()]
    assert len(paths) >= 19, paths  # the 15 units of the product library + the four lag-3 units
    n_mfma = n_trans = 0
    for path in paths:
        bad, n = B.check_mfma_use(path)
        assert not bad, (path, bad[:2])
        n_mfma += n
        bad, n = B.check_trans_use(path)
        assert not bad, (path, bad[:2])
        n_trans += n
    assert n_mfma >= 20000 and n_trans >= 10000, (n_mfma, n_trans)


def test_check_mfma_use_and_trans_use_flag_what_the_hardware_gets_wrong(tmp_path):
    from nerf_atlas_amd import build as B
    lst = tmp_path / "fake.s"
    lst.write_text(
        "_Z1av:\n"
        "\tv_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\n"
        "\tv_mov_b32_e32 v1, v2\n"
        "\ts_nop 7\n"
        "\tv_max3_f32 v0, v0, |v79|, |v3|\n"                                    # 10 wait states behind an 8-pass MFMA: stale
        "\tv_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[0:5], v[8:13], v[96:111], v20, v21 op_sel:[1,0,0] cbsz:2 blgp:2\n"
        "\ts_nop 7\n"
        "\ts_nop 3\n"
        "\tv_fma_mix_f32 v5, v96, 1.0, -v6 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"   # 12: fine
        "\tv_mfma_f32_32x32x16_f16 v[112:127], v[0:3], v[4:7], v[112:127]\n"
        "\tv_mfma_f32_32x32x16_f16 v[112:127], v[0:3], v[4:7], v[112:127]\n"     # accumulation: SrcC, not this rule
        "\ts_nop 1\n"
        "\tv_add_f32_e32 v9, v8, v8\n"                                           # does not touch the accumulator
        "\tv_mul_f32_e32 v9, v113, v8\n"                                         # 3 wait states: stale
        ".Lfunc_end0:\n")
    bad, n = B.check_mfma_use(str(lst))
    assert n == 4 and [b[0] for b in bad] == [2, 11] and bad[0][3] == 9 and bad[1][3] == 3, bad
    lst.write_text(
        "\tv_sin_f32_e32 v5, v1\n"
        "\tv_fma_mix_f32 v7, v5, 1.0, -v6 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"   # next instruction reads the result: stale
        "\tv_exp_f32_e32 v8, v1\n"
        "\ts_nop 0\n"
        "\tv_max3_f32 v0, v0, |v8|, |v3|\n"                                      # one wait state: fine
        "\tv_rcp_f32_e32 v9, v1\n"
        "\tv_sin_f32_e32 v10, v9\n"                                              # trans -> trans: not this rule
        "\tv_exp_f32_e32 v11, v1\n"
        "\tv_mul_f32_e32 v12, v2, v3\n"
        "\tv_mul_f32_e32 v13, v11, v3\n")                                        # second instruction behind: fine
    bad, n = B.check_trans_use(str(lst))
    assert n == 5 and [b[0] for b in bad] == [1], bad


def test_headline_kernel_isa_is_the_pinned_one():
    """the seven schedules of render_ls_kernel live in files of their own (csrc/ls_sched_*.inc) and the engine in
    ls_engine.h / ls_kernel.h; editing another MODEL's schedule, the packing code or the C ABI must not move ONE instruction of
    the headline kernel render_ls_kernel<f16x, MODEL 0>.  Its instruction stream is pinned by digest (csrc/ls_headline_isa.sha256);
    a deliberate change re-blesses it with `python tools/check_isa.py --bless-headline` and re-runs the GPU determinism tests."""
    from nerf_atlas_amd import build as B
    B.build(verbose=False, stress=True)
    path = [p for n, p in B.isa_listings() if n == "render_ls_f16x.o"][0]
    got = B.function_isa_digest(path, B.HEADLINE_SYMBOL)
    assert got is not None, "the headline instantiation is missing from the f16x listing"
    want = open(B.HEADLINE_PIN).read().split()[0]
    assert got[0] == want, (got, "re-bless on purpose: python tools/check_isa.py --bless-headline")
    assert B.function_isa_digest(path, "_ZN2na2ls16render_ls_kernelILi3ELi9EEEvNS0_4ArgsE") is None


def test_no_source_file_of_the_engine_is_a_monolith():
    """the layer-synchronous engine was one 3 600-line file with seven `if constexpr` schedules in one kernel body"""
    from nerf_atlas_amd import build as B
    for f in os.listdir(B.CSRC):
        if f.startswith(("ls_", "render_ls")):
            n = sum(1 for _ in open(os.path.join(B.CSRC, f)))
            assert n <= 1500, (f, n)


def test_store_data_overwrite_scan_flags_the_unprotected_pair(tmp_path):
    """tools/hw/store_soffset_hazard.hip: a VALU write of a 16-byte buffer store's data right behind the store is only protected by
    the compiler when the soffset is an immediate"""
    from nerf_atlas_amd import build as B
    lst = tmp_path / "fake.s"
    lst.write_text(
        "_ZN2na1kEv:\n"
        "\tbuffer_store_dwordx4 v[182:185], v176, s[20:23], s33 offen\n"
        "\tv_lshlrev_b32_e32 v182, 16, v171\n"                       # 0 wait states: the pair gfx950 gets wrong
        "\tbuffer_store_dwordx4 v[10:13], v176, s[20:23], s33 offen\n"
        "\ts_nop 1\n"
        "\tv_mov_b32_e32 v10, 0\n"                                    # 2 wait states: fine
        "\tbuffer_store_dwordx4 v[20:23], v176, s[20:23], 0 offen\n"
        "\tv_mov_b32_e32 v20, 0\n"                                    # immediate soffset: the compiler's own business
        "\tbuffer_store_dwordx4 v[30:33], v176, s[20:23], s33 offen\n"
        "\tv_mov_b32_e32 v40, v30\n"                                  # reads the data: fine
        ".Lfunc_end0:\n")
    bad, n = B.check_store_data_overwrite(str(lst))
    assert n == 3 and [b[0] for b in bad] == [2]
    B.build(verbose=False, stress=True)
    for name, path in B.hazard_listings():
        bad, _ = B.check_store_data_overwrite(path)
        assert not bad, (name, bad[:2])
