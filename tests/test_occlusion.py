"""N4 occlusion models (src/renderers.py:29-163) over a point light (src/lights.py:69-132) and SDF.intersect_mask
(src/sdf.py:123-135).  tests/golden/g16_occlusion.npz holds outputs of the reference's own classes (procedural weights,
SIREN SDF lifted by `sdf_out_bias_shift` so that the scene has lit and shadowed points): the oracle restatement is pinned
against it on the CPU, and the HIP path (nerf_atlas_amd/renderers.py + lights.py + csrc/march.hip) against both."""
import random

import pytest
import torch

import oracle as O
from conftest import load_golden, golden_params

CASES = [  # tag in the fixture, kind, uses the hit mask, component fn
    ("hard", "hard", True, None),
    ("learned", "learned", True, None),
    ("learned_const", "learned-const", False, None),
    ("all_learned", "all-learned", True, "pos-elaz"),
    ("all_learned_pos", "all-learned", False, "pos"),
    ("joint_all_const", "joint-all-const", False, "pos"),  # JointLearnedConstOcc builds AllLearnedOcc() with its default kind
]
THRESH = {"hard": 1e-3, "learned": 1e-3, "learned-const": 1e-3, "joint-all-const": 1e-3}


def sub_params(g, tag, sigma=4.0):
    sub = {"param_names": g[f"{tag}_param_names"], "param_shapes": g[f"{tag}_param_shapes"]}
    if len(sub["param_names"]) == 0:
        return {}
    return golden_params(sub, sigma=sigma)


def sdf_params(g):
    p = golden_params({"param_names": g["sdf_param_names"], "param_shapes": g["sdf_param_shapes"]})
    p["siren.out.bias"] = p["siren.out.bias"] + float(g["sdf_out_bias_shift"])
    return p


def alpha_of(p):
    for k, v in p.items():
        if k.endswith("alpha"):
            return torch.tensor(0.3)  # tools/gen_golden.py g16 sets every `alpha` to 0.3
    return None


def test_oracle_occlusion_matches_reference():
    g = load_golden("g16_occlusion")
    sp = sdf_params(g)
    sdf_fn = lambda x: O.skip_mlp(sp, "siren.", x, act="sin")
    center, intensity = g["center"][0], g["intensity"][0]
    d, s = O.occlusion(None, {}, g["pts"], center, intensity, sdf_fn, mask=g["mask"])
    assert torch.equal(d, g["none_dir"]) and (s - g["none_spectrum"]).abs().max() <= 1e-6
    for tag, kind, use_mask, comp in CASES:
        p = sub_params(g, tag)
        aux = {}
        d, s = O.occlusion(kind, p, g["pts"], center, intensity, sdf_fn, jitter=float(g["jitter"]),
                           mask=g["mask"] if use_mask else None, alpha=alpha_of(p), component=comp or "pos-elaz", aux=aux)
        assert d.shape == g[f"{tag}_dir"].shape and (d - g[f"{tag}_dir"]).abs().max() <= 1e-6, tag
        if "tput" in aux:
            assert abs(aux["far"] - float(g[f"{tag}_far"])) <= 1e-6, tag
            assert (aux["tput"] - g[f"{tag}_tput"]).abs().max() <= 2e-5, tag
        if "raw_att" in aux:
            assert (aux["raw_att"] - g[f"{tag}_raw_att"]).abs().max() <= 2e-5, tag
        assert (s - g[f"{tag}_spectrum"]).abs().max() <= 2e-5, tag
    # the fixture has lit and shadowed points
    lit = g["hard_spectrum"].abs().sum(-1) > 0
    assert bool(lit.any()) and not bool(lit.all())


@pytest.mark.gpu
def test_hip_occlusion_kinds_match_the_reference():
    from nerf_atlas_amd import config, lights, renderers, sdf as nsdf, refl
    import types
    config.set_precision("bf16x3")
    g = load_golden("g16_occlusion")
    under = nsdf.SIREN(intermediate_size=0)
    s = nsdf.SDF(under, refl.View(latent_size=0, act="upshifted", out_features=3), t_near=0.5, t_far=5.0).cuda().eval()
    sd = under.state_dict()
    for k, v in sdf_params(g).items():
        sd[k].copy_(v)
    light0 = lights.Point(center=[1.5, 2.0, -1.0], intensity=[30.0]).cuda()
    light = next(iter(light0.iter()))
    assert torch.equal(light.center.cpu(), g["center"]) and torch.equal(light.intensity.cpu(), g["intensity"])
    pts, mask = g["pts"].cuda(), g["mask"].cuda()
    with torch.no_grad():
        d, sp = renderers.lighting_wo_isect(pts, light, None, mask=mask)
        assert (d.cpu() - g["none_dir"]).abs().max() <= 1e-6 and (sp.cpu() - g["none_spectrum"]).abs().max() <= 1e-6
        for tag, kind, use_mask, comp in CASES:
            args = types.SimpleNamespace(occ_kind=kind, all_learned_occ_kind=comp)
            occ = renderers.load_occlusion_kind(args, kind, 0)
            occ = occ.cuda().eval() if isinstance(occ, torch.nn.Module) else occ
            p = sub_params(g, tag)
            osd = occ.state_dict()
            for k, v in p.items():
                osd[k].copy_(torch.tensor(0.3) if k.endswith("alpha") else v)
            random.seed(16)
            d, sp = occ(pts, light, s.intersect_mask, mask=mask if use_mask else None, latent=None)
            assert d.shape == g[f"{tag}_dir"].shape and (d.cpu() - g[f"{tag}_dir"]).abs().max() <= 1e-6, tag
            ok = torch.ones(sp.shape[:-1], dtype=torch.bool)
            if f"{tag}_tput" in g:
                # visibility is a threshold on an MLP output: compare away from the SDF's fp noise of the threshold
                ok = (g[f"{tag}_tput"] - THRESH[kind]).abs() > 2e-4
                assert (occ.last_throughput.cpu() - g[f"{tag}_tput"]).abs().max() <= 2e-4, tag
            if f"{tag}_raw_att" in g:
                assert (occ.all_learned_occ.raw_att.cpu() - g[f"{tag}_raw_att"]).abs().max() <= 1e-4, tag
            assert (sp.cpu() - g[f"{tag}_spectrum"])[ok].abs().max() <= 1e-4, tag
        # error behaviour of the reference: shadows need the hit mask (src/renderers.py:37,62), joint refuses it (:138)
        hard = renderers.load_occlusion_kind(types.SimpleNamespace(), "hard", 0)
        with pytest.raises(AttributeError):
            hard(pts, light, s.intersect_mask, mask=None)
        joint = renderers.load_occlusion_kind(types.SimpleNamespace(), "joint-all-const", 0).cuda()
        with pytest.raises(NotImplementedError):
            joint(pts, light, s.intersect_mask, mask=mask)
        with pytest.raises(NotImplementedError):
            renderers.load_occlusion_kind(types.SimpleNamespace(occ_kind="nope"), "nope", 0)
