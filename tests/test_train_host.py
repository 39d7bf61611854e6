"""Host-side logic of the training recipe that needs no GPU: argument handling (runner.py:38-437), loss table,
batch sharding."""
import pytest

from nerf_atlas_amd import train as T


def test_make_args_mirrors_the_reference_post_processing():
    a = T.make_args(data="x/", size=64, crop_size=24)
    assert a.render_size == 64 and a.feature_space == 3           # runner.py:430-432 (no neural upsampling)
    assert a.test_crop_size == 24                                  # runner.py:436
    assert a.learning_rate == 5e-4 and a.sched_min == 5e-5 and a.sigmoid_kind == "upshifted" and a.steps == 64
    with pytest.raises(AssertionError):
        T.make_args(nonsense=1)


def test_args_from_argv_accepts_reference_flags():
    a = T.args_from_argv(["-d", "scene/", "--data-kind", "dnerf", "--size", "48", "--crop-size", "24", "--batch-size", "2",
                          "--steps", "48", "--epochs", "200", "--model", "plain", "--dyn-model", "plain", "--spline", "4",
                          "--refl-kind", "view", "--near", "2", "--far", "6", "-lr", "1e-3", "--nosave", "--quiet",
                          "--notraintest", "--valid-freq", "100", "--loss-fns", "l2", "l1", "--no-sched"])
    assert (a.data, a.data_kind, a.spline, a.dyn_model, a.epochs) == ("scene/", "dnerf", 4, "plain", 200)
    assert a.learning_rate == 1e-3 and a.no_sched and a.loss_fns == ["l2", "l1"] and a.render_size == 48
    with pytest.raises(SystemExit):
        T.args_from_argv(["-d", "x/", "--rig-points", "4"])          # outside the hot path: rejected, not ignored


def test_loss_table_and_regulariser_guards():
    import torch
    x, y = torch.tensor([[0.5, 0.25]]), torch.tensor([[0.0, 0.25]])
    assert float(T.loss_map["l2"](x, y)) == pytest.approx(0.125)
    assert float(T.loss_map["l1"](x, y)) == pytest.approx(0.25)
    assert float(T.loss_map["rmse"](x, y)) == pytest.approx(0.125 ** 0.5)
    f = T.load_loss_fn(T.make_args(loss_fns=["l2", "l1"]))
    assert float(f(x, y)) == pytest.approx((0.125 + 0.25) / 2)


def test_load_state_executes_no_pickled_code_by_default(tmp_path):
    """--load: plain state_dicts load with weights_only=True; a pickled module needs --load-pickled-module"""
    import torch
    from nerf_atlas_amd import runner
    sd = {"a.weight": torch.arange(6.0).reshape(2, 3)}
    p1 = tmp_path / "sd.pt"
    torch.save(sd, p1)
    assert torch.equal(runner.load_state(str(p1))["a.weight"], sd["a.weight"])
    p2 = tmp_path / "module.pt"
    torch.save(torch.nn.Linear(3, 2), p2)  # what the reference's runner writes: the whole module
    with pytest.raises(ValueError, match="load-pickled-module"):
        runner.load_state(str(p2))
    assert set(runner.load_state(str(p2), allow_pickled_module=True)) == {"weight", "bias"}
    # I/O errors are not "a pickled module": they propagate as themselves and never reach the unsafe full unpickle
    with pytest.raises(FileNotFoundError):
        runner.load_state(str(tmp_path / "missing.pt"), allow_pickled_module=True)
    p3 = tmp_path / "truncated.pt"
    p3.write_bytes(p1.read_bytes()[:40])
    with pytest.raises(Exception) as ei:
        runner.load_state(str(p3))
    assert not isinstance(ei.value, ValueError) or "load-pickled-module" not in str(ei.value)


def test_invalidate_packed_clears_every_stream_cache():
    import torch
    from nerf_atlas_amd.utils import invalidate_packed, PACK_CACHE_ATTRS
    root = torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.Sequential(torch.nn.Linear(2, 2)))
    mods = list(root.modules())
    for i, m in enumerate(mods):
        m.__dict__[PACK_CACHE_ATTRS[i % len(PACK_CACHE_ATTRS)]] = {"bf16x3": ("stamp", object())}
    assert invalidate_packed(root) == len(mods)
    assert all(not m.__dict__[PACK_CACHE_ATTRS[i % len(PACK_CACHE_ATTRS)]] for i, m in enumerate(mods))
    assert invalidate_packed(root) == 0


def test_packed_caches_drop_on_load_state_dict_and_apply():
    """`load_state_dict` (post hook on every caching submodule) and `_apply` (`.float()`, `.to()` ...) invalidate the packed
    streams on their own; `config.repack_always` turns the version-counter stamp off altogether."""
    import torch
    from nerf_atlas_amd import config, nerf, utils
    m = nerf.PlainNeRF(steps=8, intermediate_size=64)

    def poison():
        m.first._packed[("bf16x3", "generic")] = ("stamp", object())
        m.refl.mlp._packed[("bf16x3", "generic")] = ("stamp", object())
        m.__dict__.setdefault("_packed_ls", {})["f16x"] = ("stamp", object())
        m.first.enc._stacked, m.first.enc._stamp = torch.zeros(1), ("stamp",)

    def clean():
        return (not m.first._packed and not m.refl.mlp._packed and not m.__dict__["_packed_ls"] and m.first.enc._stacked is None)

    poison()
    m.load_state_dict(m.state_dict())
    assert clean()
    poison()
    m.float()
    assert clean()
    poison()
    assert m.invalidate_packed() == 4 and clean()
    lin = m.first._linears()
    assert utils.pack_stamp(lin) is not None
    config.set_repack_always(True)
    try:
        assert utils.pack_stamp(lin) is None
    finally:
        config.set_repack_always(False)


def test_train_forward_switch_and_the_step_byte_model():
    """config.train_forward ("ls" = the one-launch training forward, "layers" = one Linear per launch) and bench.py's model of the bytes a
    training step has to move with either: the one-launch forward saves exactly the forward's input reads, 4 B per input column."""
    import importlib
    import os
    import sys
    from nerf_atlas_amd import config
    assert config.train_forward in ("ls", "layers")
    prev = config.train_forward
    try:
        config.set_train_forward("layers")
        assert config.train_forward == "layers"
        with pytest.raises(ValueError):
            config.set_train_forward("fused")
    finally:
        config.set_train_forward(prev)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    ls, layers = bench.train_algorithmic_bytes_per_sample("ls"), bench.train_algorithmic_bytes_per_sample("layers")
    cols_in = (38 + 294 + 256 * 4) + (69 + 325 + 256 * 4)
    assert layers - ls == 4 * cols_in and layers == 54312


def test_plain_nerf_on_the_cpu_never_takes_the_one_launch_forward():
    """PlainNeRF._train_forward_ls serves CUDA batches of 8 192 .. 2^22 - 1 samples only; anything else returns None (the layer path)."""
    import torch
    import nerf_atlas_amd.nerf as nerf
    m = nerf.PlainNeRF(steps=16, t_near=2.0, t_far=6.0, intermediate_size=64)
    pts = torch.zeros(16, 1, 4, 4, 3)
    assert m._train_forward_ls(torch.zeros(1, 4, 4, 6), torch.linspace(2, 6, 16), pts, torch.zeros(1, 4, 4, 3)) is None


def test_set_sigmoid_renames_the_reflectance_activation():
    """CommonNeRF.set_sigmoid (src/nerf.py:147-151) replaces refl.act; the name the fused renderers gate on (refl.act_kind) follows it."""
    import nerf_atlas_amd.nerf as nerf
    m = nerf.PlainNeRF(steps=16, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
    assert m.sigmoid_kind == "upshifted" and m.refl.act.kind == "upshifted" and m.refl.act_kind == "upshifted"
    m.set_sigmoid("fat")
    assert m.refl.act.kind == "fat" and m.refl.act_kind == "fat"
