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
