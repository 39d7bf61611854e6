"""Command-line entry with the reference's flags for the supported subset (runner.py:38-424, :1221-1322):

    python -m nerf_atlas_amd.runner -d data/nerf_synthetic/lego/ --data-kind original --size 64 --crop-size 24 \\
        --epochs 5000 --model plain --refl-kind view --near 2 --far 6 --batch-size 4 --outdir outputs/ [--save model.pt]
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m nerf_atlas_amd.runner ...   # replicas + gradient all-reduce

seeds, loads the dataset (loaders.py), builds the model through the registries, trains (train.py), renders the test set
with the fused kernels and writes `results.txt` in the reference's format (runner.py:977-995) plus `test_NNN.png`
(expected | rendered, side by side).  Flags outside the hot path are rejected by argparse rather than ignored."""
import os
import sys

import numpy as np
import torch

from . import train as T


def summary(name, psnrs, training=False):
    return (f"[Summary {name} ({'training' if training else 'test'}) @ nerf_atlas_amd]:\n"
            f"\tmean {np.mean(psnrs):.03f}\n\tmedian {np.median(psnrs):.03f}\n\tmin {min(psnrs):.03f}\n"
            f"\tmax {max(psnrs):.03f}\n\tvar {np.var(psnrs):.03f}")


def load_state(path, allow_pickled_module=False):
    """--load: a state_dict written by --save here (tensors only: loaded with weights_only=True, which executes no pickled
    code), or -- only with --load-pickled-module -- a reference checkpoint (runner.py:1141-1166 pickles the whole module with
    torch.save(model, path); unpickling it runs arbitrary code from the file, so it must be a file you trust).  Anything
    exposing .state_dict() is accepted; parameter names are the reference's, so its tensors load into this package's modules
    unchanged."""
    import pickle
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        # not a plain tensor dict: weights_only refuses the file's globals (UnpicklingError; RuntimeError in older torch).
        # I/O problems -- missing file, permissions, a truncated archive (EOFError / OSError / zipfile errors) -- are NOT caught:
        # they say what is wrong themselves, and must never lead to the unsafe full unpickle below
        if isinstance(e, RuntimeError) and not any(k in str(e) for k in ("Unsupported", "weights_only", "GLOBAL", "unpickl")):
            raise
        if not allow_pickled_module:
            raise ValueError(f"{path} is not a plain state_dict ({type(e).__name__}: {str(e)[:120]}); a checkpoint that pickles "
                             f"a whole module executes code from the file when loaded: pass --load-pickled-module to accept it") from e
        obj = torch.load(path, map_location="cpu", weights_only=False)
    if hasattr(obj, "state_dict"):
        obj = obj.state_dict()
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: neither a state_dict nor a module")
    return obj


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    outdir, save, load, keep, pickled = "outputs/", None, None, [], False
    it = iter(argv)
    for a in it:  # flags owned by this wrapper
        if a == "--outdir": outdir = next(it)
        elif a == "--save": save = next(it)
        elif a == "--load": load = next(it)
        elif a == "--load-pickled-module": pickled = True
        else: keep.append(a)
    args = T.args_from_argv(keep)
    state = load_state(load, pickled) if load else None
    quiet = args.quiet

    def on_iter(i, l2):
        if not quiet and (i % 50 == 0 or i == args.epochs - 1):
            print(f"[{i:06}] l2 {l2:.04f}", flush=True)
    init = None
    if state is not None:
        def init(model):
            missing, unexpected = model.load_state_dict(state, strict=False)
            if unexpected:
                raise ValueError(f"--load: parameters of another architecture: {sorted(unexpected)[:4]} ...")
            if missing:  # always reported: a silently half-loaded model trains / renders from random weights
                print(f"--load: {len(missing)} parameters are not in the checkpoint and keep their initial values: "
                      f"{sorted(missing)[:6]}{' ...' if len(missing) > 6 else ''}", file=sys.stderr, flush=True)
            from .utils import invalidate_packed
            invalidate_packed(model)
    res = T.fit(args, on_iter=on_iter, init=init)
    if res["rank"] != 0:
        return res
    os.makedirs(outdir, exist_ok=True)
    lines = []
    labels = res["test_labels"][0] if type(res["test_labels"]) is tuple else res["test_labels"]
    for i, (p, got) in enumerate(zip(res["test_psnr"], res["frames"])):
        mse = 10 ** (-p / 10)
        lines.append(f"[{i:03}]: L2 {mse:.03f} PSNR {p:.03f}")
        print(lines[-1])
        try:
            from PIL import Image
            both = torch.cat([labels[i, ..., :3].cpu(), got.clamp(0, 1).cpu()], dim=1)
            Image.fromarray((both.numpy() * 255).round().astype(np.uint8)).save(os.path.join(outdir, f"test_{i:03}.png"))
        except ImportError:
            pass
    s = summary("", res["test_psnr"])
    print(s)
    with open(os.path.join(outdir, "results.txt"), "w") as f:
        f.write(s)
        for l in lines:
            f.write("\n" + l)
    if save:
        torch.save(res["model"].state_dict(), save)  # state_dict keys = the reference's
        print(f"Saved to {save}")
    return res


if __name__ == "__main__":
    main()
