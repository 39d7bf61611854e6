#!/usr/bin/env python3
"""The BUILD's own end-point spread on the chaotic D-NeRF recipes (tests/test_gpu_train.py's `dnerf`, `dnerf_div`): the same
recipe, seed and replayed random stream, trained `--runs` times with fp32-atomic accumulation (config.set_deterministic(False):
the summation order of the hash-table scatter and of the weight-gradient partials changes from run to run, a last-bit
perturbation) plus once in the deterministic mode the test uses.  Together with the reference's own runs at different thread
counts (tools/ref_train_fixture.py --threads) this is what tests/golden/train_spread.json records.

    python tools/train_spread.py dnerf dnerf_div [--runs 4] [--out gpurun_out/r04/train_spread_build.json]     (GPU box)
"""
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import nerf_atlas_amd.train as T
    from nerf_atlas_amd import config
    from tools.make_scene import make_scene
    from test_gpu_train import procedural_init
    argv = sys.argv[1:]
    runs, out = 4, os.path.join(REPO, "gpurun_out", "r04", "train_spread_build.json")
    if "--runs" in argv:
        i = argv.index("--runs"); runs = int(argv[i + 1]); del argv[i:i + 2]
    if "--out" in argv:
        i = argv.index("--out"); out = argv[i + 1]; del argv[i:i + 2]
    res = {}
    for name in argv:
        fx = json.load(open(os.path.join(REPO, "tests", "golden", f"train_parity_{name}.json")))
        with tempfile.TemporaryDirectory() as tmp:
            data = make_scene(os.path.join(tmp, "scene"), **fx["scene"]) + "/"
            a = [x for x in fx["argv"] if x not in ("-d", "--outdir")]
            rows = []
            for prec in ("bf16x3", "fp32"):
                for k in range(runs + 1):
                    det = k == 0
                    config.set_precision("bf16x3")
                    config.set_train_precision(prec)
                    config.set_deterministic(det)
                    try:
                        r = T.fit(T.args_from_argv(["-d", data] + a), replay_reference_rng=True, init=procedural_init)
                    finally:
                        config.set_deterministic(False)
                    rows.append({"train_precision": prec, "deterministic": det, "test_psnr": [float(x) for x in r["test_psnr"]],
                                 "test_psnr_mean": float(r["test_psnr_mean"]), "loss_first10": [float(x) for x in r["losses"][:10]],
                                 "loss_last20_mean": float(sum(r["losses"][-20:]) / 20),
                                 "losses": [float(x) for x in r["losses"]] if det else None,
                                 "reg_terms": [float(x) for x in r.get("reg_terms", [])] if det else None})
                    print(name, prec, "det" if det else f"run {k}", [round(x, 3) for x in r["test_psnr"]], round(r["test_psnr_mean"], 4),
                          flush=True)
            res[name] = {"reference": {"test_psnr": fx["test_psnr"], "test_psnr_mean": fx["test_psnr_mean"]}, "build": rows}
    config.set_train_precision("bf16x3")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
