#!/usr/bin/env python3
"""tests/golden/train_spread.json: the end-point ENSEMBLES of the two chaotic D-NeRF recipes (`dnerf`, `dnerf_div`), from which
tests/test_gpu_train.py derives its end-point bars (a bar must come from the reference's own spread).

  reference_runs: the real reference (tools/ref_train_fixture.py NAME --threads N --out ...) at several thread counts -- same
                  recipe, seed and random stream, another summation order inside its CPU kernels: the reference's own response to
                  a last-bit perturbation.  The run of tests/golden/train_parity_NAME.json is the first entry.
  build_runs:     this build with fp32-atomic accumulation (run-to-run summation order) + its deterministic run, per training
                  arithmetic (tools/train_spread.py on the GPU box -> profiles/r04/train_spread_build.json).

    python tools/make_train_spread_fixture.py /tmp/spread_*.json      (build container; the build side is read from profiles/r04)
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = {}
    build = json.load(open(os.path.join(REPO, "profiles", "r04", "train_spread_build.json")))
    for name in ("dnerf", "dnerf_div"):
        fx = json.load(open(os.path.join(REPO, "tests", "golden", f"train_parity_{name}.json")))
        runs = [{"threads": fx.get("threads", 8), "test_psnr": fx["test_psnr"], "test_psnr_mean": fx["test_psnr_mean"],
                 "loss_last20_mean": sum(fx["losses"][-20:]) / 20, "source": f"tests/golden/train_parity_{name}.json"}]
        for path in sorted(sys.argv[1:]):
            d = json.load(open(path))
            if d.get("name") != name or d.get("seed") != fx.get("seed") or any(r["threads"] == d["threads"] for r in runs):
                continue  # (a thread count that is already in the list: the instrumented run repeats one)
            assert d["argv"] == fx["argv"] and len(d["losses"]) == len(fx["losses"]), path
            runs.append({"threads": d["threads"], "test_psnr": d["test_psnr"], "test_psnr_mean": d["test_psnr_mean"],
                         "loss_last20_mean": sum(d["losses"][-20:]) / 20, "source": "tools/ref_train_fixture.py --threads %d" % d["threads"]})
        reg = []  # reference runs made after tools/ref_train_fixture.py learnt to record the term (one trace per thread count)
        for path in sorted(sys.argv[1:]):
            d = json.load(open(path))
            if (d.get("name") == name and d.get("seed") == fx.get("seed") and d.get("reg_terms")
                    and not any(r["threads"] == d["threads"] for r in reg)):
                reg.append({"threads": d["threads"], "reg_terms": d["reg_terms"]})
        reg = reg or None
        out[name] = {"reference_reg": reg, "reference_runs": runs,
                     "build_runs": [{k: r[k] for k in ("train_precision", "deterministic", "test_psnr", "test_psnr_mean", "loss_last20_mean")}
                                    for r in build[name]["build"]]}
        print(name, "reference runs:", [round(r["test_psnr_mean"], 3) for r in runs])
    path = os.path.join(REPO, "tests", "golden", "train_spread.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
