#!/usr/bin/env python3
"""First losses of a golden training recipe in deterministic mode (bit-reproducible per build): compare builds / environment
switches line by line.    python tools/fit_losses.py dnerf_div [epochs=12] [train_prec=bf16x3]"""
import json, os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from tools.make_scene import make_scene
from test_gpu_train import procedural_init
import nerf_atlas_amd.train as T
from nerf_atlas_amd import config
name = sys.argv[1]; epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 12; prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
fx = json.load(open(os.path.join(REPO, "tests", "golden", f"train_parity_{name}.json")))
tmp = tempfile.mkdtemp()
data = make_scene(os.path.join(tmp, "scene"), **fx["scene"]) + "/"
argv = [x for x in fx["argv"] if x not in ("-d", "--outdir")]
i = argv.index("--epochs"); argv[i + 1] = str(epochs)
args = T.args_from_argv(["-d", data] + argv + ["--notraintest"] if "--notraintest" not in argv else ["-d", data] + argv)
config.set_precision("bf16x3"); config.set_train_precision(prec); config.set_deterministic(True)
res = T.fit(args, replay_reference_rng=True, init=procedural_init)
print("LOSSES " + " ".join(f"{v:.9e}" for v in res["losses"]))
