#!/bin/bash
# Is the layer-synchronous renderer power-limited?  Samples rocm-smi (power, clocks, temperature) while tools/ls_variant.py loops
# the full-frame render in the given precision.   bash tools/power_probe.sh [f16x|bf16|bf16x3|f16] > gpurun_out/power_probe.log
PREC=${1:-f16x}
echo "== idle"; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -30
python - <<PY &
import sys, math, torch, time
sys.path.insert(0, ".")
import bench
from nerf_atlas_amd import ops, config
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
size, T = bench.SIZE, bench.STEPS_PER_RAY
focal = 0.5 * size / math.tan(0.5 * bench.FOV)
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
rays = ops.raygen(c2w, focal, size, (0, 0, size, size))
ts, _ = ops.compute_ts(bench.NEAR, bench.FAR, T, dev)
tables = model.first.enc.tables()
packed = model.packed_ls("$PREC")
t0 = time.time()
n = 0
while time.time() - t0 < 14:
    for _ in range(5):
        ops.render_plain_view_ls(rays, ts, tables, packed, "$PREC", "upshifted", "black", want_weights=False)
    torch.cuda.synchronize(); n += 5
dt = time.time() - t0
print(f"$PREC: {n} frames in {dt:.1f} s = {n * size * size * T / dt / 1e6:.0f} Msamples/s", flush=True)
PY
sleep 6
for i in 1 2 3; do echo "== under load ($PREC) sample $i"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|socclk\|temp" | head -14; sleep 2; done
wait
echo "== power cap"; rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head
