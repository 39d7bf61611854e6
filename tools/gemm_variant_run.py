import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import _lib
N = 262144
x = torch.randn(N, 256, device="cuda"); W = torch.randn(256, 256, device="cuda") * 0.06; b = torch.zeros(256, device="cuda"); y = torch.empty(N, 256, device="cuda")
MODE = "fwd"
if sys.argv[1] in ("fwd", "dgrad", "wgrad", "dgrad_skip", "fwd_skip"):
    MODE = sys.argv.pop(1)
x1 = torch.randn(N, 38, device="cuda"); Ws = torch.randn(256, 294, device="cuda") * 0.06; Wst = Ws.t().contiguous()
gx1 = torch.empty(N, 38, device="cuda")
dWb = torch.zeros(256, 256, device="cuda"); dbb = torch.zeros(256, device="cuda")
gy = torch.randn(N, 256, device="cuda"); Wt = W.t().contiguous(); gx = torch.empty(N, 256, device="cuda")
for name in sys.argv[1:]:
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(REPO, "nerf_atlas_amd", "libnerf_atlas_amd.so") if name == "shipped" else os.path.join(REPO, "gpurun_ablate", f"lib_var_{name}.so")
    lib = C.CDLL(path)
    fn = lib.na_linear_bf16x3
    fn.argtypes = _lib.SIGNATURES["na_linear_bf16x3"][1]; fn.restype = C.c_int
    st = torch.cuda.current_stream().cuda_stream
    fd = lib.na_linear_dgrad_bf16x3
    fd.argtypes = _lib.SIGNATURES["na_linear_dgrad_bf16x3"][1]; fd.restype = C.c_int
    def f():
        if MODE == "wgrad":
            fw = lib.na_linear_wgrad_bf16x3
            fw.argtypes = _lib.SIGNATURES["na_linear_wgrad_bf16x3"][1]; fw.restype = C.c_int
            assert fw(x.data_ptr(), 256, None, 0, N, gy.data_ptr(), 256, 1, dWb.data_ptr(), dbb.data_ptr(), st) == 0
        elif MODE == "fwd_skip": assert fn(x.data_ptr(), 256, x1.data_ptr(), 38, N, Ws.data_ptr(), b.data_ptr(), 256, 1, y.data_ptr(), st) == 0
        elif MODE == "dgrad_skip": assert fd(gy.data_ptr(), 256, N, Wst.data_ptr(), x.data_ptr(), 256, x1.data_ptr(), 38, 1, gx.data_ptr(), gx1.data_ptr(), st) == 0
        elif MODE == "fwd": assert fn(x.data_ptr(), 256, None, 0, N, W.data_ptr(), b.data_ptr(), 256, 1, y.data_ptr(), st) == 0
        else: assert fd(gy.data_ptr(), 256, N, Wt.data_ptr(), x.data_ptr(), 256, None, 0, 1, gx.data_ptr(), None, st) == 0
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    print(f"{name:10s} {MODE} 262144 x 256{'+38' if 'skip' in MODE else ''} x 256: {(time.perf_counter() - t0) / 10 * 1e6:.0f} us", flush=True)
    if hasattr(lib, "na_debug_tgl_trace"):
        import numpy as np
        buf = np.zeros((3, 32, 4), dtype=np.uint64)
        lib.na_debug_tgl_trace.argtypes = [C.c_void_p]
        assert lib.na_debug_tgl_trace(buf.ctypes.data) == 0
        t0 = int(buf[0, 0, 0])
        print("  consumer wave 0 (unit: start, MFMA done, epilogue done, barrier passed), cycles since unit 0:")
        for u in range(16): print("   ", u, [int(v) - t0 for v in buf[0, u]], " MFMA", int(buf[0, u, 1]) - int(buf[0, u, 0]), "epi", int(buf[0, u, 2]) - int(buf[0, u, 1]), "wait", int(buf[0, u, 3]) - int(buf[0, u, 2]))
        print("  producer wave 4 (unit: start, converted, loads issued, barrier passed):")
        for u in range(16): print("   ", u, [int(v) - t0 for v in buf[1, u]], " convert(+wait for rows)", int(buf[1, u, 1]) - int(buf[1, u, 0]), "issue", int(buf[1, u, 2]) - int(buf[1, u, 1]), "wait", int(buf[1, u, 3]) - int(buf[1, u, 2]))
        print("  mover wave 8 (unit: start, x arrived, tile stored (issue), x parked + requested):")
        for u in range(16): print("   ", u, [int(v) - t0 for v in buf[2, u]], " wait x", int(buf[2, u, 1]) - int(buf[2, u, 0]), "stores", int(buf[2, u, 2]) - int(buf[2, u, 1]), "park", int(buf[2, u, 3]) - int(buf[2, u, 2]))
