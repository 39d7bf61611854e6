"""Layer-synchronous training GEMMs (csrc/train_gemm.hip, round 3): forward, input gradient and weight gradient of one
SkipConnMLP Linear (src/neural_blocks.py:288-296 differentiated) at batch sizes that take the layer-synchronous kernels
(N >= 2048), against an fp64 torch reference, over the shapes the five configs use: hidden 256 x 256, the encoder inputs (38, 69
columns: unaligned rows), the skip concatenations [256 | 38], [256 | 69] (two passes over the output columns / two launches of the
weight gradient), the narrow outputs 65 and 3, ragged batch sizes (partial last tile), every activation.
The small-batch path (the K-staged kernels) is covered by tests/test_gpu_backward.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ACTS = {"none": (lambda v: v, lambda v: torch.ones_like(v)),
        "leaky_relu": (lambda v: torch.where(v > 0, v, 0.01 * v), lambda v: torch.where(v > 0, 1.0, 0.01).to(v.dtype)),
        "sin": (torch.sin, torch.cos)}
# (N, in0, in1, out)
SHAPES = [(4096, 256, 0, 256), (2048 + 37, 256, 0, 256), (8192 + 5, 38, 0, 256), (4096, 256, 38, 256), (4096 + 63, 256, 69, 256),
          (4096, 69, 0, 256), (4096 + 1, 256, 0, 65), (4096, 256, 0, 3), (16384 + 64 * 256 + 9, 256, 0, 64), (40000, 16, 0, 256),
          # beyond the five configs: four passes over 1000 output columns, eight k chunks, an aligned second source, 512 x 512
          (4096, 512, 0, 512), (5000, 100, 0, 1000), (3000, 1000, 24, 72), (2048, 128, 128, 128),
          # a k chunk that holds columns of both sources (the loader's own instantiation), rows of 38 and 63 floats (16-byte
          # fetches at 4-byte alignment, tails zeroed at conversion), 6 and 7 column groups (a pair / three waves in the last pass)
          (4096 + 3, 64, 38, 64), (4096, 63, 0, 128), (4096 + 17, 256, 0, 325), (2048 + 5, 128, 0, 448),
          # both sources ragged: 16-byte pieces of the gradient straddle the boundary between the two outputs at column 38
          (4096 + 1, 38, 69, 128)]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from nerf_atlas_amd import ops
    return ops


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "N%d_in%d+%d_out%d" % s)
@pytest.mark.parametrize("act", ["leaky_relu", "sin", "none"])
def test_layer_synchronous_gemms_vs_fp64(ops, shape, act):
    N, in0, in1, out = shape
    torch.manual_seed(N + in0 + 7 * in1 + out)
    dev = "cuda"
    x0 = torch.randn(N, in0, device=dev)
    x1 = torch.randn(N, in1, device=dev) if in1 else None
    W = torch.randn(out, in0 + in1, device=dev) * (1.0 / (in0 + in1)) ** 0.5
    b = torch.randn(out, device=dev)
    gy = torch.randn(N, out, device=dev)
    f, df = ACTS[act]
    xin = (torch.cat([x0, x1], 1) if in1 else x0).double()
    y_ref = f(xin) @ W.double().t() + b.double()
    g_ref = (gy.double() @ W.double()) * df(xin)
    dW_ref = gy.double().t() @ f(xin)
    db_ref = gy.double().sum(0)

    y = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True)
    g0, g1 = ops.linear_dgrad(gy, W, x0, act, x1=x1)
    g = torch.cat([g0, g1], 1) if in1 else g0
    dW, db = ops.linear_wgrad(x0, gy, act, x1=x1, split_bf16=True)

    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    # 2-way split bf16 products, fp32 accumulation: ~2^-16 relative per dot product (measured 2e-6 .. 1e-5)
    assert rel(y, y_ref) < 3e-5
    assert rel(g, g_ref) < 3e-5
    assert rel(dW, dW_ref) < 3e-5
    assert rel(db, db_ref) < 3e-5
    # the weight gradient accumulates into its output (the caller zero-fills): a second call doubles it, bit for bit twice the same
    dW2, db2 = ops.linear_wgrad(x0, gy, act, x1=x1, split_bf16=True)
    if out <= 256 and in0 <= 256 and in1 <= 256:  # (wider shapes take the K-staged kernel: fp32 atomics, or the fixed-point mode)
        assert torch.equal(dW, dW2) and torch.equal(db, db2), "the layer-synchronous weight gradient is run-to-run reproducible"
    else:
        assert rel(dW2, dW_ref) < 3e-5


def test_only_requested_gradient_halves_are_written(ops):
    """want0 / want1 (a frozen encoder input): the unrequested half is neither allocated nor written"""
    N = 4096
    torch.manual_seed(1)
    x0, x1 = torch.randn(N, 256, device="cuda"), torch.randn(N, 38, device="cuda")
    W = torch.randn(256, 294, device="cuda") * 0.05
    gy = torch.randn(N, 256, device="cuda")
    full0, full1 = ops.linear_dgrad(gy, W, x0, "leaky_relu", x1=x1)
    only0, none1 = ops.linear_dgrad(gy, W, x0, "leaky_relu", x1=x1, want1=False)
    none0, only1 = ops.linear_dgrad(gy, W, x0, "leaky_relu", x1=x1, want0=False)
    assert none1 is None and none0 is None
    assert torch.equal(only0, full0) and torch.equal(only1, full1)


def test_layer_synchronous_and_k_staged_kernels_agree(tmp_path):
    """The two implementations of the forward / input gradient (the choice is made once per process: two subprocesses) form the
    same three bf16 products per k in fp32 and add them in another order: equal to a few units in the last place of the O(1)
    values (measured 2e-7 .. 1e-6 absolute), not bit for bit."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gemm_ls_vs_tiled.py")
    files = {}
    for mode in ("ls", "tiled"):
        env = dict(os.environ)
        env.pop("NA_TRAIN_GEMM", None)
        if mode == "tiled":
            env["NA_TRAIN_GEMM"] = "tiled"
        files[mode] = str(tmp_path / f"{mode}.pt")
        r = subprocess.run([sys.executable, tool, files[mode]], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    a, b = torch.load(files["ls"]), torch.load(files["tiled"])
    worst = 0.0
    for k in a:
        for ta, tb in zip(a[k], b[k]):
            if ta is None:
                assert tb is None
                continue
            assert ta.shape == tb.shape
            worst = max(worst, float((ta - tb).abs().max() / tb.abs().max()))
    print(f"layer-synchronous vs K-staged: worst relative difference {worst:.2e}")
    assert worst <= 2e-6, worst


PK_SHAPES = [(4096, 256, 0, 256), (2048 + 37, 38, 0, 256), (4096 + 63, 256, 69, 256), (4096, 256, 38, 256), (4096 + 1, 256, 0, 65),
             (4096, 256, 0, 3), (5000, 100, 0, 1000), (4096 + 3, 64, 38, 64), (40000 + 31, 256, 0, 128), (2048 + 1, 69, 0, 256),
             (4096 + 9, 128, 100, 256), (262144, 256, 38, 256)]


@pytest.mark.parametrize("shape", PK_SHAPES, ids=lambda s: "N%d_in%d+%d_out%d" % s)
def test_packed_operands_are_the_same_gemms(ops, shape):
    """Round 5: ops.train_pack_many builds the bf16 hi / lo fragments of W and of W^T (read straight from W: no transposing copy)
    in ONE launch; the _pk entry points run the same kernels on them: forward and input gradient bit for bit equal to the entry
    points that pack per call."""
    N, in0, in1, out = shape
    torch.manual_seed(out + in0)
    x0 = torch.randn(N, in0, device="cuda")
    x1 = torch.randn(N, in1, device="cuda") if in1 else None
    W = torch.randn(out, in0 + in1, device="cuda") * (1.0 / (in0 + in1)) ** 0.5
    W2 = torch.randn(77, 300, device="cuda")          # an unrelated matrix in the same launch: entries do not disturb each other
    b = torch.randn(out, device="cuda")
    gy = torch.randn(N, out, device="cuda")
    assert ops.train_gemm_packed_ok(N, out) and ops.train_gemm_packed_ok(N, in0 + in1) and not ops.train_gemm_packed_ok(100, out)
    pf, p2, pt = ops.train_pack_many([(W, False), (W2, True), (W, True)])
    # narrow outputs with K = 256 take the row-stream kernel (csrc/train_gemm.hip nrw: the same three products per k added in
    # another order) -- equal to a few units in the last place, like the two wide implementations among themselves; a skip layer's
    # wide half then runs as a plain single-pass layer: bit for bit the wide kernel on [W_h] alone
    nf = ((in1 == 0 and in0 == 256 and out <= 128) or (in0 == 256 and out == 256 and in1 <= 80 and N >= 8192)  # (+ train_fwd.hip: W resident in registers)
          or (in1 == 0 and in0 <= 80 and out == 256 and N >= 8192))
    n0 = in1 == 0 and out == 256 and in0 <= 128
    n1 = in1 > 0 and out == 256 and in0 % 64 == 0 and in1 <= 128
    same = lambda a, b, narrow: torch.equal(a, b) if not narrow else float((a - b).abs().max()) <= 3e-6 * float(b.abs().max())
    for act in ("leaky_relu", "sin", "none"):
        y = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True)
        y_pk = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True, packed=pf)
        assert same(y_pk, y, nf), (act, float((y - y_pk).abs().max()))
        g0, g1 = ops.linear_dgrad(gy, W, x0, act, x1=x1)
        h0, h1 = ops.linear_dgrad(gy, W, x0, act, x1=x1, packed_t=pt)
        assert same(h0, g0, n0), (act, float((g0 - h0).abs().max()))
        assert g1 is None or same(h1, g1, n1), (act, float((g1 - h1).abs().max()))
        if n1:  # only one half requested
            o0, z1 = ops.linear_dgrad(gy, W, x0, act, x1=x1, want1=False, packed_t=pt)
            z0, o1 = ops.linear_dgrad(gy, W, x0, act, x1=x1, want0=False, packed_t=pt)
            assert z1 is None and z0 is None and torch.equal(o0, h0) and torch.equal(o1, h1)
        # against fp64, at the bar of the unpacked entry points
        f, df = ACTS[act]
        xin = (torch.cat([x0, x1], 1) if in1 else x0).double()
        rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
        assert rel(y_pk, f(xin) @ W.double().t() + b.double()) < 3e-5
        assert rel(torch.cat([h0, h1], 1) if in1 else h0, (gy.double() @ W.double()) * df(xin)) < 3e-5
    # the transposed entry is what packing the materialised transpose gives
    (pt_ref,) = ops.train_pack_many([(W.t().contiguous(), False)])
    assert torch.equal(pt, pt_ref)
    (p2_ref,) = ops.train_pack_many([(W2.t().contiguous(), False)])
    assert torch.equal(p2, p2_ref)


def test_pack_many_takes_more_than_one_launch_worth_of_matrices(ops):
    torch.manual_seed(3)
    mats = [(torch.randn(16 + 3 * i, 40 + i, device="cuda"), bool(i & 1)) for i in range(70)]  # > 32 entries: several launches
    packed = ops.train_pack_many(mats)
    for (w, t), p in zip(mats, packed):
        (ref,) = ops.train_pack_many([(w.t().contiguous(), False)] if t else [(w, False)])
        assert torch.equal(p, ref)


@pytest.mark.parametrize("shape", [(4096, 256, 0, 256), (4096 + 5, 256, 38, 256), (3000, 1000, 24, 72), (100, 256, 0, 65)],
                         ids=lambda s: "N%d_in%d+%d_out%d" % s)
def test_weight_gradient_accumulating_and_overwriting_entry_points(ops, shape):
    """na_linear_wgrad_bf16x3 ACCUMULATES into dW / db (its contract since round 1); na_linear_wgrad_bf16x3_ow (round 5, what
    ops.linear_wgrad calls: no zero fill) WRITES them -- onto whatever the buffer held."""
    import ctypes as C
    from nerf_atlas_amd import _lib
    lib = _lib.load()
    N, in0, in1, out = shape
    torch.manual_seed(5)
    x0 = torch.randn(N, in0, device="cuda")
    x1 = torch.randn(N, in1, device="cuda") if in1 else None
    gy = torch.randn(N, out, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    dW, db = ops.linear_wgrad(x0, gy, "sin", x1=x1, split_bf16=True)
    junk_W = torch.full((out, in0 + in1), float("nan"), device="cuda")
    junk_b = torch.full((out,), -3.0, device="cuda")
    _lib.check(lib.na_linear_wgrad_bf16x3_ow(ptr(x0), in0, ptr(x1), in1, N, ptr(gy), out, 2, ptr(junk_W), ptr(junk_b), st))
    ls = out <= 256 and in0 <= 256 and in1 <= 256 and N >= 2048
    close = (lambda a, b: torch.equal(a, b)) if ls else (lambda a, b: float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()))
    assert close(junk_W, dW) and close(junk_b, db)
    acc_W = torch.ones(out, in0 + in1, device="cuda")
    acc_b = torch.full((out,), 2.0, device="cuda")
    _lib.check(lib.na_linear_wgrad_bf16x3(ptr(x0), in0, ptr(x1), in1, N, ptr(gy), out, 2, ptr(acc_W), ptr(acc_b), st))
    assert close(acc_W, 1.0 + dW) and close(acc_b, 2.0 + db)


def test_training_forward_packs_every_linear_once_and_changes_nothing(ops, monkeypatch):
    """SkipConnMLP._forward_train packs all Linears of the network (W for the forward, W^T for the input gradient) with one launch;
    outputs and every gradient are bit for bit those of the per-call packing."""
    from nerf_atlas_amd.neural_blocks import SkipConnMLP, HashEncoder
    torch.manual_seed(11)
    m = SkipConnMLP(in_size=3, out=65, enc=HashEncoder(), num_layers=4, hidden_size=256).cuda()
    p = (torch.rand(4096 + 7, 3, device="cuda") * 2 - 1)
    w = torch.randn(4096 + 7, 65, device="cuda")

    def run():
        for q in m.parameters():
            q.grad = None
        (m(p) * w).sum().backward()
        # (the Linears' gradients are bit-reproducible; the hash tables' scatter adds with fp32 atomics in arrival order)
        return [q.grad.clone() for l in m._linears() for q in l.parameters()], [e.weight.grad.clone() for e in m.enc.embs]
    calls = []
    real = ops.train_pack_many
    monkeypatch.setattr(ops, "train_pack_many", lambda mats: (calls.append(len(mats)), real(mats))[1])
    got, got_t = run()
    assert calls == [12]   # 6 Linears x (W, W^T): ONE call = one launch
    monkeypatch.setattr(SkipConnMLP, "_train_packs", lambda self, init: [None] * len(self._linears()))
    ref, ref_t = run()
    # (the 256 -> 65 out Linear and the 38-column input gradients run the row-stream kernel with packed operands: last-place
    # differences in those rows and in what is back-propagated through them)
    assert len(got) == 12 and all(float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) for a, b in zip(got, ref))
    assert all(float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) for a, b in zip(got_t, ref_t))


# (N, in1, out): in0 = 256 always; ragged batches (partial last stage, slices of unequal length), the out Linears' 65 / 3 columns
# (dY fetched as dwords, 5 / 1 k steps, idle weight-gradient waves), a skip layer's second source left to the other kernels
BWD_SHAPES = [(8192, 0, 256), (8192 + 37, 0, 256), (65536 + 1, 0, 256), (12000, 38, 256), (8192 + 5, 0, 65), (9000, 0, 3),
              (8192, 0, 64), (262144, 0, 256), (16384 + 33, 0, 128)]


@pytest.mark.parametrize("shape", BWD_SHAPES, ids=lambda s: "N%d_in256+%d_out%d" % s)
@pytest.mark.parametrize("act", ["leaky_relu", "sin", "none"])
def test_fused_backward_vs_fp64_and_vs_the_two_launch_path(ops, shape, act):
    """csrc/train_bwd.hip (round 5): input gradient + weight gradient + bias gradient of a Linear in one pass over dY and x."""
    N, in1, out = shape
    in0 = 256
    if N > 100000 and act != "leaky_relu":
        pytest.skip("the full-size batch once")
    assert ops.linear_bwd_fused_ok(N, out, in0)
    torch.manual_seed(N + 7 * in1 + out)
    dev = "cuda"
    x0 = torch.randn(N, in0, device=dev)
    W = torch.randn(out, in0 + in1, device=dev) * (1.0 / (in0 + in1)) ** 0.5
    gy = torch.randn(N, out, device=dev)
    f, df = ACTS[act]
    (pt,) = ops.train_pack_many([(W, True)])
    g0, dW, db = ops.linear_bwd_fused(gy, x0, act, pt, in1=in1)
    torch.cuda.synchronize()
    # the two-launch path on the same operands
    x1 = torch.randn(N, in1, device=dev) if in1 else None
    h0, _ = ops.linear_dgrad(gy, W, x0, act, x1=x1, want1=False, packed_t=pt)
    dW2, db2 = ops.linear_wgrad(x0, gy, act, x1=x1, split_bf16=True)

    def rel(a, r):
        return float((a.double() - r.double()).abs().max() / r.double().abs().max())
    if N <= 100000:
        g_ref = (gy.double() @ W.double()[:, :in0]) * df(x0.double())
        dW_ref = gy.double().t() @ f(x0.double())
        assert rel(g0, g_ref) < 3e-5
        assert rel(dW[:, :in0], dW_ref) < 3e-5
        assert rel(db, gy.double().sum(0)) < 3e-5
    # same three products per k: the input gradient differs from the standalone kernel's by the order of the additions at most
    assert rel(g0, h0) < 2e-6
    assert rel(dW[:, :in0], dW2[:, :in0]) < 2e-6
    assert rel(db, db2) < 2e-6
    # run-to-run reproducible (fixed-order partial sums, no atomics)
    g0b, dWb, dbb = ops.linear_bwd_fused(gy, x0, act, pt, in1=in1)
    assert torch.equal(g0, g0b) and torch.equal(dW[:, :in0], dWb[:, :in0]) and torch.equal(db, dbb)


@pytest.mark.parametrize("in1", [0, 38, 69, 134, 200, 259])  # (134, 200: the 129..256-column second source, ADVICE r05 -- weight gradient by columns + packed input gradient with only g_x1 requested)
def test_linear_fn_backward_through_the_fused_kernel_vs_fp64(ops, in1):
    """autograd.LinearFn with packed operands: a 256 wide first source takes the one-pass backward, its narrow second source
    (a skip layer's [256 | 38], [256 | 69]) the narrow kernels; a second source wider than 256 (the Fourier SDF network's skip)
    keeps the two-launch path.  All four gradients against fp64 autograd."""
    from nerf_atlas_amd import autograd as ag
    N, in0, out = 9000 + in1, 256, 256
    torch.manual_seed(in1)
    dev = "cuda"
    x0 = torch.randn(N, in0, device=dev, requires_grad=True)
    x1 = torch.randn(N, in1, device=dev, requires_grad=True) if in1 else None
    W = (torch.randn(out, in0 + in1, device=dev) * (1.0 / (in0 + in1)) ** 0.5).requires_grad_()
    b = torch.randn(out, device=dev, requires_grad=True)
    gy = torch.randn(N, out, device=dev)
    pk = ops.train_pack_many([(W, False), (W, True)])
    y = ag.LinearFn.apply(x0, x1, W, b, "leaky_relu", (pk[0], pk[1]))
    y.backward(gy)
    xin = (torch.cat([x0, x1], 1) if in1 else x0).detach().double().requires_grad_()
    Wd, bd = W.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    (torch.nn.functional.leaky_relu(xin, 0.01) @ Wd.t() + bd).backward(gy.double())

    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    assert rel(x0.grad, xin.grad[:, :in0]) < 3e-5
    if in1:
        assert rel(x1.grad, xin.grad[:, in0:]) < 3e-5
    assert rel(W.grad, Wd.grad) < 3e-5
    assert rel(b.grad, bd.grad) < 3e-5


# (N, in0 of the narrow source, c0 = first row of the source in W^T = width of the source in front of it, out)
NARROW_BWD = [(8192 + 3, 38, 0, 256), (9000, 69, 0, 256), (8192, 38, 256, 256), (12000 + 1, 69, 256, 256), (8192, 64, 0, 256),
              (8200, 128, 0, 256), (9001, 100, 64, 128), (8192, 3, 256, 65)]


@pytest.mark.parametrize("shape", NARROW_BWD, ids=lambda s: "N%d_in%d_at%d_out%d" % s)
@pytest.mark.parametrize("act", ["leaky_relu", "sin", "none"])
def test_fused_backward_of_a_narrow_source_vs_fp64(ops, shape, act):
    """The one-pass backward on a narrow source (an init Linear's 38 / 69 columns; the second source of a skip layer, whose rows
    of W^T start at column group c0 / 64 of the packed stream and whose columns of dW start at c0): unaligned rows are fetched
    and stored as dwords, column tiles past the source idle."""
    N, in0, c0, out = shape
    assert ops.linear_bwd_fused_ok(N, out, in0)
    torch.manual_seed(N + in0 + c0 + out)
    dev = "cuda"
    x = torch.randn(N, in0, device=dev)
    W = torch.randn(out, c0 + in0, device=dev) * (1.0 / (c0 + in0)) ** 0.5
    gy = torch.randn(N, out, device=dev)
    f, df = ACTS[act]
    (pt,) = ops.train_pack_many([(W, True)])
    dW = torch.full((out, c0 + in0), 7.0, device=dev)
    g, dW_, _ = ops.linear_bwd_fused(gy, x, act, pt, dW=dW, col0=c0)
    assert dW_ is dW

    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    g_ref = (gy.double() @ W.double()[:, c0:]) * df(x.double())
    dW_ref = gy.double().t() @ f(x.double())
    assert rel(g, g_ref) < 3e-5
    assert rel(dW[:, c0:], dW_ref) < 3e-5
    assert bool((dW[:, :c0] == 7.0).all()), "columns of the other source are not touched"
    if c0 == 0:  # a single narrow source: fresh buffers, the bias gradient too
        g2, dW2, db2 = ops.linear_bwd_fused(gy, x, act, pt)
        assert torch.equal(g2, g) and torch.equal(dW2, dW)
        assert rel(db2, gy.double().sum(0)) < 3e-5


# (N, in1): 256 -> 256 with an optional narrow second source
FWD_SHAPES = [(8192, 0), (8192 + 37, 0), (65536 + 1, 0), (12000, 38), (9000 + 5, 69), (8192, 80), (8192 + 31, 3), (262144, 38)]


@pytest.mark.parametrize("shape", FWD_SHAPES, ids=lambda s: "N%d_in256+%d" % s)
@pytest.mark.parametrize("act", ["leaky_relu", "sin", "none"])
def test_register_resident_forward_vs_fp64_and_vs_the_streaming_kernel(ops, shape, act):
    """csrc/train_fwd.hip (round 5): the forward of a 256 wide Linear (hidden layers, skip layers [256 | 38], [256 | 69] in ONE
    pass) with W resident in registers, through na_linear_bf16x3_pk."""
    N, in1 = shape
    if N > 100000 and act != "leaky_relu":
        pytest.skip("the full-size batch once")
    torch.manual_seed(N + in1)
    dev = "cuda"
    x0 = torch.randn(N, 256, device=dev)
    x1 = torch.randn(N, in1, device=dev) if in1 else None
    W = torch.randn(256, 256 + in1, device=dev) * (1.0 / (256 + in1)) ** 0.5
    b = torch.randn(256, device=dev)
    f, _ = ACTS[act]
    (pf,) = ops.train_pack_many([(W, False)])
    y = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True, packed=pf)
    y_stream = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True)   # (unpacked entry point: lsnt::kernel<0>)
    y_nobias = ops.linear_f32(x0, W, None, pre_act=act, x1=x1, split_bf16=True, packed=pf)

    def rel(a, r):
        return float((a.double() - r.double()).abs().max() / r.double().abs().max())
    if N <= 100000:
        xin = (torch.cat([x0, x1], 1) if in1 else x0).double()
        assert rel(y, f(xin) @ W.double().t() + b.double()) < 3e-5
    assert rel(y, y_stream) < 2e-6
    assert float((y_nobias + b - y).abs().max()) <= 1e-5
    assert torch.equal(y, ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True, packed=pf))


def test_register_resident_forward_with_sin_in_its_own_process():
    """lsfw::kernel with the sin activation against fp64 in a fresh process (the first version of the kernel split the output columns
    over two workgroups and was not the default for sin; since the full-width version it is: this keeps a process-level check that
    the environment switch parses -- NA_TRAIN_FUSED_FWD=all is simply "on")."""
    import os
    import subprocess
    import sys
    code = r'''
import torch, sys
sys.path.insert(0, %r)
from nerf_atlas_amd import ops
torch.manual_seed(5)
for N, in1 in ((8192 + 7, 0), (9000, 38), (8192, 69)):
    x0 = torch.randn(N, 256, device="cuda"); x1 = torch.randn(N, in1, device="cuda") if in1 else None
    W = torch.randn(256, 256 + in1, device="cuda") / 17; b = torch.randn(256, device="cuda")
    (pf,) = ops.train_pack_many([(W, False)])
    y = ops.linear_f32(x0, W, b, pre_act="sin", x1=x1, split_bf16=True, packed=pf)
    xin = (torch.cat([x0, x1], 1) if in1 else x0).double()
    ref = torch.sin(xin) @ W.double().t() + b.double()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err < 3e-5, (N, in1, err)
print("OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NA_TRAIN_FUSED_FWD="all")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_whole_network_node_matches_the_per_layer_nodes(ops, monkeypatch):
    """autograd.MlpTrainFn (one node per SkipConnMLP, partial gradients of all Linears reduced by ONE launch) against the chain of
    LinearFn nodes: same forward bits, gradients equal to the last place (same kernels, same order of the partial sums)."""
    from nerf_atlas_amd.neural_blocks import SkipConnMLP, HashEncoder
    torch.manual_seed(11)
    m = SkipConnMLP(in_size=3, out=65, latent_size=0, enc=HashEncoder(input_dims=3), num_layers=5, hidden_size=256).cuda()
    p = (torch.rand(9000, 3, device="cuda") * 2 - 1)
    gy = torch.randn(9000, 65, device="cuda")
    res = {}
    for mode in ("node", "layers"):
        if mode == "layers":
            monkeypatch.setattr(SkipConnMLP, "_mlp_fn_ok", lambda self, init, packs: False)
        m.zero_grad(set_to_none=True)
        with torch.enable_grad():
            y = m(p)
            y.backward(gy)
        res[mode] = (y.detach().clone(), [q.grad.clone() for q in m.parameters()])
    assert torch.equal(res["node"][0], res["layers"][0])
    for a, b in zip(res["node"][1], res["layers"][1]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12


@pytest.mark.parametrize("shape", [(8192 + 3, 38), (9000, 69), (8192, 64), (12000 + 1, 80), (8192, 3)], ids=lambda s: "N%d_in%d" % s)
@pytest.mark.parametrize("act", ["none", "leaky_relu", "sin"])
def test_register_resident_forward_of_a_narrow_source_alone(ops, shape, act):
    """lsfw::kernel MODE 2: an init Linear (38 / 69 -> 256): the narrow source alone, unaligned rows as dwords, 3-5 k steps."""
    N, in0 = shape
    torch.manual_seed(N + in0)
    x = torch.randn(N, in0, device="cuda")
    W = torch.randn(256, in0, device="cuda") * (1.0 / in0) ** 0.5
    b = torch.randn(256, device="cuda")
    f, _ = ACTS[act]
    (pf,) = ops.train_pack_many([(W, False)])
    y = ops.linear_f32(x, W, b, pre_act=act, split_bf16=True, packed=pf)
    ref = f(x.double()) @ W.double().t() + b.double()
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 3e-5
    y_stream = ops.linear_f32(x, W, b, pre_act=act, split_bf16=True)
    assert float((y - y_stream).abs().max()) <= 2e-6 * float(y_stream.abs().max())


@pytest.mark.parametrize("shape", [(9000, 65), (8192 + 1, 3), (12000, 128), (8200, 100), (8192, 32)], ids=lambda s: "N%d_out%d" % s)
@pytest.mark.parametrize("act", ["leaky_relu", "sin"])
def test_register_resident_forward_with_a_narrow_output(ops, shape, act):
    """lsfw::kernel NOUT: the out Linears (256 -> 65 / 3): column tiles past the output idle, the tile leaves as dwords."""
    N, out = shape
    torch.manual_seed(N + out)
    x = torch.randn(N, 256, device="cuda")
    W = torch.randn(out, 256, device="cuda") / 16
    b = torch.randn(out, device="cuda")
    f, _ = ACTS[act]
    (pf,) = ops.train_pack_many([(W, False)])
    y = ops.linear_f32(x, W, b, pre_act=act, split_bf16=True, packed=pf)
    ref = f(x.double()) @ W.double().t() + b.double()
    assert y.shape == (N, out) and float((y.double() - ref).abs().max() / ref.abs().max()) < 3e-5
    y0 = ops.linear_f32(x, W, None, pre_act=act, split_bf16=True, packed=pf)
    assert float((y0 + b - y).abs().max()) <= 1e-5


@pytest.mark.parametrize("shape", [(9000, 38, 256, "none"), (8192 + 7, 69, 256, "sin"), (8200, 128, 65, "leaky_relu"), (8192, 64, 256, "leaky_relu")],
                         ids=lambda s: "N%d_in%d_out%d_%s" % s)
def test_one_pass_backward_adds_another_gradient_of_the_same_tensor(ops, shape):
    """g_add of na_linear_bwd_partials_bf16x3_pk: g_x = (dY . W) * act'(x) + add, the bits of the separate torch add."""
    N, in0, out, act = shape
    torch.manual_seed(N + in0)
    x = torch.randn(N, in0, device="cuda"); W = torch.randn(out, in0, device="cuda") / in0 ** 0.5
    gy = torch.randn(N, out, device="cuda"); add = torch.randn(N, in0, device="cuda")
    (pt,) = ops.train_pack_many([(W, True)])
    g_plain, ws0, n0 = ops.linear_bwd_partials(gy, x, act, pt)
    g_sum, ws1, n1 = ops.linear_bwd_partials(gy, x, act, pt, add=add)
    assert n0 == n1   # (the partial gradients do not see the addend: checked through their reduction below -- rows past `out` of a workspace are never written)
    assert torch.equal(g_sum, g_plain + add)
    dW = torch.empty(out, in0, device="cuda"); db = torch.empty(out, device="cuda")
    ops.train_reduce_many([(ws1, n1, out, in0, dW, 0, db)])
    _, dW_ref, db_ref = ops.linear_bwd_fused(gy, x, act, pt)
    assert torch.equal(dW, dW_ref) and torch.equal(db, db_ref)


def test_library_scratch_grows_and_retires_buffers_without_changing_results(ops):
    """the per-(device, stream) scratch behind workspace = NULL grows geometrically and frees a replaced buffer once the
    work queued before its replacement has finished (it used to leak every earlier buffer).  A batch that grows call by call, on two
    streams, with launches still in flight when the buffer is replaced: every result equals the caller-workspace call bit for bit."""
    from nerf_atlas_amd import _lib
    lib = _lib.load()
    torch.manual_seed(5)
    W = torch.randn(256, 256, device="cuda") / 16
    (pt,) = ops.train_pack_many([(W, True)])
    side = torch.cuda.Stream()
    for rep in range(2):
        for N in (8192, 40000, 70000, 300000, 20000, 600000):
            x = torch.randn(N, 256, device="cuda")
            gy = torch.randn(N, 256, device="cuda")
            g_ref, dW_ref, db_ref = ops.linear_bwd_fused(gy, x, "leaky_relu", pt)   # (torch's allocator provides the workspace)
            for stream in (torch.cuda.current_stream(), side):
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):
                    g = torch.empty_like(x)
                    acc = torch.empty(256 * 256 + 256, device="cuda")
                    dW, db = acc[:256 * 256].view(256, 256), acc[256 * 256:]
                    for _ in range(3):   # back to back: the next call may replace the buffer while this one's kernels are queued
                        _lib.check(lib.na_linear_bwd_bf16x3_pk(gy.data_ptr(), 256, N, pt.data_ptr(), x.data_ptr(), 256, ops.ACT["leaky_relu"],
                                                               g.data_ptr(), dW.data_ptr(), 256, db.data_ptr(), None, stream.cuda_stream))
                stream.synchronize()
                assert torch.equal(g, g_ref) and torch.equal(dW, dW_ref) and torch.equal(db, db_ref), (rep, N)
