"""GPU parity of the TRAINING step (second slice; `python -m pytest tests -m gpu`).

  * every operator of csrc/yf_train_kernels.hip, through the C ABI (`yf_train_*`), against the same torch operator evaluated in
    float64 on the CPU, on every layer geometry the network has (tolerances are fp32 rounding of sums of that length);
  * the whole iteration of src/model_training/train.py:111-132 -- model.train(), forward on batch statistics, the two-head loss,
    backward, Adam -- against the reference's own run of it (tests/golden/golden_train_256.npz, made by make_golden.py main_train):
    heads, losses, all 256 parameter gradients, and after two iterations the parameters and the BatchNorm running statistics.
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights", "yolo_fastest_256x320_epoch28.pth")


@pytest.fixture(scope="module")
def yf():
    import yolo_fastest_amd
    return yolo_fastest_amd


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(yf, dev):
    from yolo_fastest_amd import training
    return training._Ops(dev)


def _g(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def _close(got, want, rel, what):
    got = got.cpu().numpy().astype(np.float64)
    want = want.detach().numpy().astype(np.float64)
    assert got.shape == want.shape, what
    err = np.abs(got - want).max()
    assert err <= rel * max(np.abs(want).max(), 1e-30), (what, err, np.abs(want).max())


# (Cin, Cout, k, stride, depthwise, H, W): every conv geometry of yolo_fastest.py:78-146, on small maps (odd sizes too: the
# reference's padding arithmetic, not only the even sizes the network sees)
CONVS = [(1, 8, 3, 2, 0, 16, 12), (24, 24, 3, 2, 0, 10, 8), (8, 8, 3, 1, 1, 9, 7), (32, 32, 3, 2, 1, 12, 10), (136, 136, 3, 2, 1, 6, 8),
         (96, 96, 5, 1, 1, 7, 9), (128, 128, 5, 1, 1, 4, 5), (4, 24, 1, 1, 0, 6, 5), (232, 96, 1, 1, 0, 4, 6), (224, 48, 1, 1, 0, 3, 4),
         (24, 136, 1, 1, 0, 5, 5), (3, 5, 3, 2, 0, 7, 9), (48, 48, 3, 1, 1, 8, 12), (96, 96, 5, 1, 1, 8, 12), (32, 32, 3, 2, 1, 16, 16),
         (16, 96, 1, 1, 0, 8, 10), (24, 24, 3, 2, 0, 16, 24),
         # io_params other than the shipped ones (yolo_fastest.py:72-78,138,148): conv0 on 3 input channels, heads of A * (5 + C) channels
         (3, 8, 3, 2, 0, 64, 96), (3, 8, 3, 2, 0, 18, 14), (128, 30, 1, 1, 0, 2, 3), (96, 30, 1, 1, 0, 4, 6), (96, 18, 1, 1, 0, 4, 6),
         (128, 75, 1, 1, 0, 8, 10), (96, 255, 1, 1, 0, 16, 20), (128, 16, 1, 1, 0, 2, 3)]


@pytest.mark.parametrize("geom", CONVS)
def test_conv_forward_backward_match_torch(ops, dev, geom):
    Cin, Cout, k, stride, dw, H, W = geom
    rng = np.random.default_rng(sum(geom))
    N = 3
    x = rng.normal(size=(N, Cin, H, W)).astype(np.float32)
    w = rng.normal(size=(Cout, 1 if dw else Cin, k, k)).astype(np.float32)
    b = rng.normal(size=(Cout,)).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    bt = torch.from_numpy(b).double().requires_grad_(True)
    yt = F.conv2d(xt, wt, bt, stride=stride, padding=(k - 1) // 2, groups=Cin if dw else 1)
    gy = rng.normal(size=tuple(yt.shape)).astype(np.float32)
    yt.backward(torch.from_numpy(gy).double())
    xd, wd, bd, gyd = _g(x, dev), _g(w, dev), _g(b, dev), _g(gy, dev)
    y = torch.empty(tuple(yt.shape), device=dev)
    ops.call("yf_train_conv_forward", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), N, Cin, H, W, Cout, k, stride, dw)
    _close(y, yt, 2e-6, "forward")
    y0 = torch.empty_like(y)
    ops.call("yf_train_conv_forward", xd.data_ptr(), wd.data_ptr(), None, y0.data_ptr(), N, Cin, H, W, Cout, k, stride, dw)
    _close(y0, yt - bt.detach()[None, :, None, None], 2e-6, "forward without bias")
    gx = torch.full_like(xd, float("nan"))
    ops.call("yf_train_conv_backward_data", gyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout, k, stride, dw)
    _close(gx, xt.grad, 2e-6, "backward data")
    gw = torch.full_like(wd, float("nan"))
    ops.call("yf_train_conv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), N, Cin, H, W, Cout, k, stride, dw, ops.scratch,
             ops.scratch_bytes)
    _close(gw, wt.grad, 5e-6, "backward weight")
    gw2 = torch.full_like(wd, float("nan"))                       # without scratch: no split (or the atomics fallback)
    ops.call("yf_train_conv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw2.data_ptr(), N, Cin, H, W, Cout, k, stride, dw, None, 0)
    _close(gw2, wt.grad, 5e-6, "backward weight, no scratch")
    gb = torch.full_like(bd, float("nan"))
    ops.call("yf_train_channel_sum", gyd.data_ptr(), gb.data_ptr(), N, Cout, y.shape[2] * y.shape[3])
    _close(gb, bt.grad, 2e-6, "bias gradient")


# the bandwidth kernels of the large maps (float4 pointwise GEMM, multi-wave weight gradient, row-block depthwise kernels) only take
# over above a size the small geometries above never reach: (N, Cin, Cout, k, stride, depthwise, H, W)
BIG = [(64, 8, 32, 1, 1, 0, 64, 80), (40, 48, 8, 1, 1, 0, 64, 80), (24, 16, 96, 1, 1, 0, 32, 80), (6, 8, 8, 3, 1, 1, 64, 160), (5, 16, 16, 5, 1, 1, 32, 80),
       (3, 8, 8, 3, 1, 1, 36, 72), (12, 24, 24, 3, 2, 0, 64, 80), (32, 1, 8, 3, 2, 0, 64, 160),
       (64, 136, 24, 1, 1, 0, 16, 32), (40, 232, 96, 1, 1, 0, 16, 52), (300, 96, 96, 5, 1, 1, 16, 20),     # weight-stationary GEMM; small planes, many
       (16, 3, 8, 3, 2, 0, 256, 320), (64, 96, 30, 1, 1, 0, 16, 20), (64, 128, 255, 1, 1, 0, 8, 10), (256, 96, 75, 1, 1, 0, 16, 20)]   # RGB conv0, other heads


@pytest.mark.parametrize("geom", BIG)
def test_large_map_kernels_match_torch(ops, dev, geom):
    N, Cin, Cout, k, stride, dw, H, W = geom
    rng = np.random.default_rng(sum(geom))
    x = rng.normal(size=(N, Cin, H, W)).astype(np.float32)
    w = rng.normal(size=(Cout, 1 if dw else Cin, k, k)).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    yt = F.conv2d(xt, wt, None, stride=stride, padding=(k - 1) // 2, groups=Cin if dw else 1)
    gy = rng.normal(size=tuple(yt.shape)).astype(np.float32)
    yt.backward(torch.from_numpy(gy).double())
    xd, wd, gyd = _g(x, dev), _g(w, dev), _g(gy, dev)
    y = torch.full(tuple(yt.shape), float("nan"), device=dev)
    ops.call("yf_train_conv_forward", xd.data_ptr(), wd.data_ptr(), None, y.data_ptr(), N, Cin, H, W, Cout, k, stride, dw)
    _close(y, yt, 3e-6, "forward")
    gx = torch.full_like(xd, float("nan"))
    ops.call("yf_train_conv_backward_data", gyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout, k, stride, dw)
    _close(gx, xt.grad, 3e-6, "backward data")
    gw = torch.full_like(wd, float("nan"))
    ops.call("yf_train_conv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), N, Cin, H, W, Cout, k, stride, dw, ops.scratch,
             ops.scratch_bytes)
    _close(gw, wt.grad, 3e-5, "backward weight")
    gw2 = torch.empty_like(gw)
    ops.call("yf_train_conv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw2.data_ptr(), N, Cin, H, W, Cout, k, stride, dw, ops.scratch,
             ops.scratch_bytes)
    assert torch.equal(gw, gw2)                                   # fixed summation order
    if not dw and k == 1:                                         # the split bias-gradient sum of the heads, at a size where it splits
        gb = torch.full((Cout,), float("nan"), device=dev)
        ops.call("yf_train_channel_sum_split", gyd.data_ptr(), gb.data_ptr(), N, Cout, H * W, ops.scratch)
        _close(gb, torch.from_numpy(gy).double().sum((0, 2, 3)), 2e-6, "bias gradient, split")


def test_conv_backward_weight_long_reduction(ops, dev):
    """The chunked reduction (several workgroups per weight element + atomics) at a reduction length of the real batch."""
    rng = np.random.default_rng(5)
    N, Cin, Cout, H, W = 16, 8, 32, 64, 80
    x = rng.normal(size=(N, Cin, H, W)).astype(np.float32)
    gy = rng.normal(size=(N, Cout, H, W)).astype(np.float32)
    want = torch.einsum("nchw,nohw->oc", torch.from_numpy(x).double(), torch.from_numpy(gy).double())[:, :, None, None]
    xd, gyd = _g(x, dev), _g(gy, dev)
    gw = torch.full((Cout, Cin, 1, 1), float("nan"), device=dev)
    ops.call("yf_train_conv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, ops.scratch, ops.scratch_bytes)
    _close(gw, want, 2e-5, "backward weight, 81920-long sums")
    gw2 = torch.empty_like(gw)
    ops.call("yf_train_conv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw2.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, ops.scratch, ops.scratch_bytes)
    assert torch.equal(gw, gw2)                                   # fixed summation order: bit-reproducible
    for k, dw, C2 in ((3, 1, 32), (5, 1, 32)):                    # depthwise, many chunks
        w_ = torch.zeros(C2, 1, k, k, dtype=torch.float64, requires_grad=True)
        xt = torch.from_numpy(rng.normal(size=(N, C2, H, W)).astype(np.float32))
        gt_ = torch.from_numpy(rng.normal(size=(N, C2, H, W)).astype(np.float32))
        F.conv2d(xt.double(), w_, None, padding=(k - 1) // 2, groups=C2).backward(gt_.double())
        g, xg, gg = torch.empty(C2, 1, k, k, device=dev), xt.to(dev), gt_.to(dev)
        ops.call("yf_train_conv_backward_weight", xg.data_ptr(), gg.data_ptr(), g.data_ptr(), N, C2, H, W, C2, k, 1, 1, ops.scratch, ops.scratch_bytes)
        _close(g, w_.grad, 2e-5, "depthwise backward weight, long sums")


def test_deconv_forward_backward_match_torch(ops, dev):
    rng = np.random.default_rng(9)
    for N, Cin, Cout, H, W in ((2, 96, 96, 4, 5), (3, 5, 7, 3, 3), (16, 96, 96, 8, 10), (5, 20, 70, 6, 6), (64, 12, 20, 12, 12), (120, 96, 96, 8, 10)):
        x = rng.normal(size=(N, Cin, H, W)).astype(np.float32)
        w = rng.normal(size=(Cin, Cout, 2, 2)).astype(np.float32)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        wt = torch.from_numpy(w).double().requires_grad_(True)
        yt = F.conv_transpose2d(xt, wt, stride=2)
        gy = rng.normal(size=tuple(yt.shape)).astype(np.float32)
        yt.backward(torch.from_numpy(gy).double())
        xd, wd, gyd = _g(x, dev), _g(w, dev), _g(gy, dev)
        y = torch.empty(tuple(yt.shape), device=dev)
        ops.call("yf_train_deconv_forward", xd.data_ptr(), wd.data_ptr(), y.data_ptr(), N, Cin, H, W, Cout)
        _close(y, yt, 2e-6, "forward")
        gx = torch.full_like(xd, float("nan"))
        ops.call("yf_train_deconv_backward_data", gyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout)
        _close(gx, xt.grad, 2e-6, "backward data")
        gw = torch.full_like(wd, float("nan"))
        ops.call("yf_train_deconv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), N, Cin, H, W, Cout, ops.scratch, ops.scratch_bytes)
        _close(gw, wt.grad, 5e-6, "backward weight")


@pytest.mark.parametrize("relu", [0, 1])
def test_batchnorm_train_mode_matches_torch(ops, dev, relu):
    rng = np.random.default_rng(11 + relu)
    for N, C, H, W in ((4, 8, 9, 7), (16, 136, 4, 5), (2, 3, 1, 1), (16, 8, 64, 80), (3, 232, 20, 16), (40, 5, 30, 30), (16, 24, 40, 32)):
        x = (rng.normal(size=(N, C, H, W)) * rng.uniform(0.5, 3, (1, C, 1, 1)) + rng.normal(size=(1, C, 1, 1))).astype(np.float32)
        gamma, beta = rng.normal(1, 0.3, C).astype(np.float32), rng.normal(0, 0.5, C).astype(np.float32)
        rm, rv = rng.normal(size=C).astype(np.float32), rng.uniform(0.5, 2, C).astype(np.float32)
        bn = torch.nn.BatchNorm2d(C).double().train()
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy(gamma)); bn.bias.copy_(torch.from_numpy(beta))
            bn.running_mean.copy_(torch.from_numpy(rm)); bn.running_var.copy_(torch.from_numpy(rv))
        xt = torch.from_numpy(x).double().requires_grad_(True)
        yt = bn(xt)
        if relu:
            yt = F.relu(yt)
        gy = rng.normal(size=x.shape).astype(np.float32)
        yt.backward(torch.from_numpy(gy).double())
        xd, gd, bd, rmd, rvd, gyd = _g(x, dev), _g(gamma, dev), _g(beta, dev), _g(rm, dev), _g(rv, dev), _g(gy, dev)
        stats, y = torch.empty(2 * C, device=dev), torch.empty_like(xd)
        ops.call("yf_train_bn_forward", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), stats.data_ptr(),
                 y.data_ptr(), N, C, H * W, relu, ops.bn_scratch)
        _close(y, yt, 3e-6, "forward")
        _close(rmd, bn.running_mean, 1e-6, "running_mean")
        _close(rvd, bn.running_var, 1e-6, "running_var (unbiased)")
        dg, db, gx = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.full_like(xd, float("nan"))
        y_keep = y.clone()
        y.fill_(float("nan"))                                          # the backward must not need the forward's output
        ops.call("yf_train_bn_backward", xd.data_ptr(), gyd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), dg.data_ptr(),
                 db.data_ptr(), gx.data_ptr(), N, C, H * W, relu, ops.bn_scratch)
        _close(dg, bn.weight.grad, 5e-6, "dgamma")
        _close(db, bn.bias.grad, 5e-6, "dbeta")
        if N * H * W > 2:        # with two samples per channel xhat = +-1 and dx is a difference of nearly equal numbers
            _close(gx, xt.grad, 2e-5, "dx")
        # running statistics are optional
        y2 = torch.empty_like(xd)
        ops.call("yf_train_bn_forward", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), None, None, stats.data_ptr(), y2.data_ptr(), N, C, H * W, relu,
                 ops.bn_scratch)
        assert torch.equal(y_keep, y2)


def test_add_slice_are_exact(ops, dev):
    rng = np.random.default_rng(3)
    a, b = _g(rng.normal(size=(3, 7, 5, 4)), dev), _g(rng.normal(size=(3, 7, 5, 4)), dev)
    out = torch.empty_like(a)
    ops.call("yf_train_add", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel())
    assert torch.equal(out, a + b)
    a2 = a.clone()
    ops.call("yf_train_add", a2.data_ptr(), b.data_ptr(), a2.data_ptr(), a.numel())            # in place
    assert torch.equal(a2, a + b)
    c = _g(rng.normal(size=(3, 4, 5, 4)), dev)
    cat = torch.full((3, 11, 5, 4), float("nan"), device=dev)
    ops.call("yf_train_channel_slice", a.data_ptr(), cat.data_ptr(), 3, 7, 20, 7, 0, 11, 0)
    ops.call("yf_train_channel_slice", c.data_ptr(), cat.data_ptr(), 3, 4, 20, 4, 0, 11, 7)
    assert torch.equal(cat, torch.cat((a, c), 1))
    back = torch.empty_like(c)
    ops.call("yf_train_channel_slice", cat.data_ptr(), back.data_ptr(), 3, 4, 20, 11, 7, 4, 0)
    assert torch.equal(back, c)


def test_adam_matches_torch(yf, dev):
    """training.Adam against torch.optim.Adam (train.py:84's optimizer) on the CPU: five steps, a learning-rate edit in between
    (train.py:106-109), gradients spanning 1e-9 .. 10 so that the eps term matters."""
    from yolo_fastest_amd import training
    rng = np.random.default_rng(2)
    p0 = rng.normal(size=(37, 11)).astype(np.float32)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    our = torch.nn.Parameter(torch.from_numpy(p0.copy()).to(dev))
    o_ref = torch.optim.Adam([ref], lr=0.001, betas=(0.9, 0.999), eps=1e-08)
    o_our = training.Adam([our], lr=0.001, betas=(0.9, 0.999), eps=1e-08)
    for it in range(5):
        g = (rng.normal(size=p0.shape) * 10.0 ** rng.uniform(-9, 1, p0.shape)).astype(np.float32)
        ref.grad = torch.from_numpy(g.copy())
        our.grad = torch.from_numpy(g.copy()).to(dev)
        if it == 3:
            for grp in o_ref.param_groups + o_our.param_groups:
                grp["lr"] = 0.00037
        o_ref.step(); o_our.step()
        assert np.abs(our.detach().cpu().numpy() - ref.detach().numpy()).max() <= 2e-7 * (it + 1), it
    st = o_our.state[our]
    assert st["step"] == 5
    assert np.allclose(st["exp_avg"].cpu().numpy(), o_ref.state[ref]["exp_avg"].numpy(), rtol=1e-5, atol=1e-12)
    assert np.allclose(st["exp_avg_sq"].cpu().numpy(), o_ref.state[ref]["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-20)
    with pytest.raises(RuntimeError):
        cpu = torch.nn.Parameter(torch.zeros(3)); cpu.grad = torch.ones(3)
        training.Adam([cpu]).step()
    # a tensor that joins later (no gradient in the first step): its own step count, its own launch
    a0, b0 = rng.normal(size=(5, 3)).astype(np.float32), rng.normal(size=(7,)).astype(np.float32)
    ra, rb = torch.nn.Parameter(torch.from_numpy(a0.copy())), torch.nn.Parameter(torch.from_numpy(b0.copy()))
    oa, ob = torch.nn.Parameter(torch.from_numpy(a0.copy()).to(dev)), torch.nn.Parameter(torch.from_numpy(b0.copy()).to(dev))
    o_ref, o_our = torch.optim.Adam([ra, rb], lr=0.01), training.Adam([oa, ob], lr=0.01)
    for it in range(3):
        ga, gb = rng.normal(size=a0.shape).astype(np.float32), rng.normal(size=b0.shape).astype(np.float32)
        ra.grad, oa.grad = torch.from_numpy(ga.copy()), torch.from_numpy(ga.copy()).to(dev)
        if it > 0:
            rb.grad, ob.grad = torch.from_numpy(gb.copy()), torch.from_numpy(gb.copy()).to(dev)
        o_ref.step(); o_our.step()
        assert np.abs(oa.detach().cpu().numpy() - ra.detach().numpy()).max() <= 1e-6 and np.abs(ob.detach().cpu().numpy() - rb.detach().numpy()).max() <= 1e-6
    assert o_our.state[oa]["step"] == 3 and o_our.state[ob]["step"] == 2


GRAD_RATIO = 1.5      # median and 90th percentile (over the parameter tensors) of our distance to the exact gradient, over the reference's
GRAD_CAP = 8e-2        # no single tensor further than this from exact (relative to its largest element); the reference's worst: 1.2e-2


def _split(flat, sizes):
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [flat[off[i]:off[i + 1]] for i in range(len(sizes))]


def _setup_step(yf, golden, dev):
    from yolo_fastest_amd import training, validation as val
    gt = golden("golden_train_256")
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev)
    m.load_state_dict(torch.load(WEIGHTS, map_location=dev))
    m.train()
    x = ((torch.from_numpy(gt["input_u8"].astype(np.float32))[:, None] - 128.0) / 255.0).to(dev)
    targets = torch.from_numpy(gt["targets"]).to(dev)
    crit = [val.YOLOLossV3(io["anchors"][i], io["num_cls"], io["input_shape"], dev, model=m) for i in range(2)]
    opt = training.Adam(m.parameters(), lr=0.001, betas=(0.9, 0.999), eps=1e-08)          # train.py:84, _config.py:43
    return gt, m, x, targets, crit, opt


def _iteration(m, crit, x, targets):
    """train.py:114-131 up to and including loss.backward()"""
    pred = m(x)
    losses = [[] for _ in range(7)]
    for i, item_pred in enumerate(pred):
        for j, v in enumerate(crit[i](item_pred, targets)):
            losses[j].append(v)
    losses = [sum(v) for v in losses]
    losses[0].backward()
    return pred, np.array([float(v.detach()) if torch.is_tensor(v) else float(v) for v in losses])


def test_training_step_matches_the_reference(yf, golden, dev, capsys):
    """Two iterations of train.py:111-132 at the reference's batch size (16 frames, 256x320) from the shipped checkpoint, against the
    reference's own run.  Both sides are fp32 with different summation orders, so this is a tolerance test:
      heads 1e-4 of the logit range, losses 1e-4 relative, in both iterations.
      Gradients: measured against the EXACT backward (the fp32 head gradients pushed back through the network in float64,
      grads_1_exact), the reference's own fp32 gradients are off by 1.4e-3 (median over the tensors, relative to each tensor's largest
      element) up to 1.2e-2 -- not rounding of sums but ReLU decisions: of ~30 M pre-activations a few dozen lie within fp32 rounding of
      zero and each flips one mask element on one side (tools/train_oracle_report.py shows one: a step of 1e-1 at the layer, 1e-2 in
      everything upstream, in torch's fp32 as in ours).  Which ones flip is chance, so the test is on the distribution: our median and
      90th percentile within 1.5x of the reference's, ours the closer one for at least 35 % of the tensors (measured: 47 %), no tensor
      beyond 8e-2.  The flip-free check of the same backward is test_training_network_against_the_fp64_oracle and the per-operator tests.
      23 BatchNorm biases have a gradient that is ZERO in exact arithmetic (their output only feeds convolutions followed by train-mode
      BatchNorm; 1e-16 in the reference's float64 run): only the magnitude is checked (<= 1e-4; the reference's fp32: 1e-6).
      Optimizer: Adam divides by the gradient's magnitude, so relative gradient noise IS step noise (and the 23 zero-gradient biases
      take +-lr steps of random sign, in the reference too).  To compare the update itself, the reference's gradients are put in
      p.grad before optimizer.step() ("teacher forcing"); then iteration 2 starts from the reference's parameters, its heads and
      losses are tight again, and the parameters after two steps agree to 2e-7.  test_training_free_running_two_steps is the same
      without forcing."""
    gt, m, x, targets, crit, opt = _setup_step(yf, golden, dev)
    names = [n for n, _ in m.named_parameters()]
    assert names == [str(s) for s in gt["param_names"]]
    sizes = gt["param_sizes"]
    zero = gt["grad_absmax_f64"] < 1e-9
    assert zero.sum() == 23
    report = []
    for it in (1, 2):
        opt.zero_grad()
        pred, got = _iteration(m, crit, x, targets)
        for h, name in zip(pred, ("head_large", "head_small")):
            want = gt["%s_%d" % (name, it)]
            err = np.abs(h.detach().cpu().numpy() - want).max()
            report.append("iteration %d %s: max |dlogit| %.2e of %.1f" % (it, name, err, np.abs(want).max()))
            assert err <= 1e-4 * np.abs(want).max(), (it, name, err)
        want = gt["losses_%d" % it]
        report.append("iteration %d losses %s vs %s" % (it, np.round(got, 6).tolist(), np.round(want, 6).tolist()))
        assert np.allclose(got, want, rtol=1e-4), (it, got, want)
        ours, theirs, vs_ref = [], [], []
        exacts = _split(gt["grads_1_exact"], sizes) if it == 1 else [None] * len(sizes)
        for p, name, ref32, exact, z in zip(m.parameters(), names, _split(gt["grads_%d" % it], sizes), exacts, zero):
            g = p.grad.detach().cpu().numpy().ravel()
            assert np.isfinite(g).all(), name
            if z:
                assert np.abs(g).max() <= 1e-4, (name, np.abs(g).max())
                continue
            scale = np.abs(ref32).max()
            vs_ref.append(np.abs(g - ref32).max() / scale)
            assert vs_ref[-1] <= GRAD_CAP, (it, name, vs_ref[-1])
            if exact is not None:
                scale = np.abs(exact).max()
                ours.append(np.abs(g - exact).max() / scale)
                theirs.append(np.abs(ref32 - exact).max() / scale)
                assert ours[-1] <= GRAD_CAP, (name, ours[-1], theirs[-1])
        vs_ref = np.array(vs_ref)
        report.append("iteration %d gradients vs the reference's, per tensor relative to its largest element: median %.2e / 90%% %.2e / max %.2e"
                      % (it, np.median(vs_ref), np.quantile(vs_ref, 0.9), vs_ref.max()))
        assert np.median(vs_ref) <= 5e-3 and np.quantile(vs_ref, 0.9) <= 1.5e-2
        if it == 1:
            ours, theirs = np.array(ours), np.array(theirs)
            report.append("  vs the exact backward of the same head gradients: ours median %.2e / 90%% %.2e / max %.2e; the reference's fp32 median "
                          "%.2e / 90%% %.2e / max %.2e; ours is the closer one in %d of %d"
                          % (np.median(ours), np.quantile(ours, 0.9), ours.max(), np.median(theirs), np.quantile(theirs, 0.9), theirs.max(),
                             (ours < theirs).sum(), ours.size))
            assert np.median(ours) <= GRAD_RATIO * np.median(theirs) and np.quantile(ours, 0.9) <= GRAD_RATIO * np.quantile(theirs, 0.9)
            assert (ours < theirs).sum() >= 0.35 * ours.size
        with torch.no_grad():                                                  # teacher forcing: the reference's gradients
            for p, ref32 in zip(m.parameters(), _split(gt["grads_%d" % it], sizes)):
                p.grad.copy_(torch.from_numpy(ref32.reshape(tuple(p.shape))))
        opt.step()

    worst = 0.0
    for p, name, want in zip(m.parameters(), names, _split(gt["params_2"], sizes)):
        worst = max(worst, np.abs(p.detach().cpu().numpy().ravel() - want).max())
    report.append("parameters after two Adam steps on the reference's gradients: max |d| %.2e (lr 1e-3)" % worst)
    assert worst <= 2e-7
    # BatchNorm running statistics after two train-mode forwards, num_batches_tracked
    bufs = dict((n, b) for n, b in m.named_buffers())
    off = 0
    for n in [str(s) for s in gt["buffer_names"]]:
        b = bufs[n].detach().cpu().numpy().ravel()
        want = gt["buffers_2"][off:off + b.size]; off += b.size
        assert np.abs(b - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), n
    assert [int(b) for n, b in m.named_buffers() if n.endswith("num_batches_tracked")] == gt["num_batches_tracked_2"].tolist()
    with capsys.disabled():
        print("\n[training step vs reference] " + "\n[training step vs reference] ".join(report))

    # back to inference: eval() re-packs the trained weights (BN fold of the updated parameters and running statistics)
    m.eval()
    from oracle import backbone_oracle as bo
    with torch.no_grad():
        hl, hs = m(x[:4])
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    wl, ws = bo.forward(sd, x[:4].cpu())
    assert np.abs(hl.cpu().numpy() - wl.numpy()).max() <= 2e-3 and np.abs(hs.cpu().numpy() - ws.numpy()).max() <= 2e-3


CLEAN_TOL = 1e-4      # gradient elements that no flipped ReLU decision can reach: relative to the tensor's largest element (measured: ~1e-5)


def test_gradient_error_is_confined_to_flipped_relu_decisions(yf, golden, dev, capsys):
    """The first iteration's parameter gradients against the exact backward, ELEMENT-wise flip-aware.
    The excess over rounding in test_training_step_matches_the_reference (worst tensor 3.45e-2) is attributed to ReLU decisions that
    fall on the other side of zero than in exact arithmetic.  Here that is checked instead of narrated:
      1. the ReLU masks of OUR train-mode forward (y > 0 of every conv+BN+ReLU unit, read from the per-block path's tape, which is
         bit-identical to the trainer's) are compared with the masks of the oracle's float64 forward: the mismatches ARE the flips;
      2. tests/flip_reach.py propagates them to the parameter-gradient elements they can reach (with train-mode BatchNorm: channel c of
         the flipped layer's own parameters and every parameter upstream of it);
      3. every element they can NOT reach must agree with the exact backward to CLEAN_TOL = 1e-4 of its tensor's largest element
         (50x tighter than the 5e-3 the review asked for); elements they can reach keep the cap GRAD_CAP.
    Flips also occur in the last ReLU layers of both heads, so most of the network is reachable and the clean set is the parameters
    downstream of / beside the flips -- but there the agreement is at rounding level, which is what the claim predicts.  The test prints
    the flips, the clean set's size and worst error, and which tensor carries the overall worst error."""
    from oracle import backbone_oracle as bo
    from yolo_fastest_amd import training
    import flip_reach as fr
    gt, m, x, targets, crit, opt = _setup_step(yf, golden, dev)
    names = [n for n, _ in m.named_parameters()]
    sizes = gt["param_sizes"]
    zero = gt["grad_absmax_f64"] < 1e-9
    # 1. our masks (a forward that leaves the BatchNorm buffers as they were) and the float64 masks
    saved = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    with torch.no_grad():
        _, _, tape = training.train_forward(m, x)
    m.load_state_dict(saved, strict=False)
    m.train()
    sd64 = bo.training_state(bo.load_state_dict(WEIGHTS), torch.float64)
    pre = {}
    bo.forward(sd64, x.detach().cpu().double(), train=True, pre=pre)
    flips, n_flips, n_relu = {}, 0, 0
    for name, d in pre.items():
        ours = (tape[name][2] > 0).cpu()             # tape: (x, conv output, y = relu(BatchNorm(conv output)), ...): y > 0 <=> pre-activation > 0
        diff = ours != (d["z"] > 0)
        n_relu += diff.numel()
        if diff.any():
            flips[name] = torch.nonzero(diff.any(0).any(-1).any(-1)).ravel().tolist()
            n_flips += int(diff.sum())
    assert 0 < n_flips < 500, n_flips            # a few dozen of ~30 M decisions (measured: see the printed line)
    # 2. what they reach
    masks = fr.reach_masks(flips, names, [tuple(p.shape) for p in m.parameters()])
    # 3. our gradients against the exact backward
    opt.zero_grad()
    _iteration(m, crit, x, targets)
    worst_clean, worst_reach, n_clean_elems, n_clean_tensors = ("", 0.0), ("", 0.0), 0, 0
    for p, name, exact, mask, z in zip(m.parameters(), names, _split(gt["grads_1_exact"], sizes), masks, zero):
        if z:
            continue
        g = p.grad.detach().cpu().numpy().ravel()
        err = np.abs(g - exact) / np.abs(exact).max()
        if (~mask).any():
            e = float(err[~mask].max())
            n_clean_elems += int((~mask).sum()); n_clean_tensors += 1
            assert e <= CLEAN_TOL, (name, e, "an element no flipped ReLU decision can reach is off by more than rounding")
            if e > worst_clean[1]:
                worst_clean = (name, e)
        if mask.any():
            e = float(err[mask].max())
            assert e <= GRAD_CAP, (name, e)
            if e > worst_reach[1]:
                worst_reach = (name, e)
    assert n_clean_tensors >= 8 and n_clean_elems >= 20000, (n_clean_tensors, n_clean_elems)
    with capsys.disabled():
        print("\n[flip-aware gradients] %d of %d ReLU decisions differ from the float64 forward, in %s" % (n_flips, n_relu, {k: len(v) for k, v in flips.items()}))
        print("[flip-aware gradients] unreachable from any flip: %d elements in %d tensors, worst %.2e (%s); reachable: worst %.2e (%s)"
              % (n_clean_elems, n_clean_tensors, worst_clean[1], worst_clean[0], worst_reach[1], worst_reach[0]))


def test_training_free_running_two_steps(yf, golden, dev, capsys):
    """The same two iterations with our own gradients in the optimizer (training.train_step = train.py:111-132 verbatim).  Adam's
    first step is lr * g / (|g| + eps) = +-lr: only the SIGN of each gradient element matters, so the parameters after step 1 differ
    from the reference's by 2 lr where a sign differs (gradient elements at the noise level) and by ~0 elsewhere; step 2 mixes the two
    gradients.  Pinned: every element within the hard bound (2 steps of lr on either side); elements whose gradient is above 10 % of
    the tensor's largest agree to a quarter of lr in >= 99 %; the second iteration's loss within 2e-3 (measured 2e-4)."""
    from yolo_fastest_amd import training
    gt, m, x, targets, crit, opt = _setup_step(yf, golden, dev)
    sizes = gt["param_sizes"]
    zero = gt["grad_absmax_f64"] < 1e-9
    l1 = training.train_step(m, crit, opt, x, targets)
    l2 = training.train_step(m, crit, opt, x, targets)
    assert np.allclose(float(l1[0].detach()), gt["losses_1"][0], rtol=1e-4) and np.allclose(float(l2[0].detach()), gt["losses_2"][0], rtol=2e-3)
    assert float(l2[0].detach()) < float(l1[0].detach())
    lr = 0.001
    tot = bad = 0
    for p, want, g1, g2, z in zip(m.parameters(), _split(gt["params_2"], sizes), _split(gt["grads_1"], sizes), _split(gt["grads_2"], sizes), zero):
        got = p.detach().cpu().numpy().ravel()
        assert np.abs(got - want).max() <= 4.01 * lr
        if z:
            continue
        ok = (np.abs(g1) > 0.1 * np.abs(g1).max()) & (np.abs(g2) > 0.1 * np.abs(g2).max())
        tot += ok.sum(); bad += (np.abs(got - want)[ok] > 0.25 * lr).sum()
    with capsys.disabled():
        print("\n[free-running] losses %.5f -> %.5f (reference %.5f -> %.5f); %d of %d large-gradient elements differ by more than lr/4"
              % (float(l1[0].detach()), float(l2[0].detach()), gt["losses_1"][0], gt["losses_2"][0], bad, tot))
    assert tot > 1000 and bad <= 0.01 * tot, (bad, tot)


# Flip-level caps of test_training_network_against_the_fp64_oracle.  One flipped mask element of a BatchNorm'd channel
# moves what is merely upstream of it by its share of the channel's batch -- measured 3e-2 .. 2e-1 for these seeds, torch's own fp32 1e-2 ..
# 1e-1 -- and the flipped channel's own filter / gamma / beta by an O(1) share.  An upstream backward kernel that is wrong by O(0.3) of a
# tensor's range no longer fits under UP_CAP (it was 0.5), a wrong own-channel reduction no longer under OWN_CAP (it was 2.0).
UP_CAP, OWN_CAP = 0.25, 1.0


@pytest.mark.parametrize("seed,tol", [(1, 1e-3), (2, 2e-2)])
def test_training_network_against_the_fp64_oracle(yf, dev, seed, tol):
    """The composition (residual adds, the conv4_2 / conv5_2 branch points, deconv, torch.cat, both heads, the loss) on random weights
    and a small batch: our train-mode heads against oracle/backbone_oracle.py in float64, and our parameter gradients against the
    oracle's backward of OUR head gradients.  Every gradient element that no flipped ReLU decision of our fp32 forward can reach is
    within 1e-3 of its tensor's largest element for seed 1 (measured 3.5e-4 when the forward has no flip at all -- round 2's kernels; a
    forward with another summation order flips one decision in res5_5.conv2, and the tensors upstream of it then carry 3e-2 .. 2e-1, as
    torch's own fp32 does: 1e-2 .. 1e-1).  Seed 2 has a near-zero activation in conv4_1_4: 2e-2.
    (tools/train_oracle_report.py prints the per-tensor table, ours and torch-fp32.)"""
    from oracle import backbone_oracle as bo
    from yolo_fastest_amd import validation as val
    torch.manual_seed(seed)
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io)
    m.initialize_weights()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.bias.normal_(0, 0.2); mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
        m.head_4.bias.normal_(0, 0.5); m.head_5.bias.normal_(0, 0.5)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    N, H, W = 3, 64, 96
    x = torch.rand(N, 1, H, W) - 0.5
    rng = np.random.default_rng(seed)
    t = np.zeros((N, 8, 6), np.float32)
    for b in range(N):
        k = 2 + b % 5
        t[b, :k, 0:2] = rng.uniform(0.05, 0.95, (k, 2)); t[b, :k, 2:4] = rng.uniform(0.05, 0.6, (k, 2))
        t[b, :k, 4] = rng.integers(0, 3, k); t[b, :k, 5] = 255.0
    shape = [H, W, 1]
    pred = m(x.to(dev))
    for p in pred:
        p.retain_grad()
    loss = sum(val.YOLOLossV3(io["anchors"][i], 3, shape, dev, model=m)(p, torch.from_numpy(t).to(dev))[0] for i, p in enumerate(pred))
    loss.backward()
    sd = bo.training_state(sd0, torch.float64)
    keys = bo.parameter_keys(sd)
    pre = {}
    want = bo.forward(sd, x.double(), train=True, pre=pre)
    for got, w in zip(pred, want):
        assert np.abs(got.detach().cpu().numpy() - w.detach().numpy()).max() <= 2e-4 * max(1.0, float(w.detach().abs().max()))
    g64 = torch.autograd.grad(list(want), [sd[k] for k in keys], [p.grad.cpu().double() for p in pred])
    # ReLU decisions of OUR forward that differ from the float64 one (none for these seeds with some kernel generations, one with
    # others: which side of zero a pre-activation at rounding level lands on depends on the summation order of the forward kernels).
    # What a flip can reach (tests/flip_reach.py) only keeps the flip-level cap; everything else keeps `tol`.
    import flip_reach as fr
    from yolo_fastest_amd import training
    saved = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    with torch.no_grad():
        _, _, tape = training.train_forward(m, x.to(dev))
    m.load_state_dict(saved, strict=False)
    flips, n_flips = {}, 0
    for name, d in pre.items():
        diff = (tape[name][2] > 0).cpu() != (d["z"] > 0)
        if diff.any():
            flips[name] = torch.nonzero(diff.any(0).any(-1).any(-1)).ravel().tolist()
            n_flips += int(diff.sum())
    # a handful at most: with N = 3 frames of 64x96 there are ~1 M ReLU decisions, of which 0 .. 2 land on the other side of zero
    # (without a bound on the flips a late one would mark everything upstream "reachable" and the test would check nothing)
    assert n_flips <= 8, (n_flips, flips)
    names = [n for n, _ in m.named_parameters()]
    masks = fr.reach_masks(flips, names, [tuple(p.shape) for p in m.parameters()], parts=True)
    n_clean = n_clean_elems = n_elems = 0
    worst_up = worst_own = 0.0
    for (name, p), w, (own, up) in zip(m.named_parameters(), g64, masks):
        g, w = p.grad.cpu().numpy().astype(np.float64).ravel(), w.numpy().ravel()
        scale = np.abs(w).max()
        if scale < 1e-9:                        # zero in exact arithmetic (see the step test)
            assert np.abs(g).max() <= 1e-4, name
            continue
        err = np.abs(g - w) / scale
        clean = ~(own | up)
        n_elems += err.size; n_clean_elems += int(clean.sum())
        if clean.any():
            n_clean += 1
            assert err[clean].max() <= tol, (name, err[clean].max(), flips)
        # upstream of a flip: flip-level, not O(1) -- with 3 frames one mask element is 1/18 .. 1/1152 of a channel's batch; measured
        # 3e-2 .. 2e-1 (torch's own fp32: 1e-2 .. 1e-1).  An O(1) error of an upstream backward kernel does not fit under this cap.
        if up.any():
            worst_up = max(worst_up, float(err[up].max()))
            assert err[up].max() <= UP_CAP, (name, err[up].max(), flips)
        if own.any():                           # the flipped channel's own filter / gamma / beta: one mask element is an O(1) share
            worst_own = max(worst_own, float(err[own].max()))
            assert err[own].max() <= OWN_CAP, (name, err[own].max(), flips)
    print("seed %d: %d flips %s; worst upstream-of-a-flip error %.3g, worst own-channel error %.3g" % (seed, n_flips, flips, worst_up, worst_own))
    assert n_clean >= 10, (n_clean, flips)      # the flips (if any) leave at least the large head's private layers untouched
    assert n_clean_elems >= 0.1 * n_elems, (n_clean_elems, n_elems, flips)   # (seed 2: 16 %, a flip as far down as res3_5.conv2)
    # the running statistics moved like the module's buffers
    for k in ("conv0.1.running_mean", "res5_5.conv2.1.running_var", "conv4_1_5.1.running_var", "deconv5_1.1.running_mean"):
        assert np.allclose(m.state_dict()[k].cpu().numpy(), sd[k].numpy(), rtol=1e-5, atol=1e-7), k


def test_training_passes_replay_as_graphs_with_the_same_results(yf, dev):
    """A pass whose pointers repeat is captured once and replayed as a HIP graph (yf_train_engine.hip run_pass): fifteen iterations
    (batch 4, then 2, then 4 again) in a child process with the graphs on and with YF_TRAIN_GRAPH_OFF=1 end at the same loss to the last printed digit
    (the kernels are deterministic), and the trainer reports replays only in the first."""
    import subprocess
    code = ("import sys, ctypes, torch, numpy as np; sys.path.insert(0, %r)\n"
            "import yolo_fastest_amd as yf; from yolo_fastest_amd import training, validation as val\n"
            "dev = torch.device('cuda:0'); io = yf.io_params_for(256); torch.manual_seed(3)\n"
            "m = yf.YoloFastest(io); m.initialize_weights(); m = m.to(dev).train()\n"
            "x = (torch.rand(4, 1, 128, 160) - 0.5).to(dev)\n"
            "t = np.zeros((4, 8, 6), np.float32); t[:, 0] = (0.4, 0.6, 0.3, 0.2, 1, 255.0); td = torch.from_numpy(t).to(dev)\n"
            "crit = [val.YOLOLossV3(io['anchors'][i], 3, [128, 160, 1], dev, model=m) for i in range(2)]\n"
            "opt = training.Adam(m.parameters(), lr=0.001)\n"
            "for n, k in ((4, 6), (2, 4), (4, 5)):\n"           # the batch size changes and comes back: the backward's sum table is rebuilt
            "    for _ in range(k): loss = training.train_step(m, crit, opt, x[:n], td[:n])[0]\n"
            "tr = training._trainer(m, 128, 160, dev); f, b = ctypes.c_long(), ctypes.c_long()\n"
            "tr.lib.yf_trainer_graph_replays(tr.handle, ctypes.byref(f), ctypes.byref(b))\n"
            "print('RESULT %%.9g %%d %%d' %% (float(loss.detach()), f.value, b.value))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for off in (False, True):
        env = dict(os.environ)
        env.pop("YF_TRAIN_GRAPH_OFF", None)
        if off:
            env["YF_TRAIN_GRAPH_OFF"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        out[off] = line[0].split()[1:]
    assert out[False][0] == out[True][0], out
    assert int(out[True][1]) == 0 and int(out[True][2]) == 0, out
    assert int(out[False][1]) >= 3 and int(out[False][2]) >= 1, out          # captured when a pointer set comes back, replayed after


def test_training_graphs_with_three_rotating_input_buffers(yf, dev):
    """a loop that cycles through three input tensors (alternating buffers, gradient accumulation, interleaved validation
    tensors) must not re-capture a pass every iteration.  The trainer keeps a graph per remembered pointer set (four) and stops capturing
    a pass whose pattern keeps evicting graphs before they were replayed: over the second half of 36 iterations the captures of each
    pass stay bounded while the losses stay those of the eager path (the kernels are deterministic)."""
    import ctypes
    from yolo_fastest_amd import training, validation as val
    io = yf.io_params_for(256)

    def run(graphs):
        torch.manual_seed(5)
        m = yf.YoloFastest(io)
        m.initialize_weights()
        m = m.to(dev).train()
        xs = [(torch.rand(4, 1, 64, 96, generator=torch.Generator().manual_seed(i)) - 0.5).to(dev) for i in range(3)]
        t = np.zeros((4, 8, 6), np.float32); t[:, 0] = (0.4, 0.6, 0.3, 0.2, 1, 255.0)
        td = torch.from_numpy(t).to(dev)
        crit = [val.YOLOLossV3(io["anchors"][i], 3, [64, 96, 1], dev, model=m) for i in range(2)]
        opt = training.Adam(m.parameters(), lr=0.001)
        tr = training._trainer(m, 64, 96, dev)
        assert tr.lib.yf_trainer_set_graphs(tr.handle, 1 if graphs else 0) == 0
        stats, losses = [], []
        for it in range(36):
            losses.append(float(training.train_step(m, crit, opt, xs[it % 3], td)[0].detach()))
            if graphs and it in (17, 35):
                out = (ctypes.c_long * 6)()
                tr.lib.yf_trainer_graph_stats(tr.handle, out)
                stats.append(list(out))
        return losses, stats

    if os.environ.get("YF_TRAIN_GRAPH_OFF"):
        pytest.skip("graphs are switched off in this environment")
    losses, stats = run(True)
    half, full = stats
    print("\n[rotating inputs] replays fwd/bwd %d/%d, captures %d/%d, evictions %d/%d after 36 iterations (after 18: captures %d/%d)"
          % (full[0], full[1], full[2], full[3], full[4], full[5], half[2], half[3]))
    for p in (0, 1):
        assert full[2 + p] <= 12, full                       # never "a capture per iteration"
        assert full[2 + p] - half[2 + p] <= 4, (half, full)    # and bounded in the steady state: replaying, or given up capturing
    assert all(np.isfinite(losses))
    # ... and the 36 losses ARE those of the same loop issued as plain launches (the docstring said so, the test did not)
    eager, no_stats = run(False)
    assert no_stats == [] and losses == eager, [(i, a, b) for i, (a, b) in enumerate(zip(losses, eager)) if a != b][:4]


def test_training_forward_guards(yf, dev):
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev).train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 1, 64, 64))                                 # CPU tensor
    with pytest.raises(ValueError):
        m(torch.zeros(2, 1, 60, 64, device=dev))
    with torch.no_grad():                                            # no graph: plain heads, running statistics still move
        before = m.conv0[1].running_mean.clone()
        hl, hs = m(torch.rand(2, 1, 64, 64, device=dev))
        assert not hl.requires_grad and hl.shape == (2, 24, 4, 4) and hs.shape == (2, 24, 2, 2)
        assert not torch.equal(before, m.conv0[1].running_mean)
    hl, hs = m(torch.rand(2, 1, 64, 64, device=dev))
    assert hl.requires_grad
    (hl.sum() + hs.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    with pytest.raises(RuntimeError):
        (hl.sum()).backward()                                        # the tape is freed by the first backward


def test_train_loop_mirrors_train_py(yf, golden, dev, tmp_path):
    """training.train(params, device, tbwriter, train_dataset, val_dataset) = the reference's train() (train.py:44-160): the warm-up /
    cosine learning-rate sequence (checked against the formula of :87-88, :103-109 evaluated here), the log line of :143-147, one
    checkpoint per epoch that a fresh model loads strictly (the reference's 508 keys), the mAP report after epoch 4, tensorboard
    scalars -- on the 20 bundled frames with the synthetic targets of the mAP golden."""
    import copy, logging, re
    from yolo_fastest_amd import training
    g, gm = golden("golden_256"), golden("golden_map_256")
    items = [((g["input_u8"][i % 20].astype(np.float32) - 128.0)[..., None], gm["targets"][i % 20].astype(np.float32)) for i in range(40)]
    params = copy.deepcopy(yf.config_params)
    params["io_params"]["save_path"] = str(tmp_path / "models")
    params["train_params"].update(total_epochs=6, batch_size=8, pretrained_pth=os.path.join(
        ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights", "yolo_fastest_256x320_epoch28.pth"))
    lines = []

    class H(logging.Handler):
        def emit(self, rec):
            lines.append(rec.getMessage())
    logger = logging.getLogger("train_loop_test"); logger.setLevel(logging.INFO); logger.handlers = [H()]
    scalars = []

    class TB:
        def add_scalar(self, name, value, step):
            scalars.append((name, float(value), step))
    torch.manual_seed(0)
    model = training.train(params, dev, TB(), train_dataset=items, val_dataset=items[:20], logger=logger)
    bpe, total = 5, 6                                                  # 40 items / batch 8, drop_last
    logged = [l for l in lines if l.startswith("epoch [")]
    assert len(logged) == 3                                            # iterations 10, 20, 30
    num_warm = max(3 * bpe, 1000)
    for l, step in zip(logged, (10, 20, 30)):
        m = re.match(r"epoch \[(\d+)\]: current_batch = (\d+)/5, total_iter = (\d+), loss = (\d+\.\d{5}), example/sec = (\d+\.\d{3}), "
                     r"lr = (\d\.\d{5}), remain = \d+:\d\d:\d\d$", l)
        assert m, l
        epoch, batch, it = int(m.group(1)), int(m.group(2)), int(m.group(3))
        assert it == step and epoch == (step - 1) // bpe and batch == (step - 1) % bpe + 1
        iteration = step - 1
        lr = np.interp(iteration, [0, num_warm], [0.0, 0.001 * (((1 + np.cos(epoch * np.pi / total)) / 2) * 0.8 + 0.2)])
        assert abs(float(m.group(6)) - lr) < 6e-6, (l, lr)
        assert [s for s in scalars if s[0] == "lr" and s[2] == step][0][1] == pytest.approx(lr, rel=1e-9)
    assert {s[0] for s in scalars} == {"lr", "example/sec", "total_loss", "x", "y", "w", "h", "conf", "cls"}
    assert any("Load pretrained model" in l for l in lines) and any("epoch: 5 validation results" in l for l in lines)
    assert sum("mean AP" in l for l in lines) == 1                     # only epoch 5 (> 4)
    for e in range(total):
        sd = torch.load(os.path.join(params["io_params"]["save_path"], "YOLO-Fastest_epoch_%d.pth" % e), map_location="cpu")
        assert len(sd) == 508
    fresh = yf.YoloFastest(params["io_params"])
    assert str(fresh.load_state_dict(sd)) == "<All keys matched successfully>"
    assert int(sd["conv0.1.num_batches_tracked"]) == int(torch.load(params["train_params"]["pretrained_pth"], map_location="cpu")[
        "conv0.1.num_batches_tracked"]) + bpe * total
    assert not model.training                                          # get_mAP left it in eval mode, like the reference
    with pytest.raises(ValueError):
        training.train(params, dev)


def test_training_forward_at_640x512(yf, dev):
    """The other shipped input size, batch 2, random weights: train-mode heads against the float64 oracle; the gradients of the two
    head convolutions (which no ReLU decision upstream can disturb beyond rounding) against the oracle's; everything else finite."""
    from oracle import backbone_oracle as bo
    torch.manual_seed(3)
    io = yf.io_params_for(512)
    m = yf.YoloFastest(io)
    m.initialize_weights()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    x = torch.rand(2, 1, 512, 640) - 0.5
    hl, hs = m(x.to(dev))
    assert hl.shape == (2, 24, 32, 40) and hs.shape == (2, 24, 16, 20)
    ghl, ghs = torch.randn(hl.shape), torch.randn(hs.shape)
    torch.autograd.backward([hl, hs], [ghl.to(dev), ghs.to(dev)])
    sd = bo.training_state(sd0, torch.float64)
    want = bo.forward(sd, x.double(), train=True)
    for got, w in zip((hl, hs), want):
        assert np.abs(got.detach().cpu().numpy() - w.detach().numpy()).max() <= 2e-4 * max(1.0, float(w.detach().abs().max()))
    keys = ["head_4.weight", "head_4.bias", "head_5.weight", "head_5.bias"]
    g64 = torch.autograd.grad(list(want), [sd[k] for k in keys], [ghl.double(), ghs.double()])
    named = dict(m.named_parameters())
    for k, w in zip(keys, g64):
        g = named[k].grad.cpu().numpy()
        assert np.abs(g - w.numpy()).max() <= 1e-4 * np.abs(w.numpy()).max(), k
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_training_operators_random_geometries(ops, dev):
    """Fuzz of the conv / deconv / BatchNorm operators over random shapes (channel counts that are not multiples of the MFMA tiles,
    1-pixel and odd maps, batch 1, strides, all kernel sizes): every fast path and every fallback against torch in float64."""
    rng = np.random.default_rng(2024)
    for trial in range(48):
        dw = int(rng.integers(0, 3) == 0)
        k = int(rng.choice([1, 3, 5])) if not dw else int(rng.choice([3, 5]))
        stride = int(rng.integers(1, 3)) if k != 5 else 1
        N = int(rng.integers(1, 6))
        Cin = int(rng.integers(1, 70))
        Cout = Cin if dw else int(rng.integers(1, 70))
        H, W = int(rng.integers(1, 21)), int(rng.integers(1, 25))
        if trial % 3 == 0:
            H, W = 2 * ((H + 1) // 2), 4 * ((W + 3) // 4)            # the shapes the fast paths take
        x = rng.normal(size=(N, Cin, H, W)).astype(np.float32)
        w = rng.normal(size=(Cout, 1 if dw else Cin, k, k)).astype(np.float32)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        wt = torch.from_numpy(w).double().requires_grad_(True)
        yt = F.conv2d(xt, wt, None, stride=stride, padding=(k - 1) // 2, groups=Cin if dw else 1)
        gy = rng.normal(size=tuple(yt.shape)).astype(np.float32)
        yt.backward(torch.from_numpy(gy).double())
        xd, wd, gyd = _g(x, dev), _g(w, dev), _g(gy, dev)
        tag = "trial %d: N %d Cin %d Cout %d k %d s %d dw %d %dx%d" % (trial, N, Cin, Cout, k, stride, dw, H, W)
        y = torch.full(tuple(yt.shape), float("nan"), device=dev)
        ops.call("yf_train_conv_forward", xd.data_ptr(), wd.data_ptr(), None, y.data_ptr(), N, Cin, H, W, Cout, k, stride, dw)
        _close(y, yt, 3e-6, tag + " forward")
        gx = torch.full_like(xd, float("nan"))
        ops.call("yf_train_conv_backward_data", gyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout, k, stride, dw)
        _close(gx, xt.grad, 3e-6, tag + " backward data")
        gw = torch.full_like(wd, float("nan"))
        ops.call("yf_train_conv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), N, Cin, H, W, Cout, k, stride, dw, ops.scratch,
                 ops.scratch_bytes)
        _close(gw, wt.grad, 1e-5, tag + " backward weight")
        # BatchNorm on the conv output
        C, HW = Cout, yt.shape[2] * yt.shape[3]
        if N * HW > 2:
            bn = torch.nn.BatchNorm2d(C).double().train()
            zt = yt.detach().clone().requires_grad_(True)
            relu = trial & 1
            ot = F.relu(bn(zt)) if relu else bn(zt)
            ot.backward(torch.from_numpy(gy).double())
            gam, bet = _g(np.ones(C), dev), _g(np.zeros(C), dev)
            stats, yy = torch.empty(2 * C, device=dev), torch.empty_like(y)
            zd = _g(yt.detach().numpy(), dev)
            ops.call("yf_train_bn_forward", zd.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, None, stats.data_ptr(), yy.data_ptr(), N, C, HW, relu,
                     ops.scratch)
            assert np.abs(yy.cpu().numpy() - ot.detach().numpy()).max() <= 2e-5 * max(1.0, float(ot.detach().abs().max())), tag + " bn forward"
            dg, db, gz = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.full_like(zd, float("nan"))
            ops.call("yf_train_bn_backward", zd.data_ptr(), gyd.data_ptr(), stats.data_ptr(), gam.data_ptr(), bet.data_ptr(), dg.data_ptr(), db.data_ptr(),
                     gz.data_ptr(), N, C, HW, relu, ops.scratch)
            assert torch.isfinite(gz).all(), tag
            if N * HW >= 16:                                           # a handful of samples per channel: dx is ill-conditioned, skip
                _close(db, bn.bias.grad, 2e-5, tag + " dbeta")
    for trial in range(12):
        N, Cin, Cout, H, W = (int(rng.integers(1, 5)), int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 9)),
                              int(rng.integers(1, 11)))
        x = rng.normal(size=(N, Cin, H, W)).astype(np.float32)
        w = rng.normal(size=(Cin, Cout, 2, 2)).astype(np.float32)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        wt = torch.from_numpy(w).double().requires_grad_(True)
        yt = F.conv_transpose2d(xt, wt, stride=2)
        gy = rng.normal(size=tuple(yt.shape)).astype(np.float32)
        yt.backward(torch.from_numpy(gy).double())
        xd, wd, gyd = _g(x, dev), _g(w, dev), _g(gy, dev)
        tag = "deconv trial %d: N %d Cin %d Cout %d %dx%d" % (trial, N, Cin, Cout, H, W)
        y = torch.empty(tuple(yt.shape), device=dev)
        ops.call("yf_train_deconv_forward", xd.data_ptr(), wd.data_ptr(), y.data_ptr(), N, Cin, H, W, Cout)
        _close(y, yt, 3e-6, tag + " forward")
        gx = torch.full_like(xd, float("nan"))
        ops.call("yf_train_deconv_backward_data", gyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout)
        _close(gx, xt.grad, 3e-6, tag + " backward data")
        gw = torch.full_like(wd, float("nan"))
        ops.call("yf_train_deconv_backward_weight", xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), N, Cin, H, W, Cout, ops.scratch, ops.scratch_bytes)
        _close(gw, wt.grad, 1e-5, tag + " backward weight")


def test_trainer_is_identical_to_the_per_block_path(yf, golden, dev):
    """yf_trainer_forward / yf_trainer_backward (the graph walked in C++) against the same kernels orchestrated from Python one block at
    a time (model.train_impl = "ops"): the forward's launches are the same, so heads and running statistics are bit-identical; in the
    backward the trainer takes BatchNorm's two sums per channel out of the depthwise data-gradient kernel of the layer above where it
    can (fp32 over a workgroup's pixels, then double) and the per-block path reduces dy and z itself (double throughout): gradients equal to rounding."""
    gt = golden("golden_train_256")
    x = ((torch.from_numpy(gt["input_u8"][:6].astype(np.float32))[:, None] - 128.0) / 255.0).to(dev)
    out = {}
    for impl in ("trainer", "ops"):
        m = yf.YoloFastest(yf.io_params_for(256)).to(dev)
        m.load_state_dict(torch.load(WEIGHTS, map_location=dev))
        m.train()
        m.train_impl = impl
        hl, hs = m(x)
        torch.manual_seed(1)
        ghl, ghs = torch.randn(hl.shape, device=dev), torch.randn(hs.shape, device=dev)
        torch.autograd.backward([hl, hs], [ghl, ghs])
        out[impl] = (hl.detach(), hs.detach(), [p.grad.clone() for p in m.parameters()], [b.clone() for b in m.buffers()])
    a, b = out["trainer"], out["ops"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    zero = gt["grad_absmax_f64"] < 1e-9       # gradients that are zero in exact arithmetic (a BatchNorm bias in front of a conv + BatchNorm): noise
    for (n, _), ga, gb, z in zip(yf.YoloFastest(yf.io_params_for(256)).named_parameters(), a[2], b[2], zero):
        if z:
            assert ga.abs().max() <= 1e-2 and gb.abs().max() <= 1e-2, n
            continue
        assert (ga - gb).abs().max() <= 2e-5 * gb.abs().max(), (n, float((ga - gb).abs().max()), float(gb.abs().max()))
    for ba, bb in zip(a[3], b[3]):
        assert torch.equal(ba, bb)
    # another geometry (maps of 96x160 .. 6x10: other kernel choices per layer), random weights: the same statement
    xr = (torch.rand(5, 1, 192, 320, generator=torch.Generator().manual_seed(4)) - 0.5).to(dev)
    out = {}
    for impl in ("trainer", "ops"):
        torch.manual_seed(11)
        m = yf.YoloFastest(yf.io_params_for(256))
        m.initialize_weights()
        m = m.to(dev).train()
        m.train_impl = impl
        hl, hs = m(xr)
        torch.manual_seed(1)
        ghl, ghs = torch.randn(hl.shape, device=dev), torch.randn(hs.shape, device=dev)
        torch.autograd.backward([hl, hs], [ghl, ghs])
        out[impl] = (hl.detach(), hs.detach(), [p.grad.clone() for p in m.parameters()])
    a, b = out["trainer"], out["ops"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # (with beta = 0 at initialisation some gammas have a zero gradient in exact arithmetic too -- relu(gamma xhat) in front of a conv +
    #  BatchNorm is scale-free: such tensors hold rounding noise of the others' magnitude, hence the floor relative to the largest gradient)
    floor = 1e-6 * max(float(g.abs().max()) for g in b[2])
    for (n, _), ga, gb in zip(yf.YoloFastest(yf.io_params_for(256)).named_parameters(), a[2], b[2]):
        assert (ga - gb).abs().max() <= 2e-5 * gb.abs().max() + floor, (n, float((ga - gb).abs().max()), float(gb.abs().max()))
    # two forwards before the first backward: each pass owns its tape
    m = yf.YoloFastest(yf.io_params_for(256)).to(dev)
    m.load_state_dict(torch.load(WEIGHTS, map_location=dev))
    m.train()
    h1 = m(x[:2])
    h2 = m(x[2:6])
    (h1[0].sum() + h1[1].sum()).backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    (h2[0].sum() + h2[1].sum()).backward()
    m2 = yf.YoloFastest(yf.io_params_for(256)).to(dev)
    m2.load_state_dict(torch.load(WEIGHTS, map_location=dev))
    m2.train()
    h1b = m2(x[:2])
    (h1b[0].sum() + h1b[1].sum()).backward()
    for p, g in zip(m2.parameters(), g1):
        assert torch.equal(p.grad, g)


def test_data_parallel_training_single_rank_rccl(yf, golden, dev):
    """training.data_parallel on the real backend: a 1-rank RCCL group (the box has one GPU) runs the broadcast and the all-reduce of the
    flat gradient buffer; with one rank the gradients are unchanged.  The 2-rank arithmetic is covered on gloo (tests/test_dist_gloo.py)."""
    import torch.distributed as dist
    from yolo_fastest_amd import training
    gt = golden("golden_train_256")
    x = ((torch.from_numpy(gt["input_u8"][:4].astype(np.float32))[:, None] - 128.0) / 255.0).to(dev)

    def grads(dp):
        m = yf.YoloFastest(yf.io_params_for(256)).to(dev)
        m.load_state_dict(torch.load(WEIGHTS, map_location=dev))
        m.train()
        if dp:
            training.data_parallel(m)
        hl, hs = m(x)
        (hl.sum() + hs.sum()).backward()
        return [p.grad.clone() for p in m.parameters()]
    want = grads(False)
    with pytest.raises(RuntimeError):
        training.data_parallel(yf.YoloFastest(yf.io_params_for(256)).to(dev))      # no process group yet
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=dev)
    try:
        got = grads(True)
        torch.cuda.synchronize(dev)
    finally:
        dist.destroy_process_group()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_training_large_batch_replication_property(yf, golden, dev):
    """Size-independent property at a batch the oracle cannot follow: 512 frames = 32 copies of the 16 golden frames (N * C beyond
    65535 planes, a 24 GB tape).  Batch statistics, and therefore the heads, equal those of the 16-frame batch; with the mean-type loss
    (sum of the head gradients scaled by 1 / copies) so do the parameter gradients -- up to rounding and the ReLU decisions it moves
    (see test_training_step_matches_the_reference): heads 1e-5 of the range, gradient tensors by their median / 90th percentile."""
    gt = golden("golden_train_256")
    x16 = ((torch.from_numpy(gt["input_u8"].astype(np.float32))[:, None] - 128.0) / 255.0).to(dev)
    copies = 32
    torch.manual_seed(5)
    g_hl16, g_hs16 = torch.randn(16, 24, 16, 20, device=dev), torch.randn(16, 24, 8, 10, device=dev)

    def run(x, g_hl, g_hs):
        m = yf.YoloFastest(yf.io_params_for(256)).to(dev)
        m.load_state_dict(torch.load(WEIGHTS, map_location=dev))
        m.train()
        hl, hs = m(x)
        torch.autograd.backward([hl, hs], [g_hl, g_hs])
        return hl.detach(), hs.detach(), [p.grad.clone() for p in m.parameters()]
    hl16, hs16, g16 = run(x16, g_hl16, g_hs16)
    hl, hs, g = run(x16.repeat(copies, 1, 1, 1), g_hl16.repeat(copies, 1, 1, 1) / copies, g_hs16.repeat(copies, 1, 1, 1) / copies)
    assert hl.shape[0] == 16 * copies
    for big, small in ((hl, hl16), (hs, hs16)):
        scale = float(small.abs().max())
        assert float((big[:16] - small).abs().max()) <= 1e-5 * scale
        assert float((big[-16:] - small).abs().max()) <= 1e-5 * scale
    gt_zero = gt["grad_absmax_f64"] < 1e-9
    rel = []
    for a, b, z in zip(g, g16, gt_zero):
        assert torch.isfinite(a).all()
        if z:
            continue
        rel.append(float((a - b).abs().max() / b.abs().max()))
    rel = np.array(rel)
    assert np.median(rel) <= 5e-3 and np.quantile(rel, 0.9) <= 3e-2 and rel.max() <= 0.2, (np.median(rel), np.quantile(rel, 0.9), rel.max())
