"""io_params generality of the drop-in boundary: `YoloFastest(io_params)`, `YOLO_post_process`, the
validation decode / NMS, the training loss and the training step for num_cls / input_channel / num_anchors other than the shipped
(3, 1, 3) -- the reference's constructors are parameterised on all three (yolo_fastest.py:72-78,138,148; detect.py:15-21,53-66;
yolo_loss.py:28-33,58-60).  Expected values: tests/golden/golden_io.npz, which tests/golden/make_golden.py (main_io) made by running the
REFERENCE module / post-process / loss on numpy-seeded weights and inputs for every configuration of seeded_weights.IO_CONFIGS:
  c1 (1 class), c5rgb (5 classes, 3-channel input), c20 (20 classes: 75 head channels, the run-time head loop), a2 (2 anchors),
  c80rgb (80 classes: 255 head channels, 3-channel input).
Tolerances as in test_gpu_parity.py: heads within max(3 x E, 5e-5) of the graph in fp64 (E = the reference's own fp32 distance from
it), decode + NMS bit-exact including order given identical logits."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests import io_cfg

pytestmark = pytest.mark.gpu

ACCURACY_RATIO, ACCURACY_FLOOR, SCORE_TOL = 3.0, 5e-5, 1e-4


@pytest.fixture(scope="module")
def yf():
    import yolo_fastest_amd
    return yolo_fastest_amd


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


_MODELS = {}


def _model(yf, dev, golden, tag):
    if tag not in _MODELS:
        g = golden("golden_io")
        io = io_cfg.io_for(tag)
        m = yf.YoloFastest(io).to(dev).eval()
        m.load_state_dict({k: v.to(dev) for k, v in io_cfg.state_dict_for(tag, int(g[tag + "_seed"])).items()})   # strict
        post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
        _MODELS[tag] = (m, post, io)
    return _MODELS[tag]


def _x(tag, dev):
    from oracle import backbone_oracle as bo
    C, Cin, A = io_cfg.CONFIG[tag]
    return bo.preprocess(io_cfg.io_inputs(tag, Cin), Cin).to(dev)


def _check_heads(got, ref32, ref64, tag):
    got = got.cpu().numpy()
    assert got.shape == ref32.shape
    ours, theirs = np.abs(got - ref64).max(), np.abs(ref32 - ref64).max()
    bound = max(ACCURACY_RATIO * theirs, ACCURACY_FLOOR)
    assert ours <= bound, (tag, ours, theirs)
    sig = lambda a: 1.0 / (1.0 + np.exp(-a.astype(np.float64)))
    assert np.abs(sig(got) - sig(ref32)).max() < SCORE_TOL      # every sigmoid the reference could emit: conf, class scores, x, y


@pytest.mark.parametrize("tag", io_cfg.TAGS)
@pytest.mark.parametrize("fusion", [0, 1, 2])
def test_heads_match_the_reference_for_other_io_params(yf, dev, golden, tag, fusion):
    g = golden("golden_io")
    m, _, io = _model(yf, dev, golden, tag)
    C, Cin, A = io_cfg.CONFIG[tag]
    assert m.num_out == A * (5 + C) and m.head_5.weight.shape == (A * (5 + C), 128, 1, 1) and m.conv0[0].weight.shape == (8, Cin, 3, 3)
    m.fusion = fusion
    try:
        with torch.no_grad():
            hl, hs = m(_x(tag, dev))
    finally:
        m.fusion = yf.model.DEFAULT_FUSION
    assert hl.shape == (2, A * (5 + C), 16, 20) and hs.shape == (2, A * (5 + C), 8, 10)
    _check_heads(hl, g[tag + "_head_large"], g[tag + "_head_large_f64"], tag)
    _check_heads(hs, g[tag + "_head_small"], g[tag + "_head_small_f64"], tag)


def test_more_input_channels_than_the_fused_stem_takes(yf, dev, golden):
    """input_channel = 6 (yolo_fastest.py:78 takes any; the fused stem kernel is instantiated for 1 .. 4): conv0 is a launch of its own in
    the fused plans (one more launch than the shipped model), the pre-process handles 6-channel HWC frames (`img[:, :, ::-1]` reverses the
    channel axis whatever its length), and the fused u8 entry says what to call instead."""
    from oracle import backbone_oracle as bo
    m, _, io = _model(yf, dev, golden, "ch6")
    u8 = io_cfg.io_inputs("ch6", 6)
    x = yf.preprocess_u8(m, torch.from_numpy(u8).to(dev), io["input_shape"])
    assert torch.equal(x.cpu(), bo.preprocess(u8, 6))
    ops = m.profile(x, reps=1)
    assert ops[0]["name"] == "conv0" and ops[1]["name"].startswith("conv1_2") and len(ops) == 22
    with pytest.raises(RuntimeError, match="yf_preprocess_u8"):
        m.forward_u8(torch.from_numpy(u8).to(dev), io["input_shape"])
    with pytest.raises(NotImplementedError):
        yf.YoloFastest(dict(io, input_channel=65))


@pytest.mark.parametrize("tag", io_cfg.TAGS)
def test_fusion_levels_agree_bitwise_for_other_io_params(yf, dev, golden, tag):
    """levels 1 and 2 issue the same arithmetic in the same order (the chained small head included); level 0's per-layer kernels are a
    different summation order and only agree to rounding."""
    m, _, _ = _model(yf, dev, golden, tag)
    x = _x(tag, dev)
    out = {}
    # (round 5 switched the split-sum launches off here; since round 6 they carry the bits of the one-workgroup launches: the default stays)
    try:
        for f in (1, 2):
            m.fusion = f
            with torch.no_grad():
                out[f] = [t.clone() for t in m(x)]
    finally:
        m.fusion = yf.model.DEFAULT_FUSION
    assert torch.equal(out[1][0], out[2][0]) and torch.equal(out[1][1], out[2][1])


def _fp16_model(yf, dev, golden, tag, prec):
    g = golden("golden_io")
    io = io_cfg.io_for(tag)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict({k: v.to(dev) for k, v in io_cfg.state_dict_for(tag, int(g[tag + "_seed"])).items()})
    m.precision = prec
    return g, io, m


@pytest.mark.parametrize("tag", io_cfg.TAGS)
def test_f16x3_path_for_other_io_params(yf, dev, golden, tag):
    """split-operand fp16 MFMA (the variant that conforms to SURVEY.md 8(d).3): the fp32 bounds, and inside the survey's 2e-2 absolute."""
    g, io, m = _fp16_model(yf, dev, golden, tag, "f16x3")
    with torch.no_grad():
        hl, hs = m(_x(tag, dev))
    _check_heads(hl, g[tag + "_head_large"], g[tag + "_head_large_f64"], tag)
    _check_heads(hs, g[tag + "_head_small"], g[tag + "_head_small_f64"], tag)
    for got, ref in ((hl, g[tag + "_head_large"]), (hs, g[tag + "_head_small"])):
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-2


# fp16 STORAGE (dtype 1) on the seeded-weight models, absolute: SURVEY.md 8(d).3's own 2e-2.  These models' logits reach 1.4 .. 2.9 (one
# fp16 ulp there: 1e-3 .. 2e-3) and the measured maxima are 1.9e-3 .. 3.9e-3, so here the fp16-storage variant CONFORMS; on the shipped
# checkpoints (logits to +-36) it cannot (tests/test_gpu_parity.py FP16_STORAGE_MAX, test_fp16_rounding_alone_exceeds_the_surveys_tolerance).
SURVEY_FP16_TOL = 2e-2


@pytest.mark.parametrize("tag", io_cfg.TAGS)
def test_fp16_storage_variant_for_other_io_params(yf, dev, golden, tag):
    """The throughput variant (fp16 storage, single fp16 operands): max |logit - the reference's fp32 logit| <= 2e-2 ABSOLUTE (the survey's
    figure; measured 4e-3 at most) and the fp32 path's detections on the same frames."""
    g, io, m = _fp16_model(yf, dev, golden, tag, "f16")
    C, Cin, A = io_cfg.CONFIG[tag]
    with torch.no_grad():
        hl, hs = m(_x(tag, dev))
    worst, rng = 0.0, 0.0
    for got, ref in ((hl, g[tag + "_head_large"]), (hs, g[tag + "_head_small"])):
        worst = max(worst, float(np.abs(got.cpu().numpy() - ref).max())); rng = max(rng, float(np.abs(ref).max()))
    print(tag, "fp16 storage: max |dlogit| %.4g (logits reach %.3g; survey 2e-2)" % (worst, rng))
    assert worst <= SURVEY_FP16_TOL, (tag, worst, rng)
    # detections: those of the fp32 path on the same frames (class, cell, order; corners within 1 px)
    m32, post, _ = _model(yf, dev, golden, tag)
    with torch.no_grad():
        ref_heads = m32(_x(tag, dev))
    try:
        want = post.detect(tuple(ref_heads), kmax=A * 400, with_src=True)
    except ZeroDivisionError:
        with pytest.raises(ZeroDivisionError):
            post.detect((hl, hs), kmax=A * 400, with_src=True)
        return
    got = post.detect((hl, hs), kmax=A * 400, with_src=True)
    both = total = 0
    for f, (L, Wl) in enumerate(zip(got, want)):
        # seeded random weights put hundreds of candidates per frame around the thresholds: one whose conf lies within the fp16 variant's
        # error of conf_thre may appear or vanish, and a pair whose IoU lies at nms_thre may flip (the shipped checkpoints' frames, where
        # neither happens, are held to EXACT equality in test_gpu_parity).  Here: the detections clear of the confidence threshold are
        # the same set to 95 %, by (class, cell).
        clear = lambda E: {(e[6], e[7]) for e in E if abs(e[4] - io["conf_thre"]) > 0.02}
        a, b = clear(L), clear(Wl)
        both += len(a & b); total += max(len(a), len(b))
    print(tag, "fp16 storage: %d of %d detections (clear of the confidence threshold) shared with the fp32 path" % (both, total))
    assert both >= 0.95 * total, (tag, both, total)


@pytest.mark.parametrize("tag", io_cfg.TAGS)
def test_post_process_bit_exact_for_other_io_params(yf, dev, golden, tag):
    """decode + class buckets + stable sort + NMS on the REFERENCE's logits: candidates and survivors equal the reference's own
    YOLO_post_process run, in its order; a frame where the reference raised ZeroDivisionError raises here."""
    g = golden("golden_io")
    m, post, io = _model(yf, dev, golden, tag)
    C, Cin, A = io_cfg.CONFIG[tag]
    hl = torch.from_numpy(g[tag + "_head_large"]).to(dev)
    hs = torch.from_numpy(g[tag + "_head_small"]).to(dev)
    raw = post.detect_raw((hl, hs), kmax=A * 400)
    counts = raw["counts"].cpu().numpy()
    for f in range(hl.shape[0]):
        want = io_cfg.unpack_lists(g, tag + "_final", f)
        assert counts[f] == want["count"], (tag, f)
        if want["count"] == -2:
            with pytest.raises(ZeroDivisionError):
                post.to_lists(raw)
            continue
        n = want["count"]
        assert np.array_equal(raw["src"][f, :n].cpu().numpy(), want["src"])
        assert np.array_equal(raw["boxes"][f, :n].cpu().numpy(), want["box"])
        assert np.array_equal(raw["cls"][f, :n].cpu().numpy(), want["cls"])
        sc = raw["scores"][f, :n].cpu().numpy().astype(np.float64)
        assert np.abs(sc[:, 0] - want["conf"]).max() < 1e-6 and np.abs(sc[:, 1] - want["score"]).max() < 1e-6
        # the reference-named method: candidates of batch element 0 in decode order (detect.py:41-67)
        if f == 0:
            cands = post.decode_box((hl, hs))
            wc = io_cfg.unpack_lists(g, tag + "_cand", 0)
            assert len(cands) == wc["count"]
            assert np.array_equal(np.array([c[:4] for c in cands]), wc["box"]) and [c[6] for c in cands] == wc["cls"].tolist()


@pytest.mark.parametrize("tag", ["c5rgb", "c20", "a2", "ch2"])
def test_detect_single_call_and_own_logits(yf, dev, golden, tag):
    """yf_detect (forward + decode + NMS in one C call, two lanes) on the engine's OWN logits == the C oracle on those logits."""
    from oracle import post_oracle_c as poc
    m, post, io = _model(yf, dev, golden, tag)
    C, Cin, A = io_cfg.CONFIG[tag]
    x = _x(tag, dev)
    raw = post.detect_raw_from_input(x, kmax=A * 400)
    with torch.no_grad():
        hl, hs = m(x)
    assert torch.equal(raw["head_large"], hl) and torch.equal(raw["head_small"], hs)
    for f in range(x.shape[0]):
        try:
            r = poc.post_process(hl[f].cpu().numpy(), hs[f].cpu().numpy(), io["anchors"], io["input_shape"][:2], io["conf_thre"], io["nms_thre"], C,
                                 num_anchors=A)
        except ZeroDivisionError:
            assert int(raw["counts"][f]) == -2
            continue
        n = r["count"]
        assert int(raw["counts"][f]) == n
        assert np.array_equal(raw["src"][f, :n].cpu().numpy(), r["src"]) and np.array_equal(raw["boxes"][f, :n].cpu().numpy(), r["box"])
        assert np.array_equal(raw["cls"][f, :n].cpu().numpy(), r["cls"])


@pytest.mark.parametrize("tag", ["c5rgb", "c80rgb", "ch2", "ch4"])
def test_rgb_u8_frames_fused_preprocess(yf, dev, golden, tag):
    """Multi-channel frames as cv2.imread returns them (HWC; 3 channels: BGR): yf_preprocess_u8 == detect.py:119-124's arithmetic
    (`img[:, :, ::-1]` reverses the channel axis whatever its length), and the fused u8 stem (yf_forward_u8) is bit-identical to
    pre-process + forward; also from frames of exactly twice the net size (2x2 box mean per channel)."""
    from oracle import backbone_oracle as bo
    m, _, io = _model(yf, dev, golden, tag)
    Cin = io_cfg.CONFIG[tag][1]
    u8 = io_cfg.io_inputs(tag, Cin)
    x = yf.preprocess_u8(m, torch.from_numpy(u8).to(dev), io["input_shape"])
    assert torch.equal(x.cpu(), bo.preprocess(u8, Cin))
    with torch.no_grad():
        a = m(x)
        b = m.forward_u8(torch.from_numpy(u8).to(dev), io["input_shape"])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    big = np.random.default_rng(5).integers(0, 256, size=(2, 512, 640, Cin), dtype=np.uint8)
    small = ((big[:, 0::2, 0::2].astype(np.uint16) + big[:, 0::2, 1::2] + big[:, 1::2, 0::2] + big[:, 1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x2 = yf.preprocess_u8(m, torch.from_numpy(big).to(dev), io["input_shape"])
    assert torch.equal(x2.cpu(), bo.preprocess(small, Cin))
    with torch.no_grad():
        a = m(x2)
        b = m.forward_u8(torch.from_numpy(big).to(dev), io["input_shape"])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 256, 320, device=dev))       # a multi-channel model on a 1-channel tensor


@pytest.mark.parametrize("tag", io_cfg.TAGS)
def test_validation_decode_nms_and_loss_for_other_io_params(yf, dev, golden, tag):
    from yolo_fastest_amd import validation
    from oracle import val_oracle as vo
    g = golden("golden_io")
    m, _, io = _model(yf, dev, golden, tag)
    C, Cin, A = io_cfg.CONFIG[tag]
    pred = (torch.from_numpy(g[tag + "_head_large"]).to(dev), torch.from_numpy(g[tag + "_head_small"]).to(dev))
    crit = [validation.YOLOLossV3(io["anchors"][i], C, io["input_shape"], dev, model=m) for i in range(2)]
    dec = torch.cat([crit[i](pred[i]) for i in range(2)], 1)
    assert dec.shape == (2, A * 400, 5 + C)
    ref = vo.decode(tuple(p.cpu() for p in pred), io["anchors"], C, io["input_shape"])      # the oracle, pinned by the reference where A == 3
    assert (dec.cpu() - ref).abs().max().item() < 2e-4          # exp / sigmoid on the device against the host's, boxes up to ~300 px
    if A == 3:
        np.testing.assert_allclose(dec.cpu().numpy()[:1, ::3], g[tag + "_val_decode"], rtol=0, atol=2e-4)
    # NMS on the ORACLE's decode (identical inputs): bit-identical detections, order included
    dets = validation.non_max_suppression(ref.to(dev), C, conf_thres=0.5, nms_thres=0.2, model=m)
    want = vo.non_max_suppression(ref, C, 0.5, 0.2)
    for d, w in zip(dets, want):
        assert (d is None) == (w is None)
        if d is not None:
            assert torch.equal(d.cpu(), w)
    if A == 3:
        for f, d in enumerate(dets):
            n = int(g[tag + "_val_count"][f])
            assert (0 if d is None else d.shape[0]) == n
            np.testing.assert_allclose(d.cpu().numpy(), g[tag + "_val_det"][f, :n], rtol=1e-6, atol=1e-5)   # (torch's exp on this host vs the golden's)
    # the training loss of both heads and its gradient against the reference's own run
    tt = torch.from_numpy(io_cfg.io_targets(tag, C, 2)).to(dev)
    for i, name in enumerate(("head_large", "head_small")):
        x = pred[i].clone().requires_grad_(True)
        res = crit[i](x, tt)
        res[0].backward()
        want_l = g[f"{tag}_{name}_losses"]
        got_l = np.array([float(res[0])] + [float(v) for v in res[1:]])
        np.testing.assert_allclose(got_l, want_l, rtol=2e-5, atol=1e-7)
        wg = g[f"{tag}_{name}_grad"]
        assert np.abs(x.grad.cpu().numpy() - wg).max() <= 2e-5 * np.abs(wg).max()


@pytest.mark.parametrize("tag", ["c5rgb", "a2", "ch4", "ch6"])
def test_training_step_for_other_io_params(yf, dev, golden, tag):
    """model.train(); pred = model(imgs); the two-head loss; loss.backward() (train.py:111-131) for an RGB 5-class and a 2-anchor model
    against the reference's own iteration: train-mode heads, the seven losses, a strided sample and the per-tensor sums of every
    parameter gradient, the BatchNorm running statistics."""
    from yolo_fastest_amd import validation
    from oracle import backbone_oracle as bo
    g = golden("golden_io")
    C, Cin, A = io_cfg.CONFIG[tag]
    io = io_cfg.io_for(tag, 64, 96)
    m = yf.YoloFastest(io).to(dev)
    m.load_state_dict({k: v.to(dev) for k, v in io_cfg.state_dict_for(tag, int(g[tag + "_seed"])).items()})
    m.train()
    u8 = io_cfg.io_inputs(tag + "_train", Cin, n=4, H=64, W=96)
    tt = torch.from_numpy(io_cfg.io_targets(tag + "_train", C, 4)).to(dev)
    crit = [validation.YOLOLossV3(io["anchors"][i], C, io["input_shape"], dev, model=m) for i in range(2)]
    pred = m(bo.preprocess(u8, Cin).to(dev))
    # yardstick: the train-mode graph in fp64 (the oracle, pinned by the reference's own iteration in test_oracle_golden.py).  With 4
    # frames of 64x96 a stride-32 BatchNorm normalises over 24 values, so fp32 rounding is amplified: the reference's own fp32 heads
    # are E away from fp64; ours must be within max(3 E, 5e-5) like the inference heads
    sd64 = bo.training_state(io_cfg.state_dict_for(tag, int(g[tag + "_seed"])), torch.float64)
    t64 = bo.forward(sd64, bo.preprocess(u8, Cin).double(), train=True)
    for got, ref32, ref64 in ((pred[0], g[tag + "_train_head_large"], t64[0]), (pred[1], g[tag + "_train_head_small"], t64[1])):
        ref64 = ref64.detach().numpy()
        ours, theirs = np.abs(got.detach().cpu().numpy() - ref64).max(), np.abs(ref32 - ref64).max()
        print(tag, "train-mode heads vs fp64: ours %.3g, the reference's fp32 %.3g" % (ours, theirs))
        assert ours <= max(ACCURACY_RATIO * theirs, ACCURACY_FLOOR), (ours, theirs)
    losses = [[] for _ in range(7)]
    for i, p in enumerate(pred):
        for j, v in enumerate(crit[i](p, tt)):
            losses[j].append(v)
    losses = [sum(v) for v in losses]
    np.testing.assert_allclose([float(v) for v in losses], g[tag + "_train_losses"], rtol=2e-4)
    losses[0].backward()
    names = [n for n, _ in m.named_parameters()]
    assert names == [str(k) for k in g[tag + "_train_param_names"]]
    grads = [p.grad.detach().cpu().numpy().ravel() for p in m.parameters()]
    flat, want = np.concatenate(grads)[::37], g[tag + "_train_grad_sample"]
    # Yardstick for the gradients: the same iteration in fp64 (the oracle; its fp32 run is pinned by the reference's own iteration in
    # test_oracle_golden.py).  A ReLU decision within fp32 rounding of zero that comes out on the other side flips one mask element and,
    # through train-mode BatchNorm, moves the gradients of its channel and of EVERYTHING upstream by percents ("The
    # training step") -- in the reference's own fp32 run as well (a2: 2.6e-3 from fp64 in its golden sample).  So, as in
    # test_gpu_training.test_gradient_error_is_confined_to_flipped_relu_decisions: find the flips of OUR forward against the fp64
    # forward, let tests/flip_reach.py say what they can reach, and hold every element they can NOT reach to 1e-4 of its tensor's
    # largest element, the reachable ones to the cap.
    from oracle import loss_oracle as lo
    from yolo_fastest_amd import training
    import flip_reach as fr
    torch.set_default_dtype(torch.float64)
    try:
        pre = {}
        sd64 = bo.training_state(io_cfg.state_dict_for(tag, int(g[tag + "_seed"])), torch.float64)
        t64 = bo.forward(sd64, bo.preprocess(u8, Cin).double(), train=True, pre=pre)
        parts = [lo.loss_head(h, tt.cpu().double(), io["anchors"][i], C, io["input_shape"]) for i, h in enumerate(t64)]
        keys = bo.parameter_keys(sd64)
        g64 = [v.numpy().ravel() for v in torch.autograd.grad(parts[0][0] + parts[1][0], [sd64[k] for k in keys])]
    finally:
        torch.set_default_dtype(torch.float32)
    m2 = yf.YoloFastest(io).to(dev)
    m2.load_state_dict({k: v.to(dev) for k, v in io_cfg.state_dict_for(tag, int(g[tag + "_seed"])).items()})
    m2.train()
    with torch.no_grad():
        _, _, tape = training.train_forward(m2, bo.preprocess(u8, Cin).to(dev))     # the per-block path: bit-identical to the trainer's forward
    flips, n_flips = {}, 0
    for name, d in pre.items():
        diff = (tape[name][2] > 0).cpu() != (d["z"] > 0)
        if diff.any():
            flips[name] = torch.nonzero(diff.any(0).any(-1).any(-1)).ravel().tolist()
            n_flips += int(diff.sum())
    shapes = [tuple(p.shape) for p in m.parameters()]
    masks = fr.reach_masks(flips, names, shapes, parts=True)
    # The yardstick for "rounding level" in the clean set: torch's OWN fp32 evaluation of the same iteration on the CPU (the oracle graph in
    # fp32), against the same fp64 result, on the elements neither its flips nor ours can reach -- the rule the head logits are held to
    # (ours <= max(3 E, floor)); no bound here comes from a measurement of the kernels under test.
    pre32 = {}
    sd32 = bo.training_state(io_cfg.state_dict_for(tag, int(g[tag + "_seed"])), torch.float32)
    t32 = bo.forward(sd32, bo.preprocess(u8, Cin), train=True, pre=pre32)
    parts32 = [lo.loss_head(h, tt.cpu(), io["anchors"][i], C, io["input_shape"]) for i, h in enumerate(t32)]
    g32 = [v.numpy().ravel() for v in torch.autograd.grad(parts32[0][0] + parts32[1][0], [sd32[k] for k in bo.parameter_keys(sd32)])]
    flips32 = {}
    for name, d in pre.items():
        diff = (pre32[name]["z"] > 0) != (d["z"] > 0)
        if diff.any():
            flips32[name] = torch.nonzero(diff.any(0).any(-1).any(-1)).ravel().tolist()
    masks32 = fr.reach_masks(flips32, names, shapes)
    worst_clean, torch_clean, worst_up, worst_own, n_clean, n_elems = 0.0, 0.0, 0.0, 0.0, 0, 0
    for gv, ex, e32, (own, up), m32, nm in zip(grads, g64, g32, masks, masks32, names):
        if np.abs(ex).max() < 1e-9:      # zero in exact arithmetic (a BatchNorm bias that only feeds train-mode BatchNorms): rounding noise only
            continue
        err, err32 = np.abs(gv - ex) / np.abs(ex).max(), np.abs(e32 - ex) / np.abs(ex).max()
        clean = ~(own | up | m32)
        n_elems += err.size
        if clean.any():
            worst_clean = max(worst_clean, float(err[clean].max())); torch_clean = max(torch_clean, float(err32[clean].max())); n_clean += int(clean.sum())
        if up.any():
            worst_up = max(worst_up, float(err[up].max()))
        if own.any():
            worst_own = max(worst_own, float(err[own].max()))
    f64s = np.concatenate(g64)[::37]
    print(tag, "%d ReLU decisions differ from the fp64 forward (%s; torch's own fp32: %s); gradient elements neither can reach: %d of %d, worst %.3g "
          "(torch fp32: %.3g); merely upstream of one of our flips: worst %.3g; the flipped channels' own parameters: worst %.3g; the reference's own "
          "fp32 sample vs fp64: %.3g" % (n_flips, {k: len(v) for k, v in flips.items()}, {k: len(v) for k, v in flips32.items()}, n_clean, n_elems,
                                        worst_clean, torch_clean, worst_up, worst_own, np.abs(want - f64s).max() / np.abs(f64s).max()))
    # the same structure as test_gpu_training.test_training_network_against_the_fp64_oracle -- few flips (4 frames of 64x96 hold
    # ~1.4 M ReLU decisions), what is merely UPSTREAM of a flip stays at flip level (one mask element of a stride-32 BatchNorm over 24 values
    # is 4 % of its channel's batch: tens of percents, not O(1)), only the flipped channel's own filter / gamma / beta may move by O(1), and
    # a real share of the elements is out of every flip's reach and held to rounding level.  An O(1) error of an upstream backward kernel
    # (conv0's 3-channel weight gradient included) does not fit under the upstream cap.
    assert n_flips <= 16, (n_flips, flips)
    assert n_clean >= 50000 and n_clean >= 0.1 * n_elems, (n_clean, n_elems, flips)
    assert worst_clean <= max(ACCURACY_RATIO * torch_clean, 1e-4), (worst_clean, torch_clean)
    assert worst_up <= 0.25 and worst_own <= 1.0, (worst_up, worst_own, flips)     # test_gpu_training.UP_CAP / OWN_CAP (measured here: 0.12 / 0.45 at most)
    if n_flips == 0:
        assert np.abs(flat - want).max() <= 1e-3 * np.abs(want).max()
    bufs = np.concatenate([b.detach().cpu().numpy().ravel() for n, b in m.named_buffers() if not n.endswith("num_batches_tracked")])
    assert np.abs(bufs[::7] - g[tag + "_train_buffers_sample"]).max() <= 1e-4
    # eval() after training re-packs the updated parameters for the inference engine
    m.eval()
    with torch.no_grad():
        hl, hs = m(bo.preprocess(u8, Cin).to(dev))
    assert hl.shape == (4, A * (5 + C), 4, 6) and torch.isfinite(hl).all() and torch.isfinite(hs).all()


def test_blob_header_is_checked(yf, dev, golden):
    """yf_create reads num_anchors / num_cls / input_channel from the blob header and refuses inconsistent ones."""
    import struct
    from yolo_fastest_amd import _lib, packer
    g = golden("golden_io")
    lib = _lib.lib()
    sd = io_cfg.state_dict_for("a2", int(g["a2_seed"]))
    blob = packer.pack_state_dict(sd, 16, 1, 2, 3)
    h = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(blob, len(blob))
    assert lib.yf_create(buf, len(blob), 64, 96, 2, 0, ctypes.byref(h)) == 0
    vals = [ctypes.c_int() for _ in range(4)]
    assert lib.yf_io_params(h, *[ctypes.byref(v) for v in vals]) == 0
    assert [v.value for v in vals] == [1, 2, 3, 16]
    lib.yf_destroy(h)
    bad = bytearray(blob)
    struct.pack_into("<I", bad, 8 + 4 * 4, 3)                  # num_anchors 3 with num_out 16: 3 * (5 + 3) != 16
    b2 = ctypes.create_string_buffer(bytes(bad), len(bad))
    assert lib.yf_create(b2, len(bad), 64, 96, 2, 0, ctypes.byref(h)) == _lib.YF_E_BLOB
    assert b"num_out" in lib.yf_last_error_string()
    with pytest.raises(RuntimeError, match="Missing key|size mismatch|Unexpected key"):   # strict load: a 3-class checkpoint into a 5-class model
        m = yf.YoloFastest(io_cfg.io_for("c5rgb"))
        m.load_state_dict(sd)


@pytest.mark.parametrize("tag,H,W", [("c20", 512, 640), ("c5rgb", 96, 160), ("c80rgb", 96, 160), ("a2", 512, 640)])
def test_other_io_params_at_other_input_sizes(yf, dev, golden, tag, H, W):
    """Sizes where the stride-16 / 32 tiles are not whole frames (640x512: several tiles per frame in the head kernels, no chained small
    head) and a ragged one: heads against the oracle in fp32 and fp64, detections against the C oracle on the engine's own logits."""
    from oracle import backbone_oracle as bo
    from oracle import post_oracle_c as poc
    g = golden("golden_io")
    C, Cin, A = io_cfg.CONFIG[tag]
    io = io_cfg.io_for(tag, H, W)
    sd = io_cfg.state_dict_for(tag, int(g[tag + "_seed"]))
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    u8 = io_cfg.io_inputs(tag + "_sz", Cin, n=2, H=H, W=W)
    x = bo.preprocess(u8, Cin)
    with torch.no_grad():
        hl, hs = m(x.to(dev))
    ol, os_ = bo.forward(sd, x)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    tl, ts = bo.forward(sd64, x.double())
    _check_heads(hl, ol.numpy(), tl.numpy(), tag)
    _check_heads(hs, os_.numpy(), ts.numpy(), tag)
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], A, C, io["anchors"], io["input_shape"]).bind(m)
    ncell = A * (H // 16 * (W // 16) + H // 32 * (W // 32))
    raw = post.detect_raw((hl, hs), kmax=ncell)
    for f in range(2):
        try:
            r = poc.post_process(hl[f].cpu().numpy(), hs[f].cpu().numpy(), io["anchors"], io["input_shape"][:2], io["conf_thre"], io["nms_thre"], C,
                                 num_anchors=A)
        except ZeroDivisionError:
            assert int(raw["counts"][f]) == -2
            continue
        n = r["count"]
        assert int(raw["counts"][f]) == n
        assert np.array_equal(raw["src"][f, :n].cpu().numpy(), r["src"]) and np.array_equal(raw["boxes"][f, :n].cpu().numpy(), r["box"])
        assert np.array_equal(raw["cls"][f, :n].cpu().numpy(), r["cls"])


def test_eight_anchors_and_the_cell_limit(yf, dev):
    """num_anchors = 8 (the most the post-process takes): the model runs at any size; the on-chip post-process holds at most 8191 cells
    per frame (13-bit field of its sort key / 160 KB of LDS) and says so beyond -- 8 anchors x (16x20 + 8x10) = 3200 cells at 320x256 fit,
    8 x (32x40 + 16x20) = 12800 at 640x512 do not."""
    from yolo_fastest_amd import _lib
    from oracle import backbone_oracle as bo
    from oracle import post_oracle_c as poc
    import copy
    io = copy.deepcopy(yf.config_params["io_params"])
    anchors = [[[10 + 7 * k, 13 + 5 * k] for k in range(8)], [[40 + 20 * k, 150 - 15 * k] for k in range(8)], [[1, 1]] * 8]
    io.update(num_cls=2, num_anchors=8, anchors=anchors)
    m = yf.YoloFastest(io)
    from collections import OrderedDict
    sd = OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in io_cfg.seeded_state_dict(
        OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items()), 77).items())
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    assert m.num_out == 56
    u8 = np.random.default_rng(8).integers(0, 256, size=(1, 256, 320), dtype=np.uint8)
    with torch.no_grad():
        hl, hs = m(bo.preprocess(u8).to(dev))
    ol, os_ = bo.forward(sd, bo.preprocess(u8))
    assert (hl.cpu() - ol).abs().max().item() < 1e-4 and (hs.cpu() - os_).abs().max().item() < 1e-4
    post = yf.YOLO_post_process(0.5, 0.2, 8, 2, anchors, io["input_shape"]).bind(m)
    raw = post.detect_raw((hl, hs), kmax=3200)
    r = poc.post_process(hl[0].cpu().numpy(), hs[0].cpu().numpy(), anchors, [256, 320], 0.5, 0.2, 2, num_anchors=8)
    n = r["count"]
    assert int(raw["counts"][0]) == n and n > 50
    assert np.array_equal(raw["src"][0, :n].cpu().numpy(), r["src"]) and np.array_equal(raw["boxes"][0, :n].cpu().numpy(), r["box"])
    io5 = dict(io, input_shape=[512, 640, 1])
    post5 = yf.YOLO_post_process(0.5, 0.2, 8, 2, anchors, io5["input_shape"]).bind(m)
    with torch.no_grad():
        h5 = m(torch.zeros(1, 1, 512, 640, device=dev))
    with pytest.raises(_lib.YFError, match="too many cells"):
        post5.detect_raw(h5, kmax=64)
    with pytest.raises(ValueError):
        yf.YoloFastest(dict(io, num_anchors=9))


def test_detect_yolo_driver_with_an_rgb_five_class_model(yf, dev, golden, tmp_path):
    """`Detect_YOLO(device, model_path, config_params, logger).batch_detect(data_path, result_path)` (detect.py:87-192) for io_params other
    than the shipped ones: a 5-class model on 3-channel frames read from image files -- what it logs and draws equals the direct path
    (BGR frame -> fused u8 pre-process -> model -> post-process) on the same files."""
    import logging
    from PIL import Image
    g = golden("golden_io")
    io = io_cfg.io_for("c5rgb")
    io["origin_img_shape"] = [512, 640, 3]        # frames of twice the net size: the exact-2x box mean per channel + __adjust_coord
    sd = io_cfg.state_dict_for("c5rgb", int(g["c5rgb_seed"]))
    torch.save(sd, tmp_path / "m.pth")
    data, res = tmp_path / "data", tmp_path / "res"
    data.mkdir(); res.mkdir()
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(3, 512, 640, 3), dtype=np.uint8)      # RGB as stored in the files
    for i, f in enumerate(frames):
        Image.fromarray(f).save(data / ("f%d.png" % i))
    lines = []

    class H(logging.Handler):
        def emit(self, rec):
            lines.append(rec.getMessage())
    logger = logging.getLogger("yf-rgb"); logger.setLevel(logging.INFO); logger.addHandler(H())
    det = yf.Detect_YOLO(dev, str(tmp_path / "m.pth"), {"io_params": io}, logger)
    det.batch_detect(str(data), str(res))
    assert len(lines) == 4 and lines[-1].startswith("detect avg_time") and all((res / ("result_f%d.png" % i)).exists() for i in range(3))
    bgr = torch.from_numpy(np.ascontiguousarray(frames[:, :, :, ::-1])).to(dev)   # what cv2.imread hands over
    direct = det.detect_u8(bgr, kmax=1200)
    for i in range(3):
        want = ["%s %.2f" % (io["class_names"][int(b[6])], b[4] * b[5]) for b in direct[i]]
        assert det.last_labels["f%d.png" % i] == want and ("detect finished" in lines[i]) == (len(want) > 0)
