"""Parity tests proper: the HIP path (through the C ABI) against the oracle and the reference-generated goldens.
Run with `-m gpu` on an MI355X.

Tolerances (floating point; BASELINE.json north_star: "boxes/scores within 1e-4 fp32, bit-exact NMS index order"):
  * SCORES -- sigmoid(conf logit) and sigmoid(class logit), the numbers the reference emits: 1e-4 absolute
    (SCORE_TOL) for every emitted detection at every size, and for EVERY cell of every frame at 320x256 (the
    size of BASELINE.json's metric).  At 640x512 the all-cell bound is the noise-floor bound below (the
    reference's own fp32 scores are 1.4e-4 away from its fp64 scores there, so no implementation, not even an
    exact one, can promise 1e-4 against it on every cell);
  * raw head logits: two fp32 evaluations of this 86-layer graph cannot agree to 1e-4 on logits of magnitude
    ~30: the reference's own fp32 output is 1.3e-4 (256x320) / 5.7e-4 (512x640) away from its own graph evaluated
    in fp64 (goldens head_*_f64, made with model.double()), and torch-CPU with BN-folded weights is 2.1e-4 away
    from torch-CPU unfolded.  The logit test is therefore relative to that measured noise floor E = max |ref_fp32 -
    ref_fp64|: the HIP result must be within max(ACCURACY_RATIO x E, ACCURACY_FLOOR) of the fp64 result (the same
    accuracy class as the reference itself) and hence within that + E of the reference's fp32 logits;
  * decode/NMS given identical logits: bit-exact boxes, classes and survivor ORDER (source indices);
    scores 1e-6 (computed in fp64, stored fp32)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WDIR = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights")
WEIGHTS = {256: os.path.join(WDIR, "yolo_fastest_256x320_epoch28.pth"),
           512: os.path.join(WDIR, "yolo_fastest_512x640_epoch27.pth")}
SCORE_TOL = 1e-4
ACCURACY_RATIO = 3.0
ACCURACY_FLOOR = 5e-5


def _sig(a):
    return 1.0 / (1.0 + np.exp(-a.astype(np.float64)))


def _score_err(got, want):
    """max |sigmoid(got) - sigmoid(want)| over the conf and class channels (4..7 of every anchor's 8)."""
    n, c, h, w = got.shape
    g = got.reshape(n, 3, 8, h, w)[:, :, 4:]
    r = want.reshape(n, 3, 8, h, w)[:, :, 4:]
    return np.abs(_sig(g) - _sig(r)).max()


@pytest.fixture(scope="module")
def yf():
    import yolo_fastest_amd
    return yolo_fastest_amd


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def models(yf, dev):
    out = {}
    for res, p in WEIGHTS.items():
        io = yf.io_params_for(res)
        m = yf.YoloFastest(io).to(dev).eval()
        m.load_state_dict(torch.load(p, map_location=dev))
        post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"],
                                    io["input_shape"]).bind(m)
        out[res] = (m, post, io)
    return out


def _x(u8, dev):
    from oracle import backbone_oracle as bo
    return bo.preprocess(u8).to(dev)


@pytest.fixture
def bit_stable(models):
    """The tests that assert BITWISE independence of a frame's result from the batch it travels in (chunking, position, fusion level, one
    call or two) once switched the few-frames split-sum launches off (round 5: they were another association of two channel sums).  Since round 6 those
    launches carry the bits of the large-batch kernels (test_split_sum_launches_carry_the_bits_of_the_large_batch_plan), so the tests that use this fixture
    run in the DEFAULT mode, split-sum launches on; the fixture only restores that default should a test have changed it."""
    for m, _, _ in models.values():
        m.split_sums = True
    yield
    for m, _, _ in models.values():
        m.split_sums = True


_SD64 = {}


def _truth64(res, u8):
    """The graph in fp64 (oracle/backbone_oracle.py run on a double state-dict)."""
    from oracle import backbone_oracle as bo
    if res not in _SD64:
        sd = bo.load_state_dict(WEIGHTS[res])
        _SD64[res] = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    hl, hs = bo.forward(_SD64[res], bo.preprocess(u8).double())
    return hl.numpy(), hs.numpy()


def _check_heads(hl, hs, ref32_l, ref32_s, ref64_l, ref64_s, metric_size=True):
    for got, ref32, ref64 in ((hl, ref32_l, ref64_l), (hs, ref32_s, ref64_s)):
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        ref32 = ref32.numpy() if isinstance(ref32, torch.Tensor) else ref32
        assert got.shape == ref32.shape
        ours = np.abs(got - ref64).max()
        theirs = np.abs(ref32 - ref64).max()
        bound = max(ACCURACY_RATIO * theirs, ACCURACY_FLOOR)
        assert ours <= bound, (ours, theirs)
        assert np.abs(got - ref32).max() <= bound + theirs
        # all-cell scores: 1e-4 at the metric's size; noise-floor bound (sigmoid' <= 1/4) beyond it
        stol = SCORE_TOL if metric_size else max(SCORE_TOL, 0.25 * (bound + theirs))
        assert _score_err(got, ref32) < stol, (_score_err(got, ref32), stol)


def test_extension_loaded(yf):
    import ctypes
    assert isinstance(yf._lib.lib(), ctypes.CDLL)
    assert os.path.exists(yf._lib.LIB_PATH)


@pytest.mark.parametrize("res", [256, 512])
def test_heads_match_reference_goldens(models, golden, dev, res):
    m, _, _ = models[res]
    g = golden(f"golden_{res}")
    with torch.no_grad():
        hl, hs = m(_x(g["input_u8"], dev))
    _check_heads(hl, hs, g["head_large"], g["head_small"], g["head_large_f64"], g["head_small_f64"], res == 256)
    with torch.no_grad():
        hl, hs = m(_x(g["syn_input_u8"], dev))
    t = _truth64(res, g["syn_input_u8"])
    _check_heads(hl, hs, g["syn_head_large"], g["syn_head_small"], t[0], t[1], res == 256)


@pytest.mark.parametrize("fusion", [0, 1, 2])
def test_layer_probes_match_reference(yf, models, golden, dev, fusion):
    """25 intermediate activations recorded from the reference module by forward hooks.  fusion=0: one launch
    per layer, all 25 exist; fusion=1 / 2 (2 = the product default): tensors kept on chip by a fused kernel report
    YF_E_NOPROBE, every other one must match."""
    m, _, _ = models[256]
    g = golden("golden_256")
    x = _x(g["input_u8"][1:2], dev)
    m.fusion = fusion
    bad, seen, fused_away = [], 0, []
    try:
        for k, v in g.items():
            if not k.startswith("probe_"):
                continue
            try:
                got = m.probe(x, k[6:]).cpu().numpy()[0]
            except yf._lib.YFError as e:
                assert "error -5" in str(e) and fusion >= 1, str(e)
                fused_away.append(k[6:])
                continue
            seen += 1
            assert got.shape == v.shape, (k, got.shape, v.shape)
            err = np.abs(got - v).max()
            if not err < 2e-5 * max(1.0, np.abs(v).max()):  # relative to the tensor's range
                bad.append((k, float(err)))
    finally:
        m.fusion = yf.model.DEFAULT_FUSION
    assert not bad, bad
    assert seen == (25 if fusion == 0 else 25 - len(fused_away)) and seen >= 12, (seen, fused_away)


def test_fused_and_per_layer_plans_agree(yf, models, golden, dev, bit_stable):
    m, _, _ = models[256]
    x = _x(golden("golden_256")["input_u8"][:6], dev)
    with torch.no_grad():
        a = m(x)
        try:
            m.fusion = 0
            b = m(x)
            m.fusion = 1
            c = m(x)
        finally:
            m.fusion = yf.model.DEFAULT_FUSION
    # same math, different summation order inside the fused kernels
    assert (a[0] - b[0]).abs().max().item() < 2e-4 and (a[1] - b[1]).abs().max().item() < 2e-4
    # fusion 2 only removes launch boundaries from fusion 1's plan (conv5_2 in the res5 launch, ...): the same MFMAs in the same order,
    # hence the same BITS
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


@pytest.mark.parametrize("prec", ["f32", "f16x3", "f16"])
@pytest.mark.parametrize("res,batch", [(256, 256), (512, 24), (96, 5)])
def test_deep_stage_fusion_is_bitwise_neutral(yf, dev, res, batch, prec, monkeypatch):
    """yf_set_fusion 2 (default) against 1 on noise frames: bitwise equal heads at the metric's size and batch, at 640x512 (where the
    stride-32 tile is a quarter frame, so the chains fall apart into single blocks) and at a ragged size (partial tiles), and fewer
    launches.  The split-operand engines (f16x3) fuse deconv5_1 + conv4_1_1 the same way (dcat_x3_kernel), the fp16-storage engines with
    dcat_h_kernel (the deconv result is rounded to fp16 in registers exactly as the two-launch plan rounds it on its way through HBM)."""
    import ctypes
    H, W = (res, res * 5 // 4) if res != 96 else (96, 160)
    io = dict(yf.io_params_for(256 if res != 512 else 512)); io["input_shape"] = [H, W, 1]
    # f16x3: conv5_2 inside the res5 launch runs on fp32 MFMAs (closer to fp32 than the split-operand launch it replaces, not the same
    # bits), so the bitwise claim is made with that one fusion off (developer switch, read when the plan is built)
    if prec == "f16x3":
        monkeypatch.setenv("YF_DEEP_MASK", "6")
    # fp16 storage: a tensor that stays on chip is not rounded to fp16 (conv5_2 inside the res5 launch, conv5_4 inside the small head's), so
    # the bitwise claim is dcat_h_kernel's alone, which rounds the deconv result where the two launches it replaces stored it
    if prec == "f16":
        monkeypatch.setenv("YF_DEEP_MASK", "4")
    m = yf.YoloFastest(io).to(dev).eval()
    m.precision = prec
    m.load_state_dict(torch.load(WEIGHTS[512 if res == 512 else 256], map_location=dev))
    g = torch.Generator(device="cpu").manual_seed(5)
    x = ((torch.randint(0, 256, (batch, 1, H, W), generator=g).float() - 128.0) / 255.0).to(dev)
    outs, launches = {}, {}
    for lvl in (2, 1):
        m.fusion = lvl
        with torch.no_grad():
            outs[lvl] = m(x)
        n = ctypes.c_int()
        e = m.engine(H, W, batch, dev)
        yf._lib.check(e.lib.yf_num_launches(e.handle, ctypes.byref(n)))
        launches[lvl] = n.value
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    assert launches[2] < launches[1], launches


@pytest.fixture(params=[2, 1], ids=["one-wg-per-frame", "one-wg-per-frame-and-class"])
def post_split(request, models):
    """Both forms of the post-process launch (yf_set_post_split): 2 = one workgroup per frame (post_kernel), 1 = one per frame and class with the
    frame's last workgroup assembling the classes (post_split_kernel, round 6: dense frames).  0 = auto (by kmax) is the default and is restored."""
    for m, _, _ in models.values():
        m.post_split = request.param
    yield request.param
    for m, _, _ in models.values():
        m.post_split = 0


@pytest.mark.parametrize("name", ["golden_256", "golden_512", "golden_dense_256", "golden_dense_512"])
def test_post_process_bit_exact_on_reference_logits(models, golden, dev, name, post_split):
    res = 256 if name.endswith("256") else 512
    _, post, _ = models[res]
    g = golden(name)
    pred = (torch.from_numpy(g["head_large"]).to(dev), torch.from_numpy(g["head_small"]).to(dev))
    models[res][0](_x(np.zeros((1, res, res * 5 // 4), np.uint8), dev))  # make sure the engine exists
    kmax = int(max(g["final_count"].max(), 1))
    raw = post.detect_raw(pred, kmax=kmax)
    counts = raw["counts"].cpu().numpy()
    assert np.array_equal(counts, g["final_count"])
    for f, n in enumerate(counts):
        assert np.array_equal(raw["boxes"][f, :n].cpu().numpy(), g["final_box"][f, :n])
        assert np.array_equal(raw["cls"][f, :n].cpu().numpy(), g["final_cls"][f, :n])
        assert np.array_equal(raw["src"][f, :n].cpu().numpy(), g["final_src"][f, :n])  # survivor ORDER
        sc = raw["scores"][f, :n].cpu().numpy().astype(np.float64)
        assert np.abs(sc[:, 0] - g["final_conf"][f, :n]).max(initial=0) < 1e-6
        assert np.abs(sc[:, 1] - g["final_score"][f, :n]).max(initial=0) < 1e-6
    if "adj_box" in g and res == 256:  # __adjust_coord epilogue
        raw = post.detect_raw(pred, kmax=kmax, origin_shape=(512, 640))
        for f, n in enumerate(counts):
            assert np.array_equal(raw["boxes"][f, :n].cpu().numpy(), g["adj_box"][f, :n])


def test_decode_box_api_matches_reference_candidates(models, golden, dev):
    m, post, _ = models[256]
    for name in ("golden_256", "golden_dense_256"):
        g = golden(name)
        for f in (0, 3):
            pred = (torch.from_numpy(g["head_large"][f:f + 1]).to(dev), torch.from_numpy(g["head_small"][f:f + 1]).to(dev))
            c = post.decode_box(pred)
            n = int(g["cand_count"][f])
            assert len(c) == n
            assert [e[:4] for e in c] == g["cand_box"][f, :n].tolist()
            assert [e[6] for e in c] == g["cand_cls"][f, :n].tolist()
            assert np.allclose([e[4] for e in c], g["cand_conf"][f, :n], atol=1e-6, rtol=0)


def test_nms_api_matches_oracle(models, golden, dev):
    from oracle import post_oracle as po
    m, post, io = models[256]
    g = golden("golden_dense_256")
    heads = (g["head_large"][0], g["head_small"][0])
    cands = po.decode_box(heads, io["anchors"], io["input_shape"][:2], 0.5)
    for cls in range(3):
        L = sorted([c for c in cands if c[6] == cls], key=lambda e: e[4], reverse=True)
        want_in = [list(e) for e in L]
        want = po.nms(want_in, 0.2)
        got_in = [list(e) for e in L]
        got = post.non_maxium_supression(got_in)
        assert got == want
        assert got_in == want_in  # same leftover as the reference's pop loop
    assert post.non_maxium_supression([]) == []


def test_end_to_end_boxes_on_test_data(models, golden, dev):
    """model + post on the 20 bundled frames: same boxes as the reference (a coordinate may differ by 1 px only
    if its pre-round value sits within 1e-3 of a half-integer -- none does on test_data)."""
    for res in (256, 512):
        m, post, io = models[res]
        g = golden(f"golden_{res}")
        with torch.no_grad():
            pred = m(_x(g["input_u8"], dev))
        got = post.detect(pred, with_src=True, origin_shape=(512, 640) if res == 256 else None)
        for f, L in enumerate(got):
            n = int(g["adj_count"][f])
            assert len(L) == n, (res, f)
            assert [e[:4] for e in L] == g["adj_box"][f, :n].tolist(), (res, f)
            assert [e[6] for e in L] == g["adj_cls"][f, :n].tolist()
            assert [e[7] for e in L] == g["adj_src"][f, :n].tolist()
            assert np.allclose([e[4] for e in L], g["adj_conf"][f, :n], atol=1e-4, rtol=0)
            assert np.allclose([e[5] for e in L], g["adj_score"][f, :n], atol=1e-4, rtol=0)


def test_random_batch_against_oracle(models, dev):
    from oracle import backbone_oracle as bo
    m, _, _ = models[256]
    g = np.random.default_rng(7)
    u8 = g.integers(0, 256, size=(5, 256, 320), dtype=np.uint8)
    with torch.no_grad():
        hl, hs = m(_x(u8, dev))
    ol, os_ = bo.forward(bo.load_state_dict(WEIGHTS[256]), bo.preprocess(u8))
    t = _truth64(256, u8)
    _check_heads(hl, hs, ol, os_, t[0], t[1])


def test_other_input_sizes(yf, dev):
    """Any H, W that are multiples of 32 (yolo_fastest.py has no fixed size); ragged vs the shipped shapes."""
    from oracle import backbone_oracle as bo
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev).eval()
    sd = torch.load(WEIGHTS[256], map_location="cpu")
    m.load_state_dict(sd)
    g = np.random.default_rng(3)
    # (160, 224): the stride-16/32 frames fit one tile -> chained residual launches on partial tiles; (288, 320) and (256, 352):
    # one row / column of tiles more than the chain allows -> one launch per block, ragged tiles
    for (H, W, N) in ((32, 32, 3), (64, 96, 2), (96, 32, 1), (160, 224, 2), (288, 320, 2), (256, 352, 1)):
        u8 = g.integers(0, 256, size=(N, H, W), dtype=np.uint8)
        with torch.no_grad():
            hl, hs = m(_x(u8, dev))
        ol, os_ = bo.forward(sd, bo.preprocess(u8))
        t = _truth64(256, u8)
        _check_heads(hl, hs, ol, os_, t[0], t[1])
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 100, 320, device=dev))


def test_chunked_pass_is_identical(models, golden, dev, bit_stable):
    m, _, _ = models[256]
    g = golden("golden_256")
    x = _x(g["input_u8"], dev)
    with torch.no_grad():
        a = m(x)
        m.chunk = 3          # applied to the existing engine on the next call (model.engine re-applies the knobs)
        try:
            b = m(x)
            assert m.engine(256, 320, 20, dev).chunk == 3
        finally:
            m.chunk = 0
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_full_size_batch_properties(models, golden, dev, bit_stable):
    """BASELINE config 2 size (batch 256, 320x256): frames are independent units -- the result for a frame does
    not depend on its position in the batch or on its neighbours (bitwise), and the fixture frames tiled into
    the batch reproduce the golden boxes."""
    m, post, _ = models[256]
    g = golden("golden_256")
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, size=(256, 256, 320), dtype=np.uint8)
    u8[::13][:20] = g["input_u8"][:len(u8[::13][:20])]
    x = _x(u8, dev)
    with torch.no_grad():
        hl, hs = m(x)
        perm = torch.from_numpy(rng.permutation(256)).to(dev)
        hl2, hs2 = m(x[perm])
        hl1, hs1 = m(x[39:40])
    assert torch.equal(hl[perm], hl2) and torch.equal(hs[perm], hs2)
    assert torch.equal(hl[39:40], hl1) and torch.equal(hs[39:40], hs1)
    got = post.detect((hl, hs), with_src=True)
    k = 0
    for f in range(0, 256, 13):
        if k >= 20:
            break
        n = int(g["final_count"][k])
        assert [e[:4] for e in got[f]] == g["final_box"][k, :n].tolist()
        assert [e[7] for e in got[f]] == g["final_src"][k, :n].tolist()
        k += 1


def test_preprocess_u8_matches_reference_arithmetic(yf, models, golden, dev):
    m, _, io = models[256]
    from PIL import Image
    names = golden("golden_256")["names"]
    full = np.stack([np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "test_data", str(n)))) for n in names[:4]])
    x = yf.preprocess_u8(m, torch.from_numpy(full).to(dev), io["input_shape"])
    want = _x(golden("golden_256")["input_u8"][:4], dev)
    assert torch.equal(x, want)
    m512, _, io512 = models[512]
    x = yf.preprocess_u8(m512, torch.from_numpy(full).to(dev), io512["input_shape"])
    assert torch.equal(x, _x(full, dev))


def test_post_edge_cases(models, dev):
    """empty frames, threshold strictness at logit 0, ties in decode order, capacity overflow reporting."""
    m, post, io = models[256]
    m(_x(np.zeros((1, 256, 320), np.uint8), dev))
    hl = torch.full((3, 24, 16, 20), -5.0, device=dev)
    hs = torch.full((3, 24, 8, 10), -5.0, device=dev)
    hl[1, 4, 3, 3] = 0.0        # conf == 0.5 exactly: rejected (strict >)
    hl[2, 4, 5, 5] = 1.0; hl[2, 4, 5, 6] = 1.0  # tie, overlapping, same class -> first in decode order wins
    hl[2, 2, 5, 5] = 1.5; hl[2, 2, 5, 6] = 1.5; hl[2, 3, 5, 5] = 1.5; hl[2, 3, 5, 6] = 1.5
    got = post.detect((hl, hs), with_src=True)
    assert got[0] == [] and got[1] == []
    assert [e[7] for e in got[2]] == [5 * 20 + 5]
    # overflow: kmax smaller than the survivor count -> true count reported, host raises
    hl2 = torch.full((1, 24, 16, 20), -5.0, device=dev)
    hl2[0, 0:4] = 0.0  # w,h = anchor (no zero-area boxes)
    hl2[0, 4, ::4, ::4] = 3.0
    raw = post.detect_raw((hl2, hs[:1]), kmax=2)
    assert int(raw["counts"][0]) == 20
    with pytest.raises(OverflowError):
        post.to_lists(raw)


def test_conf_ties_above_logit_22_follow_the_reference(models, dev):
    """The reference sorts a class by the fp64 conf = 1 / (1 + exp(-t)), stable (detect.py:23-25, :167): two DIFFERENT fp32 logits
    share one conf above t ~ 22 (and every logit above 36.7 gives conf == 1.0), and such ties keep decode order.  Sorting by the logit
    would put the later, larger logit first and NMS would keep the other box; the sort key is built from the conf's own fp64 value
    there (conf_order(), yf_post_kernels.hip).  Checked against the oracle's Python doubles: survivors and their order."""
    import math
    from oracle import post_oracle as po
    m, post, io = models[256]
    m(_x(np.zeros((1, 256, 320), np.uint8), dev))
    hl = np.full((2, 24, 16, 20), -5.0, np.float32)
    hs = np.full((2, 24, 8, 10), -5.0, np.float32)
    for a in range(3):
        hl[:, 8 * a:8 * a + 4] = 0.0; hs[:, 8 * a:8 * a + 4] = 0.0      # box = cell centre, anchor size (no zero-area boxes)
    hl[:, 2:4] = 1.2                                                    # large head, anchor 0: 3.3 x the anchor -> neighbours overlap
    up = lambda v, n=1: np.nextafter(np.float32(v), np.float32(100.0)) if n == 1 else up(up(v), n - 1)   # noqa: E731
    # frame 0, class 0 (channel 5 is the largest class logit everywhere: all -5, first maximum wins)
    t24, t24b = np.float32(24.0), up(24.0)
    assert t24b > t24 and 1 / (1 + math.exp(-float(t24))) == 1 / (1 + math.exp(-float(t24b)))       # the tie exists in the reference
    hl[0, 4, 5, 5], hl[0, 4, 5, 6] = t24, t24b                 # overlapping neighbours: the reference keeps (5,5), a logit sort (5,6)
    hl[0, 4, 9, 3], hl[0, 4, 9, 4] = 23.0, 25.0                # distinct confs: the later one leads
    hl[0, 4, 12, 10], hl[0, 4, 12, 11] = 40.0, 50.0            # both conf == 1.0 exactly: decode order
    hl[0, 4, 2, 15], hl[0, 4, 2, 16] = up(30.0, 3), 30.0       # a tie the other way round (already in decode order)
    # frame 1: the same on the small head and another anchor, class 2
    hs[1, 8 + 7] = 3.0
    hs[1, 8 + 4, 3, 3], hs[1, 8 + 4, 3, 4] = np.float32(26.5), up(26.5)
    hs[1, 8 + 4, 6, 7], hs[1, 8 + 4, 6, 8] = 37.0, 88.0
    got = post.detect((torch.from_numpy(hl).to(dev), torch.from_numpy(hs).to(dev)), with_src=True)
    for f in range(2):
        want = po.post_process((hl[f], hs[f]), io["anchors"], io["input_shape"][:2])
        assert [e[:4] + [e[6], e[7]] for e in got[f]] == [list(e[:4]) + [e[6], e[7]] for e in want], (f, got[f], want)
        assert len(want) >= 2
    src0 = [e[7] for e in got[0]]
    assert src0[0] == 12 * 20 + 10 and 12 * 20 + 11 not in src0      # conf == 1.0 twice: the first in decode order leads and suppresses the other
    assert 5 * 20 + 5 in src0 and 5 * 20 + 6 not in src0              # equal conf at 24.0 / 24.0 + 1 ulp: (5, 5) survives, not the larger logit


def test_int32_saturation_of_box_corners(models, dev):
    """A box whose exp(t_w) * anchor leaves int32: the reference's corners are unbounded Python ints (round() of a double, detect.py:65),
    the device stores int32 and SATURATES.  Pinned: the saturated box equals the reference's clamped to int32, and nothing else in the
    frame changes (needs |t_w| > ~16; never seen on real frames)."""
    from oracle import post_oracle as po
    m, post, io = models[256]
    m(_x(np.zeros((1, 256, 320), np.uint8), dev))
    hl = np.full((1, 24, 16, 20), -5.0, np.float32)
    hs = np.full((1, 24, 8, 10), -5.0, np.float32)
    for a in range(3):
        hl[:, 8 * a:8 * a + 4] = 0.0
    hl[0, 4, 4, 4] = 2.0; hl[0, 2, 4, 4] = 25.0; hl[0, 3, 4, 4] = 0.5; hl[0, 6, 4, 4] = 1.0     # class 1: w = e^25 * anchor_w ~ 1e12 px
    hl[0, 8 + 4, 7, 7] = 3.0; hl[0, 8 + 3, 7, 7] = 30.0; hl[0, 8 + 7, 7, 7] = 1.0             # class 2: h overflows
    hl[0, 4, 10, 10] = 1.0; hl[0, 4, 10, 12] = 0.5                                           # class 0: ordinary boxes
    got = post.detect((torch.from_numpy(hl).to(dev), torch.from_numpy(hs).to(dev)), with_src=True)[0]
    want = po.post_process((hl[0], hs[0]), io["anchors"], io["input_shape"][:2])
    lo, hi = -2 ** 31, 2 ** 31 - 1
    clamp = lambda v: max(lo, min(hi, v))   # noqa: E731
    assert len(want) == len(got) == 4
    assert any(abs(v) > 2 ** 31 for e in want for v in e[:4])          # the reference really leaves int32 here
    for g_, w_ in zip(got, want):
        assert g_[:4] == [clamp(v) for v in w_[:4]] and g_[6] == w_[6] and g_[7] == w_[7], (g_, w_)
    differ = [i for i, (g_, w_) in enumerate(zip(got, want)) if g_[:4] != list(w_[:4])]
    assert len(differ) == 2 and all(want[i][6] in (1, 2) for i in differ)


@pytest.mark.parametrize("res", [256, 512])
def test_weights_from_ncnn_files(yf, golden, dev, res):
    """The engine fed from the reference's ncnn .param/.bin (both shipped sizes) instead of the .pth: same heads as the reference."""
    io = yf.io_params_for(res)
    size = "256x320" if res == 256 else "512x640"
    m = yf.YoloFastest(io).eval().load_ncnn(os.path.join(ROOT, "tests", "golden", "ncnn", f"yolo_fastest_{size}.param"),
                                            os.path.join(ROOT, "tests", "golden", "ncnn", f"yolo_fastest_{size}.bin")).to(dev)
    g = golden(f"golden_{res}")
    with torch.no_grad():
        hl, hs = m(_x(g["input_u8"], dev))
    _check_heads(hl, hs, g["head_large"], g["head_small"], g["head_large_f64"], g["head_small_f64"], res == 256)


@pytest.mark.parametrize("res", [256, 512])
def test_weights_from_onnx_files(yf, models, golden, dev, res):
    """... and from the shipped ONNX exports: the state-dict they hold is the .pth's, so the heads are the .pth engine's bit for bit."""
    io = yf.io_params_for(res)
    size = "256x320" if res == 256 else "512x640"
    m = yf.YoloFastest(io).eval().load_onnx(os.path.join(ROOT, "tests", "golden", "onnx", f"yolo_fastest_{size}.onnx")).to(dev)
    g = golden(f"golden_{res}")
    x = _x(g["input_u8"][:6], dev)
    with torch.no_grad():
        a, b = m(x), models[res][0](x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_validation_path_decode_and_nms(yf, models, golden, dev):
    """YOLOLossV3 decode branch + general.non_max_suppression (the +1 IoU convention) on the GPU.
    Decode: fp32 transcendental functions differ from torch-CPU in the last bits -> 2e-6 relative.  NMS: only fp32
    add/mul/div after that, so fed the reference's own decode tensor it must reproduce the reference bit for bit."""
    from yolo_fastest_amd import validation as val
    m, _, io = models[256]
    m(_x(np.zeros((1, 256, 320), np.uint8), dev))
    val.bind(m)
    gv = golden("golden_val_256")
    for tag, src in (("real", golden("golden_256")), ("dense", golden("golden_dense_256"))):
        pred = (torch.from_numpy(src["head_large"].copy()).to(dev), torch.from_numpy(src["head_small"].copy()).to(dev))
        dec = torch.cat([val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev)(pred[i]) for i in range(2)], 1)
        want = gv[f"{tag}_decode"]
        got = dec[:4].cpu().numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
        # NMS on the reference's own decode tensor (the stored golden, 4 frames per set): exact
        dets = val.non_max_suppression(torch.from_numpy(want.copy()).to(dev), 3, conf_thres=0.5, nms_thres=0.2)
        for f, d in enumerate(dets):
            n = int(gv[f"{tag}_count"][f])
            assert (0 if d is None else d.shape[0]) == n, (tag, f)
            if n:
                assert np.array_equal(d.cpu().numpy(), gv[f"{tag}_det"][f, :n]), (tag, f)
        # end to end (own decode): same survivors, boxes to fp32 rounding of the decode
        dets = val.non_max_suppression(dec, 3, conf_thres=0.5, nms_thres=0.2)
        if tag == "real":
            for f, d in enumerate(dets):
                n = int(gv["real_count"][f])
                assert d is not None and d.shape[0] == n
                assert np.allclose(d.cpu().numpy(), gv["real_det"][f, :n], rtol=2e-6, atol=2e-5)
    with pytest.raises(ValueError):       # targets must be [batch, T, 6] (the training loss: test_training_loss_matches_the_reference)
        val.YOLOLossV3(io["anchors"][0], 3, io["input_shape"], dev)(pred[0], targets=torch.zeros(1))


def test_forward_u8_fused_preprocess_is_bit_identical(yf, models, golden, dev):
    """(u8-128)/255 and the exact-2x box mean fused into the stem kernel's loads."""
    from PIL import Image
    names = golden("golden_256")["names"]
    full = np.stack([np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "test_data", str(n)))) for n in names[:5]])
    full_t = torch.from_numpy(full).to(dev)
    for res in (256, 512):
        m, post, io = models[res]
        with torch.no_grad():
            a = m(yf.preprocess_u8(m, full_t, io["input_shape"]))
        b = m.forward_u8(full_t, io["input_shape"])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # any other source size (rounds 1-4 raised here): cv2.resize's INTER_LINEAR as one extra pass (tests/test_gpu_cv_preprocess.py holds
    # it to the oracle), then the same fused entry -- bitwise the two-step path; the float-output entry still takes 1x / 2x only
    m256 = models[256][0]
    odd = full_t[:, :300, :300].contiguous()
    two_step = m256.forward_u8(m256.cv_preprocess_u8(odd, (256, 320)), (256, 320))
    one_call = m256.forward_u8(odd, (256, 320))
    assert torch.equal(one_call[0], two_step[0]) and torch.equal(one_call[1], two_step[1])
    with pytest.raises(yf._lib.YFError):
        yf.preprocess_u8(m256, odd, (256, 320))
    # Detect_YOLO's batched entry on the bundled frames reproduces the reference's boxes in original coordinates
    import logging
    cfg = {"io_params": yf.io_params_for(256)}
    det = yf.Detect_YOLO(dev, WEIGHTS[256], cfg, logging.getLogger("yf-test"))
    g = golden("golden_256")
    got = det.detect_u8(torch.from_numpy(np.stack([np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "test_data", str(n))))
                                                   for n in names])).to(dev))
    for f, L in enumerate(got):
        n = int(g["adj_count"][f])
        assert [e[:4] for e in L] == g["adj_box"][f, :n].tolist() and [e[6] for e in L] == g["adj_cls"][f, :n].tolist()


@pytest.mark.parametrize("lanes,detect", [(1, False), (2, False), (2, True), (3, True)])
def test_forward_is_hip_graph_capturable(yf, golden, dev, lanes, detect):
    """yf_forward / yf_detect issue launches and event fork/join on the engine's side streams only: a pass can be captured into a HIP
    graph and replayed bit-identically -- with several lanes AND the small head's branch stream on (the default), forward only and
    with each chunk's decode + NMS behind it.  The engine issues the same stream topology eager and captured; it is one that the HIP
    runtime bundled with PyTorch can capture (a branch never joins back into a forked lane: tools/cap_repro.hip)."""
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev).eval()
    m.lanes = lanes
    assert m.branches == 1
    m.chunk = (64 + lanes - 1) // lanes if lanes > 1 else 0   # one chunk per lane (the automatic split starts at larger passes)
    m.load_state_dict(torch.load(WEIGHTS[256], map_location=dev))
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    x = _x(np.tile(golden("golden_256")["input_u8"], (4, 1, 1))[:64], dev)

    def run():
        if detect:
            r = post.detect_raw_from_input(x, kmax=16)
            return [r["head_large"], r["head_small"], r["counts"], r["boxes"], r["cls"], r["src"]]
        return list(m(x))

    with torch.no_grad():
        ref = [t.clone() for t in run()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        out = run()
    for t in out:
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    if detect:
        assert torch.equal(out[2], ref[2]) and int(ref[2].sum()) > 64
        valid = torch.arange(16, device=dev)[None, :] < ref[2][:, None]
        for a, b in zip(out[3:], ref[3:]):
            assert torch.equal(a[valid], b[valid])


def test_yf_detect_single_call_equals_two_calls(models, golden, dev, bit_stable):
    m, post, io = models[256]
    x = _x(golden("golden_256")["input_u8"], dev)
    with torch.no_grad():
        pred = m(x)
    a = post.detect_raw(pred, kmax=16, origin_shape=(512, 640))
    b = post.detect_raw_from_input(x, kmax=16, origin_shape=(512, 640))
    assert torch.equal(b["head_large"], pred[0]) and torch.equal(b["head_small"], pred[1])
    n = a["counts"].cpu().numpy()
    assert np.array_equal(n, b["counts"].cpu().numpy())
    for f, k in enumerate(n):
        for key in ("boxes", "scores", "cls", "src"):
            assert torch.equal(a[key][f, :k], b[key][f, :k]), key
    # chunks of 7 frames on two lanes: yf_detect runs each chunk's decode + NMS on the chunk's own stream
    m.chunk = 7
    try:
        c = post.detect_raw_from_input(x, kmax=16, origin_shape=(512, 640))
        assert m.engine(256, 320, 20, dev).chunk == 7
    finally:
        m.chunk = 0
    assert np.array_equal(n, c["counts"].cpu().numpy())
    for f, k in enumerate(n):
        for key in ("boxes", "scores", "cls", "src"):
            assert torch.equal(a[key][f, :k], c[key][f, :k]), key


# ---- BASELINE configs[2]: the fp16 MFMA pointwise path (parity target, SURVEY.md 8(d).3: 2e-2 on logits vs the fp32 reference) ----
# Two variants (include/yolo_fastest_hip.h, yf_create_ex dtype):
#   "f16x3" -- fp32 storage, every MFMA operand split into two fp16 halves (hi + lo), three fp16 MFMAs per k-group, fp32 accumulate.
#              THIS is the variant that meets the stated 2e-2 -- by two orders of magnitude: it is held to the SAME bounds as the
#              fp32 path (_check_heads: same accuracy class as the reference's own fp32 evaluation) and gives identical boxes.
#   "f16"   -- fp16 storage + single fp16 operands: the fastest, and 2e-2 is NOT reachable with it (tools/fp16_sim.py: rounding the
#              weights alone to fp16 moves the 640x512 logits by 4.5e-2, the activations by another 4e-2); its test below states
#              the bounds it does meet.  Treat it as the throughput variant; detections on the bundled frames are still identical.


@pytest.mark.parametrize("res", [512, 256])
def test_f16x3_path_meets_the_fp32_bounds(yf, golden, dev, res):
    io = yf.io_params_for(res)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(torch.load(WEIGHTS[res], map_location=dev))
    m.precision = "f16x3"
    g = golden(f"golden_{res}")
    with torch.no_grad():
        hl, hs = m(_x(g["input_u8"], dev))
    for got, ref in ((hl.cpu().numpy(), g["head_large"]), (hs.cpu().numpy(), g["head_small"])):
        assert np.abs(got - ref).max() <= 2e-2                     # SURVEY.md 8(d).3, as stated
    _check_heads(hl, hs, g["head_large"], g["head_small"], g["head_large_f64"], g["head_small_f64"], res == 256)   # and the fp32 path's own bounds
    e = m.engine(res, res * 5 // 4, 20, dev)
    kinds = [o["kernel_dtype"] for o in m.profile(_x(g["input_u8"][:2], dev), reps=1)]
    assert kinds.count("f16x3") >= len(kinds) - 5 and "f16" not in kinds, kinds     # only the four small-channel VALU launches run fp32
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    got = post.detect((hl, hs), with_src=True)
    for f, L in enumerate(got):
        n = int(g["final_count"][f])
        assert [e[:4] for e in L] == g["final_box"][f, :n].tolist(), (res, f)
        assert [e[7] for e in L] == g["final_src"][f, :n].tolist() and [e[6] for e in L] == g["final_cls"][f, :n].tolist()
        assert np.allclose([e[4] for e in L], g["final_conf"][f, :n], atol=1e-4, rtol=0)
    with torch.no_grad():
        hl, hs = m(_x(g["syn_input_u8"], dev))
    t = _truth64(res, g["syn_input_u8"])
    _check_heads(hl, hs, g["syn_head_large"], g["syn_head_small"], t[0], t[1], res == 256)


def test_f16x3_other_sizes_random_weights_and_u8(yf, dev):
    """The split-operand kernels on partial tiles, un-chained residual blocks (sizes whose stride-16/32 frame is not one tile), a
    random state-dict, and through the fused u8 pre-process: same bounds as the fp32 path."""
    from oracle import backbone_oracle as bo
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev).eval()
    sd = torch.load(WEIGHTS[256], map_location="cpu")
    m.load_state_dict(sd)
    m.precision = "f16x3"
    g = np.random.default_rng(5)
    for (H, W, N) in ((32, 32, 3), (64, 96, 2), (160, 224, 2), (288, 320, 2), (256, 352, 1)):
        u8 = g.integers(0, 256, size=(N, H, W), dtype=np.uint8)
        with torch.no_grad():
            hl, hs = m(_x(u8, dev))
            ul, us = m.forward_u8(torch.from_numpy(u8).to(dev), (H, W))
        ol, os_ = bo.forward(sd, bo.preprocess(u8))
        t = _truth64(256, u8)
        _check_heads(hl, hs, ol, os_, t[0], t[1])
        assert torch.equal(ul, hl) and torch.equal(us, hs)


# Absolute bounds of the fp16-STORAGE variant, per configuration, next to the survey's figure.  SURVEY.md 8(d).3 asks 2e-2 on logits for
# "activations/weights fp16, fp32 accumulate".  That figure cannot be met by ANY implementation of that arithmetic on these checkpoints:
# the logits reach +-36, where ONE fp16 ulp is 3.1e-2, and tests/test_oracle_golden.py::test_fp16_rounding_alone_exceeds_the_surveys_tolerance
# shows on the CPU (torch fp32 with a single class of tensors rounded to fp16, nothing of this library involved) that rounding the weights
# ALONE moves the 640x512 logits by 4.5e-2, the tensors between launches alone by 8e-2, the MFMA activation operands alone by 4e-2.
# So `f16` is NON-CONFORMING to 8(d).3 by arithmetic, `f16x3` (above) is the conforming configs[2] variant, and `f16` is held to what
# fp16 storage can deliver: the measured maxima (4.2e-2 / 8.5e-2) with a margin, absolute.
SURVEY_FP16_TOL = 2e-2
FP16_STORAGE_MAX = {256: 6e-2, 512: 1.2e-1}      # max |logit - reference fp32 logit| over all cells of the 20 bundled frames
FP16_STORAGE_P99 = {256: 2e-2, 512: 3e-2}
FP16_STORAGE_MEAN = {256: 3e-3, 512: 7e-3}


@pytest.mark.parametrize("res", [512, 256])
def test_fp16_storage_variant_is_nonconforming_but_detects_identically(yf, golden, dev, res):
    """dtype 1 (`model.half()` / storage_dtype = float16): fp16 in HBM, single fp16 MFMA operands, fp32 accumulation -- the throughput
    variant.  NON-CONFORMING to SURVEY.md 8(d).3's 2e-2 (see the constants above: no fp16-storage arithmetic can conform); held to stated
    absolute bounds, to score bounds, and to DETECTIONS IDENTICAL to the fp32 path's and the reference's on the 20 bundled frames."""
    io = yf.io_params_for(res)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(torch.load(WEIGHTS[res], map_location=dev))
    g = golden(f"golden_{res}")
    with torch.no_grad():
        f32_heads = [t.clone() for t in m(_x(g["input_u8"], dev))]
    m.storage_dtype = torch.float16
    with torch.no_grad():
        hl, hs = m(_x(g["input_u8"], dev))
    assert hl.dtype == torch.float32
    worst = 0.0
    for got, ref in ((hl.cpu().numpy(), g["head_large"]), (hs.cpu().numpy(), g["head_small"])):
        d = np.abs(got - ref)
        worst = max(worst, float(d.max()))
        assert d.max() <= FP16_STORAGE_MAX[res], (d.max(), np.abs(ref).max())
        assert np.quantile(d, 0.99) <= FP16_STORAGE_P99[res] and d.mean() <= FP16_STORAGE_MEAN[res], (np.quantile(d, 0.99), d.mean())
    assert worst > SURVEY_FP16_TOL, worst    # it really is outside the survey's figure (if this ever fails, rename the test and conform)
    assert _score_err(hl.cpu().numpy(), g["head_large"]) < 2.5e-2 and _score_err(hs.cpu().numpy(), g["head_small"]) < 2.5e-2
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    got = post.detect((hl, hs), with_src=True)
    # same detections as the reference (cell and class identical; corners within 1 px of the fp32 result)
    for f, L in enumerate(got):
        n = int(g["final_count"][f])
        assert [e[7] for e in L] == g["final_src"][f, :n].tolist(), (res, f)
        assert [e[6] for e in L] == g["final_cls"][f, :n].tolist()
        assert np.abs(np.array([e[:4] for e in L]).reshape(-1, 4) - g["final_box"][f, :n]).max(initial=0) <= 1
    # ... and as our own fp32 path on the same frames (cell, class, order; corners within 1 px)
    m.storage_dtype = torch.float32
    want = post.detect(tuple(f32_heads), with_src=True)
    for f, (L, Wl) in enumerate(zip(got, want)):
        assert [(e[6], e[7]) for e in L] == [(e[6], e[7]) for e in Wl], (res, f)
        assert np.abs(np.array([e[:4] for e in L]).reshape(-1, 4) - np.array([e[:4] for e in Wl]).reshape(-1, 4)).max(initial=0) <= 1
    # .half() like a reference module: half in, half out
    mh = yf.YoloFastest(io).to(dev).eval()
    mh.load_state_dict(torch.load(WEIGHTS[res], map_location=dev))
    mh = mh.half()
    with torch.no_grad():
        hh = mh(_x(g["input_u8"][:2], dev).half())
    assert hh[0].dtype == torch.float16
    # here the BN statistics themselves were rounded to fp16 by .half() (as they would be in the reference module)
    assert np.abs(hh[0].float().cpu().numpy() - g["head_large"][:2]).max() < 1.5


def test_fp16_storage_both_forms_of_the_stride2_block(yf, golden, dev, tmp_path):
    """The fp16-storage plan runs conv1_8 + conv1_9 + conv2_1 as k19h_kernel (no region buffers, conv1_8 per tap on 4x4x4 fp16 MFMAs with its
    weights rounded to fp16); YF_K19R=0 selects k19m_kernel<half_t> (rounds 2-4: conv1_8 on exact fp32 MFMAs, once per input pixel).
    Both in child processes (the switch is read once per process) on the bundled frames and on a 192 x 224 noise batch: each stays inside the
    fp16-storage bounds, and on the frames they differ from each other by no more than fp16 storage may cost against fp32."""
    import subprocess
    import sys
    code = ("import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
            "import yolo_fastest_amd as yf\n"
            "dev = torch.device('cuda:0'); io = yf.io_params_for(256)\n"
            "m = yf.YoloFastest(io).to(dev).eval(); m.load_state_dict(torch.load(%r, map_location=dev)); m.storage_dtype = torch.float16\n"
            "g = np.load(%r); x = torch.from_numpy(((g['input_u8'].astype(np.float32) - 128.0) / 255.0)[:, None]).to(dev)\n"
            "torch.manual_seed(11); y = torch.rand(3, 1, 192, 224, device=dev) - 0.5\n"
            "with torch.no_grad(): a = m(x); b = m(y)\n"
            "np.savez(sys.argv[1], hl=a[0].cpu().numpy(), hs=a[1].cpu().numpy(), nl=b[0].cpu().numpy(), ns=b[1].cpu().numpy())\n"
            "print('RESULT ok')\n") % (ROOT, WEIGHTS[256], os.path.join(ROOT, "tests", "golden", "golden_256.npz"))
    out = {}
    for old in (False, True):
        env = dict(os.environ)
        env.pop("YF_K19R", None)
        env.pop("YF_K19H_FORM", None)
        if old:
            env["YF_K19R"] = "0"
        f = str(tmp_path / ("old.npz" if old else "new.npz"))
        r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "RESULT ok" in r.stdout, r.stderr[-2000:]
        out[old] = np.load(f)
    g = golden("golden_256")
    for k, ref in (("hl", g["head_large"]), ("hs", g["head_small"])):
        for old in (False, True):
            assert np.abs(out[old][k] - ref).max() <= FP16_STORAGE_MAX[256], (k, old)
        # (two roundings of the same tensor differ by one fp16 ulp -- 1.6e-2 .. 3.1e-2 at the largest activations -- wherever a value sits at a tie)
        assert np.abs(out[False][k] - out[True][k]).max() <= FP16_STORAGE_MAX[256], k
    assert any(not np.array_equal(out[False][k], out[True][k]) for k in ("hl", "hs"))      # two different kernels did run
    for k in ("nl", "ns"):      # noise drives the logits far beyond the frames' (one fp16 ulp at 40 is 3e-2): relative to the largest logit
        assert np.isfinite(out[False][k]).all() and np.abs(out[False][k] - out[True][k]).max() <= 6e-3 * np.abs(out[True][k]).max(), k


def test_f16x3_both_forms_of_the_stride2_block(yf, golden, dev, tmp_path):
    """Round 6: the split-operand (f16x3) plan runs conv1_8 + conv1_9 + conv2_1 as k19m_kernel<x3_t> (region buffers, split once per region pixel);
    YF_K19X=43 selects k19x_kernel (no region buffers: conv1_8 exact fp32 per tap, ONE hi/lo split per tap in registers, six K = 32 fp16 MFMAs per
    tap -- built for VERDICT r5 item 3, measured a tie, kept as the A/B form).  Both in child processes (the switch is read once per process) on the bundled frames, at 640x512 and on a 192 x 224 noise batch: each
    stays inside the fp32 golden bounds (_check_heads: the bounds of the fp32 path, SURVEY's 2e-2 by a factor of 100), and they agree with each other
    like two fp32 evaluations do."""
    import subprocess
    import sys
    code = ("import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
            "import yolo_fastest_amd as yf\n"
            "dev = torch.device('cuda:0'); res = int(sys.argv[2]); io = yf.io_params_for(res)\n"
            "m = yf.YoloFastest(io).to(dev).eval(); m.load_state_dict(torch.load(sys.argv[3], map_location=dev)); m.precision = 'f16x3'\n"
            "g = np.load(sys.argv[4]); x = torch.from_numpy(((g['input_u8'].astype(np.float32) - 128.0) / 255.0)[:, None]).to(dev)\n"
            "torch.manual_seed(11); y = torch.rand(3, 1, 192, 224, device=dev) - 0.5\n"
            "with torch.no_grad(): a = m(x); b = m(y)\n"
            "np.savez(sys.argv[1], hl=a[0].cpu().numpy(), hs=a[1].cpu().numpy(), nl=b[0].cpu().numpy(), ns=b[1].cpu().numpy())\n"
            "print('RESULT ok')\n") % ROOT
    for res in (256, 512):
        out = {}
        for old in (False, True):
            env = dict(os.environ)
            for k in ("YF_K19R", "YF_K19X"):
                env.pop(k, None)
            env["YF_K19X"] = "0" if old else "43"
            f = str(tmp_path / ("old_%d.npz" % res if old else "new_%d.npz" % res))
            r = subprocess.run([sys.executable, "-c", code, f, str(res), WEIGHTS[res], os.path.join(ROOT, "tests", "golden", f"golden_{res}.npz")],
                               env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "RESULT ok" in r.stdout, r.stderr[-2000:]
            out[old] = np.load(f)
        g = golden(f"golden_{res}")
        for old in (False, True):
            _check_heads(out[old]["hl"], out[old]["hs"], g["head_large"], g["head_small"], g["head_large_f64"], g["head_small_f64"], res == 256)
        assert any(not np.array_equal(out[False][k], out[True][k]) for k in ("hl", "hs"))      # two different kernels did run
        for k in ("hl", "hs", "nl", "ns"):
            assert np.isfinite(out[False][k]).all() and np.abs(out[False][k] - out[True][k]).max() <= 2e-5 * max(1.0, np.abs(out[True][k]).max()), (res, k)


def test_validation_get_map_end_to_end(yf, models, golden, dev):
    """`Validation.get_mAP` (validate.py:27-122) with the model, the decode and the NMS on the GPU against the
    reference's own run on the same frames and synthetic targets (tests/golden/make_golden.py main_map).  The TP / FP
    decisions and the per-class target counts must be the reference's; AP within 2e-3: the reference orders a class's matches
    by the confidence PRINTED to 4 decimals, so last-bit differences between two fp32 evaluations of the net (and its own
    shuffle order among equal printed values) can swap neighbours."""
    import logging
    from yolo_fastest_amd import validation as V
    m, _, io = models[256]
    g, gm = golden("golden_256"), golden("golden_map_256")
    u8, targets = g["input_u8"], gm["targets"]

    class Frames(torch.utils.data.Dataset):   # what the reference's DetectDataset yields; collate_fn divides by 255
        def __len__(self): return len(u8)
        def __getitem__(self, i): return u8[i][:, :, None].astype(np.float32) - 128.0, targets[i].copy()

    params = {"train_params": {"batch_size": 4, "IOU_val_thre": 0.5},
              "io_params": dict(io, class_names=["carrier", "defender", "destroyer"])}
    losses = [V.YOLOLossV3(io["anchors"][i], io["num_cls"], io["input_shape"], dev) for i in range(2)]
    torch.manual_seed(0)
    val = V.Validation(params, logging.getLogger("yf-val"), Frames(), dev, losses)
    mAP = float(val.get_mAP(m, 0))
    assert val.target_num.tolist() == gm["target_num"].tolist()
    for c in range(3):
        assert len(val.match_list[c]) == len(gm[f"match_tp_{c}"])
        assert sum(t for _, t in val.match_list[c]) == int(gm[f"match_tp_{c}"].sum())
        assert abs(float(val._calculate_AP(c)) - gm["AP"][c]) < 2e-3
    assert abs(mAP - float(gm["mAP"])) < 2e-3, (mAP, float(gm["mAP"]))


def test_rccl_gather_path_single_rank(models, golden, dev):
    """the exchange step on the real backend.  The box has one GPU, so this is a 1-rank RCCL group: it exercises
    exactly the calls bench.py / dist.all_gather_detections make with backend "nccl" (init with device_id, all_gather_into_tensor of
    the packed int32 records) -- the 2-rank semantics (padding, frame order) are covered on gloo in tests/test_dist_gloo.py."""
    import torch.distributed as dist
    from yolo_fastest_amd import dist as yfd
    m, post, _ = models[256]
    g = golden("golden_256")
    pred = (torch.from_numpy(g["head_large"]).to(dev), torch.from_numpy(g["head_small"]).to(dev))
    m(_x(np.zeros((1, 256, 320), np.uint8), dev))
    raw = post.detect_raw(pred, kmax=8)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29537", rank=0, world_size=1, device_id=dev)
    try:
        out = yfd.all_gather_detections(raw, raw["counts"].shape[0])
        h = yfd.all_gather_detections_async(raw, raw["counts"].shape[0])   # the overlapped form
        out2 = h.wait()
        assert all(torch.equal(out[k], out2[k]) for k in out)
        torch.cuda.synchronize(dev)
        # ... and started from inside a batch's stream context, as bench.py does with two batches in flight
        import yolo_fastest_amd as yf_
        pipe = yf_.BatchPipeline(m, post, depth=2, kmax=8)
        x20 = _x(g["input_u8"], dev)
        REC = ("counts", "boxes", "scores", "cls", "src")
        ts = [pipe.submit(x20, then=lambda o: yfd.all_gather_detections_async({k: o[k] for k in REC}, 20)) for _ in range(3)]
        for t in ts:
            got = t.extra.wait()
            assert torch.equal(got["counts"].cpu(), raw["counts"].cpu())
        pipe.drain()
        torch.cuda.synchronize(dev)
        m.lanes, m.branches = 2, 1
    finally:
        dist.destroy_process_group()
    for k in ("counts", "boxes", "scores", "cls", "src"):
        assert torch.equal(out[k].cpu(), raw[k].cpu()), k


def test_fp16_other_input_sizes_and_u8(yf, dev):
    """fp16 storage at sizes whose tiles are partial (W/4 not a multiple of 16: the matrix-core k19 kernel's border windows and
    guard band, the stride-2 block kernel's ragged tiles), through the fused u8 pre-process too.  Bounds as in
    test_fp16_path_logits_and_boxes (relative to the logit range of the fp32 oracle)."""
    from oracle import backbone_oracle as bo
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev).eval()
    sd = torch.load(WEIGHTS[256], map_location="cpu")
    m.load_state_dict(sd)
    m.storage_dtype = torch.float16
    g = np.random.default_rng(11)
    for (H, W, N) in ((32, 32, 2), (64, 96, 3), (160, 224, 2), (256, 352, 1)):
        u8 = g.integers(0, 256, size=(N, H, W), dtype=np.uint8)
        ol, os_ = bo.forward(sd, bo.preprocess(u8))
        with torch.no_grad():
            hl, hs = m(_x(u8, dev))
            ul, us = m.forward_u8(torch.from_numpy(u8).to(dev), (H, W))
        for got, ref in ((hl, ol), (hs, os_)):
            d = (got.cpu() - ref).abs()
            # 3.5e-3 of the range here (3e-3 in the bundled-frame test): uniform-noise frames drive the logits harder, and the worst
            # of these cases sits at 3.2e-3 (160x224; measured 0.1438 on a range of 44.6)
            assert d.max().item() <= 3.5e-3 * max(1.0, ref.abs().max().item()) + 1e-2, (H, W, d.max().item())
        assert torch.equal(ul, hl) and torch.equal(us, hs)   # the fused u8 load computes the same (x - 128) / 255


@pytest.mark.parametrize("res", [256, 512])
def test_post_process_randomised_against_c_oracle(yf, models, dev, res, post_split):
    """Windowed NMS / 16-wave post kernel (both launch forms: post_split) against oracle/post_oracle.c (the reference's loop, restated) on random logit fields:
    sparse to very dense, thresholds at the extremes (nms_thres 0 and 1, a NEGATIVE one -- where even disjoint boxes suppress each
    other --, conf_thres low), quantised confidences (many exact ties), tiny boxes (zero-area -> the reference's ZeroDivisionError
    path must agree too).  Bit-exact boxes, classes and survivor order."""
    from oracle import post_oracle_c as poc
    m, _, io = models[res]
    H, W = io["input_shape"][:2]
    m(_x(np.zeros((1, H, W), np.uint8), dev))
    cases = [(0.5, 0.2, -1.0, 1.5, 0.5, None), (0.5, 0.0, -1.0, 1.5, 0.5, None), (0.5, 1.0, -2.0, 1.0, 0.5, None),
             (0.5, -0.1, -2.5, 1.0, 0.5, None), (0.2, 0.5, 0.0, 2.0, 1.0, None), (0.5, 0.3, -1.0, 1.5, 0.5, 0.5),
             (0.5, 0.2, -2.0, 1.0, -6.0, None), (0.9, 0.45, 1.0, 2.0, 0.3, 1.0)]
    for ci, (conf, nms, mu, sd, wh_mu, quant) in enumerate(cases):
        hl, hs = [], []
        for f in range(3):
            g = np.random.default_rng(1000 * ci + f)
            for (h, w), dst in (((H // 16, W // 16), hl), ((H // 32, W // 32), hs)):
                t = np.empty((3, 8, h, w), np.float32)
                t[:, 0:2] = g.normal(0.0, 1.0, (3, 2, h, w)); t[:, 2:4] = g.normal(wh_mu, 0.5, (3, 2, h, w))
                t[:, 4] = g.normal(mu, sd, (3, h, w)); t[:, 5:8] = g.normal(0.0, 2.0, (3, 3, h, w))
                if quant:
                    t[:, 4:8] = np.round(t[:, 4:8] / quant) * quant   # exact ties in confidence and class scores
                dst.append(t.reshape(24, h, w))
        post = yf.YOLO_post_process(conf, nms, 3, 3, io["anchors"], io["input_shape"]).bind(m)
        pred = (torch.from_numpy(np.stack(hl)).to(dev), torch.from_numpy(np.stack(hs)).to(dev))
        kmax = 3 * (hl[0].shape[1] * hl[0].shape[2] + hs[0].shape[1] * hs[0].shape[2])
        raw = post.detect_raw(pred, kmax=kmax)
        counts = raw["counts"].cpu().numpy()
        for f in range(3):
            try:
                r = poc.post_process(hl[f], hs[f], io["anchors"], io["input_shape"][:2], conf_thres=conf, nms_thres=nms)
            except ZeroDivisionError:       # the reference raises (detect.py:39): the kernel reports count -2
                assert counts[f] == -2, (res, ci, f, counts[f])
                continue
            assert counts[f] == r["count"], (res, ci, f, counts[f], r["count"], r["n_candidates"])
            n = r["count"]
            if n > 0:
                assert np.array_equal(raw["src"][f, :n].cpu().numpy(), r["src"]), (res, ci, f)
                assert np.array_equal(raw["boxes"][f, :n].cpu().numpy(), r["box"]), (res, ci, f)
                assert np.array_equal(raw["cls"][f, :n].cpu().numpy(), r["cls"]), (res, ci, f)


@pytest.mark.parametrize("N", [10, 64])      # 64 = BASELINE configs[4]'s per-GPU share (512 dense 640x512 frames over 8 GPUs)
def test_dense_post_process_per_class_workgroups_give_the_same_records(yf, models, dev, N):
    """the per-(frame, class) form of the post-process (post_split_kernel; chosen automatically at kmax >= 256) against the
    one-workgroup-per-frame form on the same logits: identical counts, boxes, classes, sources, scores -- on dense synthetic fields (SURVEY.md
    8(d).5's recipe), with a capacity overflow (the count stays the true number, the first kmax records are stored), with the packed record
    block, inside yf_detect on two lanes (each lane's frames use their own scratch rows), and launched repeatedly (the ticket is reset)."""
    m, post, io = models[512]
    H, W = io["input_shape"][:2]
    hl, hs = [], []
    for f in range(N):
        g = np.random.default_rng(500 + f)
        for (h, w), dst in (((H // 16, W // 16), hl), ((H // 32, W // 32), hs)):
            t = np.empty((3, 8, h, w), np.float32)
            t[:, 0:2] = g.normal(0.0, 1.0, (3, 2, h, w)); t[:, 2:4] = g.normal(0.0, 0.5, (3, 2, h, w))
            t[:, 4] = g.normal(-1.0, 1.5, (3, h, w)); t[:, 5:8] = g.normal(0.0, 2.0, (3, 3, h, w))
            if f == 3:
                t[:, 5] += 50.0        # one frame whose candidates all fall into class 0 (two workgroups of the frame find nothing)
            if f == 4:
                t[:, 4] = -30.0        # one frame without candidates
            dst.append(t.reshape(24, h, w))
    pred = (torch.from_numpy(np.stack(hl)).to(dev), torch.from_numpy(np.stack(hs)).to(dev))
    m(_x(np.zeros((1, H, W), np.uint8), dev))
    keys = ("counts", "boxes", "scores", "cls", "src")

    def same(a, b, kmax):
        assert torch.equal(a["counts"], b["counts"])
        valid = torch.arange(kmax, device=dev)[None, :] < a["counts"].clamp(max=kmax)[:, None]
        for k in keys[1:]:
            assert torch.equal(a[k][valid], b[k][valid]), k
    refs = {}
    try:
        for kmax in (1024, 100):             # 100: fewer records than survivors (capacity overflow)
            m.post_split = 2
            ref = refs[kmax] = {k: v.clone() for k, v in post.detect_raw(pred, kmax=kmax).items() if k in keys}
            assert int(ref["counts"].max()) > 200 and int(ref["counts"][4]) == 0
            m.post_split = 1
            for rep in range(3):
                got = post.detect_raw(pred, kmax=kmax)
                same(ref, got, kmax)
            same(ref, post.detect_raw(pred, kmax=kmax, packed=True), kmax)
        m.post_split = 0                     # auto: kmax 1024 takes the per-class form, kmax 100 the per-frame form
        for kmax in (1024, 100):
            same(refs[kmax], post.detect_raw(pred, kmax=kmax), kmax)
        if N == 64:     # ... and a few frames of the full-size share against the reference's loop restated in C (oracle/post_oracle.c)
            from oracle import post_oracle_c as poc
            got = post.detect_raw(pred, kmax=1024)
            for f in (0, 3, 31, 63):
                r = poc.post_process(hl[f], hs[f], io["anchors"], io["input_shape"][:2])
                n = r["count"]
                assert int(got["counts"][f]) == n and np.array_equal(got["src"][f, :n].cpu().numpy(), r["src"])
                assert np.array_equal(got["boxes"][f, :n].cpu().numpy(), r["box"]) and np.array_equal(got["cls"][f, :n].cpu().numpy(), r["cls"])
        # inside yf_detect, two lanes: low thresholds make the shipped model's own logits dense enough to matter
        rng = np.random.default_rng(9)
        x = _x(rng.integers(0, 256, (6, H, W), dtype=np.uint8), dev)
        lowpost = yf.YOLO_post_process(0.001, 0.9, 3, 3, io["anchors"], io["input_shape"]).bind(m)
        m.post_split = 2
        a = {k: v.clone() for k, v in lowpost.detect_raw_from_input(x, kmax=2048).items() if k in keys}
        assert int(a["counts"].min()) > 20
        m.post_split = 1
        for rep in range(2):
            same(a, lowpost.detect_raw_from_input(x, kmax=2048), 2048)
    finally:
        m.post_split = 0


@pytest.mark.parametrize("seed", [0, 1])
def test_random_weights_against_oracle(yf, dev, seed):
    """Not only the two shipped checkpoints: a random 508-key state-dict (He-style conv weights, BN statistics away from the
    identity, negative biases) through every kernel of the fused plan, fp32 and fp16 storage, against the oracle evaluated in
    fp32 and fp64 (same accuracy-class bound as for the shipped weights)."""
    from oracle import backbone_oracle as bo
    from yolo_fastest_amd import packer
    io = yf.io_params_for(256)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, kind, cin, cout, k, stride, relu in packer.layer_table(24, 1):
        if kind == packer.KIND_HEAD:
            sd[name + ".weight"] = torch.randn((cout, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5
            sd[name + ".bias"] = torch.randn((cout,), generator=g) * 0.5
            continue
        shape = {packer.KIND_PW: (cout, cin, 1, 1), packer.KIND_DENSE: (cout, cin, k, k), packer.KIND_DW: (cout, 1, k, k),
                 packer.KIND_DECONV: (cin, cout, 2, 2)}[kind]
        fan = cin * (k * k if kind == packer.KIND_DENSE else 1) if kind != packer.KIND_DW else k * k
        sd[name + ".0.weight"] = torch.randn(shape, generator=g) * (1.0 / fan) ** 0.5   # keeps the 86-layer chain in range
        sd[name + ".1.weight"] = 0.5 + torch.rand((cout,), generator=g)
        sd[name + ".1.bias"] = torch.randn((cout,), generator=g) * 0.2
        sd[name + ".1.running_mean"] = torch.randn((cout,), generator=g) * 0.2
        sd[name + ".1.running_var"] = 0.5 + torch.rand((cout,), generator=g)
        sd[name + ".1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(sd)
    u8 = np.random.default_rng(seed).integers(0, 256, size=(3, 256, 320), dtype=np.uint8)
    with torch.no_grad():
        hl, hs = m(_x(u8, dev))
    ol, os_ = bo.forward(sd, bo.preprocess(u8))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    tl, ts = bo.forward(sd64, bo.preprocess(u8).double())
    _check_heads(hl, hs, ol, os_, tl.numpy(), ts.numpy(), metric_size=False)
    m.precision = "f16x3"       # split-operand fp16 MFMA: the fp32 bounds
    with torch.no_grad():
        xl, xs = m(_x(u8, dev))
    _check_heads(xl, xs, ol, os_, tl.numpy(), ts.numpy(), metric_size=False)
    m.precision = None
    m.storage_dtype = torch.float16
    with torch.no_grad():
        fl, fs = m(_x(u8, dev))
    for got, ref in ((fl, ol), (fs, os_)):
        d = (got.cpu() - ref).abs()
        assert d.max().item() <= 5e-3 * max(1.0, ref.abs().max().item()), d.max().item()


# ---- round 2: full-size 640x512 batch, pinned all-cell score error at 640x512, result writer, two engines in one process ----

# Measured on MI355X (this test prints them; deterministic arithmetic, so the pins are tight): all-cell score error
# |sigmoid(ours) - sigmoid(.)| over ALL cells of the 20 bundled frames at 640x512:
#   head_large: vs the reference's fp32 scores 1.552e-4, vs its graph in fp64 8.76e-5 (the reference's own fp32 vs fp64: 1.147e-4)
#   head_small: vs the reference's fp32 scores 1.788e-4, vs fp64 1.195e-4 (reference: 1.325e-4)
# i.e. on every cell this path is CLOSER to the real-number result than the reference's own fp32 evaluation; the distance to the
# reference's fp32 scores is bounded by the sum of the two.  (Logits: 7.6e-4 / 8.5e-4 vs ref fp32, reference itself 4.9e-4 / 5.7e-4.)
PIN_SCORE_512_VS_REF32 = 1.9e-4
PIN_SCORE_512_VS_F64 = 1.3e-4


def test_all_cell_score_error_at_640x512_is_pinned(models, golden, dev, capsys):
    m, _, _ = models[512]
    g = golden("golden_512")
    with torch.no_grad():
        hl, hs = m(_x(g["input_u8"], dev))
    rows = []
    for name, got in (("head_large", hl.cpu().numpy()), ("head_small", hs.cpu().numpy())):
        e32 = _score_err(got, g[name]); e64 = _score_err(got, g[name + "_f64"]); r = _score_err(g[name], g[name + "_f64"])
        l32 = np.abs(got - g[name]).max(); l64 = np.abs(got - g[name + "_f64"]).max(); lr = np.abs(g[name] - g[name + "_f64"]).max()
        rows.append((name, e32, e64, r, l32, l64, lr))
    with capsys.disabled():
        for name, e32, e64, r, l32, l64, lr in rows:
            print(f"\n[640x512 all-cell, 20 frames] {name}: score err vs ref fp32 {e32:.3e}, vs fp64 {e64:.3e} (reference fp32 vs its fp64 "
                  f"{r:.3e}); logit err vs ref fp32 {l32:.3e}, vs fp64 {l64:.3e} (reference {lr:.3e})")
    for name, e32, e64, r, l32, l64, lr in rows:
        assert e32 <= PIN_SCORE_512_VS_REF32, (name, e32)
        assert e64 <= PIN_SCORE_512_VS_F64, (name, e64)
        assert e64 <= r, (name, e64, r)                        # closer to the real-number result than the reference's own fp32 run


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "f16"])
def test_full_size_640x512_batch128_two_lanes(yf, golden, dev, dtype):
    """BASELINE configs[2] at full size: 640x512, batch 128 -- the size at which the engine splits the batch over its two stream
    lanes by itself (chunk_frames: >= 64 frames of 640x512).  Frames are independent units: bitwise the same result at any position
    in the batch, alone, and on one lane; the 20 bundled frames tiled into the batch reproduce the reference's goldens."""
    io = yf.io_params_for(512)
    g = golden("golden_512")
    rng = np.random.default_rng(1)
    u8 = rng.integers(0, 256, size=(128, 512, 640), dtype=np.uint8)
    slots = np.arange(0, 120, 6)
    u8[slots] = g["input_u8"]
    x = _x(u8, dev)

    def make(lanes):
        m = yf.YoloFastest(io).to(dev).eval()
        m.lanes = lanes
        m.precision = dtype
        m.load_state_dict(torch.load(WEIGHTS[512], map_location=dev))
        return m
    m = make(2)
    with torch.no_grad():
        hl, hs = m(x)
        perm = torch.from_numpy(rng.permutation(128)).to(dev)
        hl2, hs2 = m(x[perm])                       # a frame lands in the other lane's half
        hl1, hs1 = m(x[77:78])
        hl3, hs3 = make(1)(x)                       # single lane, whole batch in one pass
    assert torch.equal(hl[perm], hl2) and torch.equal(hs[perm], hs2)
    assert torch.equal(hl[77:78], hl1) and torch.equal(hs[77:78], hs1)
    assert torch.equal(hl, hl3) and torch.equal(hs, hs3)
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    got = post.detect((hl, hs), with_src=True)
    raw2 = post.detect_raw_from_input(x, kmax=64)    # yf_detect: each lane's decode + NMS on its own stream
    assert torch.equal(raw2["head_large"], hl) and torch.equal(raw2["head_small"], hs)
    got2 = post.to_lists(raw2, with_src=True)
    assert got2 == got
    sl = torch.from_numpy(slots).to(dev)
    if dtype != "f16":
        _check_heads(hl[sl], hs[sl], g["head_large"], g["head_small"], g["head_large_f64"], g["head_small_f64"], metric_size=False)
    for k, f in enumerate(slots):
        n = int(g["final_count"][k])
        assert [e[7] for e in got[f]] == g["final_src"][k, :n].tolist(), (dtype, k)
        assert [e[6] for e in got[f]] == g["final_cls"][k, :n].tolist()
        if dtype != "f16":
            assert [e[:4] for e in got[f]] == g["final_box"][k, :n].tolist()
        else:
            assert np.abs(np.array([e[:4] for e in got[f]]).reshape(-1, 4) - g["final_box"][k, :n]).max(initial=0) <= 1


@pytest.mark.parametrize("res", [256, 512])
def test_batch_detect_writes_the_references_results(yf, golden, dev, res, tmp_path):
    """SURVEY.md 8(f).3 writer half: `Detect_YOLO.batch_detect(data_path, result_path)` (detect.py:141-192) over the bundled
    test_data for both checkpoints -- one log line per image in the reference's format with the reference's has-target flag
    (its own logs, tests/golden/golden_results.npz), `result_<name>` images with each box framed in its class colour where the
    reference's own result images have it, labels '%s %.2f' % (name, conf * cls_score), untouched image for 'no targets'."""
    import logging
    import re
    from PIL import Image
    io = yf.io_params_for(res)
    r, g = golden("golden_results"), golden(f"golden_{res}")
    lines = []

    class H(logging.Handler):
        def emit(self, rec):
            lines.append(rec.getMessage())
    logger = logging.getLogger(f"yf-batch-detect-{res}")
    logger.setLevel(logging.INFO); logger.addHandler(H()); logger.propagate = False
    det = yf.Detect_YOLO(dev, WEIGHTS[res], {"io_params": io}, logger)
    data = os.path.join(ROOT, "tests", "golden", "test_data")
    det.batch_detect(data, str(tmp_path), batch_size=8)       # 20 frames: 8 + 8 + 4
    names = [str(n) for n in g["names"]]
    assert len(lines) == 21
    pat = re.compile(r"^image_name:(\S+) -> (detect finished|no targets), infer time:\d+\.\d\dms, post_process time:\d+\.\d\dms, "
                     r"total time:\d+\.\d\dms$")
    for k, (line, name) in enumerate(zip(lines[:20], names)):
        mm = pat.match(line)
        assert mm and mm.group(1) == name, line
        assert (mm.group(2) == "detect finished") == bool(r[f"finished_{res}"][k]), line
    assert re.match(r"^detect avg_time: \d+\.\d\dms$", lines[20])
    colors = {0: (205, 90, 106), 1: (20, 97, 199), 2: (105, 128, 112)}
    for k, name in enumerate(names):
        out = np.asarray(Image.open(os.path.join(str(tmp_path), "result_" + name)).convert("RGB")).astype(np.int32)
        ori = np.asarray(Image.open(os.path.join(data, name)).convert("RGB")).astype(np.int32)
        assert out.shape == (512, 640, 3)
        n = int(g["adj_count"][k])
        want_labels = ['%s %.2f' % (io["class_names"][int(g["adj_cls"][k, j])], g["adj_conf"][k, j] * g["adj_score"][k, j]) for j in range(n)]
        assert det.last_labels[name] == want_labels, (name, det.last_labels[name], want_labels)
        if n == 0:
            d = np.abs(out - ori)                                # re-encoded JPEG (quality 95, cv2.imwrite's default) of the untouched frame
            assert d.mean() <= 1.5 and d.max() <= 40, (name, d.mean(), d.max())
            continue
        for j in range(n):
            x1, y1, x2, y2 = [int(v) for v in g["adj_box"][k, j]]
            c = np.array(colors[int(g["adj_cls"][k, j])])
            pts = [((x1 + x2) // 2, y1), ((x1 + x2) // 2, y2), (x1, (y1 + y2) // 2), (x2, (y1 + y2) // 2)]
            for i, (x, y) in enumerate(pts):
                ours = out[min(max(y, 0), 511), min(max(x, 0), 639)]
                assert np.abs(ours - c).max() <= 40, (name, j, i, ours)
                # and it is what the reference's own result image shows at that pixel (JPEG tolerance on both sides)
                assert np.abs(ours - r[f"edge_rgb_{res}"][k, j, i].astype(np.int32)).max() <= 60, (name, j, i)
    # with more than one batch the driver goes through BatchPipeline (two batches in flight); one batch at a time
    # (in_flight=1, and a single batch of all 20) writes the same flags, labels and images
    piped_labels, piped_flags = dict(det.last_labels), [pat.match(l).group(2) for l in lines[:20]]
    assert det.model.lanes == 2 and det.model.branches == 1        # the pipeline's engine settings were restored
    for kw in (dict(batch_size=8, in_flight=1), dict(batch_size=32)):
        del lines[:]
        out2 = tmp_path / ("serial_%d" % kw["batch_size"])
        out2.mkdir()
        det.batch_detect(data, str(out2), **kw)
        assert det.last_labels == piped_labels and [pat.match(l).group(2) for l in lines[:20]] == piped_flags and len(lines) == 21
        for name in names[:4]:
            a = np.asarray(Image.open(os.path.join(str(tmp_path), "result_" + name)))
            b = np.asarray(Image.open(os.path.join(str(out2), "result_" + name)))
            assert np.array_equal(a, b), name


def test_two_models_of_different_sizes_in_one_process(yf, models, golden, dev):
    """Engine binding hygiene: the validation-path calls and YOLO_post_process.non_maxium_supression pick the engine
    by the model they were given and the tensor's size / device, not 'whichever engine was created first'."""
    from yolo_fastest_amd import validation as val
    (m256, post256, io256), (m512, post512, io512) = models[256], models[512]
    g256, g512 = golden("golden_256"), golden("golden_512")
    with torch.no_grad():                                    # both engines exist, the 512 one was used last
        p256 = m256(_x(g256["input_u8"][:3], dev))
        p512 = m512(_x(g512["input_u8"][:3], dev))
    for io, m, pred, g in ((io256, m256, p256, g256), (io512, m512, p512, g512)):
        losses = [val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev, model=m) for i in range(2)]
        dec = torch.cat([losses[i](pred[i]) for i in range(2)], 1)
        assert dec.shape[1] == 3 * (pred[0].shape[2] * pred[0].shape[3] + pred[1].shape[2] * pred[1].shape[3])
        dets = val.non_max_suppression(dec, 3, conf_thres=0.5, nms_thres=0.2, model=m)
        assert len(dets) == 3
        # decode geometry follows THIS model's input shape: centres of the kept boxes lie inside it
        for d in dets:
            if d is not None:
                assert float(d[:, 2].max()) <= io["input_shape"][1] + 64 and float(d[:, 3].max()) <= io["input_shape"][0] + 64
    # interleaved per-class NMS through the reference-named method of each post-processor
    for post, g in ((post256, g256), (post512, g512), (post256, g256)):
        n = int(g["cand_count"][0])
        L = sorted([list(g["cand_box"][0, k]) + [float(g["cand_conf"][0, k]), float(g["cand_score"][0, k]), int(g["cand_cls"][0, k])]
                    for k in range(n) if g["cand_cls"][0, k] == g["cand_cls"][0, 0]], key=lambda e: e[4], reverse=True)
        keep = post.non_maxium_supression([list(e) for e in L])
        assert len(keep) >= 1 and keep[0][:4] == L[0][:4]
    # changing lanes / chunk after the engine exists is applied (model.engine re-applies them like fusion)
    m256.chunk = 2
    try:
        e = m256.engine(256, 320, 3, dev)
        with torch.no_grad():
            q = m256(_x(g256["input_u8"][:3], dev))
        assert torch.equal(q[0], p256[0]) and torch.equal(q[1], p256[1])
    finally:
        m256.chunk = 0
        m256.engine(256, 320, 3, dev)


def test_small_head_branch_stream_is_identical(yf, golden, dev):
    """The small head's launches run on a side stream of their lane beside the large head's (yf_set_branches, default on): same bits as
    issuing them in line -- one lane, two lanes, several groups of chunks, through yf_detect, repeated (a slot shared between the two
    concurrent branches would show up as a difference sooner or later)."""
    io = yf.io_params_for(256)
    g = golden("golden_256")
    x = _x(np.tile(g["input_u8"], (7, 1, 1))[:132], dev)
    outs = {}
    for lanes, chunk, branches in ((1, 0, 0), (1, 0, 1), (2, 0, 1), (2, 33, 1), (2, 20, 0)):
        m = yf.YoloFastest(io).to(dev).eval()
        m.lanes, m.chunk, m.branches = lanes, chunk, branches
        m.load_state_dict(torch.load(WEIGHTS[256], map_location=dev))
        post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
        with torch.no_grad():
            for rep in range(3):
                hl, hs = m(x)
                raw = post.detect_raw_from_input(x, kmax=16)
                valid = torch.arange(16, device=dev)[None, :] < raw["counts"][:, None]      # entries beyond a frame's count are unwritten
                key = (hl, hs, raw["counts"], raw["boxes"] * valid[:, :, None], raw["src"] * valid)
                if "ref" not in outs:
                    outs["ref"] = key
                for a, b in zip(outs["ref"], key):
                    assert torch.equal(a, b), (lanes, chunk, branches, rep)
                assert torch.equal(raw["head_large"], hl) and torch.equal(raw["head_small"], hs)


def test_stream_overlap_probe_and_lane_assignment(yf, golden, dev):
    """yf_streams_overlap: the probe the engine (lane / branch streams per caller stream) and BatchPipeline.tune_streams use to stay off
    stream pairs that share a hardware queue.  It answers for pool streams, rejects a stream paired with itself, finds an overlapping
    partner among a handful of pool streams (there are at least two hardware queues), and a two-lane pass gives the same bits whatever
    stream it is called on (the assignment is made per caller stream and cached)."""
    import ctypes
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev).eval()
    m.lanes, m.chunk = 2, 10
    m.load_state_dict(torch.load(WEIGHTS[256], map_location=dev))
    x = _x(golden("golden_256")["input_u8"], dev)
    with torch.no_grad():
        ref = m(x)
    e = m.engine(256, 320, 20, dev)
    streams = [torch.cuda.Stream(dev) for _ in range(6)]
    ov = ctypes.c_int(-1)
    found = 0
    for s in streams[1:]:
        yf._lib.check(e.lib.yf_streams_overlap(e.handle, ctypes.c_void_p(streams[0].cuda_stream), ctypes.c_void_p(s.cuda_stream), ctypes.byref(ov)))
        assert ov.value in (0, 1)
        found += ov.value
    assert found >= 1
    with pytest.raises(yf._lib.YFError):
        yf._lib.check(e.lib.yf_streams_overlap(e.handle, ctypes.c_void_p(streams[0].cuda_stream), ctypes.c_void_p(streams[0].cuda_stream), ctypes.byref(ov)))
    for s in streams[:3] + streams[:2]:        # new caller streams are probed, known ones come from the cache
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            out = m(x)
        s.synchronize()
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])


@pytest.mark.parametrize("depth", [2, 3])
def test_batch_pipeline_is_identical(yf, golden, dev, depth):
    """yolo_fastest_amd.BatchPipeline (several batches in flight on separate streams, each with its own engine): every batch's heads
    and detections are bitwise those of the one-at-a-time path, tickets come back in submit order, for DIFFERENT inputs per batch
    (a shared workspace or a missing stream dependency would mix them)."""
    io = yf.io_params_for(256)
    g = golden("golden_256")
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(torch.load(WEIGHTS[256], map_location=dev))
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    rng = np.random.default_rng(3)
    batches = []
    for b in range(7):
        u8 = rng.integers(0, 256, size=(96, 256, 320), dtype=np.uint8)
        u8[b::7][:20] = g["input_u8"][:len(u8[b::7][:20])]
        batches.append(_x(u8, dev))
    want = []
    with torch.no_grad():
        for x in batches:
            pred = m(x)
            want.append((pred, post.detect_raw(pred, kmax=16)))
    torch.cuda.synchronize()
    pipe = yf.BatchPipeline(m, post, depth=depth, kmax=16)
    # streams that share a hardware queue are replaced (probe: yf_streams_overlap), then candidate sets are timed; results must not care
    assert pipe.tune_streams() == [] and len(pipe.streams) == depth
    rates = pipe.tune_streams(batches[0], candidates=2, batches=3)
    assert len(rates) == 2 and all(r > 0 for r in rates) and len(pipe.streams) == depth
    tickets = [pipe.submit(x) for x in batches]
    for (pred, raw), t in zip(want, tickets):
        out = t.result()
        assert torch.equal(out["head_large"], pred[0]) and torch.equal(out["head_small"], pred[1])
        assert torch.equal(out["counts"], raw["counts"])
        valid = torch.arange(16, device=dev)[None, :] < raw["counts"][:, None]
        for k in ("boxes", "cls", "src", "scores"):
            assert torch.equal(out[k][valid], raw[k][valid]), k
    pipe.drain()
    assert len({id(e) for e in m._engines.values()}) >= depth      # one engine per stream
    # `then`: work queued behind ONE batch inside its stream context (what bench.py uses for the RCCL exchange)
    seen = []
    t = pipe.submit(batches[0], then=lambda out: seen.append(int(out["counts"].shape[0])) or "tag")
    assert t.extra == "tag" and seen == [96]
    assert torch.equal(t.synchronize()["head_small"], want[0][0][1])
    with pytest.raises(RuntimeError):
        yf.BatchPipeline(yf.YoloFastest(io).eval(), post)


def test_bench_line_and_multi_gpu_rehearsal(dev):
    """bench.py end to end on the one GPU of the box: the default line (two batches in flight) carries the contract's keys, and
    `--exchange-at-1` runs the N > 1 code path -- RCCL process group, the all-gather of every step's records started from inside each
    batch's stream context, barrier + max-over-ranks timing -- with a 1-rank group."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--batch", "64"]
    for extra in ([], ["--exchange-at-1", "--no-variants", "--no-configs", "--no-train"],
                  ["--exchange-at-1", "--no-variants", "--no-configs", "--no-train", "--in-flight", "1"]):
        r = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        j = json.loads(r.stdout.strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline"):
            assert k in j, k
        assert j["n_gpus"] == 1 and j["steps"] == 4 and j["value"] > 1000 and j["config"]["world_size"] == 1
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j["roofline"])
        assert 0 < j["roofline"]["frac"] < 1
        if not extra:
            # the default line: HBM bytes of the dominant launch measured by the counters in THIS run, the other BASELINE configs
            assert j["roofline"]["traffic"] and "measured in this run" in j["roofline"]["traffic_source"], j["roofline"]
            assert [c["dtype"] for c in j["configs"]] == ["f16x3", "f16", "f16x3", "f32"]
            assert j["configs"][0]["detections_identical_to_f32_on_this_batch"] and j["configs"][0]["max_abs_logit_diff_vs_f32_on_this_batch"] < 2e-2
            assert j["configs"][2]["survivors_per_frame"] >= 200 and j["configs"][2]["detections_identical_on_re_evaluation"]
            assert [t["batch"] for t in j["training"]] == [16, 256] and all(t["value"] > 100 and t["loss_finite"] for t in j["training"])
            assert j["training"][1]["roofline"]["hbm"]["traffic"] > 1e9 and j["training"][1]["roofline"]["dominant_kernel"]["name"]
            assert j["config"]["in_flight"] == 2 and j["one_batch_in_flight"]["detections_identical"]
            assert j["variants"][0]["dtype"] == "f16x3" and j["variants"][0]["detections_identical_to_f32_on_this_batch"]
            assert j["variants"][0]["max_abs_logit_diff_vs_f32_on_this_batch"] < 1e-3


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs: runs the day a multi-GPU node is leased")
def test_two_gpu_bench_gathers_the_single_gpu_records(tmp_path):
    """SURVEY.md 8(e) parity on REAL RCCL with N = 2: `bench.py --gpus 2 --frames fixtures` (one rank per GPU, 20 bundled frames each, the
    ranks' record blocks all-gathered over xGMI) must hand rank 0 exactly the records a single GPU computes for the same 40 frames,
    in frame order; the line says world_size 2."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = ["--steps", "3", "--warmup", "1", "--frames", "fixtures", "--no-cpu-baseline", "--no-variants", "--no-configs", "--no-train"]
    two, one = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "20", "--dump-records", two] + common,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["n_gpus"] == 2 and j["config"]["world_size"] == 2 and j["config"]["global_batch"] == 40
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "40", "--dump-records", one] + common,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(two), np.load(one)
    assert int(a["world_size"]) == 2 and int(b["world_size"]) == 1
    assert np.array_equal(a["counts"], b["counts"]) and a["counts"].shape == (40,) and a["counts"].sum() > 40
    valid = np.arange(a["cls"].shape[1])[None, :] < a["counts"][:, None]
    for k in ("boxes", "scores", "cls", "src"):
        assert np.array_equal(a[k][valid], b[k][valid]), k


def test_two_rank_bench_rehearsal_on_one_gpu(tmp_path):
    """Round 6: the N > 1 path of bench.py has never met a multi-GPU node (SCALE_r01..r05: skipped).  `--share-gpu` runs it with two REAL ranks on the one
    GPU there is -- bench.py's own launcher (torch.distributed.run, 127.0.0.1), WORLD_SIZE / rank checks, per-rank CPU slices, per-rank frames, barriers and
    max-over-ranks timing, the asynchronous exchange of every step's packed records (gloo here, RCCL refuses two ranks on one GPU), rank 0's line: the
    gathered records of the 40 frames are exactly what one process computes for them, in frame order."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = ["--steps", "3", "--warmup", "1", "--frames", "fixtures", "--no-cpu-baseline", "--no-variants", "--no-configs", "--no-train", "--no-live-traffic"]
    two, one = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--batch", "20", "--dump-records", two] + common,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["world_size"] == 2 and j["config"]["global_batch"] == 40 and "REHEARSAL" in j["config"]["parallelism"]
    assert j["scaling"] == "weak" and j["value"] > 0 and j["config"]["cpu_affinity"] is not None
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "40", "--dump-records", one] + common,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(two), np.load(one)
    assert int(a["world_size"]) == 2 and int(b["world_size"]) == 1
    assert np.array_equal(a["counts"], b["counts"]) and a["counts"].shape == (40,) and a["counts"].sum() > 40
    valid = np.arange(a["cls"].shape[1])[None, :] < a["counts"][:, None]
    for k in ("boxes", "scores", "cls", "src"):
        assert np.array_equal(a[k][valid], b[k][valid]), k


def test_training_loss_matches_the_reference(yf, models, golden, dev):
    """SURVEY.md 8(f).4, first slice: `YOLOLossV3(...)(input, targets)` (loss/yolo_loss.py:48-97, get_target :144-196) and the
    gradient `loss.backward()` leaves in the head tensor (train.py:131), on the GPU, against the reference's own run
    (tests/golden/golden_loss_256.npz: its heads of the 20 bundled frames, synthetic targets incl. two targets in one cell, a skipped
    zero-size target, the end marker).  fp32 transcendental functions and the summation order differ from torch-CPU: 2e-5 relative."""
    from yolo_fastest_amd import validation as val
    from oracle import loss_oracle as lo
    m, _, io = models[256]
    g, gl = golden("golden_256"), golden("golden_loss_256")
    targets = torch.from_numpy(gl["targets"]).to(dev)
    total = 0
    for i, name in enumerate(("head_large", "head_small")):
        x = torch.from_numpy(g[name].copy()).to(dev).requires_grad_(True)
        out = val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev, model=m)(x, targets)
        want = gl[name + "_losses"]
        got = np.array([out[0].item()] + list(out[1:]), np.float32)
        assert np.allclose(got, want, rtol=2e-5, atol=1e-8), (name, got, want)
        out[0].backward()
        gw = gl[name + "_grad"]
        assert x.grad.shape == x.shape
        assert np.abs(x.grad.cpu().numpy() - gw).max() <= 2e-5 * np.abs(gw).max() + 1e-10, name
        assert (x.grad.cpu().numpy() != 0).sum() == (gw != 0).sum() or np.abs(x.grad.cpu().numpy()[gw == 0]).max() < 1e-12
        total = total + out[0]
    # the training loop's sum over the two heads (train.py:124-130) is differentiable too, and scales through backward
    x = torch.from_numpy(g["head_small"].copy()).to(dev).requires_grad_(True)
    (3.0 * val.YOLOLossV3(io["anchors"][1], 3, io["input_shape"], dev, model=m)(x, targets)[0]).backward()
    assert np.abs(x.grad.cpu().numpy() - 3.0 * gl["head_small_grad"]).max() <= 1e-4 * np.abs(gl["head_small_grad"]).max()
    # random logits and targets against the oracle (more positives, saturated sigmoids, empty images, no positive at all -> nan cls loss)
    rng = np.random.default_rng(0)
    for trial, (bs, T, npos) in enumerate(((3, 8, 5), (2, 4, 0), (5, 64, 40))):
        h = torch.from_numpy(rng.normal(0, 4.0, size=(bs, 24, 16, 20)).astype(np.float32))
        t = np.zeros((bs, T, 6), np.float32)
        for b in range(bs):
            k = min(T, npos if b else max(npos - 2, 0))
            t[b, :k, 0:2] = rng.uniform(0.02, 0.98, (k, 2)); t[b, :k, 2:4] = rng.uniform(0.02, 0.9, (k, 2))
            t[b, :k, 4] = rng.integers(0, 3, k); t[b, :k, 5] = 255.0
        wl, wg = lo.loss_and_grad(h, torch.from_numpy(t), io["anchors"][0], 3, io["input_shape"])
        x = h.clone().to(dev).requires_grad_(True)
        out = val.YOLOLossV3(io["anchors"][0], 3, io["input_shape"], dev, model=m)(x, torch.from_numpy(t).to(dev))
        got = np.array([out[0].item()] + list(out[1:]), np.float32)
        assert np.allclose(got, wl, rtol=3e-5, atol=1e-7, equal_nan=True), (trial, got, wl)
        out[0].backward()
        if np.isfinite(wl[0]):
            assert np.abs(x.grad.cpu().numpy() - wg.numpy()).max() <= 3e-5 * np.abs(wg.numpy()).max() + 1e-10, trial
    bad = np.zeros((1, 2, 6), np.float32); bad[0, 0] = [1.0, 0.5, 0.1, 0.1, 0, 255.0]       # x == 1.0: column index == width
    with pytest.raises(IndexError):
        val.YOLOLossV3(io["anchors"][0], 3, io["input_shape"], dev, model=m)(torch.zeros(1, 24, 16, 20, device=dev), torch.from_numpy(bad).to(dev))


@pytest.mark.gpu
def test_profile_with_repeated_launches_changes_nothing(yf, golden, dev):
    """yf_set_profile_repeats: every launch of the profiled pass is issued N times back to back between its two events (the event
    packets and dispatch gap of a lone launch drop out of the per-launch figure).  A launch writes its whole output from inputs it does
    not modify, so the pass's result is the same bits, and the per-launch times stay in the range of the one-launch-per-pair ones."""
    io = yf.io_params_for(256)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(torch.load(WEIGHTS[256], map_location=dev))
    g = golden("golden_256")
    x = _x(g["input_u8"], dev)
    with torch.no_grad():
        a = m(x)
    one, h1 = m.profile(x, reps=3, launch_repeats=1, return_heads=True)
    four, h4 = m.profile(x, reps=3, launch_repeats=4, return_heads=True)
    # the heads the profiled passes THEMSELVES wrote (an op with an aliased slot would be relaunched on other data than its
    # first launch saw and this is where it would show)
    for h in (h1, h4):
        assert torch.equal(h[0], a[0]) and torch.equal(h[1], a[1])
    assert [o["name"] for o in one] == [o["name"] for o in four] and all(o["ms"] > 0 for o in four)
    assert sum(o["ms"] for o in four) <= 1.25 * sum(o["ms"] for o in one)
    with torch.no_grad():
        b = m(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with pytest.raises(RuntimeError):     # _lib.YFError
        m.profile(x, reps=1, launch_repeats=0)
    m.profile(x, reps=1, launch_repeats=1)


@pytest.mark.parametrize("split", [False, True])
def test_batch_1_against_the_reference_goldens_in_both_sum_modes(models, golden, dev, split):
    """VERDICT r5 item 7 / ADVICE r5: the reference's real calling pattern is ONE frame per call (detect.py:146-171).  Every golden frame goes
    through alone -- the small-batch plan -- with the split-sum launches off and on (default; the same bits since round 6): heads within the golden bounds of test_heads_match_reference_goldens, boxes / classes / survivor order those of
    test_end_to_end_boxes_on_test_data."""
    for res in (256, 512):
        m, post, io = models[res]
        g = golden(f"golden_{res}")
        x = _x(g["input_u8"], dev)
        m.split_sums = split
        try:
            with torch.no_grad():
                heads = [m(x[f:f + 1].contiguous()) for f in range(20)]
                heads = [(a.clone(), b.clone()) for a, b in heads]
            _check_heads(torch.cat([h[0] for h in heads]), torch.cat([h[1] for h in heads]), g["head_large"], g["head_small"],
                         g["head_large_f64"], g["head_small_f64"], res == 256)
            for f in range(20):
                L = post.detect(heads[f], with_src=True, origin_shape=(512, 640) if res == 256 else None)[0]
                n = int(g["adj_count"][f])
                assert len(L) == n and [e[:4] for e in L] == g["adj_box"][f, :n].tolist(), (res, f, split)
                assert [e[6] for e in L] == g["adj_cls"][f, :n].tolist() and [e[7] for e in L] == g["adj_src"][f, :n].tolist(), (res, f, split)
                assert np.allclose([e[4] for e in L], g["adj_conf"][f, :n], atol=1e-4, rtol=0)
                assert np.allclose([e[5] for e in L], g["adj_score"][f, :n], atol=1e-4, rtol=0)
        finally:
            m.split_sums = True


def test_split_sum_launches_carry_the_bits_of_the_large_batch_plan(models, golden, dev):
    """at <= 9 frames (fp32, 320x256) the stride-32 chain and the small head run as split-sum launches (chunk c of a channel sum on
    its own workgroup, partial sums added in chunk order at the launch boundary).  The large-batch kernels form the same per-chunk partial sums in the
    same order, so a frame's logits are THE SAME BITS at every batch size 1 .. 10, with the split-sum launches on (default) or off, alone or inside a
    large batch -- and so are its detections.  Golden frames mixed with noise frames, every grouping."""
    m, post, io = models[256]
    g = golden("golden_256")
    rng = np.random.default_rng(23)
    u8 = np.concatenate([g["input_u8"], rng.integers(0, 256, (7,) + g["input_u8"].shape[1:], dtype=np.uint8)])
    order = rng.permutation(len(u8))
    x = _x(u8[order], dev)
    with torch.no_grad():
        big = [t.clone() for t in m(torch.cat([x] * 6)[:160].contiguous())]     # 160 frames: the whole-frame (chained) launches
    try:
        for n in range(1, 11):
            for f0 in range(0, len(u8) - n + 1, n):
                xb = x[f0:f0 + n].contiguous()
                with torch.no_grad():
                    m.split_sums = True
                    a = [t.clone() for t in m(xb)]
                    m.split_sums = False
                    b = [t.clone() for t in m(xb)]
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (n, f0)
                assert torch.equal(a[0], big[0][f0:f0 + n]) and torch.equal(a[1], big[1][f0:f0 + n]), (n, f0)
    finally:
        m.split_sums = yf_default_split_sums(m)
    # the launches really differ: fewer dispatches with the split sums off at one frame
    import ctypes
    counts = {}
    for on in (True, False):
        m.split_sums = on
        e = m.engine(256, 320, 1, dev)
        tot = 0
        nl = ctypes.c_int()
        e.lib.yf_num_launches(e.handle, ctypes.byref(nl))
        for op in range(nl.value):
            d = ctypes.c_int()
            assert e.lib.yf_op_dispatches(e.handle, op, 1, ctypes.byref(d)) == 0
            tot += d.value
        counts[on] = tot
    m.split_sums = yf_default_split_sums(m)
    assert counts[True] > counts[False], counts


def yf_default_split_sums(m):
    import yolo_fastest_amd as yf
    return yf.YoloFastest(yf.io_params_for(256)).split_sums


@pytest.mark.parametrize("prec", ["f32", "f16x3", "f16"])
@pytest.mark.parametrize("res", [256, 512])
def test_small_batch_plan_against_the_large_batch_plan_and_the_goldens(yf, golden, dev, res, prec):
    """at N x tiles < #CU every per-frame launch spreads a frame over several workgroups (DESIGN.md section 4 "Small
    batches").  Where that only regroups pixels -- everything but the two split-sum launches -- frames pushed through alone (N = 1), in twos
    and in eights carry the SAME BITS as the same frames inside a batch large enough for the whole-frame launches; the split-sum launches
    (yf_set_split_sums(1), fp32, 320x256) carry the same bits as well since round 6 (the large-batch kernels sum per chunk like they do), with and without them.  Both
    settings are held to the reference's goldens, and the one-frame detections are the reference's."""
    io = yf.io_params_for(res)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(torch.load(WEIGHTS[res], map_location=dev))
    if prec != "f32":
        m.precision = prec
    g = golden(f"golden_{res}")
    rng = np.random.default_rng(11)
    nbig = 160 if res == 256 else 48         # 2 x (frames x whole-frame tiles) > 256 CUs: the large-batch launches
    noise = rng.integers(0, 256, (nbig - 20,) + g["input_u8"].shape[1:], dtype=np.uint8)
    u8 = np.concatenate([g["input_u8"], noise])
    x = _x(u8, dev)
    with torch.no_grad():
        big = [t.clone() for t in m(x)]
        for n, picks in ((1, (0, 7, 19, 25)), (2, (0, 18, 30)), (8, (0, 12, 24))):
            for f0 in picks:
                for split in (True, False):
                    m.split_sums = split
                    small = m(x[f0:f0 + n].contiguous())
                    # every regrouping keeps the bits -- since round 6 also the split-sum launches (fp32, 320x256, <= 9 frames: the stride-32 chain
                    # and the small head spread their channel sums over several workgroups, mres_esplit_kernel / mdw2_esplit_kernel): the
                    # large-batch kernels form the same per-chunk partial sums in the same order (PSUM in yf_mres_kernels.hip / yf_mdw_kernels.hip)
                    assert torch.equal(small[0], big[0][f0:f0 + n]) and torch.equal(small[1], big[1][f0:f0 + n]), (res, prec, n, f0, split)
        m.split_sums = True
    if prec != "f16":
        _check_heads(big[0][:20], big[1][:20], g["head_large"], g["head_small"], g["head_large_f64"], g["head_small_f64"], res == 256)
        # ... and the small-batch launches themselves against the reference: the 20 golden frames four at a time, and one by one for the detections
        with torch.no_grad():
            four = [m(x[f0:f0 + 4].contiguous()) for f0 in range(0, 20, 4)]
        _check_heads(torch.cat([h[0] for h in four]), torch.cat([h[1] for h in four]), g["head_large"], g["head_small"], g["head_large_f64"],
                     g["head_small_f64"], res == 256)
        post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
        for f in (0, 5, 13, 19):
            with torch.no_grad():
                L = post.detect(m(x[f:f + 1].contiguous()), with_src=True)[0]
            k = int(g["final_count"][f])
            assert [e[:4] for e in L] == g["final_box"][f, :k].tolist() and [e[7] for e in L] == g["final_src"][f, :k].tolist(), (res, prec, f)
