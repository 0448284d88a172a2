"""CPU simulation of the fp16 variant's rounding points (no GPU): which of them costs how much logit error.

  python tools/fp16_sim.py [res] [nframes]

Folded-BN forward in fp32 (torch CPU) with optional fp16 rounding of
  W  the pointwise / dense weights (MFMA operands),
  T  the tensors that live in HBM between launches of the fused plan (the narrow residual trunk + head-branch tensors),
  A  the activation operand of every pointwise MFMA (block input as the expansion's operand, depthwise result as the
     projection's operand),
  E  the expanded tensor kept in LDS between expansion and depthwise.
Compared with the reference's fp32 head logits in tests/golden/golden_<res>.npz.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import backbone_oracle as bo  # noqa: E402  (tools/ is not product)
from oracle.fp16_rounding_sim import Sim, fold  # noqa: E402

WDIR = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights")
WEIGHTS = {256: os.path.join(WDIR, "yolo_fastest_256x320_epoch28.pth"), 512: os.path.join(WDIR, "yolo_fastest_512x640_epoch27.pth")}


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    g = np.load(os.path.join(ROOT, "tests", "golden", f"golden_{res}.npz"))
    fw = fold(bo.load_state_dict(WEIGHTS[res]))
    x = bo.preprocess(g["input_u8"][:nf])
    ref = (g["head_large"][:nf], g["head_small"][:nf])
    for label, kw in (("fp32 folded", {}), ("W", dict(W=True)), ("T", dict(T=True)), ("A", dict(A=True)), ("E", dict(E=True)),
                      ("W+A+E (trunk fp32)", dict(W=True, A=True, E=True)), ("W+T+A+E (round 1)", dict(W=True, T=True, A=True, E=True)),
                      ("T+A+E", dict(T=True, A=True, E=True)), ("A+E", dict(A=True, E=True))):
        out = Sim(fw, **kw).forward(x)
        s = []
        for o, r in zip(out, ref):
            d = np.abs(o.numpy() - r)
            s.append(f"max {d.max():.4f} p99 {np.quantile(d, 0.99):.4f} mean {d.mean():.5f}")
        print(f"{label:24s} large: {s[0]}   small: {s[1]}")


if __name__ == "__main__":
    main()
