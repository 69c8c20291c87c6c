"""Golden vectors for SigmoidFocalLoss from the reference's CUDA FORMULA, evaluated in float64 numpy.

The reference ships no CPU C++ focal loss and no test of its CUDA kernel (oracle/README.md); its Python composite
(layers/sigmoid_focal_loss.py:40-50, captured in focal_python_composite.npz) only pins |x| <~ 15.  This generator restates
the arithmetic of maskrcnn_benchmark/csrc/cuda/SigmoidFocalLoss_cuda.cu in float64 over logits up to |x| = 100, so the
oracle's C restatement (and through it the HIP kernel) is pinned where the fp32 kernel clamps and saturates:

  forward  (SigmoidFocalLoss_cuda.cu:29-57)
      c1 = (t == d + 1);  c2 = (t >= 0 & t != d + 1);  zn = 1 - alpha;  zp = alpha             (:35-40)
      p = 1 / (1 + exp(-x))                                                                      (:43)
      term1 = (1 - p)**gamma * log(max(p, FLT_MIN))                                              (:46)
      term2 = p**gamma * (-x * (x >= 0) - log(1 + exp(x - 2 * x * (x >= 0))))                    (:49-51)
      loss = -c1 * term1 * zp - c2 * term2 * zn                                                  (:53-55)
  backward (SigmoidFocalLoss_cuda.cu:61-99)
      term1 = (1 - p)**gamma * (1 - p - p * gamma * log(max(p, FLT_MIN)))                        (:85-86)
      term2 = p**gamma * ((-x * (x >= 0) - log(1 + exp(x - 2 * x * (x >= 0)))) * (1 - p) * gamma - p)   (:89-92)
      d_logits = (-c1 * term1 * zp - c2 * term2 * zn) * d_losses                                 (:93-96)

FLT_MIN is the float32 constant the kernel uses (1.17549435e-38): a float64 evaluation must keep that clamp to describe
the same function.  Run in the build container:  python tests/golden/make_golden_focal_formula.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FLT_MIN = np.float64(np.finfo(np.float32).tiny)


def forward(x, t, gamma, alpha):
    x = x.astype(np.float64)
    d = np.arange(x.shape[1])[None, :]
    tt = t.astype(np.int64)[:, None]
    c1 = (tt == d + 1).astype(np.float64)
    c2 = ((tt >= 0) & (tt != d + 1)).astype(np.float64)
    zn, zp = 1.0 - alpha, alpha
    with np.errstate(over="ignore"):
        p = 1.0 / (1.0 + np.exp(-x))
        ge = (x >= 0).astype(np.float64)
        term1 = (1.0 - p) ** gamma * np.log(np.maximum(p, FLT_MIN))
        term2 = p ** gamma * (-1.0 * x * ge - np.log(1.0 + np.exp(x - 2.0 * x * ge)))
    return -c1 * term1 * zp + -c2 * term2 * zn


def backward(x, t, d_losses, gamma, alpha):
    x = x.astype(np.float64)
    d = np.arange(x.shape[1])[None, :]
    tt = t.astype(np.int64)[:, None]
    c1 = (tt == d + 1).astype(np.float64)
    c2 = ((tt >= 0) & (tt != d + 1)).astype(np.float64)
    zn, zp = 1.0 - alpha, alpha
    with np.errstate(over="ignore"):
        p = 1.0 / (1.0 + np.exp(-x))
        ge = (x >= 0).astype(np.float64)
        term1 = (1.0 - p) ** gamma * (1.0 - p - (p * gamma * np.log(np.maximum(p, FLT_MIN))))
        term2 = p ** gamma * ((-1.0 * x * ge - np.log(1.0 + np.exp(x - 2.0 * x * ge))) * (1.0 - p) * gamma - p)
    return (-c1 * term1 * zp + -c2 * term2 * zn) * d_losses.astype(np.float64)


def main():
    rng = np.random.RandomState(42)
    R, C = 240, 80
    x = rng.uniform(-100.0, 100.0, (R, C)).astype(np.float32)
    # rows of exact probe values: the clamp (p < FLT_MIN below x = -87.34), float32 exp overflow (|x| > 88.7), saturation
    probes = np.array([-100.0, -95.0, -90.0, -88.8, -88.0, -87.4, -87.3, -60.0, -30.0, -20.0, -17.0, -10.0, -1.0, -1e-3, 0.0,
                       1e-3, 1.0, 10.0, 16.7, 17.0, 20.0, 30.0, 60.0, 87.3, 88.0, 88.8, 90.0, 95.0, 100.0], np.float32)
    for i, v in enumerate(probes):
        x[i % R, :] = v
    x[:64] += rng.uniform(-0.5, 0.5, (64, C)).astype(np.float32) * (np.arange(64)[:, None] >= len(probes))
    t = rng.randint(-1, C + 1, size=R).astype(np.int32)       # -1 ignore, 0 background, 1..C classes
    t[:len(probes)] = np.resize(np.array([1, 0, 5, -1, 80, 2], np.int32), len(probes))
    d = rng.randn(R, C).astype(np.float32)
    out = {"logits": x, "targets": t, "d_losses": d}
    for tag, (gamma, alpha) in {"a": (2.0, 0.25), "b": (1.5, 0.4)}.items():
        out["cfg_" + tag] = np.array([gamma, alpha], np.float64)
        out["losses_" + tag] = forward(x, t, gamma, alpha)
        out["d_logits_" + tag] = backward(x, t, d, gamma, alpha)
    np.savez_compressed(os.path.join(HERE, "focal_cuda_formula_fp64.npz"), **out)
    print("focal_cuda_formula_fp64.npz:", x.shape, "max |x| =", float(np.abs(x).max()))


if __name__ == "__main__":
    main()
