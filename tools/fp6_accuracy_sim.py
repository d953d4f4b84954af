"""CPU estimate (round 4): accuracy of the colour MLP when the two low-order terms of the fp16 split run on the fp6 (e2m3) path of
v_mfma_scale_f32_32x32x64_f8f6f4 with one power-of-two block scale per lane and 32 K values (dynamic for the activations: from the
block's largest magnitude; static for the weights), next to the shipped fp8 (e4m3, fixed scales) form and the three-term fp16 form.
Blocks of 32 follow the kernel's grouping only approximately (32 consecutive K values of one lane half); per-sample |d rgb| is an
upper bound for the composited error."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonerf_amd import synth
from tools.fp8_accuracy_sim import f16, e4m3, pe  # noqa: E402  (prints that module's table first)


def e2m3(x):
    """round to nearest fp6 e2m3 (bias 1: subnormal step 0.125 below 1, max 7.5), saturating"""
    x = np.asarray(x, np.float64)
    s, a = np.sign(x), np.minimum(np.abs(x), 7.5)
    e = np.clip(np.floor(np.log2(np.maximum(a, 1e-30))), 0.0, 2.0)
    q = 2.0 ** (e - 3)
    return s * np.minimum(np.round(a / q) * q, 7.5)


def block_scale(a, top):
    """power of two s with max|a| / s in [top/2, top) per block (last axis), 1 for an all-zero block"""
    m = np.abs(a).max(-1, keepdims=True)
    e = np.floor(np.log2(np.maximum(m, 2.0 ** -60)))
    return np.where(m > 0, 2.0 ** (e + 1 - np.log2(top)), 1.0)


def q6(a, blk=32, shift=0.0):
    """quantise the last axis in blocks of blk with a per-block scale; shift: extra exponent offset of the scale"""
    n = a.shape[-1]
    pad = (-n) % blk
    ap = np.concatenate([a, np.zeros(a.shape[:-1] + (pad,))], -1).reshape(a.shape[:-1] + (-1, blk))
    s = block_scale(ap, 8.0) * 2.0 ** shift
    return (e2m3(ap / s) * s).reshape(a.shape[:-1] + (-1,))[..., :n]


def q6_resid(x, xh, blk=32):
    """residual x - xh quantised with the block scale of x times 2^-11"""
    n = x.shape[-1]
    pad = (-n) % blk
    z = np.zeros(x.shape[:-1] + (pad,))
    xp = np.concatenate([x, z], -1).reshape(x.shape[:-1] + (-1, blk))
    rp = np.concatenate([x - xh, z], -1).reshape(xp.shape)
    s = block_scale(xp, 8.0) * 2.0 ** -11
    return (e2m3(rp / s) * s).reshape(x.shape[:-1] + (-1,))[..., :n]


def layer(W, b, X, mode):
    Wh = f16(W); Xh = f16(X)
    if mode == "f64":
        return X @ W.T + b
    main = Xh @ Wh.T
    if mode == "f16x3":
        corr = Xh @ f16(W - Wh).T + f16(X - Xh) @ Wh.T
    elif mode == "fp8corr":
        sw = sx = 2.0 ** 11
        corr = (e4m3(Xh) @ e4m3((W - Wh) * sw).T) / sw + (e4m3((X - Xh) * sx) @ e4m3(Wh).T) / sx
    elif mode == "fp6corr":
        # static scales for the weights: per row and block of 32 (the residual block is scaled by its own maximum)
        corr = q6(Xh) @ q6(W - Wh).T + q6_resid(X, Xh) @ q6(Wh).T
    elif mode == "f16only":
        corr = 0.0
    return main + corr + b


if __name__ == "__main__":
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=1234)
    W1, b1, W2, b2, W3, b3 = (np.asarray(w[k], np.float64) for k in ("renderModule.mlp.0.weight", "renderModule.mlp.0.bias",
                                                                       "renderModule.mlp.2.weight", "renderModule.mlp.2.bias",
                                                                       "renderModule.mlp.4.weight", "renderModule.mlp.4.bias"))
    rng = np.random.default_rng(0)
    n = 20000
    for fscale in (0.7, 3.0):
        feat = rng.normal(0, fscale, (n, 27))
        d = rng.normal(0, 1, (n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pf, pd = pe(feat), pe(d)
        x = np.concatenate([feat, d, pf[:, 27:], pd[:, 3:]], 1)
        res = {}
        for mode in ("f64", "f16x3", "fp8corr", "fp6corr", "f16only"):
            h1 = np.maximum(layer(W1, b1, x, mode), 0)
            h2 = np.maximum(layer(W2, b2, h1, mode), 0)
            res[mode] = 1 / (1 + np.exp(-(h2 @ W3.T + b3)))
        print(f"feature sigma {fscale}")
        for mode in ("f16x3", "fp8corr", "fp6corr", "f16only"):
            e = np.abs(res[mode] - res["f64"])
            print(f"  {mode:8s}: per-sample |d rgb| max {e.max():.2e}  99.9 % {np.quantile(e, 0.999):.2e}  rms {np.sqrt((e ** 2).mean()):.2e}")
