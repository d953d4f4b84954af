"""CPU check of the MFMA K-permutation bookkeeping used by egonerf_amd/csrc/ego_shade.hip: the numpy wave
emulator fed with the packed weights must reproduce the plain basis + MLP_Fea arithmetic."""
import numpy as np

from egonerf_amd import synth
from tests import mfma_emulator as em


def plain(w, v, flag, d):
    feat = np.where(flag[:, None] == 0, v @ w["basis_mat_yin.weight"].T.astype(np.float64),
                    v @ w["basis_mat_yang.weight"].T.astype(np.float64))
    pe = lambda x: np.concatenate([np.sin((x[..., None] * [1, 2]).reshape(x.shape[0], -1)),
                                   np.cos((x[..., None] * [1, 2]).reshape(x.shape[0], -1))], 1)
    x = np.concatenate([feat, d, pe(feat), pe(d)], 1)
    h = np.maximum(x @ w["renderModule.mlp.0.weight"].T + w["renderModule.mlp.0.bias"], 0)
    h = np.maximum(h @ w["renderModule.mlp.2.weight"].T + w["renderModule.mlp.2.bias"], 0)
    o = h @ w["renderModule.mlp.4.weight"].T + w["renderModule.mlp.4.bias"]
    return feat, 1 / (1 + np.exp(-o))


def test_emulated_wave_matches_plain_mlp():
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=77)
    w["renderModule.mlp.4.bias"] = np.array([0.3, -0.2, 0.1], np.float32)  # exercise b3 (zero-init in the reference)
    packed = em.pack_mlp(w)
    assert packed.shape == (em.PACKED_FLOATS,) and em.PACKED_FLOATS == 46852
    v = synth.hash_normal(5, 0, 32 * 144).reshape(32, 144) * 0.3
    flag = (synth.hash_uniform(5, 1, 32) > 0.5).astype(np.int64)  # mixed yin / yang wave
    d = synth.make_rays(32, seed=3)[:, 3:6].astype(np.float64)
    feat, rgb = em.emulate_tile(packed, v, flag, d)
    feat_ref, rgb_ref = plain(w, v, flag, d)
    assert np.abs(feat - feat_ref).max() < 1e-12
    assert np.abs(rgb - rgb_ref).max() < 1e-12
    # asymmetric sanity: the answer really depends on the sample -> lane mapping
    assert np.abs(rgb - rgb_ref[::-1]).max() > 1e-3


def test_every_mlp_input_column_is_used_exactly_once():
    for h_cols in ([em.x_channel(kk, h) for kk in range(em.KS1) for h in (0, 1)],):
        cols = sorted(c for c in h_cols if c >= 0)
        assert cols == list(range(em.MLP_IN))
    rows = sorted(em.slot_row(r, h) for r in range(16) for h in (0, 1))
    assert rows == list(range(32))


def test_f16f6_packed_operands_mean_what_the_kernel_assumes():
    """The f16f6 image (k_pack_mlp_f6 == em.pack_mlp_f6 bit for bit, checked on the GPU): decoding the packed fp6 blocks, their element
    orders (f6_value) and scale bytes the way the MFMA reads them, with the activations quantised by the kernel's rule (block scale
    from the largest |x| of a lane's 32 values; layer 1: from 1 and the unbounded inputs only), must give both layers' products to
    ~2^-16 relative - >= 10x closer to float64 than the fp16 main term alone."""
    cfg = synth.SceneConfig(n_voxel=20 ** 3)
    w = synth.make_weights(cfg, seed=1234)
    n = 24
    h1 = np.maximum(synth.hash_normal(3, 0, n * 128).reshape(n, 128), 0).astype(np.float32)
    W2 = np.asarray(w["renderModule.mlp.2.weight"], np.float64)
    X = [np.stack([h1[:, (kk >> 4) * 32 + em.slot_row(kk & 15, h)] for kk in range(em.KS2)], 1) for h in range(2)]
    ref = h1.astype(np.float64) @ W2.T
    e6 = np.abs(em.f6_layer_reference(w, True, X) - ref).max()
    e16 = np.abs(h1.astype(np.float16).astype(np.float64) @ W2.astype(np.float32).astype(np.float16).astype(np.float64).T - ref).max()
    assert e6 < 0.1 * e16 and e6 < 1.5e-4, (e6, e16)
    feat = (synth.hash_normal(3, 1, n * 27).reshape(n, 27) * 0.7).astype(np.float32)
    d = synth.make_rays(n, seed=3)[:, 3:6].astype(np.float32)
    pe = lambda x: np.concatenate([np.sin((x[..., None] * [1, 2]).reshape(x.shape[0], -1)), np.cos((x[..., None] * [1, 2]).reshape(x.shape[0], -1))], 1)
    x = np.concatenate([feat, d, pe(feat), pe(d)], 1).astype(np.float32)
    W1 = np.asarray(w["renderModule.mlp.0.weight"], np.float64)
    X = []
    for h in range(2):
        cols = [em.x_channel(kk, h) for kk in range(em.KS1)]
        X.append(np.stack([x[:, c] if c >= 0 else np.zeros(n, np.float32) for c in cols], 1))
    ref = x.astype(np.float64) @ W1.T
    e6 = np.abs(em.f6_layer_reference(w, False, X) - ref).max()
    e16 = np.abs(x.astype(np.float16).astype(np.float64) @ W1.astype(np.float32).astype(np.float16).astype(np.float64).T - ref).max()
    assert e6 < 0.1 * e16 and e6 < 1.5e-4, (e6, e16)
    # every K value of a lane half appears exactly once per term across a layer's groups
    for layer2, groups, nk in ((False, em.G6_1, em.KS1), (True, em.G6_2, em.KS2)):
        for term in range(2):
            ks = sorted(k for g in range(groups) for e in range(32) if (k := em.f6_value(layer2, g, term, e)) >= 0 and k < nk)
            assert ks == list(range(nk)), (layer2, term)
