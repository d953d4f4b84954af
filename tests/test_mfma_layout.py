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
