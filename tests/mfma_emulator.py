"""Test infrastructure: numpy model of (a) the packed-weight layout ego_pack_mlp produces and (b) the
wave64 dataflow of k_shade (v_mfma_f32_32x32x2_f32 operand/accumulator lane maps from the CDNA4 ISA:
A[i][k] <- lane (i + 32k), B[k][j] <- lane (j + 32k), D[i][j] -> lane j + 32*((i>>2)&1), reg (i&3) + 4*(i>>3)).
Lets the CPU test-suite check the K-permutation bookkeeping without a GPU, and gives the GPU test a
bit-exact expectation for the device-side packer."""
import numpy as np

APP_C, APP_HALF, APP_DIM, HID, NSLOT = 48, 24, 27, 128, 14
KS_BASIS, KS1, KS2, MLP_IN = 72, 80, 64, 150
OFF_W1 = 0
OFF_W2 = OFF_W1 + KS1 * 4 * 64
OFF_B1 = OFF_W2 + KS2 * 4 * 64
OFF_B2 = OFF_B1 + 128
OFF_W3 = OFF_B2 + 128
OFF_B3 = OFF_W3 + 512
OFF_BASIS = OFF_B3 + 4
PACKED_FLOATS = OFF_BASIS + 2 * (KS_BASIS // 4) * 64 * 4


def slot_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def app_channel(kk, h):
    """Appearance channel gathered by lane half h as its kk-th product (halves interleave per float4 quad)."""
    return (kk // APP_HALF) * APP_C + ((kk % APP_HALF) // 4) * 8 + 4 * h + kk % 4


def x_channel(kk, h):
    if kk < 5 * NSLOT:
        kind, r = kk % 5, kk // 5
        f = 2 * r + h
        if f >= APP_DIM:
            return -1
        pe0, pe1 = APP_DIM + 3, APP_DIM + 3 + 2 * APP_DIM
        return [f, pe0 + 2 * f, pe0 + 2 * f + 1, pe1 + 2 * f, pe1 + 2 * f + 1][kind]
    if kk < 5 * NSLOT + 8:
        t = kk - 5 * NSLOT + 8 * h
        return APP_DIM + t if t < 3 else (138 + (t - 3) if t < 15 else -1)
    return -1


def pack_mlp(w):
    """w: reference-layout dict (synth.make_weights keys) -> packed float32 blob."""
    w1, b1 = w["renderModule.mlp.0.weight"], w["renderModule.mlp.0.bias"]
    w2, b2 = w["renderModule.mlp.2.weight"], w["renderModule.mlp.2.bias"]
    w3, b3 = w["renderModule.mlp.4.weight"], w["renderModule.mlp.4.bias"]
    out = np.zeros(PACKED_FLOATS, np.float32)
    W1 = out[OFF_W1:OFF_W2].reshape(KS1 // 4, 4, 64, 4)
    W2 = out[OFF_W2:OFF_B1].reshape(KS2 // 4, 4, 64, 4)
    for lane in range(64):
        i, h = lane & 31, lane >> 5
        for m in range(4):
            for kk in range(KS1):
                ch = x_channel(kk, h)
                if ch >= 0:
                    W1[kk // 4, m, lane, kk % 4] = w1[m * 32 + i, ch]
            for kk in range(KS2):
                W2[kk // 4, m, lane, kk % 4] = w2[m * 32 + i, (kk >> 4) * 32 + slot_row(kk & 15, h)]
    B1 = out[OFF_B1:OFF_B2].reshape(4, 2, 16)
    B2 = out[OFF_B2:OFF_W3].reshape(4, 2, 16)
    W3 = out[OFF_W3:OFF_B3].reshape(4, 2, 16, 4)
    for m in range(4):
        for h in range(2):
            for r in range(16):
                u = m * 32 + slot_row(r, h)
                B1[m, h, r], B2[m, h, r] = b1[u], b2[u]
                W3[m, h, r, :3] = w3[:, u]
    out[OFF_B3:OFF_B3 + 3] = b3
    BAS = out[OFF_BASIS:].reshape(2, KS_BASIS // 4, 64, 4)
    for g, key in enumerate(("basis_mat_yin.weight", "basis_mat_yang.weight")):
        bm = w[key]
        for lane in range(64):
            i, h = lane & 31, lane >> 5
            rh, r = (i >> 2) & 1, (i & 3) + 4 * (i >> 3)
            f = 2 * r + rh
            if r < NSLOT and f < APP_DIM:
                for kk in range(KS_BASIS):
                    BAS[g, kk // 4, lane, kk % 4] = bm[f, app_channel(kk, h)]
    return out


def mfma_32x32x2(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64,16] per-lane accumulators.  Returns the updated acc."""
    A = np.stack([a[:32], a[32:]], 1).astype(np.float64)   # A[i][k]
    B = np.stack([b[:32], b[32:]], 0).astype(np.float64)   # B[k][j]
    D = A @ B                                                # [32 i][32 j]
    out = acc.astype(np.float64).copy()
    for i in range(32):
        h, r = (i >> 2) & 1, (i & 3) + 4 * (i >> 3)
        out[h * 32 + np.arange(32), r] += D[i]
    return out


def emulate_tile(packed, v, grid_flag, viewdirs):
    """One wave: v [32 samples][144] products (reference channel order), grid_flag [32] (0 yin / 1 yang),
    viewdirs [32][3] -> (feat [32][27], rgb [32][3]) following k_shade's register dataflow."""
    lane = np.arange(64)
    j, h = lane & 31, lane >> 5
    # lane's B operands for the basis: its half of each plane's channels
    vl = np.zeros((64, KS_BASIS))
    for kk in range(KS_BASIS):
        vl[:, kk] = v[j, app_channel(kk, h)]
    g = grid_flag[j]
    BAS = packed[OFF_BASIS:].reshape(2, KS_BASIS // 4, 64, 4)
    fe = np.zeros((64, 16))
    for gi in range(2):
        for kk in range(KS_BASIS):
            fe = mfma_32x32x2(BAS[gi, kk // 4, :, kk % 4], np.where(g == gi, vl[:, kk], 0.0), fe)
    feat = np.zeros((32, APP_DIM))
    for r in range(NSLOT):
        for hh in range(2):
            if 2 * r + hh < APP_DIM:
                feat[:, 2 * r + hh] = fe[hh * 32 + np.arange(32), r]
    # layer-1 operands
    d = viewdirs[j]
    vlist = np.concatenate([d, np.stack([np.sin(d[:, 0]), np.sin(2 * d[:, 0]), np.sin(d[:, 1]), np.sin(2 * d[:, 1]),
                                         np.sin(d[:, 2]), np.sin(2 * d[:, 2]), np.cos(d[:, 0]), np.cos(2 * d[:, 0]),
                                         np.cos(d[:, 1]), np.cos(2 * d[:, 1]), np.cos(d[:, 2]), np.cos(2 * d[:, 2])], 1),
                            np.zeros((64, 1))], 1)
    X = np.zeros((64, KS1))
    for r in range(NSLOT):
        f = fe[:, r]
        X[:, 5 * r:5 * r + 5] = np.stack([f, np.sin(f), np.sin(2 * f), np.cos(f), np.cos(2 * f)], 1)
    for t in range(8):
        X[:, 5 * NSLOT + t] = np.where(h == 1, vlist[:, 8 + t], vlist[:, t])
    W1 = packed[OFF_W1:OFF_W2].reshape(KS1 // 4, 4, 64, 4)
    W2 = packed[OFF_W2:OFF_B1].reshape(KS2 // 4, 4, 64, 4)
    B1 = packed[OFF_B1:OFF_B2].reshape(4, 2, 16)
    B2 = packed[OFF_B2:OFF_W3].reshape(4, 2, 16)
    W3 = packed[OFF_W3:OFF_B3].reshape(4, 2, 16, 4)
    H = [B1[m][h].astype(np.float64) for m in range(4)]
    for kk in range(KS1):
        for m in range(4):
            H[m] = mfma_32x32x2(W1[kk // 4, m, :, kk % 4], X[:, kk], H[m])
    H = [np.maximum(x, 0) for x in H]
    G = [B2[m][h].astype(np.float64) for m in range(4)]
    for kk in range(KS2):
        for m in range(4):
            G[m] = mfma_32x32x2(W2[kk // 4, m, :, kk % 4], H[kk >> 4][:, kk & 15], G[m])
    o = np.zeros((64, 3))
    for m in range(4):
        o += np.einsum("lr,lrc->lc", np.maximum(G[m], 0), W3[m][h][:, :, :3])
    o = o[:32] + o[32:]
    rgb = 1.0 / (1.0 + np.exp(-(o + packed[OFF_B3:OFF_B3 + 3])))
    return feat, rgb


def pack_mlp_f16(w):
    """fp16-split blob (second half of what ego_pack_mlp writes): same offsets, matrix regions hold
    [k-step][m-tile][term hi|lo][lane][8 k] fp16 with hi = fp16(w), lo = fp16(w - hi)."""
    f32 = pack_mlp(w)
    out = f32.copy()

    def split(x):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        return hi, lo

    def region(src_f32_frag, steps, tiles):
        # src: [k4][m][lane][4] fp32 fragments -> per-lane K list [m][lane][kk]
        kk = src_f32_frag.reshape(steps * 2, tiles, 64, 4).transpose(1, 2, 0, 3).reshape(tiles, 64, steps * 8)
        hi, lo = split(kk)
        dst = np.zeros((steps, tiles, 2, 64, 8), np.float16)
        for st in range(steps):
            dst[st, :, 0] = hi[:, :, st * 8:st * 8 + 8]
            dst[st, :, 1] = lo[:, :, st * 8:st * 8 + 8]
        return dst.reshape(-1).view(np.float32)

    out[OFF_W1:OFF_W2] = region(f32[OFF_W1:OFF_W2], KS1 // 8, 4)
    out[OFF_W2:OFF_B1] = region(f32[OFF_W2:OFF_B1], KS2 // 8, 4)
    # basis fragments of the f16x3 kernel follow the team-gather K order (app_channel_g), not the fp32 kernel's
    bas = np.zeros((2, KS_BASIS // 4, 64, 4), np.float32)
    for g, key in enumerate(("basis_mat_yin.weight", "basis_mat_yang.weight")):
        for lane in range(64):
            i, h = lane & 31, lane >> 5
            rh, r = (i >> 2) & 1, (i & 3) + 4 * (i >> 3)
            f = 2 * r + rh
            if r < NSLOT and f < APP_DIM:
                for kk in range(KS_BASIS):
                    bas[g, kk // 4, lane, kk % 4] = w[key][f, app_channel_g(kk, h)]
    for g in range(2):
        n = (KS_BASIS // 8) * 2 * 64 * 8 // 2
        out[OFF_BASIS + g * n: OFF_BASIS + (g + 1) * n] = region(bas[g].reshape(KS_BASIS // 4, 1, 64, 4), KS_BASIS // 8, 1)
    return out


def app_channel_g(kk, h):
    """K order of the f16x3 kernel's 4-lane-team gather: product kk = plane*24 + half*12 + i*4 + c of lane half h is
    channel plane*48 + 16 i + 4 (h + 2 half) + c."""
    within = kk % APP_HALF
    return (kk // APP_HALF) * APP_C + ((within % 12) // 4) * 16 + 4 * (h + 2 * (within // 12)) + kk % 4


def e4m3_bytes(x):
    """float -> OCP e4m3fn code (round to nearest even, subnormals down to 2^-9, saturating at 448: what v_cvt_pk_fp8_f32 writes
    for in-range inputs, tools/fp8_layout_probe.hip)."""
    x = np.asarray(x, np.float64)
    s = (np.signbit(x)).astype(np.uint8) << 7
    a = np.minimum(np.abs(x), 448.0)
    e = np.maximum(np.floor(np.log2(np.maximum(a, 2.0 ** -40))), -6.0)
    m = np.rint(a / 2.0 ** (e - 3))                     # 8..16 for normals (16 = carry into the next binade), 0..8 below 2^-6
    carry = m >= 16
    e = np.where(carry, e + 1, e)
    m = np.where(carry, 8, m)
    normal = m >= 8
    code = np.where(normal, ((e + 7).astype(np.int64) << 3) | (m.astype(np.int64) - 8), m.astype(np.int64))
    return (code.astype(np.uint8) | s).astype(np.uint8)


F8_FLOATS = OFF_B1


def pack_mlp_f8(w):
    """Fourth region of the packed blob (k_pack_mlp_f8): per layer the fp16 hi fragments [step][m-tile][lane][8], then the fp8
    fragments [pair][m-tile][part][lane][16 bytes]; operand byte pos: [0..7] e4m3(w_lo 2^11) step 2p, [8..15] step 2p+1,
    [16..23] e4m3(w_hi) step 2p, [24..31] step 2p+1."""
    f32 = pack_mlp(w)
    out = np.zeros(F8_FLOATS, np.float32)

    def region(src_f32_frag, steps):
        kk = src_f32_frag.reshape(steps * 2, 4, 64, 4).transpose(1, 2, 0, 3).reshape(4, 64, steps * 8)   # [mt][lane][k]
        hi = kk.astype(np.float16)
        lo = kk - hi.astype(np.float32)
        hi_part = np.zeros((steps, 4, 64, 8), np.float16)
        for st in range(steps):
            hi_part[st] = hi[:, :, st * 8:st * 8 + 8]
        f8 = np.zeros((steps // 2, 4, 2, 64, 16), np.uint8)
        for p in range(steps // 2):
            op = np.zeros((4, 64, 32), np.uint8)
            for half in range(2):
                sl = slice((2 * p + half) * 8, (2 * p + half) * 8 + 8)
                op[:, :, 8 * half:8 * half + 8] = e4m3_bytes(lo[:, :, sl].astype(np.float64) * 2048.0)
                op[:, :, 16 + 8 * half:16 + 8 * half + 8] = e4m3_bytes(hi[:, :, sl].astype(np.float64))
            f8[p, :, 0] = op[:, :, :16]
            f8[p, :, 1] = op[:, :, 16:]
        return np.concatenate([hi_part.reshape(-1).view(np.float32), f8.reshape(-1).view(np.float32)])

    out[OFF_W1:OFF_W2] = region(f32[OFF_W1:OFF_W2], KS1 // 8)
    out[OFF_W2:OFF_B1] = region(f32[OFF_W2:OFF_B1], KS2 // 8)
    return out


# ---- f16f6: fp16 main term + fp6 (e2m3) correction terms with per-lane block scales (k_pack_mlp_f6) ----------------------------------
G6_1, G6_2 = 3, 2
F6_FLOATS = OFF_B1
F6_HI1, F6_HI2 = (KS1 // 8) * 4 * 64 * 4, (KS2 // 8) * 4 * 64 * 4
F6I_HI1, F6I_HI2 = 0, F6_HI1
F6I_Q1 = F6I_HI2 + F6_HI2
F6I_Q2 = F6I_Q1 + G6_1 * 4 * 3 * 256
F6I_SC = F6I_Q2 + G6_2 * 4 * 3 * 256


def f6_value(layer2, grp, term, e):
    """K value (index into the lane half's K order) carried by element e of a lane's 32-element fp6 operand, -1 = none."""
    if not layer2 and grp == G6_1 - 1:
        return -1 if e & 1 else 64 + (e >> 1)
    if term == 0:
        return 32 * grp + e
    return 32 * grp + (16 + (e >> 1) if e & 1 else (e >> 1))


def e2m3_codes(v):
    """v (already divided by the block scale) -> 6-bit e2m3 codes, round to nearest even, saturating at 7.5"""
    v = np.asarray(v, np.float64)
    s = np.where(v < 0, 32, 0)
    a = np.minimum(np.abs(v), 7.5)
    sub = np.rint(a * 8.0)
    e = np.where(a < 2.0, 0, np.where(a < 4.0, 1, 2))
    m = np.rint(a * 2.0 ** (3 - e))
    carry = m == 16
    ee = np.where(carry, e + 1, e)
    m = np.where(carry, 8, m)
    norm = np.where(ee > 2, 31, ((ee + 1) << 3) | (m.astype(np.int64) - 8))
    return (s | np.where(a < 1.0, sub.astype(np.int64), norm)).astype(np.uint32)


def e2m3_decode(c):
    c = np.asarray(c, np.int64)
    e, m = (c >> 3) & 3, c & 7
    f = np.where(e == 0, m / 8.0, (1.0 + m / 8.0) * 2.0 ** (e - 1))
    return np.where(c & 32, -f, f)


def f6_block(v, term):
    """v [..., 32] float32 values of one lane's operand -> (6 dwords [..., 6], scale byte [...]) as k_pack_mlp_f6 writes them"""
    v = np.asarray(v, np.float32)
    amax = np.abs(v).max(-1)
    E = np.where(amax > 0, np.frexp(np.where(amax > 0, amax, 1.0))[1] - 1, -100)
    E = np.maximum(E, -100)
    E = np.where(np.ldexp(amax.astype(np.float64), 2 - E) > 7.75, E + 1, E)
    codes = e2m3_codes(v.astype(np.float64) * np.ldexp(1.0, 2 - E)[..., None]).astype(np.uint64)
    bits = np.zeros(v.shape[:-1] + (3,), np.uint64)            # 192 bits as three 64-bit words
    for e in range(32):
        bit = 6 * e
        wi, sh = bit >> 6, bit & 63
        bits[..., wi] |= (codes[..., e] << np.uint64(sh)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        if sh > 58:
            bits[..., wi + 1] |= codes[..., e] >> np.uint64(64 - sh)
    dw = np.zeros(v.shape[:-1] + (6,), np.uint32)
    for i in range(6):
        dw[..., i] = ((bits[..., i >> 1] >> np.uint64(32 * (i & 1))) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    byte = np.clip((E - 2) + 127 - (2 if term == 0 else 13), 0, 254).astype(np.uint32)
    return dw, byte


def mlp_k_matrix(w, layer2):
    """[m-tile][lane][K] weights in the lane halves' K order (K = 96 for layer 1: the 80 values + one padding group's worth of zeros)"""
    f32 = pack_mlp(w)
    if layer2:
        frag, steps = f32[OFF_W2:OFF_B1], KS2 // 8
    else:
        frag, steps = f32[OFF_W1:OFF_W2], KS1 // 8
    kk = frag.reshape(steps * 2, 4, 64, 4).transpose(1, 2, 0, 3).reshape(4, 64, steps * 8)
    if not layer2:
        kk = np.concatenate([kk, np.zeros((4, 64, 16), np.float32)], -1)
    return kk


def pack_mlp_f6(w):
    """Fifth region of the packed blob (the f16f6 LDS image, csrc/ego_shade.hip): hi fragments of both layers, the fp6 operands
    [group][m-tile][quad][lane][4] (quad t: term t's dwords 0-3; quad 2: term 0's dwords 4-5, term 1's dwords 4-5), scale bytes [group][lane][term][m-tile]."""
    out = np.zeros(F6_FLOATS, np.uint32)
    for layer2, hi0, q0, groups in ((False, F6I_HI1, F6I_Q1, G6_1), (True, F6I_HI2, F6I_Q2, G6_2)):
        kk = mlp_k_matrix(w, layer2)
        steps = (KS2 if layer2 else KS1) // 8
        hi = kk.astype(np.float16)
        lo = kk - hi.astype(np.float32)
        hi_part = np.zeros((steps, 4, 64, 8), np.float16)
        for st in range(steps):
            hi_part[st] = hi[:, :, st * 8:st * 8 + 8]
        out[hi0:hi0 + steps * 4 * 64 * 4] = hi_part.reshape(-1).view(np.uint32)
        for grp in range(groups):
            sc0 = F6I_SC + ((G6_1 if layer2 else 0) + grp) * 128
            for term in range(2):
                src = lo if term == 0 else hi.astype(np.float32)
                v = np.zeros((4, 64, 32), np.float32)
                for e in range(32):
                    k = f6_value(layer2, grp, term, e)
                    if k >= 0:
                        v[:, :, e] = src[:, :, k]
                dw, byte = f6_block(v, term)
                for mt in range(4):
                    blk = out[q0 + (grp * 4 + mt) * 768: q0 + (grp * 4 + mt + 1) * 768].reshape(3, 64, 4)
                    blk[term, :, :] = dw[mt, :, :4]
                    blk[2, :, 2 * term:2 * term + 2] = dw[mt, :, 4:]
                    out[sc0 + term: sc0 + 128: 2] |= byte[mt] << np.uint32(8 * mt)
    return out.view(np.float32)


def f6_layer_reference(w, layer2, X):
    """What the f16f6 arithmetic computes for one layer before the bias, in float64, from the PACKED operands: X [n][K] activations in
    the lane halves' K order for both halves: X[h] [n][K].  Returns [n][128] (unit m * 32 + i).  Test infrastructure: checks that the
    packed blocks, the element orders and the scale bytes mean what the kernel assumes."""
    blob = pack_mlp_f6(w).view(np.uint32)
    hi0, q0, groups = (F6I_HI2, F6I_Q2, G6_2) if layer2 else (F6I_HI1, F6I_Q1, G6_1)
    steps = (KS2 if layer2 else KS1) // 8
    n = X[0].shape[0]
    out = np.zeros((n, 128))
    hi_frag = blob[hi0:hi0 + steps * 1024].view(np.float16).reshape(steps, 4, 64, 8).astype(np.float64)
    for h in range(2):
        x = np.asarray(X[h], np.float32)
        kpad = 32 * groups
        xp = np.concatenate([x, np.zeros((n, kpad - x.shape[1]), np.float32)], 1) if x.shape[1] < kpad else x
        xh = xp.astype(np.float16)
        xr = (xp - xh.astype(np.float32)).astype(np.float64)
        for mt in range(4):
            for i in range(32):
                lane = i + 32 * h
                wrow = np.concatenate([hi_frag[st, mt, lane] for st in range(steps)])
                acc = xh[:, :steps * 8].astype(np.float64) @ wrow
                for grp in range(groups):
                    sc0 = F6I_SC + ((G6_1 if layer2 else 0) + grp) * 128
                    half = (not layer2) and grp == G6_1 - 1
                    xg = xp[:, 32 * grp:32 * grp + (16 if half else 32)]
                    if layer2:
                        amax = np.abs(xg).max(1)
                    else:   # the kernel's rule for layer 1: 1 and the unbounded values (features, raw view direction)
                        ks = [k for k in range(32 * grp, 32 * grp + xg.shape[1]) if (k < 70 and k % 5 == 0) or 70 <= k < 73]
                        amax = np.maximum(np.abs(xp[:, ks]).max(1), np.float32(1.0))
                    eb = np.maximum(amax.view(np.uint32) >> 23, 14).astype(np.int64)
                    for term in range(2):
                        sword = int(blob[sc0 + lane * 2 + term]) >> (8 * mt)
                        blk = blob[q0 + (grp * 4 + mt) * 768: q0 + (grp * 4 + mt + 1) * 768].reshape(3, 64, 4)
                        dw = np.concatenate([blk[term, lane], blk[2, lane, 2 * term:2 * term + 2]])
                        big = sum(int(dw[j]) << (32 * j) for j in range(6))
                        a = e2m3_decode(np.array([(big >> (6 * e)) & 63 for e in range(32)])) * 2.0 ** ((sword & 255) - 127)
                        src = xh.astype(np.float64) if term == 0 else xr
                        scale = 2.0 ** (eb - 127 - (2 if term == 0 else 13))
                        bvals = np.zeros((n, 32))
                        for e in range(32):
                            k = f6_value(layer2, grp, term, e)
                            if k >= 0:
                                bvals[:, e] = e2m3_decode(e2m3_codes(src[:, k] / scale)) * 2.0 ** (eb - 127)
                        acc = acc + bvals @ a
                out[:, mt * 32 + i] += acc
    return out
