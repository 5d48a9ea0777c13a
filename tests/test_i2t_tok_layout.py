"""CPU check of the data layout of the token-owner image->token kernel (micro_sam_amd/csrc/decfold_tok.hip): the operand
image that fold_frag_kernel writes (fragment order, permuted score rows / output rows, folded scale and out_proj bias) and
the per-wave tile algorithm of i2t_tok_kernel are restated with numpy on the MFMA fragment maps of csrc/common.h
(A[row = l & 15][k = 8 (l >> 4) + i], B[k = 8 (l >> 4) + i][col = l & 15], C reg r = C[row = 4 (l >> 4) + r][col = l & 15])
and compared with the unfolded formulation of the reference (segment_anything TwoWayAttentionBlock step 4):
    out = LayerNorm(x + softmax(((x + pe) Wq^T + bq) k^T / 4) v Wo^T + bo).
No rounding is emulated: any index mistake shows up as an O(1) error."""
import math

import numpy as np
import pytest

C, CI, FRAG = 256, 128, 512          # FRAG in 16-bit elements (64 lanes x 8)
KT_OFF, KF_OFF, VF_OFF = 0, 4 * 2 * FRAG, 4 * 2 * FRAG + 4 * 8 * FRAG
OPER = VF_OFF + 2 * 16 * FRAG
SCALE = 0.25 * 1.4426950408889634


def fold_frag(ktok, vtok, wq, wo, bo, Nt):
    """fold_frag_kernel: ktok / vtok [Nt,128], wq [128,256], wo [256,128], bo [256] -> operand image [OPER]."""
    img = np.zeros(OPER)
    kk = np.zeros((8, CI)); vv = np.zeros((8, CI))
    kk[:Nt] = ktok; vv[:Nt] = vtok
    for c in range(256):
        ks, fg, ii = c >> 5, (c >> 3) & 3, c & 7
        for h in range(8):
            w = wq[h * 16:(h + 1) * 16, c]
            for t in range(8):
                acc = float(kk[t, h * 16:(h + 1) * 16] @ w)
                m, fr = 2 * (h >> 2) + (t >> 2), 4 * (h & 3) + (t & 3)
                img[KF_OFF + ((m * 8 + ks) * 64 + fg * 16 + fr) * 8 + ii] = acc * SCALE
    for j in range(8 * 8 * 16):
        h, t, d = j >> 7, (j >> 4) & 7, j & 15
        m, e, fr, fg = 2 * (h >> 2) + (t >> 2), (h >> 1) & 1, 4 * (h & 3) + (t & 3), 2 * (h & 1) + (d >> 3)
        img[KT_OFF + ((m * 2 + e) * 64 + fg * 16 + fr) * 8 + (d & 7)] = kk[t, h * 16 + d] * SCALE
    for c in range(256):
        cp, within = c >> 5, c & 31
        rho, ct = 4 * (within >> 3) + (within & 3), 2 * cp + ((within >> 2) & 1)
        for h in range(8):
            w = wo[c, h * 16:(h + 1) * 16]
            a, fg = h >> 2, h & 3
            for t in range(8):
                val = float(w @ vv[t, h * 16:(h + 1) * 16]) + bo[c] / 8 if t < Nt else 0.0
                img[VF_OFF + ((a * 16 + ct) * 64 + fg * 16 + rho) * 8 + t] = val
    return img


def mfma(afrag, bfrag, c):
    """c [64 lanes, 4 regs] += A B with the 16x16x32 fragment maps."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, (l >> 4) * 8:(l >> 4) * 8 + 8] = afrag[l]
        B[(l >> 4) * 8:(l >> 4) * 8 + 8, l & 15] = bfrag[l]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[(l >> 4) * 4 + r, l & 15]
    return out


def tile_kernel(x_tile, tab_tile, img, lnw, lnb, Nt, eps=1e-5):
    """i2t_tok_kernel on one 16-token tile: x_tile [16,256], tab_tile [16,128] -> [16,256]."""
    lanes = np.arange(64); fr = lanes & 15; fg = lanes >> 4
    frag = lambda off: img[off:off + FRAG].reshape(64, 8)
    b = [np.stack([x_tile[fr[l], 32 * s + 8 * fg[l]:32 * s + 8 * fg[l] + 8] for l in range(64)]) for s in range(8)]
    tb = [np.stack([tab_tile[fr[l], 32 * s + 8 * fg[l]:32 * s + 8 * fg[l] + 8] for l in range(64)]) for s in range(4)]
    sel = []
    for e in range(2):
        f = np.zeros((64, 8))
        for l in range(64):
            if fg[l] == (fr[l] >> 2):
                f[l, 4 * e + (fr[l] & 3)] = 1.0
        sel.append(f)
    s = []
    for m in range(4):
        init = np.zeros((64, 4))
        for r in range(4):
            if 4 * (m & 1) + r >= Nt:
                init[:, r] = -1e30
        s.append(init)
    for g in range(10):
        for m in range(4):
            if g < 8:
                s[m] = mfma(frag(KF_OFF + (m * 8 + g) * FRAG), b[g], s[m])
            else:
                s[m] = mfma(frag(KT_OFF + (m * 2 + (g - 8)) * FRAG), tb[2 * (m >> 1) + (g - 8)], s[m])
    o = [mfma(sel[ct & 1], b[ct >> 1], np.zeros((64, 4))) for ct in range(16)]
    pk = []
    for a2 in range(2):
        v8 = np.concatenate([s[2 * a2], s[2 * a2 + 1]], axis=1)            # [64, 8]
        e8 = np.exp2(v8 - v8.max(1, keepdims=True))
        pk.append(e8 / e8.sum(1, keepdims=True))
    for q in range(8):
        for j in range(4):
            ct = 4 * (q & 3) + j
            o[ct] = mfma(frag(VF_OFF + ((q >> 2) * 16 + ct) * FRAG), pk[q >> 2], o[ct])
    vals = np.stack(o, axis=1)                                               # [64, 16, 4]
    s1 = vals.reshape(64, -1).sum(1); s2 = (vals.reshape(64, -1) ** 2).sum(1)
    s1 = s1 + s1[lanes ^ 16]; s2 = s2 + s2[lanes ^ 16]
    s1 = s1 + s1[lanes ^ 32]; s2 = s2 + s2[lanes ^ 32]
    mean = s1 / C; rstd = 1 / np.sqrt(np.maximum(s2 / C - mean * mean, 0) + eps)
    out = np.zeros((16, 256))
    for l in range(64):
        for c2 in range(8):
            ch = 32 * c2 + 8 * fg[l]
            x8 = np.concatenate([o[2 * c2][l], o[2 * c2 + 1][l]])
            out[fr[l], ch:ch + 8] = (x8 - mean[l]) * rstd[l] * lnw[ch:ch + 8] + lnb[ch:ch + 8]
    return out


@pytest.mark.parametrize("Nt", [7, 8, 3, 1])
def test_token_owner_layout_matches_the_reference_formula(Nt):
    g = np.random.default_rng(10 + Nt)
    x = g.standard_normal((16, 256)); pe = g.standard_normal((16, 256))
    wq = g.standard_normal((128, 256)) / 16; bq = g.standard_normal(128)
    wo = g.standard_normal((256, 128)) / math.sqrt(128); bo = g.standard_normal(256)
    lnw = g.standard_normal(256) * 0.2 + 1; lnb = g.standard_normal(256)
    ktok = g.standard_normal((Nt, 128)); vtok = g.standard_normal((Nt, 128))
    tab = pe @ wq.T + bq
    out = tile_kernel(x, tab, fold_frag(ktok, vtok, wq, wo, bo, Nt), lnw, lnb, Nt)
    q = (x @ wq.T + tab).reshape(16, 8, 16).transpose(1, 0, 2)                 # [8 heads, 16 tokens, 16]
    kh = ktok.reshape(Nt, 8, 16).transpose(1, 0, 2); vh = vtok.reshape(Nt, 8, 16).transpose(1, 0, 2)
    sc = q @ kh.transpose(0, 2, 1) / 4.0
    a = np.exp(sc - sc.max(-1, keepdims=True)); a /= a.sum(-1, keepdims=True)
    attn = (a @ vh).transpose(1, 0, 2).reshape(16, 128)
    y = x + attn @ wo.T + bo
    ref = (y - y.mean(1, keepdims=True)) / np.sqrt(y.var(1, keepdims=True) + 1e-5) * lnw + lnb
    assert np.abs(out - ref).max() < 1e-9


# ---------------------------------------------------------------------------------------------------------------------
# token -> image attention in token-owner form (t2i_block / fold_attnfrag_kernel / wv_frag_kernel of decfold_tok.hip)
QD_OFF, QF_OFF = 0, 4 * FRAG
AOPER = QF_OFF + 4 * 8 * FRAG


def fold_attnfrag(qtok, wk, Nt):
    img = np.zeros(AOPER)
    qq = np.zeros((8, CI)); qq[:Nt] = qtok
    for c in range(256):
        ks, fg, ii = c >> 5, (c >> 3) & 3, c & 7
        for h in range(8):
            w = wk[h * 16:(h + 1) * 16, c]
            for t in range(8):
                ct, fr = h >> 1, (h & 1) * 8 + t
                img[QF_OFF + ((ct * 8 + ks) * 64 + fg * 16 + fr) * 8 + ii] = float(qq[t, h * 16:(h + 1) * 16] @ w) * SCALE
    for j in range(8 * 8 * 16):
        h, t, d = j >> 7, (j >> 4) & 7, j & 15
        ct, fr, fg = h >> 1, (h & 1) * 8 + t, 2 * (h & 1) + (d >> 3)
        img[QD_OFF + (ct * 64 + fg * 16 + fr) * 8 + (d & 7)] = qq[t, h * 16 + d] * SCALE
    return img


def wv_frag(wv):
    out = np.zeros(8 * 8 * FRAG)
    for row in range(128):
        for ch in range(32):
            dv, fr, ks, fg = row >> 4, row & 15, ch >> 2, ch & 3
            o = ((dv * 8 + ks) * 64 + fg * 16 + fr) * 8
            out[o:o + 8] = wv[row, ch * 8:ch * 8 + 8]
    return out


def t2i_tile(keys_tile, tab_tile, aimg, wvimg, st):
    """t2i_block on one 16-token tile; st = dict(o [8][64,4], m [4][64], l [4][64])."""
    lanes = np.arange(64); fr = lanes & 15; fg = lanes >> 4
    y = [np.stack([keys_tile[fr[l], 32 * s + 8 * fg[l]:32 * s + 8 * fg[l] + 8] for l in range(64)]) for s in range(8)]
    tk = [np.stack([tab_tile[fr[l], 32 * s + 8 * fg[l]:32 * s + 8 * fg[l] + 8] for l in range(64)]) for s in range(4)]
    frag = lambda img, off: img[off:off + FRAG].reshape(64, 8)
    s = [np.zeros((64, 4)) for _ in range(4)]
    for g in range(9):
        for ct in range(4):
            if g < 8:
                s[ct] = mfma(y[g], frag(aimg, QF_OFF + (ct * 8 + g) * FRAG), s[ct])
            else:
                s[ct] = mfma(tk[ct], frag(aimg, QD_OFF + ct * FRAG), s[ct])
    pb = []
    for ct in range(4):
        mloc = s[ct].max(1)
        mx = np.maximum(mloc, mloc[lanes ^ 16]); mx = np.maximum(mx, mx[lanes ^ 32])
        mn = np.maximum(st["m"][ct], mx); alpha = np.exp2(st["m"][ct] - mn)
        st["m"][ct] = mn; st["l"][ct] = st["l"][ct] * alpha
        st["o"][2 * ct] = st["o"][2 * ct] * alpha[:, None]; st["o"][2 * ct + 1] = st["o"][2 * ct + 1] * alpha[:, None]
        e = np.exp2(s[ct] - st["m"][ct][:, None])
        st["l"][ct] = st["l"][ct] + e.sum(1)
        pb.append(np.concatenate([e, np.zeros((64, 4))], axis=1))
    for h in range(8):
        v = np.zeros((64, 4))
        for ks in range(8):
            v = mfma(y[ks], frag(wvimg, (h * 8 + ks) * FRAG), v)
        va = np.concatenate([v, np.zeros((64, 4))], axis=1)
        st["o"][h] = mfma(va, pb[h >> 1], st["o"][h])


@pytest.mark.parametrize("Nt", [7, 8, 2])
def test_token_owner_attention_layout_matches_the_reference_formula(Nt):
    g = np.random.default_rng(20 + Nt)
    ntile = 3
    keys = g.standard_normal((16 * ntile, 256)); pe = g.standard_normal((16 * ntile, 256))
    wk = g.standard_normal((128, 256)) / 16; bk = g.standard_normal(128)
    wv = g.standard_normal((128, 256)) / 16; bv = g.standard_normal(128)
    qtok = g.standard_normal((Nt, 128)) * 1.5
    tab = pe @ wk.T + bk
    aimg, wvimg = fold_attnfrag(qtok, wk, Nt), wv_frag(wv)
    st = dict(o=[np.zeros((64, 4)) for _ in range(8)], m=[np.full(64, -1e30) for _ in range(4)], l=[np.zeros(64) for _ in range(4)])
    for n in range(ntile):
        t2i_tile(keys[16 * n:16 * n + 16], tab[16 * n:16 * n + 16], aimg, wvimg, st)
    lanes = np.arange(64)
    out = np.zeros((Nt, 128))
    for ct in range(4):
        st["l"][ct] = st["l"][ct] + st["l"][ct][lanes ^ 16]; st["l"][ct] = st["l"][ct] + st["l"][ct][lanes ^ 32]
    for h in range(8):
        for l in range(64):
            fr, fgl = l & 15, l >> 4
            if (fr >> 3) == (h & 1) and (fr & 7) < Nt:
                for r in range(4):
                    out[fr & 7, h * 16 + fgl * 4 + r] = st["o"][h][l, r] / st["l"][h >> 1][l] + bv[h * 16 + fgl * 4 + r]
    K = keys @ wk.T + tab; V = keys @ wv.T + bv
    qh = qtok.reshape(Nt, 8, 16).transpose(1, 0, 2); kh = K.reshape(-1, 8, 16).transpose(1, 0, 2); vh = V.reshape(-1, 8, 16).transpose(1, 0, 2)
    sc = qh @ kh.transpose(0, 2, 1) / 4.0
    a = np.exp(sc - sc.max(-1, keepdims=True)); a /= a.sum(-1, keepdims=True)
    ref = (a @ vh).transpose(1, 0, 2).reshape(Nt, 128)
    assert np.abs(out - ref).max() < 1e-9


# ---------------------------------------------------------------------------------------------------------------------
# second form of the chained attention (i2t0_t2i_v2): the LayerNorm affine is folded into the operands (gamma into Q' and Wv, beta
# into per-column constants) and the V projection is taken BEFORE the LayerNorm by linearity:
#     x = src + P0 V''0 (+ bo inside V''0),  x^ = (x - mean) rstd,   V = x^ (gamma Wv)^T
#       = rstd (src (gWv)^T + P0 (V''0 (gWv)^T) - mean rowsum(gWv)) = rstd (tabV + P0 M - mean gd)
# tabV [4096,128] and gd are prompt independent, M [64,128] = per-prompt 16-MAC products with WoWv = (gWv) Wo.
def test_attention_v2_algebra_and_layouts():
    g = np.random.default_rng(77)
    Nt, ntile = 7, 2
    T = 16 * ntile
    src = g.standard_normal((T, 256)); pe = g.standard_normal((T, 256))
    # layer-0 image->token block
    wq0 = g.standard_normal((128, 256)) / 16; bq0 = g.standard_normal(128)
    wo0 = g.standard_normal((256, 128)) / math.sqrt(128); bo0 = g.standard_normal(256)
    gam = g.standard_normal(256) * 0.2 + 1; bet = g.standard_normal(256)
    k0 = g.standard_normal((Nt, 128)); v0 = g.standard_normal((Nt, 128))
    # layer-1 token->image attention
    wk = g.standard_normal((128, 256)) / 16; bk = g.standard_normal(128)
    wv = g.standard_normal((128, 256)) / 16; bv = g.standard_normal(128)
    qtok = g.standard_normal((Nt, 128)) * 1.5
    tabk = pe @ wk.T + bk
    q0 = src @ wq0.T + (pe @ wq0.T + bq0)
    # ---- reference: keys1 = LN(src + attn0 Wo^T + bo) gamma + beta, then the attention on keys1
    qh = q0.reshape(T, 8, 16).transpose(1, 0, 2)
    kh = k0.reshape(Nt, 8, 16).transpose(1, 0, 2); vh = v0.reshape(Nt, 8, 16).transpose(1, 0, 2)
    sc = qh @ kh.transpose(0, 2, 1) / 4.0
    p0 = np.exp(sc - sc.max(-1, keepdims=True)); p0 /= p0.sum(-1, keepdims=True)               # [8, T, Nt]
    x = src + (p0 @ vh).transpose(1, 0, 2).reshape(T, 128) @ wo0.T + bo0
    mean = x.mean(1, keepdims=True); rstd = 1 / np.sqrt(x.var(1, keepdims=True) + 1e-5)
    keys1 = (x - mean) * rstd * gam + bet
    K = keys1 @ wk.T + tabk; V = keys1 @ wv.T + bv
    qh1 = qtok.reshape(Nt, 8, 16).transpose(1, 0, 2); kh1 = K.reshape(T, 8, 16).transpose(1, 0, 2); vh1 = V.reshape(T, 8, 16).transpose(1, 0, 2)
    s1 = qh1 @ kh1.transpose(0, 2, 1) / 4.0
    a1 = np.exp(s1 - s1.max(-1, keepdims=True)); a1 /= a1.sum(-1, keepdims=True)
    ref = (a1 @ vh1).transpose(1, 0, 2).reshape(Nt, 128)
    # ---- the kernel's form.  Prompt-independent tables:
    wvg = wv * gam[None, :]
    tabv = src @ wvg.T                                            # [T,128]
    gd = wvg.sum(1); bwv = bv + wv @ bet
    wowv = wvg @ wo0                                              # [128 d, 128 e]
    cd = wvg @ bo0
    # per prompt: M[(h,t)][d] = sum_e v0[t][16h+e] WoWv[d][16h+e] + cd[d] / 8   (the bo / 8 of V''0)
    M = np.zeros((64, 128))
    for h in range(8):
        for t in range(Nt):
            M[h * 8 + t] = wowv[:, h * 16:(h + 1) * 16] @ v0[t, h * 16:(h + 1) * 16] + cd / 8
    # Q' with gamma folded, column constants beta . Q'
    qp = np.zeros((64, 256))
    for h in range(8):
        for t in range(Nt):
            qp[h * 8 + t] = wk[h * 16:(h + 1) * 16].T @ qtok[t, h * 16:(h + 1) * 16]
    qpg = qp * gam[None, :] * SCALE
    colc = (qp @ bet) * SCALE
    # lane-level: fragments of M ([a2][dv]: B[k = (h,t) = 32 a2 + 8 fg + i][col d = 16 dv + fr]) and tabV in blocked-C layout
    lanes = np.arange(64); fr = lanes & 15; fg = lanes >> 4
    mfrag = lambda a2, dv: np.stack([M[32 * a2 + 8 * fg[l]:32 * a2 + 8 * fg[l] + 8, 16 * dv + fr[l]] for l in range(64)])
    st = dict(o=[np.zeros((64, 4)) for _ in range(8)], m=[np.full(64, -1e30) for _ in range(4)], l=[np.zeros(64) for _ in range(4)])
    for n in range(ntile):
        sl = slice(16 * n, 16 * n + 16)
        xh = (x[sl] - mean[sl]) * rstd[sl]                         # x^ of the tile (from the layer-0 block)
        # P0 packed as the layer-0 block leaves it: lane (token fr, group fg): head 4 a2 + fg, tokens 0..7
        pk = [np.stack([np.concatenate([p0[4 * a2 + fg[l], 16 * n + fr[l], :], np.zeros(8 - Nt)]) for l in range(64)]) for a2 in range(2)]
        y = [np.stack([xh[fr[l], 32 * s_ + 8 * fg[l]:32 * s_ + 8 * fg[l] + 8] for l in range(64)]) for s_ in range(8)]
        tk = [np.stack([tabk[16 * n + fr[l], 32 * s_ + 8 * fg[l]:32 * s_ + 8 * fg[l] + 8] for l in range(64)]) for s_ in range(4)]
        s = [np.tile(np.array([colc[16 * ct + fr[l]] for l in range(64)])[:, None], (1, 4)) for ct in range(4)]     # column constants
        qf = lambda ct, ks: np.stack([qpg[16 * ct + fr[l], 32 * ks + 8 * fg[l]:32 * ks + 8 * fg[l] + 8] for l in range(64)])
        qdf = []
        for ct in range(4):
            f = np.zeros((64, 8))
            for l in range(64):
                h, t = 2 * ct + (fr[l] >> 3), fr[l] & 7
                if (fg[l] >> 1) == (fr[l] >> 3) and t < Nt:
                    f[l] = qtok[t, 32 * ct + 8 * fg[l]:32 * ct + 8 * fg[l] + 8] * SCALE
            qdf.append(f)
        for ct in range(4):
            for ks in range(8):
                s[ct] = mfma(y[ks], qf(ct, ks), s[ct])
            s[ct] = mfma(tk[ct], qdf[ct], s[ct])
        pb = []
        for ct in range(4):
            mx = s[ct].max(1); mx = np.maximum(mx, mx[lanes ^ 16]); mx = np.maximum(mx, mx[lanes ^ 32])
            mn = np.maximum(st["m"][ct], mx); al = np.exp2(st["m"][ct] - mn)
            st["m"][ct] = mn; st["l"][ct] *= al; st["o"][2 * ct] *= al[:, None]; st["o"][2 * ct + 1] *= al[:, None]
            e = np.exp2(s[ct] - mn[:, None]); st["l"][ct] = st["l"][ct] + e.sum(1)
            pb.append(np.concatenate([e, np.zeros((64, 4))], axis=1))
        # V phase: A = P0 (same registers as pk), B = M fragments; then the per-row / per-column correction
        row_mean = np.stack([[mean[16 * n + 4 * fg[l] + r, 0] for r in range(4)] for l in range(64)])     # what the shuffles fetch
        row_rstd = np.stack([[rstd[16 * n + 4 * fg[l] + r, 0] for r in range(4)] for l in range(64)])
        for dv in range(8):
            v = np.zeros((64, 4))
            for a2 in range(2):
                v = mfma(pk[a2], mfrag(a2, dv), v)
            tv = np.stack([[tabv[16 * n + 4 * fg[l] + r, 16 * dv + fr[l]] for r in range(4)] for l in range(64)])   # blocked-C tile
            gdl = np.array([gd[16 * dv + fr[l]] for l in range(64)])[:, None]
            v = row_rstd * (v + tv - row_mean * gdl)
            va = np.concatenate([v, np.zeros((64, 4))], axis=1)
            st["o"][dv] = mfma(va, pb[dv >> 1], st["o"][dv])
    out = np.zeros((Nt, 128))
    for ct in range(4):
        st["l"][ct] = st["l"][ct] + st["l"][ct][lanes ^ 16]; st["l"][ct] = st["l"][ct] + st["l"][ct][lanes ^ 32]
    for h in range(8):
        for l in range(64):
            if (fr[l] >> 3) == (h & 1) and (fr[l] & 7) < Nt:
                for r in range(4):
                    d = h * 16 + fg[l] * 4 + r
                    out[fr[l] & 7, d] = st["o"][h][l, r] / st["l"][h >> 1][l] + bwv[d]
    assert np.abs(out - ref).max() < 1e-9
