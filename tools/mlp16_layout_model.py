#!/usr/bin/env python3
"""Host model of the 16x16x4 MFMA data movement of csrc/mlp16.hip (no GPU): the packed-image index formula, the chain
order of the k-steps, the transposed image, the sample-major publish / operand reads of the weight-gradient rounds and
the block -> gradient-tensor store map, each checked against plain matrix products.  Mirrors the C++ index arithmetic
line by line; run it after touching any layout there (tests/test_mlp16_layout_model.py does)."""
import numpy as np

LANES = np.arange(64)
I, G = LANES & 15, LANES >> 4


def mfma_16x16x4(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64, 4].  A[i][k] = a[i + 16k], B[k][j] = b[j + 16k], D[4g + r][j] -> acc[j + 16g][r]."""
    A = a.reshape(4, 16).T          # [i][k]
    B = b.reshape(4, 16)            # [k][j]
    D = A @ B                       # [16 i][16 j]
    out = acc.copy()
    for g in range(4):
        for r in range(4):
            out[16 * g + np.arange(16), r] += D[4 * g + r, :]
    return out


def pack16_layer(W, n_out, n_k, KS, natural, transposed):
    Q = max(n_out // 64, 1)
    img = np.zeros(KS * Q * 256, dtype=W.dtype)
    for idx in range(img.size):
        e, lane, q, ks = idx & 3, (idx >> 2) & 63, (idx >> 8) % Q, idx // (Q * 256)
        i, g = lane & 15, lane >> 4
        out = 16 * (4 * q + e) + i
        k = 4 * ks + g if natural else 16 * (ks >> 2) + 4 * g + (ks & 3)
        if out < n_out and k < n_k:
            img[idx] = W[k, out] if transposed else W[out, k]
    return img


def to_acc_layout(X, T):
    """X [16 samples][16 T features] -> [T][64 lanes][4]: lane (s, g) register r of tile t = X[s][16t + 4g + r]."""
    out = np.zeros((T, 64, 4), dtype=X.dtype)
    for t in range(T):
        for r in range(4):
            out[t, :, r] = X[I, 16 * t + 4 * G + r]
    return out


def from_acc_layout(acc):
    T = acc.shape[0]
    X = np.zeros((16, 16 * T), dtype=acc.dtype)
    for t in range(T):
        for r in range(4):
            X[I, 16 * t + 4 * G + r] = acc[t, :, r]
    return X


def chain_layer(img, T_in_ksteps, T_out, inp_acc=None, xr=None):
    """The k-loop of layer16 / first16: B operand = in[ks>>2][ks&3] (chain) or xr[ks] (first layer)."""
    Q = T_out // 4
    out = np.zeros((T_out, 64, 4), dtype=img.dtype)
    for ks in range(T_in_ksteps):
        b = xr[ks] if xr is not None else inp_acc[ks >> 2, :, ks & 3]
        for q in range(Q):
            w = img[((ks * Q + q) * 64 + LANES)[:, None] * 4 + np.arange(4)[None, :]]   # float4 per lane
            for e in range(4):
                out[4 * q + e] = mfma_16x16x4(w[:, e], b, out[4 * q + e])
    return out


def dw_job(A_tiles, B_tiles, TA, TB, RT=4, nwaves=4):
    """dw16: A_tiles / B_tiles = per-wave accumulator-layout tiles [wave][T][64][4]; rounds of RT tiles (RT = 4: one round of
    all 64 samples; RT = 2: two rounds).  Returns dW [16 TA][16 TB] assembled through the kernel's publish / read / store maps."""
    NBLK = TA * TB
    TOT = (NBLK + nwaves - 1) // nwaves
    NB = 4 if TB >= 4 else TB
    PERP = min(TOT, 16)
    NA = max(PERP // NB, 1)
    PR = TB // NB
    NPATCH = (TA // NA) * PR
    PASSES = (NPATCH + nwaves - 1) // nwaves
    PA, PB = TA * 16 + 20, TB * 16 + 20
    ROWS, NKS = RT * 16, RT * 4
    dW = np.zeros((16 * TA, 16 * TB))
    bias = np.zeros(16 * TA)
    for p in range(PASSES):
        for wave in range(nwaves):
            patch = p * nwaves + wave
            if patch >= NPATCH:
                continue
            a0, b0 = (patch // PR) * NA, (patch % PR) * NB
            acc = np.zeros((NA, NB, 64, 4))
            bsum = np.zeros((NA, 64))
            for t in range(nwaves // RT):
                X = np.zeros(ROWS * PA + ROWS * PB)
                for w in range(nwaves):          # publish: PubAcc16 into rows (w % RT) * 16 + s of round w // RT
                    if w // RT != t:
                        continue
                    rows = ((w % RT) * 16 + I) * PA
                    for tt in range(TA):
                        for r in range(4):
                            X[rows + tt * 16 + 4 * G + r] = A_tiles[w][tt][:, r]
                    rowsb = ROWS * PA + ((w % RT) * 16 + I) * PB
                    for tt in range(TB):
                        for r in range(4):
                            X[rowsb + tt * 16 + 4 * G + r] = B_tiles[w][tt][:, r]
                for ks in range(NKS):
                    for x in range(NA):
                        av = X[G * PA + a0 * 16 + I + 4 * ks * PA + x * 16]
                        bsum[x] += av
                        for y in range(NB):
                            bv = X[ROWS * PA + G * PB + b0 * 16 + I + 4 * ks * PB + y * 16]
                            acc[x, y] = mfma_16x16x4(av, bv, acc[x, y])
            for x in range(NA):
                for y in range(NB):
                    for r in range(4):
                        dW[(a0 + x) * 16 + 4 * G + r, (b0 + y) * 16 + I] = acc[x, y][:, r]
                bs = bsum[x].reshape(4, 16).sum(0)
                if b0 == 0:
                    bias[(a0 + x) * 16 + np.arange(16)] = bs
    return dW, bias


def self_check(md=64, D=7, seed=0):
    rng = np.random.default_rng(seed)
    T = md // 16
    Dp = (D + 3) & ~3
    KS0 = Dp // 4
    W0, W1 = rng.standard_normal((md, D)), rng.standard_normal((md, md))
    x = rng.standard_normal((16, D))
    # first layer, natural order: xr[ks][lane (s, g)] = x[s][4ks + g]
    xr = np.zeros((KS0, 64))
    for ks in range(KS0):
        c = 4 * ks + G
        xr[ks] = np.where(c < D, x[I, np.minimum(c, D - 1)], 0.0)
    h1 = chain_layer(pack16_layer(W0, md, D, KS0, True, False), KS0, T, xr=xr)
    assert np.allclose(from_acc_layout(h1), x @ W0.T), "first layer"
    # hidden layer, chain order
    h2 = chain_layer(pack16_layer(W1, md, md, md // 4, False, False), md // 4, T, inp_acc=h1)
    assert np.allclose(from_acc_layout(h2), from_acc_layout(h1) @ W1.T), "chain layer"
    # backward chain through the transposed image: dH1 = dZ2 @ W1
    dz = rng.standard_normal((16, md))
    dh = chain_layer(pack16_layer(W1, md, md, md // 4, False, True), md // 4, T, inp_acc=to_acc_layout(dz, T))
    assert np.allclose(from_acc_layout(dh), dz @ W1), "transposed chain layer"
    # weight gradient over a 64-sample group: 4 waves x 16 samples
    dZ = rng.standard_normal((4, 16, md))
    H = rng.standard_normal((4, 16, md))
    for RT in (4, 2):
        dW, db = dw_job([to_acc_layout(dZ[w], T) for w in range(4)], [to_acc_layout(H[w], T) for w in range(4)], T, T, RT)
        assert np.allclose(dW, np.einsum("wsi,wsj->ij", dZ, H)), "dW hidden"
        assert np.allclose(db, dZ.sum((0, 1))), "bias gradient"
    # first-layer weight gradient: B = the state rows (PubX16), one or two 16-column tiles
    TB = 1 if Dp <= 16 else 2
    X4 = rng.standard_normal((4, 16, D))
    Xpad = np.zeros((4, 16, 16 * TB))
    Xpad[:, :, :D] = X4
    dW0, _ = dw_job([to_acc_layout(dZ[w], T) for w in range(4)], [to_acc_layout(Xpad[w], TB) for w in range(4)], T, TB)
    assert np.allclose(dW0[:, :D], np.einsum("wsi,wsj->ij", dZ, X4)), "dW first layer"
    return True


if __name__ == "__main__":
    for md, D in ((64, 3), (64, 30), (128, 4)):
        self_check(md, D)
        print("ok", md, D)
