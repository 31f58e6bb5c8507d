#!/usr/bin/env python3
"""Host model of the 16-lane-tile policy forward (csrc/rollout_policy.hpp: policy_forward16): the fused rollout / evaluation /
off-policy kernels run v_mfma_f32_16x16x4_f32 on the packed image that was laid out for v_mfma_f32_32x32x2_f32 (mlp_device.hpp:
pack_mfma / pack_vec / pack_first).  The model packs random weights with the packers' formulas, then evaluates a layer exactly as
the kernel addresses it -- lane (c = lane & 15, g = lane >> 4) owns features 16 ot' + 4 g + r of sample c, k-step (ot_k, r_k)
multiplies feature 16 ot_k + 4 g + r_k, the A fragment of output tile ot' is component ot' >> 1 of the OT-wide word
((kt 16 + s) 64 + lane32) with kt = ot_k >> 1, s = r_k + 8 (ot_k & 1) + 4 (g >> 1), lane32 = 16 (ot' & 1) + c + 32 (g & 1) --
and compares with W x + b.  No GPU needed."""
import numpy as np


def feat32(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def pack_mfma(W, KT, OT):
    """dst[(ks * 64 + lane) * OT + ot] = W[ot * 32 + (lane & 31)][kt * 32 + feat32(s, lane >> 5)], ks = kt * 16 + s"""
    dst = np.empty(KT * 16 * 64 * OT)
    for idx in range(dst.size):
        ot, lane, ks = idx % OT, (idx // OT) & 63, idx // (OT * 64)
        kt, s = ks >> 4, ks & 15
        dst[idx] = W[ot * 32 + (lane & 31), kt * 32 + feat32(s, lane >> 5)]
    return dst


def pack_vec(v, OT):
    dst = np.empty(OT * 32)
    for idx in range(dst.size):
        h, r, ot = idx // (OT * 16), idx & 15, (idx >> 4) % OT
        dst[idx] = v[ot * 32 + feat32(r, h)]
    return dst


def vec16_off(OT32, ot, g):
    return (g & 1) * (OT32 * 16) + (ot >> 1) * 16 + 8 * (ot & 1) + 4 * (g >> 1)


def layer16_on32(img, bias_img, KT32, OT32, x):
    """x: [K][16 samples] -> [O][16].  Executes the 64 lanes' MFMA operands as the kernel forms them."""
    K, O = KT32 * 32, OT32 * 32
    out = np.zeros((O, 16))
    lanes = [(l & 15, l >> 4) for l in range(64)]
    # accumulator init: lane (c, g) reads the 4 bias values of tile ot at vec16_off
    for c, g in lanes:
        for ot in range(OT32 * 2):
            off = vec16_off(OT32, ot, g)
            out[16 * ot + 4 * g:16 * ot + 4 * g + 4, c] = bias_img[off:off + 4]
    for ok in range(KT32 * 2):
        for rk in range(4):
            word = ((ok >> 1) * 16 + rk + 8 * (ok & 1)) * 64
            # one MFMA per output tile: D[i][j] += sum_k A[i][k] B[k][j];  lane l gives A[l & 15][l >> 4] and B[l >> 4][l & 15]
            A = np.zeros((OT32 * 2, 16, 4))
            B = np.zeros((4, 16))
            for c, g in lanes:
                base = c + 32 * (g & 1) + 256 * (g >> 1)
                B[g, c] = x[16 * ok + 4 * g + rk, c]
                for o in range(OT32):
                    A[2 * o, c, g] = img[(base + word) * OT32 + o]
                    A[2 * o + 1, c, g] = img[(base + word + 16) * OT32 + o]
            for ot in range(OT32 * 2):
                out[16 * ot:16 * ot + 16] += A[ot] @ B
    return out


def self_check(KT32=4, OT32=4, seed=0):
    rng = np.random.default_rng(seed)
    K, O = KT32 * 32, OT32 * 32
    W = rng.integers(-8, 8, size=(O, K)).astype(np.float64)
    b = rng.integers(-8, 8, size=O).astype(np.float64)
    x = rng.integers(-8, 8, size=(K, 16)).astype(np.float64)
    got = layer16_on32(pack_mfma(W, KT32, OT32), pack_vec(b, OT32), KT32, OT32, x)
    assert np.array_equal(got, W @ x + b[:, None])
    return True


if __name__ == "__main__":
    for kt, ot in ((4, 4), (4, 2), (2, 2), (2, 1)):
        print(f"layer {kt * 32} -> {ot * 32}:", "ok" if self_check(kt, ot) else "FAILED")
