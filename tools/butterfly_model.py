#!/usr/bin/env python3
"""Host model of `half_sums16` (csrc/ppo_fused.hip): the register-halving lane butterfly behind the head weight gradient of the
fused PPO kernels.  Sixteen registers of a wave (one value per lane) are summed over the 32 lanes of each lane half; every level
adds a PAIR of registers into one, each lane group keeping the sum of the register it owns:

    rows (16 lanes)   v_permlane16_swap + add                 16 -> 8 registers
    8-lane groups     x + row_ror:8(x), bank-masked select     8 -> 4
    banks (4 lanes)   x + row_half_mirror(x), masked select    4 -> 2
    quad              two quad_perm adds                       (all four lanes of a quad hold the sum)

The model executes the cross-lane primitives on 64-element arrays exactly as the ISA defines them and checks the kernel's claim:
out[k] in lane (h, li) = sum over the half's 32 lanes of register 8 k + 4 bit2(li) + 2 bit3(li) + bit4(li).  No GPU needed."""
import numpy as np

LANES = 64


def permlane16_swap(vdst, src0):
    """v_permlane16_swap_b32: the odd rows of vdst trade places with the even rows of src0 (rows = 16 lanes)."""
    a, b = vdst.copy(), src0.copy()
    for row in (0, 2):
        lo, hi = slice(16 * row, 16 * row + 16), slice(16 * (row + 1), 16 * (row + 1) + 16)
        a[hi], b[lo] = src0[lo].copy(), vdst[hi].copy()
    return a, b


def row_ror(v, n):
    out = np.empty_like(v)
    for lane in range(LANES):
        row, i = divmod(lane, 16)
        out[lane] = v[16 * row + (i - n) % 16]   # rotate right: lane i receives lane i - n of its row
    return out


def row_half_mirror(v):
    out = np.empty_like(v)
    for lane in range(LANES):
        base, i = lane & ~7, lane & 7
        out[lane] = v[base + 7 - i]
    return out


def quad_perm(v, perm):
    out = np.empty_like(v)
    for lane in range(LANES):
        out[lane] = v[(lane & ~3) + perm[lane & 3]]
    return out


def bank_select(old, new, bank_mask):
    """v_mov_b32_dpp quad_perm:[0,1,2,3] bank_mask: `new` in the banks (groups of 4 lanes inside a row) of the mask, `old` elsewhere."""
    out = old.copy()
    for lane in range(LANES):
        if bank_mask >> ((lane % 16) // 4) & 1:
            out[lane] = new[lane]
    return out


def half_sums16(p):
    """p: [16][64].  Returns out: [2][64]."""
    a8 = []
    for i in range(8):
        x, y = permlane16_swap(p[2 * i], p[2 * i + 1])
        a8.append(x + y)
    a4 = [bank_select(a8[2 * i] + row_ror(a8[2 * i], 8), a8[2 * i + 1] + row_ror(a8[2 * i + 1], 8), 0xC) for i in range(4)]
    out = []
    for i in range(2):
        v = bank_select(a4[2 * i] + row_half_mirror(a4[2 * i]), a4[2 * i + 1] + row_half_mirror(a4[2 * i + 1]), 0xA)
        v = v + quad_perm(v, (1, 0, 3, 2))
        out.append(v + quad_perm(v, (2, 3, 0, 1)))
    return out


def owned_register(k, li):
    return 8 * k + 4 * ((li >> 2) & 1) + 2 * ((li >> 3) & 1) + ((li >> 4) & 1)


def feature_slot(k, h, li):
    """Where the kernel adds out[k]: feature index inside a 32-feature tile (accumulator register r of lane half h is feature
    (r & 3) + 8 (r >> 2) + 4 h)."""
    return 4 * h + ((li >> 3) & 1) * 2 + ((li >> 4) & 1) + 8 * ((li >> 2) & 1) + 16 * k


def self_check(seed=0):
    rng = np.random.default_rng(seed)
    p = rng.integers(-1000, 1000, size=(16, LANES)).astype(np.float64)   # integers: sums are exact in any order
    out = half_sums16(list(p))
    covered = set()
    for lane in range(LANES):
        h, li = lane >> 5, lane & 31
        for k in range(2):
            r = owned_register(k, li)
            assert out[k][lane] == p[r, 32 * h:32 * h + 32].sum(), (lane, k)
            assert feature_slot(k, h, li) == (r & 3) + 8 * (r >> 2) + 4 * h
            if li & 3 == 0:
                covered.add((h, r))
    assert covered == {(h, r) for h in range(2) for r in range(16)}   # the writer lanes (li & 3 == 0) cover every feature once
    return True


if __name__ == "__main__":
    print("half_sums16 model:", "ok" if self_check() else "FAILED")
