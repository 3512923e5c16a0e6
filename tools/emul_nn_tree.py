"""CPU emulation of nn_tree_min32 (3dpointcloudattack_amd/csrc/nn.hip): the lane algebra of the halving butterfly with the
instruction semantics measured by tools/exp/lane_probe.hip on gfx950. Prints ok when lane L ends up with min(c[L>>1])."""
import numpy as np
rng = np.random.default_rng(0)
c = [rng.random(64).astype(np.float32) for _ in range(32)]
truth = np.array([v.min() for v in c])
lanes = np.arange(64)
def swap32(v, s):   # lanes 32..63 of v <-> lanes 0..31 of s
    v2, s2 = v.copy(), s.copy()
    v2[32:] = s[:32]; s2[:32] = v[32:]
    return v2, s2
def swap16(v, s):   # odd rows of v <-> even rows of s
    v2, s2 = v.copy(), s.copy()
    for r in (0, 2):
        v2[(r+1)*16:(r+2)*16] = s[r*16:(r+1)*16]
        s2[r*16:(r+1)*16] = v[(r+1)*16:(r+2)*16]
    return v2, s2
def dpp(old, src, srcmap, bank_mask):
    out = old.copy()
    for L in range(64):
        bank = (L % 16) // 4
        if not (bank_mask >> bank) & 1: continue
        s = srcmap(L)
        if s is None: continue
        out[L] = src[s]
    return out
def ror8(L): r = L // 16 * 16; return r + ((L - r) - 8) % 16
def shl4(L): r = L // 16 * 16; return L + 4 if (L - r) + 4 < 16 else None
def shr4(L): r = L // 16 * 16; return L - 4 if (L - r) - 4 >= 0 else None
def qp(sel):
    return lambda L: (L // 4) * 4 + sel[L % 4]
for p in range(16):
    c[p], c[p+16] = swap32(c[p], c[p+16]); c[p] = np.minimum(c[p], c[p+16])
for p in range(8):
    c[p], c[p+8] = swap16(c[p], c[p+8]); c[p] = np.minimum(c[p], c[p+8])
for p in range(4):
    x = dpp(c[p+4], c[p], ror8, 0x3); y = dpp(c[p], c[p+4], ror8, 0xC); c[p] = np.minimum(x, y)
for p in range(2):
    x = dpp(c[p+2], c[p], shl4, 0x5); y = dpp(c[p], c[p+2], shr4, 0xA); c[p] = np.minimum(x, y)
bit1 = (lanes & 2) != 0
keep = np.where(bit1, c[1], c[0]); give = np.where(bit1, c[0], c[1])
c0 = np.minimum(keep, dpp(give, give, qp([2,3,0,1]), 0xF))
c0 = np.minimum(c0, dpp(c0, c0, qp([1,0,3,2]), 0xF))
print("ok" if np.array_equal(c0, truth[lanes >> 1]) else "MISMATCH", (c0 != truth[lanes>>1]).sum())
