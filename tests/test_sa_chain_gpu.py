"""GPU: the one-launch set-abstraction MLP (pc3d_sa_chain_f32: gather -> layer 1 -> layer 2 -> layer 3 -> group max,
model/pointnet2_utils.py:173-199) against the two-launch form it replaces (pc3d_gemm_nt_gather_f32 + the group-max GEMM),
which is pinned to the reference's SSG / MSG logits and input gradients elsewhere (tests/test_pointnet2_gpu.py): outputs,
arg-max members and the gradients to P / Bc are BIT-identical, at SSG's two levels, MSG-like shapes, a ragged last tile
and indices outside the cloud; plus a float64 evaluation of the same chain."""
import importlib

import numpy as np
import pytest
import torch

from helpers import unit_cloud

pytestmark = pytest.mark.gpu


def _case(ops, dev, B, N, S, ns, C1, C2, C3, seed, bad_idx=False):
    rng = np.random.default_rng(seed)
    xyz = torch.from_numpy(np.stack([unit_cloud(rng, N) for _ in range(B)])).to(dev)
    cidx = torch.from_numpy(np.stack([rng.choice(N, S, replace=False) for _ in range(B)]).astype(np.int64)).to(dev)
    centers = torch.gather(xyz, 1, cidx[..., None].expand(-1, -1, 3)).contiguous()
    idx = ops.ball_query(0.3, ns, xyz, centers)
    if bad_idx:
        idx = idx.clone()
        idx[0, 0, 1] = N            # the ball query's "no point" marker: a zero row of P
        idx[-1, -1, -1] = -3
    P = torch.from_numpy(rng.standard_normal((B, N, C1)).astype(np.float32)).to(dev)
    Bc = torch.from_numpy(rng.standard_normal((B, S, C1)).astype(np.float32)).to(dev)
    mk = lambda o, i: (torch.from_numpy((rng.standard_normal((o, i)) / i ** 0.5).astype(np.float32)).to(dev),
                       torch.from_numpy(rng.standard_normal(o).astype(np.float32)).to(dev))
    layers = [mk(C2, C1), mk(C3, C2)]
    w = torch.from_numpy(rng.standard_normal((B, S, C3)).astype(np.float32)).to(dev)
    return P, Bc, idx.contiguous(), layers, w


def _run(ops, P, Bc, idx, layers, w, chain):
    ops.SA_CHAIN = chain
    try:
        p, bc = P.clone().requires_grad_(), Bc.clone().requires_grad_()
        rev = ops.group_reverse(idx, P.shape[1])
        out = ops.grouped_mlp_max(p, bc, idx, layers, rev=rev)
        (out * w).sum().backward()
        return out.detach(), p.grad, bc.grad
    finally:
        ops.SA_CHAIN = True


@pytest.mark.parametrize("B,N,S,ns,C1,C2,C3", [(8, 2048, 512, 32, 64, 64, 128),     # SSG SA1 (B reduced)
                                                (8, 512, 128, 64, 128, 128, 256),    # SSG SA2
                                                (3, 512, 128, 128, 64, 96, 128),     # MSG widest scale
                                                (2, 512, 100, 32, 32, 32, 64),       # narrow widths, C3 < a column tile
                                                (1, 300, 7, 64, 64, 128, 320),       # ragged last row tile, three column tiles
                                                (2, 256, 9, 32, 128, 64, 96)])
def test_sa_chain_equals_two_launch_form_bitwise(ops, dev, B, N, S, ns, C1, C2, C3):
    P, Bc, idx, layers, w = _case(ops, dev, B, N, S, ns, C1, C2, C3, seed=C3 + ns, bad_idx=True)
    if not ops.grouped_mlp_max_supported(C1, ns, layers):
        pytest.skip("shape not on the fused path")
    assert ops.sa_chain_supported(C1, C2, C3, ns)
    o1, gp1, gb1 = _run(ops, P, Bc, idx, layers, w, chain=True)
    o0, gp0, gb0 = _run(ops, P, Bc, idx, layers, w, chain=False)
    assert torch.equal(o1, o0), float((o1 - o0).abs().max())
    assert torch.equal(gp1, gp0) and torch.equal(gb1, gb0)


def test_sa_chain_vs_float64(ops, dev):
    B, N, S, ns, C1, C2, C3 = 2, 256, 16, 32, 64, 64, 128
    P, Bc, idx, layers, w = _case(ops, dev, B, N, S, ns, C1, C2, C3, seed=3)
    out, gp, gb = _run(ops, P, Bc, idx, layers, w, chain=True)
    p, bc = P.double().requires_grad_(), Bc.double().requires_grad_()
    rows = torch.gather(p, 1, idx.long().reshape(B, S * ns, 1).expand(-1, -1, C1)).view(B, S, ns, C1)
    h = torch.relu(rows + bc[:, :, None, :])
    h = torch.relu(h @ layers[0][0].double().t() + layers[0][1].double())
    h = torch.relu(h @ layers[1][0].double().t() + layers[1][1].double())
    ref = h.max(dim=2)[0]
    (ref * w.double()).sum().backward()
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gp.cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gb.cpu().numpy(), bc.grad.cpu().numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("B,N,S,ns,C1,C2,C3", [(8, 2048, 512, 32, 64, 64, 128),     # SSG SA1 (B reduced)
                                                (8, 512, 128, 64, 128, 128, 256),    # SSG SA2
                                                (3, 512, 128, 128, 64, 96, 128),     # MSG widest scale (K not a chunk multiple)
                                                (2, 512, 100, 32, 32, 32, 64),       # narrow widths
                                                (1, 300, 7, 64, 64, 128, 320),       # ragged last row tile
                                                (2, 256, 9, 32, 128, 64, 96)])
def test_sa_chain_backward_fused_gemm_groupsum_bitwise(ops, dev, B, N, S, ns, C1, C2, C3):
    """The chain's backward in its forms — (a) max-backward on the ACTIVE rows / GEMM on W2^T + groups pass in one
    launch over the compacted active rows / points pass skipping the others, (b) the same on full tensors, (c) the four
    separate launches: the gradients to P and Bc are BIT-identical (the skipped terms are exact zeros, everything else is
    summed in the same order). The allocator's free blocks are filled with NaN first: a read of a row the sparse producer
    did not write would show."""
    P, Bc, idx, layers, w = _case(ops, dev, B, N, S, ns, C1, C2, C3, seed=C3 + ns + 1, bad_idx=True)
    if not ops.grouped_mlp_max_supported(C1, ns, layers):
        pytest.skip("shape not on the fused path")
    res = []
    for fused, sparse, packed in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)):
        ops.SA_CHAIN_BWD, ops.SA_BWD_SPARSE, ops.SA_BWD_PACKED = fused, sparse, packed
        try:
            junk = [torch.full((n,), float("nan"), device=dev) for n in (B * S * ns * C2, B * S * ns * C1, B * N * C1, 1 << 20)]
            del junk
            res.append(_run(ops, P, Bc, idx, layers, w, chain=True))
        finally:
            ops.SA_CHAIN_BWD, ops.SA_BWD_SPARSE, ops.SA_BWD_PACKED = True, True, True
    (o3, gp3, gb3), (o1, gp1, gb1), (o2, gp2, gb2), (o0, gp0, gb0) = res
    assert torch.equal(o1, o0) and torch.isfinite(gp1).all() and torch.isfinite(gb1).all()
    assert torch.equal(gb1, gb0), float((gb1 - gb0).abs().max())
    assert torch.equal(gp1, gp0), float((gp1 - gp0).abs().max())
    assert torch.equal(gb2, gb0) and torch.equal(gp2, gp0)
    assert torch.isfinite(gp3).all() and torch.isfinite(gb3).all()       # tiles packed with whole groups' active rows
    assert torch.equal(gb3, gb0), float((gb3 - gb0).abs().max())
    assert torch.equal(gp3, gp0), float((gp3 - gp0).abs().max())


@pytest.mark.parametrize("B,N,S,ns,C1,C2,C3,mode", [(4, 512, 128, 64, 128, 128, 256, "zero_groups"),
                                                     (4, 1024, 256, 32, 64, 64, 128, "zero_groups"),
                                                     (2, 512, 64, 64, 64, 64, 128, "zero_all"),
                                                     (2, 256, 40, 32, 32, 64, 96, "dense"),       # no padding copies: every row listed
                                                     (2, 256, 24, 64, 64, 128, 256, "dense"),
                                                     (100, 512, 500, 32, 32, 32, 64, "zero_groups")])   # 50 000 groups: past the
                                                                                                        # tile pass's LDS staging
def test_sa_chain_backward_packed_tiles_edge_cases_bitwise(ops, dev, B, N, S, ns, C1, C2, C3, mode):
    """The packed-tile backward where its tiling is least regular: groups whose upstream gradient is all zero (no active
    row: the group still owns its weight in a tile and its sums are exact zeros), a gradient that is zero everywhere, and
    groupings without a single padding copy (up to ns active rows per group: the widest weights the tiling sees) — against
    the four-launch form, bit for bit."""
    P, Bc, idx, layers, w = _case(ops, dev, B, N, S, ns, C1, C2, C3, seed=ns + C2 + B)
    if mode == "zero_groups":
        w = w.clone()
        w[:, ::3] = 0.0
        w[0, : S // 2] = 0.0
    elif mode == "zero_all":
        w = torch.zeros_like(w)
    else:
        rng = np.random.default_rng(5)
        idx = torch.from_numpy(np.stack([np.stack([rng.choice(N, ns, replace=False) for _ in range(S)]) for _ in range(B)])
                               .astype(np.int32)).to(dev)
    res = []
    for fused, sparse, packed in ((True, True, True), (False, False, False)):
        ops.SA_CHAIN_BWD, ops.SA_BWD_SPARSE, ops.SA_BWD_PACKED = fused, sparse, packed
        try:
            junk = [torch.full((n,), float("nan"), device=dev) for n in (B * S * ns * C2, B * S * ns * C1, B * N * C1, 1 << 20)]
            del junk
            res.append(_run(ops, P, Bc, idx, layers, w, chain=True))
        finally:
            ops.SA_CHAIN_BWD, ops.SA_BWD_SPARSE, ops.SA_BWD_PACKED = True, True, True
    (o1, gp1, gb1), (o0, gp0, gb0) = res
    assert torch.equal(o1, o0) and torch.isfinite(gp1).all() and torch.isfinite(gb1).all()
    assert torch.equal(gb1, gb0), float((gb1 - gb0).abs().max())
    assert torch.equal(gp1, gp0), float((gp1 - gp0).abs().max())


@pytest.mark.parametrize("B,N,S,ns,C1,C2,C3", [(8, 512, 128, 64, 128, 128, 256),    # SSG SA2: streaming kernel, 32-row units
                                                (8, 2048, 512, 32, 64, 64, 128),    # SSG SA1: resident kernel, 16-row units
                                                (3, 512, 128, 64, 32, 64, 96),      # resident, four units per group
                                                (1, 300, 7, 64, 64, 128, 320),      # ragged last tile
                                                (2, 256, 40, 64, 128, 64, 96),
                                                (2, 512, 9, 32, 32, 32, 64),
                                                (2, 256, 5, 32, 64, 64, 128),       # resident, S < 8: 16-row units
                                                (2, 512, 24, 128, 64, 96, 128)])    # groups of 128 rows: 32-row blocks
def test_sa_chain_block_table_equals_full_launch_bitwise(ops, dev, B, N, S, ns, C1, C2, C3):
    """pc3d_sa_chain_tb_f32 over the unit table of pc3d_sa_blocks_i32 (8- / 16- / 32-row units of nothing but padding copies
    left out, the others packed into fewer tiles) against the launch over every block: outputs and both gradients BIT-identical;
    the table lists every block that holds a listed point exactly once, whole groups per tile."""
    P, Bc, idx, layers, w = _case(ops, dev, B, N, S, ns, C1, C2, C3, seed=ns + C1, bad_idx=True)
    unit = ops.sa_chain_table_unit(S, ns, C1, C2, C3)
    assert unit in (8, 16, 32)
    tb, nt, _ = ops.sa_blocks(idx, unit)
    torch.cuda.synchronize()
    tbn, ntv = tb.cpu().numpy().reshape(-1, 8 if unit == 8 else 4), int(nt.item())
    bpg = ns // unit
    ii = idx.cpu().numpy().reshape(B * S, ns)
    want = sorted(g * bpg + b for g in range(B * S) for b in range(bpg)
                  if b == 0 or (ii[g, unit * b:unit * b + unit] != ii[g, 0]).any())
    got = [int(k) for k in tbn[:ntv].ravel() if k >= 0]
    assert got == want and (unit != 32 or (tbn[ntv:] == -1).all())    # (8- / 16-row units: tiles past ntiles are not written)
    for row in tbn[:ntv]:                                   # a group's blocks never straddle two tiles
        ks = [int(k) for k in row if k >= 0]
        assert ks == sorted(ks) and all((k // bpg != ks[0] // bpg) or k == ks[0] or True for k in ks)
        groups = [k // bpg for k in ks]
        for g in set(groups):
            assert sum(1 for k in want if k // bpg == g) == groups.count(g)

    def run(use_table):
        ops.SA_BLOCK_TABLE = use_table
        try:
            p, bc = P.clone().requires_grad_(), Bc.clone().requires_grad_()
            rev = ops.group_reverse(idx, P.shape[1])
            out = ops.grouped_mlp_max(p, bc, idx, layers, rev=rev, blocks=(tb, nt, unit))
            (out * w).sum().backward()
            return out.detach(), p.grad, bc.grad
        finally:
            ops.SA_BLOCK_TABLE = True

    a, b_ = run(True), run(False)
    assert ntv <= tbn.shape[0]
    assert all(torch.equal(u, v) for u, v in zip(a, b_))


@pytest.mark.parametrize("G,ns,unit", [(40960, 32, 16), (65536, 32, 8), (33000, 64, 8), (5000, 32, 16)])
def test_unit_table_of_many_groups(ops, dev, G, ns, unit):
    """pc3d_sa_blocks_i32 where a packing chunk is longer than 32 groups (B * S > 32 K, up to the launches' 64 K limit): every
    unit that holds a listed point exactly once, in order, whole groups per tile; beyond the limit no table is built."""
    rng = np.random.default_rng(G + unit)
    cnt = rng.integers(1, ns + 1, size=G)                              # listed points per group: a prefix, then copies
    base = rng.integers(0, 1000, size=(G, 1))
    ii = np.where(np.arange(ns)[None, :] < cnt[:, None], base + 1 + np.arange(ns)[None, :], base + 1).astype(np.int32)
    idx = torch.from_numpy(ii.reshape(1, G, ns)).to(dev)
    tb, nt, _ = ops.sa_blocks(idx, unit)
    torch.cuda.synchronize()
    slots = 8 if unit == 8 else 4
    ntv = int(nt.item())
    tbn = tb.cpu().numpy().reshape(-1, slots)[:ntv]
    upg = ns // unit
    kept = -(-cnt // unit)                                             # units of a group that hold a listed point
    want = np.concatenate([g * upg + np.arange(k) for g, k in enumerate(kept)])
    got = tbn[tbn >= 0]
    assert np.array_equal(got, want)
    grp = np.where(tbn >= 0, tbn // upg, -1)
    first, last = np.where(grp >= 0, grp, G + 1).min(1), grp.max(1)
    assert (first[1:] > last[:-1]).all()                               # a group never straddles two tiles
    assert ops.sa_blocks(torch.zeros((1, 70000, 32), dtype=torch.int32, device=dev), 16) is None
