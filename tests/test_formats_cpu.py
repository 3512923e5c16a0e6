"""On-disk formats vs the reference's own dataset code (fixtures: make_golden.py formats; the two
input files under tests/golden/data/ are data files the reference ships)."""
import importlib
import os

import numpy as np
import torch

from conftest import GOLDEN

io = importlib.import_module("3dpointcloudattack_amd.dataset.cloud_io")
DATA = os.path.join(GOLDEN, "data")


def test_bosphorus_text_branch_matches_reference():
    fx = np.load(os.path.join(GOLDEN, "formats.npz"))
    np.random.seed(4242)
    xyz, rest = io.load_cloud_txt(os.path.join(DATA, "face0424.txt"), npoint=4000)     # ',' detected; u,v kept aside
    assert xyz.shape == (4000, 3) and rest.shape == (4000, 2)
    norm, center, dist = io.normalize_cloud(xyz)
    np.testing.assert_array_equal(norm, fx["bosphorus_pc"])                             # same shuffle, same arithmetic
    assert abs(np.max(np.linalg.norm(norm, axis=1)) - 1.0) < 1e-12 and np.allclose(norm.mean(0), 0, atol=1e-12)
    np.testing.assert_allclose(io.denormalize_cloud(norm, center, dist), xyz, rtol=0, atol=1e-9)


def test_read_pc_matches_reference(tmp_path):
    fx = np.load(os.path.join(GOLDEN, "formats.npz"))
    A, ori, tar = io.read_PC(0, DATA)
    np.testing.assert_array_equal(A, fx["advdata_A"])
    assert [ori, tar] == fx["advdata_ori_tar"].tolist() == [88, 63]
    # quirks kept: short file -> remaining rows repeat its last line; missing index -> zeros and (idx, idx)
    p = tmp_path / io.adv_filename(7, 3, 5)
    p.write_text("1.0 2.0 3.0\n4.0 5.0 6.0\n")
    B, o, t = io.read_PC(7, str(tmp_path), npoint=5)
    assert (o, t) == (3, 5) and np.array_equal(B, [[1, 2, 3]] + [[4, 5, 6]] * 4)
    Z, o, t = io.read_PC(9, str(tmp_path), npoint=5)
    assert (o, t) == (9, 9) and not Z.any()
    ds = io.AdvData_Dataset(str(tmp_path), npoint=5)
    assert len(ds) == 1 and np.array_equal(ds[7][0], B)


def test_file_to_file_round_trip(tmp_path):
    """read -> normalise -> (identity 'attack' that nudges z) -> de-normalise -> save with %.04f: extra columns
    untouched, coordinates within the format's 5e-5 quantisation, file re-readable by the same loader."""
    np.random.seed(1)
    seen = {}

    def attack_fn(pc):
        assert pc.shape == (1, 600, 3) and pc.dtype == torch.float32
        seen["pc"] = pc.clone()
        out = pc.clone()
        out[..., 2] += 0.01
        return out.numpy()
    path, adv = io.attack_cloud_file(attack_fn, os.path.join(DATA, "face0424.txt"), str(tmp_path / "out"),
                                     io.adv_filename(0, 105, 3), npoint=600)
    assert os.path.basename(path) == "0-105-3.txt"
    back = np.loadtxt(path)
    assert back.shape == (600, 5)
    np.testing.assert_allclose(back[:, :3], adv, atol=5.1e-5)
    np.random.seed(1)
    xyz, rest = io.load_cloud_txt(os.path.join(DATA, "face0424.txt"), npoint=600)
    np.testing.assert_array_equal(back[:, 3:], rest)
    _, center, dist = io.normalize_cloud(xyz)
    np.testing.assert_allclose(adv[:, 2] - xyz[:, 2], 0.01 * dist, rtol=1e-4)
    with open(path) as f:
        assert all(len(tok.split('.')[1]) == 4 for tok in f.readline().split())
    x2, r2 = io.load_cloud_txt(path)                                                    # whitespace detected
    assert x2.shape == (600, 3) and r2.shape == (600, 2)
