"""On-disk formats of the step either side of the attack loop: plain-text point clouds in,
normalised float tensors to the attack, de-normalised ``%.04f`` text out — the conventions of
dataset/bosphorus_dataset.py:23-27,59-84, attack/KNN/Eval_KNN.py:60-96, attack/additional_exp/Test_CW.py:69-118 and
dataset/AdvData_dataset.py:21-38. Host-side numpy only (file parsing is not GPU work); the attack in the middle is
whatever attacker object the caller passes (this package's CW / CWKNN / CWTAOF / additional_exp.CW).
"""
import os

import numpy as np
import torch


def rand_row(array, dim_needed):
    """dataset/bosphorus_dataset.py:23-27 — `dim_needed` rows in the order of a shuffle drawn from numpy's global RNG."""
    row_sequence = np.arange(array.shape[0])
    np.random.shuffle(row_sequence)
    return array[row_sequence[0:dim_needed], :]


def load_cloud_txt(path, delimiter=None, npoint=None):
    """x,y,z[,u,v,...] rows -> (xyz [N,3] float64, rest [N,C-3] float64). delimiter None: ',' if the first line has a
    comma (AddData/*.txt, bosphorus_dataset.py:61) else whitespace (the attacks' own outputs, Eval_KNN.py:62).
    npoint: keep that many rows via rand_row (the reference keeps 4000, :62 / Eval_KNN.py:63)."""
    if delimiter is None:
        with open(path) as f:
            delimiter = ',' if ',' in f.readline() else None
    data = np.loadtxt(path, delimiter=delimiter, ndmin=2)
    if npoint is not None:
        data = rand_row(data, npoint)
    return data[:, 0:3], data[:, 3:]


def normalize_cloud(xyz):
    """bosphorus_dataset.py:70-76 — NaNs to 0, centre on the mean, scale the farthest point to the unit sphere.
    Returns (normalised [N,3] float64, center [1,3], dist scalar) so the result can be mapped back."""
    xyz = np.array(xyz, dtype=np.float64)
    if np.any(np.isnan(xyz)):
        xyz[np.isnan(xyz)] = 0
    center = np.expand_dims(np.mean(xyz, axis=0), 0)
    xyz = xyz - center
    dist = np.max(np.sqrt(np.sum(xyz ** 2, axis=1)), 0)
    return xyz / dist, center, dist


def denormalize_cloud(xyz, center, dist):
    """Eval_KNN.py:89 / Test_CW.py:104."""
    return np.asarray(xyz) * dist + center


def adv_filename(idx, orig, target):
    """`idx-orig-target.txt` (the naming AdvData_dataset.read_PC parses, dataset/AdvData_dataset.py:27-31)."""
    return '{}-{}-{}.txt'.format(int(idx), int(orig), int(target))


def save_cloud_txt(path, xyz, rest=None):
    """Eval_KNN.py:90-93 / Test_CW.py:106-111 — optional extra columns appended, fmt='%.04f', space separated."""
    out = np.asarray(xyz, dtype=np.float64)
    if rest is not None and np.size(rest):
        out = np.hstack((out, np.asarray(rest, dtype=np.float64)))
    np.savetxt(path, out, fmt='%.04f')
    return path


def read_PC(idx, path, npoint=4000):
    """dataset/AdvData_dataset.py:21-38 — the adversarial cloud whose file name starts with `idx-`:
    (A [npoint,3] float64, orig label, target label). Kept quirks: only the first `npoint` lines are used, a shorter
    file leaves the remaining rows equal to its LAST line (`A[row:] = line`), no file -> zeros and (idx, idx)."""
    A = np.zeros((npoint, 3), dtype=float)
    ori = tar = idx
    for file in os.listdir(path):
        parts = file.split('-')
        try:
            hit = int(parts[0]) == idx
        except ValueError:
            continue
        if hit:
            ori = int(parts[1])
            tar = int(parts[2].split('.')[0])
            with open(os.path.join(path, file)) as f:
                for row, line in enumerate(f):
                    if row >= npoint:
                        break
                    A[row:] = [float(v) for v in line.strip('\n').split(' ')[0:3]]
            break
    return A, ori, tar


class AdvData_Dataset(torch.utils.data.Dataset):
    """dataset/AdvData_dataset.py:41-82 over a directory of `idx-orig-target.txt` clouds (length = number of files;
    the reference hard-codes 1341)."""

    def __init__(self, data_path, npoint=4000):
        self.path = data_path
        self.npoint = npoint
        self.len = len([f for f in os.listdir(data_path) if f.split('-')[0].isdigit()])

    def __len__(self):
        return self.len

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return read_PC(idx, self.path, self.npoint)


def attack_cloud_file(attack_fn, in_path, out_dir, out_name, npoint=None, delimiter=None):
    """The whole file-to-file step of Eval_KNN.py:60-96 / Test_CW.py:69-111: read -> (sub-sample) -> normalise ->
    `attack_fn(pc [1,N,3] float32 tensor) -> adversarial cloud [1,N,3] (numpy or tensor)` -> de-normalise -> append
    the untouched extra columns -> save. Returns (path written, adversarial cloud in file coordinates [N,3])."""
    xyz, rest = load_cloud_txt(in_path, delimiter, npoint)
    norm, center, dist = normalize_cloud(xyz)
    adv = attack_fn(torch.from_numpy(norm.astype(np.float32)).unsqueeze(0))
    adv = adv.detach().cpu().numpy() if torch.is_tensor(adv) else np.asarray(adv)
    adv = denormalize_cloud(adv.reshape(-1, 3), center, dist)
    os.makedirs(out_dir, exist_ok=True)
    return save_cloud_txt(os.path.join(out_dir, out_name), adv, rest), adv
