"""CPU: the hipGraph victim wrapper (3dpointcloudattack_amd/graphed.py) as plain Python — which models it wraps, that it
stays out of state_dicts, copies and pickles, and that attribute access falls through to the victim."""
import copy
import importlib
import io
import pickle

import torch

graphed = importlib.import_module("3dpointcloudattack_amd.graphed")
pointnet = importlib.import_module("3dpointcloudattack_amd.model.pointnet")


def test_wrap_only_deterministic_victims_and_cache_on_the_model():
    m = pointnet.PointNetCls(k=5, feature_transform=False).eval()
    g = graphed.wrap(m)
    assert isinstance(g, graphed.GraphedVictim) and graphed.wrap(m) is g and graphed.wrap(g) is g
    assert graphed.wrap(m, enable=False) is m
    assert graphed.wrap(torch.nn.Linear(3, 3)).__class__ is torch.nn.Linear        # no deterministic_forward declared
    assert "_pc3d_graphed" not in dict(m.named_modules()) and not any("graphed" in k for k in m.state_dict())
    assert hasattr(g, "fused_loss_and_grad") and g.deterministic_forward is True    # falls through to the victim
    assert set(g.state_dict()) == set(m.state_dict())
    g.load_state_dict(m.state_dict())


def test_copies_and_pickles_drop_the_captures():
    m = pointnet.PointNetCls(k=5, feature_transform=False).eval()
    g = graphed.wrap(m)
    g._slots["fake"] = [object()]
    g.stats["captures"] = 3
    m2 = copy.deepcopy(m)
    g2 = m2.__dict__["_pc3d_graphed"]
    assert g2 is not g and g2.model is m2 and g2._slots == {} and graphed.wrap(m2) is g2
    g._slots.clear()                                    # the fake entry is not picklable, as real captures are not
    g._slots["k"] = ["capture"]
    buf = io.BytesIO()
    pickle.dump(g, buf)
    g3 = pickle.loads(buf.getvalue())
    assert g3._slots == {} and g3.stats["captures"] == 0
    for a, b in zip(g3.model.state_dict().values(), m.state_dict().values()):
        assert torch.equal(a, b)
