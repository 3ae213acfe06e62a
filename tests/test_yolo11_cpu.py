"""YOLO11 host side (no GPU): the graph/plan of caesar_yolo_amd/yolo11_graph.py is internally consistent and agrees with
the independent oracle (oracle/yolo11_ref.py) on names and shapes; published parameter counts; the CYW2 container; the
`.pt` importer on a synthetic checkpoint of the ultralytics pickle structure."""
import sys
import types
import numpy as np
import pytest
import torch
import torch.nn as nn
from caesar_yolo_amd import yolo11_graph as G
from caesar_yolo_amd import weights as W
from caesar_yolo_amd import pt_import as PI

PUBLISHED_MPARAMS = {"n": 2.6, "s": 9.4, "m": 20.1, "l": 25.3, "x": 56.9}      # ultralytics model table, nc = 80


@pytest.mark.parametrize("scale", list("nsmlx"))
def test_parameter_count_matches_published(scale):
    g = G.build(scale, 80)
    params = sum(c.cout * (c.cin // c.groups) * c.k * c.k + (2 * c.cout if c.bn else c.cout) for c in g.convs)
    assert abs(params / 1e6 - PUBLISHED_MPARAMS[scale]) < 0.1, params


@pytest.mark.parametrize("scale,nc", [("n", 3), ("l", 5)])
def test_plan_is_consistent(scale, nc):
    g = G.build(scale, nc)
    used = [o["conv"] for o in g.ops if o["conv"] >= 0]
    assert sorted(used) == list(range(len(g.convs)))                       # every convolution runs exactly once
    for o in g.ops:
        lev0, c_in = g.tensors[o["in0"]]
        assert 0 <= o["in0_coff"] and o["in0_coff"] + (o["c0"] if o["kind"] != G.OPK_STEM else 3) <= c_in + (1 if o["kind"] == G.OPK_STEM else 0)
        if o["conv"] >= 0 and o["kind"] == G.OPK_CONV:
            cs = g.convs[o["conv"]]
            assert o["c0"] + o["c1"] == cs.cin, cs.name
        if o["out"] >= 0:
            lev1, c_out = g.tensors[o["out"]]
            width = g.convs[o["conv"]].cout if o["conv"] >= 0 else (o["c0"] if o["kind"] == G.OPK_POOL else o["p0"] * o["p2"])
            assert o["out_coff"] + width <= c_out
            if o["kind"] in (G.OPK_POOL, G.OPK_DWCONV, G.OPK_ATTN):
                assert lev1 == lev0
        else:
            assert o["pred_level"] in (0, 1, 2) and o["pred_coff"] in (0, 64)
        if o["res"] >= 0:
            assert o["res_coff"] + g.convs[o["conv"]].cout <= g.tensors[o["res"]][1]
        for t in (o["in0"], o["in1"], o["out"], o["res"]):
            assert t < len(g.tensors)
    attn = [o for o in g.ops if o["kind"] == G.OPK_ATTN]
    assert attn and all(o["p1"] == 32 and o["p2"] == 64 and o["p0"] == g.tensors[o["out"]][1] // 64 for o in attn)


def test_graph_names_and_shapes_agree_with_oracle():
    """The oracle consumes {name: (W, b)} keyed by the SAME ultralytics state_dict names: run it on zero-size-checked random
    weights of the graph's shapes; a wrong channel count anywhere makes F.conv2d raise."""
    sys.path.insert(0, "tests")
    from yolo11_common import seeded_folded
    from oracle import yolo11_ref as O
    for scale in ("n", "m"):
        g, wd = seeded_folded(scale, 4, probe_hw=64)
        net = O.Net11(wd, scale, 4)
        with torch.no_grad():
            raw = net.forward(torch.zeros(1, 3, 64, 96))
        assert tuple(raw.shape) == (1, 64 + 4, 8 * 12 + 4 * 6 + 2 * 3)
        assert set(wd) == {c.name for c in g.convs}


def test_cyw2_header_roundtrip(tmp_path):
    g = G.build("n", 2)
    rng = np.random.default_rng(0)
    folded = [(cs, rng.standard_normal((cs.cout, cs.cin // cs.groups, cs.k, cs.k)).astype(np.float32),
               rng.standard_normal(cs.cout).astype(np.float32)) for cs in g.convs]
    p = str(tmp_path / "a.cyw")
    W.write_cyw2(p, g, folded, {0: "x", 1: "y"})
    assert W.read_cyw_header(p) == ("n", {0: "x", 1: "y"}, 2, len(g.convs))
    assert open(p, "rb").read(4) == b"CYW2"
    with pytest.raises(ValueError):
        W.write_cyw2(p, g, folded, {0: "x"})


# ---- synthetic ultralytics-style YOLO11 checkpoint (throw-away modules, removed again before the import)
FAKE = ["ultralytics", "ultralytics.nn", "ultralytics.nn.tasks", "ultralytics.nn.modules", "ultralytics.nn.modules.conv",
        "ultralytics.nn.modules.block", "ultralytics.nn.modules.head"]


def _fake_classes():
    mods = {n: types.ModuleType(n) for n in FAKE}
    sys.modules.update(mods)

    def cls(mod, name):
        c = type(name, (nn.Module,), {"__module__": mod})
        setattr(mods[mod], name, c)
        return c
    K = {n: cls("ultralytics.nn.modules.conv", n) for n in ("Conv", "DWConv", "Concat")}
    K.update({n: cls("ultralytics.nn.modules.block", n) for n in ("C3k2", "C3k", "Bottleneck", "SPPF", "C2PSA", "PSABlock",
                                                                   "Attention", "DFL")})
    K["Detect"] = cls("ultralytics.nn.modules.head", "Detect")
    K["DetectionModel"] = cls("ultralytics.nn.tasks", "DetectionModel")
    return K


def _module_tree(K, g, ck):
    """nn.Module tree whose state_dict() has exactly the checkpoint's names: containers are created along every dotted path,
    typed like the ultralytics module at that position where the importer looks at types (the top-level sequence)."""
    top = {}
    for cs in g.convs:
        i = int(cs.name.split(".")[1])
        top.setdefault(i, []).append(cs)
    layers = []
    for i, kind in enumerate(PI._YOLO11_LAYERS):
        if kind == "Upsample":
            layers.append(nn.Upsample(scale_factor=2.0, mode="nearest"))
            continue
        m = K[kind]()
        for cs in top.get(i, []):
            parts = cs.name.split(".")[2:]
            node = m
            for p_ in parts:
                if not hasattr(node, p_):
                    node.add_module(p_, nn.Module())
                node = getattr(node, p_)
            co, ci, k = cs.cout, cs.cin // cs.groups, cs.k
            if cs.bn:
                node.conv = nn.Conv2d(ci * cs.groups, co, k, groups=cs.groups, bias=False)
                node.conv.weight.data = torch.from_numpy(ck[cs.name + ".conv.weight"])
                node.bn = nn.BatchNorm2d(co, eps=1e-3)
                node.bn.weight.data = torch.from_numpy(ck[cs.name + ".bn.weight"])
                node.bn.bias.data = torch.from_numpy(ck[cs.name + ".bn.bias"])
                node.bn.running_mean = torch.from_numpy(ck[cs.name + ".bn.running_mean"].copy())
                node.bn.running_var = torch.from_numpy(ck[cs.name + ".bn.running_var"].copy())
            else:                                           # plain nn.Conv2d(c, n, 1) at the end of a head branch
                node.weight = nn.Parameter(torch.from_numpy(ck[cs.name + ".weight"]))
                node.bias = nn.Parameter(torch.from_numpy(ck[cs.name + ".bias"]))
        layers.append(m)
    dm = K["DetectionModel"]()
    dm.model = nn.Sequential(*layers)
    dm.names = {i: "c%d" % i for i in range(g.nc)}
    return dm


def _random_checkpoint(g, seed=0):
    rng = np.random.default_rng(seed)
    ck = {}
    for cs in g.convs:
        w = rng.standard_normal((cs.cout, cs.cin // cs.groups, cs.k, cs.k)).astype(np.float32) * 0.1
        if cs.bn:
            ck[cs.name + ".conv.weight"] = w
            ck[cs.name + ".bn.weight"] = rng.uniform(0.5, 1.5, cs.cout).astype(np.float32)
            ck[cs.name + ".bn.bias"] = rng.standard_normal(cs.cout).astype(np.float32) * 0.1
            ck[cs.name + ".bn.running_mean"] = rng.standard_normal(cs.cout).astype(np.float32) * 0.1
            ck[cs.name + ".bn.running_var"] = rng.uniform(0.5, 1.5, cs.cout).astype(np.float32)
        else:
            ck[cs.name + ".weight"] = w
            ck[cs.name + ".bias"] = rng.standard_normal(cs.cout).astype(np.float32) * 0.1
    return ck


@pytest.mark.parametrize("scale", ["n", "m"])
def test_import_yolo11_checkpoint(tmp_path, scale):
    nc = 3
    g = G.build(scale, nc)
    ck = _random_checkpoint(g)
    K = _fake_classes()
    try:
        torch.save({"model": _module_tree(K, g, ck), "ema": None, "train_args": {}}, tmp_path / "y11.pt")
    finally:
        for n in FAKE:
            sys.modules.pop(n, None)
    r = PI.import_pt(str(tmp_path / "y11.pt"))
    assert "ultralytics" not in sys.modules
    assert (r["arch"], r["scale"], r["nc"]) == ("yolo11", scale, nc) and r["names"] == {0: "c0", 1: "c1", 2: "c2"}
    for k, v in ck.items():
        np.testing.assert_array_equal(r["sd"][k], v, err_msg=k)
    PI.convert_pt_to_cyw(str(tmp_path / "y11.pt"), str(tmp_path / "y11.cyw"))
    assert W.read_cyw_header(str(tmp_path / "y11.cyw")) == (scale, r["names"], nc, len(g.convs))
    with pytest.raises(PI.PtImportError):
        PI.import_ultralytics_pt(str(tmp_path / "y11.pt"))                 # the YOLOv8-only entry refuses it
