"""Ultralytics `.pt` importer on a SYNTHETIC checkpoint of the real pickle structure.

No trained checkpoint ships with the reference and ultralytics is not installed, so the test writes its own: throw-away
modules named `ultralytics.nn.tasks` / `ultralytics.nn.modules.*` are registered, a DetectionModel-shaped nn.Module tree
(Sequential of Conv/C2f/SPPF/Upsample/Concat/Detect with conv+bn leaves, fp16 like a trained checkpoint) is filled with
the seeded weights and saved with torch.save as {'model': ..., 'ema': ..., 'train_args': ...}.  The fake modules are
then REMOVED again, so the importer has to get through the pickle with none of its classes importable -- which is the
situation on a machine without ultralytics.  Parity with a real ultralytics file is unpinned."""
import sys
import types
import numpy as np
import pytest
import torch
import torch.nn as nn
from caesar_yolo_amd import pt_import as PI
from caesar_yolo_amd import weights as W
from caesar_yolo_amd import yolov8_spec as S

FAKE = ["ultralytics", "ultralytics.nn", "ultralytics.nn.tasks", "ultralytics.nn.modules", "ultralytics.nn.modules.conv",
        "ultralytics.nn.modules.block", "ultralytics.nn.modules.head"]


def _install_fakes():
    mods = {n: types.ModuleType(n) for n in FAKE}
    sys.modules.update(mods)

    def cls(mod, name, base=nn.Module):
        c = type(name, (base,), {"__module__": mod})
        setattr(mods[mod], name, c)
        return c
    K = {n: cls("ultralytics.nn.modules.conv", n) for n in ("Conv", "Concat")}
    K.update({n: cls("ultralytics.nn.modules.block", n) for n in ("C2f", "Bottleneck", "SPPF", "DFL", "C3k2")})
    K["Detect"] = cls("ultralytics.nn.modules.head", "Detect")
    K["DetectionModel"] = cls("ultralytics.nn.tasks", "DetectionModel")
    K["IterableSimpleNamespace"] = cls("ultralytics.nn.tasks", "IterableSimpleNamespace", object)
    return K


def _remove_fakes():
    for n in FAKE:
        sys.modules.pop(n, None)


def _conv(K, ck, name, fused=False):
    m = K["Conv"]()
    w = torch.from_numpy(ck[name + ".conv.weight"])
    co, ci, k, _ = w.shape
    m.conv = nn.Conv2d(ci, co, k, bias=fused)
    m.conv.weight.data = w
    if fused:
        m.conv.bias.data = torch.from_numpy(ck[name + ".conv.bias"])
    else:
        m.bn = nn.BatchNorm2d(co, eps=1e-3)
        for a, b in (("weight", "weight"), ("bias", "bias")):
            getattr(m.bn, a).data = torch.from_numpy(ck[name + ".bn." + b])
        m.bn.running_mean = torch.from_numpy(ck[name + ".bn.running_mean"].copy())
        m.bn.running_var = torch.from_numpy(ck[name + ".bn.running_var"].copy())
    m.act = nn.SiLU()
    return m


def _build(K, ck, scale, nc, names, fused=False, yolo11=False):
    c = S.channels(scale)

    def c2f(i, n):
        m = K["C3k2" if yolo11 and i == 2 else "C2f"]()
        m.cv1, m.cv2 = _conv(K, ck, "model.%d.cv1" % i, fused), _conv(K, ck, "model.%d.cv2" % i, fused)
        bl = []
        for j in range(n):
            b = K["Bottleneck"]()
            b.cv1, b.cv2 = _conv(K, ck, "model.%d.m.%d.cv1" % (i, j), fused), _conv(K, ck, "model.%d.m.%d.cv2" % (i, j), fused)
            bl.append(b)
        m.m = nn.ModuleList(bl)
        return m
    layers = []
    for i, kind in enumerate(PI._YOLOV8_LAYERS):
        if kind == "Conv":
            layers.append(_conv(K, ck, "model.%d" % i, fused))
        elif kind == "C2f":
            layers.append(c2f(i, c["n6"] if i in (4, 6) else c["n3"]))
        elif kind == "SPPF":
            m = K["SPPF"]()
            m.cv1, m.cv2 = _conv(K, ck, "model.9.cv1", fused), _conv(K, ck, "model.9.cv2", fused)
            m.m = nn.MaxPool2d(5, 1, 2)
            layers.append(m)
        elif kind == "Upsample":
            layers.append(nn.Upsample(scale_factor=2.0, mode="nearest"))
        elif kind == "Concat":
            layers.append(K["Concat"]())
        else:
            d = K["Detect"]()
            for br in ("cv2", "cv3"):
                seqs = []
                for lv in range(3):
                    last = nn.Conv2d(1, 1, 1)
                    last.weight.data = torch.from_numpy(ck["model.22.%s.%d.2.weight" % (br, lv)])
                    last.bias.data = torch.from_numpy(ck["model.22.%s.%d.2.bias" % (br, lv)])
                    seqs.append(nn.Sequential(_conv(K, ck, "model.22.%s.%d.0" % (br, lv), fused),
                                              _conv(K, ck, "model.22.%s.%d.1" % (br, lv), fused), last))
                setattr(d, br, nn.ModuleList(seqs))
            d.dfl = K["DFL"]()
            d.dfl.conv = nn.Conv2d(16, 1, 1, bias=False)
            d.dfl.conv.weight.data = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)
            d.nc, d.reg_max, d.stride = nc, 16, torch.tensor([8., 16., 32.])
            layers.append(d)
    dm = K["DetectionModel"]()
    dm.model = nn.Sequential(*layers)
    dm.names, dm.yaml, dm.stride = names, {"nc": nc, "scale": scale}, torch.tensor([8., 16., 32.])
    ns = K["IterableSimpleNamespace"]()
    ns.imgsz, ns.task = 640, "detect"
    dm.args = ns
    return dm


def _fused_ck(ck, scale, nc):
    out = dict(ck)
    for cs, w, b in W.fold(ck, scale, nc):
        if cs.bn:
            out[cs.name + ".conv.weight"], out[cs.name + ".conv.bias"] = w, b
    return out


@pytest.mark.parametrize("scale,half", [("n", True), ("s", False)])
def test_import_unfused_checkpoint(tmp_path, scale, half):
    nc, names = 5, dict(S.DEFAULT_NAMES)
    ck = W.seeded_checkpoint(scale, nc, seed=7)
    K = _install_fakes()
    try:
        dm = _build(K, ck, scale, nc, names)
        if half:
            dm = dm.half()
        torch.save({"epoch": -1, "model": None, "ema": dm, "train_args": {"imgsz": 640}, "date": "2025-01-01",
                    "version": "8.3.0"}, tmp_path / "w.pt")
    finally:
        _remove_fakes()
    assert "ultralytics" not in sys.modules
    sd, got_names, got_scale, got_nc = PI.import_ultralytics_pt(str(tmp_path / "w.pt"))
    assert "ultralytics" not in sys.modules                 # nothing from the pickle's module list was imported
    assert (got_scale, got_nc, got_names) == (scale, nc, names)
    for k, v in ck.items():
        ref = v.astype(np.float16).astype(np.float32) if half else v
        np.testing.assert_array_equal(sd[k], ref, err_msg=k)
    # -> CYW file, readable by the C-ABI loader's Python twin, identical to folding the source checkpoint
    PI.convert_pt_to_cyw(str(tmp_path / "w.pt"), str(tmp_path / "w.cyw"))
    sc, nm, wd, _ = W.read_cyw(str(tmp_path / "w.cyw"))
    src = {k: (v.astype(np.float16).astype(np.float32) if half else v) for k, v in ck.items()}
    for cs, w, b in W.fold(src, scale, nc):
        np.testing.assert_array_equal(wd[cs.name][0], w)
        np.testing.assert_array_equal(wd[cs.name][1], b)
    assert sc == scale and nm == names


def test_import_fused_checkpoint_and_model_key(tmp_path):
    scale, nc = "n", 3
    ck = W.seeded_checkpoint(scale, nc, seed=11)
    fck = _fused_ck(ck, scale, nc)
    K = _install_fakes()
    try:
        torch.save({"model": _build(K, fck, scale, nc, ["a", "b", "c"], fused=True), "ema": None}, tmp_path / "f.pt")
    finally:
        _remove_fakes()
    sd, names, got_scale, got_nc = PI.import_ultralytics_pt(str(tmp_path / "f.pt"))
    assert names == {0: "a", 1: "b", 2: "c"} and got_scale == scale and got_nc == nc
    for (cs, w, b), (_, w0, b0) in zip(W.fold(sd, scale, nc), W.fold(ck, scale, nc)):
        np.testing.assert_array_equal(w, w0)
        np.testing.assert_array_equal(b, b0)


def test_import_refuses_what_it_cannot_run(tmp_path):
    scale, nc = "n", 2
    ck = W.seeded_checkpoint(scale, nc, seed=3)
    K = _install_fakes()
    try:
        torch.save({"model": _build(K, ck, scale, nc, {0: "a", 1: "b"}, yolo11=True)}, tmp_path / "y11.pt")
    finally:
        _remove_fakes()
    with pytest.raises(PI.PtImportError, match="YOLO11"):
        PI.import_ultralytics_pt(str(tmp_path / "y11.pt"))
    (tmp_path / "junk.pt").write_bytes(b"not a zip")
    with pytest.raises(PI.PtImportError, match="not a torch zip"):
        PI.import_ultralytics_pt(str(tmp_path / "junk.pt"))


def test_pickle_globals_are_never_resolved(tmp_path):
    """A checkpoint whose pickle names a callable with side effects: the importer must not call (or import) it."""
    import pickle
    import zipfile
    marker = tmp_path / "executed"

    class Evil(object):
        def __reduce__(self):
            import os
            return (os.system, ("touch %s" % marker,))
    payload = pickle.dumps({"model": None, "ema": None, "x": Evil()}, protocol=2)
    with zipfile.ZipFile(tmp_path / "evil.pt", "w") as zf:
        zf.writestr("archive/data.pkl", payload)
        zf.writestr("archive/byteorder", "little")
    with pytest.raises(PI.PtImportError):
        PI.import_ultralytics_pt(str(tmp_path / "evil.pt"))
    assert not marker.exists()
    obj = PI.load_checkpoint_objects(str(tmp_path / "evil.pt"))
    assert isinstance(obj["x"], PI.Placeholder) and not marker.exists()


def test_yolo_shim_accepts_pt(tmp_path, monkeypatch):
    """YOLO('<file>.pt') converts through the importer into the cache directory (no GPU needed until .engine())."""
    from caesar_yolo_amd.model import YOLO
    monkeypatch.setenv("CAESAR_YOLO_CACHE", str(tmp_path / "cache"))
    scale, nc = "n", 4
    ck = W.seeded_checkpoint(scale, nc, seed=5)
    K = _install_fakes()
    try:
        torch.save({"model": _build(K, ck, scale, nc, {0: "w", 1: "x", 2: "y", 3: "z"})}, tmp_path / "best.pt")
    finally:
        _remove_fakes()
    m = YOLO(str(tmp_path / "best.pt"))
    assert m.names == {0: "w", 1: "x", 2: "y", 3: "z"} and m.scale == "n"
    assert m._wpath.endswith(".cyw") and str(tmp_path / "cache") in m._wpath
    m2 = YOLO(str(tmp_path / "best.pt"))                    # second construction reuses the cached conversion
    assert m2._wpath == m._wpath


def test_tensor_views_outside_their_storage_are_refused():
    """A crafted checkpoint must not make the importer read outside a storage blob (negative offset / stride, oversized view)."""
    import numpy as np
    from caesar_yolo_amd import pt_import as P
    st = (np.arange(16, dtype=np.float32), "FloatStorage")
    assert P._rebuild_tensor(st, 2, (3, 4), (4, 1)).shape == (3, 4)
    assert P._rebuild_tensor(st, 15, (), ()).shape == ()
    for off, size, stride in [(-1, (4,), (1,)), (0, (4,), (-1,)), (8, (3, 4), (4, 1)), (0, (5, 4), (4, 1)), (16, (), ()),
                              (0, (2, 2), (1,)), (0, (-1,), (1,))]:
        with pytest.raises(P.PtImportError):
            P._rebuild_tensor(st, off, size, stride)
