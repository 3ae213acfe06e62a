"""Read an ultralytics YOLOv8 detection checkpoint (`*.pt`) WITHOUT ultralytics and without executing anything from it.

The reference hands the path straight to `ultralytics.YOLO(weights)` (scripts/run.py:347), which unpickles a full
`DetectionModel` object graph.  Here the file is treated as data:

  * a torch zip checkpoint is `<root>/data.pkl` + one raw little-endian blob per storage (`<root>/data/<key>`);
  * `data.pkl` is walked by a RESTRICTED unpickler: tensors are rebuilt as numpy arrays from the blobs, plain
    containers (dict / OrderedDict / list / tuple / set) are kept, and EVERY other global the pickle names - the
    ultralytics / torch.nn classes, functions, anything else - is replaced by an inert placeholder that only records
    its constructor arguments and state.  No class or function named by the file is ever imported or called;
  * the placeholder graph is then flattened exactly like `nn.Module.state_dict()` (`_modules` / `_parameters` /
    `_buffers`), which yields the `model.N....conv.weight`, `...bn.running_mean` names that `weights.fold` consumes.

Two detect architectures are accepted, recognised by their layer-class sequence: YOLOv8 (Conv / C2f / SPPF / Upsample /
Concat / Detect) and YOLO11 (C3k2 / C2PSA / depth-wise class branch; yolo11_graph.py).  Anything else is refused with a
clear message.
No trained checkpoint ships with the reference (README.md:192-206 are links), so this importer is tested against a
synthetic checkpoint of the same pickle structure written by the test-suite (tests/test_pt_import.py): parity with a
real ultralytics file is unpinned.
"""
import io
import math
import pickle
import zipfile
from collections import OrderedDict
import numpy as np
from . import yolov8_spec as S

_STORAGE_DTYPES = {
    "FloatStorage": np.dtype("<f4"), "HalfStorage": np.dtype("<f2"), "DoubleStorage": np.dtype("<f8"),
    "LongStorage": np.dtype("<i8"), "IntStorage": np.dtype("<i4"), "ShortStorage": np.dtype("<i2"),
    "CharStorage": np.dtype("i1"), "ByteStorage": np.dtype("u1"), "BoolStorage": np.dtype("?"),
    "BFloat16Storage": np.dtype("<u2"),          # widened to fp32 on rebuild
}
_YOLOV8_LAYERS = ["Conv", "Conv", "C2f", "Conv", "C2f", "Conv", "C2f", "Conv", "C2f", "SPPF", "Upsample", "Concat", "C2f",
                  "Upsample", "Concat", "C2f", "Conv", "Concat", "C2f", "Conv", "Concat", "C2f", "Detect"]
_YOLO11_LAYERS = ["Conv", "Conv", "C3k2", "Conv", "C3k2", "Conv", "C3k2", "Conv", "C3k2", "SPPF", "C2PSA", "Upsample", "Concat",
                  "C3k2", "Upsample", "Concat", "C3k2", "Conv", "Concat", "C3k2", "Conv", "Concat", "C3k2", "Detect"]
_UNSUPPORTED_BLOCKS = ("C2fCIB", "SCDown", "PSA", "v10Detect", "RepC3", "RTDETRDecoder", "Segment", "Pose", "OBB", "A2C2f")


class PtImportError(Exception):
    pass


class _StorageTag(object):
    """Stands for a `torch.<X>Storage` class named by the pickle: only its dtype is used."""
    def __init__(self, name):
        self.name, self.dtype = name, _STORAGE_DTYPES.get(name)


class Placeholder(object):
    """Inert stand-in for any object whose class the pickle names.  Records what the pickle gave it, runs nothing."""
    _cy_qualname = "?"

    def __init__(self, *args, **kwargs):
        self.__dict__["_cy_args"] = args

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):      # (dict, slots)
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])
        else:
            self.__dict__["_cy_state"] = state

    def __call__(self, *args, **kwargs):          # a REDUCE on a placeholder "function": yield another placeholder
        return Placeholder()

    # containers pickled through append / setitem (list / dict subclasses)
    def append(self, x):
        self.__dict__.setdefault("_cy_items", []).append(x)

    def extend(self, xs):
        self.__dict__.setdefault("_cy_items", []).extend(xs)

    def __setitem__(self, k, v):
        self.__dict__.setdefault("_cy_map", OrderedDict())[k] = v


def _placeholder_class(module, name):
    return type(str(name), (Placeholder,), {"_cy_qualname": "%s.%s" % (module, name)})


def _rebuild_tensor(storage, offset, size, stride, *unused):
    arr, dtype_name = storage
    offset = int(offset)
    size, stride = tuple(int(s) for s in size), tuple(int(s) for s in stride)
    # a checkpoint is untrusted data: every element the view can reach must lie inside the storage blob (as_strided checks
    # nothing: a negative offset or stride, or an oversized extent, would read process memory into the "weights")
    if offset < 0 or len(size) != len(stride) or any(n < 0 for n in size) or any(st < 0 for st in stride):
        raise PtImportError("malformed tensor view (negative offset, size or stride)")
    if len(size) == 0:
        if offset + 1 > arr.size:
            raise PtImportError("tensor view exceeds its storage")
        out = arr[offset:offset + 1].reshape(())
    else:
        need = offset + sum((n - 1) * st for n, st in zip(size, stride)) + 1 if all(n > 0 for n in size) else 0
        if need > arr.size:
            raise PtImportError("tensor view exceeds its storage")
        out = np.lib.stride_tricks.as_strided(arr[offset:], shape=size, strides=tuple(st * arr.itemsize for st in stride))
    out = np.array(out)                                   # own, contiguous copy
    if dtype_name == "BFloat16Storage":
        out = (out.astype(np.uint32) << 16).view(np.float32)
    return out


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None, *unused):
    return data


def _rebuild_parameter_with_state(data, requires_grad=False, backward_hooks=None, state=None, *unused):
    return data


class _Unpickler(pickle.Unpickler):
    def __init__(self, fp, read_blob):
        pickle.Unpickler.__init__(self, fp, encoding="utf-8")
        self._read_blob, self._blobs = read_blob, {}

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch._utils":
            if name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
                return _rebuild_tensor
            if name == "_rebuild_parameter":
                return _rebuild_parameter
            if name == "_rebuild_parameter_with_state":
                return _rebuild_parameter_with_state
        if module in ("torch", "torch.storage") and name.endswith("Storage"):
            return _StorageTag(name)
        if module == "torch" and name == "Size":
            return tuple
        if module in ("builtins", "__builtin__") and name in ("set", "frozenset", "list", "dict", "tuple", "int", "float",
                                                              "bool", "str", "bytes", "complex", "slice", "range", "object"):
            return {"set": set, "frozenset": frozenset, "list": list, "dict": dict, "tuple": tuple, "int": int,
                    "float": float, "bool": bool, "str": str, "bytes": bytes, "complex": complex, "slice": slice,
                    "range": range, "object": Placeholder}[name]
        return _placeholder_class(module, name)       # never imported, never executed

    def persistent_load(self, pid):
        # ('storage', storage_type, key, location, numel)
        if not isinstance(pid, tuple) or len(pid) < 5 or pid[0] != "storage":
            raise PtImportError("unsupported persistent id in checkpoint")
        tag, key = pid[1], str(pid[2])
        if isinstance(tag, _StorageTag):
            name, dt = tag.name, tag.dtype
        else:                                              # torch >= 2.x may pickle a dtype placeholder instead
            name = getattr(tag, "_cy_qualname", "?").split(".")[-1]
            dt = {"float32": np.dtype("<f4"), "float16": np.dtype("<f2"), "float64": np.dtype("<f8"), "int64": np.dtype("<i8"),
                  "int32": np.dtype("<i4"), "uint8": np.dtype("u1"), "bool": np.dtype("?")}.get(name)
        if dt is None:
            raise PtImportError("unsupported storage type %r" % name)
        if key not in self._blobs:
            self._blobs[key] = np.frombuffer(self._read_blob(key), dtype=dt)
        return (self._blobs[key], name)


def load_checkpoint_objects(path):
    """-> the unpickled top-level object of a torch zip checkpoint, with placeholders for every foreign class."""
    try:
        zf = zipfile.ZipFile(path)
    except zipfile.BadZipFile:
        raise PtImportError("%s is not a torch zip checkpoint (legacy pre-1.6 .pt files are not supported)" % path)
    with zf:
        pk = [n for n in zf.namelist() if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pk) != 1:
            raise PtImportError("no data.pkl in %s" % path)
        root = pk[0][:-len("data.pkl")]
        if root + "byteorder" in zf.namelist() and zf.read(root + "byteorder").strip() != b"little":
            raise PtImportError("big-endian checkpoints are not supported")
        up = _Unpickler(io.BytesIO(zf.read(pk[0])), lambda key: zf.read("%sdata/%s" % (root, key)))
        return up.load()


def _children(obj):
    m = getattr(obj, "_modules", None)
    return m if isinstance(m, dict) else {}


def state_dict_of(module, prefix=""):
    """nn.Module.state_dict() over the placeholder graph: parameters, then buffers, then children, depth first."""
    out = OrderedDict()
    for group in ("_parameters", "_buffers"):
        g = getattr(module, group, None)
        if isinstance(g, dict):
            for k, v in g.items():
                if isinstance(v, np.ndarray):
                    out[prefix + k] = v
    for name, child in _children(module).items():
        if child is not None:
            out.update(state_dict_of(child, prefix + name + "."))
    return out


def _class_name(obj):
    return getattr(type(obj), "_cy_qualname", type(obj).__name__).split(".")[-1]


def _detect_scale(sd):
    w0 = sd.get("model.0.conv.weight")
    if w0 is None or w0.ndim != 4 or w0.shape[1] != 3:
        raise PtImportError("model.0.conv.weight missing or not a 3-channel stem: not a YOLOv8 detection checkpoint")
    for scale in S.SCALES:
        c = S.channels(scale)
        n_m2 = len({k.split(".")[3] for k in sd if k.startswith("model.2.m.")})
        n_m4 = len({k.split(".")[3] for k in sd if k.startswith("model.4.m.")})
        if c["c1"] == w0.shape[0] and c["n3"] == n_m2 and c["n6"] == n_m4:
            return scale
    raise PtImportError("stem width %d / depths do not match any YOLOv8 scale (n,s,m,l,x)" % w0.shape[0])


def _detect_scale11(sd):
    from . import yolo11_graph as G
    w0 = sd.get("model.0.conv.weight")
    if w0 is None or w0.ndim != 4 or w0.shape[1] != 3:
        raise PtImportError("model.0.conv.weight missing or not a 3-channel stem: not a YOLO11 detection checkpoint")
    n_m2 = len({k.split(".")[3] for k in sd if k.startswith("model.2.m.")})
    c3k_at_2 = any(k.startswith("model.2.m.0.cv3.") for k in sd)
    for scale, (d, w, mc) in G.SCALES.items():
        width = int(math.ceil(min(64, mc) * w / 8) * 8)
        if width == w0.shape[0] and max(round(2 * d), 1) == n_m2 and (scale in "mlx") == c3k_at_2:
            return scale
    raise PtImportError("stem width %d / depth %d do not match any YOLO11 scale (n,s,m,l,x)" % (w0.shape[0], n_m2))


def _fused_to_identity_bn(sd, convs):
    """Fused checkpoints (Conv2d with bias, no bn) -> the layout fold() expects, with an identity BatchNorm."""
    from .weights import BN_EPS
    for cs in convs:
        if not cs.bn:
            if cs.name + ".weight" not in sd or cs.name + ".bias" not in sd:
                raise PtImportError("missing %s.weight/.bias" % cs.name)
            continue
        w = sd.get(cs.name + ".conv.weight")
        if w is None:
            raise PtImportError("missing %s.conv.weight" % cs.name)
        groups = getattr(cs, "groups", 1)
        if w.shape != (cs.cout, cs.cin // groups, cs.k, cs.k):
            raise PtImportError("%s.conv.weight has shape %s, expected %s" % (cs.name, w.shape, (cs.cout, cs.cin // groups, cs.k, cs.k)))
        if cs.name + ".bn.weight" not in sd:
            b = sd.get(cs.name + ".conv.bias")
            if b is None:
                raise PtImportError("%s has neither BatchNorm statistics nor a fused bias" % cs.name)
            sd[cs.name + ".bn.weight"] = np.full(cs.cout, np.sqrt(np.float32(1.0) + np.float32(BN_EPS)), np.float32)
            sd[cs.name + ".bn.bias"] = b.astype(np.float32)
            sd[cs.name + ".bn.running_mean"] = np.zeros(cs.cout, np.float32)
            sd[cs.name + ".bn.running_var"] = np.ones(cs.cout, np.float32)


def import_pt(path):
    """-> dict(sd={name: fp32 ndarray} in ultralytics naming, names={int: str}, scale, nc, arch='yolov8'|'yolo11').

    Accepts both un-fused checkpoints (Conv2d without bias + BatchNorm2d: what ultralytics saves after training) and
    fused ones (Conv2d with bias, no bn)."""
    ck = load_checkpoint_objects(path)
    model = None
    if isinstance(ck, dict):
        model = ck.get("ema") if ck.get("ema") is not None else ck.get("model")
    names, arch = None, None
    if model is not None and not isinstance(model, dict):
        seq = _children(model).get("model")
        if seq is None:
            raise PtImportError("checkpoint 'model' has no .model Sequential: not an ultralytics detection model")
        layers = [_class_name(m) for m in _children(seq).values()]
        bad = sorted(set(l for l in layers if l in _UNSUPPORTED_BLOCKS))
        if bad:
            raise PtImportError("blocks %s are not supported by the HIP detect path (YOLOv8 and YOLO11 detection only)" % bad)
        if layers == _YOLOV8_LAYERS:
            arch = "yolov8"
        elif layers == _YOLO11_LAYERS:
            arch = "yolo11"
        else:
            raise PtImportError("layer sequence %s is neither the YOLOv8 nor the YOLO11 detect graph" % layers)
        sd = state_dict_of(model)
        names = getattr(model, "names", None)
    else:
        sd = OrderedDict((k, v) for k, v in (model if isinstance(model, dict) else ck).items() if isinstance(v, np.ndarray))
        if not sd:
            raise PtImportError("no tensors found in %s" % path)
        arch = "yolo11" if any(k.startswith("model.23.") for k in sd) else "yolov8"
    sd = OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items() if v.dtype.kind == "f")
    head = "model.23" if arch == "yolo11" else "model.22"
    cls_last = sd.get(head + ".cv3.0.2.weight")
    if cls_last is None:
        raise PtImportError("detect head (%s.cv3) missing" % head)
    nc = int(cls_last.shape[0])
    if isinstance(names, (list, tuple)):
        names = {i: n for i, n in enumerate(names)}
    if not isinstance(names, dict) or len(names) != nc:
        names = {i: "class%d" % i for i in range(nc)}
    names = {int(k): str(v) for k, v in names.items()}
    if arch == "yolo11":
        from . import yolo11_graph as G
        scale = _detect_scale11(sd)
        _fused_to_identity_bn(sd, G.build(scale, nc).convs)
    else:
        scale = _detect_scale(sd)
        _fused_to_identity_bn(sd, S.conv_list(scale, nc))
    return dict(sd=sd, names=names, scale=scale, nc=nc, arch=arch)


def import_ultralytics_pt(path):
    """YOLOv8 checkpoints: -> (state_dict, names, scale, nc)  (kept for callers that only handle YOLOv8)."""
    r = import_pt(path)
    if r["arch"] != "yolov8":
        raise PtImportError("%s is a %s checkpoint; use import_pt / convert_pt_to_cyw" % (path, r["arch"]))
    return r["sd"], r["names"], r["scale"], r["nc"]


def convert_pt_to_cyw(pt_path, cyw_path):
    """ultralytics `.pt` -> CYW file for cy_load_weights (CYW1 for YOLOv8, CYW2 with the plan inside for YOLO11);
    returns (scale, names)."""
    from . import weights as W
    r = import_pt(pt_path)
    if r["arch"] == "yolo11":
        from . import yolo11_graph as G
        g = G.build(r["scale"], r["nc"])
        W.write_cyw2(cyw_path, g, W.fold_graph(r["sd"], g, with_scale=True), r["names"])
    else:
        W.write_cyw(cyw_path, W.fold(r["sd"], r["scale"], r["nc"], with_scale=True), r["names"], r["scale"])
    return r["scale"], r["names"]
