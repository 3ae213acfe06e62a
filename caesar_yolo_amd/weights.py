"""Weight containers for the HIP detector.

* `seeded_checkpoint(scale, nc, seed)`  -- deterministic random-init YOLOv8 weights in the ultralytics
  state_dict layout (conv weight + BatchNorm gamma/beta/mean/var).  No trained weights ship with the
  reference (README.md:192-206 are download links; test/run_inference.sh:4 names a file that is not in
  the repo), so every config here runs on these.
* `fold(checkpoint)`                   -- Conv2d(bias=False)+BatchNorm2d(eps=1e-3) -> conv weight + bias, fp32
  (what ultralytics `fuse()` does before inference).
* `write_cyw(path, ...)` / `read_cyw(path)` -- the flat "CYW1" file the C-ABI loads with `cy_load_weights`.

CYW2 (write_cyw2) carries the execution plan as well -- tensors, ops and convolutions of a yolo11_graph.Graph -- so the
runtime needs no built-in knowledge of the architecture:
  "CYW2" u32 version=2 | char arch[8] | char scale[4] | u32 nc nnames ntensors nops nconv | u32 feat_level[3]
  nnames x { u32 len | bytes (padded to 4) }      ntensors x { u32 level C }      nops x { i32 x 20 (yolo11_graph op fields) }
  nconv  x { u32 cout cin k s act groups | u32 name_len | name (padded to 4) | f32 W[cout][cin/groups][k][k] | f32 b[cout] }

CYW1 layout (little endian):
  "CYW1" u32 version=1 | char scale[4] | u32 nc | u32 nconv | u32 nnames
  nnames x { u32 len | bytes (padded to 4) }
  nconv  x { u32 cout cin k s act | u32 name_len | name (padded to 4) | f32 W[cout][cin][k][k] | f32 b[cout] }
version 2 (CYW1) / 3 (CYW2): every conv header carries a `flags` word in front of name_len; flags bit 0: `f32 scale[cout]` follows
the bias, with W[n] = fl32(w16[n] * scale[n]) and w16 the checkpoint's own fp16 filter (ultralytics checkpoints are stored in fp16
and Conv + BatchNorm are folded in fp32 when the model is loaded: scale = gamma / sqrt(var + eps)).  W and b are still the folded
fp32 tensors every context and the oracle use; the scale only lets the fp16x3 context recognise that a layer's filter is exact in
fp16 and run it in two passes instead of three (csrc/conv_igemm.hip: x3_passes).
"""
import json
import os
import struct
import numpy as np
from . import yolov8_spec as S

BN_EPS = 1e-3
# class-logit bias of the detect head for seeded weights: chosen once against a synthetic S16k-style tile so that
# conf 0.7 leaves O(10-100) candidates per 512x512 tile ("Seeded weights")
SEEDED_CLS_BIAS = -3.4
_SILU_M, _SILU_Q = 0.20645, 0.35568        # E[silu(z)], E[silu(z)^2] for z ~ N(0,1)


def _input_moments(scale, nc):
    """Mean-square of every conv's input, propagated analytically through the graph (independence
    assumption) so that the seeded BatchNorm statistics keep every pre-activation ~N(0,1):
    SiLU output (m,q)=(0.206,0.356); shortcut add: m+=0.206, q+=0.356+2*m_prev*0.206; concat = channel mean."""
    c = S.channels(scale)
    n3, n6 = c["n3"], c["n6"]
    q = {}
    act = (_SILU_M, _SILU_Q)

    def c2f(i, q_in, n, shortcut):
        q["model.%d.cv1" % i] = q_in
        m_cur, q_cur = act
        segs = [act[1], act[1]]
        for j in range(n):
            q["model.%d.m.%d.cv1" % (i, j)] = q_cur
            q["model.%d.m.%d.cv2" % (i, j)] = act[1]
            if shortcut:
                q_cur = q_cur + act[1] + 2 * m_cur * act[0]
                m_cur = m_cur + act[0]
            else:
                m_cur, q_cur = act
            segs.append(q_cur)
        q["model.%d.cv2" % i] = float(np.mean(segs))
    q["model.0"] = 0.25
    for i in (1, 3, 5, 7, 16, 19):
        q["model.%d" % i] = act[1]
    c2f(2, act[1], n3, True); c2f(4, act[1], n6, True); c2f(6, act[1], n6, True); c2f(8, act[1], n3, True)
    q["model.9.cv1"] = act[1]
    q["model.9.cv2"] = float(np.mean([act[1], 1.2, 1.9, 2.4]))       # a, mp5(a), mp9(a), mp13(a): rough
    for i in (12, 15, 18, 21):
        c2f(i, act[1], n3, False)
    for lvl in range(3):
        for br in ("cv2", "cv3"):
            for k in range(3):
                q["model.22.%s.%d.%d" % (br, lvl, k)] = act[1]
    return q


def _variance_table(scale, nc):
    """Per-conv variance of the raw conv output: the analytic estimate, refined by the table that
    tests/tools/calibrate_seeded.py measured once (seeded_calibration.json, plain data)."""
    t = _input_moments(scale, nc)
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seeded_calibration.json")
    if os.path.exists(p):
        with open(p) as fp:
            t.update(json.load(fp).get(scale, {}))
    return t


def _as_fp16_checkpoint(t):
    """what `torch.save(model.half())` leaves of a tensor: the nearest fp16 value (ultralytics stores its checkpoints that way,
    README.md:192-199 of the reference: 83.6 MB for 43.6 M parameters)"""
    return np.asarray(t, np.float32).astype(np.float16).astype(np.float32)


def seeded_checkpoint(scale="l", nc=5, seed=20260104, cls_bias=SEEDED_CLS_BIAS, var_table=None, half=True):
    """half=True (default): every tensor holds fp16 values, like an ultralytics checkpoint on disk."""
    ck = _seeded_checkpoint_f32(scale, nc, seed, cls_bias, var_table)
    return {k: _as_fp16_checkpoint(v) for k, v in ck.items()} if half else ck


def _seeded_checkpoint_f32(scale, nc, seed, cls_bias, var_table):
    rng = np.random.default_rng(seed)
    qin = var_table if var_table is not None else _variance_table(scale, nc)
    ck = {}
    for cs in S.conv_list(scale, nc):
        fan_in = cs.cin * cs.k * cs.k
        w = rng.standard_normal((cs.cout, cs.cin, cs.k, cs.k), dtype=np.float32) * np.float32(1.0 / np.sqrt(fan_in))
        v = qin[cs.name]                      # expected variance of the raw conv output (unit-variance weights/fan_in)
        if cs.bn:
            ck[cs.name + ".conv.weight"] = w
            ck[cs.name + ".bn.weight"] = rng.uniform(0.9, 1.1, cs.cout).astype(np.float32)
            ck[cs.name + ".bn.bias"] = (rng.standard_normal(cs.cout) * 0.1).astype(np.float32)
            ck[cs.name + ".bn.running_mean"] = (rng.standard_normal(cs.cout) * 0.1 * np.sqrt(v)).astype(np.float32)
            ck[cs.name + ".bn.running_var"] = (rng.uniform(0.9, 1.1, cs.cout) * v).astype(np.float32)
        else:
            boost = 1.0 if ".cv3." in cs.name else 2.0       # wider DFL logits -> varied box sizes
            ck[cs.name + ".weight"] = (w * np.float32(boost / np.sqrt(v))).astype(np.float32)
            if ".cv3." in cs.name:
                b = np.full(cs.cout, cls_bias, np.float32) + (rng.standard_normal(cs.cout) * 0.05).astype(np.float32)
            else:                              # DFL logits biased to low bins -> boxes of a few cells
                b = np.tile(1.0 - 0.6 * np.arange(S.REG_MAX, dtype=np.float32), 4)
            ck[cs.name + ".bias"] = b.astype(np.float32)
    return ck


def fold(ck, scale="l", nc=5, with_scale=False):
    """-> list of (ConvSpec, W fp32 [co,ci,k,k], b fp32 [co]) in canonical order; with_scale: 4-tuples with the per-channel
    factor the filter was multiplied by (gamma / sqrt(var + eps); ones for a plain Conv2d) as the last element."""
    out = []
    for cs in S.conv_list(scale, nc):
        if cs.bn:
            w = ck[cs.name + ".conv.weight"].astype(np.float32)
            g, beta = ck[cs.name + ".bn.weight"], ck[cs.name + ".bn.bias"]
            mu, var = ck[cs.name + ".bn.running_mean"], ck[cs.name + ".bn.running_var"]
            sc = (g / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32)
            wf = (w * sc[:, None, None, None]).astype(np.float32)
            bf = (beta - mu * sc).astype(np.float32)
        else:
            wf = ck[cs.name + ".weight"].astype(np.float32)
            bf = ck[cs.name + ".bias"].astype(np.float32)
            sc = np.ones(cs.cout, np.float32)
        assert wf.shape == (cs.cout, cs.cin, cs.k, cs.k), (cs.name, wf.shape)
        t = (cs, np.ascontiguousarray(wf), np.ascontiguousarray(bf))
        out.append(t + (np.ascontiguousarray(sc, np.float32),) if with_scale else t)
    return out


def _pad4(b):
    return b + b"\0" * ((-len(b)) % 4)


def write_cyw(path, folded, names, scale="l"):
    """folded: 3-tuples (cs, W, b) -> version 1; 4-tuples (cs, W, b, scale) (fold(with_scale=True)) -> version 2."""
    nc = len(names)
    v2 = any(len(t) > 3 for t in folded)
    with open(path, "wb") as fp:
        fp.write(b"CYW1" + struct.pack("<I", 2 if v2 else 1) + scale.encode().ljust(4, b"\0") +
                 struct.pack("<III", nc, len(folded), nc))
        for i in range(nc):
            nb = str(names[i]).encode()
            fp.write(struct.pack("<I", len(nb)) + _pad4(nb))
        for t in folded:
            cs, w, b = t[:3]
            nb = cs.name.encode()
            fp.write(struct.pack("<IIIII", cs.cout, cs.cin, cs.k, cs.s, int(cs.act)))
            if v2:
                fp.write(struct.pack("<I", 1 if len(t) > 3 else 0))
            fp.write(struct.pack("<I", len(nb)) + _pad4(nb))
            fp.write(w.astype("<f4").tobytes())
            fp.write(b.astype("<f4").tobytes())
            if len(t) > 3:
                fp.write(np.ascontiguousarray(t[3], "<f4").tobytes())


def read_cyw(path):
    """-> (scale, names dict, {conv name: (W, b)}, [(name,cout,cin,k,s,act)] in file order)."""
    with open(path, "rb") as fp:
        buf = fp.read()
    if buf[:4] != b"CYW1":
        raise ValueError("%s: not a CYW1 weight file" % path)
    ver, = struct.unpack_from("<I", buf, 4)
    if ver not in (1, 2):
        raise ValueError("unsupported CYW version %d" % ver)
    scale = buf[8:12].rstrip(b"\0").decode()
    nc, nconv, nnames = struct.unpack_from("<III", buf, 12)
    off = 24
    names = {}
    for i in range(nnames):
        n, = struct.unpack_from("<I", buf, off)
        off += 4
        names[i] = buf[off:off + n].decode()
        off += (n + 3) // 4 * 4
    w, order = {}, []
    for _ in range(nconv):
        co, ci, k, s, act = struct.unpack_from("<IIIII", buf, off)
        off += 20
        flags = 0
        if ver == 2:
            flags, = struct.unpack_from("<I", buf, off)
            off += 4
        n, = struct.unpack_from("<I", buf, off)
        off += 4
        name = buf[off:off + n].decode()
        off += (n + 3) // 4 * 4
        cnt = co * ci * k * k
        W = np.frombuffer(buf, "<f4", cnt, off).reshape(co, ci, k, k)
        off += 4 * cnt
        b = np.frombuffer(buf, "<f4", co, off)
        off += 4 * co
        if flags & 1:                      # per-channel scale annex (only the fp16x3 context's packer reads it)
            off += 4 * co
        w[name] = (W, b)
        order.append((name, co, ci, k, s, act))
    return scale, names, w, order


def make_seeded_file(path, scale="l", nc=5, seed=20260104, names=None):
    names = names or {i: S.DEFAULT_NAMES.get(i, "class%d" % i) for i in range(nc)}
    ck = seeded_checkpoint(scale, nc, seed)
    write_cyw(path, fold(ck, scale, nc, with_scale=True), names, scale)
    return path


def read_cyw_header(path):
    """-> (scale, names dict, nc, nconv) without touching the tensor payload (CYW1 and CYW2)."""
    with open(path, "rb") as fp:
        magic = fp.read(4)
        if magic == b"CYW1":
            head = magic + fp.read(20)
            scale = head[8:12].rstrip(b"\0").decode()
            nc, nconv, nnames = struct.unpack_from("<III", head, 12)
        elif magic == b"CYW2":
            head = fp.read(4 + 8 + 4 + 20 + 12)
            scale = head[12:16].rstrip(b"\0").decode()
            nc, nnames, _, _, nconv = struct.unpack_from("<IIIII", head, 16)
        else:
            raise ValueError("%s: not a CYW weight file" % path)
        names = {}
        for i in range(nnames):
            n, = struct.unpack("<I", fp.read(4))
            names[i] = fp.read((n + 3) // 4 * 4)[:n].decode()
    return scale, names, nc, nconv


OP_FIELDS = ("kind", "conv", "in0", "in0_coff", "c0", "up0", "in1", "in1_coff", "c1", "out", "out_coff", "res", "res_coff",
             "pred_level", "pred_coff", "p0", "p1", "p2", "p3", "_pad")


def fold_graph(ck, graph, with_scale=False):
    """Conv(+BatchNorm eps 1e-3) folding over a yolo11_graph.Graph's conv list (grouped / depth-wise convs included:
    the weight is [cout][cin/groups][k][k]) -> list of (ConvSpec, W, b) in state_dict order (with_scale: + the per-channel factor)."""
    out = []
    for cs in graph.convs:
        if cs.bn:
            w = np.asarray(ck[cs.name + ".conv.weight"], np.float32)
            g, beta = ck[cs.name + ".bn.weight"], ck[cs.name + ".bn.bias"]
            mu, var = ck[cs.name + ".bn.running_mean"], ck[cs.name + ".bn.running_var"]
            sc = (g / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32)
            wf = (w * sc[:, None, None, None]).astype(np.float32)
            bf = (beta - mu * sc).astype(np.float32)
        else:
            wf = np.asarray(ck[cs.name + ".weight"], np.float32)
            bf = np.asarray(ck[cs.name + ".bias"], np.float32)
            sc = np.ones(cs.cout, np.float32)
        if wf.shape != (cs.cout, cs.cin // cs.groups, cs.k, cs.k):
            raise ValueError("%s: weight shape %s, expected %s" % (cs.name, wf.shape, (cs.cout, cs.cin // cs.groups, cs.k, cs.k)))
        t = (cs, np.ascontiguousarray(wf), np.ascontiguousarray(bf))
        out.append(t + (np.ascontiguousarray(sc, np.float32),) if with_scale else t)
    return out


def write_cyw2(path, graph, folded, names):
    nc = len(names)
    if nc != graph.nc:
        raise ValueError("names (%d) do not match the graph's class count (%d)" % (nc, graph.nc))
    v3 = any(len(t) > 3 for t in folded)
    with open(path, "wb") as fp:
        fp.write(b"CYW2" + struct.pack("<I", 3 if v3 else 2) + graph.arch.encode().ljust(8, b"\0") + graph.scale.encode().ljust(4, b"\0"))
        fp.write(struct.pack("<IIIII", nc, nc, len(graph.tensors), len(graph.ops), len(folded)))
        fp.write(struct.pack("<III", *graph.feat_level))
        for i in range(nc):
            nb = str(names[i]).encode()
            fp.write(struct.pack("<I", len(nb)) + _pad4(nb))
        for lev, C in graph.tensors:
            fp.write(struct.pack("<II", lev, C))
        for o in graph.ops:
            fp.write(struct.pack("<20i", *[int(o.get(k, 0)) for k in OP_FIELDS]))
        for t in folded:
            cs, w, b = t[:3]
            nb = cs.name.encode()
            fp.write(struct.pack("<IIIIII", cs.cout, cs.cin, cs.k, cs.s, int(cs.act), cs.groups))
            if v3:
                fp.write(struct.pack("<I", 1 if len(t) > 3 else 0))
            fp.write(struct.pack("<I", len(nb)) + _pad4(nb))
            fp.write(np.ascontiguousarray(w, "<f4").tobytes())
            fp.write(np.ascontiguousarray(b, "<f4").tobytes())
            if len(t) > 3:
                fp.write(np.ascontiguousarray(t[3], "<f4").tobytes())


# ---------------------------------------------------------------------------------------------- seeded YOLO11
def seeded11_draw(graph, seed=11):
    """The raw random draw of every convolution of a yolo11_graph.Graph (weight ~ N(0, 1/fan_in), bias ~ 0.1 N(0,1)) in graph
    order -> {name: (W, b)}.  tests/tools/calibrate_seeded11.py measures one normalisation factor per conv on these."""
    rng = np.random.default_rng(seed)
    wd = {}
    for cs in graph.convs:
        fan = (cs.cin // cs.groups) * cs.k * cs.k
        w = (rng.standard_normal((cs.cout, cs.cin // cs.groups, cs.k, cs.k)) / np.sqrt(fan)).astype(np.float32)
        b = (rng.standard_normal(cs.cout) * 0.1).astype(np.float32)
        wd[cs.name] = (w, b)
    return wd


def seeded11_folded(scale="l", nc=5, seed=11, cls_bias=SEEDED_CLS_BIAS):
    """Deterministic random-init YOLO11 weights (folded, fp16-valued like a checkpoint on disk) with activations kept O(1) by
    the tabulated per-conv factors (seeded11_calibration.json, plain data measured once by tests/tools/calibrate_seeded11.py);
    class-logit bias as for the seeded YOLOv8 weights.  -> (graph, {name: (W, b)})."""
    from . import yolo11_graph as G
    g = G.build(scale, nc)
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seeded11_calibration.json")
    key = "%s:%d:%d" % (scale, nc, seed)
    table = json.load(open(p)).get(key) if os.path.exists(p) else None
    if table is None:
        raise ValueError("no calibration for seeded YOLO11 weights %s (python tests/tools/calibrate_seeded11.py %s %d %d)" % (key, scale, nc, seed))
    wd = {}
    for name, (w, b) in seeded11_draw(g, seed).items():
        w = _as_fp16_checkpoint(w / np.float32(table[name]))
        b = _as_fp16_checkpoint(b)
        if cls_bias is not None and ".cv3." in name and name.endswith(".2"):
            b = _as_fp16_checkpoint(np.full_like(b, cls_bias))
        wd[name] = (w, b)
    return g, wd


def make_seeded11_file(path, scale="l", nc=5, seed=11, names=None):
    names = names or {i: S.DEFAULT_NAMES.get(i, "class%d" % i) for i in range(nc)}
    g, wd = seeded11_folded(scale, nc, seed)
    write_cyw2(path, g, [(cs, wd[cs.name][0], wd[cs.name][1], np.ones(cs.cout, np.float32)) for cs in g.convs], names)
    return path
