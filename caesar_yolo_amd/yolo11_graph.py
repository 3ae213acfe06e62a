"""YOLO11 detection graph: convolution list in ultralytics state_dict naming + the NHWC execution
plan the HIP runtime runs, for a (scale, nc) pair.

The reference only ever sees this network as `YOLO("yolo11*.pt")` (README.md:200-207 lists yolo11n/l weights); its
definition is the third-party ultralytics package (unpinned, absent here), so this follows the public yolo11.yaml and
module definitions (Conv, C3k2 / C3k / Bottleneck, SPPF, C2PSA / PSABlock / Attention, Detect with depth-wise class
branch).  Nothing in the reference's tests pins it: parity is against this repo's own torch restatement
(oracle/yolo11_ref.py, written independently of this file) and is UNPINNED with respect to ultralytics.

Unlike YOLOv8 (whose plan the C++ runtime derives itself from scale/nc), the plan travels inside the weight file
("CYW2", weights.write_cyw2): tensors, ops and convolutions below are serialised as they are.

Plan conventions (same as csrc/cy_plan.h): a tensor is (level, C) = [B, H>>level, W>>level, C] NHWC; ops read and write
CHANNEL SLICES of tensors, so chunk/cat are free; `up0` reads segment 0 through a nearest x2 upsample.
"""
import math

SCALES = {  # depth, width, max_channels  (yolo11.yaml)
    "n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512),
    "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512),
}
OPK_STEM, OPK_CONV, OPK_POOL, OPK_DWCONV, OPK_ATTN = 0, 1, 2, 3, 4


class ConvSpec(object):
    __slots__ = ("name", "cin", "cout", "k", "s", "act", "bn", "groups")

    def __init__(self, name, cin, cout, k, s, act=True, bn=True, groups=1):
        self.name, self.cin, self.cout, self.k, self.s, self.act, self.bn, self.groups = name, cin, cout, k, s, act, bn, groups


class Graph(object):
    def __init__(self, scale, nc):
        self.arch, self.scale, self.nc = "yolo11", scale, nc
        self.convs, self.idx = [], {}
        self.tensors, self.ops = [(0, 4)], []          # tensor 0 = network input [B,H,W,4]
        self.feat_level = [3, 4, 5]

    # ---- declarations
    def add_conv(self, name, cin, cout, k, s, act=True, bn=True, groups=1):
        self.idx[name] = len(self.convs)
        self.convs.append(ConvSpec(name, cin, cout, k, s, act, bn, groups))

    def T(self, level, C):
        self.tensors.append((level, C))
        return len(self.tensors) - 1

    # ---- ops
    def op(self, kind, name=None, **kw):
        o = dict(kind=kind, conv=self.idx[name] if name else -1, in0=-1, in0_coff=0, c0=0, up0=0, in1=-1, in1_coff=0, c1=0,
                 out=-1, out_coff=0, res=-1, res_coff=0, pred_level=-1, pred_coff=0, p0=0, p1=0, p2=0, p3=0)
        o.update(kw)
        self.ops.append(o)
        return o

    def conv(self, name, src, src_coff, dst, dst_coff, res=-1, res_coff=0, **kw):
        cs = self.convs[self.idx[name]]
        return self.op(OPK_CONV, name, in0=src, in0_coff=src_coff, c0=kw.pop("c0", cs.cin), out=dst, out_coff=dst_coff,
                       res=res, res_coff=res_coff, **kw)


def _make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


def build(scale="n", nc=80):
    """-> Graph for yolo11<scale> with nc classes."""
    d, w, mc = SCALES[scale]
    ch = lambda c: _make_divisible(min(c, mc) * w, 8)
    rep = lambda n: max(round(n * d), 1) if n > 1 else n
    c3k_all = scale in "mlx"                              # parse_model: C3k2 gets c3k=True for the m/l/x scales
    g = Graph(scale, nc)
    c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    n2 = rep(2)

    # ------------------------------------------------------------------ declarations (state_dict order)
    def decl_c3k2(i, c1, c2, n, c3k, e):
        c = int(c2 * e)
        b = "model.%d" % i
        g.add_conv(b + ".cv1", c1, 2 * c, 1, 1)
        g.add_conv(b + ".cv2", (2 + n) * c, c2, 1, 1)
        for j in range(n):
            m = "%s.m.%d" % (b, j)
            if c3k or c3k_all:
                c_ = int(c * 0.5)
                g.add_conv(m + ".cv1", c, c_, 1, 1)
                g.add_conv(m + ".cv2", c, c_, 1, 1)
                g.add_conv(m + ".cv3", 2 * c_, c, 1, 1)
                for q in range(2):
                    g.add_conv("%s.m.%d.cv1" % (m, q), c_, c_, 3, 1)
                    g.add_conv("%s.m.%d.cv2" % (m, q), c_, c_, 3, 1)
            else:
                g.add_conv(m + ".cv1", c, c // 2, 3, 1)
                g.add_conv(m + ".cv2", c // 2, c, 3, 1)
        return c

    g.add_conv("model.0", 3, c64, 3, 2)
    g.add_conv("model.1", c64, c128, 3, 2)
    cs = {}
    cs[2] = decl_c3k2(2, c128, c256, n2, False, 0.25)
    g.add_conv("model.3", c256, c256, 3, 2)
    cs[4] = decl_c3k2(4, c256, c512, n2, False, 0.25)
    g.add_conv("model.5", c512, c512, 3, 2)
    cs[6] = decl_c3k2(6, c512, c512, n2, True, 0.5)
    g.add_conv("model.7", c512, c1024, 3, 2)
    cs[8] = decl_c3k2(8, c1024, c1024, n2, True, 0.5)
    g.add_conv("model.9.cv1", c1024, c1024 // 2, 1, 1)
    g.add_conv("model.9.cv2", c1024 * 2, c1024, 1, 1)
    pc = int(c1024 * 0.5)                                 # C2PSA hidden width
    heads, hd = pc // 64, 64
    if heads < 1 or pc % 64:
        raise ValueError("C2PSA width %d is not a multiple of 64 (unsupported scale)" % pc)
    kd = hd // 2
    g.add_conv("model.10.cv1", c1024, 2 * pc, 1, 1)
    g.add_conv("model.10.cv2", 2 * pc, c1024, 1, 1)
    for j in range(n2):
        m = "model.10.m.%d" % j
        g.add_conv(m + ".attn.qkv", pc, pc + 2 * heads * kd, 1, 1, act=False)
        g.add_conv(m + ".attn.proj", pc, pc, 1, 1, act=False)
        g.add_conv(m + ".attn.pe", pc, pc, 3, 1, act=False, groups=pc)
        g.add_conv(m + ".ffn.0", pc, 2 * pc, 1, 1)
        g.add_conv(m + ".ffn.1", 2 * pc, pc, 1, 1, act=False)
    cs[13] = decl_c3k2(13, c1024 + c512, c512, n2, False, 0.5)
    cs[16] = decl_c3k2(16, c512 + c512, c256, n2, False, 0.5)
    g.add_conv("model.17", c256, c256, 3, 2)
    cs[19] = decl_c3k2(19, c256 + c512, c512, n2, False, 0.5)
    g.add_conv("model.20", c512, c512, 3, 2)
    cs[22] = decl_c3k2(22, c512 + c1024, c1024, n2, True, 0.5)
    chs = (c256, c512, c1024)
    cb = max(16, chs[0] // 4, 64)
    cc = max(chs[0], min(nc, 100))
    for l in range(3):
        b = "model.23.cv2.%d" % l
        g.add_conv(b + ".0", chs[l], cb, 3, 1)
        g.add_conv(b + ".1", cb, cb, 3, 1)
        g.add_conv(b + ".2", cb, 64, 1, 1, act=False, bn=False)
    for l in range(3):
        b = "model.23.cv3.%d" % l
        g.add_conv(b + ".0.0", chs[l], chs[l], 3, 1, groups=chs[l])
        g.add_conv(b + ".0.1", chs[l], cc, 1, 1)
        g.add_conv(b + ".1.0", cc, cc, 3, 1, groups=cc)
        g.add_conv(b + ".1.1", cc, cc, 1, 1)
        g.add_conv(b + ".2", cc, nc, 1, 1, act=False, bn=False)

    # ------------------------------------------------------------------ execution plan
    def dwconv(name, src, src_coff, dst, dst_coff, res=-1, res_coff=0, blk=0, gstride=0, goff=0):
        c = g.convs[g.idx[name]].cout
        return g.op(OPK_DWCONV, name, in0=src, in0_coff=src_coff, c0=c, out=dst, out_coff=dst_coff, res=res, res_coff=res_coff,
                    p0=blk, p1=gstride, p2=goff)

    def c3k2(i, first, level, c2, n, c3k, dst, dst_coff):
        """C2f data flow: cv1 -> [y0|y1] at the head of a (2+n)c buffer, block j reads slice 1+j and appends slice 2+j."""
        c = cs[i]
        b = "model.%d" % i
        buf = g.T(level, (2 + n) * c)
        first.update(out=buf, out_coff=0)
        g.ops.append(first)
        for j in range(n):
            m = "%s.m.%d" % (b, j)
            src_off, dst_off = (1 + j) * c, (2 + j) * c
            if c3k or c3k_all:
                c_ = int(c * 0.5)
                cat = g.T(level, 2 * c_)                  # [m(cv1(x)) | cv2(x)]
                ta, tb = g.T(level, c_), g.T(level, c_)
                g.conv(m + ".cv1", buf, src_off, ta, 0)
                g.conv(m + ".cv2", buf, src_off, cat, c_)
                # two bottlenecks c_ -> c_ (3x3, 3x3, + shortcut); the second writes into the cat buffer
                g.conv(m + ".m.0.cv1", ta, 0, tb, 0)
                t1 = g.T(level, c_)
                g.conv(m + ".m.0.cv2", tb, 0, t1, 0, res=ta, res_coff=0)
                g.conv(m + ".m.1.cv1", t1, 0, tb, 0)
                g.conv(m + ".m.1.cv2", tb, 0, cat, 0, res=t1, res_coff=0)
                g.conv(m + ".cv3", cat, 0, buf, dst_off)
            else:
                tmp = g.T(level, c // 2)
                g.conv(m + ".cv1", buf, src_off, tmp, 0)
                g.conv(m + ".cv2", tmp, 0, buf, dst_off, res=buf, res_coff=src_off)
        g.conv(b + ".cv2", buf, 0, dst, dst_coff)

    def cv1_of(i, src, src_coff):
        cs_ = g.convs[g.idx["model.%d.cv1" % i]]
        return dict(kind=OPK_CONV, conv=g.idx["model.%d.cv1" % i], in0=src, in0_coff=src_coff, c0=cs_.cin, up0=0, in1=-1,
                    in1_coff=0, c1=0, out=-1, out_coff=0, res=-1, res_coff=0, pred_level=-1, pred_coff=0, p0=0, p1=0, p2=0, p3=0)

    t0, t1 = g.T(1, c64), g.T(2, c128)
    t2, t3, t4 = g.T(2, c256), g.T(3, c256), g.T(3, c512)
    t5, t6, t7, t8 = g.T(4, c512), g.T(4, c512), g.T(5, c1024), g.T(5, c1024)
    sppf = g.T(5, 2 * c1024)                              # [a | mp5 | mp9 | mp13], each c1024/2
    t9 = g.T(5, c1024)
    cat21 = g.T(5, c512 + c1024)                          # Concat[20, 10]: [model.20 out | model.10 out]
    cat18 = g.T(4, c256 + c512)                           # Concat[17, 13]: [model.17 out | model.13 out]
    t16, t19, t22 = g.T(3, c256), g.T(4, c512), g.T(5, c1024)
    g.op(OPK_STEM, "model.0", in0=0, c0=3, out=t0)
    g.conv("model.1", t0, 0, t1, 0)
    c3k2(2, cv1_of(2, t1, 0), 2, c256, n2, False, t2, 0)
    g.conv("model.3", t2, 0, t3, 0)
    c3k2(4, cv1_of(4, t3, 0), 3, c512, n2, False, t4, 0)
    g.conv("model.5", t4, 0, t5, 0)
    c3k2(6, cv1_of(6, t5, 0), 4, c512, n2, True, t6, 0)
    g.conv("model.7", t6, 0, t7, 0)
    c3k2(8, cv1_of(8, t7, 0), 5, c1024, n2, True, t8, 0)
    g.conv("model.9.cv1", t8, 0, sppf, 0)
    for j in range(3):
        g.op(OPK_POOL, None, in0=sppf, in0_coff=j * (c1024 // 2), c0=c1024 // 2, out=sppf, out_coff=(j + 1) * (c1024 // 2))
    g.conv("model.9.cv2", sppf, 0, t9, 0)
    # C2PSA: cv1 -> [a | b]; b <- PSABlock x n in place; cv2([a | b]) -> cat21[c512:]
    psa = g.T(5, 2 * pc)
    g.conv("model.10.cv1", t9, 0, psa, 0)
    qkv = g.T(5, pc + 2 * heads * kd)
    att, mix, ffn = g.T(5, pc), g.T(5, pc), g.T(5, 2 * pc)
    for j in range(n2):
        m = "model.10.m.%d" % j
        g.conv(m + ".attn.qkv", psa, pc, qkv, 0)
        g.op(OPK_ATTN, None, in0=qkv, in0_coff=0, c0=pc + 2 * heads * kd, out=att, out_coff=0, p0=heads, p1=kd, p2=hd)
        # x = (v @ attn^T) + pe(v): v is the last hd channels of every head's (2kd+hd)-channel block of qkv
        dwconv(m + ".attn.pe", qkv, 0, mix, 0, res=att, res_coff=0, blk=hd, gstride=2 * kd + hd, goff=2 * kd)
        g.conv(m + ".attn.proj", mix, 0, psa, pc, res=psa, res_coff=pc)       # b = b + attn(b)
        g.conv(m + ".ffn.0", psa, pc, ffn, 0)
        g.conv(m + ".ffn.1", ffn, 0, psa, pc, res=psa, res_coff=pc)           # b = b + ffn(b)
    g.conv("model.10.cv2", psa, 0, cat21, c512)
    o = cv1_of(13, cat21, c512)                           # Concat[Upsample(10), 6]
    o.update(c0=c1024, up0=1, in1=t6, in1_coff=0, c1=c512)
    c3k2(13, o, 4, c512, n2, False, cat18, c256)
    o = cv1_of(16, cat18, c256)                           # Concat[Upsample(13), 4]
    o.update(c0=c512, up0=1, in1=t4, in1_coff=0, c1=c512)
    c3k2(16, o, 3, c256, n2, False, t16, 0)
    g.conv("model.17", t16, 0, cat18, 0)
    c3k2(19, cv1_of(19, cat18, 0), 4, c512, n2, False, t19, 0)
    g.conv("model.20", t19, 0, cat21, 0)
    c3k2(22, cv1_of(22, cat21, 0), 5, c1024, n2, True, t22, 0)
    feats = (t16, t19, t22)
    for l in range(3):
        lev = 3 + l
        b0, b1 = g.T(lev, cb), g.T(lev, cb)
        d0, k0, d1, k1 = g.T(lev, chs[l]), g.T(lev, cc), g.T(lev, cc), g.T(lev, cc)
        bb, kk = "model.23.cv2.%d" % l, "model.23.cv3.%d" % l
        g.conv(bb + ".0", feats[l], 0, b0, 0)
        g.conv(bb + ".1", b0, 0, b1, 0)
        g.conv(bb + ".2", b1, 0, -1, 0, pred_level=l, pred_coff=0)
        dwconv(kk + ".0.0", feats[l], 0, d0, 0)
        g.conv(kk + ".0.1", d0, 0, k0, 0)
        dwconv(kk + ".1.0", k0, 0, d1, 0)
        g.conv(kk + ".1.1", d1, 0, k1, 0)
        g.conv(kk + ".2", k1, 0, -1, 0, pred_level=l, pred_coff=64)
    return g
