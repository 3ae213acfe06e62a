// Builds the conv list (canonical order) and the NHWC execution plan for a (scale, nc) pair.
#include "cy_plan.h"
#include <cmath>
#include <map>

namespace cy {

namespace {
struct Scale { double d, w; int mc; };
bool scale_of(char s, Scale* o) {
    switch (s) {
        case 'n': *o = {0.33, 0.25, 1024}; return true;
        case 's': *o = {0.33, 0.50, 1024}; return true;
        case 'm': *o = {0.67, 0.75, 768}; return true;
        case 'l': *o = {1.00, 1.00, 512}; return true;
        case 'x': *o = {1.00, 1.25, 512}; return true;
    }
    return false;
}
int make_div8(double x) { return (int)std::ceil(x / 8.0) * 8; }
int py_round(double x) {                     // Python round(): half to even
    double f = std::floor(x), d = x - f;
    if (d > 0.5) return (int)f + 1;
    if (d < 0.5) return (int)f;
    return ((long)f % 2 == 0) ? (int)f : (int)f + 1;
}

struct Builder {
    Plan p;
    std::map<std::string, int> idx;
    int T(int level, int C) { p.tensors.push_back({level, C}); return (int)p.tensors.size() - 1; }
    void add_conv(const std::string& name, int cin, int cout, int k, int s, int act = 1) {
        idx[name] = (int)p.convs.size();
        p.convs.push_back({name, cin, cout, k, s, act});
    }
    void decl_c2f(int i, int cin, int cout, int n) {
        const int h = cout / 2;
        const std::string b = "model." + std::to_string(i);
        add_conv(b + ".cv1", cin, 2 * h, 1, 1);
        add_conv(b + ".cv2", (2 + n) * h, cout, 1, 1);
        for (int j = 0; j < n; ++j) {
            add_conv(b + ".m." + std::to_string(j) + ".cv1", h, h, 3, 1);
            add_conv(b + ".m." + std::to_string(j) + ".cv2", h, h, 3, 1);
        }
    }
    Op base(const std::string& name) {
        Op o{};
        o.kind = OPK_CONV; o.conv = idx.at(name);
        o.in1 = -1; o.res = -1; o.pred_level = -1;
        return o;
    }
    void conv(const std::string& name, int in, int in_coff, int out, int out_coff, int res = -1, int res_coff = 0) {
        Op o = base(name);
        o.in0 = in; o.in0_coff = in_coff; o.c0 = p.convs[o.conv].cin;
        o.out = out; o.out_coff = out_coff; o.res = res; o.res_coff = res_coff;
        p.ops.push_back(o);
    }
    // C2f: cv1 -> [y0|y1] into the head of a (2+n)*h-channel buffer; bottleneck j reads slice 1+j and appends slice 2+j
    // (in place "concat"); cv2 reads the whole buffer.  `first` is the already-filled cv1 op description.
    void c2f(int i, Op first, int level, int cout, int n, bool shortcut, int dst, int dst_coff) {
        const int h = cout / 2;
        const std::string b = "model." + std::to_string(i);
        const int buf = T(level, (2 + n) * h), tmp = T(level, h);
        first.out = buf; first.out_coff = 0;
        p.ops.push_back(first);
        for (int j = 0; j < n; ++j) {
            const std::string m = b + ".m." + std::to_string(j);
            conv(m + ".cv1", buf, (1 + j) * h, tmp, 0);
            conv(m + ".cv2", tmp, 0, buf, (2 + j) * h, shortcut ? buf : -1, (1 + j) * h);
        }
        conv(b + ".cv2", buf, 0, dst, dst_coff);
    }
    Op cv1_single(int i, int in, int coff) {
        Op o = base("model." + std::to_string(i) + ".cv1");
        o.in0 = in; o.in0_coff = coff; o.c0 = p.convs[o.conv].cin;
        return o;
    }
};
}  // namespace

Plan build_plan(char scale, int nc) {
    Builder B;
    Plan& p = B.p;
    p.scale = scale; p.nc = nc; p.ok = false;
    Scale sc;
    if (!scale_of(scale, &sc) || nc < 1 || nc > 1000) { p.err = "unknown scale or bad nc"; return p; }
    auto ch = [&](int c) { return make_div8(std::min(c, sc.mc) * sc.w); };
    auto rep = [&](int n) { return std::max(py_round(n * sc.d), 1); };
    const int c1 = ch(64), c2 = ch(128), c3 = ch(256), c4 = ch(512), c5 = ch(1024), n3 = rep(3), n6 = rep(6);

    // ---- conv declarations, canonical order (must match caesar_yolo_amd/yolov8_spec.py:conv_list)
    B.add_conv("model.0", 3, c1, 3, 2);
    B.add_conv("model.1", c1, c2, 3, 2);
    B.decl_c2f(2, c2, c2, n3);
    B.add_conv("model.3", c2, c3, 3, 2);
    B.decl_c2f(4, c3, c3, n6);
    B.add_conv("model.5", c3, c4, 3, 2);
    B.decl_c2f(6, c4, c4, n6);
    B.add_conv("model.7", c4, c5, 3, 2);
    B.decl_c2f(8, c5, c5, n3);
    B.add_conv("model.9.cv1", c5, c5 / 2, 1, 1);
    B.add_conv("model.9.cv2", c5 * 2, c5, 1, 1);
    B.decl_c2f(12, c5 + c4, c4, n3);
    B.decl_c2f(15, c4 + c3, c3, n3);
    B.add_conv("model.16", c3, c3, 3, 2);
    B.decl_c2f(18, c3 + c4, c4, n3);
    B.add_conv("model.19", c4, c4, 3, 2);
    B.decl_c2f(21, c4 + c5, c5, n3);
    const int chs[3] = {c3, c4, c5};
    const int cb = std::max(16, std::max(chs[0] / 4, 64));
    const int cc = std::max(chs[0], std::min(nc, 100));
    for (int l = 0; l < 3; ++l) {
        const std::string b = "model.22.cv2." + std::to_string(l);
        B.add_conv(b + ".0", chs[l], cb, 3, 1);
        B.add_conv(b + ".1", cb, cb, 3, 1);
        B.add_conv(b + ".2", cb, 64, 1, 1, 0);
    }
    for (int l = 0; l < 3; ++l) {
        const std::string b = "model.22.cv3." + std::to_string(l);
        B.add_conv(b + ".0", chs[l], cc, 3, 1);
        B.add_conv(b + ".1", cc, cc, 3, 1);
        B.add_conv(b + ".2", cc, nc, 1, 1, 0);
    }

    // ---- execution plan
    const int in = B.T(0, 4);
    const int t0 = B.T(1, c1), t1 = B.T(2, c2), t2 = B.T(2, c2), t3 = B.T(3, c3), t4 = B.T(3, c3);
    const int t5 = B.T(4, c4), t6 = B.T(4, c4), t7 = B.T(5, c5), t8 = B.T(5, c5);
    const int sppf = B.T(5, 2 * c5);             // [a | mp5 | mp9 | mp13], each c5/2
    const int cat20 = B.T(5, c4 + c5);           // Concat[19, 9]: [model.19 out | model.9 out]
    const int cat17 = B.T(4, c3 + c4);           // Concat[16, 12]: [model.16 out | model.12 out]
    const int t15 = B.T(3, c3), t18 = B.T(4, c4), t21 = B.T(5, c5);
    {   // model.0
        Op o = B.base("model.0");
        o.kind = OPK_STEM; o.in0 = in; o.c0 = 3; o.out = t0;
        p.ops.push_back(o);
    }
    B.conv("model.1", t0, 0, t1, 0);
    B.c2f(2, B.cv1_single(2, t1, 0), 2, c2, n3, true, t2, 0);
    B.conv("model.3", t2, 0, t3, 0);
    B.c2f(4, B.cv1_single(4, t3, 0), 3, c3, n6, true, t4, 0);
    B.conv("model.5", t4, 0, t5, 0);
    B.c2f(6, B.cv1_single(6, t5, 0), 4, c4, n6, true, t6, 0);
    B.conv("model.7", t6, 0, t7, 0);
    B.c2f(8, B.cv1_single(8, t7, 0), 5, c5, n3, true, t8, 0);
    // SPPF: cv1 -> slice 0; three chained 5x5 max pools -> slices 1..3; cv2 -> cat20[c4:]
    B.conv("model.9.cv1", t8, 0, sppf, 0);
    for (int j = 0; j < 3; ++j) {
        Op o{};
        o.kind = OPK_POOL; o.conv = -1; o.in0 = sppf; o.in0_coff = j * (c5 / 2); o.c0 = c5 / 2;
        o.in1 = -1; o.res = -1; o.pred_level = -1; o.out = sppf; o.out_coff = (j + 1) * (c5 / 2);
        p.ops.push_back(o);
    }
    B.conv("model.9.cv2", sppf, 0, cat20, c4);
    {   // model.12: Concat[Upsample(9), 6] folded into cv1's gather
        Op o = B.base("model.12.cv1");
        o.in0 = cat20; o.in0_coff = c4; o.c0 = c5; o.up0 = 1;
        o.in1 = t6; o.in1_coff = 0; o.c1 = c4;
        B.c2f(12, o, 4, c4, n3, false, cat17, c3);
    }
    {   // model.15: Concat[Upsample(12), 4]
        Op o = B.base("model.15.cv1");
        o.in0 = cat17; o.in0_coff = c3; o.c0 = c4; o.up0 = 1;
        o.in1 = t4; o.in1_coff = 0; o.c1 = c3;
        B.c2f(15, o, 3, c3, n3, false, t15, 0);
    }
    B.conv("model.16", t15, 0, cat17, 0);
    B.c2f(18, B.cv1_single(18, cat17, 0), 4, c4, n3, false, t18, 0);
    B.conv("model.19", t18, 0, cat20, 0);
    B.c2f(21, B.cv1_single(21, cat20, 0), 5, c5, n3, false, t21, 0);
    const int feats[3] = {t15, t18, t21};
    for (int l = 0; l < 3; ++l) {
        const int lev = 3 + l;
        p.feat_level[l] = lev;
        const int b0 = B.T(lev, cb), b1 = B.T(lev, cb), k0 = B.T(lev, cc), k1 = B.T(lev, cc);
        const std::string bb = "model.22.cv2." + std::to_string(l), kk = "model.22.cv3." + std::to_string(l);
        B.conv(bb + ".0", feats[l], 0, b0, 0);
        B.conv(bb + ".1", b0, 0, b1, 0);
        { Op o = B.base(bb + ".2"); o.in0 = b1; o.c0 = cb; o.out = -1; o.pred_level = l; o.pred_coff = 0; p.ops.push_back(o); }
        B.conv(kk + ".0", feats[l], 0, k0, 0);
        B.conv(kk + ".1", k0, 0, k1, 0);
        { Op o = B.base(kk + ".2"); o.in0 = k1; o.c0 = cc; o.out = -1; o.pred_level = l; o.pred_coff = 64; p.ops.push_back(o); }
    }
    p.ok = true;
    return p;
}

}  // namespace cy
