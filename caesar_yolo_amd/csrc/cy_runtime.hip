// Runtime behind the C-ABI (include/caesar_yolo_hip.h): context, weight upload, workspace, and the host loops that
// enqueue the gfx950 kernels.  One context per GPU / per process (one process per GPU under torch.distributed).
#include "../../include/caesar_yolo_hip.h"
#include "cy_kernels.h"
#include "cy_plan.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace cy;

struct DevConv { void* w = nullptr; float* bias = nullptr; size_t wbytes = 0; float* stem_w = nullptr; void* w32 = nullptr; size_t w32bytes = 0;
                 float* oscale = nullptr;       // fp16x3 context: 2^-e per output channel (undoes the weight scale in the epilogue)
                 void* w2f = nullptr; size_t w2fbytes = 0;   // on the SECOND layer of a back-to-back pair: its weights in the first layer's accumulator order (pack_weights_fused2)
                 int passes = 3;                // fp16x3 context: passes over K -- 3, or 2 when the filter is fp16-exact up to a per-channel scale (then oscale = that scale)
                 float* dw_w = nullptr;         // dw_w: depth-wise 3x3 weights [9][C] fp32 (YOLO11)
                 void* bneck = nullptr; };      // on a bottleneck's cv1: register-fragment weights of the fused cv1+cv2 kernel (bneck64.hip)

struct cy_ctx {
    int device = 0;
    cy_config cfg{};
    Precision prec = PREC_F16;
    std::string err;
    bool loaded = false;
    Plan plan;
    std::vector<std::string> names;
    std::vector<DevConv> dconv;
    // workspace
    char* ws = nullptr; size_t ws_bytes = 0;           // activations
    // second workspace + stream: cy_detect_tiles runs batches of 64..239 tiles as two concurrent half-batches (forward_split)
    char* ws2 = nullptr; size_t ws2_bytes = 0; hipStream_t s_fwd2 = nullptr; hipEvent_t ev_split[2] = {nullptr, nullptr};
    char* ws3 = nullptr; size_t ws3_bytes = 0; hipStream_t s_small = nullptr;     // small-batch lane: its own workspace and (lowest-priority) stream
    std::vector<size_t> toff; std::vector<size_t> tbytes;   // per tensor, for the last forward geometry
    int lastB = 0, lastH = 0, lastW = 0;
    // stage buffers (sized at load for max_batch), two sets: cy_detect_tiles software-pipelines consecutive batches
    // (preprocessing of batch i+1 and post-processing of batch i-1 run on side streams beside the forward of batch i)
    struct StageBufs {
        void* netin = nullptr; float* pred = nullptr;
        float* cand = nullptr; int* cand_anchor = nullptr; int* cand_count = nullptr; uint64_t* keys = nullptr;
        float* det = nullptr; int* det_anchor = nullptr; int* det_count = nullptr; int* merge_err = nullptr; int* out_src = nullptr;
        double* pre_params = nullptr; double* pre_histeq = nullptr; double* pre_scratch = nullptr;
    } sb[3];                                            // [0], [1]: alternating sets of the batch pipeline; [2]: small batches (side forward stream)
    int slot = 0;                                       // buffer set used by the stage entry points
    StageBufs& S() { return sb[slot]; }
    size_t pre_scratch_elems = 0;
    int cap = 0, cap_pow2 = 0;
    hipStream_t s_pre = nullptr, s_post = nullptr;      // side streams of the pipelined cy_detect_tiles
    hipEvent_t ev_call = nullptr, ev_pre[3] = {nullptr, nullptr, nullptr}, ev_fwd[3] = {nullptr, nullptr, nullptr}, ev_post[3] = {nullptr, nullptr, nullptr};
    unsigned long batches = 0;                          // cy_detect_tiles calls on the main pipeline since load / flush
    unsigned long small_batches = 0;                    // ... and on the small-batch lane (buffer set 2, forward on s_fwd2)
    bool mosaic_dirty = true;                           // cy_mosaic_prepare / cy_detect_fence ran on the caller's stream since the last cy_detect_tiles
    const void* seen_mosaic[16] = {nullptr}; int n_seen = 0;   // mosaic buffers already ordered behind the caller's stream in this pipeline
    int* counters = nullptr;                            // device: [0] degenerate boxes dropped by the IoU merge, [1] tiles whose candidates overflowed `cap`
    // optional per-launch timing of the forward ops (hipEvents on the caller's stream)
    bool profiling = false;
    bool split_last = false;                             // the last forward ran as two half-batches (debug reads see only one)
    bool stem_fused_last = false;                        // the last forward ran model.0 + model.1 as one kernel (no model.0 tensor)
    int pw_fused_conv = -1;                              // the last forward ran this conv + the 1x1 behind it as one kernel (its tensor was not written)
    bool bneck_fused_last = false;                       // ... and 64-channel bottlenecks as one kernel each (their cv1 outputs do not exist)
    int prof_stride = 1; unsigned long fwd_calls = 0;   // profiling on: every prof_stride-th cy_forward call is timed
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    struct ProfRec { size_t e0, e1; int kind; double flops; int conv; int lane; };    // lane: 0 main stream(s), 1 small-batch lane
    std::vector<ProfRec> prof;
};

namespace {
thread_local std::string g_err;
int fail(cy_ctx* c, int code, const std::string& m) { if (c) c->err = m; g_err = m; return code; }
#define HIPCHK(c, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(c, CY_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline size_t esize(Precision p) { return p == PREC_F16 ? 2 : 4; }      // bytes per activation value (fp16x3: two fp16 halves)
inline Precision io_prec(Precision p) { return p == PREC_F16X3 ? PREC_F32 : p; }   // type of the network input buffer

int py_round_half_even(double x) {
    double f = std::floor(x), d = x - f;
    if (d > 0.5) return (int)f + 1;
    if (d < 0.5) return (int)f;
    return (((long)f) % 2 == 0) ? (int)f : (int)f + 1;
}

size_t tensor_elems_per_tile(const Plan& p, int H, int W) {
    size_t t = 0;
    for (auto& x : p.tensors) t += (size_t)(H >> x.level) * (W >> x.level) * x.C;
    return t;
}

void free_all(cy_ctx* c) {
    for (auto e : c->ev_pool) hipEventDestroy(e);
    c->ev_pool.clear(); c->ev_used = 0; c->prof.clear();
    for (auto& d : c->dconv) { if (d.w) hipFree(d.w); if (d.bias) hipFree(d.bias); if (d.stem_w) hipFree(d.stem_w); if (d.w32) hipFree(d.w32); if (d.dw_w) hipFree(d.dw_w); if (d.bneck) hipFree(d.bneck); if (d.oscale) hipFree(d.oscale); if (d.w2f) hipFree(d.w2f); }
    c->dconv.clear();
    if (c->ws) hipFree(c->ws);
    c->ws = nullptr;
    if (c->ws2) hipFree(c->ws2);
    c->ws2 = nullptr; c->ws2_bytes = 0;
    if (c->s_fwd2) { hipStreamDestroy(c->s_fwd2); c->s_fwd2 = nullptr; }
    if (c->ws3) hipFree(c->ws3);
    c->ws3 = nullptr; c->ws3_bytes = 0;
    c->s_small = nullptr;                                   // (an alias of s_fwd2)
    for (auto& e : c->ev_split) if (e) { hipEventDestroy(e); e = nullptr; }
    for (auto& b : c->sb) {
        void* ptrs[] = {b.netin, b.pred, b.cand, b.cand_anchor, b.cand_count, b.keys, b.det, b.det_anchor, b.det_count,
                        b.merge_err, b.out_src, b.pre_params, b.pre_histeq, b.pre_scratch};
        for (void* p : ptrs) if (p) hipFree(p);
        b = cy_ctx::StageBufs();
    }
    if (c->counters) { hipFree(c->counters); c->counters = nullptr; }
    if (c->s_pre) { hipStreamDestroy(c->s_pre); c->s_pre = nullptr; }
    if (c->s_post) { hipStreamDestroy(c->s_post); c->s_post = nullptr; }
    hipEvent_t* evs[] = {&c->ev_call, &c->ev_pre[0], &c->ev_pre[1], &c->ev_pre[2], &c->ev_fwd[0], &c->ev_fwd[1], &c->ev_fwd[2], &c->ev_post[0], &c->ev_post[1], &c->ev_post[2]};
    for (hipEvent_t* e : evs) if (*e) { hipEventDestroy(*e); *e = nullptr; }
    c->batches = 0; c->small_batches = 0;
}

struct Reader {
    const unsigned char* p; size_t n, o = 0; bool bad = false;
    uint32_t u32() { if (o + 4 > n) { bad = true; return 0; } uint32_t v; memcpy(&v, p + o, 4); o += 4; return v; }
    std::string str(uint32_t len) { if (o + len > n) { bad = true; return ""; } std::string s((const char*)p + o, len); o += (len + 3) / 4 * 4; return s; }
    const float* f32(size_t cnt) { if (o + 4 * cnt > n) { bad = true; return nullptr; } const float* r = (const float*)(p + o); o += 4 * cnt; return r; }
};

// CYW2: the execution plan travels in the file (caesar_yolo_amd/weights.py:write_cyw2, yolo11_graph.py)
int parse_plan_v2(cy_ctx* c, Reader& r, Plan* plan, uint32_t* nconv_out, std::vector<std::string>* names, uint32_t* version) {
    *version = r.u32();                                   // 3: conv headers carry a flags word (bit 0: a per-channel scale vector follows the bias)
    if (*version != 2 && *version != 3) return fail(c, CY_ERR_IO, "unsupported CYW2 version");
    if (r.o + 12 > r.n) return fail(c, CY_ERR_IO, "truncated weight file");
    plan->arch = std::string((const char*)r.p + r.o, strnlen((const char*)r.p + r.o, 8)); r.o += 8;
    plan->scale = (char)r.p[r.o]; r.o += 4;
    const uint32_t nc = r.u32(), nnames = r.u32(), nt = r.u32(), nops = r.u32(), nconv = r.u32();
    for (int l = 0; l < 3; ++l) plan->feat_level[l] = (int)r.u32();
    if (r.bad || nc < 1 || nc > 1000 || nt < 2 || nt > 4096 || nops < 1 || nops > 8192 || nconv < 1 || nconv > 4096)
        return fail(c, CY_ERR_IO, "implausible CYW2 header");
    plan->nc = (int)nc;
    for (uint32_t i = 0; i < nnames; ++i) { uint32_t l = r.u32(); names->push_back(r.str(l)); }
    for (uint32_t i = 0; i < nt; ++i) { const int lev = (int)r.u32(), C = (int)r.u32(); plan->tensors.push_back({lev, C}); }
    for (uint32_t i = 0; i < nops; ++i) {
        int v[20];
        for (int j = 0; j < 20; ++j) v[j] = (int)r.u32();
        Op o{};
        o.kind = (OpKind)v[0]; o.conv = v[1]; o.in0 = v[2]; o.in0_coff = v[3]; o.c0 = v[4]; o.up0 = v[5];
        o.in1 = v[6]; o.in1_coff = v[7]; o.c1 = v[8]; o.out = v[9]; o.out_coff = v[10]; o.res = v[11]; o.res_coff = v[12];
        o.pred_level = v[13]; o.pred_coff = v[14]; o.p0 = v[15]; o.p1 = v[16]; o.p2 = v[17]; o.p3 = v[18];
        auto tok = [&](int t, bool opt) { return (opt && t < 0) || (t >= 0 && t < (int)nt); };
        if (r.bad || v[0] < 0 || v[0] > OPK_ATTN || !tok(o.in0, false) || !tok(o.in1, true) || !tok(o.out, true) || !tok(o.res, true) ||
            (o.kind != OPK_POOL && o.kind != OPK_ATTN && (o.conv < 0 || o.conv >= (int)nconv)) ||
            (o.out < 0 && (o.pred_level < 0 || o.pred_level > 2)))
            return fail(c, CY_ERR_IO, "malformed op in CYW2 plan");
        plan->ops.push_back(o);
    }
    if (r.bad) return fail(c, CY_ERR_IO, "truncated weight file");
    for (auto& t : plan->tensors) if (t.level < 0 || t.level > 5 || t.C < 1 || t.C > 65536) return fail(c, CY_ERR_IO, "malformed tensor in CYW2 plan");
    *nconv_out = nconv;
    plan->ok = true;
    return CY_OK;
}

// ops[i], ops[i+1] = the two 3x3 convs of a 64-channel Bottleneck (x [+] cv2(cv1(x))) whose intermediate has no other reader:
// with CY_BNECK_FUSE=1 they run as ONE kernel in the fp16 context (bneck64.hip; off by default: measured slower so far)
bool bneck_pair(const Plan& p, size_t i) {
    if (i + 1 >= p.ops.size()) return false;
    const Op& o = p.ops[i]; const Op& n = p.ops[i + 1];
    if (o.kind != OPK_CONV || n.kind != OPK_CONV || o.conv < 0 || n.conv < 0 || o.out < 0 || n.out < 0) return false;
    const ConvDesc& d1 = p.convs[o.conv]; const ConvDesc& d2 = p.convs[n.conv];
    auto c64 = [](const ConvDesc& d) { return d.k == 3 && d.s == 1 && d.cin == 64 && d.cout == 64 && d.act && d.groups == 1; };
    if (!c64(d1) || !c64(d2)) return false;
    if (o.in1 >= 0 || o.up0 || o.res >= 0 || o.c0 != 64 || n.in1 >= 0 || n.up0 || n.c0 != 64) return false;
    if (n.in0 != o.out || n.in0_coff != o.out_coff) return false;
    if (p.tensors[o.in0].level != p.tensors[o.out].level || p.tensors[o.in0].level != p.tensors[n.out].level) return false;
    if (n.res >= 0 && (n.res != o.in0 || n.res_coff != o.in0_coff)) return false;      // the shortcut adds the bottleneck's own input
    if (n.out == o.in0 && n.out_coff == o.in0_coff) return false;                        // in place: patches would read overwritten halos
    // the intermediate must be dead after cv2: the next op that touches its tensor (C2f reuses one scratch tensor for all its
    // bottlenecks) has to overwrite exactly this slice before anything reads it
    for (size_t j = i + 2; j < p.ops.size(); ++j) {
        const Op& q = p.ops[j];
        if (q.in0 == o.out || q.in1 == o.out || q.res == o.out) return false;
        if (q.out == o.out) return q.out_coff == o.out_coff && q.conv >= 0 && p.convs[q.conv].cout == 64;
    }
    return true;
}

// ops[i] = a convolution whose 256 output channels all sit in one tile of the pixels-direct kernel (3x3 stride 2 or 1x1, SiLU, no
// residual, one input segment), ops[i+1] = the 1x1 convolution (256 -> <= 256 channels, one input segment, no residual) that is the ONLY
// reader of its output: in the fp16 context the pair runs as one kernel, the first layer's accumulators feeding the second GEMM in
// registers (conv1x1_direct_kernel<..., FUSE2>; yolov8l: model.3 -> model.4.cv1).  CY_FUSE_PW=0 turns it off.
bool pw_pair(const Plan& p, size_t i) {
    if (i + 1 >= p.ops.size()) return false;
    const Op& o = p.ops[i]; const Op& n = p.ops[i + 1];
    if (o.kind != OPK_CONV || n.kind != OPK_CONV || o.conv < 0 || n.conv < 0 || o.out < 0 || n.out < 0) return false;
    const ConvDesc& d1 = p.convs[o.conv]; const ConvDesc& d2 = p.convs[n.conv];
    if (d1.groups != 1 || d2.groups != 1 || d1.cout != 256 || !d1.act || d1.cin % 64) return false;
    if (!((d1.k == 3 && d1.s == 2 && d1.cin >= 128) || (d1.k == 1 && d1.s == 1))) return false;
    if (d2.k != 1 || d2.s != 1 || d2.cin != 256 || d2.cout > 256 || d2.cout % 16) return false;
    if (o.in1 >= 0 || o.up0 || o.res >= 0 || n.in1 >= 0 || n.up0 || n.res >= 0 || n.c0 != 256) return false;
    if (n.in0 != o.out || n.in0_coff != o.out_coff || p.tensors[o.out].level != p.tensors[n.out].level) return false;
    if (n.out == o.out) return false;
    int readers = 0;                                         // nothing else may read the first layer's output slice
    for (const Op& q : p.ops) readers += (q.in0 == o.out) + (q.in1 == o.out) + (q.res == o.out);
    if (readers != 1) return false;
    for (size_t j = i + 2; j < p.ops.size(); ++j)            // (a later op may overwrite that tensor; none may read it first: counted above)
        if (p.ops[j].out == o.out) break;
    return true;
}

// A CYW2 file carries its own execution plan; it is untrusted data.  Every channel slice an op names must lie inside its
// tensor, or the kernels would address outside the workspace (the tensors' extents are the buffer-resource bounds).
std::string validate_plan(const Plan& p) {
    auto T = [&](int t) -> const Tensor& { return p.tensors[t]; };
    auto inside = [&](int t, int coff, int n) { return coff >= 0 && n >= 1 && (long)coff + n <= T(t).C; };      // (untrusted u32 fields: no int overflow)
    for (size_t i = 0; i < p.ops.size(); ++i) {
        const Op& o = p.ops[i];
        const std::string at = " in op " + std::to_string(i);
        if (o.kind == OPK_POOL) {
            if (o.out < 0 || !inside(o.in0, o.in0_coff, o.c0) || !inside(o.out, o.out_coff, o.c0) || T(o.in0).level != T(o.out).level)
                return "pool slice outside its tensor" + at;
            continue;
        }
        if (o.kind == OPK_ATTN) {
            const long need = (long)o.p0 * (2L * o.p1 + o.p2);
            if (o.out < 0 || o.p0 < 1 || o.p1 < 1 || o.p2 < 1 || need > 65536 || !inside(o.in0, o.in0_coff, (int)need) ||
                (long)o.p0 * o.p2 > 65536 || !inside(o.out, o.out_coff, o.p0 * o.p2))
                return "attention slice outside its tensor" + at;
            continue;
        }
        const ConvDesc& d = p.convs[o.conv];
        if (o.kind == OPK_DWCONV) {
            // input channel of output channel ch: (ch / blk) * gstride + goff + ch % blk   (blk = 0: ch)
            const int last = d.cout - 1;
            const long cin_last = o.p0 > 0 ? (long)(last / o.p0) * o.p1 + o.p2 + last % o.p0 : last;
            if (o.out < 0 || o.p0 < 0 || o.p1 < 0 || o.p2 < 0 || (o.p0 > 0 && o.p0 % 8) || o.in0_coff < 0 ||
                o.in0_coff + cin_last + 1 > T(o.in0).C || !inside(o.out, o.out_coff, d.cout) ||
                (o.res >= 0 && !inside(o.res, o.res_coff, d.cout)))
                return "depth-wise slice outside its tensor" + at;
            continue;
        }
        if (o.kind == OPK_STEM) {
            if (o.out < 0 || !inside(o.out, o.out_coff, d.cout) || T(o.in0).C != 4 || d.cin != 3) return "malformed stem" + at;
            continue;
        }
        if (!inside(o.in0, o.in0_coff, o.c0) || (o.in1 >= 0 ? !inside(o.in1, o.in1_coff, o.c1) : o.c1 != 0) || o.c0 + o.c1 != d.cin)
            return "conv input slice does not match " + d.name + at;
        if (o.in1 >= 0 && (d.k != 1 || o.c0 % 64)) return "two-segment input needs a 1x1 conv and a 64-aligned boundary" + at;
        if (o.up0 && (d.k != 1 || T(o.in0).level < 1)) return "upsampled input needs a 1x1 conv" + at;
        if (o.out >= 0 && !inside(o.out, o.out_coff, d.cout)) return "conv output slice outside its tensor" + at;
        if (o.out < 0 && (o.pred_coff < 0 || o.pred_coff + d.cout > 64 + p.nc)) return "head slice outside the prediction row" + at;
        if (o.res >= 0 && (!inside(o.res, o.res_coff, d.cout) || (o.out >= 0 && T(o.res).level != T(o.out).level)))
            return "residual slice outside its tensor" + at;
    }
    return "";
}

int upload_weights(cy_ctx* c, const void* buf, size_t nbytes) {
    Reader r{(const unsigned char*)buf, nbytes};
    if (nbytes < 24 || (memcmp(buf, "CYW1", 4) != 0 && memcmp(buf, "CYW2", 4) != 0)) return fail(c, CY_ERR_IO, "not a CYW weight file");
    const bool v2 = memcmp(buf, "CYW2", 4) == 0;
    r.o = 4;
    Plan plan;
    uint32_t nc = 0, nconv = 0, fver = 0;
    std::vector<std::string> names;
    bool has_flags = false;                                  // CYW1 version 2 / CYW2 version 3: a flags word per conv header
    if (v2) {
        plan.ok = false;
        int rc = parse_plan_v2(c, r, &plan, &nconv, &names, &fver);
        if (rc) return rc;
        nc = (uint32_t)plan.nc;
        has_flags = fver == 3;
    } else {
        fver = r.u32();
        if (fver != 1 && fver != 2) return fail(c, CY_ERR_IO, "unsupported CYW version");
        has_flags = fver == 2;
        const char scale = (char)r.p[r.o]; r.o += 4;
        nc = r.u32(); nconv = r.u32();
        const uint32_t nnames = r.u32();
        plan = build_plan(scale, (int)nc);
        if (!plan.ok) return fail(c, CY_ERR_UNSUPPORTED, plan.err);
        if (plan.convs.size() != nconv) return fail(c, CY_ERR_IO, "weight file conv count does not match the graph");
        for (uint32_t i = 0; i < nnames; ++i) { uint32_t l = r.u32(); names.push_back(r.str(l)); }
    }
    HIPCHK(c, hipSetDevice(c->device));
    free_all(c);
    c->dconv.resize(nconv);
    std::vector<const float*> Wsrc(nconv, nullptr);          // folded fp32 weights inside `buf` (for the fused-bottleneck packing below)
    std::vector<char> packed;
    int stem_conv = -1;
    for (auto& o : plan.ops) if (o.kind == OPK_STEM) stem_conv = o.conv;
    for (uint32_t i = 0; i < nconv; ++i) {
        const uint32_t co = r.u32(), ci = r.u32(), k = r.u32(), s = r.u32(), act = r.u32();
        const uint32_t groups = v2 ? r.u32() : 1u;
        const uint32_t flags = has_flags ? r.u32() : 0u;
        const uint32_t nl = r.u32();
        const std::string name = r.str(nl);
        if (r.bad) return fail(c, CY_ERR_IO, "truncated weight file");
        if (v2) {
            if (co < 1 || ci < 1 || groups < 1 || ci % groups || (k != 1 && k != 3) || (s != 1 && s != 2))
                return fail(c, CY_ERR_IO, "malformed conv header: " + name);
            ConvDesc d; d.name = name; d.cin = (int)ci; d.cout = (int)co; d.k = (int)k; d.s = (int)s; d.act = (int)act; d.groups = (int)groups;
            plan.convs.push_back(d);
        } else {
            const ConvDesc& d = plan.convs[i];
            if (name != d.name || (int)co != d.cout || (int)ci != d.cin || (int)k != d.k || (int)s != d.s || (int)act != d.act)
                return fail(c, CY_ERR_IO, "weight file layer " + name + " does not match graph layer " + d.name);
        }
        const float* W = r.f32((size_t)co * (ci / groups) * k * k);
        const float* b = r.f32(co);
        // flags bit 0: scale[cout] with W[n] = fl32(w16[n] * scale[n]), w16 the checkpoint's fp16 filter (weights.py: fold(with_scale=True))
        const float* wscale = (flags & 1u) ? r.f32(co) : nullptr;
        if (r.bad) return fail(c, CY_ERR_IO, "truncated weight file");
        Wsrc[i] = W;
        DevConv& dc = c->dconv[i];
        const int cp = (co + 127) / 128 * 128;
        std::vector<float> bias(cp, 0.0f);
        memcpy(bias.data(), b, 4 * co);
        HIPCHK(c, hipMalloc(&dc.bias, 4 * cp));
        HIPCHK(c, hipMemcpy(dc.bias, bias.data(), 4 * cp, hipMemcpyHostToDevice));
        if (groups > 1) {   // depth-wise 3x3 (YOLO11): [9][C] fp32 for dwconv3x3_kernel
            if (groups != ci || co != ci || k != 3 || s != 1 || co % 8) return fail(c, CY_ERR_UNSUPPORTED, "only depth-wise 3x3 stride-1 grouped convs are supported: " + name);
            std::vector<float> dw(9 * (size_t)co);
            for (uint32_t ch = 0; ch < co; ++ch) for (int t = 0; t < 9; ++t) dw[(size_t)t * co + ch] = W[(size_t)ch * 9 + t];
            HIPCHK(c, hipMalloc(&dc.dw_w, 4 * dw.size()));
            HIPCHK(c, hipMemcpy(dc.dw_w, dw.data(), 4 * dw.size(), hipMemcpyHostToDevice));
            continue;
        }
        if ((int)i == stem_conv) {   // stem: [27][Cout] fp32, t = (kh*3+kw)*3 + c over the 3 network input channels
            if (co > 64 || co % 16) return fail(c, CY_ERR_UNSUPPORTED, "stem width not supported by the gfx950 stem kernel");
            std::vector<float> sw(27 * co);
            for (uint32_t o = 0; o < co; ++o) for (int cch = 0; cch < 3; ++cch) for (int t = 0; t < 9; ++t)
                sw[(size_t)(t * 3 + cch) * co + o] = W[((size_t)o * 3 + cch) * 9 + t];
            HIPCHK(c, hipMalloc(&dc.stem_w, 4 * sw.size()));
            HIPCHK(c, hipMemcpy(dc.stem_w, sw.data(), 4 * sw.size(), hipMemcpyHostToDevice));
            if (c->prec == PREC_F16 && co == 64) {
                std::vector<char> pk(64 * 32 * 2 + 64 * 64 * 2);          // panel of stem_mfma_kernel, then the one of stem_down_kernel
                pack_stem_weights(W, (int)co, pk.data());
                pack_stem_weights2(W, (int)co, pk.data() + 64 * 32 * 2);
                HIPCHK(c, hipMalloc(&dc.w, pk.size()));
                HIPCHK(c, hipMemcpy(dc.w, pk.data(), pk.size(), hipMemcpyHostToDevice));
            }
            continue;
        }
        if (ci % 8) return fail(c, CY_ERR_UNSUPPORTED, "conv input channels must be a multiple of 8: " + name);
        if (c->prec == PREC_F16X3) {
            std::vector<float> osc(cp);
            // two passes when the filter is exactly an fp16 filter times a per-channel scale (CY_X3_PASSES=3 forces the general form)
            dc.passes = env_knob("CY_X3_PASSES", 0) == 3 ? 3 : x3_passes(W, (int)co, (int)ci, (int)k, wscale);
            dc.wbytes = packed_weight_bytes_x3(co, ci, k, 128, dc.passes);
            packed.resize(dc.wbytes);
            pack_weights_x3(W, co, ci, k, packed.data(), osc.data(), 128, dc.passes, wscale);
            HIPCHK(c, hipMalloc(&dc.w, dc.wbytes));
            HIPCHK(c, hipMemcpy(dc.w, packed.data(), dc.wbytes, hipMemcpyHostToDevice));
            HIPCHK(c, hipMalloc(&dc.oscale, 4 * cp));
            HIPCHK(c, hipMemcpy(dc.oscale, osc.data(), 4 * cp, hipMemcpyHostToDevice));
            if (k == 3 && s == 1 && ci % 64 == 0) {          // second copy with 64-byte K chunks (conv3x3_wide_kernel)
                dc.w32bytes = packed_weight_bytes_x3(co, ci, k, 64, dc.passes);
                packed.resize(dc.w32bytes);
                pack_weights_x3(W, co, ci, k, packed.data(), osc.data(), 64, dc.passes, wscale);
                HIPCHK(c, hipMalloc(&dc.w32, dc.w32bytes));
                HIPCHK(c, hipMemcpy(dc.w32, packed.data(), dc.w32bytes, hipMemcpyHostToDevice));
            }
            continue;
        }
        dc.wbytes = packed_weight_bytes(c->prec, co, ci, k);
        packed.resize(dc.wbytes);
        pack_weights(c->prec, W, co, ci, k, packed.data());
        HIPCHK(c, hipMalloc(&dc.w, dc.wbytes));
        HIPCHK(c, hipMemcpy(dc.w, packed.data(), dc.wbytes, hipMemcpyHostToDevice));
        // (also the 3x3 s2 64->128 layer behind the stem: stem_down_kernel reads its weight fragments from this copy)
        if (c->prec == PREC_F16 && k == 3 && ((s == 1 && ((ci % 64 == 0 && co >= 33) || (ci == 32 && co <= 64))) || (s == 2 && ci == 64 && co == 128))) {   // second copy (64-byte K chunks) for conv3x3_wide_kernel / conv3x3_c64_kernel<32>
            dc.w32bytes = packed_weight_bytes(c->prec, co, ci, k, 64);
            packed.resize(dc.w32bytes);
            pack_weights(c->prec, W, co, ci, k, packed.data(), 64);
            HIPCHK(c, hipMalloc(&dc.w32, dc.w32bytes));
            HIPCHK(c, hipMemcpy(dc.w32, packed.data(), dc.w32bytes, hipMemcpyHostToDevice));
        }
    }
    if (v2) {
        const std::string bad = validate_plan(plan);
        if (!bad.empty()) { free_all(c); return fail(c, CY_ERR_IO, "malformed CYW2 plan: " + bad); }
    }
    if (c->prec == PREC_F16) {
        for (size_t i = 0; i + 1 < plan.ops.size(); ++i) {      // back-to-back pairs: the second layer's weights in accumulator order
            if (!pw_pair(plan, i)) continue;
            const int c2 = plan.ops[i + 1].conv;
            const ConvDesc& d2 = plan.convs[c2];
            DevConv& dc2 = c->dconv[c2];
            dc2.w2fbytes = packed_weight_bytes(PREC_F16, d2.cout, d2.cin, 1);
            std::vector<char> pk(dc2.w2fbytes);
            pack_weights_fused2(Wsrc[c2], d2.cout, d2.cin, pk.data());
            HIPCHK(c, hipMalloc(&dc2.w2f, dc2.w2fbytes));
            HIPCHK(c, hipMemcpy(dc2.w2f, pk.data(), dc2.w2fbytes, hipMemcpyHostToDevice));
        }
        std::vector<char> wf(BNECK_WFRAG_BYTES);
        for (size_t i = 0; i + 1 < plan.ops.size(); ++i) {
            if (!bneck_pair(plan, i)) continue;
            const int c1 = plan.ops[i].conv, c2 = plan.ops[i + 1].conv;
            pack_bneck_weights(Wsrc[c1], Wsrc[c2], wf.data());
            HIPCHK(c, hipMalloc(&c->dconv[c1].bneck, wf.size()));
            HIPCHK(c, hipMemcpy(c->dconv[c1].bneck, wf.data(), wf.size(), hipMemcpyHostToDevice));
        }
    }
    c->plan = plan; c->names = names;
    // ---- workspace for (max_batch, max_h, max_w)
    const cy_config& g = c->cfg;
    const size_t es = esize(c->prec);
    size_t act = 0;
    for (auto& t : plan.tensors) act += align_up((size_t)g.max_batch * (g.max_h >> t.level) * (g.max_w >> t.level) * t.C * es, 256);
    c->ws_bytes = act;
    HIPCHK(c, hipMalloc(&c->ws, c->ws_bytes));
    const int A = cy_num_anchors(g.max_h, g.max_w);
    // candidates per tile: ultralytics keeps at most max_nms = 30000 by score; with cap >= the anchor count of the largest
    // letterboxed tile an overflow cannot happen (8400 anchors at 640x640).  A smaller explicit max_cand is honoured, and a
    // tile that overflows it is counted (cy_detect_counters) -- its surplus candidates are dropped in arrival order.
    c->cap = g.max_cand > 0 ? g.max_cand : A;
    if (c->cap > 30000) c->cap = 30000;
    c->cap = (c->cap + 63) / 64 * 64;
    c->cap_pow2 = 1; while (c->cap_pow2 < c->cap) c->cap_pow2 <<= 1;
    const size_t Bm = g.max_batch;
    c->pre_scratch_elems = Bm * 3 * (size_t)g.max_h * g.max_w;
    for (auto& b : c->sb) {
        HIPCHK(c, hipMalloc(&b.netin, Bm * g.max_h * g.max_w * 4 * es));
        HIPCHK(c, hipMalloc(&b.pred, Bm * A * (64 + nc) * sizeof(float)));
        HIPCHK(c, hipMalloc(&b.cand, Bm * c->cap * 6 * sizeof(float)));
        HIPCHK(c, hipMalloc(&b.cand_anchor, Bm * c->cap * sizeof(int)));
        HIPCHK(c, hipMalloc(&b.cand_count, Bm * sizeof(int)));
        HIPCHK(c, hipMalloc(&b.keys, Bm * c->cap_pow2 * sizeof(uint64_t)));
        HIPCHK(c, hipMalloc(&b.det, Bm * CY_MAX_DET * 6 * sizeof(float)));
        HIPCHK(c, hipMalloc(&b.det_anchor, Bm * CY_MAX_DET * sizeof(int)));
        HIPCHK(c, hipMalloc(&b.det_count, Bm * sizeof(int)));
        HIPCHK(c, hipMalloc(&b.merge_err, Bm * sizeof(int)));
        HIPCHK(c, hipMalloc(&b.out_src, Bm * CY_MAX_DET * sizeof(int)));
        HIPCHK(c, hipMalloc(&b.pre_params, Bm * 3 * CY_MAX_STAGES * 4 * sizeof(double)));
        HIPCHK(c, hipMalloc(&b.pre_histeq, Bm * 3 * 520 * sizeof(double)));
        HIPCHK(c, hipMalloc(&b.pre_scratch, c->pre_scratch_elems * sizeof(double)));
    }
    HIPCHK(c, hipMalloc(&c->counters, 4 * sizeof(int)));
    HIPCHK(c, hipMemset(c->counters, 0, 4 * sizeof(int)));
    // side streams at the LOWEST priority: preprocessing and decode/NMS/merge fill the gaps of the conv stack on the caller's
    // stream instead of competing with it for CUs (CY_SIDE_PRIO=0 restores default-priority streams).  Round 4: confining them to a
    // subset of the CUs instead (hipExtStreamCreateWithCUMask, 64 / 32 / 16 CUs spread over the XCDs) costs 10-12 % of the S16k rate
    // (8950 -> 8000 / 8000 / 7910 tiles/s): the pass then waits for its side kernels.
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);          // lo = numerically greatest = lowest priority
    const bool low = !(getenv("CY_SIDE_PRIO") && atoi(getenv("CY_SIDE_PRIO")) == 0);
    HIPCHK(c, hipStreamCreateWithPriority(&c->s_pre, hipStreamNonBlocking, low ? prio_lo : 0));
    HIPCHK(c, hipStreamCreateWithPriority(&c->s_post, hipStreamNonBlocking, low ? prio_lo : 0));
    hipEvent_t* evs[] = {&c->ev_call, &c->ev_pre[0], &c->ev_pre[1], &c->ev_pre[2], &c->ev_fwd[0], &c->ev_fwd[1], &c->ev_fwd[2], &c->ev_post[0], &c->ev_post[1], &c->ev_post[2]};
    for (hipEvent_t* e : evs) HIPCHK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    c->loaded = true;
    return CY_OK;
}

int layout_tensors(cy_ctx* c, int B, int H, int W, size_t capacity) {
    const size_t es = esize(c->prec);
    c->toff.assign(c->plan.tensors.size(), 0);
    c->tbytes.assign(c->plan.tensors.size(), 0);
    size_t off = 0;
    for (size_t i = 0; i < c->plan.tensors.size(); ++i) {
        const Tensor& t = c->plan.tensors[i];
        const size_t b = (size_t)B * (H >> t.level) * (W >> t.level) * t.C * es;
        c->toff[i] = off; c->tbytes[i] = b;
        off += align_up(b, 256);
        if (b >= 0xFFFFFF00ull) return fail(c, CY_ERR_ARG, "a tensor exceeds the 4 GiB buffer-addressing limit; lower the batch");
    }
    if (off > capacity) return fail(c, CY_ERR_ARG, "batch/shape exceeds the workspace sized at cy_create");
    c->lastB = B; c->lastH = H; c->lastW = W;
    return CY_OK;
}
}  // namespace

extern "C" {

int cy_create(int device, const cy_config* cfg, cy_ctx** out) {
    if (!cfg || !out) return fail(nullptr, CY_ERR_ARG, "null argument");
    if (cfg->max_batch < 1 || cfg->max_h < 32 || cfg->max_w < 32 || cfg->max_h % 32 || cfg->max_w % 32)
        return fail(nullptr, CY_ERR_ARG, "max_batch >= 1 and max_h/max_w multiples of 32 required");
    if (cfg->precision != CY_F16 && cfg->precision != CY_F32 && cfg->precision != CY_F16X3) return fail(nullptr, CY_ERR_ARG, "bad precision");
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device < 0 || device >= n)
        return fail(nullptr, CY_ERR_HIP, std::string("no such HIP device (this library has no CPU fallback): hipGetDeviceCount -> ") +
                                             hipGetErrorString(e) + ", " + std::to_string(n) + " device(s), requested " + std::to_string(device));
    cy_ctx* c = new cy_ctx();
    c->device = device; c->cfg = *cfg; c->prec = cfg->precision == CY_F16 ? PREC_F16 : (cfg->precision == CY_F16X3 ? PREC_F16X3 : PREC_F32);
    *out = c;
    return CY_OK;
}

int cy_destroy(cy_ctx* c) {
    if (!c) return CY_OK;
    hipSetDevice(c->device);
    free_all(c);
    delete c;
    return CY_OK;
}

const char* cy_last_error(const cy_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

int cy_load_weights_mem(cy_ctx* c, const void* buf, size_t nbytes) {
    if (!c || !buf) return fail(c, CY_ERR_ARG, "null argument");
    return upload_weights(c, buf, nbytes);
}

int cy_load_weights(cy_ctx* c, const char* path) {
    if (!c || !path) return fail(c, CY_ERR_ARG, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(c, CY_ERR_IO, std::string("cannot open ") + path);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> buf(n);
    const size_t got = fread(buf.data(), 1, n, f);
    fclose(f);
    if ((long)got != n) return fail(c, CY_ERR_IO, "short read");
    return upload_weights(c, buf.data(), buf.size());
}

int cy_num_classes(const cy_ctx* c) { return c && c->loaded ? c->plan.nc : -1; }
int cy_weight_passes(const cy_ctx* c, int* out2) {
    if (!c || !c->loaded || !out2) return CY_ERR_ARG;
    out2[0] = out2[1] = 0;
    if (c->prec == PREC_F16X3)
        for (const auto& d : c->dconv) if (d.oscale) out2[d.passes == 2 ? 0 : 1] += 1;
    return CY_OK;
}
const char* cy_class_name(const cy_ctx* c, int i) {
    if (!c || !c->loaded || i < 0 || i >= (int)c->names.size()) return nullptr;
    return c->names[i].c_str();
}

int cy_plan_num_convs(char scale, int nc) { Plan p = build_plan(scale, nc); return p.ok ? (int)p.convs.size() : CY_ERR_ARG; }
int cy_plan_conv_desc(char scale, int nc, int idx, cy_conv_desc* out) {
    Plan p = build_plan(scale, nc);
    if (!p.ok || !out || idx < 0 || idx >= (int)p.convs.size()) return CY_ERR_ARG;
    const ConvDesc& d = p.convs[idx];
    memset(out, 0, sizeof(*out));
    strncpy(out->name, d.name.c_str(), sizeof(out->name) - 1);
    out->cin = d.cin; out->cout = d.cout; out->k = d.k; out->s = d.s; out->act = d.act;
    return CY_OK;
}

int cy_letterbox_geometry(int h0, int w0, int imgsz, cy_letterbox* o) {
    // ultralytics LetterBox(auto=True, scaleup=True, center=True, stride=32): SURVEY.md Appendix A.1 step 2
    if (!o || h0 < 1 || w0 < 1 || imgsz < 32) return CY_ERR_ARG;
    const double r = std::fmin((double)imgsz / h0, (double)imgsz / w0);
    o->new_w = py_round_half_even(w0 * r); o->new_h = py_round_half_even(h0 * r);
    double dw = (imgsz - o->new_w) % 32, dh = (imgsz - o->new_h) % 32;
    if (dw < 0) dw += 32;
    if (dh < 0) dh += 32;
    dw /= 2; dh /= 2;
    o->top = py_round_half_even(dh - 0.1); o->left = py_round_half_even(dw - 0.1);
    const int bottom = py_round_half_even(dh + 0.1), right = py_round_half_even(dw + 0.1);
    o->H = o->new_h + o->top + bottom; o->W = o->new_w + o->left + right;
    return CY_OK;
}

int cy_num_anchors(int H, int W) { int n = 0; for (int s = 8; s <= 32; s *= 2) n += ((H + s - 1) / s) * ((W + s - 1) / s); return n; }
size_t cy_pred_elems(const cy_ctx* c, int B, int H, int W) {
    return c && c->loaded ? (size_t)B * cy_num_anchors(H, W) * (64 + c->plan.nc) : 0;
}

int cy_mosaic_prepare(cy_ctx* c, float* d_data, size_t n, int big_endian, void* stream) {
    if (!c || !d_data) return fail(c, CY_ERR_ARG, "null argument");
    HIPCHK(c, launch_mosaic_prepare(d_data, n, big_endian, (hipStream_t)stream));
    c->mosaic_dirty = true;
    return CY_OK;
}

static int forward_on(cy_ctx* c, const void* d_netin, int B, int H, int W, float* d_pred, hipStream_t s, char* ws, size_t ws_cap,
                      bool may_profile);

int cy_forward(cy_ctx* c, const void* d_netin, int B, int H, int W, float* d_pred, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    c->split_last = false;
    return forward_on(c, d_netin, B, H, W, d_pred, (hipStream_t)stream, c->ws, c->ws_bytes, true);
}

// The forward of one batch as TWO half-batches on two streams (own workspace each).  Every layer is one kernel launch
// whose last round of workgroups leaves part of the chip idle, and kernels of one stream cannot overlap: the other half's
// kernels fill those tails.  Measured on MI355X (tools/concurrent_forward.py, forward only): -6.8 % at 208 tiles, -3.7 % at
// 224, -1.3 % at 254.  Used for 64..239 tiles (the per-rank shares at N >= 2); a full batch stays on one stream, so that the
// per-launch event timing of bench.py at N = 1 means exclusive use of the GPU.  CY_DUAL_FORWARD: 0 off, 2 from 2 tiles on.
// ONE extra forward stream serves both the second half of a split batch and the small-batch lane.  With a stream of its own per
// role a context had five streams (caller's, preprocessing, post-processing, split, small lane) and the HIP runtime maps streams
// onto FOUR hardware queues: whichever two shared a queue serialised -- measured as a per-rank pass of 41.6 instead of 24.6 ms when
// one process used both roles (a rank of the N = 2 / 4 partitions: full batches of 64..239 tiles AND ragged classes; order of
// stream creation decided which pair collided).  A small batch now queues behind the second half of a split batch when both are
// in flight; at N = 1 (no split batches) and N = 8 (a rank has either kind) nothing changes.
static int ensure_side_forward_stream(cy_ctx* c) {
    if (c->s_fwd2) return CY_OK;
    HIPCHK(c, hipStreamCreateWithFlags(&c->s_fwd2, hipStreamNonBlocking));
    return CY_OK;
}
static int ensure_second_workspace(cy_ctx* c) {
    if (c->ws2) return CY_OK;
    c->ws2_bytes = c->ws_bytes / 2 + (1u << 20);            // the second half is never the larger one
    HIPCHK(c, hipMalloc(&c->ws2, c->ws2_bytes));
    int rc = ensure_side_forward_stream(c);
    if (rc) return rc;
    for (auto& ev : c->ev_split) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return CY_OK;
}
// The small-batch lane (cy_detect_tiles): small batches run their forward on the side forward stream with a workspace of their
// own, so that their ~105 launches of a few workgroups each take what the full batch beside them leaves free.
static int ensure_small_lane(cy_ctx* c) {
    if (c->ws3) return CY_OK;
    c->ws3_bytes = c->ws_bytes / 3 + (1u << 20);            // a small batch holds at most max_batch / 3 tiles
    HIPCHK(c, hipMalloc(&c->ws3, c->ws3_bytes));
    int rc = ensure_side_forward_stream(c);
    if (rc) return rc;
    c->s_small = c->s_fwd2;                                 // (shared: see ensure_side_forward_stream)
    return CY_OK;
}
static int dual_mode() { const char* e = getenv("CY_DUAL_FORWARD"); return e ? atoi(e) : 1; }

static int forward_split(cy_ctx* c, const void* d_netin, int B, int H, int W, float* d_pred, hipStream_t sm) {
    const int mode = dual_mode();
    const bool split = c->prec == PREC_F16 && mode != 0 && B >= 2 && (mode > 1 || (B >= 64 && B < 240));      // (fp16x3: a 128-tile batch already fills the chip 3x longer per launch)
    c->split_last = split;
    if (!split) return forward_on(c, d_netin, B, H, W, d_pred, sm, c->ws, c->ws_bytes, true);
    int rc0 = ensure_second_workspace(c);
    if (rc0) return rc0;
    const int h = B - B / 2;
    const size_t in_half = (size_t)h * H * W * 4 * esize(c->prec);
    const size_t pred_half = (size_t)h * cy_num_anchors(H, W) * (64 + c->plan.nc);
    HIPCHK(c, hipEventRecord(c->ev_split[0], sm));           // everything the forward waits for is already queued on sm
    HIPCHK(c, hipStreamWaitEvent(c->s_fwd2, c->ev_split[0], 0));
    int rc = forward_on(c, d_netin, h, H, W, d_pred, sm, c->ws, c->ws_bytes, true);
    if (rc) return rc;
    rc = forward_on(c, (const char*)d_netin + in_half, B - h, H, W, d_pred + pred_half, c->s_fwd2, c->ws2, c->ws2_bytes, false);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_split[1], c->s_fwd2));
    HIPCHK(c, hipStreamWaitEvent(sm, c->ev_split[1], 0));
    return CY_OK;
}

static int forward_on(cy_ctx* c, const void* d_netin, int B, int H, int W, float* d_pred, hipStream_t s, char* ws, size_t ws_cap,
                      bool may_profile) {
    if (!d_netin || !d_pred || B < 1 || H % 32 || W % 32 || H < 32 || W < 32) return fail(c, CY_ERR_ARG, "bad forward arguments");
    int rc = layout_tensors(c, B, H, W, ws_cap);
    if (rc) return rc;
    const Plan& p = c->plan;
    const size_t es = esize(c->prec);
    const bool x3 = c->prec == PREC_F16X3;
    const int cm = x3 ? 2 : 1;                           // halves per activation value: pixel strides are cm * C, the low halves C behind
    const int A = cy_num_anchors(H, W);
    int a_off[3], acc = 0;
    for (int l = 0; l < 3; ++l) { a_off[l] = acc; acc += (H >> (3 + l)) * (W >> (3 + l)); }
    auto tptr = [&](int t) -> char* { return t == 0 ? (char*)const_cast<void*>(d_netin) : ws + c->toff[t]; };
    auto stamp = [&]() -> size_t {
        if (c->ev_used == c->ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); c->ev_pool.push_back(e); }
        hipEventRecord(c->ev_pool[c->ev_used], s);
        return c->ev_used++;
    };
    const bool prof_now = may_profile && c->profiling && (c->fwd_calls++ % (unsigned long)c->prof_stride) == 0;
    size_t ev_prev = prof_now ? stamp() : 0;
    int cur_conv = -1;
    auto prof_done = [&](int kind, double flops) {
        if (!prof_now) return;
        const size_t e = stamp();
        c->prof.push_back({ev_prev, e, kind, flops, cur_conv, (c->ws3 && ws == c->ws3) ? 1 : 0});
        ev_prev = e;
    };
    // Optional (CY_SUB=n): run the full-resolution head of the graph (stem .. model.2, levels 1-2; tensors of 134-537 MB
    // at batch 64) in sub-batches so that producer and consumer could meet in the 256 MiB Infinity Cache; only the
    // stage output and the network input are addressed per sub-batch.  Measured on MI355X: 1-6 % SLOWER at n = 4/8/16
    // (the extra launches cost more than the cache buys), so it is off by default and kept as a tuning knob.
    static const int sub_env = getenv("CY_SUB") ? atoi(getenv("CY_SUB")) : 0;   // measured: no gain on MI355X (profiles/), off by default
    size_t n_head = 0;
    while (n_head < p.ops.size()) {
        const Op& o = p.ops[n_head];
        const int lo = o.out >= 0 ? p.tensors[o.out].level : 9;
        if (lo > 2 || o.in1 >= 0) break;
        ++n_head;
    }
    std::vector<char> crosses(p.tensors.size(), 0);      // written in the head, read after it
    for (size_t i = n_head; i < p.ops.size(); ++i) {
        const Op& o = p.ops[i];
        if (o.in0 >= 0) crosses[o.in0] = 1;
        if (o.in1 >= 0) crosses[o.in1] = 1;
        if (o.res >= 0) crosses[o.res] = 1;
    }
    crosses[0] = 1;
    const int sub = (sub_env > 0 && sub_env < B && n_head > 0) ? sub_env : B;

    // model.0 + model.1 in one kernel (fp16 context) when the stem's only reader is a 3x3 s2 64->128 SiLU conv and the launch
    // fills the chip.  CY_STEM_FUSE: 0 = off, 2 = regardless of size (parity tests); read per call.
    const int fuse_env = getenv("CY_STEM_FUSE") ? atoi(getenv("CY_STEM_FUSE")) : 1;
    auto stem_fusable = [&](const Op& o, const Op& n) -> bool {
        if (c->prec != PREC_F16 || !fuse_env || o.kind != OPK_STEM || n.kind != OPK_CONV || n.conv < 0 || n.out < 0) return false;
        const ConvDesc& d0 = p.convs[o.conv]; const ConvDesc& d = p.convs[n.conv];
        if (d0.cout != 64 || !c->dconv[o.conv].w || !c->dconv[n.conv].w32) return false;
        if (d.k != 3 || d.s != 2 || d.cin != 64 || d.cout != 128 || !d.act) return false;
        if (n.in0 != o.out || n.in0_coff != o.out_coff || n.c0 != 64 || n.in1 >= 0 || n.res >= 0 || n.up0) return false;
        if (p.tensors[o.out].C != 64) return false;
        int readers = 0;
        for (const Op& q : p.ops) readers += (q.in0 == o.out) + (q.in1 == o.out) + (q.res == o.out);
        return readers == 1;
    };
    c->stem_fused_last = false;
    c->pw_fused_conv = -1;
    const int pw_env = env_knob("CY_FUSE_PW", 1);             // back-to-back pairs (pw_pair); read per call: the parity tests run both forms

    struct HeadPend { ConvArgs a; double flops; int conv; bool on; };
    HeadPend pend[3] = {};                                    // deferred box-branch output convolutions, per stride level
    const int bneck_env = env_knob("CY_BNECK_FUSE", 0);       // off by default (slower than two launches so far, see bneck64.hip); read per call: the parity tests run both forms
    auto run_op = [&](const Op& o, const Op* next, int b0, int Bn, bool* fused) -> int {
        auto tp = [&](int t) -> char* {                    // tensor base for images [b0, b0+Bn)
            char* base = tptr(t);
            if (b0 && crosses[t]) base += (size_t)b0 * (H >> p.tensors[t].level) * (W >> p.tensors[t].level) * p.tensors[t].C * es;
            return base;
        };
        cur_conv = o.conv;
        if (bneck_env && c->prec == PREC_F16 && o.kind == OPK_CONV && next && c->dconv[o.conv].bneck &&
            bneck_pair(p, (size_t)(&o - p.ops.data()))) {
            const Op& n = *next;
            const Tensor& ti = p.tensors[o.in0]; const Tensor& to = p.tensors[n.out];
            BneckArgs a{};
            a.in = tp(o.in0); a.in_ct = ti.C; a.in_coff = o.in0_coff;
            a.B = Bn; a.H = H >> ti.level; a.W = W >> ti.level;
            a.in_bytes = (uint32_t)((size_t)Bn * a.H * a.W * ti.C * es);
            a.out = tp(n.out); a.out_ct = to.C; a.out_coff = n.out_coff;
            a.wfrag = c->dconv[o.conv].bneck; a.bias1 = c->dconv[o.conv].bias; a.bias2 = c->dconv[n.conv].bias;
            a.shortcut = n.res >= 0;
            HIPCHK(c, launch_bneck64(a, s));
            prof_done(CONV_NUM_VARIANTS + 4, 2.0 * (2.0 * Bn * a.H * a.W * 64.0 * 64.0 * 9.0));
            *fused = true; c->bneck_fused_last = true;
            return CY_OK;
        }
        if (o.kind == OPK_STEM && next && stem_fusable(o, *next)) {
            const Op& n = *next;
            const Tensor& ti = p.tensors[o.in0]; const Tensor& t1 = p.tensors[o.out]; const Tensor& to = p.tensors[n.out];
            StemDownArgs a{};
            a.B = Bn; a.Hi = H >> ti.level; a.Wi = W >> ti.level; a.H1 = H >> t1.level; a.W1 = W >> t1.level;
            a.Ho = a.H1 / 2; a.Wo = a.W1 / 2;
            a.in = tp(o.in0); a.in_bytes = (uint32_t)((size_t)Bn * a.Hi * a.Wi * ti.C * es);
            a.wpk2 = reinterpret_cast<const char*>(c->dconv[o.conv].w) + 64 * 32 * 2; a.bias0 = c->dconv[o.conv].bias;
            a.wgt32 = c->dconv[n.conv].w32; a.wgt32_bytes = (uint32_t)c->dconv[n.conv].w32bytes; a.bias1 = c->dconv[n.conv].bias;
            a.out = tp(n.out); a.out_ct = to.C; a.out_coff = n.out_coff;
            if (ti.C == 4 && (fuse_env > 1 || batch_invariant() || stem_down_blocks(a) >= 256)) {
                HIPCHK(c, launch_stem_down(a, s));
                prof_done(CONV_NUM_VARIANTS, 2.0 * Bn * a.H1 * a.W1 * 64 * 27.0 + 2.0 * Bn * a.Ho * a.Wo * 128.0 * 64 * 9);
                *fused = true; c->stem_fused_last = true;
                return CY_OK;
            }
        }
        if (o.kind == OPK_STEM) {
            const Tensor& ti = p.tensors[o.in0]; const Tensor& to = p.tensors[o.out];
            StemArgs a{};
            a.in = tp(o.in0); a.out = tp(o.out); a.w = c->dconv[o.conv].stem_w; a.bias = c->dconv[o.conv].bias; a.wpk = c->dconv[o.conv].w;
            a.B = Bn; a.Hi = H >> ti.level; a.Wi = W >> ti.level; a.Ho = H >> to.level; a.Wo = W >> to.level;
            a.Cout = p.convs[o.conv].cout; a.out_ct = cm * to.C; a.out_coff = o.out_coff; a.out_lo = x3 ? to.C : 0;
            HIPCHK(c, launch_stem(c->prec, a, s));
            prof_done(CONV_NUM_VARIANTS, 2.0 * Bn * a.Ho * a.Wo * a.Cout * 27.0);
        } else if (o.kind == OPK_POOL) {
            const Tensor& t = p.tensors[o.in0];
            PoolArgs a{};
            a.src = tp(o.in0); a.dst = tp(o.out); a.ct = cm * t.C; a.src_coff = o.in0_coff; a.dst_coff = o.out_coff; a.lo = x3 ? t.C : 0;
            a.C = o.c0; a.B = Bn; a.H = H >> t.level; a.W = W >> t.level;
            HIPCHK(c, launch_pool5(c->prec, a, s));
            prof_done(CONV_NUM_VARIANTS + 1, 0.0);
        } else if (o.kind == OPK_DWCONV) {
            const ConvDesc& d = p.convs[o.conv];
            const Tensor& ti = p.tensors[o.in0]; const Tensor& to = p.tensors[o.out];
            DwArgs a{};
            a.in = tp(o.in0); a.in_ct = cm * ti.C; a.in_coff = o.in0_coff; a.out = tp(o.out); a.out_ct = cm * to.C; a.out_coff = o.out_coff;
            a.in_lo = ti.C; a.out_lo = to.C;
            if (o.res >= 0) { a.res = tp(o.res); a.res_ct = cm * p.tensors[o.res].C; a.res_coff = o.res_coff; a.res_lo = p.tensors[o.res].C; }
            a.w = c->dconv[o.conv].dw_w; a.bias = c->dconv[o.conv].bias;
            a.B = Bn; a.H = H >> ti.level; a.W = W >> ti.level; a.C = d.cout; a.act = d.act;
            a.blk = o.p0; a.gstride = o.p1; a.goff = o.p2;
            if (!a.w || ti.level != to.level) return fail(c, CY_ERR_STATE, "malformed depth-wise op");
            HIPCHK(c, launch_dwconv(c->prec, a, s));
            prof_done(CONV_NUM_VARIANTS + 2, 2.0 * Bn * a.H * a.W * (double)a.C * 9.0);
        } else if (o.kind == OPK_ATTN) {
            const Tensor& ti = p.tensors[o.in0]; const Tensor& to = p.tensors[o.out];
            AttnArgs a{};
            a.qkv = tp(o.in0); a.ct = cm * ti.C; a.coff = o.in0_coff; a.out = tp(o.out); a.out_ct = cm * to.C; a.out_coff = o.out_coff;
            a.lo = ti.C; a.out_lo = to.C;
            a.B = Bn; a.N = (H >> ti.level) * (W >> ti.level); a.heads = o.p0; a.kd = o.p1; a.hd = o.p2;
            a.scale = 1.0f / sqrtf((float)a.kd);
            if (a.heads < 1 || ti.level != to.level) return fail(c, CY_ERR_STATE, "malformed attention op");
            hipError_t e = launch_attention(c->prec, a, s);
            if (e == hipErrorInvalidValue) return fail(c, CY_ERR_UNSUPPORTED, "attention map too large for this kernel (stride-32 map of the input must have <= 10240 pixels)");
            HIPCHK(c, e);
            prof_done(CONV_NUM_VARIANTS + 3, 2.0 * Bn * a.heads * (double)a.N * a.N * (a.kd + a.hd));
        } else {
            const ConvDesc& d = p.convs[o.conv];
            ConvArgs a{};
            const Tensor& t0 = p.tensors[o.in0];
            auto span = [&](int t) -> uint32_t {           // bytes addressable from tp(t) for Bn images
                const Tensor& tt = p.tensors[t];
                return (uint32_t)((size_t)Bn * (H >> tt.level) * (W >> tt.level) * tt.C * es);
            };
            a.in0 = tp(o.in0); a.in0_ct = cm * t0.C; a.in0_coff = o.in0_coff; a.c0 = o.c0; a.up0 = o.up0;
            a.in0_bytes = span(o.in0);
            a.split = x3 ? c->dconv[o.conv].passes : 0; a.in0_lo = t0.C; a.oscale = c->dconv[o.conv].oscale;
            int lev_in = o.up0 ? t0.level - 1 : t0.level;
            if (o.in1 >= 0) {
                const Tensor& t1 = p.tensors[o.in1];
                a.in1 = tp(o.in1); a.in1_ct = cm * t1.C; a.in1_coff = o.in1_coff; a.c1 = o.c1; a.in1_bytes = span(o.in1); a.in1_lo = t1.C;
                lev_in = t1.level;
            }
            a.wgt = c->dconv[o.conv].w; a.wgt_bytes = (uint32_t)c->dconv[o.conv].wbytes; a.bias = c->dconv[o.conv].bias;
            a.wgt32 = c->dconv[o.conv].w32; a.wgt32_bytes = (uint32_t)c->dconv[o.conv].w32bytes;
            a.B = Bn; a.Hi = H >> lev_in; a.Wi = W >> lev_in; a.k = d.k; a.s = d.s; a.act = d.act;
            a.Ho = d.s == 2 ? a.Hi / 2 : a.Hi; a.Wo = d.s == 2 ? a.Wi / 2 : a.Wi;
            a.Cin = d.cin; a.Cout = d.cout;
            if (o.out >= 0) {
                const Tensor& to = p.tensors[o.out];
                a.out = tp(o.out); a.out_ct = cm * to.C; a.out_coff = o.out_coff; a.out_bs = a.Ho * a.Wo; a.out_ro = 0; a.out_f32 = 0; a.out_lo = to.C;
            } else {
                a.out = d_pred; a.out_ct = 64 + p.nc; a.out_coff = o.pred_coff; a.out_bs = A; a.out_ro = a_off[o.pred_level]; a.out_f32 = 1;
            }
            if (o.res >= 0) { a.res = tp(o.res); a.res_ct = cm * p.tensors[o.res].C; a.res_coff = o.res_coff; a.res_lo = p.tensors[o.res].C; }
            double flops = 2.0 * Bn * a.Ho * a.Wo * (double)a.Cout * a.Cin * a.k * a.k;
            if (pw_env && c->prec == PREC_F16 && next && next->conv >= 0 && c->dconv[next->conv].w2f && conv_variant(c->prec, a) == CONV_DIRECT_256 &&
                pw_pair(p, (size_t)(&o - p.ops.data()))) {
                const Op& n = *next;
                const ConvDesc& d2 = p.convs[n.conv];
                const Tensor& to2 = p.tensors[n.out];
                a.wgt2 = c->dconv[n.conv].w2f; a.wgt2_bytes = (uint32_t)c->dconv[n.conv].w2fbytes; a.bias2 = c->dconv[n.conv].bias;
                a.act2 = d2.act; a.cout2 = d2.cout;
                a.out2 = tp(n.out); a.out2_ct = to2.C; a.out2_coff = n.out_coff;
                flops += 2.0 * Bn * a.Ho * a.Wo * (double)d2.cout * d2.cin;
                *fused = true; c->pw_fused_conv = o.conv;
            }
            // the two output convolutions of a detect-head level (box branch at column 0, class branch at column 64 of the same
            // prediction rows) as ONE launch: the box branch's launch is deferred until the class branch's arguments are known
            // (nothing else reads the prediction rows inside the forward pass); CY_HEAD_PAIR=0 / other shapes: one launch each
            if (o.out < 0 && o.pred_coff == 0 && c->prec != PREC_F32 && env_knob("CY_HEAD_PAIR", 1) && conv_variant(c->prec, a) == CONV_HEAD_1X1) {
                HeadPend& h = pend[o.pred_level];
                if (h.on) { cur_conv = h.conv; HIPCHK(c, launch_conv(c->prec, h.a, s)); prof_done(conv_variant(c->prec, h.a), h.flops); cur_conv = o.conv; }
                h.a = a; h.flops = flops; h.conv = o.conv; h.on = true;
                return CY_OK;
            }
            if (o.out < 0 && o.pred_coff == 64 && pend[o.pred_level].on) {
                HeadPend& h = pend[o.pred_level];
                h.on = false;
                if (head_pair_ok(c->prec, h.a, a)) {
                    HIPCHK(c, launch_head_pair(c->prec, h.a, a, s));
                    prof_done(CONV_HEAD_1X1, flops + h.flops);
                    return CY_OK;
                }
                cur_conv = h.conv; HIPCHK(c, launch_conv(c->prec, h.a, s)); prof_done(conv_variant(c->prec, h.a), h.flops); cur_conv = o.conv;
            }
            HIPCHK(c, launch_conv(c->prec, a, s));
            prof_done(conv_variant(c->prec, a), flops);
        }
        return CY_OK;
    };
    auto run_range = [&](size_t lo, size_t hi, int b0, int Bn) -> int {
        for (size_t i = lo; i < hi; ++i) {
            bool fused = false;
            const int r = run_op(p.ops[i], i + 1 < hi ? &p.ops[i + 1] : nullptr, b0, Bn, &fused);
            if (r) return r;
            if (fused) ++i;                                  // the next op ran inside this one's kernel
        }
        return CY_OK;
    };
    for (int b0 = 0; b0 < B; b0 += sub) {
        const int Bn = B - b0 < sub ? B - b0 : sub;
        rc = run_range(0, n_head, b0, Bn); if (rc) return rc;
    }
    rc = run_range(n_head, p.ops.size(), 0, B); if (rc) return rc;
    for (HeadPend& h : pend)                                  // a box branch whose class branch never came (no such plan today)
        if (h.on) { h.on = false; cur_conv = h.conv; HIPCHK(c, launch_conv(c->prec, h.a, s)); prof_done(conv_variant(c->prec, h.a), h.flops); }
    return CY_OK;
}

int cy_profile_enable(cy_ctx* c, int on) {
    if (!c) return CY_ERR_ARG;
    c->profiling = on != 0;
    c->prof_stride = on > 1 ? on : 1;                    // on = N > 1: time every N-th forward call only (an event per launch costs ~3 % at N = 1)
    c->fwd_calls = 0;
    c->prof.clear(); c->ev_used = 0;
    return CY_OK;
}

static int profile_summary_lane(cy_ctx* c, cy_prof_entry* out, int cap, int lane);
int cy_profile_summary(cy_ctx* c, cy_prof_entry* out, int cap) { return profile_summary_lane(c, out, cap, -1); }
int cy_profile_summary_lane(cy_ctx* c, cy_prof_entry* out, int cap, int lane) { return profile_summary_lane(c, out, cap, lane); }

static int profile_summary_lane(cy_ctx* c, cy_prof_entry* out, int cap, int lane) {
    // one entry per forward kernel variant (conv variants in ConvVariant order, then stem, pool)
    const int n = CONV_NUM_VARIANTS + 5;
    if (!c || !out || cap < n) return fail(c, CY_ERR_ARG, "bad arguments");
    HIPCHK(c, hipDeviceSynchronize());
    for (int k = 0; k < n; ++k) {
        memset(&out[k], 0, sizeof(out[k]));
        static const char* const extra[5] = {"stem_kernel", "pool5_kernel", "dwconv3x3_kernel", "attention_kernel",
                                             "bneck64_kernel fused 3x3+3x3 64ch bottleneck, weights in registers"};
        const char* nm = k < CONV_NUM_VARIANTS ? conv_variant_name(k) : extra[k - CONV_NUM_VARIANTS];
        strncpy(out[k].kernel, nm, sizeof(out[k].kernel) - 1);
    }
    for (const auto& r : c->prof) {
        if (lane >= 0 && r.lane != lane) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[r.e0], c->ev_pool[r.e1]) != hipSuccess) continue;
        out[r.kind].ms += ms; out[r.kind].flops += r.flops; out[r.kind].launches += 1;
    }
    return n;
}

int cy_profile_layers(cy_ctx* c, cy_prof_entry* out, int cap) {
    if (!c || !c->loaded || !out) return fail(c, CY_ERR_ARG, "bad arguments");
    const int n = (int)c->plan.convs.size();
    if (cap < n) return fail(c, CY_ERR_ARG, "need one entry per conv");
    HIPCHK(c, hipDeviceSynchronize());
    for (int i = 0; i < n; ++i) { memset(&out[i], 0, sizeof(out[i])); strncpy(out[i].kernel, c->plan.convs[i].name.c_str(), sizeof(out[i].kernel) - 1); }
    for (const auto& r : c->prof) {
        if (r.conv < 0) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[r.e0], c->ev_pool[r.e1]) != hipSuccess) continue;
        out[r.conv].ms += ms; out[r.conv].flops += r.flops; out[r.conv].launches += 1;
    }
    return n;
}

int cy_debug_read_conv(cy_ctx* c, const char* conv_name, float* h_out, size_t cap, int* dims4) {
    if (!c || !c->loaded || !conv_name || !h_out) return fail(c, CY_ERR_ARG, "bad arguments");
    if (c->lastB == 0) return fail(c, CY_ERR_STATE, "no forward has run");
    if (c->split_last) return fail(c, CY_ERR_STATE, "the last batch ran as two half-batches (CY_DUAL_FORWARD=0 keeps it in one workspace)");
    const Plan& p = c->plan;
    for (const Op& o : p.ops) {
        if (o.conv < 0 || p.convs[o.conv].name != conv_name) continue;      // pool and attention ops carry no convolution
        if (o.out < 0) return fail(c, CY_ERR_UNSUPPORTED, "head outputs are read from d_pred");
        if (c->bneck_fused_last && c->dconv[o.conv].bneck && env_knob("CY_BNECK_FUSE", 0))
            return fail(c, CY_ERR_STATE, "this bottleneck ran fused: its cv1 output was not materialised; unset CY_BNECK_FUSE");
        if (o.conv == c->pw_fused_conv && c->pw_fused_conv >= 0 && env_knob("CY_FUSE_PW", 1))
            return fail(c, CY_ERR_STATE, "this layer ran fused with the 1x1 convolution behind it: its output was not materialised; CY_FUSE_PW=0 runs the two layers apart");
        if (o.kind == OPK_STEM && c->stem_fused_last)
            return fail(c, CY_ERR_STATE, "the stem output was not materialised (fused into the next layer's kernel); set CY_STEM_FUSE=0");
        const Tensor& t = p.tensors[o.out];
        const int Ho = c->lastH >> t.level, Wo = c->lastW >> t.level, C = p.convs[o.conv].cout, B = c->lastB;
        const size_t n = (size_t)B * C * Ho * Wo;
        if (n > cap) return fail(c, CY_ERR_ARG, "output buffer too small");
        HIPCHK(c, hipDeviceSynchronize());
        std::vector<char> host(c->tbytes[o.out]);
        HIPCHK(c, hipMemcpy(host.data(), c->ws + c->toff[o.out], host.size(), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) for (int h = 0; h < Ho; ++h) for (int w = 0; w < Wo; ++w) for (int ch = 0; ch < C; ++ch) {
            const size_t src = (((size_t)b * Ho + h) * Wo + w) * t.C + o.out_coff + ch;
            float v;
            if (c->prec == PREC_F16X3) {
                const _Float16* hp = reinterpret_cast<_Float16*>(host.data()) + (((size_t)b * Ho + h) * Wo + w) * 2 * t.C + o.out_coff + ch;
                v = (float)hp[0] + (float)hp[t.C];
            } else v = c->prec == PREC_F16 ? (float)reinterpret_cast<_Float16*>(host.data())[src]
                                           : reinterpret_cast<float*>(host.data())[src];
            h_out[(((size_t)b * C + ch) * Ho + h) * Wo + w] = v;
        }
        if (dims4) { dims4[0] = B; dims4[1] = C; dims4[2] = Ho; dims4[3] = Wo; }
        return CY_OK;
    }
    return fail(c, CY_ERR_ARG, std::string("no such conv: ") + conv_name);
}

int cy_conv_bn_silu(cy_ctx* c, const void* d_in, int B, int Hi, int Wi, int Cin, const float* h_w, const float* h_b,
                    int Cout, int k, int s, int act, const void* d_res, void* d_out, void* stream) {
    if (!c || !d_in || !h_w || !h_b || !d_out) return fail(c, CY_ERR_ARG, "null argument");
    if ((k != 1 && k != 3) || (s != 1 && s != 2) || Cin % 8 || B < 1) return fail(c, CY_ERR_ARG, "unsupported conv geometry");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->prec == PREC_F16X3) {
        // fp16x3 context: caller tensors are fp32 NHWC; they are split into high / low halves here, the layer runs on the split
        // kernels exactly as inside the forward pass, and the result is merged back to fp32
        hipStream_t st = (hipStream_t)stream;
        const int pad = k / 2, Ho = (Hi + 2 * pad - k) / s + 1, Wo = (Wi + 2 * pad - k) / s + 1, cp = (Cout + 127) / 128 * 128;
        const size_t nin = (size_t)B * Hi * Wi, nout = (size_t)B * Ho * Wo;
        const int passes = env_knob("CY_X3_PASSES", 0) == 3 ? 3 : x3_passes(h_w, Cout, Cin, k, nullptr);     // 2 when every weight is an fp16 value
        const size_t wb = packed_weight_bytes_x3(Cout, Cin, k, 128, passes), wb32 = (k == 3 && s == 1 && Cin % 64 == 0) ? packed_weight_bytes_x3(Cout, Cin, k, 64, passes) : 0;
        std::vector<char> packed(wb), p32(wb32);
        std::vector<float> osc(cp), bias(cp, 0.0f);
        pack_weights_x3(h_w, Cout, Cin, k, packed.data(), osc.data(), 128, passes);
        if (wb32) pack_weights_x3(h_w, Cout, Cin, k, p32.data(), osc.data(), 64, passes);
        memcpy(bias.data(), h_b, 4 * Cout);
        void *dw = nullptr, *dw32 = nullptr, *xin = nullptr, *xres = nullptr, *xout = nullptr; float *db = nullptr, *dsc = nullptr;
        auto cleanup = [&]() { for (void* q : {dw, dw32, xin, xres, xout, (void*)db, (void*)dsc}) if (q) hipFree(q); };
        hipError_t e = hipMalloc(&dw, wb);
        if (e == hipSuccess && wb32) e = hipMalloc(&dw32, wb32);
        if (e == hipSuccess) e = hipMalloc(&db, 4 * cp);
        if (e == hipSuccess) e = hipMalloc(&dsc, 4 * cp);
        if (e == hipSuccess) e = hipMalloc(&xin, nin * Cin * 4);
        if (e == hipSuccess) e = hipMalloc(&xout, nout * Cout * 4);
        if (e == hipSuccess && d_res) e = hipMalloc(&xres, nout * Cout * 4);
        if (e == hipSuccess) e = hipMemcpy(dw, packed.data(), wb, hipMemcpyHostToDevice);
        if (e == hipSuccess && wb32) e = hipMemcpy(dw32, p32.data(), wb32, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(db, bias.data(), 4 * cp, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dsc, osc.data(), 4 * cp, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_x3_split(reinterpret_cast<const float*>(d_in), xin, (long)nin, Cin, st);
        if (e == hipSuccess && d_res) e = launch_x3_split(reinterpret_cast<const float*>(d_res), xres, (long)nout, Cout, st);
        if (e == hipSuccess) {
            ConvArgs a{};
            a.in0 = xin; a.in0_ct = 2 * Cin; a.c0 = Cin; a.in0_bytes = (uint32_t)(nin * Cin * 4); a.in0_lo = Cin;
            a.wgt = dw; a.wgt_bytes = (uint32_t)wb; a.bias = db; a.wgt32 = dw32; a.wgt32_bytes = (uint32_t)wb32; a.oscale = dsc; a.split = passes;
            a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.Cin = Cin; a.Cout = Cout; a.k = k; a.s = s; a.act = act;
            a.out = xout; a.out_ct = 2 * Cout; a.out_lo = Cout; a.out_bs = Ho * Wo;
            if (d_res) { a.res = xres; a.res_ct = 2 * Cout; a.res_lo = Cout; }
            e = launch_conv(c->prec, a, st);
        }
        if (e == hipSuccess) e = launch_x3_merge(xout, reinterpret_cast<float*>(d_out), (long)nout, Cout, st);
        const hipError_t e2 = hipStreamSynchronize(st);
        cleanup();
        if (e != hipSuccess) return fail(c, CY_ERR_HIP, hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(c, CY_ERR_HIP, hipGetErrorString(e2));
        return CY_OK;
    }
    const size_t es = esize(c->prec);
    const size_t wb = packed_weight_bytes(c->prec, Cout, Cin, k);
    std::vector<char> packed(wb);
    pack_weights(c->prec, h_w, Cout, Cin, k, packed.data());
    const int cp = (Cout + 127) / 128 * 128;
    std::vector<float> bias(cp, 0.0f);
    memcpy(bias.data(), h_b, 4 * Cout);
    void* dw = nullptr; float* db = nullptr;
    HIPCHK(c, hipMalloc(&dw, wb));
    HIPCHK(c, hipMalloc(&db, 4 * cp));
    HIPCHK(c, hipMemcpy(dw, packed.data(), wb, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(db, bias.data(), 4 * cp, hipMemcpyHostToDevice));
    void* dw32 = nullptr; size_t wb32 = 0;
    if (c->prec == PREC_F16 && k == 3 && s == 1 && ((Cin % 64 == 0 && Cout >= 33) || (Cin == 32 && Cout <= 64))) {
        wb32 = packed_weight_bytes(c->prec, Cout, Cin, k, 64);
        std::vector<char> p32(wb32);
        pack_weights(c->prec, h_w, Cout, Cin, k, p32.data(), 64);
        HIPCHK(c, hipMalloc(&dw32, wb32));
        HIPCHK(c, hipMemcpy(dw32, p32.data(), wb32, hipMemcpyHostToDevice));
    }
    ConvArgs a{};
    const int pad = k / 2;
    a.in0 = d_in; a.in0_ct = Cin; a.c0 = Cin; a.in0_bytes = (uint32_t)((size_t)B * Hi * Wi * Cin * es);
    a.wgt = dw; a.wgt_bytes = (uint32_t)wb; a.bias = db; a.wgt32 = dw32; a.wgt32_bytes = (uint32_t)wb32;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = (Hi + 2 * pad - k) / s + 1; a.Wo = (Wi + 2 * pad - k) / s + 1;
    a.Cin = Cin; a.Cout = Cout; a.k = k; a.s = s; a.act = act;
    a.out = d_out; a.out_ct = Cout; a.out_bs = a.Ho * a.Wo;
    if (d_res) { a.res = d_res; a.res_ct = Cout; }
    hipError_t e = launch_conv(c->prec, a, (hipStream_t)stream);
    hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
    hipFree(dw); hipFree(db); if (dw32) hipFree(dw32);
    if (e != hipSuccess) return fail(c, CY_ERR_HIP, hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(c, CY_ERR_HIP, hipGetErrorString(e2));
    return CY_OK;
}

int cy_bottleneck64(cy_ctx* c, const void* d_in, int B, int H, int W, const float* h_w1, const float* h_b1, const float* h_w2,
                    const float* h_b2, int shortcut, void* d_out, void* stream) {
    if (!c || !d_in || !h_w1 || !h_b1 || !h_w2 || !h_b2 || !d_out || B < 1 || H < 1 || W < 1) return fail(c, CY_ERR_ARG, "bad arguments");
    if (c->prec != PREC_F16) return fail(c, CY_ERR_UNSUPPORTED, "the fused bottleneck kernel exists in the fp16 context only");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<char> wf(BNECK_WFRAG_BYTES);
    pack_bneck_weights(h_w1, h_w2, wf.data());
    void* dw = nullptr; float* db = nullptr;
    HIPCHK(c, hipMalloc(&dw, wf.size()));
    HIPCHK(c, hipMalloc(&db, 128 * sizeof(float)));
    HIPCHK(c, hipMemcpy(dw, wf.data(), wf.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(db, h_b1, 64 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(db + 64, h_b2, 64 * sizeof(float), hipMemcpyHostToDevice));
    BneckArgs a{};
    a.in = d_in; a.in_ct = 64; a.in_coff = 0; a.in_bytes = (uint32_t)((size_t)B * H * W * 64 * 2);
    a.out = d_out; a.out_ct = 64; a.out_coff = 0; a.wfrag = dw; a.bias1 = db; a.bias2 = db + 64;
    a.B = B; a.H = H; a.W = W; a.shortcut = shortcut != 0;
    hipError_t e = launch_bneck64(a, (hipStream_t)stream);
    hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
    hipFree(dw); hipFree(db);
    if (e != hipSuccess) return fail(c, CY_ERR_HIP, hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(c, CY_ERR_HIP, hipGetErrorString(e2));
    return CY_OK;
}

int cy_debug_stamps(unsigned long long* out8, int reset) {
    if (!out8) return CY_ERR_ARG;
    if (reset < 0) { debug_read_wg_stamps(out8, -reset); return CY_OK; }      // raw per-workgroup records: out8 holds (-reset) x 4 values
    if (reset >= 2) debug_read_pre_stamps(out8, reset == 3);       // 2 / 3: the statistics kernel's phase stamps (read / read + reset)
    else debug_read_stamps(out8, reset != 0);
    return CY_OK;
}

int cy_debug_fastdiv(const double* d_a, const double* d_b, double* d_fast, double* d_ref, int n) {
    if (!d_a || !d_b || !d_fast || !d_ref || n < 1) return CY_ERR_ARG;
    debug_fastdiv(d_a, d_b, d_fast, d_ref, n);
    return hipGetLastError() == hipSuccess ? CY_OK : CY_ERR_HIP;
}

int cy_debug_cand_counts(cy_ctx* c, int* h_out, int B) {
    if (!c || !c->loaded || !h_out || B < 1 || B > c->cfg.max_batch) return fail(c, CY_ERR_ARG, "bad arguments");
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(h_out, c->S().cand_count, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
    return CY_OK;
}

int cy_preproc_params(cy_ctx* c, double* h_out, int B) {
    if (!c || !c->loaded || !h_out || B < 1 || B > c->cfg.max_batch) return fail(c, CY_ERR_ARG, "bad arguments");
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(h_out, c->S().pre_params, (size_t)B * 3 * CY_MAX_STAGES * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return CY_OK;
}

static int fill_pre_args(cy_ctx* c, const float* d_mosaic, int MH, int MW, const int* h_tiles, int B, int th, int tw,
                         const cy_preproc_cfg* cfg, int* d_status, PreArgs& a) {
    if (cfg->nprog != 0 && cfg->nprog != 1 && cfg->nprog != 3) return fail(c, CY_ERR_ARG, "nprog must be 0, 1 or 3");
    if (B > MAX_PRE_BATCH) return fail(c, CY_ERR_ARG, "at most 256 tiles per preprocessing call");
    for (int b = 0; b < B; ++b) {
        a.txy[2 * b] = h_tiles[2 * b]; a.txy[2 * b + 1] = h_tiles[2 * b + 1];
        if (h_tiles[2 * b] < 0 || h_tiles[2 * b + 1] < 0 || h_tiles[2 * b] + tw > MW || h_tiles[2 * b + 1] + th > MH)
            return fail(c, CY_ERR_ARG, "tile outside the mosaic");
    }
    // the statistics passes address a tile through a raw buffer resource with 32-bit byte offsets
    if (((long)(th - 1) * MW + tw) * 4L > 0xFFFFFF00L)
        return fail(c, CY_ERR_UNSUPPORTED, "tile rows span more than 4 GiB of the mosaic: crop the image or run it in tiles");
    a.mosaic = d_mosaic; a.MH = MH; a.MW = MW; a.B = B; a.th = th; a.tw = tw;
    a.nprog = cfg->nprog;
    for (int i = 0; i < 3; ++i) {
        a.prog[i].n = cfg->prog[i].n;
        if (a.prog[i].n < 0 || a.prog[i].n > MAX_STAGES) return fail(c, CY_ERR_ARG, "too many stages");
        bool reversed = false;               // a decreasing map so far (MINMAX with norm_max < norm_min)
        for (int j = 0; j < a.prog[i].n; ++j) {
            const cy_pre_stage& st = cfg->prog[i].st[j];
            if (st.op < CY_OP_BKG || st.op > CY_OP_MINMAX) return fail(c, CY_ERR_ARG, "unknown preprocessing op");
            // the statistics kernel selects medians by the order of the RAW pixels, which needs every earlier map to be
            // non-decreasing (cy_preproc.hip); the CLI order (scripts/run.py:272-302) puts MINMAX last, so this never fires there
            if (reversed && (st.op == CY_OP_BKG || st.op == CY_OP_SHIFT || st.op == CY_OP_CLIP))
                return fail(c, CY_ERR_UNSUPPORTED, "a sigma-clip stage after a MINMAX stage with norm_max < norm_min is not supported");
            if (st.op == CY_OP_MINMAX && st.p1 < st.p0) reversed = !reversed;
            a.prog[i].st[j] = PreStage{st.op, st.p0, st.p1, st.p2, st.flag};
        }
    }
    a.params = c->S().pre_params; a.histeq = c->S().pre_histeq; a.status = d_status; a.counters = c->counters;
    return CY_OK;
}

int cy_preproc(cy_ctx* c, const float* d_mosaic, int MH, int MW, const int* h_tiles, int B, int th, int tw, int imgsz,
               const cy_preproc_cfg* cfg, void* d_netin, int* d_status, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    if (!d_mosaic || !h_tiles || !cfg || !d_netin || !d_status || B < 1 || B > c->cfg.max_batch)
        return fail(c, CY_ERR_ARG, "bad preproc arguments");
    cy_letterbox lb;
    if (cy_letterbox_geometry(th, tw, imgsz, &lb)) return fail(c, CY_ERR_ARG, "bad tile/imgsz");
    if (lb.H > c->cfg.max_h || lb.W > c->cfg.max_w) return fail(c, CY_ERR_ARG, "letterboxed tile exceeds max_h/max_w of the context");
    PreArgs a{};
    int rc = fill_pre_args(c, d_mosaic, MH, MW, h_tiles, B, th, tw, cfg, d_status, a);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    a.out = d_netin; a.out_prec = io_prec(c->prec); a.H = lb.H; a.W = lb.W; a.top = lb.top; a.left = lb.left;
    a.new_h = lb.new_h; a.new_w = lb.new_w;
    const bool resize = (lb.new_h != th) || (lb.new_w != tw);
    if (resize && (size_t)B * 3 * th * tw > c->pre_scratch_elems) return fail(c, CY_ERR_ARG, "resize scratch too small");
    a.scratch = resize ? c->S().pre_scratch : nullptr;
    HIPCHK(c, launch_preproc(a, s));
    return CY_OK;
}

int cy_preproc_planes(cy_ctx* c, const float* d_mosaic, int MH, int MW, const int* h_tiles, int B, int th, int tw,
                      const cy_preproc_cfg* cfg, double* d_planes, int* d_status, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    if (!d_mosaic || !h_tiles || !cfg || !d_planes || !d_status || B < 1 || B > c->cfg.max_batch || th < 1 || tw < 1)
        return fail(c, CY_ERR_ARG, "bad preproc arguments");
    PreArgs a{};
    int rc = fill_pre_args(c, d_mosaic, MH, MW, h_tiles, B, th, tw, cfg, d_status, a);
    if (rc) return rc;
    a.scratch = d_planes;
    HIPCHK(c, launch_preproc_planes(a, (hipStream_t)stream));
    return CY_OK;
}

int cy_letterbox_pack(cy_ctx* c, const double* d_planes, int B, int h0, int w0, int imgsz, void* d_netin, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    if (!d_planes || !d_netin || B < 1 || B > c->cfg.max_batch) return fail(c, CY_ERR_ARG, "bad arguments");
    cy_letterbox lb;
    if (cy_letterbox_geometry(h0, w0, imgsz, &lb)) return fail(c, CY_ERR_ARG, "bad image/imgsz");
    if (lb.H > c->cfg.max_h || lb.W > c->cfg.max_w) return fail(c, CY_ERR_ARG, "letterboxed image exceeds max_h/max_w of the context");
    PreArgs a{};
    a.B = B; a.th = h0; a.tw = w0; a.scratch = const_cast<double*>(d_planes);
    a.out = d_netin; a.out_prec = io_prec(c->prec); a.H = lb.H; a.W = lb.W; a.top = lb.top; a.left = lb.left;
    a.new_h = lb.new_h; a.new_w = lb.new_w;
    HIPCHK(c, launch_letterbox_pack(a, (hipStream_t)stream));
    return CY_OK;
}

int cy_decode_nms(cy_ctx* c, const float* d_pred, int B, int H, int W, int h0, int w0, float conf, float iou,
                  float* d_det, int* d_det_anchor, int* d_count, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    if (!d_pred || !d_det || !d_det_anchor || !d_count || B < 1 || B > c->cfg.max_batch) return fail(c, CY_ERR_ARG, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DecodeArgs d{};
    d.pred = d_pred; d.B = B; d.A = cy_num_anchors(H, W); d.nc = c->plan.nc; d.conf = conf;
    for (int l = 0; l < 3; ++l) { d.lvl_h[l] = H >> (3 + l); d.lvl_w[l] = W >> (3 + l); }
    d.cand = c->S().cand; d.cand_anchor = c->S().cand_anchor; d.cand_count = c->S().cand_count; d.cap = c->cap;
    HIPCHK(c, hipMemsetAsync(c->S().cand_count, 0, B * sizeof(int), s));
    HIPCHK(c, launch_decode(d, s));
    NmsArgs n{};
    n.cand = c->S().cand; n.cand_anchor = c->S().cand_anchor; n.cand_count = c->S().cand_count; n.cap = c->cap; n.B = B; n.iou = iou;
    n.max_det = CY_MAX_DET;
    // ultralytics scale_boxes: gain / pad from the letterboxed and original shapes
    const double gain = std::fmin((double)H / h0, (double)W / w0);
    n.gain = (float)gain;
    n.padw = py_round_half_even((W - w0 * gain) / 2 - 0.1); n.padh = py_round_half_even((H - h0 * gain) / 2 - 0.1);
    n.w0 = w0; n.h0 = h0;
    n.det = d_det; n.det_anchor = d_det_anchor; n.det_count = d_count; n.keys = c->S().keys; n.mask = nullptr;
    n.counters = c->counters;
    HIPCHK(c, launch_nms(n, s));
    return CY_OK;
}

int cy_iou_merge(cy_ctx* c, const float* d_det, const int* d_count, int B, float score_thr, double soft, double hard,
                 float* d_out, int* d_out_count, int* d_out_src, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    if (!d_det || !d_count || !d_out || !d_out_count || B < 1 || B > c->cfg.max_batch) return fail(c, CY_ERR_ARG, "bad arguments");
    MergeArgs m{};
    m.det = d_det; m.det_count = d_count; m.B = B; m.max_det = CY_MAX_DET; m.score_thr = score_thr; m.soft = soft; m.hard = hard;
    m.out = d_out; m.out_count = d_out_count; m.out_src = d_out_src ? d_out_src : c->S().out_src; m.err = c->S().merge_err;
    m.counters = c->counters;
    HIPCHK(c, launch_iou_merge(m, (hipStream_t)stream));
    return CY_OK;
}

int cy_detect_tiles(cy_ctx* c, const float* d_mosaic, int MH, int MW, const int* h_tiles, int B, int th, int tw, int imgsz,
                    const cy_preproc_cfg* cfg, float conf, float iou, double soft, double hard,
                    float* d_out, int* d_out_count, int* d_status, void* stream) {
    // Software pipeline over consecutive calls (batches): three streams, two buffer sets (+ the small-batch lane below).
    //   s_pre  : preprocessing of batch i      (waits: forward of batch i-2 has consumed this set's network input)
    //   stream : forward of batch i            (waits: preprocessing i; post-processing i-2 has consumed this set's head output)
    //   s_post : decode/NMS/IoU-merge of batch i (waits: forward i)
    // so the latency-bound statistics / NMS / merge kernels (a few dozen workgroups) run beside the conv stack of the
    // neighbouring batches instead of serialising with it.  Outputs are complete after cy_detect_flush(ctx, stream).
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    cy_letterbox lb;
    if (cy_letterbox_geometry(th, tw, imgsz, &lb)) return fail(c, CY_ERR_ARG, "bad tile/imgsz");
    hipStream_t sm = (hipStream_t)stream;
    // Small batches (the ragged edge classes of a grid: 39 + 39 + 1 of the 1600 tiles of the 16k mosaic) take their own lane:
    // buffer set 2, forward on the second stream / workspace.  A batch of 1-39 tiles is a chain of ~105 kernels of a few
    // workgroups each -- 4.2 ms for ONE tile, 6.2 ms for 39, i.e. 8 % of a pass for 4 % of its pixels when run in line -- but
    // hardly any work: beside the full batches of the main lane it costs about its share of the chip.  The caller
    // interleaves them with the full batches (TileEngine).  CY_SMALL_LANE=0 keeps every batch on the main lane.
    // "small" is relative to the context: at most a third of max_batch (and < 64 tiles).  With the CLI default --tile_batch 64
    // the equal-sized batches of 50-63 tiles stay on the main lane with its two buffer sets (the lane has ONE set: consecutive
    // small batches serialise preprocessing behind forward).  CY_SMALL_LANE is read per call (the tests compare both settings).
    const bool small = env_knob("CY_SMALL_LANE", 1) && c->prec != PREC_F32 && B < 64 && (long)B * 3 <= c->cfg.max_batch;
    int rc = CY_OK;
    if (small) { rc = ensure_small_lane(c); if (rc) return rc; }
    const int sl = small ? 2 : (int)(c->batches & 1);
    const bool reuse = small ? c->small_batches >= 1 : c->batches >= 2;
    hipStream_t sf = small ? c->s_small : sm;
    // Stream budget.  The HIP runtime maps streams onto four hardware queues; two BUSY streams on one queue serialise (DESIGN.md
    // section 5).  A context owns four: the caller's, preprocessing, post-processing and the second forward stream (small-batch lane /
    // second half of a split batch).  A collective library in the process (RCCL: one stream per communicator) is a fifth, and even idle
    // it displaces one of ours onto a shared queue: measured at N = 1 with a one-rank RCCL group, 176.7 -> 187.7 ms per pass.
    // CY_SIDE_STREAMS (read per call; TileEngine sets 3 when a process group exists): 2 = four streams (default); 1 = post-processing
    // shares the preprocessing stream (the NMS of batch i-1 then delays the statistics of batch i+1: 191.7 ms); 3 = post-processing
    // shares the SECOND FORWARD stream (decode / NMS / merge are ~1 ms per batch next to a small-batch forward that runs beside the
    // main lane anyway), i.e. three streams + the library's = four queues.
    const int side_mode = env_knob("CY_SIDE_STREAMS", 2);
    if (side_mode == 3) { rc = ensure_side_forward_stream(c); if (rc) return rc; }
    hipStream_t spost = side_mode == 1 ? c->s_pre : (side_mode == 3 ? c->s_fwd2 : c->s_post);
    c->slot = sl;
    // order the side streams after whatever the caller already queued on `stream` -- only where that matters: the first batch
    // after load / flush, or after cy_mosaic_prepare.  Later batches are ordered by ev_fwd / ev_post alone, so that the
    // preprocessing of batch i does not wait for the forward of batch i-1 that is already queued on `stream`.
    // A mosaic buffer this pipeline has not seen yet may still be the target of an upload queued on `stream`: order behind it too.
    bool seen = false;
    for (int i = 0; i < c->n_seen; ++i) seen = seen || c->seen_mosaic[i] == (const void*)d_mosaic;
    if (c->batches + c->small_batches == 0 || c->mosaic_dirty || !seen) {
        HIPCHK(c, hipEventRecord(c->ev_call, sm));
        HIPCHK(c, hipStreamWaitEvent(c->s_pre, c->ev_call, 0));
        if (c->mosaic_dirty || c->batches + c->small_batches == 0) c->n_seen = 0;
        if (c->n_seen < 16) c->seen_mosaic[c->n_seen++] = d_mosaic;        // (a 17th distinct buffer is simply ordered again every time)
        c->mosaic_dirty = false;
    }
    if (reuse) HIPCHK(c, hipStreamWaitEvent(c->s_pre, c->ev_fwd[sl], 0));
    rc = cy_preproc(c, d_mosaic, MH, MW, h_tiles, B, th, tw, imgsz, cfg, c->S().netin, d_status, c->s_pre);
    if (rc) { c->slot = 0; return rc; }
    HIPCHK(c, hipEventRecord(c->ev_pre[sl], c->s_pre));
    HIPCHK(c, hipStreamWaitEvent(sf, c->ev_pre[sl], 0));
    if (reuse) HIPCHK(c, hipStreamWaitEvent(sf, c->ev_post[sl], 0));
    if (small) {
        c->split_last = false;
        rc = forward_on(c, c->S().netin, B, lb.H, lb.W, c->S().pred, sf, c->ws3, c->ws3_bytes, true);      // (profiled like the main lane)
    } else {
        rc = forward_split(c, c->S().netin, B, lb.H, lb.W, c->S().pred, sm);
    }
    if (rc) { c->slot = 0; return rc; }
    HIPCHK(c, hipEventRecord(c->ev_fwd[sl], sf));
    HIPCHK(c, hipStreamWaitEvent(spost, c->ev_fwd[sl], 0));
    rc = cy_decode_nms(c, c->S().pred, B, lb.H, lb.W, th, tw, conf, iou, c->S().det, c->S().det_anchor, c->S().det_count, spost);
    if (!rc) rc = cy_iou_merge(c, c->S().det, c->S().det_count, B, conf, soft, hard, d_out, d_out_count, nullptr, spost);
    if (rc) { c->slot = 0; return rc; }
    HIPCHK(c, hipEventRecord(c->ev_post[sl], spost));
    if (small) c->small_batches++; else c->batches++;
    c->slot = 0;
    return CY_OK;
}

int cy_detect_fence(cy_ctx* c, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    (void)stream;
    c->mosaic_dirty = true;                  // the next cy_detect_tiles records an event on ITS stream and orders the side streams behind it
    return CY_OK;
}

int cy_detect_flush(cy_ctx* c, void* stream) {
    if (!c || !c->loaded) return fail(c, CY_ERR_STATE, "weights not loaded");
    hipStream_t sm = (hipStream_t)stream;
    const unsigned long n = c->batches < 2 ? c->batches : 2;
    for (unsigned long k = 0; k < n; ++k) {
        const int sl = (int)((c->batches - 1 - k) & 1);
        HIPCHK(c, hipStreamWaitEvent(sm, c->ev_post[sl], 0));
        HIPCHK(c, hipStreamWaitEvent(sm, c->ev_pre[sl], 0));
    }
    if (c->small_batches) {          // the small-batch lane (one buffer set: its batches are ordered among themselves)
        HIPCHK(c, hipStreamWaitEvent(sm, c->ev_post[2], 0));
        HIPCHK(c, hipStreamWaitEvent(sm, c->ev_pre[2], 0));
    }
    c->batches = 0; c->small_batches = 0; c->n_seen = 0;     // the next call starts a new pipeline: it orders the side streams behind `stream` again
    return CY_OK;
}

int cy_compact_records(const float* d_gathered, long long n_rows, const long long* d_perm, int T, int row_floats, int* d_hdr, float* d_out, void* stream) {
    if (!d_gathered || !d_perm || !d_hdr || !d_out || T < 1 || n_rows < 1 || row_floats != CY_MAX_DET * CY_DET_STRIDE + 3)
        return fail(nullptr, CY_ERR_ARG, "bad arguments");
    const hipError_t e = launch_compact_records(d_gathered, n_rows, d_perm, T, row_floats, d_hdr, d_out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, CY_ERR_HIP, hipGetErrorString(e));
    return CY_OK;
}

int cy_compact_records_ctx(cy_ctx* c, const float* d_gathered, long long n_rows, const long long* d_perm, int T, int row_floats, int* d_hdr, float* d_out, void* stream) {
    if (!c) return CY_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = cy_compact_records(d_gathered, n_rows, d_perm, T, row_floats, d_hdr, d_out, stream);
    if (rc != CY_OK) return fail(c, rc, "cy_compact_records failed (bad arguments or launch error)");
    return CY_OK;
}

int cy_detect_counters(cy_ctx* c, long long* out4, int reset) {
    if (!c || !c->loaded || !out4) return fail(c, CY_ERR_ARG, "bad arguments");
    int h[4] = {0, 0, 0, 0};
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out4[i] = h[i];
    if (reset) HIPCHK(c, hipMemset(c->counters, 0, sizeof(h)));
    return CY_OK;
}

}  // extern "C"
