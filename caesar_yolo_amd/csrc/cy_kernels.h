// Internal launch interfaces between the runtime (cy_context.cpp) and the gfx950 kernels.
// Nothing here is part of the C-ABI (see include/caesar_yolo_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace cy {

// Environment switches come in two classes.  env_knob(): selection among kernels / schedules that all compute the same
// result (CY_BATCH_INVARIANT, CY_DUAL_FORWARD, CY_STEM_FUSE, CY_WIDE_DUAL, CY_DIRECT_MIN_BLOCKS, ... -- the parity tests force
// every variant through them), always available.  dev_knob(): developer ablation switches that SKIP work and therefore
// invalidate results (CY_DBG, CY_SD_DBG): read only in a diagnostic build (-DCY_DEV_KNOBS=1, `CY_STAMPS=1 python
// __graft_entry__.py --force`); in the product binary they are the constant default.
#ifndef CY_DEV_KNOBS
#define CY_DEV_KNOBS 0
#endif
inline int env_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline int dev_knob(const char* name, int dflt) { return CY_DEV_KNOBS ? env_knob(name, dflt) : dflt; }

// PREC_F16X3 ("fp16x3", the fast parity context): every activation is stored as TWO fp16 values hi = fp16(x), lo = fp16(x - hi)
// (22 significand bits), weights likewise after a per-output-channel power-of-two scale that keeps their low halves out of the
// fp16 subnormal range; a product x*w is evaluated as hi*hi + lo*hi + hi*lo on the fp16 matrix cores with fp32 accumulation
// (the dropped lo*lo term is 2^-22 relative), i.e. a layer is the tuned fp16 kernel over 3x the K.  Activation buffers keep the
// NHWC channel-slice scheme with 2*C halves per pixel: [C high halves | C low halves].
enum Precision { PREC_F16 = 0, PREC_F32 = 1, PREC_F16X3 = 2 };

// One fused Conv2d(+folded BN bias)(+SiLU)(+residual) over NHWC activations, as an implicit GEMM:
//   out[pix, n] = act( sum_{tap,c} in[pix(tap), c] * w[n, tap, c] + bias[n] ) (+ res[pix, n])
// Input may be the channel-concat of two NHWC segments (k == 1 only); segment 0 may be read through a
// nearest x2 upsample (nn.Upsample + Concat of yolov8 layers 10-11 / 13-14 folded into the consumer).
struct ConvArgs {
    const void* in0; int in0_ct, in0_coff, c0, up0;   // segment 0: base, channels per pixel of the buffer, channel offset, #channels
    const void* in1; int in1_ct, in1_coff, c1;        // segment 1 (c1 == 0: absent)
    const void* wgt;                                   // packed [K chunk][k*k][Cout_pad128][128 B]; rows permuted per 64 (pack_weights)
    const void* wgt32; uint32_t wgt32_bytes;           // second copy with 64-byte K chunks (3x3 s1 fp16 layers, conv3x3_wide_kernel) or null
    const float* bias;                                 // [Cout_pad64]
    void* out; int out_ct, out_coff;                   // NHWC destination slice
    int out_bs, out_ro;                                // destination pixel = b*out_bs + out_ro + (ho*Wo+wo)
    int out_f32;                                       // 1: store fp32 regardless of the activation type (detect head)
    const void* res; int res_ct, res_coff;             // residual source (same pixel grid) or null
    int B, Hi, Wi, Ho, Wo, Cin, Cout, k, s, act;
    uint32_t in0_bytes, in1_bytes, wgt_bytes;          // buffer extents for the hardware range check
    int dbg;                                           // developer ablation bits (0 in production)
    // fp16x3 context (split = passes over K: 3, or 2 when the layer's weights are fp16-exact up to a per-channel scale; 0 elsewhere): *_ct are in halves (2 * channels of the tensor); the low halves of a slice sit *_lo halves behind
    // its high halves.  wgt / wgt32 then hold 3 passes over K: [w_hi | w_lo | w_hi] against inputs [x_lo | x_hi | x_hi]; oscale[n] is
    // the power of two that undoes the weight scale of output channel n (applied to the accumulator before the bias).
    int split, in0_lo, in1_lo, out_lo, res_lo;
    const float* oscale;                               // [Cout_pad128] or null
    // back-to-back fusion (fp16 context, pixels-direct kernel with a 256-channel tile that holds ALL output channels of its pixels): the
    // fused Conv+bias+SiLU result of this layer is not stored but multiplied, in registers, by the 1x1 convolution that is its only
    // reader: wgt2 = that layer's packed weights with its INPUT channels permuted to the accumulator order of the kernel
    // (pack_weights_fused2), bias2 / act2 / out2* its bias, activation and destination slice (same pixel grid).  wgt2 == null: off.
    const void* wgt2; uint32_t wgt2_bytes; const float* bias2; int act2, cout2;
    void* out2; int out2_ct, out2_coff;
};

// First layer (Cin = 3 stored as 4, k=3, s=2): direct convolution.
struct StemArgs {
    const void* in; void* out; const float* w; const float* bias;   // w: [27][Cout] fp32 (tap-major, then c)
    const void* wpk;                                                // fp16 context: [64][32] packed panel (pack_stem_weights)
    int B, Hi, Wi, Ho, Wo, Cout, out_ct, out_coff;
    int out_lo;                                                     // fp16x3 context: fp32 input, output as high / low halves (see ConvArgs)
};

// model.0 + model.1 of the YOLOv8/YOLO11 graphs in one kernel (fp16 context): the 64-channel half-resolution map (8 MB
// per 512x512 tile) never goes to HBM
struct StemDownArgs {
    const void* in; uint32_t in_bytes; int B, Hi, Wi;                  // NHWC4 fp16 network input
    const void* wpk2; const float* bias0;                               // stem panel [64][64] fp16 (pack_stem_weights2), bias
    const void* wgt32; uint32_t wgt32_bytes; const float* bias1;        // 3x3 s2 64->128 weights, 64-byte K chunks (pack_weights(..., 64))
    void* out; int out_ct, out_coff;
    int H1, W1, Ho, Wo;                                                 // stem map, output map
    int dbg;                                                            // developer timing switches (CY_SD_DBG): skip a phase
};

// Fused Bottleneck(64 -> 64 -> 64, 3x3 + 3x3 stride 1, SiLU, optional shortcut) on NHWC fp16 channel slices (bneck64.hip)
struct BneckArgs {
    const void* in; int in_ct, in_coff; uint32_t in_bytes;     // input slice (also the residual when shortcut)
    void* out; int out_ct, out_coff;                            // output slice (same pixel grid)
    const void* wfrag;                                          // pack_bneck_weights: [conv][block][k-step][lane] x 16 B
    const float* bias1; const float* bias2;                     // folded biases of cv1 / cv2
    int B, H, W, shortcut;
    int dbg;                                                    // developer ablation bits (0 in production)
};
constexpr size_t BNECK_WFRAG_BYTES = 2 * 4 * 18 * 1024;
hipError_t launch_bneck64(const BneckArgs& a, hipStream_t s);
void pack_bneck_weights(const float* W1, const float* W2, void* dst);      // W: [64][64][3][3] fp32

struct PoolArgs {   // MaxPool2d(5,1,2) on a channel slice of an NHWC buffer, -inf padding
    const void* src; void* dst; int ct, src_coff, dst_coff, C, B, H, W;
    int lo;                                            // fp16x3 context: offset of the low halves (0: plain)
};

// YOLO11 operators (cy_extra.hip)
struct DwArgs {     // depth-wise 3x3 stride 1 over NHWC channel slices; w: [9][C] fp32, bias [C] fp32
    const void* in; int in_ct, in_coff; void* out; int out_ct, out_coff; const void* res; int res_ct, res_coff;
    const float* w; const float* bias; int B, H, W, C, act;
    int blk, gstride, goff;                            // input channel of output channel c: (c/blk)*gstride + goff + c%blk (blk = 0: c)
    int in_lo, out_lo, res_lo;                         // fp16x3 context: offsets of the low halves (see ConvArgs)
    int xcd;                                           // set by the launch: workgroups in XCD-contiguous order
};
struct AttnArgs {   // softmax(q^T k * scale) applied to v, per head; qkv channels per head: [q kd | k kd | v hd]
    const void* qkv; int ct, coff; void* out; int out_ct, out_coff; int B, N, heads, kd, hd; float scale;
    int lo, out_lo;                                    // fp16x3 context: offsets of the low halves of qkv / out
};
hipError_t launch_dwconv(Precision p, const DwArgs& a, hipStream_t s);
hipError_t launch_attention(Precision p, const AttnArgs& a, hipStream_t s);

hipError_t launch_conv(Precision p, const ConvArgs& a, hipStream_t s);
enum ConvVariant { CONV_GENERIC_128 = 0, CONV_GENERIC_64, CONV_HALO8_128, CONV_PP_64, CONV_PP_128, CONV_HALO16_128, CONV_C64_PERSIST, CONV_GENERIC_BIG, CONV_WIDE_128, CONV_DIRECT_256, CONV_DIRECT_128, CONV_WIDE_64, CONV_WIDE_DUAL, CONV_STRIP_128, CONV_HEAD_1X1, CONV_NUM_VARIANTS };
int conv_variant(Precision p, const ConvArgs& a);          // which kernel launch_conv picks
const char* conv_variant_name(int v);
// the two output 1x1s of a detect-head level (box branch a, class branch b) in one launch of head1x1_pair_kernel (conv_igemm.hip)
bool head_pair_ok(Precision p, const ConvArgs& a, const ConvArgs& b);
hipError_t launch_head_pair(Precision p, const ConvArgs& a, const ConvArgs& b, hipStream_t s);
void debug_read_stamps(unsigned long long* out8, bool reset);
void debug_read_wg_stamps(unsigned long long* out, int n);      // raw per-workgroup phase records (n x 4), stamped builds   // developer diagnostics (CY_DBG=64)
void debug_read_pre_stamps(unsigned long long* out8, bool reset);
void debug_fastdiv(const double* d_a, const double* d_b, double* d_fast, double* d_ref, int n);      // fast_div vs `/` (cy_preproc.hip)   // phases of pre_stats_kernel (cy_preproc.hip), stamped builds only
hipError_t launch_stem(Precision p, const StemArgs& a, hipStream_t s);
hipError_t launch_stem_down(const StemDownArgs& a, hipStream_t s);
long stem_down_blocks(const StemDownArgs& a);
bool batch_invariant();                                            // CY_BATCH_INVARIANT=1 (conv_igemm.hip)
void pack_stem_weights2(const float* W, int cout, void* dst);     // 64*64 fp16, k = tap*4 + c (taps 0..7), 32 + c (tap 8)
hipError_t launch_pool5(Precision p, const PoolArgs& a, hipStream_t s);

// Host-side weight packing into the layout launch_conv expects.
//   W: [Cout][Cin][k][k] fp32 -> dst: [Cout_pad64][k*k][Cin] (fp16 or fp32), rows permuted within each 64-row group
//   so that MFMA row (ni, rr) holds channel 64*blk + 16*(rr>>2) + 4*ni + (rr&3).
size_t packed_weight_bytes(Precision p, int cout, int cin, int k, int chunk_bytes = 128);
void pack_weights(Precision p, const float* W, int cout, int cin, int k, void* dst, int chunk_bytes = 128);
// second layer of a back-to-back pair (ConvArgs::wgt2): [cout2][cin2] 1x1 weights, cin2 = 256, packed like pack_weights(PREC_F16, ...,
// k = 1) with input channel 64 c + 16 q + 8 kk + j stored at K position 64 c + 32 kk + 8 q + j (what lane quarter q of the first
// layer's accumulators holds, in the order the MFMA B operand wants it)
void pack_weights_fused2(const float* W2, int cout2, int cin2, void* dst);
// fp16x3 context: scaled [hi | lo | hi] passes (each pass padded to whole K chunks); oscale: [pad128(cout)] floats, 2^-s per channel.
// passes = 2: [w16 | w16] of a filter that is exactly fp16 values times a per-channel scale (oscale = scale); x3_passes() tells which
// form a layer admits (scale: [cout] or null = ones).
int x3_passes(const float* W, int cout, int cin, int k, const float* scale);
size_t packed_weight_bytes_x3(int cout, int cin, int k, int chunk_bytes = 128, int passes = 3);
void pack_weights_x3(const float* W, int cout, int cin, int k, void* dst, float* oscale, int chunk_bytes = 128, int passes = 3, const float* scale = nullptr);
// fp32 NHWC [npix][C] <-> high/low halves [npix][2C] (test entry cy_conv_bn_silu, debug reads)
hipError_t launch_x3_split(const float* in, void* out, long npix, int C, hipStream_t s);
hipError_t launch_x3_merge(const void* in, float* out, long npix, int C, hipStream_t s);
void pack_stem_weights(const float* W, int cout, void* dst);      // 64*32 fp16
// Workgroups b and b + 8 share an XCD (round-robin dispatch over the 8 XCDs, each with its own L2): give each XCD a contiguous run of
// work items so that neighbours in the index space (adjacent image rows of a stencil, channel blocks of the same pixels) meet in one L2.
// Bijective for any number of workgroups.
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
__host__ __device__ inline int pad64(int c) { return (c + 63) / 64 * 64; }
__host__ __device__ inline int pad128(int c) { return (c + 127) / 128 * 128; }

// ---- detection post-processing --------------------------------------------------------------
struct DecodeArgs {
    const float* pred;       // [B][A][64+nc] raw head output (box logits, class logits)
    int B, A, nc;
    int lvl_h[3], lvl_w[3];  // grid of each stride-8/16/32 level
    float conf;
    // outputs: candidates per tile (fixed capacity cap), in anchor order
    float* cand;             // [B][cap][6]  x1,y1,x2,y2 (letterboxed px), score, class
    int* cand_anchor;        // [B][cap]
    int* cand_count;         // [B] (may exceed cap; clamped by consumers)
    int cap;
};
hipError_t launch_decode(const DecodeArgs& a, hipStream_t s);

struct NmsArgs {
    const float* cand; const int* cand_anchor; const int* cand_count; int cap;   // from decode
    int B; float iou; int max_det;
    // letterbox undo (ultralytics scale_boxes): x = (x - padw)/gain, clamp to [0,w0]x[0,h0]
    float gain; int padw, padh, w0, h0;
    float* det;              // [B][max_det][6]
    int* det_anchor;         // [B][max_det] anchor index of each kept detection (parity witness)
    int* det_count;          // [B]
    // workspace
    uint64_t* keys;          // [B][cap_pow2]
    uint64_t* mask;          // [B][cap][cap/64]
    int* counters;           // context counters (cy_detect_counters) or null: [1] += 1 for a tile whose candidates overflowed cap
};
hipError_t launch_nms(const NmsArgs& a, hipStream_t s);

struct MergeArgs {          // Analyzer.process_detections on device
    const float* det; const int* det_count; int B, max_det;
    float score_thr; double soft, hard;
    float* out; int* out_count; int* out_src;    // [B][max_det][6], [B], [B][max_det] (index into det rows)
    int* err;               // [B] number of degenerate boxes dropped (reference would assert)
    int* counters;          // context counters or null: [0] += degenerate boxes dropped
};
hipError_t launch_iou_merge(const MergeArgs& a, hipStream_t s);
// gathered tile records -> per-tile counts / status / prefix (hdr: 3 T + 1 ints) and the valid detections in tile-id order (out)
hipError_t launch_compact_records(const float* g, long long rows, const long long* perm, int T, int stride, int* hdr, float* out, hipStream_t s);

// ---- preprocessing ---------------------------------------------------------------------------
enum PreOp { OP_BKG = 1, OP_SHIFT = 2, OP_CLIP = 3, OP_ZSCALE = 4, OP_HISTEQ = 5, OP_MINMAX = 6 };
struct PreStage { int op; double p0, p1, p2; int flag; };   // op parameters (see cy_preproc.hip)
constexpr int MAX_STAGES = 8;
constexpr int MAX_PRE_BATCH = 256;
struct PreProgram { int n; PreStage st[MAX_STAGES]; };
struct PreArgs {
    const float* mosaic; int MH, MW;      // resident mosaic, native-endian fp32, non-finite already 0
    int txy[2 * 256];                      // tile origins x0,y0 in mosaic pixels, B <= 256 (kernel argument: no staging copy)
    int B, th, tw;                         // tile box of this shape class
    PreProgram prog[3]; int nprog;         // 1: one program broadcast to 3 channels; 3: per-channel programs
    double* params;                        // [B][3][MAX_STAGES][4] solved stage parameters (workspace / witness)
    double* histeq;                        // [B][3][520] bin centres [0:256] + cdf [256:512] for OP_HISTEQ (workspace)
    int* status;                           // [B] 0 ok, 1 preprocessing returned None, 2 constant-row check failed
    // network input
    void* out; int out_prec; int H, W, top, left;   // NHWC4 letterboxed canvas, fill 114/255
    double* scratch;                       // [B][3][th*tw] fp64 preprocessed image when a resize is needed, else null
    int new_h, new_w;                      // resized size (== th,tw when no resize)
    int* counters;                         // context counters or null: [2] += median-bracket hits, [3] += misses (cy_preproc.hip)
    int variant;                           // developer A/B switch of the statistics kernel (CY_PRE_VARIANT; 0 = shipped form)
    int fuse01;                            // set by the launch: programs 0 and 1 run in one workgroup (shared initial set)
};
hipError_t launch_preproc(const PreArgs& a, hipStream_t s);
hipError_t launch_preproc_planes(const PreArgs& a, hipStream_t s);    // statistics + checks + float64 planes into a.scratch (no packing)
hipError_t launch_letterbox_pack(const PreArgs& a, hipStream_t s);   // uses scratch/th/tw/new_*/H/W/top/left/out only
hipError_t launch_mosaic_prepare(float* data, size_t n, int big_endian, hipStream_t s);

}  // namespace cy
