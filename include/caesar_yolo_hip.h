/* caesar_yolo_hip.h -- C-ABI of the MI355X-native tiled-YOLO detect path (libcaesar_yolo_hip.so, gfx950).
 *
 * Drop-in boundary for the one hot path of SKA-INAF/caesar-yolo: everything `Analyzer.predict` does between
 * receiving a tile and handing back merged boxes.  Citations are reference files (relative to the reference
 * checkout) whose behaviour each entry point replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative cy_status; nothing throws across the boundary
 *     (the reference catches any exception from the model call and skips the tile: caesar_yolo/evaluation.py:194-196,
 *     caesar_yolo/inference.py:615-618); `cy_last_error` gives the text;
 *   - `d_*` pointers are DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr()), `h_*` are host
 *     pointers; `stream` is a hipStream_t passed as void* (NULL = default stream); all launches are asynchronous;
 *   - one context per GPU; a context is not thread-safe; no allocation happens after cy_load_weights.
 */
#ifndef CAESAR_YOLO_HIP_H
#define CAESAR_YOLO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cy_ctx cy_ctx;

enum cy_status { CY_OK = 0, CY_ERR_ARG = -1, CY_ERR_HIP = -2, CY_ERR_IO = -3, CY_ERR_STATE = -4, CY_ERR_UNSUPPORTED = -5 };
/* CY_F16: fp16 operands, fp32 accumulate (fast); CY_F32: exact fp32 FMA chains on v_mfma_f32_16x16x4_f32 (reference arithmetic,
 * slow); CY_F16X3: the fast parity context -- activations and weights carried as fp16 high + low halves (22 significand bits),
 * every product evaluated as hi*hi + lo*hi + hi*lo on the fp16 matrix cores with fp32 accumulation (3x the K of CY_F16).
 * Range: in CY_F16 and CY_F16X3 an activation is stored through an fp16 high half, so |activation| must stay below 65504 (beyond
 * that the value becomes inf; only CY_F32 has the fp32 range).  Weights have no such limit (scaled per output channel).
 * Buffers the caller hands to cy_forward / cy_preproc / cy_letterbox_pack / cy_conv_bn_silu are fp32 in the CY_F32 and CY_F16X3
 * contexts and fp16 in CY_F16. */
enum cy_precision { CY_F16 = 0, CY_F32 = 1, CY_F16X3 = 2 };

#define CY_MAX_DET 300        /* ultralytics max_det */
#define CY_DET_STRIDE 6       /* x1,y1,x2,y2,score,class */
#define CY_MAX_STAGES 8

typedef struct cy_config {
    int precision;            /* cy_precision */
    int max_batch;            /* tiles per launch the workspace is sized for */
    int max_h, max_w;         /* largest letterboxed network input (multiple of 32) */
    int max_cand;             /* candidate capacity per tile before NMS (<= 30000 = ultralytics max_nms); 0 -> the anchor
                                 count of a max_h x max_w input (no overflow possible); a tile that overflows a smaller
                                 explicit capacity is counted in cy_detect_counters */
} cy_config;

/* Preprocessing program: the CLI-fixed stage order of scripts/run.py:272-302, one op list per output channel.
 * ops follow caesar_yolo/preprocessing.py: BKG :591-658, SHIFT :664-717, CLIP :723-771, ZSCALE :934-971,
 * HISTEQ :977-1012, MINMAX :75-111.  Chan3Trasformer (:1020-1072) = three different programs. */
enum cy_pre_op { CY_OP_BKG = 1, CY_OP_SHIFT = 2, CY_OP_CLIP = 3, CY_OP_ZSCALE = 4, CY_OP_HISTEQ = 5, CY_OP_MINMAX = 6 };
typedef struct cy_pre_stage {
    int op;
    double p0, p1, p2;        /* BKG: sigma, mask_fract | SHIFT: sigma | CLIP: sigma_low, sigma_up | ZSCALE: contrast | MINMAX: norm_min, norm_max */
    int flag;                 /* BKG: use_mask_box */
} cy_pre_stage;
typedef struct cy_pre_program { int n; cy_pre_stage st[CY_MAX_STAGES]; } cy_pre_program;
typedef struct cy_preproc_cfg {
    int nprog;                /* 0: no preprocessing (raw pixel values go to the network, as without --preprocessing);
                                 1: one program, result replicated to 3 channels; 3: one program per channel */
    cy_pre_program prog[3];
} cy_preproc_cfg;

typedef struct cy_conv_desc { char name[48]; int cin, cout, k, s, act; } cy_conv_desc;

typedef struct cy_letterbox {  /* ultralytics LetterBox geometry for an (h0,w0) image at imgsz */
    int new_h, new_w, top, left, H, W;
} cy_letterbox;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* replaces `model = YOLO(weights_path)` (scripts/run.py:347) + device selection (caesar_yolo/inference.py:205-217) */
int cy_create(int device, const cy_config* cfg, cy_ctx** out);
int cy_destroy(cy_ctx* ctx);
const char* cy_last_error(const cy_ctx* ctx);
int cy_load_weights(cy_ctx* ctx, const char* path);                 /* CYW1 file (caesar_yolo_amd/weights.py) */
int cy_load_weights_mem(cy_ctx* ctx, const void* buf, size_t nbytes);
int cy_num_classes(const cy_ctx* ctx);                               /* len(model.names), caesar_yolo/evaluation.py:46-47 */
const char* cy_class_name(const cy_ctx* ctx, int i);
/* fp16x3 context: how many convolutions run the two-pass form (filter = fp16 values x a per-channel scale, exactly: what a
 * checkpoint stored in fp16 and folded with its BatchNorm in fp32 is -- ultralytics' own load path) and how
 * many the general three-pass form.  out2[0] = two-pass layers, out2[1] = three-pass layers (both 0 in the other contexts). */
int cy_weight_passes(const cy_ctx* ctx, int* out2);

/* ---- host-only helpers (no GPU needed) ------------------------------------------------------- */
int cy_plan_num_convs(char scale, int nc);
int cy_plan_conv_desc(char scale, int nc, int idx, cy_conv_desc* out);
int cy_letterbox_geometry(int h0, int w0, int imgsz, cy_letterbox* out);
int cy_num_anchors(int H, int W);
size_t cy_pred_elems(const cy_ctx* ctx, int B, int H, int W);        /* floats in the raw head output [B][A][64+nc] */

/* ---- stages ---------------------------------------------------------------------------------- */
/* utils.read_fits / read_fits_crop value semantics (caesar_yolo/utils.py:219, :394): big-endian FITS floats ->
 * native, non-finite -> 0, in place on the HBM-resident mosaic */
int cy_mosaic_prepare(cy_ctx* ctx, float* d_data, size_t n, int big_endian, void* stream);

/* Analyzer.predict steps evaluation.py:146-176 for B tiles of one shape (th x tw) cropped from the resident mosaic:
 * 3-channel cube, DataPreprocessor pipeline, None / constant-row rejection, then the model's own LetterBox + BGR
 * flip + /255.  h_tiles: B x {x0, y0} tile origins in mosaic pixels.
 * d_netin: [B][H][W][4] (fp16 or fp32 per context precision); d_status[B]: 0 ok, 1 pipeline returned None, 2 row check */
int cy_preproc(cy_ctx* ctx, const float* d_mosaic, int MH, int MW, const int* h_tiles, int B, int th, int tw,
               int imgsz, const cy_preproc_cfg* cfg, void* d_netin, int* d_status, void* stream);
/* the same statistics and rejection checks, but the output is the preprocessed image itself, d_planes [B][3][th*tw] float64
 * in image channel order = what `DataPreprocessor.__call__` hands back to Analyzer.predict (caesar_yolo/evaluation.py:157-161,
 * before the model's LetterBox): full-precision parity witness, and the picture Analyzer.draw_results plots (:351-411) */
int cy_preproc_planes(cy_ctx* ctx, const float* d_mosaic, int MH, int MW, const int* h_tiles, int B, int th, int tw,
                      const cy_preproc_cfg* cfg, double* d_planes, int* d_status, void* stream);
/* witness for parity tests: solved per-stage parameters of the last cy_preproc call, [B][3][CY_MAX_STAGES][4] doubles */
int cy_preproc_params(cy_ctx* ctx, double* h_out, int B);

/* the model's own input handling for caller-supplied images (the `model(ndarray)` surface, evaluation.py:181-193):
 * d_planes [B][3][h0][w0] float64 (image channel order, nominal range [0,255]) -> LetterBox + flip + /255 -> d_netin */
int cy_letterbox_pack(cy_ctx* ctx, const double* d_planes, int B, int h0, int w0, int imgsz, void* d_netin, void* stream);

/* DetectionModel.forward: d_netin [B][H][W][4] -> d_pred [B][A][64+nc] fp32 raw head output */
int cy_forward(cy_ctx* ctx, const void* d_netin, int B, int H, int W, float* d_pred, void* stream);
/* per-launch timing of the forward kernels with hipEvents on the caller's stream (bench.py roofline): enable, run
 * cy_forward / cy_detect_tiles as usual, then read the totals per kernel variant (up to 8 entries: the conv variants,
 * stem, pool); flops are the ALGORITHMIC 2*MACs of the launches timed */
typedef struct cy_prof_entry { char kernel[64]; double ms; double flops; long launches; } cy_prof_entry;
int cy_profile_enable(cy_ctx* ctx, int on);    /* 0 off, 1 every cy_forward call, N > 1 every N-th call (sampling) */
int cy_profile_summary(cy_ctx* ctx, cy_prof_entry* out, int cap);
/* the same for the launches of one lane of cy_detect_tiles only: 0 = the caller's stream (full batches, incl. the second stream of
 * a split batch), 1 = the small-batch lane, -1 = all */
int cy_profile_summary_lane(cy_ctx* ctx, cy_prof_entry* out, int cap, int lane);
int cy_profile_layers(cy_ctx* ctx, cy_prof_entry* out, int cap);    /* the same, one entry per convolution (graph order) */
/* copy the output of one named convolution of the last cy_forward to host as fp32 [B][C][Ho][Wo] (test hook) */
int cy_debug_read_conv(cy_ctx* ctx, const char* conv_name, float* h_out, size_t cap_elems, int* dims4);

/* Detect decode + non_max_suppression + scale_boxes:
 * d_det [B][300][6], d_det_anchor [B][300] (anchor index of each kept box), d_count [B] */
int cy_decode_nms(cy_ctx* ctx, const float* d_pred, int B, int H, int W, int h0, int w0, float conf, float iou,
                  float* d_det, int* d_det_anchor, int* d_count, void* stream);

/* developer diagnostics: in-kernel cycle stamps of the 3x3 halo kernel (enabled with CY_DBG=64) */
int cy_debug_stamps(unsigned long long* out8, int reset);
/* developer diagnostics: the pack kernel's quotient by a per-tile constant (d_fast) beside the float64 division (d_ref), element-wise on
 * device arrays of n doubles: must agree bit for bit (tests/test_gpu_preproc.py) */
int cy_debug_fastdiv(const double* d_a, const double* d_b, double* d_fast, double* d_ref, int n);
/* number of pre-NMS candidates per tile of the last cy_decode_nms call (diagnostics) */
int cy_debug_cand_counts(cy_ctx* ctx, int* h_out, int B);

/* Analyzer.process_detections (caesar_yolo/evaluation.py:252-346): score re-filter, IoU graph, connected components,
 * best score per component.  d_out [B][300][6], d_out_count [B], d_out_src [B][300] = row of d_det kept */
int cy_iou_merge(cy_ctx* ctx, const float* d_det, const int* d_count, int B, float score_thr, double thr_soft,
                 double thr_hard, float* d_out, int* d_out_count, int* d_out_src, void* stream);

/* the whole per-tile path for B same-shape tiles; status/det/count as above (merged detections in tile pixels).
 * Consecutive calls are software-pipelined on internal side streams (preprocessing / post-processing of neighbouring
 * batches overlap the forward pass): give every in-flight call its own output buffers and call cy_detect_flush before
 * reading them -- after it, work queued on `stream` is ordered behind all outstanding batches */
int cy_detect_tiles(cy_ctx* ctx, const float* d_mosaic, int MH, int MW, const int* h_tiles, int B, int th, int tw,
                    int imgsz, const cy_preproc_cfg* cfg, float conf, float iou, double thr_soft, double thr_hard,
                    float* d_out, int* d_out_count, int* d_status, void* stream);

int cy_detect_flush(cy_ctx* ctx, void* stream);
/* Ordering contract of unflushed cy_detect_tiles calls.  The internal streams are ordered behind `stream` (everything the caller
 * queued on it so far) at: the first call after cy_load_weights / cy_detect_flush; the first call after cy_mosaic_prepare; the
 * first call that names a d_mosaic buffer this pipeline has not seen yet; the first call after cy_detect_fence.  Any OTHER work
 * the caller queues on `stream` between two unflushed calls -- a re-upload into a mosaic buffer already used, a memset of an
 * output buffer -- is NOT ordered before the internal streams touch those buffers: call cy_detect_fence after queuing it (the
 * next cy_detect_tiles then waits for it; costs the overlap of that one batch's preprocessing with the previous forward), or
 * prepare every buffer before the first call (what caesar_yolo_amd.inference.TileEngine does).
 * "A buffer this pipeline has seen" is its ADDRESS: a caching allocator (PyTorch's) may hand a NEW mosaic tensor the address of one
 * freed inside the same unflushed pipeline, and that buffer then counts as seen although its upload is still queued on `stream`.
 * A caller that allocates mosaic buffers between unflushed calls must call cy_detect_fence after every upload (uploads through
 * cy_mosaic_prepare are fenced by the library itself). */
int cy_detect_fence(cy_ctx* ctx, void* stream);

/* Rank 0 after the gather (replaces the unpickling of the workers' source lists, caesar_yolo/inference.py:936-984): d_gathered holds
 * the ranks' fixed-capacity tile records, n_rows rows of row_floats = 300*6 + 3 floats {detections | count | status | tile id};
 * d_perm[t] = row index (over all ranks) of tile t (an index outside [0, n_rows) makes tile t a rejected tile with status CY_ERR_ARG).  Writes d_hdr = {count[T] (0 for a rejected tile) | status[T] | exclusive prefix[T] | total}
 * (3 T + 1 ints) and the valid detections in tile-id order into d_out (room for T * 300 * 6 floats): what cy_make_tile_records takes. */
int cy_compact_records(const float* d_gathered, long long n_rows, const long long* d_perm, int T, int row_floats, int* d_hdr, float* d_out, void* stream);
/* the same with the context named: launches on the context's device whatever device is current (the form above launches on the
 * CURRENT device: the caller must have selected the device that owns the buffers) */
int cy_compact_records_ctx(cy_ctx* ctx, const float* d_gathered, long long n_rows, const long long* d_perm, int T, int row_floats, int* d_hdr, float* d_out, void* stream);

/* events the reference would not survive silently, accumulated over cy_decode_nms / cy_iou_merge / cy_detect_tiles calls:
 * out4[0] degenerate boxes (x1 >= x2 or y1 >= y2) dropped before the IoU merge -- the reference aborts on them inside
 * get_iou's assert (caesar_yolo/utils.py:78-81); out4[1] tiles with more candidates than the
 * context's capacity (ultralytics keeps the top max_nms = 30000 by score; here the surplus is dropped in arrival order, so
 * a non-zero count means "raise max_cand"); out4[2], out4[3]: performance diagnostics of the sigma-clip statistics (median
 * selections served by the one-pass bracket / fallen back to the three-pass radix select).  Synchronises the device.
 * reset != 0 clears them. */
int cy_detect_counters(cy_ctx* ctx, long long* out4, int reset);

/* single fused Conv+bias+SiLU layer on caller tensors (kernel-level parity tests).
 * d_in [B][Hi][Wi][Cin], d_out [B][Ho][Wo][Cout] in the context precision; h_w [Cout][Cin][k][k], h_b [Cout] fp32;
 * d_res optional residual [B][Ho][Wo][Cout] */
int cy_conv_bn_silu(cy_ctx* ctx, const void* d_in, int B, int Hi, int Wi, int Cin, const float* h_w, const float* h_b,
                    int Cout, int k, int s, int act, const void* d_res, void* d_out, void* stream);

/* one fused Bottleneck(64, 64, shortcut) = x [+] SiLU(cv2(SiLU(cv1(x)))) with two folded 3x3 convs, as the forward runs the
 * stride-4 C2f blocks of yolov8l in the fp16 context (kernel-level parity tests).  d_in, d_out [B][H][W][64] fp16;
 * h_w1, h_w2 [64][64][3][3], h_b1, h_b2 [64] fp32 */
int cy_bottleneck64(cy_ctx* ctx, const void* d_in, int B, int H, int W, const float* h_w1, const float* h_b1,
                    const float* h_w2, const float* h_b2, int shortcut, void* d_out, void* stream);

/* ---- catalog records and cross-tile merge (host code, no GPU) --------------------------------- */
/* Analyzer.make_json_results (caesar_yolo/evaluation.py:418-469: int() truncation, tile-local edge rule, tile origin)
 * followed by SFinder.find_sources_at_edge (caesar_yolo/inference.py:663-726).
 * det: n x 6 floats {x1,y1,x2,y2,score,class} in TILE pixels, grouped by tile in ascending tile order;
 * det_tile: n tile ids; tiles: T x 4 ints {xmin, xmax_excl, ymin, ymax_excl} (utils.generate_tiles order);
 * rec: n x 8 doubles {x1,y1,x2,y2,score,class_id,tile_id,edge}, edge = 0 | 1 (tile-local rule only, int in the
 * reference JSON) | 2 (set True by find_sources_at_edge) */
int cy_make_tile_records(const float* det, const int* det_tile, int n, const int* tiles, int T, double* rec);
/* SFinder.merge_edge_sources (caesar_yolo/inference.py:731-931) on those records.
 * out: up to n x 8 doubles {x1,y1,x2,y2,score,class_id,edge,merged}; returns the number of output sources (>= 0) */
int cy_merge_edge_sources(const double* rec, int n, const int* tiles, int T, double* out);

#ifdef __cplusplus
}
#endif
#endif
