// Execution plan of the YOLOv8 detection graph on NHWC buffers (host side, no HIP).
// Graph per the public yolov8.yaml as summarised in SURVEY.md Appendix A.1 step 4; what the reference runs through
// `self.model(...)` at caesar_yolo/evaluation.py:181-193.
#pragma once
#include <string>
#include <vector>

namespace cy {

struct ConvDesc { std::string name; int cin, cout, k, s, act; int groups = 1; };

struct Tensor { int level; int C; };          // spatial size = (H >> level, W >> level), C channels per pixel

enum OpKind { OPK_STEM = 0, OPK_CONV = 1, OPK_POOL = 2, OPK_DWCONV = 3, OPK_ATTN = 4 };   // 3, 4: YOLO11 plans (CYW2 files)

struct Op {
    OpKind kind;
    int conv;                                  // index into Plan::convs (STEM/CONV)
    int in0, in0_coff, c0, up0;                // segment 0 (tensor id, channel offset, channels, read through x2 nearest upsample)
    int in1, in1_coff, c1;                     // segment 1 or in1 = -1
    int out, out_coff;                         // destination tensor slice, or out = -1 for the head output
    int res, res_coff;                         // residual tensor slice or res = -1
    int pred_level, pred_coff;                 // when out == -1: which stride level, channel offset inside [64+nc]
    int p0 = 0, p1 = 0, p2 = 0, p3 = 0;        // DWCONV: input channel map (blk, gstride, goff); ATTN: heads, key_dim, head_dim
};

struct Plan {
    char scale; int nc;
    std::string arch = "yolov8";
    std::vector<ConvDesc> convs;               // canonical (state_dict) order == weight-file order
    std::vector<Tensor> tensors;               // tensor 0 is the network input [B,H,W,4]
    std::vector<Op> ops;                       // execution order
    int feat_level[3];
    bool ok; std::string err;
};

Plan build_plan(char scale, int nc);

}  // namespace cy
