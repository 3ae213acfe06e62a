// Detection post-processing on device (gfx950): Detect/DFL decode + confidence filter, wavefront-scan NMS with the
// letterbox undo, and the reference's per-tile IoU graph merge.
//
//   decode  : ultralytics Detect inference branch (DFL softmax-expectation, dist2bbox(xywh) * stride, sigmoid) and the
//             candidate filter of non_max_suppression -- SURVEY.md Appendix A.1 steps 5-6
//   nms     : torchvision.ops.nms semantics on class-offset boxes (+ cls*7680), IoU > thr (strict), stable score order,
//             first 300 kept, then scale_boxes/clip_boxes -- Appendix A.1 steps 6-7
//   merge   : Analyzer.process_detections (caesar_yolo/evaluation.py:252-346) with get_iou (caesar_yolo/utils.py:54-107)
//             and Graph.connectedComponents (caesar_yolo/graph.py:2-41)
//
// All threshold comparisons are discontinuities, so the arithmetic is written operation by operation in the reference's
// precision and order, with floating-point contraction disabled for this translation unit.
#include "cy_kernels.h"
#pragma clang fp contract(off)

namespace cy {

constexpr int MAXDET = 300;          // ultralytics max_det (CY_MAX_DET)
constexpr int MAX_NMS = 30000;       // ultralytics max_nms
constexpr float MAX_WH = 7680.0f;    // ultralytics max_wh (class offset)

// ------------------------------------------------------------------------------------------------ decode
// One thread per anchor, class logits first.  Alone on the GPU the kernel takes 59 us per 256 tiles of 512^2 (it touches the cache
// lines that hold the class logits: 190 MB, 3.2 TB/s); inside the pipelined pass, on the low-priority post-processing stream beside the
// conv stack, its launches last ~430 us.  Round 4 tried a STAGED form (a workgroup's 256 x 69 floats copied to LDS with coalesced
// 16-byte loads, rows read from there): it moves twice the bytes and 70 KB of LDS per workgroup: 111 us alone, 590 us in the pass.
__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.A) return;
    // anchor -> (level, y, x): levels are concatenated stride 8, 16, 32, each row-major
    int lvl = 0, r = i;
    const int n0 = a.lvl_h[0] * a.lvl_w[0], n1 = a.lvl_h[1] * a.lvl_w[1];
    if (r >= n0) { r -= n0; lvl = 1; if (r >= n1) { r -= n1; lvl = 2; } }
    const int gw = a.lvl_w[lvl];
    const float ax = (float)(r % gw) + 0.5f, ay = (float)(r / gw) + 0.5f;
    const float stride = (float)(8 << lvl);
    const float* p = a.pred + ((size_t)b * a.A + i) * (64 + a.nc);
    // class scores first: most anchors fail the confidence test and skip the DFL work
    float best = -1.0f; int bj = 0;
    for (int c = 0; c < a.nc; ++c) {
        const float s = 1.0f / (1.0f + expf(-p[64 + c]));
        if (s > best) { best = s; bj = c; }        // first maximum wins, as torch.max(1)
    }
    if (!(best > a.conf)) return;
    float d[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
        const float* q = p + side * 16;
        float m = q[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) m = fmaxf(m, q[k]);
        float e[16], s = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { e[k] = expf(q[k] - m); s += e[k]; }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += (e[k] / s) * (float)k;
        d[side] = acc;
    }
    // dist2bbox(xywh=True) then * stride, then xywh2xyxy of non_max_suppression
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    const float cx = ((x1 + x2) / 2.0f) * stride, cy_ = ((y1 + y2) / 2.0f) * stride;
    const float w = (x2 - x1) * stride, h = (y2 - y1) * stride;
    const float hw = w / 2.0f, hh = h / 2.0f;
    const int slot = atomicAdd(a.cand_count + b, 1);
    if (slot >= a.cap) return;
    float* o = a.cand + ((size_t)b * a.cap + slot) * 6;
    o[0] = cx - hw; o[1] = cy_ - hh; o[2] = cx + hw; o[3] = cy_ + hh; o[4] = best; o[5] = (float)bj;
    a.cand_anchor[(size_t)b * a.cap + slot] = i;
}

hipError_t launch_decode(const DecodeArgs& a, hipStream_t s) {
    if (a.A >= 65536 || a.cap >= 65536) return hipErrorInvalidValue;     // key packing below: 16 + 16 bits
    hipLaunchKernelGGL(decode_kernel, dim3((a.A + 255) / 256, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ NMS
// One 256-thread workgroup per tile.  (1) order candidates by (score desc, anchor asc) with a bitonic network on 64-bit
// keys -- in LDS when they fit, else through L2; (2) wave 0 scans the ordered list 64 boxes at a time: every lane tests
// its box against the boxes kept so far (<= 300, in LDS), then the 64 survivors are resolved against each other in
// order with ballots; the scan stops at max_det kept boxes, which are exactly torchvision's keep[:max_det].
constexpr int NMS_LDS_KEYS = 8192;     // sort keys held in (dynamic) LDS: 64 KiB; more candidates than this sort in global memory

__device__ __forceinline__ bool iou_gt(float ix1, float iy1, float ix2, float iy2, float iarea,
                                       float jx1, float jy1, float jx2, float jy2, float jarea, float thr) {
    const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1), xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (iarea + jarea - inter);
    return ovr > thr;
}

constexpr int NMS_NT = 1024, NMS_NW = NMS_NT / 64;    // the sort and the kept-list test use every wave; the in-order resolution is wave 0's
__global__ __launch_bounds__(NMS_NT) void nms_kernel(const NmsArgs a, int cap_pow2) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lkeys[];      // NMS_LDS_KEYS entries (launch_nms)
    __shared__ float kx1[MAXDET], ky1[MAXDET], kx2[MAXDET], ky2[MAXDET], kar[MAXDET];
    __shared__ int kslot[MAXDET];
    const int b = blockIdx.x, tid = threadIdx.x;
    int n = a.cand_count[b];
    if (n > a.cap) { n = a.cap; if (tid == 0 && a.counters) atomicAdd(a.counters + 1, 1); }
    if (n == 0) { if (tid == 0) a.det_count[b] = 0; return; }
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    uint64_t* keys = np2 <= NMS_LDS_KEYS ? lkeys : a.keys + (size_t)b * cap_pow2;
    const float* cand = a.cand + (size_t)b * a.cap * 6;
    const int* canch = a.cand_anchor + (size_t)b * a.cap;
    for (int i = tid; i < np2; i += NMS_NT) {
        uint64_t k = ~0ull;
        if (i < n) {
            const unsigned sb = __float_as_uint(cand[i * 6 + 4]);            // scores are positive: bit order = value order
            k = ((uint64_t)(~sb) << 32) | ((uint64_t)(unsigned)canch[i] << 16) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (np2 >> 1); p += NMS_NT) {   // thread -> compare-exchange pair (i, i | j): every thread has work
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
                const bool up = (i & kk) == 0;
                const uint64_t x = keys[i], y = keys[ixj];
                if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
            }
            __syncthreads();
        }
    if (n > MAX_NMS) n = MAX_NMS;
    // ---- scan in score order, 64 candidates per round.  The test of a round's candidates against the boxes kept so far is
    // split over the sixteen waves (wave w takes kept boxes w, w+16, ...: the dense tiles that set this kernel's duration have
    // thousands of candidates against up to 300 kept boxes); wave 0 then resolves the round's survivors among themselves
    // in order, exactly as a one-wave scan would.
    __shared__ unsigned long long amask[NMS_NW];
    __shared__ int s_nk;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_nk = 0;
    __syncthreads();
    int nk = 0;
    for (int base = 0; base < n && nk < a.max_det; base += 64) {
        const int idx = base + lane;
        const bool valid = idx < n;
        int slot = 0;
        float x1 = 0, y1 = 0, x2 = 0, y2 = 0, area = 0;
        if (valid) {
            slot = (int)(keys[idx] & 0xFFFFu);
            const float* c = cand + slot * 6;
            const float off = c[5] * MAX_WH;            // class-aware: boxes + cls * max_wh
            x1 = c[0] + off; y1 = c[1] + off; x2 = c[2] + off; y2 = c[3] + off;
            area = (x2 - x1) * (y2 - y1);
        }
        bool alive = valid;
        for (int t = wave; t < nk; t += 4 * NMS_NW) {        // four kept boxes per trip: their LDS reads are in flight together
            float qx1[4], qy1[4], qx2[4], qy2[4], qar[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tt = t + NMS_NW * q;
                const bool in = tt < nk;                     // past the list: an empty box (intersection 0 -> never suppresses)
                qx1[q] = in ? kx1[tt] : 0.0f; qy1[q] = in ? ky1[tt] : 0.0f; qx2[q] = in ? kx2[tt] : 0.0f; qy2[q] = in ? ky2[tt] : 0.0f;
                qar[q] = in ? kar[tt] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (alive && iou_gt(qx1[q], qy1[q], qx2[q], qy2[q], qar[q], x1, y1, x2, y2, area, a.iou)) alive = false;
            if (__ballot(alive) == 0ull) break;
        }
        const unsigned long long mw = __ballot(alive);
        if (lane == 0) amask[wave] = mw;
        __syncthreads();
        if (wave == 0) {
            unsigned long long m = amask[0];
#pragma unroll
            for (int w = 1; w < NMS_NW; ++w) m &= amask[w];
            alive = (m >> lane) & 1ull;
            while (m != 0ull && nk < a.max_det) {
                const int j = __ffsll((long long)m) - 1;    // lowest alive lane = next kept box (wave-uniform)
                const float jx1 = __shfl(x1, j), jy1 = __shfl(y1, j), jx2 = __shfl(x2, j), jy2 = __shfl(y2, j);
                const float jar = __shfl(area, j);
                const int jslot = __shfl(slot, j);
                if (lane == 0) { kx1[nk] = jx1; ky1[nk] = jy1; kx2[nk] = jx2; ky2[nk] = jy2; kar[nk] = jar; kslot[nk] = jslot; }
                ++nk;
                if (lane > j && alive && iou_gt(jx1, jy1, jx2, jy2, jar, x1, y1, x2, y2, area, a.iou)) alive = false;
                if (lane == j) alive = false;
                m = __ballot(alive);
            }
            if (lane == 0) s_nk = nk;
        }
        __syncthreads();                                // kept[] and the count are visible to every wave's next round
        nk = s_nk;
    }
    // ---- emit: undo the letterbox (scale_boxes + clip_boxes) on the un-offset boxes, in kept (= score) order
    if (tid == 0) a.det_count[b] = nk;
    for (int t = tid; t < nk; t += NMS_NT) {
        const float* c = cand + kslot[t] * 6;
        float bx1 = (c[0] - (float)a.padw) / a.gain, by1 = (c[1] - (float)a.padh) / a.gain;
        float bx2 = (c[2] - (float)a.padw) / a.gain, by2 = (c[3] - (float)a.padh) / a.gain;
        bx1 = fminf(fmaxf(bx1, 0.0f), (float)a.w0); bx2 = fminf(fmaxf(bx2, 0.0f), (float)a.w0);
        by1 = fminf(fmaxf(by1, 0.0f), (float)a.h0); by2 = fminf(fmaxf(by2, 0.0f), (float)a.h0);
        float* o = a.det + ((size_t)b * a.max_det + t) * 6;
        o[0] = bx1; o[1] = by1; o[2] = bx2; o[3] = by2; o[4] = c[4]; o[5] = c[5];
        a.det_anchor[(size_t)b * a.max_det + t] = canch[kslot[t]];
    }
}

hipError_t launch_nms(const NmsArgs& a, hipStream_t s) {
    if (a.max_det > MAXDET) return hipErrorInvalidValue;
    int cp2 = 1;
    while (cp2 < a.cap) cp2 <<= 1;
    static bool attr_set = false;
    const size_t lds = (size_t)NMS_LDS_KEYS * sizeof(uint64_t);
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 16384);
        attr_set = true;
    }
    hipLaunchKernelGGL(nms_kernel, dim3(a.B), dim3(NMS_NT), lds, s, a, cp2);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ IoU graph merge
// One workgroup (four waves) per tile, N <= 300 detections.  Pair tests are lane-parallel (one ballot fills 64 adjacency
// bits) and row-parallel over the waves (LDS atomic OR); connected components are walked by thread 0 in exactly the
// reference's DFS preorder, because the survivor of a component is the FIRST member in that order with the strictly
// largest score (evaluation.py:322-331).
__global__ __launch_bounds__(256) void iou_merge_kernel(const MergeArgs a) {
    constexpr int W64 = (MAXDET + 63) / 64;
    __shared__ int sel[MAXDET];
    __shared__ float bx[MAXDET][4];
    __shared__ float sc[MAXDET];
    __shared__ int cl[MAXDET];
    __shared__ unsigned long long adj[MAXDET][W64];
    __shared__ int stack[MAXDET];
    __shared__ int s_n, s_nerr;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nin = a.det_count[b];
    if (nin > a.max_det) nin = a.max_det;
    const float* det = a.det + (size_t)b * a.max_det * 6;
    // ---- score re-filter (evaluation.py:282 drops score < thr), order preserved; degenerate boxes are dropped and
    // counted (the reference would abort on get_iou's assert)
    int n = 0, nerr = 0;
    if (wave == 0) {
    for (int base = 0; base < nin; base += 64) {
        const int i = base + lane;
        bool keep = false, bad = false;
        if (i < nin) {
            const float* d = det + i * 6;
            keep = !(d[4] < a.score_thr);
            bad = keep && !(d[0] < d[2] && d[1] < d[3]);
            keep = keep && !bad;
        }
        const unsigned long long m = __ballot(keep);
        nerr += __popcll(__ballot(bad));
        if (keep) {
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            const float* d = det + i * 6;
            sel[pos] = i; bx[pos][0] = d[0]; bx[pos][1] = d[1]; bx[pos][2] = d[2]; bx[pos][3] = d[3];
            sc[pos] = d[4]; cl[pos] = (int)d[5];
        }
        n += __popcll(m);
    }
    if (lane == 0) { s_n = n; s_nerr = nerr; }
    }
    __syncthreads();
    n = s_n; nerr = s_nerr;
    for (int i = tid; i < n * W64; i += 256) adj[i / W64][i % W64] = 0ull;
    __syncthreads();
    // ---- adjacency: mergeable = iou >= hard or (same class and iou >= soft); iou = f32 areas, f64 division
    for (int i = wave; i < n - 1; i += 4) {                 // rows are independent: one per wave at a time
        const float ax1 = bx[i][0], ay1 = bx[i][1], ax2 = bx[i][2], ay2 = bx[i][3];
        const float aarea = (ax2 - ax1) * (ay2 - ay1);
        const int acl = cl[i];
        for (int w = i / 64; w * 64 < n; ++w) {
            const int j = w * 64 + lane;
            bool e = false;
            if (j > i && j < n) {
                const float xl = fmaxf(ax1, bx[j][0]), yt = fmaxf(ay1, bx[j][1]);
                const float xr = fminf(ax2, bx[j][2]), yb = fminf(ay2, bx[j][3]);
                double iou = 0.0;
                if (!(xr < xl || yb < yt)) {
                    const float inter = (xr - xl) * (yb - yt);
                    const float barea = (bx[j][2] - bx[j][0]) * (bx[j][3] - bx[j][1]);
                    const float uni = aarea + barea - inter;
                    iou = (double)inter / (double)uni;
                }
                e = (iou >= a.hard) || (acl == cl[j] && iou >= a.soft);
                if (e) atomicOr(&adj[j][i >> 6], 1ull << (i & 63));      // other waves may be setting other bits of this word
            }
            const unsigned long long m = __ballot(e);
            if (lane == 0 && m) atomicOr(&adj[i][w], m);
        }
    }
    __syncthreads();
    // ---- connected components in DFS preorder (graph.py:9-41), survivor = first strict maximum of the score
    if (tid == 0) {
        unsigned long long vis[W64];
        for (int w = 0; w < W64; ++w) vis[w] = 0ull;
        int nout = 0;
        float* out = a.out + (size_t)b * a.max_det * 6;
        int* osrc = a.out_src + (size_t)b * a.max_det;
        for (int v0 = 0; v0 < n; ++v0) {
            if ((vis[v0 >> 6] >> (v0 & 63)) & 1ull) continue;
            float best = 0.0f; int ibest = -1;
            int sp = 0;
            stack[sp++] = v0;
            vis[v0 >> 6] |= 1ull << (v0 & 63);
            if (sc[v0] > best) { best = sc[v0]; ibest = v0; }
            while (sp > 0) {
                const int v = stack[sp - 1];
                int u = -1;
                unsigned long long row[W64];
#pragma unroll
                for (int w = 0; w < W64; ++w) row[w] = adj[v][w];          // all words in flight at once: one LDS latency per step
#pragma unroll
                for (int w = 0; w < W64; ++w) {
                    const unsigned long long c = row[w] & ~vis[w];
                    if (c && u < 0) u = w * 64 + __ffsll((long long)c) - 1;
                }
                if (u < 0) { --sp; continue; }
                vis[u >> 6] |= 1ull << (u & 63);
                if (sc[u] > best) { best = sc[u]; ibest = u; }
                stack[sp++] = u;
            }
            if (ibest < 0) ibest = v0;          // all scores <= 0: the reference would index [-1]; keep the root instead
            float* o = out + nout * 6;
            o[0] = bx[ibest][0]; o[1] = bx[ibest][1]; o[2] = bx[ibest][2]; o[3] = bx[ibest][3];
            o[4] = sc[ibest]; o[5] = (float)cl[ibest];
            osrc[nout] = sel[ibest];
            ++nout;
        }
        a.out_count[b] = nout;
        a.err[b] = nerr;
        if (nerr && a.counters) atomicAdd(a.counters, nerr);
    }
}

hipError_t launch_iou_merge(const MergeArgs& a, hipStream_t s) {
    if (a.max_det > MAXDET) return hipErrorInvalidValue;
    hipLaunchKernelGGL(iou_merge_kernel, dim3(a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ gathered records -> merge input
// After the all-gather rank 0 holds [R ranks][rows][300*6 + 3] floats (detections | count | status | tile id).  The cross-tile merge
// wants the valid detections in tile-id order on the host.  Two launches compact them on device, so that only the detections cross
// PCIe: (1) one workgroup walks the tiles in tile-id order (perm[t] = row of tile t), writes count (0 for a rejected tile) and status
// per tile and the exclusive prefix of the counts, (2) one workgroup per tile copies its rows to out[prefix[t] ..].
__global__ __launch_bounds__(1024) void records_scan_kernel(const float* __restrict__ g, long long rows, const long long* __restrict__ perm, int T, int stride,
                                                            int* __restrict__ hdr) {        // hdr: [T] counts | [T] status | [T] prefix | total
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int t = base + (int)threadIdx.x;
        int cnt = 0;
        if (t < T) {
            const long long pr = perm[t];
            int st = -1;                                       // (CY_ERR_ARG) a row index outside the gathered buffer: the tile counts as rejected
            if (pr >= 0 && pr < rows) {
                const float* row = g + pr * (long long)stride;
                st = (int)row[stride - 2];
                cnt = st == 0 ? (int)row[stride - 3] : 0;
                cnt = cnt < 0 ? 0 : (cnt > 300 ? 300 : cnt);
            }
            hdr[t] = cnt; hdr[T + t] = st;
        }
        part[threadIdx.x] = cnt;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                   // inclusive scan of the 1024 counts of this round
            const int v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (t < T) hdr[2 * T + t] = carry + part[threadIdx.x] - cnt;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) hdr[3 * T] = carry;
}

__global__ __launch_bounds__(64) void records_copy_kernel(const float* __restrict__ g, const long long* __restrict__ perm, int T, int stride,
                                                          const int* __restrict__ hdr, float* __restrict__ out) {
    const int t = blockIdx.x;
    const int n = hdr[t] * 6;                                 // 0 for a rejected tile or a row index out of range (records_scan_kernel)
    if (n == 0) return;
    const float* row = g + perm[t] * (long long)stride;
    float* dst = out + (long long)hdr[2 * T + t] * 6;
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = row[i];
}

hipError_t launch_compact_records(const float* g, long long rows, const long long* perm, int T, int stride, int* hdr, float* out, hipStream_t s) {
    if (T < 1 || stride < 9 || rows < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(records_scan_kernel, dim3(1), dim3(1024), 0, s, g, rows, perm, T, stride, hdr);
    hipLaunchKernelGGL(records_copy_kernel, dim3(T), dim3(64), 0, s, g, perm, T, stride, (const int*)hdr, out);
    return hipGetLastError();
}

}  // namespace cy
