// YOLO11-only operators: depth-wise 3x3 convolution and the C2PSA attention core.
// Both are a fraction of a percent of the network's work (the detect head's depth-wise branch moves ~100 MB per 64-tile
// batch, attention runs on the 16x16 stride-32 map only), so they are written for clarity: HBM-bound element kernels in
// the activation type with fp32 arithmetic, not MFMA tiles.  Definitions follow the public ultralytics modules
// (Conv with g = c, Attention in nn/modules/block.py); parity is against oracle/yolo11_ref.py (unpinned to ultralytics).
#include "cy_kernels.h"
#include <cstdlib>

namespace cy {

typedef _Float16 f16;

template <typename T> struct Vec8;
template <> struct Vec8<f16> { typedef f16 type __attribute__((ext_vector_type(8))); };
template <> struct Vec8<float> { typedef float type __attribute__((ext_vector_type(8))); };

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// fp16 context: hardware exp2 / rcp (1 ulp each; the result is rounded to fp16 anyway), as in the convolution epilogues
__device__ __forceinline__ float silu_fast_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f)); }
// fp16x3 context (SPLIT): a value is the sum of two fp16 halves, the low one `lo` elements behind the high one (cy_kernels.h)
template <bool SPLIT, typename T> __device__ __forceinline__ float ld1(const T* p, int lo) {
    if constexpr (SPLIT) return (float)p[0] + (float)p[lo]; else return (float)p[0];
}
template <bool SPLIT, typename T> __device__ __forceinline__ void st1(T* p, int lo, float v) {
    if constexpr (SPLIT) { const T h = (T)v; p[0] = h; p[lo] = (T)(v - (float)h); } else p[0] = (T)v;
}

// ------------------------------------------------------------------------------------------------ depth-wise 3x3, stride 1
// out[b,y,x,c] = act( sum_{kh,kw} in[b,y+kh-1,x+kw-1,map(c)] * w[kh*3+kw][c] + bias[c] ) (+ res[b,y,x,c])
// One thread = DW_PX pixels of a row x 8 channels (16-byte loads in fp16).  map(c) is the identity, or - for the positional-encoding
// conv of the attention block - the channel of `v` inside the per-head [q|k|v] blocks of the qkv tensor:
// map(c) = (c / blk) * gstride + goff + c % blk  (blk is a multiple of 8).
// A thread owns DW_PX consecutive pixels of a row: the 3 x (DW_PX + 2) input vectors and the 9 x 8 weights are loaded once for
// DW_PX outputs (18 + 18 loads for four pixels instead of 36 + 72).  Per output pixel the products are added in the same
// (bias, kh, kw) order as a one-pixel-per-thread walk, so the result does not depend on DW_PX.  (Weights and bias staged in LDS
// per workgroup instead of read through the vector cache: 7-9 % slower.)
constexpr int DW_PX = 4;
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const DwArgs a) {
    typedef typename Vec8<T>::type v8;
    const int cg = a.C / 8, xb = (a.W + DW_PX - 1) / DW_PX;
    const long total = (long)a.B * a.H * xb * cg;
    // the launch is one pass of whole workgroups; consecutive workgroups (rows y, y + 1 of an image) go to different XCDs, whose L2s
    // would each fetch the three input rows of their outputs (PMC: 3.0 x the written bytes fetched): XCD-contiguous order instead
    const long wg = a.xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    for (long idx = wg * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int g8 = (int)(idx % cg);
        const long pb = idx / cg;
        const int x0 = (int)(pb % xb) * DW_PX, y = (int)((pb / xb) % a.H), b = (int)(pb / ((long)xb * a.H));
        const int c = g8 * 8;
        const int cin = a.blk ? (c / a.blk) * a.gstride + a.goff + c % a.blk : c;
        float acc[DW_PX][8];
#pragma unroll
        for (int i = 0; i < DW_PX; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = a.bias[c + j];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            // loads at clamped coordinates, unconditional: written as `if (inside) load`, every load sat in its own branch with a
            // full vmcnt(0) wait behind it (18 serialised round trips per thread); the validity only gates the FMAs
            const int yy = y + kh - 1;
            const bool rowok = (unsigned)yy < (unsigned)a.H;
            const int yyc = yy < 0 ? 0 : (yy >= a.H ? a.H - 1 : yy);
            float w[3][8];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int j = 0; j < 8; ++j) w[kw][j] = a.w[(kh * 3 + kw) * a.C + c + j];
            float in[DW_PX + 2][8];
            bool ok[DW_PX + 2];
            const T* row = reinterpret_cast<const T*>(a.in) + (((long)b * a.H + yyc) * a.W) * a.in_ct + a.in_coff + cin;
            v8 raw[DW_PX + 2], rawl[SPLIT ? DW_PX + 2 : 1];
#pragma unroll
            for (int t = 0; t < DW_PX + 2; ++t) {
                const int xx = x0 + t - 1;
                ok[t] = rowok && (unsigned)xx < (unsigned)a.W;
                const int xxc = xx < 0 ? 0 : (xx >= a.W ? a.W - 1 : xx);
                const T* ip = row + (long)xxc * a.in_ct;
                raw[t] = *reinterpret_cast<const v8*>(ip);
                if constexpr (SPLIT) rawl[t] = *reinterpret_cast<const v8*>(ip + a.in_lo);
            }
#pragma unroll
            for (int t = 0; t < DW_PX + 2; ++t) {
                if constexpr (SPLIT) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) in[t][j] = (float)raw[t][j] + (float)rawl[t][j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) in[t][j] = (float)raw[t][j];
                }
            }
#pragma unroll
            for (int i = 0; i < DW_PX; ++i)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if (ok[i + kw]) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(in[i + kw][j], w[kw][j], acc[i][j]);
                    }
        }
#pragma unroll
        for (int i = 0; i < DW_PX; ++i) {
            const int x = x0 + i;
            if (x >= a.W) break;
            v8 o;
            const long opix = ((long)b * a.H + y) * a.W + x;
            if constexpr (SPLIT) {
                float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (a.res) {
                    const T* rp = reinterpret_cast<const T*>(a.res) + opix * a.res_ct + a.res_coff + c;
                    const v8 r = *reinterpret_cast<const v8*>(rp), rl = *reinterpret_cast<const v8*>(rp + a.res_lo);
#pragma unroll
                    for (int j = 0; j < 8; ++j) rr[j] = (float)r[j] + (float)rl[j];
                }
                v8 ol;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = (a.act ? silu_f(acc[i][j]) : acc[i][j]) + rr[j];
                    o[j] = (T)v; ol[j] = (T)(v - (float)o[j]);
                }
                T* op = reinterpret_cast<T*>(a.out) + opix * a.out_ct + a.out_coff + c;
                *reinterpret_cast<v8*>(op) = o;
                *reinterpret_cast<v8*>(op + a.out_lo) = ol;
                continue;
            }
            // fp16 context: hardware exp2 / rcp as in the convolution epilogues; exact division in the fp32 context
            if (a.res) {
                const v8 r = *reinterpret_cast<const v8*>(reinterpret_cast<const T*>(a.res) + opix * a.res_ct + a.res_coff + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (T)((a.act ? (sizeof(T) == 2 ? silu_fast_f(acc[i][j]) : silu_f(acc[i][j])) : acc[i][j]) + (float)r[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (T)(a.act ? (sizeof(T) == 2 ? silu_fast_f(acc[i][j]) : silu_f(acc[i][j])) : acc[i][j]);
            }
            *reinterpret_cast<v8*>(reinterpret_cast<T*>(a.out) + opix * a.out_ct + a.out_coff + c) = o;
        }
    }
}

hipError_t launch_dwconv(Precision p, const DwArgs& a, hipStream_t s) {
    if (a.C % 8 || (a.blk && a.blk % 8)) return hipErrorInvalidValue;
    const long total = (long)a.B * a.H * ((a.W + DW_PX - 1) / DW_PX) * (a.C / 8);
    // one pass of whole workgroups (a capped grid with a grid-stride loop leaves a ragged second round)
    const long blocks = (total + 255) / 256;
    const int grid = (int)(blocks < (1L << 22) ? blocks : (1L << 22));
    DwArgs a2 = a;
    a2.xcd = blocks == grid && env_knob("CY_XCD_ORDER", 1);        // single pass only (the remap is a permutation of the launched workgroups)
    if (p == PREC_F16) hipLaunchKernelGGL(dwconv3x3_kernel<f16>, dim3(grid), dim3(256), 0, s, a2);
    else if (p == PREC_F16X3) hipLaunchKernelGGL((dwconv3x3_kernel<f16, true>), dim3(grid), dim3(256), 0, s, a2);
    else hipLaunchKernelGGL(dwconv3x3_kernel<float>, dim3(grid), dim3(256), 0, s, a2);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ attention core (C2PSA)
// qkv: [B][N][heads*(2*kd+hd)] (N = H*W pixels of the stride-32 map), per head the channels [q: kd | k: kd | v: hd].
// out[b][n][head*hd + c] = sum_m softmax_m( scale * <q_n, k_m> ) * v[m][c],  scale = kd^-0.5.
// One wave per query: lanes stride over the keys for the scores (kept in LDS), wave-reduce max and sum, then lane c
// accumulates channel c over all keys (hd = 64 = one channel per lane).  fp32 arithmetic throughout.
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
    extern __shared__ float sc[];                            // [4 waves][N]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = a.kd * 2 + a.hd;
    const long q_id = (long)blockIdx.x * 4 + wave;           // (b, head, n)
    const long nq = (long)a.B * a.heads * a.N;
    if (q_id >= nq) return;
    const int n = (int)(q_id % a.N), head = (int)((q_id / a.N) % a.heads), b = (int)(q_id / ((long)a.N * a.heads));
    const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * a.N * a.ct + a.coff + head * per;
    float* s = sc + wave * a.N;
    float q[64];                                             // kd <= 64
    const T* qp = base + (long)n * a.ct;
    for (int d = 0; d < a.kd; ++d) q[d] = ld1<SPLIT>(qp + d, a.lo);
    float mx = -INFINITY;
    for (int m = lane; m < a.N; m += 64) {
        const T* kp = base + (long)m * a.ct + a.kd;
        float t = 0.0f;
        for (int d = 0; d < a.kd; ++d) t = fmaf(q[d], ld1<SPLIT>(kp + d, a.lo), t);
        t *= a.scale;
        s[m] = t;
        mx = fmaxf(mx, t);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
    for (int m = lane; m < a.N; m += 64) { const float e = expf(s[m] - mx); s[m] = e; sum += e; }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // this wave's score row is complete in LDS
    const float inv = 1.0f / sum;
    for (int c = lane; c < a.hd; c += 64) {
        float acc = 0.0f;
        const T* vp = base + 2 * a.kd + c;
        for (int m = 0; m < a.N; ++m) acc = fmaf(s[m], ld1<SPLIT>(vp + (long)m * a.ct, a.lo), acc);
        st1<SPLIT>(reinterpret_cast<T*>(a.out) + ((long)b * a.N + n) * a.out_ct + a.out_coff + head * a.hd + c, a.out_lo, acc * inv);
    }
}

// Fast path: one workgroup per (image, head); K and V of that head are staged in LDS once (fp32), each thread owns one
// query (or a few, for maps with more than 256 pixels) and walks the keys with an online softmax, reading K[m] and V[m] as
// wave-wide broadcasts.  256 tokens: 96 KiB of LDS, 0.15 ms for all heads of a 64-image batch (the per-query kernel above
// re-reads K and V from L2 for every query and took 1.4 ms); used whenever K and V fit in 160 KiB.
template <typename T, int KD, int HD, bool SPLIT = false>
__global__ __launch_bounds__(256) void attention_lds_kernel(const AttnArgs a) {
    extern __shared__ float kv[];                            // K [N][KD] | V [N][HD]
    float* Ks = kv;
    float* Vs = kv + (size_t)a.N * KD;
    const int head = blockIdx.x % a.heads, b = blockIdx.x / a.heads;
    constexpr int PER = 2 * KD + HD;
    const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * a.N * a.ct + a.coff + head * PER;
    for (int i = threadIdx.x; i < a.N * (KD / 8); i += 256) {
        const int m = i / (KD / 8), c8 = (i % (KD / 8)) * 8;
        const T* p = base + (long)m * a.ct + KD + c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) Ks[m * KD + c8 + j] = ld1<SPLIT>(p + j, a.lo);
    }
    for (int i = threadIdx.x; i < a.N * (HD / 8); i += 256) {
        const int m = i / (HD / 8), c8 = (i % (HD / 8)) * 8;
        const T* p = base + (long)m * a.ct + 2 * KD + c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) Vs[m * HD + c8 + j] = ld1<SPLIT>(p + j, a.lo);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < a.N; n += 256) {
        float q[KD], o[HD];
        const T* qp = base + (long)n * a.ct;
#pragma unroll
        for (int d = 0; d < KD; ++d) q[d] = ld1<SPLIT>(qp + d, a.lo) * a.scale;
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = 0.0f;
        float mx = -INFINITY, sum = 0.0f;
        for (int m = 0; m < a.N; ++m) {
            const float* kp = Ks + m * KD;
            float t = 0.0f;
#pragma unroll
            for (int d = 0; d < KD; ++d) t = fmaf(q[d], kp[d], t);
            if (t > mx) {                                    // rescale the running sums to the new maximum
                const float corr = expf(mx - t);
                sum *= corr;
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] *= corr;
                mx = t;
            }
            const float e = expf(t - mx);
            sum += e;
            const float* vp = Vs + m * HD;
#pragma unroll
            for (int c = 0; c < HD; ++c) o[c] = fmaf(e, vp[c], o[c]);
        }
        const float inv = 1.0f / sum;
        T* op = reinterpret_cast<T*>(a.out) + ((long)b * a.N + n) * a.out_ct + a.out_coff + head * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) st1<SPLIT>(op + c, a.out_lo, o[c] * inv);
    }
}

hipError_t launch_attention(Precision p, const AttnArgs& a, hipStream_t s) {
    const size_t fast_lds = (size_t)a.N * (a.kd + a.hd) * sizeof(float);
    if (a.kd == 32 && a.hd == 64 && fast_lds <= 160 * 1024 && !(getenv("CY_ATTN_SLOW") && atoi(getenv("CY_ATTN_SLOW")))) {
        static bool fast_attr = false;
        if (!fast_attr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attention_lds_kernel<f16, 32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(attention_lds_kernel<float, 32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(attention_lds_kernel<f16, 32, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            fast_attr = true;
        }
        const int grid = a.B * a.heads;
        if (p == PREC_F16) hipLaunchKernelGGL((attention_lds_kernel<f16, 32, 64>), dim3(grid), dim3(256), fast_lds, s, a);
        else if (p == PREC_F16X3) hipLaunchKernelGGL((attention_lds_kernel<f16, 32, 64, true>), dim3(grid), dim3(256), fast_lds, s, a);
        else hipLaunchKernelGGL((attention_lds_kernel<float, 32, 64>), dim3(grid), dim3(256), fast_lds, s, a);
        return hipGetLastError();
    }
    if (a.kd > 64 || a.kd < 1 || a.hd < 1 || a.N < 1 || (size_t)a.N * 16 > 160 * 1024) return hipErrorInvalidValue;
    const long nq = (long)a.B * a.heads * a.N;
    const size_t lds = (size_t)4 * a.N * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<f16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = (int)((nq + 3) / 4);
    if (p == PREC_F16) hipLaunchKernelGGL(attention_kernel<f16>, dim3(grid), dim3(256), lds, s, a);
    else if (p == PREC_F16X3) hipLaunchKernelGGL((attention_kernel<f16, true>), dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(attention_kernel<float>, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace cy
