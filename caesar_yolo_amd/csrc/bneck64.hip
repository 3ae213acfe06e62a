// Fused Bottleneck(64 -> 64 -> 64, two 3x3 stride-1 Conv+BN+SiLU, optional shortcut) on NHWC fp16 channel slices (gfx950).
//
// Replaces two launches of the C2f bottleneck at the network's stride-4 level (ultralytics `Bottleneck.forward`:
// x + cv2(cv1(x)); graph: SURVEY.md Appendix A.1 step 4, model.2.m.* of yolov8l), which are HBM-bound as separate convs:
// per 512x512 tile each bottleneck reads its 64-channel input, writes and re-reads the 64-channel intermediate and reads the
// input again as the residual (10.5 MB), where the data it actually needs is input + output (4.2 MB).  Here the
// intermediate never leaves the CU:
//
//   workgroup (4 waves, or 8 with CY_BNECK_WAVES=8; persistent: one per CU walking the patches) owns an 8 x 32-pixel output patch;
//   Y : the 12 x 36-pixel input halo, LDS-DMA, double-buffered (the next patch streams in under this patch's MFMAs);
//   T : cv1's output on the 10 x 34 pixels cv2 needs (zero outside the image = cv2's padding), written to LDS as fp16
//       exactly as the unfused path would round it to HBM;
//   cv2 reads T, adds the residual straight from the centre of Y, stores the 8 x 32 x 64 output.
//
// Both 3x3 weight panels (2 x 72 KB) do not fit in LDS next to Y and T, so the kernel is WEIGHT-STATIONARY IN REGISTERS:
// wave w owns 16 output channels (w & 3) and keeps the MFMA A-fragments of ONE conv at a time (18 k-steps x f16x8 = 72
// VGPRs; the other conv's set is re-fetched from L2 behind the last MFMA of a phase); the pixels are the B operand, read from
// LDS in one continuous stream of (k-step, row) units with the reads three or four units ahead of the MFMAs.
// Per patch: 8 x 414 (or 4 x 828) MFMAs = 13.2k cycles of each SIMD's matrix pipe, 12.9k LDS-array cycles, ~7k cycles of HBM
// time for the 87 KB of the patch.
//
// STATUS: OPT-IN (CY_BNECK_FUSE=1), NOT the default.  It is correct (tests/test_gpu_conv.py::test_fused_bottleneck64 and the
// forward parity test run it) but slower than the two launches it replaces: at batch 256 (16384 patches, 64 per CU) the two
// conv3x3_c64_kernel launches of a bottleneck take 0.43 + 0.48 = 0.91 ms (4.6 TB/s of HBM), this kernel 1.02-1.16 ms.
// Variants measured on MI355X (round 2), none under 1.0 ms:
//   4 waves x 16 ch, row pairs, reads one k-step ahead                       1.05 ms   (first version)
//   4 waves x 32 ch x half the rows (half the LDS reads), reads two ahead    1.05 ms   (so not LDS bandwidth, not read latency)
//   8 waves (two per SIMD) x 16 ch x half the rows, both weight sets held    1.02-1.08 ms  (256 VGPRs + 68 B of spills)
//   8 waves, deeper read-ahead rings (172 B of spills)                       1.40 ms
//   4 waves, streamed units, one weight set at a time (this file, default)   1.16 ms;  8 waves, same (280 B of spills): 1.78 ms
// Ablation of the 8-wave version (diagnostic build, CY_BK_DBG): no stores 0.93, no halo DMA 0.92, neither 0.83 (so memory is
// 0.19 ms of it and the stream is compute-bound); cv1 alone 0.47-0.56 ms, cv2 alone 0.33-0.43 ms, against 0.28 + 0.15 ms of pure
// matrix time: the MFMA pipe is ~50 % busy.  What is left: a third of cv1's MFMAs is wasted on the 16+16+2-column fragments of
// the 34-column T frame, 184 SiLUs per lane run on the same in-order issue port as the MFMAs, and at 256 VGPRs (two waves per
// SIMD) the register-resident weights leave no room for the accumulators + read ring without spills.  Next idea: cv1 and cv2
// on different waves of a SIMD (producer / consumer on a double-buffered T, 4 x 32-pixel patches so that it fits in LDS).
#include "cy_kernels.h"
#include <cstring>

namespace cy {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr unsigned OOB = 0xFFFFFF00u;
constexpr int OH = 8, OW = 32;                    // output patch
constexpr int TH = OH + 2, TP = 36;               // T: 10 rows, pitch 36 (34 used): 2 * pitch = 0 mod 8 keeps the swizzle phase of a row pair
constexpr int YH = OH + 4, YP = 36;               // Y: 12 x 36
constexpr int YPIECES = 56, YBYTES = YPIECES * 1024;          // 448 rows of 128 B (432 used; the tail is zero-filled and read by masked lanes)
constexpr int TBYTES = TH * TP * 128;
constexpr int LDS_BYTES = 2 * YBYTES + TBYTES;    // 160768 <= 163840

__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
}  // namespace

template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void bneck64_kernel(const BneckArgs a) {
    constexpr int PH = NWAVE / 4, R1 = TH / PH, R2 = OH / PH;       // pixel halves; T rows / output rows per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    char* const Yl = smem;
    char* const Tl = smem + 2 * YBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cq = wave & 3, ph = wave >> 2;                  // channel quarter (16 channels = one MFMA block), pixel half (0 when NWAVE = 4)
    const int fr = lane & 15, fq = lane >> 4;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + OW - 1) / OW, tiles_y = (H + OH - 1) / OH;
    const int npatch = a.B * tiles_y * tiles_x;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.in_bytes, 0x00020000);

    // This wave's weight fragments of ONE conv at a time (18 k-steps x f16x8 = 72 VGPRs): cv1's while cv1 runs, reloaded with
    // cv2's behind cv1's last MFMA, and back behind cv2's last (147 KB per patch and workgroup from L2, against the 55 KB halo)
    // -- both sets at once (144 VGPRs) do not fit beside the accumulators and the read-ahead ring at two waves per SIMD.
    const f16x8* const wf1 = reinterpret_cast<const f16x8*>(a.wfrag) + (size_t)((0 * 4 + cq) * 18) * 64 + lane;
    const f16x8* const wf2 = reinterpret_cast<const f16x8*>(a.wfrag) + (size_t)((1 * 4 + cq) * 18) * 64 + lane;
    f16x8 w[18];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) w[ks] = wf1[ks * 64];
    float b1[4], b2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { b1[j] = a.bias1[cq * 16 + fq * 4 + j]; b2[j] = a.bias2[cq * 16 + fq * 4 + j]; }

    // LDS rows are pixels (64 ch x fp16 = 128 B), 16-byte chunk c of row r stored at chunk c ^ (r & 7) (conflict-free b128 reads
    // at any row offset).  A fragment read = lane (fr, fq) takes chunk 4*half + fq of row base + fr: with base known at compile
    // time only base & 7 matters, so eight per-lane offsets (x2 for the channel half) cover every tap of every fragment.
    unsigned lp[2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const unsigned o = (unsigned)(fr * 128 + ((fq ^ ((c + fr) & 7)) << 4));
        lp[0][c] = o; lp[1][c] = o ^ 64u;
    }

    auto dma_patch = [&](int buf, int pidx) {
        const int tx = pidx % tiles_x, ty = (pidx / tiles_x) % tiles_y, b = pidx / (tiles_x * tiles_y);
        const int y0 = ty * OH - 2, x0 = tx * OW - 2;
#pragma unroll
        for (int j = 0; j < YPIECES / NWAVE; ++j) {
            const int pc = j * NWAVE + wave;
            const int r = pc * 8 + (lane >> 3);
            const int ry = r / YP, rx = r - ry * YP;
            const int y = y0 + ry, x = x0 + rx;
            const int q = (lane & 7) ^ (r & 7);
            const bool ok = ry < YH && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const unsigned off = ok ? (unsigned)(((b * H + y) * W + x) * a.in_ct + a.in_coff + q * 8) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_void*)(Yl + buf * YBYTES + pc * 1024), 16, off, 0, 0, 0);
        }
    };

    int pidx = blockIdx.x;
    if (pidx < npatch) dma_patch(0, pidx);
    int it = 0;
    for (; pidx < npatch; pidx += gridDim.x, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this patch's halo, cv1's weights (and the previous patch's stores)
        __builtin_amdgcn_s_barrier();
        const int nxt = pidx + gridDim.x;
        if (nxt < npatch && !(a.dbg & 2)) dma_patch((it + 1) & 1, nxt);          // that buffer's last readers finished before the barrier above
        const char* Y = Yl + (it & 1) * YBYTES;
        const int tx = pidx % tiles_x, ty = (pidx / tiles_x) % tiles_y, b = pidx / (tiles_x * tiles_y);
        const int y0 = ty * OH, x0 = tx * OW;

        // ---------------- cv1: this wave's five T rows (5 ph .. 5 ph + 4; image rows y0 - 1 + ..), three fragments per row
        // (cols 0..47 of the T frame, 34 used).  ONE MFMA stream over units u = (k-step, row): the three fragment reads of unit
        // u + 3 are issued before the three MFMAs of unit u (register ring of four units), i.e. nine MFMAs = 144 cycles ahead:
        // the stream never waits for LDS, and there is no per-row prologue.
        if (!(a.dbg & 4)) {
            const char* Yw = Y + ph * (R1 * YP * 128);           // NWAVE = 8: 5 rows = 180 pixels, 180 & 7 = 4 -> the phase shift below
            const unsigned sh64 = (unsigned)ph << 6;             // rows of this window sit (5 * 36) & 7 = 4 phases later: slot ^ 4 = offset ^ 64
            f32x4 acc[R1][3];
#pragma unroll
            for (int r = 0; r < R1; ++r)
#pragma unroll
                for (int f = 0; f < 3; ++f) acc[r][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int D = 3, NU = 18 * R1;
            f16x8 ring[D + 1][3];
            auto rd = [&](int u, f16x8* dst) {
                const int ks = u / R1, r = u % R1, tap = ks >> 1, half = ks & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const int base = (r + kh) * YP + f * 16 + kw;                      // compile-time
                    dst[f] = *reinterpret_cast<const f16x8*>(Yw + base * 128 + (lp[half][base & 7] ^ sh64));
                }
            };
#pragma unroll
            for (int u = 0; u < D; ++u) rd(u, ring[u]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + D < NU) rd(u + D, ring[(u + D) % (D + 1)]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < 3; ++f)
                    acc[u % R1][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u / R1], ring[u % (D + 1)][f], acc[u % R1][f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // cv2's weights: requested now, they land under the epilogue below and the barrier
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) w[ks] = wf2[ks * 64];
            // bias + SiLU -> fp16 -> T (zero where the pixel lies outside the image: cv2 pads with zeros, not with cv1 of padding)
#pragma unroll
            for (int r = 0; r < R1; ++r) {
                const int trow = R1 * ph + r;
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const int tcol = f * 16 + fr;
                    const int iy = y0 - 1 + trow, ix = x0 - 1 + tcol;
                    const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    if (tcol < 34) {
                        const int row = trow * TP + tcol;
                        f16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = inside ? (f16)silu_fast(acc[r][f][j] + b1[j]) : (f16)0.0f;
                        const int chunk = (2 * cq + (fq >> 1)) ^ (row & 7);
                        *reinterpret_cast<f16x4*>(Tl + row * 128 + chunk * 16 + (fq & 1) * 8) = o;
                    }
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) w[ks] = wf2[ks * 64];
        }
        __syncthreads();                                          // T complete

        // ---------------- cv2: this wave's four output rows (4 ph .. 4 ph + 3), two fragments per row; same stream form
        // (units of two MFMAs, reads four units = 128 cycles ahead); + bias, SiLU, + residual (centre of Y), store
        if (!(a.dbg & 8)) {
            const char* Tw = Tl + ph * (R2 * TP * 128);          // NWAVE = 8: 4 rows = 144 pixels, 144 & 7 = 0 -> no phase shift
            f32x4 acc[R2][2];
#pragma unroll
            for (int r = 0; r < R2; ++r)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[r][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int D = 4, NU = 18 * R2;
            f16x8 ring[D + 1][2];
            auto rd = [&](int u, f16x8* dst) {
                const int ks = u / R2, r = u % R2, tap = ks >> 1, half = ks & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int base = (r + kh) * TP + f * 16 + kw;
                    dst[f] = *reinterpret_cast<const f16x8*>(Tw + base * 128 + lp[half][base & 7]);
                }
            };
#pragma unroll
            for (int u = 0; u < D; ++u) rd(u, ring[u]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + D < NU) rd(u + D, ring[(u + D) % (D + 1)]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < 2; ++f)
                    acc[u % R2][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u / R2], ring[u % (D + 1)][f], acc[u % R2][f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // cv1's weights for the next patch
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) w[ks] = wf1[ks * 64];
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                const int orow = R2 * ph + r;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int ocol = f * 16 + fr;
                    const int iy = y0 + orow, ix = x0 + ocol;
                    if (iy >= H || ix >= W) continue;
                    const long pix = ((long)b * H + iy) * W + ix;
                    const int row = (orow + 2) * YP + ocol + 2;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = silu_fast(acc[r][f][j] + b2[j]);
                    if (a.shortcut) {
                        const int chunk = (2 * cq + (fq >> 1)) ^ (row & 7);
                        const f16x4 rv = *reinterpret_cast<const f16x4*>(Y + row * 128 + chunk * 16 + (fq & 1) * 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
                    }
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (f16)v[j];
                    if (!(a.dbg & 1)) *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cq * 16 + fq * 4) = o;
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) w[ks] = wf1[ks * 64];
        }
        // (the loop-top barrier of the next patch separates these T / Y reads from the next writes)
    }
}

hipError_t launch_bneck64(const BneckArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(bneck64_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(bneck64_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int npatch = a.B * ((a.H + OH - 1) / OH) * ((a.W + OW - 1) / OW);
    int cus = 256;
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    const int grid = npatch < cus ? npatch : cus;                  // one persistent workgroup per CU
    BneckArgs b2 = a;
    b2.dbg = dev_knob("CY_BK_DBG", 0);          // ablation bits of diagnostic builds: 1 no stores, 2 no halo DMA after the first, 4 no cv1, 8 no cv2
    static const int nwave = env_knob("CY_BNECK_WAVES", 4);
    if (nwave == 8) hipLaunchKernelGGL(bneck64_kernel<8>, dim3(grid), dim3(512), LDS_BYTES, s, b2);
    else hipLaunchKernelGGL(bneck64_kernel<4>, dim3(grid), dim3(256), LDS_BYTES, s, b2);
    return hipGetLastError();
}

// W1, W2: [64][64][3][3] fp32 (folded Conv+BN) -> dst[conv][block nb][k-step ks][lane] x 8 halves:
// lane (fr = l & 15, fq = l >> 4) of k-step ks = 2*tap + half holds W[16 nb + fr][32 half + 8 fq + j][tap], j = 0..7
// (the A operand of v_mfma_f32_16x16x32_f16: row fr, k = 8 fq + j)
void pack_bneck_weights(const float* W1, const float* W2, void* dst) {
    f16* o = reinterpret_cast<f16*>(dst);
    for (int c = 0; c < 2; ++c) {
        const float* Wc = c ? W2 : W1;
        for (int nb = 0; nb < 4; ++nb)
            for (int ks = 0; ks < 18; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int fr = l & 15, fq = l >> 4, tap = ks >> 1, half = ks & 1;
                        const int ch = 16 * nb + fr, cin = 32 * half + 8 * fq + j;
                        o[((((size_t)c * 4 + nb) * 18 + ks) * 64 + l) * 8 + j] = (f16)Wc[((size_t)ch * 64 + cin) * 9 + tap];
                    }
    }
}

}  // namespace cy
