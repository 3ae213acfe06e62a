// Fused Conv2d + folded-BN bias + SiLU (+ residual, + concat/upsample gather, + channel-slice store) as an
// implicit GEMM on CDNA4 matrix cores (gfx950 only).
//
// Replaces ultralytics `Conv.forward_fuse` / `Bottleneck` / `Concat` / `nn.Upsample` as executed inside the
// reference's model call (caesar_yolo/evaluation.py:181-193; graph: SURVEY.md Appendix A.1 step 4, Appendix B).
//
// GEMM view (per launch): D[n, m] = sum_k W[n, k] * X[m, k]
//   m = output pixel (b, ho, wo) over the whole tile batch, n = output channel, k = (tap, input channel).
//   The weights are the MFMA "A" operand and the im2col rows the "B" operand, so an accumulator lane holds
//   FOUR CONSECUTIVE CHANNELS of ONE pixel: with the 64-row weight permutation done at pack time a lane owns 16
//   contiguous channels of a pixel and the NHWC store is 16-byte vectors (no LDS transpose in the epilogue).
//   f16: v_mfma_f32_16x16x32_f16 (fp32 accumulate).  f32 (parity mode): v_mfma_f32_16x16x4_f32 = exact fp32 FMA chain.
//
// Data movement: K is walked in 128-byte slabs per row (64 halves / 32 floats of one filter tap).  Each lane moves
// 16-byte pieces global -> LDS directly (`buffer_load_dwordx4 ... lds`, no VGPR staging, no ds_write); the buffer
// range check supplies the zero padding of the 3x3 halo and of ragged channel counts (an out-of-range offset
// writes 0).  The LDS image is double-buffered and XOR-swizzled through the SOURCE address so that ds_read_b128
// fragment reads are conflict-free (slot = chunk ^ (row & 7): conflict-free for any 16 consecutive rows, whatever the first row -- the 3x3 taps of the
// halo kernels read at arbitrary row offsets; a 256-byte bank row holds two 128-byte tile rows).
#include "cy_kernels.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace cy {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CY_OOB 0xFFFFFF00u
#define CY_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// developer diagnostics: per-segment s_memtime sums of the 3x3 kernels' stage loop, summed over waves.  Compiled in only
// with -DCY_STAMPS_ENABLED=1 (then enabled at run time by CY_DBG bit 6); a stamped build is for SHARES, not for timing.
#ifndef CY_STAMPS_ENABLED
#define CY_STAMPS_ENABLED 0
#endif
__device__ unsigned long long g_stamps[8];
constexpr int WG_STAMP_SLOTS = 4096;
__device__ unsigned long long g_wg_stamps[WG_STAMP_SLOTS * 4];     // per-workgroup phase records of the wide kernel (stamped builds)
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long stamp_real() {       // constant 100 MHz counter: wall time in units of 10 ns
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }
// fp16 context: x * sigmoid(x) with the hardware exp2/rcp (1 ulp each; the result is rounded to fp16 anyway):
// 5 VALU instructions per element instead of the ~50 of an IEEE-exact division
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
// The epilogue of the fp16 kernels for the 16 values a lane holds of one pixel (four accumulators of four channels): bias +
// SiLU with two values per VALU instruction where the ISA has a packed fp32 form (add, mul; exp2 and rcp stay scalar), and
// the activation switch as ONE uniform branch (written per value it becomes a v_cndmask per value behind an unconditional SiLU).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bias_act16(const f32x4& a0, const f32x4& a1, const f32x4& a2, const f32x4& a3, const float (&bv)[16],
                                           bool act, float (&v)[16]) {
    const f32x4 acc[4] = {a0, a1, a2, a3};
    if (act) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 t = f32x2{acc[ni][2 * h], acc[ni][2 * h + 1]} + f32x2{bv[ni * 4 + 2 * h], bv[ni * 4 + 2 * h + 1]};
                f32x2 e = t * f32x2{-1.44269504088896341f, -1.44269504088896341f};
                e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
                t = t * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                v[ni * 4 + 2 * h] = t[0]; v[ni * 4 + 2 * h + 1] = t[1];
            }
    } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[ni * 4 + j] = acc[ni][j] + bv[ni * 4 + j];
    }
}

// fp16x3 context: the same epilogue with the accumulator first multiplied by the power of two that undoes the weight scale of its
// output channel, and the result stored as two fp16 halves hi = fp16(v), lo = fp16(v - hi) (lo_off halves behind hi)
__device__ __forceinline__ void scale_bias_act16(const f32x4& a0, const f32x4& a1, const f32x4& a2, const f32x4& a3, const float (&bv)[16],
                                                 const float (&sc)[16], bool act, float (&v)[16]) {
    const f32x4 acc[4] = {a0, a1, a2, a3};
    if (act) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 t = f32x2{acc[ni][2 * h], acc[ni][2 * h + 1]} * f32x2{sc[ni * 4 + 2 * h], sc[ni * 4 + 2 * h + 1]} +
                          f32x2{bv[ni * 4 + 2 * h], bv[ni * 4 + 2 * h + 1]};
                f32x2 e = t * f32x2{-1.44269504088896341f, -1.44269504088896341f};
                e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
                t = t * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                v[ni * 4 + 2 * h] = t[0]; v[ni * 4 + 2 * h + 1] = t[1];
            }
    } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[ni * 4 + j] = acc[ni][j] * sc[ni * 4 + j] + bv[ni * 4 + j];
    }
}
__device__ __forceinline__ void store_split16(f16* dst, int lo_off, const float (&v)[16]) {
    f16x8 h0, h1, l0, l1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h0[j] = (f16)v[j]; h1[j] = (f16)v[8 + j];
        l0[j] = (f16)(v[j] - (float)h0[j]); l1[j] = (f16)(v[8 + j] - (float)h1[j]);
    }
    *reinterpret_cast<f16x8*>(dst) = h0;
    *reinterpret_cast<f16x8*>(dst + 8) = h1;
    *reinterpret_cast<f16x8*>(dst + lo_off) = l0;
    *reinterpret_cast<f16x8*>(dst + lo_off + 8) = l1;
}
__device__ __forceinline__ void store_split1(f16* dst, int lo_off, float v) {
    const f16 h = (f16)v;
    dst[0] = h; dst[lo_off] = (f16)(v - (float)h);
}
// virtual K chunk of the fp16x3 passes [x_lo | x_hi | x_hi] (against weights [w_hi | w_lo | w_hi]) -> physical chunk; lo = 1 in the
// first pass.  The two small cross terms come FIRST: the fp32 accumulator of the MFMA rounds at every step by an amount relative
// to its current magnitude, so they are summed while it is still ~2^-11 of the final value (their rounding is then negligible)
// and the x_hi * w_hi chain runs last, exactly as long as in the fp16 context.
__device__ __forceinline__ int x3_chunk(int v, int per_pass, int& lo) {
    lo = v < per_pass;
    if (v >= 2 * per_pass) return v - 2 * per_pass;
    if (v >= per_pass) return v - per_pass;
    return v;
}

template <typename T> struct Elem;
template <> struct Elem<f16> { static constexpr int BKE = 64, EPC = 8, ES = 2; };
template <> struct Elem<float> { static constexpr int BKE = 32, EPC = 4, ES = 4; };

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles so that
    // neighbouring tiles (same activation rows, different channel blocks) hit one L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// NSTAGE = 2: the slab for step s+1 is requested while step s computes and drained at the barrier (two workgroups of
// four waves per CU hide each other's drains).  NSTAGE = 3 (the 256x128 tile, eight waves, one workgroup per CU): slabs
// are requested TWO steps ahead and the wait before a barrier is a counted vmcnt that leaves the newest request in
// flight, so an L2 round trip (about one step's worth of MFMA time for K-slabs of 64) is off the critical path.
// Non-template wrappers: inside a template kernel a buffer builtin whose soffset is not a constant makes this clang drop
// the kernel's host-side instantiation without a diagnostic; called through these, the builtin is never value-dependent.
typedef __attribute__((address_space(3))) void lds_ptr_t;
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, lds_ptr_t* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);     // soffset is NOT range-checked (voffset is)
}
__device__ __forceinline__ u32x4 load_b128(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}

template <typename T, int WM, int WN, int MI, int NSTAGE = 2, bool SPLIT = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 2 : 1)) void conv_igemm_kernel(const ConvArgs a) {
    static_assert(!SPLIT || sizeof(T) == 2, "fp16x3: fp16 operands");
    constexpr int BM = WM * MI * 16, BN = WN * 64, NT = WM * WN * 64, RPR = NT / 8;
    constexpr int BKE = Elem<T>::BKE, EPC = Elem<T>::EPC, ES = Elem<T>::ES;
    constexpr int AROWS = BM / RPR, BROWS = BN / RPR;        // rows each thread stages per slab (RPR rows per round)
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int DIST = NSTAGE - 1, PER = AROWS + BROWS;    // slabs in flight; DMA instructions per slab per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int M = a.B * a.Ho * a.Wo;
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;
    const int taps = a.k * a.k, cpad = pad128(a.Cout);
    const int pchunks = (a.Cin + BKE - 1) / BKE;             // K chunks of one pass over the input channels
    const int cchunks = SPLIT ? a.split * pchunks : pchunks; // fp16x3: three passes (x_lo w_hi, x_hi w_lo, x_hi w_hi), or two (x_lo w, x_hi w: fp16-exact weights)
    const int nslab = taps * cchunks;

    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c1 ? a.in1 : a.in0), 0,
                                                       a.c1 ? a.in1_bytes : a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt), 0, a.wgt_bytes, 0x00020000);

    // ---- per-thread staging geometry: chunk q of rows (tid>>3) + 32*i
    const int q = tid & 7, r0 = tid >> 3;
    int pix0[AROWS], pix1[AROWS];
    unsigned vmask[AROWS];
    const int pad = a.k >> 1;
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + r0 + RPR * i;
        unsigned vm = 0;
        int p0 = 0, p1 = 0;
        if (m < M) {
            const int b = m / HoWo, r = m - b * HoWo;
            const int ho = r / a.Wo, wo = r - ho * a.Wo;
            const int hi0 = ho * a.s - pad, wi0 = wo * a.s - pad;
            for (int kh = 0; kh < a.k; ++kh)
                for (int kw = 0; kw < a.k; ++kw)
                    if ((unsigned)(hi0 + kh) < (unsigned)a.Hi && (unsigned)(wi0 + kw) < (unsigned)a.Wi)
                        vm |= 1u << (kh * a.k + kw);
            if (a.up0) p0 = (b * (a.Hi >> 1) + (ho >> 1)) * (a.Wi >> 1) + (wo >> 1);
            else p0 = (b * a.Hi + hi0) * a.Wi + wi0;
            p1 = (b * a.Hi + hi0) * a.Wi + wi0;
        }
        pix0[i] = p0; pix1[i] = p1; vmask[i] = vm;
    }

    // LDS-DMA staging: `buffer_load_dwordx4 ... lds` writes wave-uniform base + lane*16, i.e. one wave-instruction
    // fills 8 consecutive 128-byte tile rows.  Row r = 8*wave + 32*i + (lane>>3), slot = lane&7 is exactly lane-linear,
    // so the XOR swizzle moves to the SOURCE: the lane that owns slot s of row r fetches logical chunk s ^ ((r>>1)&7)
    // (the same involution the fragment reads apply).  ((r>>1)&7 does not depend on i because 32*i>>1 is 0 mod 8.)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int cq = q ^ (r0 & 7);
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma = [&](int stage, int tap, int cc) {
        int lo = 0;
        const int ccp = SPLIT ? x3_chunk(cc, pchunks, lo) : cc;       // physical chunk of the input (the weights are packed per virtual chunk)
        const int lo0 = SPLIT && lo ? a.in0_lo : 0, lo1 = SPLIT && lo ? a.in1_lo : 0;
        const int c = ccp * BKE + cq * EPC;
        const int kh = tap / a.k, kw = tap - kh * a.k;
        const int dpix = kh * a.Wi + kw;
        const bool cin_ok = c < a.Cin;
        const bool seg1 = ccp * BKE >= a.c0;              // wave-uniform: segment boundaries are multiples of BKE
        char* A = smem + stage * STAGE + wave_u * (8 * 128);
        char* Bm = A + A_BYTES;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const bool ok = cin_ok && ((vmask[i] >> tap) & 1u);
            unsigned off;
            if (!seg1) off = (unsigned)((pix0[i] + (a.up0 ? 0 : dpix)) * a.in0_ct + a.in0_coff + lo0 + c) * ES;
            else       off = (unsigned)((pix1[i] + dpix) * a.in1_ct + a.in1_coff + lo1 + (c - a.c0)) * ES;
            off = ok ? off : CY_OOB;
            lds_void* dst = (lds_void*)(A + i * (RPR * 128));
            if (seg1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, off, 0, 0, 0);
            else      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const int n = n0 + r0 + RPR * i;
            const unsigned off = (unsigned)(((cc * taps + tap) * cpad + n) * 128 + cq * 16);      // rows are padded to 128: always valid
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(Bm + i * (RPR * 128)), 16, off, 0, 0, 0);
        }
    };

    f32x4 acc[4][MI];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int stage) {
        const char* A = smem + stage * STAGE;
        const char* Bm = A + A_BYTES;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int qf = fq + 4 * kk;
                f16x8 xa[MI], wb[4];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int r = wm * (MI * 16) + mi * 16 + fr;
                    xa[mi] = *reinterpret_cast<const f16x8*>(A + r * 128 + ((qf ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int r = wn * 64 + ni * 16 + fr;
                    wb[ni] = *reinterpret_cast<const f16x8*>(Bm + r * 128 + ((qf ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float xa[MI], wb[4];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int r = wm * (MI * 16) + mi * 16 + fr;
                    xa[mi] = *reinterpret_cast<const float*>(A + r * 128 + ((ks ^ (r & 7)) << 4) + fq * 4);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int r = wn * 64 + ni * 16 + fr;
                    wb[ni] = *reinterpret_cast<const float*>(Bm + r * 128 + ((ks ^ (r & 7)) << 4) + fq * 4);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
            }
        }
    };

    // ---- main loop: the DMA for slab s+1 is in flight while slab s is on the matrix pipe; one barrier per slab
    // (__syncthreads drains vmcnt, which is what orders the landed DMA before the next slab's ds_reads)
    if constexpr (NSTAGE == 2) {
        int tap = 0, cc = 0;
        dma(0, 0, 0);
        __syncthreads();
        for (int s = 0; s < nslab; ++s) {
            int ntap = tap, ncc = cc + 1;
            if (ncc == cchunks) { ncc = 0; ++ntap; }
            if ((s + 1) < nslab) dma((s + 1) & 1, ntap, ncc);
            compute(s & 1);
            __syncthreads();
            tap = ntap; cc = ncc;
        }
    } else {
        // ring of NSTAGE slabs, requests DIST steps ahead.  Slot (s+DIST) % NSTAGE last held slab s-1, whose readers are
        // all past the barrier that ended step s-1 when step s issues into it.
        int itap = 0, icc = 0, islot = 0;                    // issue cursor
        auto issue = [&]() {
            dma(islot, itap, icc);
            if (++icc == cchunks) { icc = 0; ++itap; }
            if (++islot == NSTAGE) islot = 0;
        };
#pragma unroll
        for (int d = 0; d < DIST; ++d)
            if (d < nslab) issue();
        if (nslab >= DIST) { CY_WAIT_VM((DIST - 1) * PER); } else { CY_WAIT_VM(0); }
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int s = 0; s < nslab; ++s) {
            const bool more = s + DIST < nslab;
            if (more) issue();
            compute(slot);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragment reads of this slot are done (WAR on the ring)
            if (more) { CY_WAIT_VM((DIST - 1) * PER); } else { CY_WAIT_VM(0); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (++slot == NSTAGE) slot = 0;
        }
    }

    // ---- epilogue: bias + SiLU (+ residual) and 16 contiguous channels per lane per pixel
    const int cbase = n0 + wn * 64 + fq * 16;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = a.bias[cbase + j];       // bias is padded to Cout_pad64
    float sc[SPLIT ? 16 : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[j] = a.oscale[cbase + j];
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * (MI * 16) + mi * 16 + fr;
        if (m >= M) continue;
        const int b = m / HoWo, r = m - b * HoWo;
        const long opix = (long)b * a.out_bs + a.out_ro + r;
        float v[16];
        if constexpr (SPLIT) {
            scale_bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bv, sc, a.act != 0, v);
            if (!a.out_f32) {
                f16* dst = reinterpret_cast<f16*>(a.out) + opix * a.out_ct + a.out_coff + cbase;
                const f16* rp = a.res ? reinterpret_cast<const f16*>(a.res) + (long)m * a.res_ct + a.res_coff + cbase : nullptr;
                if (cbase + 16 <= a.Cout) {
                    if (rp) {
                        const f16x8 r0v = *reinterpret_cast<const f16x8*>(rp), r1v = *reinterpret_cast<const f16x8*>(rp + 8);
                        const f16x8 q0v = *reinterpret_cast<const f16x8*>(rp + a.res_lo), q1v = *reinterpret_cast<const f16x8*>(rp + a.res_lo + 8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) { v[j] += (float)r0v[j] + (float)q0v[j]; v[8 + j] += (float)r1v[j] + (float)q1v[j]; }
                    }
                    store_split16(dst, a.out_lo, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (cbase + j >= a.Cout) continue;
                        float x = v[j];
                        if (rp) x += (float)rp[j] + (float)rp[a.res_lo + j];
                        store_split1(dst + j, a.out_lo, x);
                    }
                }
                continue;
            }
        } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = acc[ni][mi][j] + bv[ni * 4 + j];
                if (a.act) x = (sizeof(T) == 2) ? silu_fast(x) : silu_exact(x);
                v[ni * 4 + j] = x;
            }
        }
        const bool full = (cbase + 16 <= a.Cout) && !a.out_f32;
        if (full) {
            T* dst = reinterpret_cast<T*>(a.out) + opix * a.out_ct + a.out_coff + cbase;
            if (a.res) {
                const T* rp = reinterpret_cast<const T*>(a.res) + (long)m * a.res_ct + a.res_coff + cbase;
                if constexpr (sizeof(T) == 2) {
                    const f16x8 r0v = *reinterpret_cast<const f16x8*>(rp), r1v = *reinterpret_cast<const f16x8*>(rp + 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] += (float)r0v[j]; v[8 + j] += (float)r1v[j]; }
                } else {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(rp + 4 * j4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[4 * j4 + j] += rv[j];
                    }
                }
            }
            if constexpr (sizeof(T) == 2) {
                f16x8 o0, o1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
                *reinterpret_cast<f16x8*>(dst) = o0;
                *reinterpret_cast<f16x8*>(dst + 8) = o1;
            } else {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4)
                    *reinterpret_cast<f32x4*>(dst + 4 * j4) = f32x4{v[4 * j4], v[4 * j4 + 1], v[4 * j4 + 2], v[4 * j4 + 3]};
            }
        } else if (a.out_f32 && !a.res && cbase + 16 <= a.Cout) {
            // fp32 head output (box logits): rows of the prediction buffer are 64+nc floats, so only dword-aligned; four
            // 16-byte stores per lane through a 4-byte-aligned vector type instead of sixteen scattered dword stores
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            float* dst = reinterpret_cast<float*>(a.out) + opix * a.out_ct + a.out_coff + cbase;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
                *reinterpret_cast<f32x4u*>(dst + 4 * j4) = f32x4u{v[4 * j4], v[4 * j4 + 1], v[4 * j4 + 2], v[4 * j4 + 3]};
        } else {
            // ragged channel count (class logits, nc channels): scalar stores
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = cbase + j;
                if (c >= a.Cout) continue;
                float x = v[j];
                if (a.res) x += (float)(reinterpret_cast<const T*>(a.res)[(long)m * a.res_ct + a.res_coff + c]);
                if (a.out_f32) reinterpret_cast<float*>(a.out)[opix * a.out_ct + a.out_coff + c] = x;
                else reinterpret_cast<T*>(a.out)[opix * a.out_ct + a.out_coff + c] = (T)x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ detect-head output 1x1
// The last convolution of each head branch (ultralytics Detect: cv2.x.2 = Conv2d(64, 4 * reg_max, 1), cv3.x.2 = Conv2d(c3, nc, 1);
// no activation, fp32 rows of the prediction buffer) is a GEMM with K = 64..256 and <= 64 output channels: 1-4 K slabs, so the
// generic 128 px x 64 ch tile spends its time in prologue, barriers and epilogue (35 TFLOP/s; the layer is HBM-bound at 0.4-0.5 KB
// per pixel).  Here the WEIGHTS ARE STATIONARY: a workgroup copies the layer's whole packed filter (8-48 KB) to LDS once, in the
// generic kernel's swizzled image, and its four waves then stream 32-pixel groups of the flattened batch (grid-stride over groups):
// a lane loads its MFMA B operand -- 16 bytes of a pixel -- straight from global memory one K chunk ahead of the MFMAs that use
// it, across group boundaries, so the request stream never drains; no barrier after the fill.
// Per output value the MFMA chain is the generic kernel's (same instruction, same operands, same K order: chunks ascending, two
// K steps of 32 each, fp16x3: the virtual chunks [x_lo | x_hi (| x_hi)]) and so is the epilogue: results are bit-identical
// (tests/test_gpu_forward.py::test_head_output_kernel_is_bit_identical).  Rows of 16 packed weight rows that hold no real channel
// (nc = 5: two of four) are neither copied nor multiplied.  nrows = 16 * (row groups kept).
template <bool SPLIT>
__global__ __launch_bounds__(256) void head1x1_kernel(const ConvArgs a, const int nrows) {
    constexpr int MI = 2, PG = 16 * MI, D = 4;              // D: K chunks in flight per wave (register ring)
    constexpr int TP = 68;                                   // pitch (floats) of the store-transpose rows: conflict-free b128 writes and reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int M = a.B * a.Ho * a.Wo, HoWo = a.Ho * a.Wo, cpad = pad128(a.Cout);
    const int pch = a.Cin / 64, vch = SPLIT ? a.split * pch : pch;
    const int niv = nrows >> 4;
    const bool tstore = a.Cout == 64 && a.out_f32;           // whole 64-float slices: stores go through a per-wave LDS transpose
    float* const tbuf = reinterpret_cast<float*>(smem + vch * nrows * 128) + wave * (PG * TP);

    for (int i = tid; i < vch * nrows * 8; i += 256) {
        const int q = i & 7, rr = i >> 3, r = rr % nrows, cc = rr / nrows;
        const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.wgt) + ((size_t)cc * cpad + r) * 128 + q * 16);
        *reinterpret_cast<u32x4*>(smem + (cc * nrows + r) * 128 + ((q ^ (r & 7)) << 4)) = w;
    }
    __syncthreads();

    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const int ngrp = (M + PG - 1) / PG, stride = gridDim.x * 4;
    const int cbase = fq * 16;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = a.bias[cbase + j];
    float sc[SPLIT ? 16 : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[j] = a.oscale[cbase + j];
    }

    u32x4 x[D][MI][2];
    int gi = blockIdx.x * 4 + wave, vi = 0;                 // issue cursor (group, virtual chunk)
    auto issue = [&](u32x4 (&xs)[MI][2]) {
        if (gi >= ngrp) return;
        int lo = 0;
        const int ccp = SPLIT ? x3_chunk(vi, pch, lo) : vi;
        const int lo0 = SPLIT && lo ? a.in0_lo : 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = gi * PG + mi * 16 + fr;
            const unsigned base = (unsigned)(m * a.in0_ct + a.in0_coff + lo0 + ccp * 64 + fq * 8) * 2;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xs[mi][kk] = load_b128(rs0, m < M ? base + kk * 64 : CY_OOB, 0);
        }
        if (++vi == vch) { vi = 0; gi += stride; }
    };

    f32x4 acc[4][MI];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    int g = gi, v = 0;                                       // compute cursor
    auto step = [&](u32x4 (&xc)[MI][2], u32x4 (&xn)[MI][2]) {
        issue(xn);                                           // the slot the previous step consumed
        const char* Wl = smem + v * nrows * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int qf = fq + 4 * kk;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if (ni < niv) {
                    const int r = ni * 16 + fr;
                    const f16x8 wb = *reinterpret_cast<const f16x8*>(Wl + r * 128 + ((qf ^ (r & 7)) << 4));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, __builtin_bit_cast(f16x8, xc[mi][kk]), acc[ni][mi], 0, 0, 0);
                }
            }
        }
        if (++v < vch) return;
        v = 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = g * PG + mi * 16 + fr;
            float o[16];
            if constexpr (SPLIT) {
                scale_bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bv, sc, a.act != 0, o);
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float xo = acc[ni][mi][j] + bv[ni * 4 + j];
                        if (a.act) xo = silu_fast(xo);
                        o[ni * 4 + j] = xo;
                    }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!a.out_f32) {
                // narrow 1x1 layers inside the network (YOLO11 C3k branches, small YOLOv8 scales): fp16 (or high / low halves) into a
                // channel slice, optional residual -- the generic kernel's epilogue, value for value
                if (m >= M) continue;
                f16* dst = reinterpret_cast<f16*>(a.out) + (long)m * a.out_ct + a.out_coff + cbase;
                const f16* rp = a.res ? reinterpret_cast<const f16*>(a.res) + (long)m * a.res_ct + a.res_coff + cbase : nullptr;
                if (cbase + 16 <= a.Cout) {
                    if (rp) {
                        const f16x8 r0v = *reinterpret_cast<const f16x8*>(rp), r1v = *reinterpret_cast<const f16x8*>(rp + 8);
                        if constexpr (SPLIT) {
                            const f16x8 q0v = *reinterpret_cast<const f16x8*>(rp + a.res_lo), q1v = *reinterpret_cast<const f16x8*>(rp + a.res_lo + 8);
#pragma unroll
                            for (int j = 0; j < 8; ++j) { o[j] += (float)r0v[j] + (float)q0v[j]; o[8 + j] += (float)r1v[j] + (float)q1v[j]; }
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) { o[j] += (float)r0v[j]; o[8 + j] += (float)r1v[j]; }
                        }
                    }
                    if constexpr (SPLIT) {
                        store_split16(dst, a.out_lo, o);
                    } else {
                        f16x8 o0, o1;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { o0[j] = (f16)o[j]; o1[j] = (f16)o[8 + j]; }
                        *reinterpret_cast<f16x8*>(dst) = o0;
                        *reinterpret_cast<f16x8*>(dst + 8) = o1;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (cbase + j >= a.Cout) continue;
                        float xo = o[j];
                        if constexpr (SPLIT) {
                            if (rp) xo += (float)rp[j] + (float)rp[a.res_lo + j];
                            store_split1(dst + j, a.out_lo, xo);
                        } else {
                            if (rp) xo += (float)rp[j];
                            dst[j] = (f16)xo;
                        }
                    }
                }
                continue;
            }
            if (tstore) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4)
                    *reinterpret_cast<f32x4*>(tbuf + (mi * 16 + fr) * TP + cbase + 4 * j4) = f32x4{o[4 * j4], o[4 * j4 + 1], o[4 * j4 + 2], o[4 * j4 + 3]};
                continue;
            }
            if (m >= M) continue;
            const int b = m / HoWo, r = m - b * HoWo;
            float* dst = reinterpret_cast<float*>(a.out) + ((long)b * a.out_bs + a.out_ro + r) * a.out_ct + a.out_coff + cbase;
            if (cbase + 16 <= a.Cout) {
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4)
                    *reinterpret_cast<f32x4u*>(dst + 4 * j4) = f32x4u{o[4 * j4], o[4 * j4 + 1], o[4 * j4 + 2], o[4 * j4 + 3]};
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (cbase + j < a.Cout) dst[j] = o[j];
            }
        }
        if (tstore) {
            // wave-private transpose: a store instruction now covers four pixels' 256-byte slices (16 lanes x 16 bytes each) instead
            // of 16-byte pieces of sixteen different rows
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < PG / 4; ++i) {
                const int px = 4 * i + fq, m = g * PG + px;
                const f32x4 t = *reinterpret_cast<const f32x4*>(tbuf + px * TP + fr * 4);
                if (m < M) {
                    const int b = m / HoWo, r = m - b * HoWo;
                    float* dst = reinterpret_cast<float*>(a.out) + ((long)b * a.out_bs + a.out_ro + r) * a.out_ct + a.out_coff + fr * 4;
                    *reinterpret_cast<f32x4u*>(dst) = f32x4u{t[0], t[1], t[2], t[3]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the transpose rows are free again before the next group writes them
        }
        g += stride;
    };

#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(x[d]);
    while (true) {
        if (g >= ngrp) break;
        step(x[0], x[3]);
        if (g >= ngrp) break;
        step(x[1], x[0]);
        if (g >= ngrp) break;
        step(x[2], x[1]);
        if (g >= ngrp) break;
        step(x[3], x[2]);
    }
}

// Both output convolutions of one stride level in ONE launch (default; CY_HEAD_PAIR=0: one launch each): a workgroup (8 waves) holds
// both filters in LDS, a wave streams the K chunks of its 32 pixels from the box branch (64 channels) and then from the class
// branch (256) through the same register ring, keeps two accumulator sets, and writes WHOLE prediction rows: the 64 + nc floats
// of 16 pixels go through a wave-private LDS tile and leave as one contiguous run of dword stores (16 x 276 bytes for nc = 5)
// instead of a 256-byte slice and a 20-byte slice per row from two kernels.  Same MFMA chains, same epilogue arithmetic: bit-identical.
template <bool SPLIT>
__global__ __launch_bounds__(512) void head1x1_pair_kernel(const ConvArgs a, const ConvArgs b, const int nrows_b, const int TP) {
    constexpr int MI = 2, PG = 16 * MI, D = 4, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int M = a.B * a.Ho * a.Wo, HoWo = a.Ho * a.Wo;
    const int pcha = a.Cin / 64, vcha = SPLIT ? a.split * pcha : pcha;
    const int pchb = b.Cin / 64, vchb = SPLIT ? b.split * pchb : pchb;
    const int vch = vcha + vchb, nivb = nrows_b >> 4, rowlen = a.out_ct;
    char* const Wa = smem;
    char* const Wb = smem + vcha * 64 * 128;
    float* const tbuf = reinterpret_cast<float*>(Wb + vchb * nrows_b * 128) + wave * (16 * TP);

    {
        const int cpa = pad128(a.Cout), cpb = pad128(b.Cout);
        for (int i = tid; i < vcha * 64 * 8; i += NW * 64) {
            const int q = i & 7, rr = i >> 3, r = rr & 63, cc = rr >> 6;
            const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.wgt) + ((size_t)cc * cpa + r) * 128 + q * 16);
            *reinterpret_cast<u32x4*>(Wa + (cc * 64 + r) * 128 + ((q ^ (r & 7)) << 4)) = w;
        }
        for (int i = tid; i < vchb * nrows_b * 8; i += NW * 64) {
            const int q = i & 7, rr = i >> 3, r = rr % nrows_b, cc = rr / nrows_b;
            const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(b.wgt) + ((size_t)cc * cpb + r) * 128 + q * 16);
            *reinterpret_cast<u32x4*>(Wb + (cc * nrows_b + r) * 128 + ((q ^ (r & 7)) << 4)) = w;
        }
    }
    __syncthreads();

    const auto rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(b.in0), 0, b.in0_bytes, 0x00020000);
    const int ngrp = (M + PG - 1) / PG, stride = gridDim.x * NW;
    const int cbase = fq * 16;
    float bva[16], bvb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { bva[j] = a.bias[cbase + j]; bvb[j] = b.bias[cbase + j]; }
    float sca[SPLIT ? 16 : 1], scb[SPLIT ? 16 : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { sca[j] = a.oscale[cbase + j]; scb[j] = b.oscale[cbase + j]; }
    }

    u32x4 x[D][MI][2];
    int gi = blockIdx.x * NW + wave, vi = 0;
    auto issue = [&](u32x4 (&xs)[MI][2]) {
        if (gi >= ngrp) return;
        const bool sb = vi >= vcha;                          // wave-uniform: which branch this chunk belongs to
        const int vv = sb ? vi - vcha : vi, pch = sb ? pchb : pcha;
        int lo = 0;
        const int ccp = SPLIT ? x3_chunk(vv, pch, lo) : vv;
        const int ct = sb ? b.in0_ct : a.in0_ct, coff = sb ? b.in0_coff : a.in0_coff;
        const int lo0 = SPLIT && lo ? (sb ? b.in0_lo : a.in0_lo) : 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = gi * PG + mi * 16 + fr;
            const unsigned base = (unsigned)(m * ct + coff + lo0 + ccp * 64 + fq * 8) * 2;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned off = m < M ? base + kk * 64 : CY_OOB;
                xs[mi][kk] = sb ? load_b128(rsb, off, 0) : load_b128(rsa, off, 0);
            }
        }
        if (++vi == vch) { vi = 0; gi += stride; }
    };

    f32x4 acca[4][MI], accb[4][MI];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { acca[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    int g = gi, v = 0;
    auto step = [&](u32x4 (&xc)[MI][2], u32x4 (&xn)[MI][2]) {
        issue(xn);
        if (v < vcha) {
            const char* Wl = Wa + v * 64 * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int qf = fq + 4 * kk;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int r = ni * 16 + fr;
                    const f16x8 wb = *reinterpret_cast<const f16x8*>(Wl + r * 128 + ((qf ^ (r & 7)) << 4));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acca[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, __builtin_bit_cast(f16x8, xc[mi][kk]), acca[ni][mi], 0, 0, 0);
                }
            }
        } else {
            const char* Wl = Wb + (v - vcha) * nrows_b * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int qf = fq + 4 * kk;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    if (ni < nivb) {
                        const int r = ni * 16 + fr;
                        const f16x8 wb = *reinterpret_cast<const f16x8*>(Wl + r * 128 + ((qf ^ (r & 7)) << 4));
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            accb[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, __builtin_bit_cast(f16x8, xc[mi][kk]), accb[ni][mi], 0, 0, 0);
                    }
                }
            }
        }
        if (++v < vch) return;
        v = 0;
        const int mg = g * PG;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float oa[16], ob[16];
            if constexpr (SPLIT) {
                scale_bias_act16(acca[0][mi], acca[1][mi], acca[2][mi], acca[3][mi], bva, sca, false, oa);
                scale_bias_act16(accb[0][mi], accb[1][mi], accb[2][mi], accb[3][mi], bvb, scb, false, ob);
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { oa[ni * 4 + j] = acca[ni][mi][j] + bva[ni * 4 + j]; ob[ni * 4 + j] = accb[ni][mi][j] + bvb[ni * 4 + j]; }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { acca[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            float* trow = tbuf + fr * TP;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
                *reinterpret_cast<f32x4*>(trow + cbase + 4 * j4) = f32x4{oa[4 * j4], oa[4 * j4 + 1], oa[4 * j4 + 2], oa[4 * j4 + 3]};
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (cbase + j < b.Cout) trow[64 + cbase + j] = ob[j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // 16 whole rows: one contiguous run in the prediction buffer unless the run crosses into the next image
            const int m0 = mg + mi * 16;
            const int npx = M - m0 < 16 ? M - m0 : 16;
            if (npx > 0) {
                const int b0 = m0 / HoWo, r0 = m0 - b0 * HoWo;
                float* const obase = reinterpret_cast<float*>(a.out);
                const int nit = (16 * rowlen + 63) >> 6;
                int px = 0, c = lane;
                for (int i = 0; i < nit; ++i) {
                    if (px < npx) {
                        int r = r0 + px, bb = b0;
                        while (r >= HoWo) { r -= HoWo; ++bb; }
                        obase[((long)bb * a.out_bs + a.out_ro + r) * rowlen + c] = tbuf[px * TP + c];
                    }
                    c += 64;
                    if (c >= rowlen) { c -= rowlen; ++px; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        g += stride;
    };

#pragma unroll
    for (int d = 0; d < D - 1; ++d) issue(x[d]);
    while (true) {
        if (g >= ngrp) break;
        step(x[0], x[3]);
        if (g >= ngrp) break;
        step(x[1], x[0]);
        if (g >= ngrp) break;
        step(x[2], x[1]);
        if (g >= ngrp) break;
        step(x[3], x[2]);
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 stride-1, halo reuse
// 90 % of the network's FLOPs are 3x3 stride-1 convolutions.  As a plain implicit GEMM every filter tap re-fetches its
// im2col rows, so a 128x128 tile moves 32 KB L2->LDS per 2.1 MFLOP (64 flop/B) and the kernel is bound by the L2->LDS
// path, not by the matrix cores.  Here a workgroup owns a TH x 16 patch of output pixels of one image: per 64-channel
// slab it stages the (TH+2) x 18 input halo ONCE and the nine taps read it at shifted rows (the MFMA B-operand row of a
// lane is a pixel, so a tap is just a row offset in the LDS image), while only the 128 x 64 weight slab changes per tap.
// L2->LDS traffic per flop drops ~3x (TH=16: 185 KB per 37.7 MFLOP = 204 flop/B).  Zero padding of the halo and of
// ragged image edges comes from the buffer range check, as in the generic kernel.

template <int WM, int RING, int PB = 2>
__global__ __launch_bounds__(WM * 128) void conv3x3_halo_kernel(const ConvArgs a) {
    constexpr int TH = 4 * WM, TW = 16, NT = WM * 128, NW = NT / 64, BN = 128;
    constexpr int PR = (TH + 2) * (TW + 2);                 // halo rows (one row = one pixel, 64 channels = 128 B)
    constexpr int NWI = (PR + 7) / 8;                        // wave-instructions per halo load (8 rows each)
    constexpr int PROUNDS = (NWI + NW - 1) / NW;             // every wave issues exactly PROUNDS pieces (uniform vmcnt)
    constexpr int P_BYTES = PROUNDS * NW * 1024, W_BYTES = BN * 128;
    constexpr int WROUNDS = (BN / 8) / NW;
    constexpr int DIST = RING - 1;                           // weight slabs in flight: tap t+DIST is fetched while tap t computes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Pbuf = smem;
    char* const Wbuf = smem + PB * P_BYTES;              // [halo 0 (| halo 1) | weight ring]
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int H = a.Hi, W = a.Wi;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int cpad = pad128(a.Cout);                      // packed weight rows (zero rows past Cout)
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = id % ntn;
    int rest = id / ntn;
    const int tx = rest % tiles_x; rest /= tiles_x;
    const int ty = rest % tiles_y;
    const int b = rest / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
    const int chunks = a.Cin / 64;

    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt), 0, a.wgt_bytes, 0x00020000);

    // ---- per-thread halo geometry: round j moves halo row (j*NW + wave)*8 + (lane>>3), LDS slot lane&7
    unsigned poff[PROUNDS];
#pragma unroll
    for (int j = 0; j < PROUNDS; ++j) {
        const int wi = j * NW + wave;
        const int r = wi * 8 + (lane >> 3);
        const int ry = r / (TW + 2), rx = r - ry * (TW + 2);
        const int y = y0 + ry - 1, x = x0 + rx - 1;
        const int q = (lane & 7) ^ (r & 7);
        const bool ok = r < PR && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        poff[j] = ok ? (unsigned)(((b * H + y) * W + x) * a.in0_ct + a.in0_coff + q * 8) * 2u : CY_OOB;
    }
    unsigned woff[WROUNDS];
#pragma unroll
    for (int j = 0; j < WROUNDS; ++j) {
        const int row = (j * NW + wave) * 8 + (lane >> 3);
        const int q = (lane & 7) ^ (row & 7);
        woff[j] = (unsigned)((n0 + row) * 128 + q * 16);
    }
    auto dma_patch = [&](int buf, int ch) {
#pragma unroll
        for (int j = 0; j < PROUNDS; ++j) {
            const int wi = j * NW + wave;                   // rows >= PR land in the padded tail of the buffer as zeros
            const unsigned off = poff[j] == CY_OOB ? CY_OOB : poff[j] + (unsigned)(ch * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_void*)(Pbuf + buf * P_BYTES + wi * 1024), 16, off, 0, 0, 0);
        }
    };
    auto dma_w = [&](int buf, int ch, int tap) {
#pragma unroll
        for (int j = 0; j < WROUNDS; ++j) {
            const unsigned off = woff[j] + (unsigned)((ch * 9 + tap) * cpad * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(Wbuf + buf * W_BYTES + (j * NW + wave) * 1024), 16, off, 0, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    auto compute = [&](const char* P, const char* Wb, int kh, int kw) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int qf = fq + 4 * kk;
            f16x8 xa[4], wb[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int r = (wm * 4 + mi + kh) * (TW + 2) + kw + fr;
                xa[mi] = *reinterpret_cast<const f16x8*>(P + r * 128 + ((qf ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int r = wn * 64 + ni * 16 + fr;
                wb[ni] = *reinterpret_cast<const f16x8*>(Wb + r * 128 + ((qf ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
        }
    };

    // ---- pipeline.  Weight slabs live in a 3-slot ring: the slab of tap t+2 is requested while tap t computes, so the
    // wait in front of a barrier only covers a DMA issued a whole tap earlier (counted vmcnt, raw s_barrier: the most
    // recent requests stay in flight across the barrier).  The next 64-channel halo is requested at tap 0 of a slab.
    const int total = chunks * 9;
    auto w_issue = [&](int n) { const int ch = n / 9; dma_w(n % RING, ch, n - ch * 9); };
    dma_patch(0, 0);
    w_issue(0);
    if (DIST > 1 && total > 1) { w_issue(1); CY_WAIT_VM((DIST - 1) * WROUNDS); } else { CY_WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();
    int it = 0;
    const bool stamps = CY_STAMPS_ENABLED && (a.dbg & 64) != 0;
    unsigned long long acc_dma = 0, acc_cmp = 0, acc_wait = 0, acc_bar = 0;
    const unsigned long long t_begin = stamps ? stamp_now() : 0;
    for (int ch = 0; ch < chunks; ++ch) {
        const char* P = Pbuf + (PB == 2 ? (ch & 1) : 0) * P_BYTES;
        if (PB == 1 && ch > 0) {
            // single halo buffer: every wave is past the last tap's barrier, so the buffer is free; fetch the next
            // 64-channel halo now and wait for it (once per nine taps) -- the LDS saved buys a third weight slot
            dma_patch(0, ch);
            CY_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++it) {
            const bool pre_p = PB == 2 && tap == 0 && ch + 1 < chunks, pre_w = it + DIST < total;
            unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if (stamps) { __builtin_amdgcn_sched_barrier(0); t0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            if (pre_p) dma_patch((ch + 1) & 1, ch + 1);
            if (pre_w) w_issue(it + DIST);
            if (stamps) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            const int kh = tap / 3, kw = tap - kh * 3;
            compute(P, Wbuf + (it % RING) * W_BYTES, kh, kw);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this tap's fragment reads are done (WAR on the ring)
            if (stamps) { __builtin_amdgcn_sched_barrier(0); t2 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            if (pre_w && DIST > 1) { CY_WAIT_VM((DIST - 1) * WROUNDS); }   // all but the newest slab request have landed
            else { CY_WAIT_VM(0); }
            if (stamps) { __builtin_amdgcn_sched_barrier(0); t3 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (stamps) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t4 = stamp_now();
                acc_dma += t1 - t0; acc_cmp += t2 - t1; acc_wait += t3 - t2; acc_bar += t4 - t3;
            }
        }
    }

    if (stamps && lane == 0) {
        const unsigned long long t_end = stamp_now();
        atomicAdd(&g_stamps[0], acc_dma); atomicAdd(&g_stamps[1], acc_cmp); atomicAdd(&g_stamps[2], acc_wait);
        atomicAdd(&g_stamps[3], acc_bar); atomicAdd(&g_stamps[4], t_end - t_begin); atomicAdd(&g_stamps[5], (unsigned long long)total);
        atomicAdd(&g_stamps[6], 1ull);
    }
    // ---- epilogue (same per-lane channel layout as the generic kernel)
    const int cbase = n0 + wn * 64 + fq * 16;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = a.bias[cbase + j];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int y = y0 + wm * 4 + mi, x = x0 + fr;
        if (y >= H || x >= W) continue;
        const long pix = ((long)b * H + y) * W + x;
        float v[16];
        bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bv, a.act != 0, v);
        if (cbase + 16 <= a.Cout) {
            f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase;
            if (a.res) {
                const f16* rp = reinterpret_cast<const f16*>(a.res) + pix * a.res_ct + a.res_coff + cbase;
                const f16x8 r0v = *reinterpret_cast<const f16x8*>(rp), r1v = *reinterpret_cast<const f16x8*>(rp + 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] += (float)r0v[j]; v[8 + j] += (float)r1v[j]; }
            }
            f16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
            *reinterpret_cast<f16x8*>(dst) = o0;
            *reinterpret_cast<f16x8*>(dst + 8) = o1;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = cbase + j;
                if (c >= a.Cout) continue;
                float t = v[j];
                if (a.res) t += (float)(reinterpret_cast<const f16*>(a.res)[pix * a.res_ct + a.res_coff + c]);
                reinterpret_cast<f16*>(a.out)[pix * a.out_ct + a.out_coff + c] = (f16)t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 stride-1, ping-pong
// Same data flow as conv3x3_halo_kernel (halo staged once per 64-channel slab, weights per tap), but the eight waves of
// the workgroup run as two groups half an iteration apart (waves 0-3 / 4-7 = the two waves of each SIMD): while one
// group issues its DMA pieces and fragment reads for tap t, its SIMD partner is on the matrix pipe with tap t (or t-1).
// PMC on the lock-step version showed every SIMD's MFMA pipe only ~45 % busy with both waves waiting at the same
// barrier; staggering by wave >= 4 is what MI355X_MICROARCH.md ("Two waves per SIMD", item 9) measures as the fix.
//   group A (waves 0-3): LOAD(t) at half-step 2t,   COMPUTE(t) at 2t+1
//   group B (waves 4-7): LOAD(t) at half-step 2t+1, COMPUTE(t) at 2t+2      (one s_barrier ends every half-step)
// Weight slabs sit in a 4-slot ring and are requested two taps ahead: slot (t+2)%4 last held tap t-2, whose last reader
// (B, half-step 2t-2) is done before A issues at 2t.  The halo of slab ch+1 is requested at tap 1 of slab ch.
// Each wave waits (counted vmcnt) for everything but its newest requests at the end of LOAD; the following barrier
// publishes the landed pieces to all readers.
template <int NI>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(const ConvArgs a) {
    constexpr int TH = 16, TW = 16, NW = 8, WN = 2, BN = WN * NI * 16, RING = 4;
    constexpr int PR = (TH + 2) * (TW + 2), NWI = (PR + 7) / 8, PROUNDS = (NWI + NW - 1) / NW;
    constexpr int P_BYTES = (NWI + 1) * 1024, W_BYTES = BN * 128;       // +1 KB dummy row block for the padded rounds
    constexpr int WPIECES = BN / 8, WROUNDS = (WPIECES + NW - 1) / NW;  // BN=128: 2 per wave; BN=64: 1 per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Pbuf = smem;
    char* const Wbuf = smem + 2 * P_BYTES;
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 1, wn = wave & 1;
    const int H = a.Hi, W = a.Wi;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int cpad = pad128(a.Cout);                      // packed weight rows (zero rows past Cout)
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = id % ntn;
    int rest = id / ntn;
    const int tx = rest % tiles_x; rest /= tiles_x;
    const int ty = rest % tiles_y;
    const int b = rest / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
    const int chunks = a.Cin / 64, total = chunks * 9;

    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt), 0, a.wgt_bytes, 0x00020000);

    unsigned poff[PROUNDS];
#pragma unroll
    for (int j = 0; j < PROUNDS; ++j) {
        const int r = (j * NW + wave) * 8 + (lane >> 3);
        const int ry = r / (TW + 2), rx = r - ry * (TW + 2);
        const int y = y0 + ry - 1, x = x0 + rx - 1;
        const int q = (lane & 7) ^ (r & 7);
        const bool ok = r < PR && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        poff[j] = ok ? (unsigned)(((b * H + y) * W + x) * a.in0_ct + a.in0_coff + q * 8) * 2u : CY_OOB;
    }
    unsigned woff[WROUNDS];
#pragma unroll
    for (int j = 0; j < WROUNDS; ++j) {
        const int row = ((j * NW + wave) % WPIECES) * 8 + (lane >> 3);
        const int q = (lane & 7) ^ (row & 7);
        woff[j] = (unsigned)((n0 + row) * 128 + q * 16);
    }
    auto dma_patch = [&](int buf, int ch) {
#pragma unroll
        for (int j = 0; j < PROUNDS; ++j) {
            const int wi = j * NW + wave;
            const unsigned off = poff[j] == CY_OOB ? CY_OOB : poff[j] + (unsigned)(ch * 128);
            char* dst = Pbuf + buf * P_BYTES + (wi < NWI ? wi : NWI) * 1024;      // rounds past the halo hit the dummy block
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_void*)dst, 16, off, 0, 0, 0);
        }
    };
    auto dma_w = [&](int n) {                               // weight slab of global tap index n -> ring slot n % RING
        const int ch = n / 9, tap = n - ch * 9;
#pragma unroll
        for (int j = 0; j < WROUNDS; ++j) {
            const unsigned off = woff[j] + (unsigned)((ch * 9 + tap) * cpad * 128);
            char* dst = Wbuf + (n % RING) * W_BYTES + ((j * NW + wave) % WPIECES) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)dst, 16, off, 0, 0, 0);
        }
    };

    f32x4 acc[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    f16x8 xa0[4], wb0[NI], xa1[4], wb1[NI];

    auto read_frags = [&](const char* P, const char* Wb, int kh, int kw, int kk, f16x8* xa, f16x8* wb) {
        const int qf = fq + 4 * kk;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int r = (wm * 4 + mi + kh) * (TW + 2) + kw + fr;
            xa[mi] = *reinterpret_cast<const f16x8*>(P + r * 128 + ((qf ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int r = wn * (NI * 16) + ni * 16 + fr;
            wb[ni] = *reinterpret_cast<const f16x8*>(Wb + r * 128 + ((qf ^ (r & 7)) << 4));
        }
    };
    auto mfma_all = [&](const f16x8* xa, const f16x8* wb) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
    };

    // ---- prologue: halo 0 and the first two weight slabs, fully landed before anyone reads
    dma_patch(0, 0);
    dma_w(0);
    if (total > 1) dma_w(1);
    CY_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();             // group B runs half an iteration behind

    int it = 0;
#pragma unroll 1
    for (int ch = 0; ch < chunks; ++ch) {
        const char* P = Pbuf + (ch & 1) * P_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++it) {
            const char* Wb = Wbuf + (it % RING) * W_BYTES;
            const int kh = tap / 3, kw = tap - kh * 3;
            // -------- LOAD(it): requests two taps ahead, fragment reads of the first 32 channels
            const bool pre_p = tap == 1 && ch + 1 < chunks, pre_w = it + 2 < total;
            if (pre_w) dma_w(it + 2);
            if (pre_p) dma_patch((ch + 1) & 1, ch + 1);      // issued after the slab so the halo may stay in flight
            read_frags(P, Wb, kh, kw, 0, xa0, wb0);
            if (pre_p) { CY_WAIT_VM(WROUNDS + PROUNDS); }    // the slab of tap it+1 (requested last iteration) has landed
            else if (pre_w) { CY_WAIT_VM(WROUNDS); }
            else { CY_WAIT_VM(0); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // -------- COMPUTE(it)
            read_frags(P, Wb, kh, kw, 1, xa1, wb1);
            __builtin_amdgcn_sched_barrier(0);
            if (a.dbg & 32) __builtin_amdgcn_s_setprio(1);
            mfma_all(xa0, wb0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_all(xa1, wb1);
            if (a.dbg & 32) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();             // pairs with group B's last barrier

    // ---- epilogue: lane owns NI*4 contiguous channels of one pixel
    // packed weight row wn*NI*16 + ni*16 + rr holds channel 64*blk + 16*(rr>>2) + 4*ni_g + (rr&3), ni_g = (wn*NI+ni)&3
    const int cbase = n0 + ((wn * NI) >> 2) * 64 + fq * 16 + ((wn * NI) & 3) * 4;
    float bv[NI * 4];
#pragma unroll
    for (int j = 0; j < NI * 4; ++j) bv[j] = a.bias[cbase + j];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int y = y0 + wm * 4 + mi, x = x0 + fr;
        if (y >= H || x >= W) continue;
        const long pix = ((long)b * H + y) * W + x;
        float v[NI * 4];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[ni][mi][j] + bv[ni * 4 + j];
                if (a.act) t = silu_fast(t);
                v[ni * 4 + j] = t;
            }
        if (cbase + NI * 4 <= a.Cout) {
            f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase;
            if (a.res) {
                const f16* rp = reinterpret_cast<const f16*>(a.res) + pix * a.res_ct + a.res_coff + cbase;
#pragma unroll
                for (int h8 = 0; h8 < NI / 2; ++h8) {
                    const f16x8 rv = *reinterpret_cast<const f16x8*>(rp + 8 * h8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[8 * h8 + j] += (float)rv[j];
                }
            }
#pragma unroll
            for (int h8 = 0; h8 < NI / 2; ++h8) {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)v[8 * h8 + j];
                *reinterpret_cast<f16x8*>(dst + 8 * h8) = o;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI * 4; ++j) {
                const int c = cbase + j;
                if (c >= a.Cout) continue;
                float t = v[j];
                if (a.res) t += (float)(reinterpret_cast<const f16*>(a.res)[pix * a.res_ct + a.res_coff + c]);
                reinterpret_cast<f16*>(a.out)[pix * a.out_ct + a.out_coff + c] = (f16)t;
            }
        }
    }
}

template <int NI>
static hipError_t launch_pp(const ConvArgs& a, hipStream_t s) {
    constexpr int BN = 2 * NI * 16, PR = 18 * 18, NWI = (PR + 7) / 8;
    const size_t lds = 2 * (NWI + 1) * 1024 + 4 * BN * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_pp_kernel<NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int blocks = a.B * ((a.Hi + 15) / 16) * ((a.Wi + 15) / 16) * ((pad64(a.Cout) + BN - 1) / BN);
    hipLaunchKernelGGL((conv3x3_pp_kernel<NI>), dim3(blocks), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 3x3 s1, Cin = 64: persistent
// The six bottleneck convolutions of the full-resolution C2f (64 -> 64 channels over 1M pixels per 64-tile batch) have
// only nine K-slabs per output patch, so a one-patch-per-workgroup kernel spends most of its time in prologue/epilogue
// (measured: 16 rounds x ~10 us).  Here a workgroup is persistent (one per CU): the whole 64 x (9 x 64) weight panel is
// staged in LDS ONCE (72 KB) next to two input-halo buffers (2 x 42 KB), and the nine taps of a patch run without a barrier.
// KC = input channels (64, or 32 with 64-byte LDS rows and the 64-byte-chunk weight copy: the 32-channel bottlenecks of the
// n/s scales and of YOLO11's C3k blocks; the output side still walks all 64 packed rows, half of them zero for Cout = 32).
//
// Two wave groups.  Round 1's form had four waves (one per SIMD) and a double-buffered halo; its in-kernel stamps showed, per
// patch and wave, 6.2k cycles of fragment reads + MFMAs, 5.2k of epilogue (bias, SiLU, residual, stores), 1.9k of LDS-DMA issue
// and next to no waiting -- with ONE wave per SIMD all of it is serial.  Now a workgroup has 8 waves = two groups of four
// (waves w and w + 4 share a SIMD); its patches alternate between the groups, and in every phase one group runs the MFMAs of
// its patch while the other runs the "back" of its previous one: request the halo of its next patch into its own (single)
// halo buffer, epilogue, wait.  One workgroup barrier per phase orders both hand-overs (halo landed -> readable; halo read ->
// writable).  256 tiles of 128x128x64: 0.43-0.48 -> 0.35-0.39 ms per conv (3.5 TB/s of HBM traffic without, 4.5 TB/s with a
// residual input: the kernel is now bound by memory, not by issue; making the halo requests and the epilogue cheaper -- uniform
// address parts in soffset, an interior-patch path without border tests, packed fp32 math -- no longer moved it).
template <int KC, bool ACT, bool RES>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const ConvArgs a) {
    constexpr int TH = 16, TW = 16, NG = 4, NI = 4;      // NG waves per group, each 64 px x 64 ch
    constexpr int RB = KC * 2, RPP = 1024 / RB, NCH = KC / 8, KK = KC / 32;
    constexpr int PR = (TH + 2) * (TW + 2), NWI = (PR + RPP - 1) / RPP, PROUNDS = (NWI + NG - 1) / NG;
    auto swz = [](int chunk, int row) { return KC == 64 ? (chunk ^ (row & 7)) : (chunk ^ (((row >> 2) & 1) << 1)); };
    constexpr int P_BYTES = (NWI + 1) * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int H = a.Hi, W = a.Wi;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int npatch = a.B * tiles_y * tiles_x;
    // halo requests: offset = (per-lane part, fixed for the whole launch, in voffset) + (per-patch part, uniform, in soffset).
    // soffset is unsigned and the per-patch part starts one row + one pixel before the patch, so the resource base is moved
    // back by that much (num_records widened to match); only voffset is range-checked: lanes outside the image get CY_OOB.
    const unsigned bias_bytes = (unsigned)((W + 1) * a.in0_ct) * 2u;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.in0)) - bias_bytes, 0,
                                                       a.in0_bytes + 2 * bias_bytes, 0x00020000);

    constexpr int W_BYTES = 9 * 64 * RB, WPIECES = W_BYTES / 1024, PPT = 64 / RPP;
    char* const Wl = smem;
    char* const P = smem + W_BYTES + grp * P_BYTES;           // this group's halo buffer
    {
        const auto rsw = KC == 64 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt), 0, a.wgt_bytes, 0x00020000)
                                  : __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt32), 0, a.wgt32_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < (WPIECES + 7) / 8; ++j) {
            const int pc = j * 8 + wave;
            if (pc >= WPIECES) break;                       // (36 pieces for KC = 32)
            const int tap = pc / PPT, row = (pc % PPT) * RPP + lane / NCH;
            const int q = swz(lane % NCH, row);
            const unsigned off = (unsigned)((tap * 128 + row) * RB + q * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(Wl + pc * 1024), 16, off, 0, 0, 0);
        }
    }
    const int cbase = fq * 16;
    f32x2 bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = f32x2{a.bias[cbase + 2 * j], a.bias[cbase + 2 * j + 1]};

    // per-lane parts of the PROUNDS halo pieces this wave requests: byte offset relative to the halo's first pixel, and the
    // halo row / column (packed) for the border test
    unsigned rel[PROUNDS];
#pragma unroll
    for (int j = 0; j < PROUNDS; ++j) {
        const int wi = j * NG + wm;
        const int r = wi * RPP + lane / NCH;
        const int ry = r / (TW + 2), rx = r - ry * (TW + 2);
        const int q = swz(lane % NCH, r);
        rel[j] = r < PR ? (unsigned)(((ry * W + rx) * a.in0_ct + q * 8) * 2) : CY_OOB;
    }
    const int rlane = lane / NCH;
    auto dma_patch = [&](int pidx) {                       // by the four waves of the group that owns the patch
        const int tx = pidx % tiles_x, ty = (pidx / tiles_x) % tiles_y, b = pidx / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const unsigned so = bias_bytes + (unsigned)((((b * H + y0 - 1) * W + x0 - 1) * a.in0_ct + a.in0_coff) * 2);
        const bool interior = y0 >= 1 && x0 >= 1 && y0 + TH + 1 <= H && x0 + TW + 1 <= W;
        if (interior) {
#pragma unroll
            for (int j = 0; j < PROUNDS; ++j) {
                const int wi = j * NG + wm;
                dma_piece(rs0, (lds_ptr_t*)(P + (wi < NWI ? wi : NWI) * 1024), rel[j], so);
            }
        } else {
            // halo rows [ylo, yhi) and columns [xlo, xhi) are inside the image
            const int ylo = y0 >= 1 ? 0 : 1, yhi = H - y0 + 1, xlo = x0 >= 1 ? 0 : 1, xhi = W - x0 + 1;
#pragma unroll
            for (int j = 0; j < PROUNDS; ++j) {
                const int wi = j * NG + wm;
                const int r = wi * RPP + rlane, ry = (r * 3641) >> 16, rx = r - ry * (TW + 2);      // r / 18 for r < 400
                const bool ok = ry >= ylo && ry < yhi && rx >= xlo && rx < xhi;
                dma_piece(rs0, (lds_ptr_t*)(P + (wi < NWI ? wi : NWI) * 1024), ok ? rel[j] : CY_OOB, so);
            }
        }
    };

    // patch n of this workgroup = blockIdx.x + n * gridDim.x, owned by group n & 1; NP of them
    const int NP = (int)blockIdx.x < npatch ? (npatch - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (grp < NP) dma_patch(blockIdx.x + grp * gridDim.x);
    // (the builtin, not inline asm: the compiler then knows that the bias loads above have landed; with the asm form it
    // put an `s_waitcnt vmcnt(0)` in front of the first use of bv[] INSIDE the loop, i.e. between the halo requests of a
    // back phase and its epilogue, which exposed the whole DMA latency there)
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    __builtin_amdgcn_s_barrier();

    // fragment addresses as (one of a few lane-dependent bases) + (a compile-time offset): the swizzle of halo row R0 + c depends
    // on (R0 + c) & 7 only, and the second K half (chunk + 4) is the swizzle of row + 4, so eight bases cover all 72 fragments
    // of a patch (computed one by one they are 36 live registers, which at two waves per SIMD spill)
    const int R0 = wm * 4 * (TW + 2) + fr;
    unsigned xb[8], wb0[KK];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) xb[k8] = (unsigned)(R0 * RB + (swz(fq, (R0 + k8) & 7) << 4));
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) wb0[kk] = (unsigned)(fr * RB + (swz(fq + 4 * kk, fr) << 4));
    f32x4 acc[NI][4];
    const bool vec_out = cbase + 16 <= a.Cout;
    const int out_lane = fr * a.out_ct + cbase, res_lane = fr * a.res_ct + cbase;       // element offsets of the lane inside a patch row
    // phase ph: group ph & 1 runs the MFMAs of patch ph; the other group the back of patch ph - 1 (+ the halo of patch ph + 1)
    for (int ph = 0; ph <= NP; ++ph) {
        if ((ph & 1) == grp) {
            if (ph < NP) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                f16x8 xa[2][4], wb[2][NI];
                auto load_group = [&](int g, f16x8* x, f16x8* w) {
                    const int tap = g / KK, kk = g % KK, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int c = (mi + kh) * (TW + 2) + kw;                     // halo row = R0 + c
                        x[mi] = *reinterpret_cast<const f16x8*>(P + xb[(c + 4 * kk) & 7] + c * RB);
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        w[ni] = *reinterpret_cast<const f16x8*>(Wl + wb0[kk] + (tap * 64 + ni * 16) * RB);
                };
                load_group(0, xa[0], wb[0]);
#pragma unroll
                for (int g = 0; g < 9 * KK; ++g) {
                    if (g + 1 < 9 * KK) load_group(g + 1, xa[(g + 1) & 1], wb[(g + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[g & 1][ni], xa[g & 1][mi], acc[ni][mi], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this group is done reading its halo
            }
        } else if (ph >= 1) {
            f16x8 rres[4][2];
            const int pidx = blockIdx.x + (ph - 1) * gridDim.x;
            const int tx = pidx % tiles_x, ty = (pidx / tiles_x) % tiles_y, b = pidx / (tiles_x * tiles_y);
            const int yw = ty * TH + wm * 4, x = tx * TW + fr;                 // first of the wave's four rows; the lane's column
            const long rowpix = ((long)b * H + yw) * W + tx * TW;              // uniform: first pixel of the wave's first row
            const bool xok = x < W;
            if (RES) {                                     // residual of the patch: its latency hides under the DMA issue and the SiLUs below
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const bool ok = yw + mi < H && xok && vec_out;
                    // (masked lanes read the start of the patch's first pixel: valid memory, never used)
                    const f16* rp = reinterpret_cast<const f16*>(a.res) + ((rowpix + (yw + mi < H ? (long)mi * W : 0)) * a.res_ct + a.res_coff) + (ok ? res_lane : 0);
                    rres[mi][0] = *reinterpret_cast<const f16x8*>(rp);
                    rres[mi][1] = *reinterpret_cast<const f16x8*>(rp + 8);
                }
            }
            if (ph + 1 < NP) dma_patch(blockIdx.x + (ph + 1) * gridDim.x);
            // bias + SiLU in place, two values per instruction where the ISA has a packed form
            if (ACT) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x2 t = f32x2{acc[ni][mi][2 * h], acc[ni][mi][2 * h + 1]} + bv[ni * 2 + h];
                            f32x2 e = t * f32x2{-1.44269504088896341f, -1.44269504088896341f};
                            e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
                            t = t * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                            acc[ni][mi][2 * h] = t[0]; acc[ni][mi][2 * h + 1] = t[1];
                        }
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ni][mi][j] += bv[ni * 2 + (j >> 1)][j & 1];
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                if (yw + mi >= H) break;                   // (uniform)
                if (!xok) continue;
                if (vec_out) {
                    f16* dst = reinterpret_cast<f16*>(a.out) + ((rowpix + (long)mi * W) * a.out_ct + a.out_coff) + out_lane;
                    float v[16];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[ni * 4 + j] = acc[ni][mi][j];
                    if (RES) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { v[j] += (float)rres[mi][0][j]; v[8 + j] += (float)rres[mi][1][j]; }
                    }
                    f16x8 o0, o1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
                    *reinterpret_cast<f16x8*>(dst) = o0;
                    *reinterpret_cast<f16x8*>(dst + 8) = o1;
                } else {
                    const long pix = rowpix + (long)mi * W + fr;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int c = cbase + j;
                        if (c >= a.Cout) continue;
                        float t = acc[j >> 2][mi][j & 3];
                        if (RES) t += (float)(reinterpret_cast<const f16*>(a.res)[pix * a.res_ct + a.res_coff + c]);
                        reinterpret_cast<f16*>(a.out)[pix * a.out_ct + a.out_coff + c] = (f16)t;
                    }
                }
            }
            CY_WAIT_VM(0);                                 // the halo requested above has landed (and the stores are out)
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int KC>
static hipError_t launch_c64(const ConvArgs& a, hipStream_t s) {
    constexpr int RPP = 1024 / (KC * 2), NWI = (18 * 18 + RPP - 1) / RPP;
    const size_t lds = 9 * 64 * KC * 2 + 2 * (NWI + 1) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<KC, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<KC, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<KC, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<KC, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int npatch = a.B * ((a.Hi + 15) / 16) * ((a.Wi + 15) / 16);
    const int grid = npatch < 256 ? npatch : 256;                                  // one persistent workgroup per CU
    if (a.act && a.res) hipLaunchKernelGGL((conv3x3_c64_kernel<KC, true, true>), dim3(grid), dim3(512), lds, s, a);
    else if (a.act) hipLaunchKernelGGL((conv3x3_c64_kernel<KC, true, false>), dim3(grid), dim3(512), lds, s, a);
    else if (a.res) hipLaunchKernelGGL((conv3x3_c64_kernel<KC, false, true>), dim3(grid), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_c64_kernel<KC, false, false>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 3x3 s1, two taps per barrier
// Variant of conv3x3_halo_kernel for 16x16-pixel patches (8 waves, one workgroup per CU, all 160 KiB of LDS): a pipeline
// stage holds the weight slabs of TWO taps, so a wave issues 64 MFMAs between barriers instead of 32 and the per-stage
// cost (DMA issue, vmcnt drain, s_barrier: ~600 cycles in the stamped build) is paid half as often.  Input channels are
// walked in pairs of 64-channel slabs (18 taps = 9 stages, fully unrolled, so every fragment address is a constant
// offset); the two halo buffers hold the even / odd slab of the pair and are refilled as soon as their last tap is done.
__global__ __launch_bounds__(512) void conv3x3_halo2_kernel(const ConvArgs a) {
    constexpr int TH = 16, TW = 16, NW = 8, BN = 128;
    constexpr int PR = (TH + 2) * (TW + 2), NWI = (PR + 7) / 8, PROUNDS = (NWI + NW - 1) / NW;
    constexpr int P_BYTES = PROUNDS * NW * 1024, SLAB = BN * 128, W_BYTES = 2 * SLAB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Pbuf = smem;
    char* const Wbuf = smem + 2 * P_BYTES;
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int H = a.Hi, W = a.Wi;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int cpad = pad128(a.Cout);                      // packed weight rows (zero rows past Cout)
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = id % ntn;
    int rest = id / ntn;
    const int tx = rest % tiles_x; rest /= tiles_x;
    const int ty = rest % tiles_y;
    const int b = rest / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
    const int chunks = a.Cin / 64, pairs = chunks / 2;      // launch_conv guarantees an even number of slabs

    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt), 0, a.wgt_bytes, 0x00020000);
    unsigned poff[PROUNDS];
#pragma unroll
    for (int j = 0; j < PROUNDS; ++j) {
        const int r = (j * NW + wave) * 8 + (lane >> 3);
        const int ry = r / (TW + 2), rx = r - ry * (TW + 2);
        const int y = y0 + ry - 1, x = x0 + rx - 1;
        const int q = (lane & 7) ^ (r & 7);
        const bool ok = r < PR && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        poff[j] = ok ? (unsigned)(((b * H + y) * W + x) * a.in0_ct + a.in0_coff + q * 8) * 2u : CY_OOB;
    }
    unsigned woff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * NW + wave) * 8 + (lane >> 3);
        woff[j] = (unsigned)((n0 + row) * 128 + ((lane & 7) ^ (row & 7)) * 16);
    }
    auto dma_patch = [&](int buf, int ch) {
#pragma unroll
        for (int j = 0; j < PROUNDS; ++j) {
            // the uniform part of the offset goes in soffset, which is not range-checked: an OOB sentinel stays OOB
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_void*)(Pbuf + buf * P_BYTES + (j * NW + wave) * 1024), 16, poff[j], ch * 128, 0, 0);
        }
    };
    auto dma_slab = [&](int buf, int t, int ch, int tap) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(Wbuf + buf * W_BYTES + t * SLAB + (j * NW + wave) * 1024), 16, woff[j], (ch * 9 + tap) * cpad * 128, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](const char* P, const char* Wb, int kh, int kw) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int qf = fq + 4 * kk;
            f16x8 xa[4], wb[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int r = (wm * 4 + mi + kh) * (TW + 2) + kw + fr;
                xa[mi] = *reinterpret_cast<const f16x8*>(P + r * 128 + ((qf ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int r = wn * 64 + ni * 16 + fr;
                wb[ni] = *reinterpret_cast<const f16x8*>(Wb + r * 128 + ((qf ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
        }
    };

    // prologue: halo of slab 0, weight stage 0 (taps 0,1 of slab 0)
    dma_patch(0, 0);
    dma_slab(0, 0, 0, 0);
    dma_slab(0, 1, 0, 1);
    __syncthreads();
    int g = 0;                                               // global stage counter: weight stage buffer = g & 1
    const bool stamps = CY_STAMPS_ENABLED && (a.dbg & 64) != 0;
    unsigned long long acc_dma = 0, acc_cmp = 0, acc_wait = 0, acc_bar = 0;
    const unsigned long long t_begin = stamps ? stamp_now() : 0;
    unsigned long long t0s = t_begin;
#pragma unroll 1
    for (int cp = 0; cp < pairs; ++cp) {
#pragma unroll
        for (int st = 0; st < 9; ++st, ++g) {
            // requests for the next stage / the halo buffers that have just become free
            if (st < 8) {
                const int u0 = 2 * (st + 1), u1 = u0 + 1;
                dma_slab((g + 1) & 1, 0, 2 * cp + u0 / 9, u0 % 9);
                dma_slab((g + 1) & 1, 1, 2 * cp + u1 / 9, u1 % 9);
            } else if (cp + 1 < pairs) {
                dma_slab((g + 1) & 1, 0, 2 * cp + 2, 0);
                dma_slab((g + 1) & 1, 1, 2 * cp + 2, 1);
            }
            if (st == 0) dma_patch(1, 2 * cp + 1);                                  // odd slab of this pair (needed from stage 4)
            if (st == 5 && cp + 1 < pairs) dma_patch(0, 2 * cp + 2);                // even slab of the next pair
            unsigned long long t1 = 0, t2 = 0, t3 = 0;
            if (stamps) { __builtin_amdgcn_sched_barrier(0); t1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int u = 2 * st + t, tap = u % 9;
                compute(Pbuf + (u / 9) * P_BYTES, Wbuf + (g & 1) * W_BYTES + t * SLAB, tap / 3, tap % 3);
            }
            if (stamps) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0); t2 = stamp_now(); __builtin_amdgcn_sched_barrier(0);
                CY_WAIT_VM(0);
                __builtin_amdgcn_sched_barrier(0); t3 = stamp_now(); __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (stamps) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t4 = stamp_now();
                acc_dma += t1 - t0s; acc_cmp += t2 - t1; acc_wait += t3 - t2; acc_bar += t4 - t3;
                t0s = t4;
            }
        }
    }

    if (stamps && lane == 0) {
        const unsigned long long t_end = stamp_now();
        atomicAdd(&g_stamps[0], acc_dma); atomicAdd(&g_stamps[1], acc_cmp); atomicAdd(&g_stamps[2], acc_wait);
        atomicAdd(&g_stamps[3], acc_bar); atomicAdd(&g_stamps[4], t_end - t_begin); atomicAdd(&g_stamps[5], (unsigned long long)(pairs * 9));
        atomicAdd(&g_stamps[6], 1ull);
    }
    const int cbase = n0 + wn * 64 + fq * 16;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = a.bias[cbase + j];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int y = y0 + wm * 4 + mi, x = x0 + fr;
        if (y >= H || x >= W) continue;
        const long pix = ((long)b * H + y) * W + x;
        float v[16];
        bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bv, a.act != 0, v);
        if (cbase + 16 <= a.Cout) {
            f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase;
            if (a.res) {
                const f16* rp = reinterpret_cast<const f16*>(a.res) + pix * a.res_ct + a.res_coff + cbase;
                const f16x8 r0v = *reinterpret_cast<const f16x8*>(rp), r1v = *reinterpret_cast<const f16x8*>(rp + 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] += (float)r0v[j]; v[8 + j] += (float)r1v[j]; }
            }
            f16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
            *reinterpret_cast<f16x8*>(dst) = o0;
            *reinterpret_cast<f16x8*>(dst + 8) = o1;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = cbase + j;
                if (c >= a.Cout) continue;
                float t = v[j];
                if (a.res) t += (float)(reinterpret_cast<const f16*>(a.res)[pix * a.res_ct + a.res_coff + c]);
                reinterpret_cast<f16*>(a.out)[pix * a.out_ct + a.out_coff + c] = (f16)t;
            }
        }
    }
}

static hipError_t launch_halo2(const ConvArgs& a, hipStream_t s) {
    constexpr int NWI = (18 * 18 + 7) / 8, PROUNDS = (NWI + 7) / 8;
    const size_t lds = 2 * PROUNDS * 8 * 1024 + 2 * 2 * 128 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int blocks = a.B * ((a.Hi + 15) / 16) * ((a.Wi + 15) / 16) * ((pad64(a.Cout) + 127) / 128);
    hipLaunchKernelGGL(conv3x3_halo2_kernel, dim3(blocks), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 3x3 s1, 512 px x 128 ch
// The halo kernels above are not limited by matrix-pipe or LDS bandwidth but by the cost of getting bytes into LDS: every
// 1 KiB LDS-DMA piece holds its issuing wave for 60-180 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), and
// conv3x3_halo2_kernel needs 41 pieces per 64 MFMAs of every wave.  Weight bytes per flop only depend on the number of
// PIXELS a workgroup owns, so this variant doubles them: a 16 x 32-pixel patch (512 px) x 128 channels, each wave a
// 128 px x 64 ch tile (acc = 128 VGPRs).  To make that fit in LDS the K slab is 32 channels (LDS rows of 64 B):
//   halo 18 x 34 px x 64 B = 38.3 KiB (x2 buffers), weight stage = two taps x 128 rows x 64 B = 16 KiB (x3 ring)  -> 128 KiB
// One stage is again 64 MFMAs per wave but only ~25 DMA pieces and 24 instead of 32 fragment reads, and the ring is
// deep enough to request weights two stages ahead (counted vmcnt; halo pieces stay in flight for 2-3 stages).
// 64-byte rows: chunk' = chunk ^ 2*((row>>2)&1) is conflict-free for the ds_read_b128 lane groups at any row offset.
// Weights come from the second packed copy with 64-byte K chunks (ConvArgs::wgt32).
// WN = 1: the 64-channel variant (BN = 64): eight waves of 64 px x 64 ch (two image rows each), one DMA piece of weights
// per wave and stage, 32 MFMAs per wave between barriers.
// DUAL = true: maps at most 16 pixels wide (the stride-32 level of a 512-px tile): the 32 patch columns are the 16 columns of
// TWO consecutive images, each with its own left/right halo column (patch rows of 36 instead of 34 pixels).
template <bool TAIL, int WN, bool DUAL = false, int TPS = 2, bool SPLIT = false, int SCHED = 0>
__global__ __launch_bounds__(512) void conv3x3_wide_kernel(const ConvArgs a) {
    // SCHED = 1 (round 4 experiment, CY_WIDE_SCHED=1): the eight fragment reads of a half-step are spread between the 16 MFMAs of the
    // previous one (sched_group_barrier: one ds_read_b128 per two MFMAs) instead of where the compiler sinks them
    // SPLIT (fp16x3 context): three passes over the input channels -- halo slabs of [x_lo | x_hi | x_hi] against the weight
    // slabs [w_hi | w_lo | w_hi] of the packed copy, i.e. the same stage loop over 3x the slab pairs; scaled / split epilogue
    // TPS = taps per stage (between two barriers): 2, or 3 (WN = 2 only: six stages of 96 MFMAs per slab pair, 152 KiB of LDS)
    static_assert(TPS == 2 || (TPS == 3 && WN == 2), "three taps per stage: 128-channel variant only");
    constexpr int NST = 18 / TPS;
    constexpr int TH = 16, TW = 32, NW = 8, BN = 64 * WN, PWID = DUAL ? 36 : TW + 2, HALF = DUAL ? 18 : 16;
    constexpr int RPW = TH / (NW / WN), MIW = 2 * RPW, WPS = WN == 2 ? TPS : 1;   // image rows / pixel fragments per wave; weight pieces per wave and stage
    constexpr int PR = (TH + 2) * PWID, NPC = (PR + 15) / 16, PROUNDS = (NPC + NW - 1) / NW;
    constexpr int P_BYTES = PROUNDS * NW * 1024, SLAB = BN * 64, W_BYTES = TPS * SLAB, RING = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Pbuf = smem;
    char* const Wbuf = smem + 2 * P_BYTES;
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int H = a.Hi, W = a.Wi;
    const int tiles_x = DUAL ? 1 : (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int cpad = pad128(a.Cout);                      // packed weight rows (zero rows past Cout)
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = id % ntn;
    int rest = id / ntn;
    const int tx = rest % tiles_x; rest /= tiles_x;
    const int ty = rest % tiles_y;
    const int b = rest / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
    const int ppairs = a.Cin / 64;                          // pairs of 32-channel slabs in one pass over the input channels
    const int pairs = SPLIT ? a.split * ppairs : ppairs;     // a.split = passes of the fp16x3 context (3, or 2 for fp16-exact weights)
    const bool stamps = CY_STAMPS_ENABLED && (a.dbg & 64) != 0;      // diagnostic builds: phase stamps of the workgroup
    const unsigned long long t_entry = stamps ? stamp_real() : 0;

    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt32), 0, a.wgt32_bytes, 0x00020000);
    unsigned poff[PROUNDS];
#pragma unroll
    for (int j = 0; j < PROUNDS; ++j) {
        const int r = (j * NW + wave) * 16 + (lane >> 2);
        const int ry = r / PWID, rx = r - ry * PWID;
        const int half = DUAL ? rx / 18 : 0;                 // DUAL: which of the block's two images this patch column belongs to
        const int bb = DUAL ? 2 * b + half : b;
        const int y = y0 + ry - 1, x = DUAL ? rx - half * 18 - 1 : x0 + rx - 1;
        const int q = (lane & 3) ^ (((r >> 2) & 1) << 1);
        const bool ok = r < PR && bb < a.B && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        poff[j] = ok ? (unsigned)(((bb * H + y) * W + x) * a.in0_ct + a.in0_coff + q * 8) * 2u : CY_OOB;
    }
    unsigned woff;
    {   // WN = 2: every wave moves rows wave*16.. of BOTH taps of a stage; WN = 1: wave -> (tap wave>>2, rows (wave&3)*16..)
        const int row = (WN == 2 ? wave : (wave & 3)) * 16 + (lane >> 2);
        woff = (unsigned)((n0 + row) * 64 + ((lane & 3) ^ (((row >> 2) & 1) << 1)) * 16);
    }
    auto dma_patch = [&](int buf, int slab) {               // slab: 32-channel slab index
        unsigned so = (unsigned)slab * 64u;
        if constexpr (SPLIT) {                               // virtual slab -> physical slab (+ the offset of the low halves in the first pass)
            int lo;
            so = (unsigned)x3_chunk(slab, 2 * ppairs, lo) * 64u;
            if (lo) so += (unsigned)a.in0_lo * 2u;
        }
#pragma unroll
        for (int j = 0; j < PROUNDS; ++j)                    // uniform part in soffset: not range-checked, so the OOB sentinel of
            dma_piece(rs0, (lds_ptr_t*)(Pbuf + buf * P_BYTES + (j * NW + wave) * 1024), poff[j], so);   // a lane survives it
    };
    auto dma_stage = [&](int ring, int slab0, int u0) {     // taps u0 .. u0+TPS-1 of the pair starting at slab slab0 (u in 0..17)
        if constexpr (WN == 2) {
#pragma unroll
            for (int t = 0; t < TPS; ++t) {
                const int u = u0 + t, sl = slab0 + u / 9, tap = u % 9;
                dma_piece(rsw, (lds_ptr_t*)(Wbuf + ring * W_BYTES + t * SLAB + wave * 1024), woff, (sl * 9 + tap) * cpad * 64);
            }
        } else {
            const int t = wave >> 2, u = u0 + t, sl = slab0 + u / 9, tap = u % 9;
            dma_piece(rsw, (lds_ptr_t*)(Wbuf + ring * W_BYTES + t * SLAB + (wave & 3) * 1024), woff, (sl * 9 + tap) * cpad * 64);
        }
    };
    f32x4 acc[4][MIW];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    // Fragment addresses = per-lane base + compile-time offset.  A halo row is r = wm*4*PWID + base + fr with `base`
    // known at compile time per (tap, mi) and wm*4*PWID = 0 mod 8, so the swizzle bit ((r>>2)&1) only depends on
    // base & 7 and the lane: eight lane bases cover every tap.  Weight rows are 0 mod 16 + fr: one lane base.
    unsigned pb[8];
    const int rw0 = wm * RPW * PWID;                         // first halo row of this wave's pixels (0 mod 8 for WN = 2, 0 or 4 for WN = 1)
#pragma unroll
    for (int c = 0; c < 8; ++c)
        pb[c] = (unsigned)(rw0 * 64 + fr * 64 + ((fq ^ ((((c + fr + (rw0 & 7)) >> 2) & 1) << 1)) << 4));
    const unsigned wl = (unsigned)(2 * P_BYTES + (wn * 64 + fr) * 64 + ((fq ^ (((fr >> 2) & 1) << 1)) << 4));
    // One stage = two taps = four half-steps of 16 MFMAs (4 channel blocks x 4 pixel fragments).  The fragment reads of
    // half-step h+1 are issued before the MFMAs of half-step h (register double buffer), and a scheduling fence after
    // each half-step keeps the compiler from hoisting more than that (the kernel sits at the 256-VGPR limit: 128 acc).
    f16x8 xa[2][4], wb[2][4];
    auto load_x = [&](f16x8* dst, int pbuf_off, int kh, int kw, int half) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int mi = half * 4 + m;
            const int base = ((mi >> 1) + kh) * PWID + (mi & 1) * HALF + kw;
            dst[m] = *reinterpret_cast<const f16x8*>(smem + pb[base & 7] + (pbuf_off + base * 64));
        }
    };
    auto load_w = [&](f16x8* dst, int wbuf_off) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) dst[ni] = *reinterpret_cast<const f16x8*>(smem + wl + (wbuf_off + ni * 1024));
    };
    auto mma = [&](const f16x8* w, const f16x8* x, int half) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[ni][half * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ni], x[m], acc[ni][half * 4 + m], 0, 0, 0);
    };
    auto stage_compute = [&](int st) {                       // taps u = TPS*st .. of the current slab pair
        const int u0 = TPS * st, u1 = u0 + 1, t0 = u0 % 9, t1 = u1 % 9;
        const int p0 = (u0 / 9) * P_BYTES, p1 = (u1 / 9) * P_BYTES, w0 = (st % 3) * W_BYTES, w1 = w0 + SLAB;
        if constexpr (TPS == 3) {                             // six half-steps, fragments double-buffered one half-step ahead
            const int u2 = u0 + 2, t2 = u2 % 9, p2 = (u2 / 9) * P_BYTES, w2 = w1 + SLAB;
            load_w(wb[0], w0);
            load_x(xa[0], p0, t0 / 3, t0 % 3, 0);
            load_x(xa[1], p0, t0 / 3, t0 % 3, 1);
            mma(wb[0], xa[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(wb[1], w1);
            load_x(xa[0], p1, t1 / 3, t1 % 3, 0);
            mma(wb[0], xa[1], 1);
            __builtin_amdgcn_sched_barrier(0);
            load_x(xa[1], p1, t1 / 3, t1 % 3, 1);
            mma(wb[1], xa[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(wb[0], w2);
            load_x(xa[0], p2, t2 / 3, t2 % 3, 0);
            mma(wb[1], xa[1], 1);
            __builtin_amdgcn_sched_barrier(0);
            load_x(xa[1], p2, t2 / 3, t2 % 3, 1);
            mma(wb[0], xa[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!TAIL) mma(wb[0], xa[1], 1);
        } else if constexpr (WN == 2) {
            load_w(wb[0], w0);
            load_x(xa[0], p0, t0 / 3, t0 % 3, 0);
            load_x(xa[1], p0, t0 / 3, t0 % 3, 1);
            mma(wb[0], xa[0], 0);
            if constexpr (SCHED != 0 && TAIL) {              // region = [tail MFMAs of the previous stage, 12 reads, 16 MFMAs]
#pragma unroll
                for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_w(wb[1], w1);
            load_x(xa[0], p1, t1 / 3, t1 % 3, 0);
            mma(wb[0], xa[1], 1);
            if constexpr (SCHED != 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_x(xa[1], p1, t1 / 3, t1 % 3, 1);
            mma(wb[1], xa[0], 0);
            if constexpr (SCHED != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!TAIL) mma(wb[1], xa[1], 1);                 // TAIL: the last 16 MFMAs are issued behind the stage barrier
        } else {                                             // one half-step (4 fragments = this wave's 64 pixels) per tap
            load_w(wb[0], w0);
            load_x(xa[0], p0, t0 / 3, t0 % 3, 0);
            load_w(wb[1], w1);
            load_x(xa[1], p1, t1 / 3, t1 % 3, 0);
            mma(wb[0], xa[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!TAIL) mma(wb[1], xa[1], 0);
        }
    };

    // Workgroup time over the layers of the benchmark: ~12 us + 1.35 us per stage (18 / 36 / 72 stages: 36 / 60 / 110 us), i.e. a third of
    // an 18-stage workgroup is launch + prologue round trip + epilogue (128 SiLUs per lane at 26 issue cycles each, two waves per
    // SIMD) + stores.  Starting the first round of workgroups in 2 / 4 / 8 phases a fraction of a workgroup apart (so that the CUs'
    // read and write bursts stop coinciding) only added the idle time it cost: +4.4 / +6.4 / +6.6 % on the forward pass.
    // Phase records of a stamped build (tools/stamp_wide.py, 256 tiles): 18-stage workgroups (Cin 128, 64x64 maps) spend 3.7 us before
    // the loop, 24.9 us in it (1.39 us per stage), 3.2 us issuing the epilogue (6.9 us with a residual input) and 0.7 us draining
    // stores = 32.6 us of a 36.2 us slot (the rest is the hand-over between workgroups); 36-stage ones 2.3 + 49.5 + 3.2 + 0.7 of 60.
    // A PERSISTENT form was tried a second time with those figures in hand (one workgroup per CU walking patches, the next patch's
    // bias / halo / first two weight stages requested right behind the last stage barrier so that they land under the epilogue,
    // residual vectors as a ring of four fragments to stay inside 256 VGPRs): 28.9-29.0 ms forward against 28.0, every layer
    // slower (residual layers +14 %).  The hardware's own hand-over between one-patch workgroups is the better pipeline here.
    // A STAGGERED form was tried as well (MI355X_MICROARCH.md "Two waves per SIMD" item 9: waves 4-7 half a stage behind waves 0-3 --
    // between two barriers 16 + 48 MFMAs of two stages for the first group, 32 + 32 for the second; four-slot weight ring addressed
    // at run time, odd halo slab requested a stage later; one stage loop per group, else the two streams cost 580 B of scratch;
    // bit-identical outputs): 29.10 vs 28.79-28.98 ms forward on the same box, 18-stage layers 5 % slower, 36- and 72-stage layers
    // unchanged.  Nor does it matter WHEN the second wave of a SIMD issues its requests (round 3: waves 4-7 issuing theirs in mid-stage,
    // behind their first 32 MFMAs, so that the two waves of a SIMD never wait at the memory pipeline together: every wide layer within
    // +-0.5 % at batch 256).  So the 1.39 us per stage is not a lock-step effect either.  It is the clock: s_memtime / s_memrealtime over the loop
    // give 1.76 GHz for 18-, 36- and 72-stage layers, at which a stage's 2 x 64 MFMAs per SIMD (2048 cycles) would take 1.16 us:
    // the loop runs at 84 % of the matrix-core rate at the clock the chip holds under this load.
    // (tried in round 2, not kept: s_setprio 1 for waves 4-7 before the loop -- static priority for the second-dispatched half,
    // MI355X_MICROARCH.md "Two waves per SIMD" item 4: 8026 vs 8048 tiles/s on the S16k benchmark; and a PERSISTENT form, one
    // workgroup per CU walking the patches with the next patch's halo / first weight stages requested behind the last stage
    // barrier so that their round trip runs under the epilogue: 7807 vs 7851 tiles/s with the same restructured body, which
    // itself cost 2.5 % through spills at the 256-VGPR limit -- the exposed prologue is not where the time goes)
    // the workgroup's BN bias values go to LDS with the very first request (wave 0; it is the oldest of that wave's requests,
    // so every counted wait below covers it): the epilogue then reads them in ~100 cycles instead of paying an exposed L2
    // round trip per workgroup (one workgroup per CU: nothing else would hide it)
    float* const bias_lds = reinterpret_cast<float*>(smem + 2 * P_BYTES + RING * W_BYTES);
    if (wave == 0) {
        const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)pad64(a.Cout) * 4u, 0x00020000);
        dma_piece(rsb, (lds_ptr_t*)bias_lds, lane * 16 < BN * 4 ? (unsigned)(n0 * 4 + lane * 16) : CY_OOB, 0);
    }
    // prologue: halo of slab 0, weight stages 0 and 1
    dma_patch(0, 0);
    dma_stage(0, 0, 0);
    dma_stage(1, 0, TPS);
    CY_WAIT_VM(WPS);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t_loop = stamps ? stamp_real() : 0, c_loop = stamps ? stamp_now() : 0;
#pragma unroll 1
    for (int cp = 0; cp < pairs; ++cp) {
        const bool more = cp + 1 < pairs;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            // stage g = NST*cp + st uses ring slot st % 3; request the weights of stage g+2 and the halo buffers just freed
            const bool has_w = st + 2 < NST || more;
            if (st + 2 < NST) dma_stage((st + 2) % 3, 2 * cp, TPS * (st + 2));
            else if (more) dma_stage((st + 2) % 3, 2 * cp + 2, TPS * (st + 2 - NST));
            // halo: odd slab of this pair (first read in stage NST/2, rounded up) and even slab of the next pair (its buffer
            // is free once the stage holding tap 8 is done)
            constexpr int ST_EVEN = TPS == 2 ? 5 : 3;
            if (st == 0) dma_patch(1, 2 * cp + 1);
            if (st == ST_EVEN && more) dma_patch(0, 2 * cp + 2);
            stage_compute(st);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the weights of stage g+1 (requested first thing in stage g-1) must have landed; younger requests may fly on
            if constexpr (TPS == 2) {
                if (st == 0 || st == 1) { CY_WAIT_VM(WPS + PROUNDS); }
                else if (st == 5 || st == 6) { if (more) { CY_WAIT_VM(WPS + PROUNDS); } else { CY_WAIT_VM(WPS); } }
                else if (has_w) { CY_WAIT_VM(WPS); }
                else { CY_WAIT_VM(0); }
            } else {
                // a halo patch requested in stage 0 / 3 is first read in stage 3 / (next pair's) 0: it may stay in flight for two stages
                if (st == 0 || st == 1) { CY_WAIT_VM(WPS + PROUNDS); }
                else if (st == 3) { if (more) { CY_WAIT_VM(WPS + PROUNDS); } else { CY_WAIT_VM(WPS); } }
                else if (st == 4) { if (more) { CY_WAIT_VM(WPS + PROUNDS); } else { CY_WAIT_VM(0); } }
                else if (has_w) { CY_WAIT_VM(WPS); }
                else { CY_WAIT_VM(0); }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (TAIL) mma(wb[TPS == 3 ? 0 : 1], xa[1], WN == 2 ? 1 : 0);    // operands are in registers: overlaps the next stage's DMA issue / first reads
        }
    }

    const unsigned long long t_epi = stamps ? stamp_real() : 0, c_epi = stamps ? stamp_now() : 0;
    const int cbase = n0 + wn * 64 + fq * 16;
    if constexpr (SPLIT) {
        // fp16x3 epilogue: acc * oscale + bias, SiLU, residual = its high + low halves (requested two pixel fragments at a
        // time: the accumulators leave no room for more), result stored as high / low halves
        float bv[16], sc[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(bias_lds + wn * 64 + fq * 16 + j * 4);
            bv[j * 4] = t[0]; bv[j * 4 + 1] = t[1]; bv[j * 4 + 2] = t[2]; bv[j * 4 + 3] = t[3];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) sc[j] = a.oscale[cbase + j];
        const bool vec = cbase + 16 <= a.Cout;
        const bool res_vec = a.res != nullptr && vec;
        const auto rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0,
                                                           a.res ? (unsigned)((long)a.B * H * W * a.res_ct * 2) : 0u, 0x00020000);
#pragma unroll
        for (int g4 = 0; g4 < MIW; g4 += 2) {
            f16x8 rv[2][4];
            if (res_vec) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int mi = g4 + m;
                    const int y = y0 + wm * RPW + (mi >> 1), x = DUAL ? fr : x0 + (mi & 1) * 16 + fr;
                    const int bb = DUAL ? 2 * b + (mi & 1) : b;
                    const bool ok = y < H && x < W && bb < a.B;
                    const unsigned ro = ok ? (unsigned)((((bb * H + y) * W + x) * a.res_ct + a.res_coff + cbase) * 2) : CY_OOB;
                    rv[m][0] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 0));
                    rv[m][1] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 16));
                    rv[m][2] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, a.res_lo * 2));
                    rv[m][3] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, a.res_lo * 2 + 16));
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int mi = g4 + m;
                const int y = y0 + wm * RPW + (mi >> 1), x = DUAL ? fr : x0 + (mi & 1) * 16 + fr;
                const int bb = DUAL ? 2 * b + (mi & 1) : b;
                if (y >= H || x >= W || bb >= a.B) continue;
                const long pix = ((long)bb * H + y) * W + x;
                float v[16];
                scale_bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bv, sc, a.act != 0, v);
                f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase;
                if (vec) {
                    if (res_vec) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            v[j] += (float)rv[m][0][j] + (float)rv[m][2][j];
                            v[8 + j] += (float)rv[m][1][j] + (float)rv[m][3][j];
                        }
                    }
                    store_split16(dst, a.out_lo, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (cbase + j >= a.Cout) continue;
                        float t = v[j];
                        if (a.res) {
                            const f16* rp = reinterpret_cast<const f16*>(a.res) + pix * a.res_ct + a.res_coff + cbase + j;
                            t += (float)rp[0] + (float)rp[a.res_lo];
                        }
                        store_split1(dst + j, a.out_lo, t);
                    }
                }
            }
        }
    } else {
    // Residual (bottleneck shortcut): all of this wave's 2*MIW vectors are requested up front (the fragment registers are
    // free now), so the epilogue pays ONE memory round trip instead of one per pixel fragment (1 workgroup per CU: nothing
    // else hides it).  Out-of-range pixels read the zero of the buffer range check.
    const bool res_vec = a.res != nullptr && cbase + 16 <= a.Cout;
    f16x8 rv[MIW][2];
    if (res_vec) {
        const auto rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res), 0, (unsigned)((long)a.B * H * W * a.res_ct * 2), 0x00020000);
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
            const int y = y0 + wm * RPW + (mi >> 1), x = DUAL ? fr : x0 + (mi & 1) * 16 + fr;
            const int bb = DUAL ? 2 * b + (mi & 1) : b;
            const bool ok = y < H && x < W && bb < a.B;
            const unsigned ro = ok ? (unsigned)((((bb * H + y) * W + x) * a.res_ct + a.res_coff + cbase) * 2) : CY_OOB;
            rv[mi][0] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 0));
            rv[mi][1] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 16));
        }
    }
    float bv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(bias_lds + wn * 64 + fq * 16 + j * 4);
        bv[j * 4] = t[0]; bv[j * 4 + 1] = t[1]; bv[j * 4 + 2] = t[2]; bv[j * 4 + 3] = t[3];
    }
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi) {
        const int y = y0 + wm * RPW + (mi >> 1), x = DUAL ? fr : x0 + (mi & 1) * 16 + fr;
        const int bb = DUAL ? 2 * b + (mi & 1) : b;
        if (y >= H || x >= W || bb >= a.B) continue;
        const long pix = ((long)bb * H + y) * W + x;
        float v[16];
        bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bv, a.act != 0, v);
        if (cbase + 16 <= a.Cout) {
            f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase;
            if (res_vec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] += (float)rv[mi][0][j]; v[8 + j] += (float)rv[mi][1][j]; }
            }
            f16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
            *reinterpret_cast<f16x8*>(dst) = o0;
            *reinterpret_cast<f16x8*>(dst + 8) = o1;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = cbase + j;
                if (c >= a.Cout) continue;
                float t = v[j];
                if (a.res) t += (float)(reinterpret_cast<const f16*>(a.res)[pix * a.res_ct + a.res_coff + c]);
                reinterpret_cast<f16*>(a.out)[pix * a.out_ct + a.out_coff + c] = (f16)t;
            }
        }
    }
    }
    // diagnostic builds: one record per workgroup (wave 0, plain stores; units of 10 ns): [0] entry -> loop (address setup, prologue
    // round trip), [1] stage loop, [2] epilogue issue, [3] the stage loop again in s_memtime ticks (clock = [3] / [1]).  (Summing with atomics from 16k waves stretched the
    // kernel 4x and the epilogue figures with it.)
    if (stamps) {
        const unsigned long long t_st = stamp_real();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = stamp_real();
        if (tid == 0) {
            unsigned long long* rec = g_wg_stamps + (size_t)(blockIdx.x % WG_STAMP_SLOTS) * 4;
            rec[0] = t_loop - t_entry; rec[1] = t_epi - t_loop; rec[2] = t_st - t_epi; rec[3] = c_epi - c_loop;   // [3]: the loop in s_memtime ticks
        }
    }
}

template <int WN, bool DUAL = false, int TPS = 2, bool SPLIT = false, int SCHED = 0>
static hipError_t launch_wide(const ConvArgs& a, hipStream_t s) {
    constexpr int PR = 18 * (DUAL ? 36 : 34), NPC = (PR + 15) / 16, PROUNDS = (NPC + 7) / 8, BN = 64 * WN;
    const size_t lds = 2 * PROUNDS * 8 * 1024 + 3 * TPS * BN * 64 + 1024;       // halo x2, weight ring, bias
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wide_kernel<false, WN, DUAL, TPS, SPLIT, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wide_kernel<true, WN, DUAL, TPS, SPLIT, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int bx = DUAL ? (a.B + 1) / 2 : a.B * ((a.Wi + 31) / 32);
    const int blocks = bx * ((a.Hi + 15) / 16) * ((pad64(a.Cout) + BN - 1) / BN);
    static const int tail = getenv("CY_WIDE_TAIL") ? atoi(getenv("CY_WIDE_TAIL")) : 1;
    if (tail) hipLaunchKernelGGL((conv3x3_wide_kernel<true, WN, DUAL, TPS, SPLIT, SCHED>), dim3(blocks), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_wide_kernel<false, WN, DUAL, TPS, SPLIT, SCHED>), dim3(blocks), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 3x3 s1, 512 px x 128 ch, 32x32x16 MFMA
// Round 4 A/B experiment (CY_WIDE_MFMA=32): conv3x3_wide_kernel<true, 2> with its K loop on v_mfma_f32_32x32x16_f16 instead of
// v_mfma_f32_16x16x32_f16 -- the same workgroup (16 x 32 px x 128 ch), wave tile (128 px x 64 ch = 4 x 2 blocks of 32 x 32, 128
// accumulator registers), stage stream (halo x2, 3-slot weight ring, two taps per barrier, counted vmcnt, tail MFMAs behind the
// barrier) and the same packed weights; half as many MFMA instructions, each holding the SIMD's vector issue for 8 of its 32 cycles
// instead of 8 of 16.  A lane's fragment is (row lane & 31, 8 channels of K-step s at chunk 2 s + (lane >> 5)), so the 16 lanes of a
// ds_read_b128 group read ONE chunk of 16 rows: the swizzle has to spread a row's chunk over all four 16-byte slots,
// chunk' = chunk ^ f with f = (rx >> 2) & 3 of the halo column rx (not of the linear row index: a tap then shifts the lane's row by a
// compile-time constant and only kw changes f -> six lane bases for the pixels, two for the weights), f = (row >> 4) & 3 for the
// weight rows.  Weight rows are read through the permutation that makes a lane's 16 accumulator registers of a block 16 contiguous
// output channels on the 16x16-ordered packed copy: MFMA row m of block ni <- packed row 16 (m >> 3) + 8 ni + 4 ((m >> 2) & 1) + (m & 3).
// SCHED = 1: the fragment reads of the next K-step are spread between the MFMAs of the current one (one ds_read_b128 per 32-cycle
// MFMA, sched_group_barrier) instead of wherever the compiler sinks them (in front of the last MFMA of the group, latency exposed).
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SCHED>
__global__ __launch_bounds__(512) void conv3x3_wide32_kernel(const ConvArgs a) {
    constexpr int NST = 9, TH = 16, TW = 32, NW = 8, BN = 128, PWID = TW + 2, RPW = 4;
    constexpr int PR = (TH + 2) * PWID, NPC = (PR + 15) / 16, PROUNDS = (NPC + NW - 1) / NW;
    constexpr int P_BYTES = PROUNDS * NW * 1024, SLAB = BN * 64, W_BYTES = 2 * SLAB, RING = 3, WPS = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Pbuf = smem;
    char* const Wbuf = smem + 2 * P_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int H = a.Hi, W = a.Wi;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int cpad = pad128(a.Cout);
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = id % ntn;
    int rest = id / ntn;
    const int tx = rest % tiles_x; rest /= tiles_x;
    const int ty = rest % tiles_y;
    const int b = rest / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
    const int pairs = a.Cin / 64;

    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt32), 0, a.wgt32_bytes, 0x00020000);
    unsigned poff[PROUNDS];
#pragma unroll
    for (int j = 0; j < PROUNDS; ++j) {
        const int r = (j * NW + wave) * 16 + (lane >> 2);
        const int ry = r / PWID, rx = r - ry * PWID;
        const int y = y0 + ry - 1, x = x0 + rx - 1;
        const int q = (lane & 3) ^ ((rx >> 2) & 3);
        const bool ok = r < PR && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        poff[j] = ok ? (unsigned)(((b * H + y) * W + x) * a.in0_ct + a.in0_coff + q * 8) * 2u : CY_OOB;
    }
    unsigned woff;
    {
        const int row = wave * 16 + (lane >> 2);
        woff = (unsigned)((n0 + row) * 64 + ((lane & 3) ^ ((row >> 4) & 3)) * 16);
    }
    auto dma_patch = [&](int buf, int slab) {
#pragma unroll
        for (int j = 0; j < PROUNDS; ++j)
            dma_piece(rs0, (lds_ptr_t*)(Pbuf + buf * P_BYTES + (j * NW + wave) * 1024), poff[j], (unsigned)slab * 64u);
    };
    auto dma_stage = [&](int ring, int slab0, int u0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int u = u0 + t, sl = slab0 + u / 9, tap = u % 9;
            dma_piece(rsw, (lds_ptr_t*)(Wbuf + ring * W_BYTES + t * SLAB + wave * 1024), woff, (sl * 9 + tap) * cpad * 64);
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[ni][mi][j] = 0.f;
    const int lr = lane & 31, lh = lane >> 5;
    unsigned pbx[6], wlx[2];
    const int rw0 = wm * RPW * PWID;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            pbx[kw * 2 + s] = (unsigned)((rw0 + kw + lr) * 64 + (((2 * s + lh) ^ (((kw + lr) >> 2) & 3)) << 4));
    {
        const int p0 = 16 * (lr >> 3) + 4 * ((lr >> 2) & 1) + (lr & 3);
#pragma unroll
        for (int s = 0; s < 2; ++s) wlx[s] = (unsigned)(2 * P_BYTES + (wn * 64 + p0) * 64 + (((2 * s + lh) ^ (lr >> 3)) << 4));
    }
    f16x8 wa[2][2], xb[2][4];
    auto load_k = [&](int bi, int pbuf_off, int wbuf_off, int kh, int kw, int s) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) wa[bi][ni] = *reinterpret_cast<const f16x8*>(smem + wlx[s] + (wbuf_off + ni * 512));
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xb[bi][mi] = *reinterpret_cast<const f16x8*>(smem + pbx[kw * 2 + s] + (pbuf_off + (mi + kh) * PWID * 64));
    };
    auto mma = [&](int bi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[bi][ni], xb[bi][mi], acc[ni][mi], 0, 0, 0);
    };
    auto stage_compute = [&](int st) {
        const int u0 = 2 * st, u1 = u0 + 1, t0 = u0 % 9, t1 = u1 % 9;
        const int p0 = (u0 / 9) * P_BYTES, p1 = (u1 / 9) * P_BYTES, w0 = (st % 3) * W_BYTES, w1 = w0 + SLAB;
        load_k(0, p0, w0, t0 / 3, t0 % 3, 0);
        load_k(1, p0, w0, t0 / 3, t0 % 3, 1);
        mma(0);
        if constexpr (SCHED != 0) {                          // region = [tail MFMAs of the previous stage, 12 reads, 8 MFMAs]
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_k(0, p1, w1, t1 / 3, t1 % 3, 0);
        mma(1);
        if constexpr (SCHED != 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_k(1, p1, w1, t1 / 3, t1 % 3, 1);
        mma(0);
        if constexpr (SCHED != 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                  // the last 8 MFMAs of the stage are issued behind the stage barrier
    };

    float* const bias_lds = reinterpret_cast<float*>(smem + 2 * P_BYTES + RING * W_BYTES);
    if (wave == 0) {
        const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)pad64(a.Cout) * 4u, 0x00020000);
        dma_piece(rsb, (lds_ptr_t*)bias_lds, lane * 16 < BN * 4 ? (unsigned)(n0 * 4 + lane * 16) : CY_OOB, 0);
    }
    dma_patch(0, 0);
    dma_stage(0, 0, 0);
    dma_stage(1, 0, 2);
    CY_WAIT_VM(WPS);
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int cp = 0; cp < pairs; ++cp) {
        const bool more = cp + 1 < pairs;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const bool has_w = st + 2 < NST || more;
            if (st + 2 < NST) dma_stage((st + 2) % 3, 2 * cp, 2 * (st + 2));
            else if (more) dma_stage((st + 2) % 3, 2 * cp + 2, 2 * (st + 2 - NST));
            if (st == 0) dma_patch(1, 2 * cp + 1);
            if (st == 5 && more) dma_patch(0, 2 * cp + 2);
            stage_compute(st);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (st == 0 || st == 1) { CY_WAIT_VM(WPS + PROUNDS); }
            else if (st == 5 || st == 6) { if (more) { CY_WAIT_VM(WPS + PROUNDS); } else { CY_WAIT_VM(WPS); } }
            else if (has_w) { CY_WAIT_VM(WPS); }
            else { CY_WAIT_VM(0); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
        }
    }

    // epilogue: lane (lr, lh) holds, per block (ni, mi), channels n0 + wn*64 + ni*32 + lh*16 + 0..15 of pixel (y0 + wm*4 + mi, x0 + lr)
    const int x = x0 + lr;
    const auto rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.in0), 0,
                                                       a.res ? (unsigned)((long)a.B * H * W * a.res_ct * 2) : 0u, 0x00020000);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int cl = wn * 64 + ni * 32 + lh * 16, cbase = n0 + cl;
        const bool vec = cbase + 16 <= a.Cout, res_vec = a.res != nullptr && vec;
        f16x8 rv[4][2];
        if (res_vec) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int y = y0 + wm * RPW + mi;
                const bool ok = y < H && x < W;
                const unsigned ro = ok ? (unsigned)((((b * H + y) * W + x) * a.res_ct + a.res_coff + cbase) * 2) : CY_OOB;
                rv[mi][0] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 0));
                rv[mi][1] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 16));
            }
        }
        float bv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(bias_lds + cl + j * 4);
            bv[j * 4] = t[0]; bv[j * 4 + 1] = t[1]; bv[j * 4 + 2] = t[2]; bv[j * 4 + 3] = t[3];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int y = y0 + wm * RPW + mi;
            if (y >= H || x >= W) continue;
            const long pix = ((long)b * H + y) * W + x;
            const f32x16& c = acc[ni][mi];
            float v[16];
            bias_act16(f32x4{c[0], c[1], c[2], c[3]}, f32x4{c[4], c[5], c[6], c[7]}, f32x4{c[8], c[9], c[10], c[11]},
                       f32x4{c[12], c[13], c[14], c[15]}, bv, a.act != 0, v);
            if (vec) {
                f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase;
                if (res_vec) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] += (float)rv[mi][0][j]; v[8 + j] += (float)rv[mi][1][j]; }
                }
                f16x8 o0, o1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
                *reinterpret_cast<f16x8*>(dst) = o0;
                *reinterpret_cast<f16x8*>(dst + 8) = o1;
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int ch = cbase + j;
                    if (ch >= a.Cout) continue;
                    float t = v[j];
                    if (a.res) t += (float)(reinterpret_cast<const f16*>(a.res)[pix * a.res_ct + a.res_coff + ch]);
                    reinterpret_cast<f16*>(a.out)[pix * a.out_ct + a.out_coff + ch] = (f16)t;
                }
            }
        }
    }
}

template <int SCHED>
static hipError_t launch_wide32(const ConvArgs& a, hipStream_t s) {
    constexpr int PR = 18 * 34, NPC = (PR + 15) / 16, PROUNDS = (NPC + 7) / 8;
    const size_t lds = 2 * PROUNDS * 8 * 1024 + 3 * 2 * 128 * 64 + 1024;        // halo x2, weight ring, bias
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wide32_kernel<SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int blocks = a.B * ((a.Wi + 31) / 32) * ((a.Hi + 15) / 16) * ((pad64(a.Cout) + 127) / 128);
    hipLaunchKernelGGL(conv3x3_wide32_kernel<SCHED>, dim3(blocks), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 3x3 s1, 512 px x 128 ch, persistent
// conv3x3_wide_kernel<true, 2> as ONE workgroup per CU walking the patches (patch n of a workgroup = blockIdx.x + n * gridDim.x
// in the same XCD-aware order).  What a one-patch workgroup pays outside its stage loop -- per-workgroup records of the stamped
// build: 2.3-3.7 us from entry to the first MFMA (the 70 KiB prologue arrives at the ~11 B/cycle a CU gets when every CU asks at
// once), 3.2-6.9 us of epilogue, 0.7 us of store drain and 3.6-4.3 us until the next workgroup runs on the CU -- is a third of an
// 18-stage (Cin = 128) patch.  Here the NEXT patch's prologue (bias, first halo slab, first two weight stages) is requested right
// behind the last stage barrier and lands under the epilogue of the current patch, and there is no hand-over.
// Two earlier persistent forms lost to the one-patch kernel (+3 %, residual layers +14 %).  The vector-memory counter is in-order:
// a residual load issued after the next patch's 70 KiB of LDS-DMA cannot be waited for without waiting for all of that DMA, so
// the epilogue stood still for the whole prologue.  Here the order of requests is  bias(next) -> residual(this) -> halo / weights
// (next)  and the epilogue waits with a counted vmcnt that leaves exactly the DMA in flight; the output stores are buffer stores
// issued by every lane (out-of-range lanes carry the out-of-range offset and are dropped by the range check), so the number of
// operations between the DMA and the next patch's first wait is a compile-time constant too.
// STRIP form (STRIP = halo pieces per requesting wave and slab: 10 for maps up to 62 px wide, 12 up to 95): a workgroup owns 512
// consecutive ENTRIES of the batch flattened to one dimension -- image after image, row after row, with ONE zero entry after each
// row and one zero row after each image (row stride RS = W + 1, image stride IS = (H + 1) * RS).  Zero padding of a 3x3 filter is
// then just those shared zero entries: tap (kh, kw) of entry e is entry e + (kh - 1) * RS + (kw - 1) for EVERY entry, so any 16
// consecutive entries are an MFMA pixel fragment and a map of any size is covered with (RS / W) * ((H + 1) / H) - 1 idle lanes
// (80 x 80: 2.5 %, 40 x 40: 5 %, 20 x 20: 10 %, 52 x 64 of a ragged tile: 3.5 %) instead of the 17-61 % that 16 x 32-pixel patches
// waste on 80 / 40 / 20-px maps.  Price: a halo of RS + 1 entries on both sides (676 staged entries for 80-px rows against 612).
struct StripGeo { int RS, IS, NE; unsigned long long mRS, mIS; };      // NE = B * IS entries; m* = ceil(2^40 / divisor)
__device__ __forceinline__ int strip_div(int e, unsigned long long m) { return (int)(((unsigned long long)(unsigned)e * m) >> 40); }

template <bool SPLIT, bool RES, int STRIP = 0>
__global__ __launch_bounds__(512) void conv3x3_widep_kernel(const ConvArgs a, const int total, const StripGeo sg) {
    // SPLIT (fp16x3 context): three passes over the halo slabs [x_lo | x_hi | x_hi] against the packed weight slabs [w_hi | w_lo | w_hi]
    // (see conv3x3_wide_kernel); epilogue acc * oscale + bias, SiLU, + residual (hi + lo), stored as high / low halves
    constexpr int NST = 9, TH = 16, TW = 32, NW = 8, BN = 128, PWID = TW + 2;
    // Only waves 0-3 (one per SIMD) issue LDS-DMA.  A wave whose request does not fit the memory pipeline waits at the issue, and
    // cannot issue MFMAs meanwhile; with every wave issuing its share at the same point of a stage both waves of a SIMD wait
    // together and the matrix pipe idles (a slab pair WITH its requests takes 13.3-15.6 us, the last pair of a patch, which has
    // none left to issue, 9 us).  With the requests on one wave per SIMD its partner (waves 4-7) has the pipe meanwhile.
    constexpr int NDW = 4;                                   // requesting waves
    constexpr int RPW = 4, MIW = 8, WPS = 2 * (NW / NDW);    // weight pieces per requesting wave and stage
    constexpr int PR = (TH + 2) * PWID, NPC = (PR + 15) / 16, PROUNDS = STRIP ? STRIP : (NPC + NDW - 1) / NDW;      // halo pieces per requesting wave and slab
    constexpr int P_BYTES = PROUNDS * NDW * 1024, SLAB = BN * 64, W_BYTES = 2 * SLAB, RING = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Pbuf = smem;
    char* const Wbuf = smem + 2 * P_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int H = a.Hi, W = a.Wi;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int cpad = pad128(a.Cout);
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int ppairs = a.Cin / 64;                          // slab pairs of one pass over the input channels
    const int pairs = SPLIT ? a.split * ppairs : ppairs;     // a.split = passes of the fp16x3 context (3, or 2 for fp16-exact weights)
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in0), 0, a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt32), 0, a.wgt32_bytes, 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)pad64(a.Cout) * 4u, 0x00020000);
    const auto rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SPLIT ? a.oscale : a.bias), 0, (unsigned)pad64(a.Cout) * 4u, 0x00020000);
    const unsigned out_bytes = (unsigned)((long)a.B * H * W * a.out_ct * 2);
    const auto rso = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, out_bytes, 0x00020000);
    const auto rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? a.res : a.in0), 0,
                                                       RES ? (unsigned)((long)a.B * H * W * a.res_ct * 2) : 0u, 0x00020000);
    const int fr = lane & 15, fq = lane >> 4;

    // patch v -> (image, first row / column, first channel)
    int pb_, py0, px0, pn0;
    auto geometry = [&](int v) {
        const int id = xcd_remap(v, total);
        pn0 = (id % ntn) * BN;
        int rest = id / ntn;
        if constexpr (STRIP != 0) { px0 = rest * 512; py0 = 0; pb_ = 0; return; }       // px0 = first entry of the strip
        px0 = (rest % tiles_x) * TW; rest /= tiles_x;
        py0 = (rest % tiles_y) * TH;
        pb_ = rest / tiles_y;
    };
    // developer ablation (diagnostic builds; results invalid): bit 0 no requests inside the stage loop, bit 1 no fragment reads, bit 2 no MFMAs
    const bool no_dma = CY_STAMPS_ENABLED && (a.dbg & 1), no_rd = CY_STAMPS_ENABLED && (a.dbg & 2), no_mma = CY_STAMPS_ENABLED && (a.dbg & 4);
    const bool dmaw = wave < NDW && !no_dma;
    // halo piece j of a requesting wave: entries (j * NDW + wave) * 16 + (lane >> 2) of the 18 x 34 halo; the offsets are computed at
    // the request (a dozen VALU instructions of a wave that is about to wait for the memory pipeline anyway) instead of living in
    // ten registers next to the 128 accumulators
    auto dma_patch = [&](int buf, int slab, int gb, int gy0, int gx0) {
        if (!dmaw) return;
        int l4 = lane >> 2;
        asm volatile("" : "+v"(l4));                         // (keeps the optimiser from hoisting the ten offsets out of the stage loop)
        const unsigned lq = (unsigned)(lane & 3);
        unsigned slab_so = (unsigned)slab * 64u;             // fp16x3: virtual slab -> physical slab (+ the low halves in the first pass)
        if constexpr (SPLIT) {
            int lo;
            slab_so = (unsigned)x3_chunk(slab, 2 * ppairs, lo) * 64u;
            if (lo) slab_so += (unsigned)a.in0_lo * 2u;
        }
        if constexpr (STRIP != 0) {
            // staged entry le of the strip = entry gx0 - RS - 1 + le of the flattened batch -> (image, row, column) or a zero
            const int hl = 514 + 2 * sg.RS;
#pragma unroll
            for (int j = 0; j < PROUNDS; ++j) {
                const int le = (j * NDW + wave) * 16 + l4;
                const int e = gx0 - sg.RS - 1 + le;
                const bool in = le < hl && (unsigned)e < (unsigned)sg.NE;
                const int ec = in ? e : 0;
                const int bb = strip_div(ec, sg.mIS), r = ec - bb * sg.IS;
                const int y = strip_div(r, sg.mRS), x = r - y * sg.RS;
                const unsigned q = lq ^ (unsigned)(((le >> 2) & 1) << 1);
                const bool ok = in && y < H && x < W;
                const unsigned off = ok ? (unsigned)(((bb * H + y) * W + x) * a.in0_ct + a.in0_coff + (int)q * 8) * 2u : CY_OOB;
                dma_piece(rs0, (lds_ptr_t*)(Pbuf + buf * P_BYTES + (j * NDW + wave) * 1024), off, slab_so);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < PROUNDS; ++j) {
            const int r = (j * NDW + wave) * 16 + l4;
            const int ry = (r * 1928) >> 16, rx = r - ry * PWID;       // r / 34 for r < 1024
            const int y = gy0 + ry - 1, x = gx0 + rx - 1;
            const unsigned q = lq ^ (unsigned)(((r >> 2) & 1) << 1);
            const bool ok = r < PR && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const unsigned off = ok ? (unsigned)(((gb * H + y) * W + x) * a.in0_ct + a.in0_coff + (int)q * 8) * 2u : CY_OOB;
            dma_piece(rs0, (lds_ptr_t*)(Pbuf + buf * P_BYTES + (j * NDW + wave) * 1024), off, slab_so);
        }
    };
    // (lane-dependent LDS / weight addresses are recomputed at the top of every patch instead of living through the epilogue,
    // where the 128 accumulators + 64 residual registers leave no room for them)
    unsigned woff = 0;
    auto weight_lane_offset = [&]() { const int row = wave * 32 + (lane >> 2); woff = (unsigned)(row * 64 + ((lane & 3) ^ (((row >> 2) & 1) << 1)) * 16); };
    auto dma_stage = [&](int ring, int slab0, int u0, int n0) {      // requesting wave w: rows w*32 .. w*32+31 of both taps
        if (!dmaw) return;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int u = u0 + t, sl = slab0 + u / 9, tap = u % 9;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                dma_piece(rsw, (lds_ptr_t*)(Wbuf + ring * W_BYTES + t * SLAB + (wave * 2 + h) * 1024), woff,
                          (unsigned)((sl * 9 + tap) * cpad * 64 + n0 * 64 + h * 1024));
        }
    };
    // two slots of 2 KiB, alternating per patch: [bias 512 B | 512 B written with zeros | oscale 512 B (fp16x3) | zeros]
    float* const bias_lds = reinterpret_cast<float*>(smem + 2 * P_BYTES + RING * W_BYTES);
    auto dma_bias = [&](int slot, int n0) {
        if (wave != 0) return;
        dma_piece(rsb, (lds_ptr_t*)(bias_lds + slot * 512), lane * 16 < BN * 4 ? (unsigned)(n0 * 4 + lane * 16) : CY_OOB, 0);
        if constexpr (SPLIT) dma_piece(rss, (lds_ptr_t*)(bias_lds + slot * 512 + 256), lane * 16 < BN * 4 ? (unsigned)(n0 * 4 + lane * 16) : CY_OOB, 0);
    };

    f32x4 acc[4][MIW];
    unsigned pb[STRIP ? 9 : 8], wl = 0;
    const int rw0 = wm * RPW * PWID;
    auto fragment_bases = [&]() {
        // opaque to the optimiser: otherwise it hoists these 9 registers out of the patch loop and spills around the epilogue
        int lz = lane;
        asm volatile("" : "+v"(lz));
        const int r_ = lz & 15, q_ = lz >> 4;
        if constexpr (STRIP != 0) {
            // strip form: one lane base per TAP -- staged row of (tap, fragment 0) = kh * RS + kw + wm * 128 + fr; fragment m is 16 m rows
            // (1 KiB, same swizzle phase) further on
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int R0 = (t / 3) * sg.RS + (t % 3) + wm * 128 + r_;
                pb[t] = (unsigned)(R0 * 64 + ((q_ ^ (((R0 >> 2) & 1) << 1)) << 4));
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                pb[c] = (unsigned)(rw0 * 64 + r_ * 64 + ((q_ ^ ((((c + r_ + (rw0 & 7)) >> 2) & 1) << 1)) << 4));
        }
        wl = (unsigned)(2 * P_BYTES + (wn * 64 + r_) * 64 + ((q_ ^ (((r_ >> 2) & 1) << 1)) << 4));
    };
    f16x8 xa[2][4], wb[2][4];
    auto load_x = [&](f16x8* dst, int pbuf_off, int kh, int kw, int half) {
        if (no_rd) return;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int mi = half * 4 + m;
            if constexpr (STRIP != 0) {
                dst[m] = *reinterpret_cast<const f16x8*>(smem + pb[kh * 3 + kw] + (pbuf_off + mi * 1024));
            } else {
                const int base = ((mi >> 1) + kh) * PWID + (mi & 1) * 16 + kw;
                dst[m] = *reinterpret_cast<const f16x8*>(smem + pb[base & 7] + (pbuf_off + base * 64));
            }
        }
    };
    auto load_w = [&](f16x8* dst, int wbuf_off) {
        if (no_rd) return;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) dst[ni] = *reinterpret_cast<const f16x8*>(smem + wl + (wbuf_off + ni * 1024));
    };
    auto mma = [&](const f16x8* w, const f16x8* x, int half) {
        if (no_mma) return;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[ni][half * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ni], x[m], acc[ni][half * 4 + m], 0, 0, 0);
    };
    auto stage_compute = [&](int st) {
        const int u0 = 2 * st, u1 = u0 + 1, t0 = u0 % 9, t1 = u1 % 9;
        const int p0 = (u0 / 9) * P_BYTES, p1 = (u1 / 9) * P_BYTES, w0 = (st % 3) * W_BYTES, w1 = w0 + SLAB;
        load_w(wb[0], w0);
        load_x(xa[0], p0, t0 / 3, t0 % 3, 0);
        load_x(xa[1], p0, t0 / 3, t0 % 3, 1);
        mma(wb[0], xa[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        load_w(wb[1], w1);
        load_x(xa[0], p1, t1 / 3, t1 % 3, 0);
        mma(wb[0], xa[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        load_x(xa[1], p1, t1 / 3, t1 % 3, 1);
        mma(wb[1], xa[0], 0);
        __builtin_amdgcn_sched_barrier(0);                  // the last 16 MFMAs of the stage are issued behind the stage barrier
    };

    const bool stamps = CY_STAMPS_ENABLED && (a.dbg & 64) != 0;      // diagnostic builds: phase records of the workgroup's SECOND patch
    unsigned long long t_l0 = 0, t_l1 = 0, t_rw = 0, t_ep = 0;
    const unsigned long long t_begin = stamps ? stamp_real() : 0;
    // ---- first patch: prologue as in the one-patch kernel
    int v = blockIdx.x, n = 0;
    if (v >= total) return;
    geometry(v);
    weight_lane_offset();
    dma_bias(0, pn0);
    dma_patch(0, 0, pb_, py0, px0);
    dma_stage(0, 0, 0, pn0);
    dma_stage(1, 0, 2, pn0);
    CY_WAIT_VM(WPS);
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (;;) {
        const int b = pb_, y0 = py0, x0 = px0, n0 = pn0;
        const int vn = v + (int)gridDim.x;
        const bool next = vn < total;
        if (stamps) {
            const unsigned long long t = stamp_real();
            if (n == 2 && tid == 0) {                             // [0] requests + residual wait, [1] stage loop, [2] epilogue, [3] (unused)
                unsigned long long* rec = g_wg_stamps + (size_t)(blockIdx.x % WG_STAMP_SLOTS) * 4;
                rec[0] = t_rw - t_l1; rec[1] = t_l1 - t_l0; rec[2] = t_ep - t_rw;
            }
            t_l0 = t;
        }
        fragment_bases();
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The stream of stages never drains at a patch boundary: in the LAST slab pair of a patch the requests that would fetch
        // "the next pair" fetch the first halo slab and the first two weight stages of the NEXT patch instead, at the same steady
        // rate (a burst of the whole 70 KiB prologue holds the issuing waves for 3.3 us: the memory pipeline takes ~11 B per cycle
        // and CU, and a wave whose request does not fit waits).  The halo offsets of the next patch replace this patch's in the same
        // registers once its last halo request is out (stage 0 of the last pair).
#pragma unroll 1
        for (int cp = 0; cp < pairs; ++cp) {
            const bool more = cp + 1 < pairs;
            const bool chain = more || next;                     // something follows this pair
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const bool has_w = st + 2 < NST || chain;
                if (st + 2 < NST) dma_stage((st + 2) % 3, 2 * cp, 2 * (st + 2), n0);
                else if (more) dma_stage((st + 2) % 3, 2 * cp + 2, 2 * (st + 2 - NST), n0);
                else if (next) dma_stage((st + 2) % 3, 0, 2 * (st + 2 - NST), pn0);
                if (st == 0) dma_patch(1, 2 * cp + 1, b, y0, x0);
                if (st == 1 && !more && next) { geometry(vn); dma_bias((n + 1) & 1, pn0); }
                if (st == 5 && more) dma_patch(0, 2 * cp + 2, b, y0, x0);
                if (st == 5 && !more && next) dma_patch(0, 0, pb_, py0, px0);
                stage_compute(st);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (st == 0 || st == 1) { CY_WAIT_VM(WPS + PROUNDS); }
                else if (st == 5 || st == 6) { if (chain) { CY_WAIT_VM(WPS + PROUNDS); } else { CY_WAIT_VM(WPS); } }
                else if (has_w) { CY_WAIT_VM(WPS); }
                else { CY_WAIT_VM(0); }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                mma(wb[1], xa[1], 1);
            }
        }
        if (stamps) t_l1 = stamp_real();
        // ---- epilogue of this patch.  Stage 0 of the next patch needs nothing but what the last barrier has already published.
        const float* bl = bias_lds + (n & 1) * 512 + wn * 64 + fq * 16;
        float bv[16];
        if constexpr (!SPLIT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(bl + j * 4);
                bv[j * 4] = t[0]; bv[j * 4 + 1] = t[1]; bv[j * 4 + 2] = t[2]; bv[j * 4 + 3] = t[3];
            }
        }
        const int cbase = n0 + wn * 64 + fq * 16;
        const bool chan_ok = cbase + 16 <= a.Cout;
        // pixel index of fragment mi of this lane = pix0 + (mi >> 1) * W + (mi & 1) * 16; rows past H are uniform per fragment pair
        const int xl = x0 + fr;
        const unsigned pix0 = (unsigned)((b * H + y0 + wm * RPW) * W + xl);
        const bool okx0 = xl < W && chan_ok, okx1 = xl + 16 < W && chan_ok;
        const unsigned cb2 = (unsigned)cbase * 2u;
        // strip form: entry of (fragment mi, this lane) -> (image, row, column) by two magic divisions (any map size: the 128 entries of a
        // wave may span several rows and, on maps of a few pixels, several images)
        const int e0 = x0 + wm * 128 + fr;
        auto pixel = [&](int mi, unsigned& pix) -> bool {
            if constexpr (STRIP != 0) {
                const int e = e0 + 16 * mi;
                const int ec = e < sg.NE ? e : 0;
                const int bb = strip_div(ec, sg.mIS), r = ec - bb * sg.IS;
                const int yy = strip_div(r, sg.mRS), xx = r - yy * sg.RS;
                pix = (unsigned)((bb * H + yy) * W + xx);
                return e < sg.NE && yy < H && xx < W && chan_ok;
            } else {
                pix = pix0 + (unsigned)((mi >> 1) * W + (mi & 1) * 16);
                return (y0 + wm * RPW + (mi >> 1) < H) && ((mi & 1) ? okx1 : okx0);
            }
        };
        if constexpr (SPLIT) {
            // residual = high + low halves: four 16-byte loads per pixel fragment, requested two fragments ahead of the arithmetic (a ring
            // of three fragments: 48 registers; the compiler places the waits).  The stores are issued by every lane (out-of-range ones
            // are dropped by the range check).
            f16x8 rq[3][4];
            auto res_load = [&](int mi) {
                unsigned pix;
                const bool ok = pixel(mi, pix);
                const unsigned ro = ok ? (pix * (unsigned)a.res_ct + (unsigned)a.res_coff) * 2u + cb2 : CY_OOB;
                rq[mi % 3][0] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 0));
                rq[mi % 3][1] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 16));
                rq[mi % 3][2] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, a.res_lo * 2));
                rq[mi % 3][3] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, a.res_lo * 2 + 16));
            };
            if (RES) { res_load(0); res_load(1); }
            if (stamps) t_rw = stamp_real();
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                if (RES && mi + 2 < MIW) res_load(mi + 2);
                float bg[16], sc[16];          // (re-read from the LDS slot per fragment: 32 registers that need not live through the epilogue)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(bl + j * 4), u = *reinterpret_cast<const f32x4*>(bl + 256 + j * 4);
                    bg[j * 4] = t[0]; bg[j * 4 + 1] = t[1]; bg[j * 4 + 2] = t[2]; bg[j * 4 + 3] = t[3];
                    sc[j * 4] = u[0]; sc[j * 4 + 1] = u[1]; sc[j * 4 + 2] = u[2]; sc[j * 4 + 3] = u[3];
                }
                float vv[16];
                scale_bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bg, sc, a.act != 0, vv);
                if (RES) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        vv[j] += (float)rq[mi % 3][0][j] + (float)rq[mi % 3][2][j];
                        vv[8 + j] += (float)rq[mi % 3][1][j] + (float)rq[mi % 3][3][j];
                    }
                }
                f16x8 h0, h1, l0, l1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    h0[j] = (f16)vv[j]; h1[j] = (f16)vv[8 + j];
                    l0[j] = (f16)(vv[j] - (float)h0[j]); l1[j] = (f16)(vv[8 + j] - (float)h1[j]);
                }
                unsigned pix;
                const bool ok = pixel(mi, pix);
                const unsigned so = ok ? (pix * (unsigned)a.out_ct + (unsigned)a.out_coff) * 2u + cb2 : CY_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h0), rso, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h1), rso, so, 16, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, l0), rso, so, a.out_lo * 2, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, l1), rso, so, a.out_lo * 2 + 16, 0);
            }
        } else {
        f16x8 rv[MIW][2];
        if (RES) {
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                unsigned pix;
                const bool ok = pixel(mi, pix);
                const unsigned ro = ok ? (pix * (unsigned)a.res_ct + (unsigned)a.res_coff) * 2u + cb2 : CY_OOB;
                rv[mi][0] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 0));
                rv[mi][1] = __builtin_bit_cast(f16x8, load_b128(rsr, ro, 16));
            }
            CY_WAIT_VM(0);                                       // (older than them: only the second weight stage of the next patch)
        }
        if (stamps) t_rw = stamp_real();
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
            float vv[16];
            bias_act16(acc[0][mi], acc[1][mi], acc[2][mi], acc[3][mi], bv, a.act != 0, vv);
            if (RES) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { vv[j] += (float)rv[mi][0][j]; vv[8 + j] += (float)rv[mi][1][j]; }
            }
            f16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (f16)vv[j]; o1[j] = (f16)vv[8 + j]; }
            unsigned pix;
            const bool ok = pixel(mi, pix);
            const unsigned so = ok ? (pix * (unsigned)a.out_ct + (unsigned)a.out_coff) * 2u + cb2 : CY_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rso, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rso, so, 16, 0);
        }
        }
        if (stamps) t_ep = stamp_real();
        if (stamps && !next && tid == 0) g_wg_stamps[(size_t)(blockIdx.x % WG_STAMP_SLOTS) * 4 + 3] = t_ep - t_begin;    // [3] the workgroup's whole life
        if (!next) break;
        v = vn; ++n;
    }
}

template <int STRIP, bool SPLIT = false>
static hipError_t launch_widep_t(const ConvArgs& a, hipStream_t s, int total, const StripGeo& sg) {
    constexpr int PR = 18 * 34, NPC = (PR + 15) / 16, PROUNDS = STRIP ? STRIP : (NPC + 3) / 4;
    const size_t lds = 2 * PROUNDS * 4 * 1024 + 3 * 2 * 128 * 64 + 4096;        // halo x2, weight ring, bias (+ oscale) x2
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_widep_kernel<SPLIT, false, STRIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_widep_kernel<SPLIT, true, STRIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    static const int ncu = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int grid = total < ncu ? total : ncu;
    if (a.res) hipLaunchKernelGGL((conv3x3_widep_kernel<SPLIT, true, STRIP>), dim3(grid), dim3(512), lds, s, a, total, sg);
    else hipLaunchKernelGGL((conv3x3_widep_kernel<SPLIT, false, STRIP>), dim3(grid), dim3(512), lds, s, a, total, sg);
    return hipGetLastError();
}

static hipError_t launch_widep(const ConvArgs& a, hipStream_t s) {
    const int total = a.B * ((a.Wi + 31) / 32) * ((a.Hi + 15) / 16) * ((pad64(a.Cout) + 127) / 128);
    return a.split ? launch_widep_t<0, true>(a, s, total, StripGeo{}) : launch_widep_t<0>(a, s, total, StripGeo{});
}

// idle share of the MFMA lanes of the 2-D patch form (16 x 32-pixel patches) and of the strip form on an H x W map
static double wide2d_cover(int H, int W) { return (double)((H + 15) / 16 * 16) * ((W + 31) / 32 * 32) / ((double)H * W); }
static double strip_cover(int H, int W) { return (double)(H + 1) * (W + 1) / ((double)H * W); }
// maps from 17 px wide (narrower ones down to 13 px belong to the dual-image / two-tap kernels), and the maps of a few pixels that no
// patch shape fits (8 x 8 and 4 x 4: the stride-16 / 32 levels of 128- and 256-px inputs): 27 % / 56 % idle lanes there instead of 75 % / 94 %
static bool strip_small(const ConvArgs& a) { return a.Wi >= 3 && a.Wi <= 12 && a.Hi >= 3 && a.Hi <= 126; }
static bool strip_fits(const ConvArgs& a) {
    return ((a.Wi >= 17 && a.Hi >= 17 && a.Wi <= 126) || strip_small(a)) && a.Cout % 16 == 0 && (long)a.B * (a.Hi + 1) * (a.Wi + 1) < (1L << 22);
}

static hipError_t launch_strip(const ConvArgs& a, hipStream_t s) {
    StripGeo sg;
    sg.RS = a.Wi + 1; sg.IS = (a.Hi + 1) * sg.RS; sg.NE = a.B * sg.IS;
    sg.mRS = ((1ull << 40) + sg.RS - 1) / sg.RS; sg.mIS = ((1ull << 40) + sg.IS - 1) / sg.IS;
    const int total = ((sg.NE + 511) / 512) * ((pad64(a.Cout) + 127) / 128);
    if (a.split) return a.Wi <= 62 ? launch_widep_t<10, true>(a, s, total, sg) : launch_widep_t<12, true>(a, s, total, sg);
    return a.Wi <= 62 ? launch_widep_t<10>(a, s, total, sg) : launch_widep_t<12>(a, s, total, sg);
}

// ------------------------------------------------------------------------------------------------ 1x1 (and 3x3 stride 2), pixels direct to registers
// A 1x1 convolution has no tap reuse, so staging the pixel operand through LDS only costs: in the tiled kernels above two
// thirds of the LDS-DMA pieces (60-180 issue cycles each) carry pixels that exactly one wave reads exactly once.  In NHWC
// the MFMA B-operand fragment of a lane (pixel fr, channels 8*(fq+4*kk)..+7) is 16 contiguous bytes in global memory, so
// here every wave loads its own pixels straight into registers (buffer_load_dwordx4, zero fill from the range check,
// RING-1 K chunks ahead) and only the weights - shared by all eight waves - go through LDS (RING slots, requested RING-1
// chunks ahead, counted vmcnt).  A wave owns MI*16 pixels x all BN = 64*NB channels of the workgroup's tile:
//   per 64-channel K chunk and wave: NB DMA pieces + 2*MI register loads for 8*NB*MI MFMAs (NB=4, MI=2: 8 for 64).
// Also handles the two-segment input (nearest-x2 upsample + concat) of layers 12 and 15: per-lane addresses anyway.
// FUSE2 (round 4; NB = 4, the tile holds all 256 output channels of its pixels): back-to-back fusion with the 1x1 convolution that is
// this layer's only reader (yolov8 model.3 -> model.4.cv1).  After the K loop a lane holds, per 64-channel block g, the 16 contiguous
// channels 64 g + 16 fq + 0..15 of its pixels: bias + SiLU, rounded to fp16 exactly as the store would, they ARE the MFMA B operand of
// the second GEMM for K chunk g once that layer's input channels are permuted at pack time (pack_weights_fused2) -- no LDS, no lane
// movement.  The second layer's four weight chunks (128 KB) are requested into the dead ring (+ one slot) right behind the loop and
// land under the first epilogue's arithmetic; its result is stored through the same epilogue form.  The 256-channel map between the
// two layers (2 MB per 512^2 tile) is never written or read.
template <int NB, int MI, int RING, bool K3, bool SPLIT = false, int NDW = 8, bool FUSE2 = false>
__global__ __launch_bounds__(512, (NB * MI <= 4 && RING == 2) ? 2 : 1) void conv1x1_direct_kernel(const ConvArgs a) {
    static_assert(!FUSE2 || (NB == 4 && !SPLIT), "back-to-back fusion: 256-channel tile, fp16 context");
    // NDW = number of waves that request the weight chunks (LDS-DMA): 8 = every wave its share; 4 = waves 0-3 only (one per SIMD),
    // so that a request waiting for room in the memory pipeline holds one wave of a SIMD while its partner keeps the matrix pipe busy
    // SPLIT (fp16x3 context): three passes over the K chunks -- pixels [x_lo | x_hi | x_hi] (the low halves a.in*_lo halves behind
    // the high ones) against the packed weight chunks [w_hi | w_lo | w_hi]; scaled / split epilogue
    constexpr int NW = 8, BN = 64 * NB, BM = NW * MI * 16, W_BYTES = BN * 128, DIST = RING - 1;
    constexpr int WPW = BN / 8 / NDW, APW = 2 * MI, PER = WPW + APW;     // weight pieces per requesting wave and chunk; pixel loads; both
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool stamps = CY_STAMPS_ENABLED && (a.dbg & 64) != 0;      // diagnostic builds: per-workgroup phase records (see the wide kernel)
    const unsigned long long t_entry = stamps ? stamp_real() : 0;
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int HoWo = a.Ho * a.Wo, M = a.B * HoWo;
    const int cpad = pad128(a.Cout);
    const int ntn = (pad64(a.Cout) + BN - 1) / BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;
    // K chunk c = slab * taps + tap (the order of the packed weights).  k = 3 (strided 3x3 layers): every tap is the same
    // per-lane centre-pixel address plus a UNIFORM shift, which rides in soffset; soffset is unsigned, so the resource base
    // is moved back by the largest negative shift (one row + one pixel) and the shift is biased by the same amount.  Lanes
    // whose tap falls outside the image get the OOB sentinel in voffset (the only part that is range-checked) -> zeros.
    // K3 = false is the plain 1x1 kernel: none of the tap arithmetic is compiled in.
    constexpr int taps = K3 ? 9 : 1, pad = K3 ? 1 : 0, kk3 = K3 ? 3 : 1;
    const int pchunks = (a.Cin / 64) * taps, c0chunks = a.c1 ? a.c0 / 64 : pchunks;     // chunks of one pass; of its first segment
    const int chunks = SPLIT ? a.split * pchunks : pchunks;
    const unsigned bias_bytes = K3 ? (unsigned)((a.Wi + 1) * a.in0_ct) * 2u : 0u;

    // (num_records widened by the bias so that the check passes whether or not the hardware adds soffset before it)
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.in0)) - bias_bytes, 0,
                                                       a.in0_bytes + 2 * bias_bytes, 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c1 ? a.in1 : a.in0), 0,
                                                       a.c1 ? a.in1_bytes : a.in0_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt), 0, a.wgt_bytes, 0x00020000);
    unsigned v0[MI], v1[MI], vmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wave * MI + mi) * 16 + fr;
        v0[mi] = v1[mi] = CY_OOB;
        vmask[mi] = 0;
        if (m < M) {
            const int b = m / HoWo, r = m - b * HoWo, ho = r / a.Wo, wo = r - ho * a.Wo;
            int p0;
            if (a.up0) p0 = (b * (a.Hi >> 1) + (ho >> 1)) * (a.Wi >> 1) + (wo >> 1);
            else p0 = (b * a.Hi + ho * a.s) * a.Wi + wo * a.s;                      // centre tap: always inside the image
            v0[mi] = (unsigned)(p0 * a.in0_ct + a.in0_coff + fq * 8) * 2u;
            v1[mi] = (unsigned)(m * a.in1_ct + a.in1_coff + fq * 8) * 2u;
            unsigned vm = 0;
#pragma unroll
            for (int kh = 0; kh < kk3; ++kh)
#pragma unroll
                for (int kw = 0; kw < kk3; ++kw)
                    if ((unsigned)(ho * a.s + kh - pad) < (unsigned)a.Hi && (unsigned)(wo * a.s + kw - pad) < (unsigned)a.Wi)
                        vm |= 1u << (kh * kk3 + kw);
            vmask[mi] = vm;
        }
    }
    // piece j of requesting wave w = rows (j * NDW + w) * 8 .. +7 of the chunk: the lane part (row & 7 = lane >> 3) is the same for
    // every j, the rest is uniform and rides in soffset
    // developer ablation (diagnostic builds; results invalid): bit 0 no weight requests in the loop, bit 1 no weight fragment reads,
    // bit 2 no MFMAs, bit 3 no pixel loads in the loop.  Round 3, 1024 -> 256 channels on 64 x 64 maps at batch 256, us per K chunk in a
    // stamped build: everything 2.57 | no weight requests 2.42 | no fragment reads 2.34 | no MFMAs 1.70 | no pixel loads 1.59 | MFMAs +
    // fragment reads 1.35 | MFMAs alone 1.16 | barriers alone 0.33.  The loop waits for its operands: per chunk a workgroup fetches
    // 32 KB of pixels and 32 KB of weights, and 64 KB per ~3900 cycles = 16-17 B per cycle and CU is the rate every fetch-heavy kernel
    // of this library tops out at (the wide kernel's requests alone: 15.7 B per cycle).  More requests in flight do not help -- touching
    // the pixel lines 1 / 2 / 4 chunks ahead of the register ring (4-byte LDS-DMA into a dump area) made every 1x1 layer 9-21 % SLOWER --
    // only fewer bytes per MFMA would, and the 256 px x 256 ch tile is the largest the 128 accumulator registers of 8 waves allow.
    // The lane -> address map of the pixel loads costs a little: the MFMA operand layout puts four DIFFERENT pixels (four 128-byte
    // lines, 16 bytes of each) into every quad of lanes.  A timing-only build whose quads read 64 contiguous bytes of ONE pixel (same
    // bytes per instruction, wrong lanes) ran the 1x1 layers 1.5-9.6 % and the strided 3x3 layers 4.6-7.5 % faster (0.39 ms of the
    // 27.4 ms forward pass at batch 256); getting the data back into operand order is a 4 x 4 lane transpose per register (four
    // ds_bpermute_b32 per load, or ~20 DPP / permlane-swap moves).  Built with ds_bpermute (the next chunk's registers transposed behind
    // the current chunk's MFMAs; bit-identical, 234 conv / forward tests green): every layer 2-14 % SLOWER (forward 28.34 vs 27.80 ms):
    // the 16 permutes per wave and chunk cost two to three times what the better access pattern gives.  Not kept.
    // Round 4: the eight requests of a chunk (four weight pieces, four pixel loads per wave) issued one behind each of the chunk's eight
    // MFMA groups instead of together at its start, where all eight waves stand behind the barrier (same order, same counted waits;
    // bit-identical, 296 conv / forward tests green): forward 27.31-27.48 vs 27.37-27.38 ms, the K-heavy layers 0-2 % SLOWER.  Not kept.
    // Also round 4: a persistent-kernel grid of ncu - 8 / 16 / 32 workgroups (CUs left to the side streams) changes nothing in the
    // config-5 pass (4864-4893 vs 4890-4905 tiles/s).
    const bool no_dma = CY_STAMPS_ENABLED && (a.dbg & 1), no_rd = CY_STAMPS_ENABLED && (a.dbg & 2), no_mma = CY_STAMPS_ENABLED && (a.dbg & 4),
               no_px = CY_STAMPS_ENABLED && (a.dbg & 8);
    const bool reqw = wave < NDW;
    const unsigned woff0 = (unsigned)((n0 + wave * 8 + (lane >> 3)) * 128 + ((lane & 7) ^ (lane >> 3)) * 16);
    u32x4 xa[RING][MI][2];
    int ltap = 0, lslab = 0;                                // cursor of load_a (called for c = 0, 1, 2, ... in order)
    auto load_a = [&](int slot, int cv) {                   // the uniform part of the address rides in soffset (not range-checked)
        if (no_px) return;
        int lo = 0;
        const int c = SPLIT ? x3_chunk(cv, pchunks, lo) : cv;           // physical chunk (the weights are packed per virtual chunk)
        const unsigned lo0 = SPLIT && lo ? (unsigned)a.in0_lo * 2u : 0u, lo1 = SPLIT && lo ? (unsigned)a.in1_lo * 2u : 0u;
        if (c >= c0chunks) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    xa[slot][mi][kk] = load_b128(rs1, v1[mi], ((c - c0chunks) * 64 + kk * 32) * 2 + lo1);
        } else if constexpr (!K3) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    xa[slot][mi][kk] = load_b128(rs0, v0[mi], (c * 64 + kk * 32) * 2 + lo0);
        } else {
            const int kh = ltap / 3, kw = ltap - kh * 3;
            const unsigned so = bias_bytes + (unsigned)((((kh - 1) * a.Wi + (kw - 1)) * a.in0_ct + lslab * 64) * 2) + lo0;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const unsigned vo = ((vmask[mi] >> ltap) & 1u) ? v0[mi] : CY_OOB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) xa[slot][mi][kk] = load_b128(rs0, vo, so + kk * 64);
            }
            if (++ltap == 9) { ltap = 0; ++lslab; if (SPLIT && lslab * 9 == pchunks) lslab = 0; }
        }
    };
    auto dma_w = [&](int slot, int c) {
        if ((NDW < NW && !reqw) || no_dma) return;
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            dma_piece(rsw, (lds_ptr_t*)(smem + slot * W_BYTES + (j * NDW + wave) * 1024), woff0, c * cpad * 128 + j * NDW * 1024);
    };
    // counted wait of a requesting wave: all but its n youngest requests have landed (the other waves have no LDS-DMA of their own;
    // the loads of their pixel registers are waited for by the compiler at the use)
    auto wait_w = [&](int n) {
        if (NDW < NW && !reqw) return;
        if (n == APW + 2 * PER) { CY_WAIT_VM(APW + 2 * PER); }
        else if (n == APW + PER) { CY_WAIT_VM(APW + PER); }
        else { CY_WAIT_VM(APW); }
    };
    f32x4 acc[NB * 4][MI];
#pragma unroll
    for (int ni = 0; ni < NB * 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute_x = [&](int slot, const u32x4 (&xs)[MI][2]) {
        const char* Wb = smem + slot * W_BYTES;
        // 2*NB groups of (4 weight fragments, 4*MI MFMAs); the fragments of group i+1 are read before the MFMAs of group i
        f16x8 wb[2][4];
        auto load_wb = [&](f16x8* dst, int gi) {
            if (no_rd) return;
            const int kk = gi / NB, g = gi % NB, qf = fq + 4 * kk;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int r = g * 64 + ni * 16 + fr;
                dst[ni] = *reinterpret_cast<const f16x8*>(Wb + r * 128 + ((qf ^ (r & 7)) << 4));
            }
        };
        load_wb(wb[0], 0);
#pragma unroll
        for (int gi = 0; gi < 2 * NB; ++gi) {
            const int kk = gi / NB, g = gi % NB;
            if (gi + 1 < 2 * NB) load_wb(wb[(gi + 1) & 1], gi + 1);
            if (!no_mma)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[g * 4 + ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        wb[gi & 1][ni], __builtin_bit_cast(f16x8, xs[mi][kk]), acc[g * 4 + ni][mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto compute = [&](int slot) { compute_x(slot, xa[slot]); };

    // the tile's BN bias values -> LDS with the very first request (wave 0; oldest request: every counted wait covers it), so
    // that the epilogue does not start with an exposed L2 round trip
    float* const bias_lds = reinterpret_cast<float*>(smem + (FUSE2 ? 4 : RING) * W_BYTES);
    if (wave == 0) {
        const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, (unsigned)pad64(a.Cout) * 4u, 0x00020000);
        dma_piece(rsb, (lds_ptr_t*)bias_lds, lane * 16 < BN * 4 ? (unsigned)(n0 * 4 + lane * 16) : CY_OOB, 0);
    }
    // prologue: chunks 0..DIST-1 requested (weights first, then pixels, per chunk: the order the counted waits assume)
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < chunks) { dma_w(d, d); load_a(d, d); }
    // weights of chunk 0 landed; what was requested after them may stay in flight (DIST is 2 or 3)
    if (chunks >= DIST) wait_w(APW + (DIST - 1) * PER);
    else if (DIST == 3 && chunks == 2) wait_w(APW + PER);
    else wait_w(APW);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t_loop = stamps ? stamp_real() : 0, c_loop = stamps ? stamp_now() : 0;
#pragma unroll 1
    for (int cb = 0; cb < chunks; cb += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int c = cb + u;
            if (c < chunks) {
                if (c + DIST < chunks) { dma_w((u + DIST) % RING, c + DIST); load_a((u + DIST) % RING, c + DIST); }
                compute(u);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // weights of chunk c+1 must have landed; everything requested after them may stay in flight
                if (c + DIST < chunks) wait_w(APW + (DIST - 1) * PER);
                else if (DIST == 3 && c + 2 < chunks) wait_w(APW + PER);
                else wait_w(APW);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    const unsigned long long t_epi = stamps ? stamp_real() : 0, c_epi = stamps ? stamp_now() : 0;
    if constexpr (FUSE2) {
        // every wave is past the loop's last barrier: the ring is free.  Second layer: chunks 0..3 -> slots 0..3 (the launch gives
        // this form a fourth slot), bias2 behind the first bias
        const auto rsw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt2), 0, a.wgt2_bytes, 0x00020000);
        const int cpad2 = pad128(a.cout2);
        if ((NDW == NW || reqw)) {
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
                for (int j = 0; j < WPW; ++j)
                    dma_piece(rsw2, (lds_ptr_t*)(smem + c2 * W_BYTES + (j * NDW + wave) * 1024), woff0, c2 * cpad2 * 128 + j * NDW * 1024);
        }
        float* const bias2_lds = reinterpret_cast<float*>(smem + 4 * W_BYTES) + 256;
        if (wave == 0) {
            const auto rsb2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias2), 0, (unsigned)pad64(a.cout2) * 4u, 0x00020000);
            dma_piece(rsb2, (lds_ptr_t*)bias2_lds, lane * 16 < BN * 4 ? (unsigned)(lane * 16) : CY_OOB, 0);
        }
        // first epilogue in registers: bias + SiLU, rounded to fp16 like the store it replaces -> B fragments of the second GEMM
        float* const bias1_lds = reinterpret_cast<float*>(smem + 4 * W_BYTES);
        u32x4 x2[4][MI][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float bv[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(bias1_lds + g * 64 + fq * 16 + j * 4);
                bv[j * 4] = t[0]; bv[j * 4 + 1] = t[1]; bv[j * 4 + 2] = t[2]; bv[j * 4 + 3] = t[3];
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float v[16];
                bias_act16(acc[g * 4][mi], acc[g * 4 + 1][mi], acc[g * 4 + 2][mi], acc[g * 4 + 3][mi], bv, a.act != 0, v);
                f16x8 h0, h1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { h0[j] = (f16)v[j]; h1[j] = (f16)v[8 + j]; }
                x2[g][mi][0] = __builtin_bit_cast(u32x4, h0); x2[g][mi][1] = __builtin_bit_cast(u32x4, h1);
            }
        }
#pragma unroll
        for (int ni = 0; ni < NB * 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        CY_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) compute_x(c2, x2[c2]);
        // second epilogue: bias2 + activation, 16 contiguous channels per lane and 64-channel block
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const int cbase = g * 64 + fq * 16;
            if (cbase >= pad64(a.cout2)) continue;
            float bv[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(bias2_lds + g * 64 + fq * 16 + j * 4);
                bv[j * 4] = t[0]; bv[j * 4 + 1] = t[1]; bv[j * 4 + 2] = t[2]; bv[j * 4 + 3] = t[3];
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m0 + (wave * MI + mi) * 16 + fr;
                if (m >= M) continue;
                const int b = m / HoWo, r = m - b * HoWo;
                const long opix = (long)b * a.out_bs + a.out_ro + r;
                float v[16];
                bias_act16(acc[g * 4][mi], acc[g * 4 + 1][mi], acc[g * 4 + 2][mi], acc[g * 4 + 3][mi], bv, a.act2 != 0, v);
                f16* dst = reinterpret_cast<f16*>(a.out2) + opix * a.out2_ct + a.out2_coff + cbase;
                if (cbase + 16 <= a.cout2) {
                    f16x8 o0, o1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
                    *reinterpret_cast<f16x8*>(dst) = o0;
                    *reinterpret_cast<f16x8*>(dst + 8) = o1;
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (cbase + j < a.cout2) dst[j] = (f16)v[j];
                }
            }
        }
        return;
    }
    // ---- epilogue: per 64-channel block the same 16-contiguous-channels-per-lane layout as the other kernels
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int cbase = n0 + g * 64 + fq * 16;
        if (cbase >= pad64(a.Cout)) continue;
        float bv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(bias_lds + g * 64 + fq * 16 + j * 4);
            bv[j * 4] = t[0]; bv[j * 4 + 1] = t[1]; bv[j * 4 + 2] = t[2]; bv[j * 4 + 3] = t[3];
        }
        float sc[SPLIT ? 16 : 1];
        if constexpr (SPLIT) {
#pragma unroll
            for (int j = 0; j < 16; ++j) sc[j] = a.oscale[cbase + j];
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + (wave * MI + mi) * 16 + fr;
            if (m >= M) continue;
            const int b = m / HoWo, r = m - b * HoWo;
            const long opix = (long)b * a.out_bs + a.out_ro + r;
            float v[16];
            if constexpr (SPLIT) {
                scale_bias_act16(acc[g * 4][mi], acc[g * 4 + 1][mi], acc[g * 4 + 2][mi], acc[g * 4 + 3][mi], bv, sc, a.act != 0, v);
                f16* dst = reinterpret_cast<f16*>(a.out) + opix * a.out_ct + a.out_coff + cbase;
                const f16* rp = a.res ? reinterpret_cast<const f16*>(a.res) + (long)m * a.res_ct + a.res_coff + cbase : nullptr;
                if (cbase + 16 <= a.Cout) {
                    if (rp) {
                        const f16x8 r0v = *reinterpret_cast<const f16x8*>(rp), r1v = *reinterpret_cast<const f16x8*>(rp + 8);
                        const f16x8 q0v = *reinterpret_cast<const f16x8*>(rp + a.res_lo), q1v = *reinterpret_cast<const f16x8*>(rp + a.res_lo + 8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) { v[j] += (float)r0v[j] + (float)q0v[j]; v[8 + j] += (float)r1v[j] + (float)q1v[j]; }
                    }
                    store_split16(dst, a.out_lo, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (cbase + j >= a.Cout) continue;
                        float t = v[j];
                        if (rp) t += (float)rp[j] + (float)rp[a.res_lo + j];
                        store_split1(dst + j, a.out_lo, t);
                    }
                }
                continue;
            }
            bias_act16(acc[g * 4][mi], acc[g * 4 + 1][mi], acc[g * 4 + 2][mi], acc[g * 4 + 3][mi], bv, a.act != 0, v);
            if (cbase + 16 <= a.Cout) {
                f16* dst = reinterpret_cast<f16*>(a.out) + opix * a.out_ct + a.out_coff + cbase;
                if (a.res) {
                    const f16* rp = reinterpret_cast<const f16*>(a.res) + (long)m * a.res_ct + a.res_coff + cbase;
                    const f16x8 r0v = *reinterpret_cast<const f16x8*>(rp), r1v = *reinterpret_cast<const f16x8*>(rp + 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] += (float)r0v[j]; v[8 + j] += (float)r1v[j]; }
                }
                f16x8 o0, o1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { o0[j] = (f16)v[j]; o1[j] = (f16)v[8 + j]; }
                *reinterpret_cast<f16x8*>(dst) = o0;
                *reinterpret_cast<f16x8*>(dst + 8) = o1;
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = cbase + j;
                    if (c >= a.Cout) continue;
                    float t = v[j];
                    if (a.res) t += (float)(reinterpret_cast<const f16*>(a.res)[(long)m * a.res_ct + a.res_coff + c]);
                    reinterpret_cast<f16*>(a.out)[opix * a.out_ct + a.out_coff + c] = (f16)t;
                }
            }
        }
    }
    if (stamps && tid == 0) {
        unsigned long long* rec = g_wg_stamps + (size_t)(blockIdx.x % WG_STAMP_SLOTS) * 4;
        rec[0] = t_loop - t_entry; rec[1] = t_epi - t_loop; rec[2] = stamp_real() - t_epi; rec[3] = c_epi - c_loop;
    }
}

template <int NB, int MI, int RING, bool K3, bool SPLIT = false, int NDW = 8, bool FUSE2 = false>
static hipError_t launch_direct(const ConvArgs& a, hipStream_t s) {
    constexpr int BN = 64 * NB, BM = 8 * MI * 16;
    const size_t lds = (FUSE2 ? 4 : RING) * BN * 128 + 2048;             // weight ring (four slots for the second layer of a fused pair), bias (x2)
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_direct_kernel<NB, MI, RING, K3, SPLIT, NDW, FUSE2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int M = a.B * a.Ho * a.Wo;
    const int blocks = ((M + BM - 1) / BM) * ((pad64(a.Cout) + BN - 1) / BN);
    hipLaunchKernelGGL((conv1x1_direct_kernel<NB, MI, RING, K3, SPLIT, NDW, FUSE2>), dim3(blocks), dim3(512), lds, s, a);
    return hipGetLastError();
}

template <int WM, int RING, int PB = 2>
static hipError_t launch_halo(const ConvArgs& a, hipStream_t s) {
    constexpr int TH = 4 * WM, NT = WM * 128;
    constexpr int PR = (TH + 2) * 18, NWI = (PR + 7) / 8, NW = NT / 64, PROUNDS = (NWI + NW - 1) / NW;
    const size_t lds = PB * PROUNDS * NW * 1024 + RING * 128 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<WM, RING, PB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int blocks = a.B * ((a.Hi + TH - 1) / TH) * ((a.Wi + 15) / 16) * ((pad64(a.Cout) + 127) / 128);
    hipLaunchKernelGGL((conv3x3_halo_kernel<WM, RING, PB>), dim3(blocks), dim3(NT), lds, s, a);
    return hipGetLastError();
}

template <typename T, int WM, int WN, int MI, int NSTAGE = 2, bool SPLIT = false>
static hipError_t launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MI * 16, BN = WN * 64;
    const int M = a.B * a.Ho * a.Wo;
    const int ntm = (M + BM - 1) / BM, ntn = (pad64(a.Cout) + BN - 1) / BN;
    const size_t lds = NSTAGE * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, WM, WN, MI, NSTAGE, SPLIT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, MI, NSTAGE, SPLIT>), dim3(ntm * ntn), dim3(WM * WN * 64), lds, s, a);
    return hipGetLastError();
}

// head1x1_kernel: which layers it takes, and its launch -- the detect head's output 1x1s (fp32 prediction rows, no activation) and, since
// the end of round 4, every other single-segment 1x1 with <= 64 output channels and Cin a multiple of 64 (fp16 / split outputs into a
// channel slice, SiLU, residual).  Geometry only (no batch threshold), so a tile's result does not depend on its batch.
static int head_rows(const ConvArgs& a) { const int c = a.Cout < 16 ? a.Cout : 16; const int g = (c + 3) / 4; return 16 * (g < 4 ? g : 4); }
static bool head_direct(const ConvArgs& a, int passes) {
    if (!env_knob("CY_HEAD_DIRECT", 1)) return false;             // read per call: the parity tests run both forms
    if (!(a.k == 1 && a.s == 1 && a.c1 == 0 && !a.up0 && a.Cin >= 64 && a.Cin % 64 == 0 && a.c0 == a.Cin && pad64(a.Cout) == 64 && a.Ho == a.Hi && a.Wo == a.Wi))
        return false;
    if (a.out_f32 ? (a.res || a.act) : (a.out_bs != a.Ho * a.Wo || a.out_ro != 0 || !env_knob("CY_NARROW_DIRECT", 1))) return false;      // head rows: no residual / activation; inside the network: plain pixel order
    return (size_t)passes * (a.Cin / 64) * head_rows(a) * 128 <= 65536;
}
template <bool SPLIT>
static hipError_t launch_head(const ConvArgs& a, hipStream_t s) {
    const int nrows = head_rows(a);
    const size_t lds = (size_t)(SPLIT ? a.split : 1) * (a.Cin / 64) * nrows * 128 + (a.Cout == 64 && a.out_f32 ? 4 * 32 * 68 * 4 : 0);     // weights + store transpose
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(head1x1_kernel<SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4 * 32 * 68 * 4);
        attr_set = true;
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long wgs = (M + 127) / 128;                             // one 32-pixel group per wave at least
    hipLaunchKernelGGL((head1x1_kernel<SPLIT>), dim3((unsigned)(wgs < 512 ? wgs : 512)), dim3(256), lds, s, a, nrows);      // persistent: two 4-wave workgroups per CU
    return hipGetLastError();
}

// the two output convolutions of a stride level as one launch (head1x1_pair_kernel): a = box branch (64 channels at column 0),
// b = class branch (nc channels at column 64) of the same pixels and prediction rows
static int head_pair_pitch(int rowlen) { int tp = rowlen; while (tp % 8 != 4) ++tp; return tp; }      // conflict-free b128 rows
static size_t head_pair_lds(Precision p, const ConvArgs& a, const ConvArgs& b) {
    const size_t pa = p == PREC_F16X3 ? a.split : 1, pb = p == PREC_F16X3 ? b.split : 1;
    return pa * (a.Cin / 64) * 64 * 128 + pb * (b.Cin / 64) * head_rows(b) * 128 + (size_t)8 * 16 * head_pair_pitch(a.out_ct) * 4;
}
bool head_pair_ok(Precision p, const ConvArgs& a, const ConvArgs& b) {
    if (p == PREC_F32 || !env_knob("CY_HEAD_PAIR", 1)) return false;       // read per call: the parity tests run both forms
    if (conv_variant(p, a) != CONV_HEAD_1X1 || conv_variant(p, b) != CONV_HEAD_1X1) return false;
    if (a.B != b.B || a.Ho != b.Ho || a.Wo != b.Wo || a.out != b.out || a.out_ct != b.out_ct || a.out_bs != b.out_bs || a.out_ro != b.out_ro) return false;
    if (a.out_coff != 0 || a.Cout != 64 || b.out_coff != 64 || b.Cout < 1 || b.Cout != a.out_ct - 64) return false;
    return head_pair_lds(p, a, b) <= 128 * 1024;
}
hipError_t launch_head_pair(Precision p, const ConvArgs& a, const ConvArgs& b, hipStream_t s) {
    const size_t lds = head_pair_lds(p, a, b);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(head1x1_pair_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(head1x1_pair_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set = true;
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long wgs = (M + 255) / 256;                             // one 32-pixel group per wave at least
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);           // persistent: one 8-wave workgroup per CU is what the register file holds
    if (p == PREC_F16X3) hipLaunchKernelGGL((head1x1_pair_kernel<true>), dim3(grid), dim3(512), lds, s, a, b, head_rows(b), head_pair_pitch(a.out_ct));
    else hipLaunchKernelGGL((head1x1_pair_kernel<false>), dim3(grid), dim3(512), lds, s, a, b, head_rows(b), head_pair_pitch(a.out_ct));
    return hipGetLastError();
}

// Kernel variants (also the keys of the profiling summary)
static const char* const kVariantNames[CONV_NUM_VARIANTS] = {
    "conv_igemm_kernel<2,2,4> generic 128x128", "conv_igemm_kernel<4,1,2> generic 128x64",
    "conv3x3_halo2_kernel 3x3 s1 16x16px x128ch (halo<2,2> for odd slab counts)", "conv3x3_pp_kernel<2> 3x3 s1 16x16px x64ch",
    "conv3x3_pp_kernel<4> 3x3 s1 16x16px x128ch", "conv3x3_halo_kernel<4,*> 3x3 s1 16x16px x128ch",
    "conv3x3_c64_kernel<64|32> 3x3 s1 persistent, Cin 64 or 32", "conv_igemm_kernel<4,2,4,3> generic 256x128, 3-slab ring",
    "conv3x3_wide_kernel 3x3 s1 16x32px x128ch, K slabs of 32",
    "conv1x1_direct_kernel<4,2> 1x1 256px x256ch, pixels to regs", "conv1x1_direct_kernel<2,2> 1x1 256px x128ch, pixels to regs",
    "conv3x3_wide_kernel<WN=1> 3x3 s1 16x32px x64ch", "conv3x3_wide_kernel<dual> 3x3 s1 2 images x 16x16px x128ch",
    "conv3x3_widep_kernel<strip> 3x3 s1 512 flattened px x128ch, persistent",
    "head1x1_kernel detect-head output 1x1, weights stationary in LDS, fp32 rows"};
const char* conv_variant_name(int v) { return v >= 0 && v < CONV_NUM_VARIANTS ? kVariantNames[v] : "?"; }

static bool s2_direct() { static const int v = getenv("CY_S2_DIRECT") ? atoi(getenv("CY_S2_DIRECT")) : 1; return v != 0; }

// Batch-invariant kernel selection (default; CY_BATCH_INVARIANT=0 turns it off): every batch-size threshold below (and the
// stem fusion) is evaluated as if the batch held >= 256 tiles, so a layer always runs the SAME kernel and a tile's fp16
// results do not depend on how many tiles share its batch: catalogs are identical for any world size / batch split, as the
// reference's are for any MPI size.  Price: big-batch kernels on small batches (1 % of the 16k-mosaic rate; a single
// image runs kernels tuned for 256).  With 0 the thresholds see the real batch: fastest per batch size, last-bit differences
// between batch sizes.
bool batch_invariant() { const char* e = getenv("CY_BATCH_INVARIANT"); return !e || atoi(e) != 0; }

// fp16x3 context: the kernels that carry the three-pass K walk -- the 512-px wide 3x3 kernel (128- / 64-channel / dual-image forms),
// the pixels-direct 1x1 / strided-3x3 kernel and the generic implicit GEMM for everything else.  The choice depends on the layer's
// geometry only, so a tile's result does not depend on the batch it travels in -- with one bound: the strip form addresses the
// flattened batch with 22 bits (strip_fits: B * (H + 1) * (W + 1) < 2^22), which a launch can only exceed beyond the 4 GiB a tensor
// may hold in this context (128 tiles of 640^2: 2^19.6 entries on the largest strip-form map), where cy_forward refuses the batch.
static int conv_variant_x3(const ConvArgs& a) {
    const bool narrow = pad64(a.Cout) <= 64;
    if (a.k == 3 && a.s == 1 && a.c1 == 0 && !a.up0 && !a.out_f32 && a.Cin % 64 == 0 && a.wgt32 && a.out_bs == a.Ho * a.Wo && a.out_ro == 0) {
        const int wpad = (a.Wi + 31) / 32 * 32;
        const bool fits = (wpad - a.Wi) * 8 <= a.Wi;
        if (narrow && a.Wi % 32 == 0) return CONV_WIDE_64;
        if (narrow && a.Cin >= 128 && env_knob("CY_STRIP", 1) && strip_small(a) && strip_fits(a)) return CONV_STRIP_128;       // (see conv_variant)
        // strip form where 16 x 32 patches would idle > 4 % more lanes (geometry only: no batch threshold in this context)
        if (!narrow && env_knob("CY_STRIP", 1) && strip_fits(a) && strip_cover(a.Hi, a.Wi) + 0.04 < wide2d_cover(a.Hi, a.Wi)) return CONV_STRIP_128;
        if (!narrow && fits) return CONV_WIDE_128;
        if (!narrow && a.Wi <= 16 && a.Wi >= 14) return CONV_WIDE_DUAL;
    }
    if (!a.out_f32 && a.Cin % 64 == 0 && !narrow &&
        ((a.k == 1 && a.s == 1 && (a.c1 == 0 || a.c0 % 64 == 0)) || (a.k == 3 && a.s == 2 && a.c1 == 0 && !a.up0)))
        return pad64(a.Cout) >= 256 ? CONV_DIRECT_256 : CONV_DIRECT_128;
    if (head_direct(a, a.split)) return CONV_HEAD_1X1;
    return narrow ? CONV_GENERIC_64 : CONV_GENERIC_128;
}

int conv_variant(Precision p, const ConvArgs& a) {
    if (p == PREC_F16X3) return conv_variant_x3(a);
    const bool narrow = pad64(a.Cout) <= 64;
    const long Bv = batch_invariant() && a.B < 256 ? 256 : a.B;     // the batch size the thresholds see
    // 3x3 stride-1 layers (fp16 context; the fp32 parity context keeps the generic kernel): halo-reuse kernels.
    //   Cout <= 64  : ping-pong kernel with 64-channel tiles (256 px x 64 ch per workgroup)
    //   Cout >= 128 : 8x16-pixel patches x 128 channels, two workgroups per CU
    if (p == PREC_F16 && a.k == 3 && a.s == 1 && a.c1 == 0 && !a.up0 && !a.out_f32 && a.Cin == 32 && narrow && a.wgt32 &&
        a.out_bs == a.Ho * a.Wo && a.out_ro == 0)
        return CONV_C64_PERSIST;                             // 32-channel variant of the persistent kernel
    if (p == PREC_F16 && a.k == 3 && a.s == 1 && a.c1 == 0 && !a.up0 && !a.out_f32 && a.Cin % 64 == 0 &&
        a.out_bs == a.Ho * a.Wo && a.out_ro == 0) {
        static const int force = getenv("CY_HALO_WM") ? atoi(getenv("CY_HALO_WM")) : 0;     // tuning override
        static const int wide64 = getenv("CY_WIDE64") ? atoi(getenv("CY_WIDE64")) : 1;
        // 64-channel variant of the wide kernel for the box-head convs with deep inputs, when it fills the chip
        if (narrow && wide64 && a.wgt32 && a.Cin >= 128 && a.Wi % 32 == 0 && force == 0 &&
            Bv * ((a.Hi + 15) / 16) * (a.Wi / 32) >= 256)
            return CONV_WIDE_64;
        // 64 output channels on maps of a few pixels with deep inputs (the box head of 128- / 256-px inputs): the strip form with half
        // of its channel tile empty still beats the 16 x 16-pixel patches of the 64-channel kernels there (68 -> ~250 TFLOP/s on 4 x 4 maps)
        if (narrow && force == 0 && a.wgt32 && a.Cin >= 128 && env_knob("CY_STRIP", 1) && strip_small(a) && strip_fits(a)) return CONV_STRIP_128;
        if (narrow && a.Cin == 64 && force != 8) return force == 9 ? CONV_GENERIC_64 : CONV_C64_PERSIST;
        if (narrow) return force == 9 ? CONV_GENERIC_64 : CONV_PP_64;
        if (force == 5) return CONV_PP_128;
        if (force == 4 || force == 43) return CONV_HALO16_128;
        static const int wide = getenv("CY_WIDE") ? atoi(getenv("CY_WIDE")) : 1;
        // strip form of the wide kernel (maps flattened to one dimension) where 16 x 32-pixel patches would leave > 4 % more of the MFMA
        // lanes idle: 80 / 40 / 20-px maps of 640-px inputs, 52 x 64 / 26 x 32 maps of the ragged 416 x 512 tiles.  CY_STRIP: 0 off,
        // 2 regardless of the launch size (tests); read per call.
        const int strip = env_knob("CY_STRIP", 1);
        if (wide && strip && a.wgt32 && strip_fits(a) && strip_cover(a.Hi, a.Wi) + 0.04 < wide2d_cover(a.Hi, a.Wi) &&
            (strip > 1 || strip_small(a) || Bv * (long)(a.Hi + 1) * (a.Wi + 1) / 512 * ((pad64(a.Cout) + 127) / 128) >= 200))      // (about one strip per CU; tiny maps: every alternative idles more)
            return CONV_STRIP_128;
        const int wpad = (a.Wi + 31) / 32 * 32;
        if (wide && a.wgt32 && (wpad - a.Wi) * 8 <= a.Wi) return CONV_WIDE_128;     // <= 12.5 % of the patch columns idle
        const int dual = getenv("CY_WIDE_DUAL") ? atoi(getenv("CY_WIDE_DUAL")) : 1;      // read per call: 2 forces it (tests)
        // two 16-px-wide images side by side: +8 % over the two-tap halo kernel once it still fills the chip (half as many
        // workgroups), slower below that
        if (wide && dual && a.wgt32 && a.Wi <= 16 && a.Wi >= 14 &&
            ((Bv + 1) / 2) * ((a.Hi + 15) / 16) * ((pad64(a.Cout) + 127) / 128) >= (dual > 1 ? 1 : 224))
            return CONV_WIDE_DUAL;
        return CONV_HALO8_128;
    }
    if (p == PREC_F16 && head_direct(a, 1)) return CONV_HEAD_1X1;
    if (narrow) return CONV_GENERIC_64;
    // 1x1: pixels-direct kernel once there is at least one 256-pixel tile per CU (below that the 128x128 tiles of the generic
    // kernel fill the chip better).  CY_DIRECT_MIN_BLOCKS is read per call so that the parity tests can force the path.
    if (p == PREC_F16 && !a.out_f32 && a.Cin % 64 == 0 &&
        ((a.k == 1 && a.s == 1 && (a.c1 == 0 || a.c0 % 64 == 0)) || (a.k == 3 && a.s == 2 && a.c1 == 0 && !a.up0 && (a.Cin >= 128 || pad64(a.Cout) < 256) && s2_direct()))) {
        const char* e = getenv("CY_DIRECT_MIN_BLOCKS");
        const long min_blocks = e ? atol(e) : 256;
        static const int bn_force = getenv("CY_DIRECT_BN") ? atoi(getenv("CY_DIRECT_BN")) : 0;      // tuning knob
        const int bn = bn_force == 128 && a.k == 1 ? 128 : (pad64(a.Cout) >= 256 ? 256 : 128);
        const long blocks = ((Bv * a.Ho * a.Wo + 255) / 256) * ((pad64(a.Cout) + bn - 1) / bn);
        if (min_blocks >= 0 && blocks >= min_blocks) return bn == 256 ? CONV_DIRECT_256 : CONV_DIRECT_128;
    }
    // 1x1 and strided convs with enough 256-pixel tiles to fill the chip: deeper-pipelined 256x128 tile
    static const int big = getenv("CY_BIG") ? atoi(getenv("CY_BIG")) : 1;
    const long M = Bv * a.Ho * a.Wo;
    if (p == PREC_F16 && big && (M / 256) * ((pad64(a.Cout) + 127) / 128) >= big * 256) return CONV_GENERIC_BIG;
    return CONV_GENERIC_128;
}

hipError_t launch_conv(Precision p, const ConvArgs& a, hipStream_t s) {
    if (p == PREC_F16X3) {
        if ((a.split != 2 && a.split != 3) || !a.oscale) return hipErrorInvalidValue;
        switch (conv_variant_x3(a)) {
            case CONV_WIDE_128:      // persistent form (bit-identical) with CY_X3_PERSIST=1: measured before it became a default
                if (env_knob("CY_X3_PERSIST", 0) && a.Cout % 16 == 0) return launch_widep(a, s);
                return launch_wide<2, false, 2, true>(a, s);
            case CONV_STRIP_128: return launch_strip(a, s);
            case CONV_WIDE_64: return launch_wide<1, false, 2, true>(a, s);
            case CONV_WIDE_DUAL: return launch_wide<2, true, 2, true>(a, s);
            case CONV_DIRECT_256: return a.k == 3 ? launch_direct<4, 2, 3, true, true>(a, s) : launch_direct<4, 2, 3, false, true>(a, s);
            case CONV_DIRECT_128: return a.k == 3 ? launch_direct<2, 4, 2, true, true>(a, s) : launch_direct<2, 2, 2, false, true>(a, s);
            case CONV_HEAD_1X1: return launch_head<true>(a, s);
            case CONV_GENERIC_64: return launch_t<f16, 4, 1, 2, 2, true>(a, s);
            default: return launch_t<f16, 2, 2, 4, 2, true>(a, s);
        }
    }
    switch (conv_variant(p, a)) {
        case CONV_C64_PERSIST: return a.Cin == 32 ? launch_c64<32>(a, s) : launch_c64<64>(a, s);
        case CONV_PP_64: return launch_pp<2>(a, s);
        case CONV_PP_128: { ConvArgs b2 = a; b2.dbg = dev_knob("CY_DBG", 0); return launch_pp<4>(b2, s); }
        case CONV_HALO16_128: return (getenv("CY_HALO_WM") && atoi(getenv("CY_HALO_WM")) == 43) ? launch_halo<4, 3>(a, s) : launch_halo<4, 2>(a, s);
        case CONV_HALO8_128: {
            ConvArgs b2 = a; b2.dbg = dev_knob("CY_DBG", 0);
            static const int v = getenv("CY_HALO_V") ? atoi(getenv("CY_HALO_V")) : 2;      // 2: two taps per barrier (default)
            if (v == 2 && (a.Cin / 64) % 2 == 0) return launch_halo2(b2, s);
            return v == 1 ? launch_halo<2, 3, 1>(b2, s) : launch_halo<2, 2>(b2, s);
        }
        case CONV_GENERIC_BIG: return launch_t<f16, 4, 2, 4, 3>(a, s);
        case CONV_WIDE_128: {
            static const int tps = getenv("CY_WIDE_TPS") ? atoi(getenv("CY_WIDE_TPS")) : 2;
            ConvArgs b2 = a; b2.dbg = dev_knob("CY_DBG", 0);
            // persistent form (same arithmetic in the same order: bit-identical outputs) when every CU gets at least two patches
            // and the output tile has whole 16-channel groups; CY_WIDE_PERSIST: 0 off, 2 regardless of the launch size (tests)
            // round-4 A/B: the K loop on 32x32x16 MFMAs (32: compiler-placed fragment reads, 33: one read per MFMA gap)
            if (const int mf = env_knob("CY_WIDE_MFMA", 16); mf >= 32) return mf == 33 ? launch_wide32<1>(b2, s) : launch_wide32<0>(b2, s);
            const int wp = env_knob("CY_WIDE_PERSIST", 1);
            const long patches = (long)a.B * ((a.Wi + 31) / 32) * ((a.Hi + 15) / 16) * ((pad64(a.Cout) + 127) / 128);
            // Measured per layer at batch 256: 18-stage layers (Cin 128: model.4 / model.15 bottlenecks) -4..-6 %, 36-stage ones
            // -1.7..+1.5 %: taken for up to two slab pairs.
            if (wp && tps == 2 && a.Cout % 16 == 0 && (wp > 1 || (patches >= 512 && a.Cin <= 128))) return launch_widep(b2, s);
            if (tps == 2 && env_knob("CY_WIDE_SCHED", 0)) return launch_wide<2, false, 2, false, 1>(b2, s);      // round-4 experiment: reads interleaved with the MFMAs
            return tps == 3 ? launch_wide<2, false, 3>(b2, s) : launch_wide<2>(b2, s);
        }
        case CONV_STRIP_128: return launch_strip(a, s);
        case CONV_WIDE_64: return launch_wide<1>(a, s);
        case CONV_WIDE_DUAL: return launch_wide<2, true>(a, s);
        case CONV_DIRECT_256: {
            ConvArgs a2 = a; a2.dbg = dev_knob("CY_DBG", 0);
            // tuning knob, off: 128 px x 256 ch tiles with two workgroups per CU for Cin <= CY_D256_V.  Measured at batch 256: 2-12 %
            // SLOWER on every 256-channel 1x1 layer but model.4.cv1 (-4 %): twice the weight pieces per MFMA cost more than the overlap buys
            static const int v256 = getenv("CY_D256_V") ? atoi(getenv("CY_D256_V")) : 0;
            if (a.k != 3 && v256 > 0 && a.Cin <= v256) return launch_direct<4, 1, 2, false>(a, s);
            // (a 4-slot ring for this tile: 256 VGPRs with spills, 3-10 % slower)
            // weight requests on 4 waves (one per SIMD) or on all 8 (CY_DIRECT_NDW forces one; same bits either way).  Measured per layer
            // at batch 256: four requesting waves are 2-5 % faster on the strided 3x3 layers (model.5, model.7) and 1-6 % slower on the 1x1s
            static const int ndw = env_knob("CY_DIRECT_NDW", 0);
            // back-to-back pair (the runtime sets wgt2 when this layer's only reader is a 256 -> <= 256 1x1 and the tile holds all channels)
            if (a.wgt2 && pad64(a.Cout) == 256 && a.c1 == 0 && !a.res && !a.out_f32)
                return a.k == 3 ? launch_direct<4, 2, 3, true, false, 4, true>(a2, s) : launch_direct<4, 2, 3, false, false, 8, true>(a2, s);
            if (a.k == 3) return ndw == 8 ? launch_direct<4, 2, 3, true>(a2, s) : launch_direct<4, 2, 3, true, false, 4>(a2, s);
            return ndw == 4 ? launch_direct<4, 2, 3, false, false, 4>(a2, s) : launch_direct<4, 2, 3, false>(a2, s);
        }
        case CONV_DIRECT_128:     // strided 3x3 (model.1): 64 px per wave, so every weight fragment feeds four MFMAs (-8 % vs 32 px)
            // 1x1 with a 128-channel tile = the HBM-bound layers (model.2.cv1/cv2): two workgroups per CU (126 VGPRs, one chunk
            // ahead) overlap each other's prologue/epilogue: -12 % against one workgroup with a 4-chunk ring
            if (a.k != 3 && !(getenv("CY_D128_V") && atoi(getenv("CY_D128_V")) == 0)) return launch_direct<2, 2, 2, false>(a, s);
            return a.k == 3 ? launch_direct<2, 4, 2, true>(a, s) : launch_direct<2, 2, 4, false>(a, s);
        case CONV_HEAD_1X1: return launch_head<false>(a, s);
        case CONV_GENERIC_64: return p == PREC_F16 ? launch_t<f16, 4, 1, 2>(a, s) : launch_t<float, 4, 1, 2>(a, s);
        default: return p == PREC_F16 ? launch_t<f16, 2, 2, 4>(a, s) : launch_t<float, 2, 2, 4>(a, s);
    }
}

void debug_read_wg_stamps(unsigned long long* out, int n) {      // raw per-workgroup records (n x 4) of stamped builds
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_stamps), (size_t)(n < WG_STAMP_SLOTS ? n : WG_STAMP_SLOTS) * 4 * sizeof(unsigned long long));
}

void debug_read_stamps(unsigned long long* out8, bool reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long));
    if (CY_STAMPS_ENABLED && out8[6] == 0) {        // no summed stamps: average the wide kernel's per-workgroup records instead
        static unsigned long long rec[WG_STAMP_SLOTS * 4];
        hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_wg_stamps), sizeof(rec));
        for (int i = 0; i < WG_STAMP_SLOTS; ++i)
            if (rec[i * 4 + 1]) { out8[6] += 1; for (int j = 0; j < 4; ++j) out8[j] += rec[i * 4 + j]; }
        out8[4] = out8[0] + out8[1] + out8[2] + out8[3];
        if (reset) { memset(rec, 0, sizeof(rec)); hipMemcpyToSymbol(HIP_SYMBOL(g_wg_stamps), rec, sizeof(rec)); }
    }
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
}

// ------------------------------------------------------------------------------------------------ weights
// Packed layout ("slab-major"): [K chunk of 128 B][tap][row in Cout_pad128][128 B] (zero rows past Cout, so no kernel
// needs a range check on weight rows).  The unit every kernel stages -- the
// rows n0..n0+BN of one (chunk, tap) -- is ONE contiguous run of BN*128 bytes, so a DMA wave-instruction (8 rows) reads
// 1 KiB of consecutive cache lines instead of 8 lines a whole filter row (k*k*Cin elements) apart.  K is zero-padded to
// a whole chunk; rows are permuted per 64 so that a lane of the MFMA result holds 16 contiguous output channels.
size_t packed_weight_bytes(Precision p, int cout, int cin, int k, int chunk_bytes) {
    const int epb = chunk_bytes / (p == PREC_F16 ? 2 : 4);
    return (size_t)((cin + epb - 1) / epb) * k * k * pad128(cout) * chunk_bytes;
}

void pack_weights(Precision p, const float* W, int cout, int cin, int k, void* dst, int chunk_bytes) {
    const int taps = k * k, cp = pad128(cout), epb = chunk_bytes / (p == PREC_F16 ? 2 : 4), chunks = (cin + epb - 1) / epb;
    memset(dst, 0, packed_weight_bytes(p, cout, cin, k, chunk_bytes));
    for (int row = 0; row < cp; ++row) {
        const int blk = row >> 6, ni = (row >> 4) & 3, rr = row & 15;
        const int n = blk * 64 + (rr >> 2) * 16 + ni * 4 + (rr & 3);      // channel held by packed row `row`
        if (n >= cout) continue;
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < cin; ++c) {
                const float v = W[((size_t)n * cin + c) * taps + t];
                const size_t o = (((size_t)(c / epb) * taps + t) * cp + row) * epb + (c % epb);
                if (p == PREC_F16) reinterpret_cast<f16*>(dst)[o] = (f16)v;
                else reinterpret_cast<float*>(dst)[o] = v;
            }
    }
    (void)chunks;
}

// second layer of a back-to-back pair: 1x1 weights with the input channels in the accumulator order of the first layer's kernel
void pack_weights_fused2(const float* W2, int cout2, int cin2, void* dst) {
    std::vector<float> perm((size_t)cout2 * cin2);
    for (int n = 0; n < cout2; ++n)
        for (int p = 0; p < cin2; ++p) {
            const int c = p >> 6, kk = (p >> 5) & 1, q = (p >> 3) & 3, j = p & 7;
            perm[(size_t)n * cin2 + p] = W2[(size_t)n * cin2 + 64 * c + 16 * q + 8 * kk + j];
        }
    pack_weights(PREC_F16, perm.data(), cout2, cin2, 1, dst);
}

// fp16x3 context.  Per output channel n the filter is scaled by 2^e(n) so that its largest weight lies in [2^13, 2^14): the low
// halves w_lo = fp16(w' - fp16(w')) of all but vanishing weights are then normal fp16 numbers (unscaled they would sit in the
// subnormal range and carry ~3e-6 relative error); oscale[n] = 2^-e(n) multiplies the accumulator in the epilogue (exact).
// K holds three passes over the (chunk-padded) input channels: w_hi, w_lo, w_hi -- against x_lo, x_hi, x_hi (see x3_chunk).
// TWO passes (round 4) when every weight of the layer is an fp16 value times its channel's scale, exactly: W[n] = fl32(w16[n] * scale[n])
// with w16 representable in fp16 -- what an ultralytics checkpoint is (its tensors are stored in fp16; Conv + BatchNorm are folded in
// fp32 at load time, so the folded filter of channel n is the fp16 filter times gamma / sqrt(var + eps)).  The layer is then
// scale[n] * sum_k (x_lo + x_hi) * w16: K holds [w16 | w16] against [x_lo | x_hi], oscale[n] = scale[n], and the weights carry no
// rounding at all.  x3_passes() decides from the numbers themselves (scale = null: all ones).
int x3_passes(const float* W, int cout, int cin, int k, const float* scale) {
    const size_t per = (size_t)cin * k * k;
    for (int n = 0; n < cout; ++n) {
        const float sc = scale ? scale[n] : 1.0f;
        if (!(sc != 0.0f) || !std::isfinite(sc)) return 3;
        for (size_t i = 0; i < per; ++i) {
            const float w = W[(size_t)n * per + i];
            const f16 h = (f16)(w / sc);
            if (!((float)h * sc == w)) return 3;
        }
    }
    return 2;
}

size_t packed_weight_bytes_x3(int cout, int cin, int k, int chunk_bytes, int passes) {
    const int epb = chunk_bytes / 2;
    return packed_weight_bytes(PREC_F16, cout, passes * ((cin + epb - 1) / epb * epb), k, chunk_bytes);
}

void pack_weights_x3(const float* W, int cout, int cin, int k, void* dst, float* oscale, int chunk_bytes, int passes, const float* scale) {
    const int taps = k * k, epb = chunk_bytes / 2, cinp = (cin + epb - 1) / epb * epb, cp = pad128(cout);
    memset(dst, 0, packed_weight_bytes_x3(cout, cin, k, chunk_bytes, passes));
    for (int i = 0; i < cp; ++i) oscale[i] = 1.0f;
    f16* o = reinterpret_cast<f16*>(dst);
    for (int row = 0; row < cp; ++row) {
        const int blk = row >> 6, ni = (row >> 4) & 3, rr = row & 15;
        const int n = blk * 64 + (rr >> 2) * 16 + ni * 4 + (rr & 3);      // channel held by packed row `row`
        if (n >= cout) continue;
        if (passes == 2) {                                   // exact fp16 filter, the channel's scale in the epilogue
            const float sc = scale ? scale[n] : 1.0f;
            oscale[n] = sc;
            for (int t = 0; t < taps; ++t)
                for (int c = 0; c < cin; ++c) {
                    const f16 h = (f16)(W[((size_t)n * cin + c) * taps + t] / sc);
                    for (int pass = 0; pass < 2; ++pass) {
                        const int cv = pass * cinp + c;
                        o[(((size_t)(cv / epb) * taps + t) * cp + row) * epb + (cv % epb)] = h;
                    }
                }
            continue;
        }
        float m = 0.0f;
        for (size_t i = 0; i < (size_t)cin * taps; ++i) m = fmaxf(m, fabsf(W[(size_t)n * cin * taps + i]));
        int e = 0;
        if (m > 0.0f && std::isfinite(m)) { int ex; frexpf(m, &ex); e = 14 - ex; }       // m = f * 2^ex, f in [0.5, 1): m * 2^e in [2^13, 2^14)
        if (e > 60) e = 60;
        if (e < -60) e = -60;
        const float up = ldexpf(1.0f, e);
        oscale[n] = ldexpf(1.0f, -e);
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < cin; ++c) {
                const float v = W[((size_t)n * cin + c) * taps + t] * up;
                const f16 hi = (f16)v, lo = (f16)(v - (float)hi);
                for (int pass = 0; pass < 3; ++pass) {
                    const int cv = pass * cinp + c;
                    o[(((size_t)(cv / epb) * taps + t) * cp + row) * epb + (cv % epb)] = pass == 1 ? lo : hi;
                }
            }
    }
}

// fp32 NHWC <-> high / low halves (kernel-level test entry and debug reads of the fp16x3 context)
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ in, f16* __restrict__ out, long n, int C) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long pix = i / C; const int c = (int)(i - pix * C);
        store_split1(out + pix * 2 * C + c, C, in[i]);
    }
}
__global__ __launch_bounds__(256) void x3_merge_kernel(const f16* __restrict__ in, float* __restrict__ out, long n, int C) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long pix = i / C; const int c = (int)(i - pix * C);
        out[i] = (float)in[pix * 2 * C + c] + (float)in[pix * 2 * C + C + c];
    }
}
hipError_t launch_x3_split(const float* in, void* out, long npix, int C, hipStream_t s) {
    const long n = npix * C;
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535)), dim3(256), 0, s, in, reinterpret_cast<f16*>(out), n, C);
    return hipGetLastError();
}
hipError_t launch_x3_merge(const void* in, float* out, long npix, int C, hipStream_t s) {
    const long n = npix * C;
    hipLaunchKernelGGL(x3_merge_kernel, dim3((unsigned)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535)), dim3(256), 0, s, reinterpret_cast<const f16*>(in), out, n, C);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ stem
// Layer 0: Conv(3, C, 3, 2) on the NHWC4 network input.  K = 27 is too small for the matrix cores to matter; the
// layer is bound by its 64-channel output write.  One thread = one output pixel x 16 output channels.
template <typename T, bool SPLIT = false>      // SPLIT (fp16x3 context): T = float input, output as fp16 high / low halves
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
    __shared__ float w[27 * 64];
    __shared__ float bs[64];
    const int co_blocks = a.Cout / 16;
    for (int i = threadIdx.x; i < 27 * a.Cout; i += 256) w[i] = a.w[i];
    for (int i = threadIdx.x; i < a.Cout; i += 256) bs[i] = a.bias[i];
    __syncthreads();
    const long total = (long)a.B * a.Ho * a.Wo * co_blocks;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cb = (int)(idx % co_blocks);
        const long pix = idx / co_blocks;
        const int wo = (int)(pix % a.Wo);
        const int ho = (int)((pix / a.Wo) % a.Ho);
        const int b = (int)(pix / ((long)a.Wo * a.Ho));
        float x[27];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hi = ho * 2 - 1 + kh, wi = wo * 2 - 1 + kw;
                const bool ok = (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
                typedef T vec4 __attribute__((ext_vector_type(4)));
                vec4 pv = {(T)0, (T)0, (T)0, (T)0};
                if (ok) pv = *reinterpret_cast<const vec4*>(reinterpret_cast<const T*>(a.in) +
                                                            (((long)b * a.Hi + hi) * a.Wi + wi) * 4);
#pragma unroll
                for (int c = 0; c < 3; ++c) x[(kh * 3 + kw) * 3 + c] = (float)pv[c];
            }
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(x[t], w[t * a.Cout + cb * 16 + j], acc[j]);
        if constexpr (SPLIT) {
            float v16[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v16[j] = silu_fast(acc[j] + bs[cb * 16 + j]);
            store_split16(reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cb * 16, a.out_lo, v16);
            continue;
        }
        T* dst = reinterpret_cast<T*>(a.out) + pix * a.out_ct + a.out_coff + cb * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v = acc[j] + bs[cb * 16 + j];
            dst[j] = (T)((sizeof(T) == 2) ? silu_fast(v) : silu_exact(v));
        }
    }
}

// The same layer with FOUR horizontally adjacent output pixels per thread (maps whose width is a multiple of 4): the form above reads
// every weight from LDS for a single FMA and is bound by that (16-byte LDS reads for 4 lanes' worth of FMAs: 20 TFLOP/s of fp32 in the
// fp16x3 context, 4.9 % of its forward pass); here a weight read feeds four pixels, and the four pixels' 3 x 9 input columns are loaded
// once (27 loads instead of 36).  Per output value the FMA chain is the one above (taps ascending, fmaf), so results are bit-identical
// (tests/test_gpu_forward.py::test_stem_four_pixel_form_is_bit_identical).  CY_STEM_QUAD=0: the form above.
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void stem_quad_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) float w[27 * 64];
    __shared__ float bs[64];
    const int co_blocks = a.Cout / 16, wq = a.Wo >> 2;
    for (int i = threadIdx.x; i < 27 * a.Cout; i += 256) w[i] = a.w[i];
    for (int i = threadIdx.x; i < a.Cout; i += 256) bs[i] = a.bias[i];
    __syncthreads();
    const long total = (long)a.B * a.Ho * wq * co_blocks;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cb = (int)(idx % co_blocks);
        const long q = idx / co_blocks;
        const int wo0 = (int)(q % wq) * 4;
        const int ho = (int)((q / wq) % a.Ho);
        const int b = (int)(q / ((long)wq * a.Ho));
        float acc[4][16];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[p][j] = 0.0f;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - 1 + kh;
            const bool row_ok = (unsigned)hi < (unsigned)a.Hi;
            const int hic = hi < 0 ? 0 : (hi >= a.Hi ? a.Hi - 1 : hi);
            vec4 r[9];
#pragma unroll
            for (int ci = 0; ci < 9; ++ci) {
                const int wi = wo0 * 2 - 1 + ci;
                // loaded at clamped coordinates and zeroed by a select (a conditional load is a branch with a full wait behind it)
                const int wic = wi < 0 ? 0 : (wi >= a.Wi ? a.Wi - 1 : wi);
                const vec4 t = *reinterpret_cast<const vec4*>(reinterpret_cast<const T*>(a.in) + (((long)b * a.Hi + hic) * a.Wi + wic) * 4);
                r[ci] = (row_ok && (unsigned)wi < (unsigned)a.Wi) ? t : vec4{(T)0, (T)0, (T)0, (T)0};
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    asm volatile("" ::: "memory");           // weights of one tap at a time (hoisted, all 432 of them would live in registers)
                    const float* wr = w + ((kh * 3 + kw) * 3 + c) * a.Cout + cb * 16;
                    float wv[16];
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(wr + 4 * j4);
                        wv[4 * j4] = t4[0]; wv[4 * j4 + 1] = t4[1]; wv[4 * j4 + 2] = t4[2]; wv[4 * j4 + 3] = t4[3];
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float xv = (float)r[2 * p + kw][c];
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc[p][j] = fmaf(xv, wv[j], acc[p][j]);
                    }
                }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long pix = ((long)b * a.Ho + ho) * a.Wo + wo0 + p;
            if constexpr (SPLIT) {
                float v16[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v16[j] = silu_fast(acc[p][j] + bs[cb * 16 + j]);
                store_split16(reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cb * 16, a.out_lo, v16);
            } else {
                T* dst = reinterpret_cast<T*>(a.out) + pix * a.out_ct + a.out_coff + cb * 16;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = acc[p][j] + bs[cb * 16 + j];
                    dst[j] = (T)((sizeof(T) == 2) ? silu_fast(v) : silu_exact(v));
                }
            }
        }
    }
}

// fp16 context: the stem as a K=32 (27 padded) MFMA GEMM.  A wave turns 16 output pixels x 64 channels per step:
// the 64x32 weight panel lives in registers for the whole kernel (A operand), each lane gathers the 8 im2col values of
// its (pixel, k-chunk) from the NHWC4 image with the halo zeroed, and stores 16 contiguous channels of its pixel.
// Bound by the 64-channel output write (8 MB per 512x512 tile), not by arithmetic.
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemArgs a, const f16* __restrict__ wpk) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    f16x8 wb[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wb[ni] = *reinterpret_cast<const f16x8*>(wpk + (ni * 16 + fr) * 32 + fq * 8);
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = a.bias[fq * 16 + j];
    // k = 8*fq + j  ->  tap = k/3 (kh = tap/3, kw = tap%3), channel = k%3; k >= 27 is zero padding
    int dh[8], dw[8], dc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * fq + j, tap = k / 3;
        dh[j] = tap / 3 - 1; dw[j] = tap % 3 - 1; dc[j] = k < 27 ? k % 3 : -1;
    }
    const long ngroups = ((long)a.B * a.Ho * a.Wo + 15) / 16;
    const long npix = (long)a.B * a.Ho * a.Wo;
    const int wave_id = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const f16* in = reinterpret_cast<const f16*>(a.in);
    const bool rows16 = (a.Wo & 15) == 0;                   // a group of 16 pixels never straddles an image row: the
    const int gw = a.Wo >> 4;                               // (b, ho, wo) split is wave-uniform -> scalar divisions
    for (int g = wave_id; g < (int)ngroups; g += nwaves) {
        const long pix = (long)g * 16 + fr;
        const bool pv = pix < npix;
        int wo, ho, b;
        if (rows16) {
            const int row = g / gw;                          // = b*Ho + ho (uniform)
            wo = (g - row * gw) * 16 + fr;
            b = row / a.Ho;
            ho = row - b * a.Ho;
        } else {
            wo = (int)(pix % a.Wo); ho = (int)((pix / a.Wo) % a.Ho); b = (int)(pix / ((long)a.Wo * a.Ho));
        }
        f16x8 xa;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int hi = 2 * ho + dh[j], wi = 2 * wo + dw[j];
            const bool ok = pv && dc[j] >= 0 && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
            xa[j] = ok ? in[(((long)b * a.Hi + hi) * a.Wi + wi) * 4 + dc[j]] : (f16)0.0f;
        }
        f32x4 acc[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa, acc[ni], 0, 0, 0);
        }
        if (pv) {
            f16x8 o0, o1;
            float v16[16];
            bias_act16(acc[0], acc[1], acc[2], acc[3], bv, true, v16);
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (f16)v16[j]; o1[j] = (f16)v16[8 + j]; }
            f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + fq * 16;
            *reinterpret_cast<f16x8*>(dst) = o0;
            *reinterpret_cast<f16x8*>(dst + 8) = o1;
        }
    }
}

hipError_t launch_stem(Precision p, const StemArgs& a, hipStream_t s) {
    if (a.Cout > 64 || a.Cout % 16) return hipErrorInvalidValue;
    if (p == PREC_F16 && a.Cout == 64 && a.wpk) {
        const long ngroups = ((long)a.B * a.Ho * a.Wo + 15) / 16;
        const int grid = (int)((ngroups + 3) / 4 < 4096 ? (ngroups + 3) / 4 : 4096);
        hipLaunchKernelGGL(stem_mfma_kernel, dim3(grid), dim3(256), 0, s, a, reinterpret_cast<const f16*>(a.wpk));
        return hipGetLastError();
    }
    if (p != PREC_F16 && a.Wo % 4 == 0 && env_knob("CY_STEM_QUAD", 1)) {      // four pixels per thread (bit-identical; read per call: tests)
        const long total4 = (long)a.B * a.Ho * (a.Wo / 4) * (a.Cout / 16);
        const int grid4 = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
        if (p == PREC_F16X3) hipLaunchKernelGGL((stem_quad_kernel<float, true>), dim3(grid4), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(stem_quad_kernel<float>, dim3(grid4), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const long total = (long)a.B * a.Ho * a.Wo * (a.Cout / 16);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (p == PREC_F16) hipLaunchKernelGGL(stem_kernel<f16>, dim3(grid), dim3(256), 0, s, a);
    else if (p == PREC_F16X3) hipLaunchKernelGGL((stem_kernel<float, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(stem_kernel<float>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// packed stem weights for stem_mfma_kernel: [64 rows, permuted like pack_weights][32] fp16, k = (kh*3+kw)*3 + c
void pack_stem_weights(const float* W, int cout, void* dst) {
    f16* o = reinterpret_cast<f16*>(dst);
    for (int row = 0; row < 64; ++row) {
        const int ni = (row >> 4) & 3, rr = row & 15;
        const int n = (rr >> 2) * 16 + ni * 4 + (rr & 3);
        for (int k = 0; k < 32; ++k) {
            float v = 0.0f;
            if (n < cout && k < 27) { const int tap = k / 3, c = k % 3; v = W[((size_t)n * 3 + c) * 9 + tap]; }
            o[row * 32 + k] = (f16)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ stem + first down conv
// model.0 (3x3 s2, 3 -> 64) and model.1 (3x3 s2, 64 -> 128) fused.  As separate layers they are the two slowest launches
// of the forward pass and both HBM-bound: the 64-channel half-resolution map is 8.4 MB per 512x512 tile, written once and
// read back ~1.6 times (the nine taps of a stride-2 conv come back long after each other: the L2 does not hold them).
// Here a workgroup owns 8 x 32 output pixels of model.1 x all 128 channels:
//   phase 1: the 17 x 65 stem pixels under them are computed on the matrix cores (K = 9 taps x 4 NHWC channels = 36,
//            padded to 64: a lane's k-chunk is two whole input pixels = two 8-byte loads) and written, bias + SiLU applied,
//            to LDS as fp16 [row][64 ch] (138 KiB).  Stem pixels outside the map are model.1's zero padding: zeros.
//            Rows are split by column parity (even columns first), so the 16 pixels of a stride-2 fragment are 16
//            CONSECUTIVE LDS rows and the usual chunk ^ (row & 7) swizzle keeps ds_read_b128 conflict-free.
//   phase 2: 9 taps x 2 K-halves; a wave owns 64 px x 64 ch, reads its pixel fragments from LDS and its weight fragments
//            straight from global memory (the 147 KB panel is L2-resident; no LDS left for it), one step ahead.
// HBM traffic: input (0.5 MB/tile x 1.08 halo) + output (4.2 MB/tile) instead of + 8.4 MB written + >= 8.4 MB read.
// Measured at batch 256 (CY_SD_DBG phase switches): 1.38 ms against 1.03 + 1.13 ms for the two layers; phase 1 0.70 ms (one
// exposed gather latency + 144 SiLU per lane per tile), phase 2 0.39 ms, epilogue + stores 0.38 ms.  An 8 x 16-pixel variant
// with two workgroups per CU (72 KiB, 114 VGPRs) was no faster: its 8-MFMA steps are too short to cover the weight fetch.
constexpr int SD_TH = 8, SD_TW = 32, SD_PH = 2 * SD_TH + 1, SD_PW = 2 * SD_TW + 1, SD_EVEN = SD_TW + 1;
constexpr int SD_ROWS = SD_PH * SD_PW, SD_FRAGS = (SD_ROWS + 15) / 16, SD_LDS = SD_FRAGS * 16 * 128;

__device__ __forceinline__ u32x2 load_b64(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
}

__global__ __launch_bounds__(512) void stem_down_kernel(const StemDownArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tiles_x = (a.Wo + SD_TW - 1) / SD_TW, tiles_y = (a.Ho + SD_TH - 1) / SD_TH;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = id % tiles_x; id /= tiles_x;
    const int ty = id % tiles_y;
    const int b = id / tiles_y;
    const int oy0 = ty * SD_TH, ox0 = tx * SD_TW;
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;          // stem-map coordinates of LDS pixel (0, 0)
    const auto rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.in_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt32), 0, a.wgt32_bytes, 0x00020000);

    // weight fragments of step (tap, h): rows wn*64 + ni*16 + fr of chunk h, bytes fq*16.. ; cpad = 128 rows of 64 B
    constexpr int SD_RING = 4;                                // ring: fragments are requested three steps ahead (a ring of 8 measured
    f16x8 wa[SD_RING][4];                                     // 6 % slower: its 28 loads per lane up front delay phase 1's gathers)
    const unsigned wl = (unsigned)((wn * 64 + fr) * 64 + fq * 16);
    auto load_wa = [&](f16x8* dst, int step) {
        const int h = step & 1, tap = step >> 1;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) dst[ni] = __builtin_bit_cast(f16x8, load_b128(rsw, wl + ni * 1024, (h * 9 + tap) * 8192));
    };
#pragma unroll
    for (int st = 0; st < SD_RING - 1; ++st) load_wa(wa[st], st);      // in flight during phase 1

    if (!(a.dbg & 1)) {   // ---- phase 1: stem pixels -> LDS
        const f16* wp = reinterpret_cast<const f16*>(a.wpk2);
        f16x8 sw0[4], sw1[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            sw0[ni] = *reinterpret_cast<const f16x8*>(wp + (ni * 16 + fr) * 64 + fq * 8);
            sw1[ni] = *reinterpret_cast<const f16x8*>(wp + (ni * 16 + fr) * 64 + 32 + fq * 8);
        }
        float bv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) bv[j] = a.bias0[fq * 16 + j];
        const int t0 = 2 * fq, t1 = 2 * fq + 1;              // the two taps of this lane's k-chunk; tap 8 rides in the second MFMA (fq = 0)
        const int dh0 = t0 / 3 - 1, dw0 = t0 % 3 - 1, dh1 = t1 / 3 - 1, dw1 = t1 % 3 - 1;
        const int d0 = (dh0 * a.Wi + dw0) * 8, d1 = (dh1 * a.Wi + dw1) * 8, d2 = (a.Wi + 1) * 8;
        constexpr int NG = (SD_FRAGS + 7) / 8;
        u32x2 q0[NG], q1[NG], q2[NG];
        unsigned inmask = 0;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = gi * 8 + wave, p = g * 16 + fr;
            const int sy = p / SD_PW, q = p - sy * SD_PW;
            const int sx = q < SD_EVEN ? 2 * q : 2 * (q - SD_EVEN) + 1;
            const int Y = sy0 + sy, X = sx0 + sx;
            const bool inmap = p < SD_ROWS && (unsigned)Y < (unsigned)a.H1 && (unsigned)X < (unsigned)a.W1;
            inmask |= inmap ? (1u << gi) : 0u;
            // input pixel (2Y + dh, 2X + dw): with Hi = 2*H1 and Wi = 2*W1 only the -1 row / column can fall outside
            const int hc = 2 * Y, wc = 2 * X;
            const int base = ((b * a.Hi + hc) * a.Wi + wc) * 8;
            const bool ok0 = inmap && ((hc + dh0) | (wc + dw0)) >= 0, ok1 = inmap && ((hc + dh1) | (wc + dw1)) >= 0;
            q0[gi] = load_b64(rsi, ok0 ? (unsigned)(base + d0) : CY_OOB);
            q1[gi] = load_b64(rsi, ok1 ? (unsigned)(base + d1) : CY_OOB);
            q2[gi] = load_b64(rsi, (inmap && fq == 0) ? (unsigned)(base + d2) : CY_OOB);
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = gi * 8 + wave;
            if (g >= SD_FRAGS) break;                         // wave-uniform
            const int p = g * 16 + fr;
            const bool inmap = (inmask >> gi) & 1u;
            // the fourth NHWC channel is padding: its weights are zero, and masking it keeps a stray NaN out of the sum
            const u32x4 u0 = {q0[gi].x, q0[gi].y & 0xFFFFu, q1[gi].x, q1[gi].y & 0xFFFFu};
            const u32x4 u1 = {q2[gi].x, q2[gi].y & 0xFFFFu, 0u, 0u};
            const f16x8 x0 = __builtin_bit_cast(f16x8, u0), x1 = __builtin_bit_cast(f16x8, u1);
            f32x4 acc[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw0[ni], x0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw1[ni], x1, acc[ni], 0, 0, 0);
            }
            f16x8 o0, o1;
            float v16[16];
            bias_act16(acc[0], acc[1], acc[2], acc[3], bv, true, v16);
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (f16)v16[j]; o1[j] = (f16)v16[8 + j]; }
            const unsigned keep = inmap ? 0xFFFFFFFFu : 0u;   // a select on the packed result: a `?:` around silu becomes 16 branches
            const u32x4 k4 = {keep, keep, keep, keep};
            char* row = smem + p * 128;
            *reinterpret_cast<u32x4*>(row + (((2 * fq) ^ (p & 7)) << 4)) = __builtin_bit_cast(u32x4, o0) & k4;
            *reinterpret_cast<u32x4*>(row + (((2 * fq + 1) ^ (p & 7)) << 4)) = __builtin_bit_cast(u32x4, o1) & k4;
        }
    }
    __syncthreads();

    // ---- phase 2: 3x3 stride 2 over the LDS patch
    f32x4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[ni][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pl = wm * 4 * SD_PW + fr;                      // LDS row of (first output row of this wave, tap (0,0), column fr)
    if (!(a.dbg & 2))
#pragma unroll
    for (int step = 0; step < 18; ++step) {
        const int tap = step >> 1, h = step & 1, kh = tap / 3, kw = tap % 3;
        if (step + SD_RING - 1 < 18) load_wa(wa[(step + SD_RING - 1) % SD_RING], step + SD_RING - 1);
        __builtin_amdgcn_sched_barrier(0);                   // (left alone the compiler sinks each load to just before its MFMAs)
        f16x8 xb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int cm = (2 * (m >> 1) + kh) * SD_PW + (m & 1) * 16 + (kw == 1 ? SD_EVEN : (kw == 2 ? 1 : 0));
            const int p = pl + cm;
            xb[m] = *reinterpret_cast<const f16x8*>(smem + p * 128 + (((h * 4 + fq) ^ (p & 7)) << 4));
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[ni][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[step % SD_RING][ni], xb[m], acc[ni][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    const int cbase = wn * 64 + fq * 16;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = a.bias1[cbase + j];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int oy = oy0 + wm * 2 + (m >> 1), ox = ox0 + (m & 1) * 16 + fr;
        if (oy >= a.Ho || ox >= a.Wo || (a.dbg & 4)) continue;
        const long pix = ((long)b * a.Ho + oy) * a.Wo + ox;
        f16x8 o0, o1;
        float v16[16];
        bias_act16(acc[0][m], acc[1][m], acc[2][m], acc[3][m], bv, true, v16);
#pragma unroll
        for (int j = 0; j < 8; ++j) { o0[j] = (f16)v16[j]; o1[j] = (f16)v16[8 + j]; }
        f16* dst = reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase;
        *reinterpret_cast<f16x8*>(dst) = o0;
        *reinterpret_cast<f16x8*>(dst + 8) = o1;
    }
}

// Two-group form of stem_down_kernel (same arithmetic, same outputs): the one-group kernel runs its three parts one after the other
// on all eight waves -- stem pixels -> LDS 0.66 ms (gather latency + 144 SiLUs per lane), 3x3 s2 from LDS 0.47 ms (MFMAs, weights
// from L2), epilogue + stores 0.30 ms per 256 tiles -- and its 138 KiB patch leaves no room for a second workgroup.  Here a
// persistent workgroup (one per CU) has two groups of four waves (waves w and w + 4 share a SIMD), each with its own 72 KiB
// patch of 8 x 16 output pixels; the patches of the workgroup alternate between the groups, and in every phase one group fills
// its patch (VALU / memory latency) while the other convolves and stores its previous one (matrix cores): one barrier per phase.
constexpr int S2_TH = 8, S2_TW = 16, S2_PH = 2 * S2_TH + 1, S2_PW = 2 * S2_TW + 1, S2_EVEN = S2_TW + 1;
constexpr int S2_ROWS = S2_PH * S2_PW, S2_FRAGS = (S2_ROWS + 15) / 16, S2_BUF = S2_FRAGS * 16 * 128, S2_LDS = 2 * S2_BUF;
static_assert(S2_FRAGS % 4 == 0, "fragments split evenly over the four waves of a group");

__global__ __launch_bounds__(512) void stem_down2_kernel(const StemDownArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wn = w4 & 1, wm = w4 >> 1;
    char* const buf = smem + grp * S2_BUF;                   // this group's patch
    const int tiles_x = (a.Wo + S2_TW - 1) / S2_TW, tiles_y = (a.Ho + S2_TH - 1) / S2_TH;
    const int npatch = a.B * tiles_y * tiles_x;
    const int NP = (int)blockIdx.x < npatch ? (npatch - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;     // patches of this workgroup
    const auto rsi = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.in_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wgt32), 0, a.wgt32_bytes, 0x00020000);

    // weight fragments of step (tap, h) of the 3x3 s2 conv: rows w4*32 + ni*16 + fr of chunk h, bytes fq*16..
    // Round 3: a wave convolves ALL 8 output rows of the patch x 32 channels (was 4 rows x 64 channels).  The weight fragments come
    // straight from L2, per wave and per patch: 4 fragments x 18 steps = 72 KB per wave, 288 KB per patch -- at the 16-17 B per cycle a
    // CU's fetch path sustains (section 4 of DESIGN.md) that alone was ~9 us per patch, the whole phase.  Two fragments per step feeding
    // eight pixel fragments halve it (the pixel fragments are LDS reads); outputs are bit-identical (same K order per output).
    // Measured: 1.17-1.25 -> 1.09-1.11 ms per 256 tiles, less than the halved fetch promised: the FILL phase (27 eight-byte gathers per
    // wave and patch + 144 SiLUs per lane) now sets the phase time.  A weight-STATIONARY form was built on that reading and thrown away
    // again: four-wave workgroups, two per CU, 14 of the 18 K steps of a wave's 32-channel weight slice held in registers for the whole
    // kernel (112 VGPRs), the stem panel in LDS, gathers three fragments at a time -- bit-identical, 1.20 ms (no fetch of weights per
    // patch, but no fill / convolve overlap inside a workgroup either, and 32 B of scratch at the 256-register limit).
    constexpr int RING = 4;
    f16x8 wa[RING][2];
    const unsigned wl = (unsigned)((w4 * 32 + fr) * 64 + fq * 16);
    auto load_wa = [&](f16x8* dst, int step) {
        const int h = step & 1, tap = step >> 1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) dst[ni] = __builtin_bit_cast(f16x8, load_b128(rsw, wl + ni * 1024, (h * 9 + tap) * 8192));
    };
    const f16* wp = reinterpret_cast<const f16*>(a.wpk2);
    const int t0 = 2 * fq, t1 = 2 * fq + 1;                  // the two taps of this lane's k-chunk; tap 8 rides in the second MFMA (fq = 0)
    const int dh0 = t0 / 3 - 1, dw0 = t0 % 3 - 1, dh1 = t1 / 3 - 1, dw1 = t1 % 3 - 1;
    const int d0 = (dh0 * a.Wi + dw0) * 8, d1 = (dh1 * a.Wi + dw1) * 8, d2 = (a.Wi + 1) * 8;

    auto coords = [&](int n, int& b, int& oy0, int& ox0) {
        int id = (int)blockIdx.x + n * (int)gridDim.x;
        const int tx = id % tiles_x; id /= tiles_x;
        oy0 = (id % tiles_y) * S2_TH; ox0 = tx * S2_TW; b = id / tiles_y;
    };

    // ---- fill: the 17 x 33 stem pixels under patch n -> this group's LDS patch (bias + SiLU applied, fp16, zeros outside the map)
    auto fill = [&](int n) {
        int b, oy0, ox0;
        coords(n, b, oy0, ox0);
        const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;      // stem-map coordinates of LDS pixel (0, 0)
        // stem panel and bias: re-read per patch (L2 hits, beside the gathers) rather than 48 registers held across the convolution
        f16x8 sw0[4], sw1[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            sw0[ni] = *reinterpret_cast<const f16x8*>(wp + (ni * 16 + fr) * 64 + fq * 8);
            sw1[ni] = *reinterpret_cast<const f16x8*>(wp + (ni * 16 + fr) * 64 + 32 + fq * 8);
        }
        float bv0[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) bv0[j] = a.bias0[fq * 16 + j];
        constexpr int NG = S2_FRAGS / 4;
        u32x2 q0[NG], q1[NG], q2[NG];
        unsigned inmask = 0;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = gi * 4 + w4, p = g * 16 + fr;
            const int sy = p / S2_PW, q = p - sy * S2_PW;
            const int sx = q < S2_EVEN ? 2 * q : 2 * (q - S2_EVEN) + 1;
            const int Y = sy0 + sy, X = sx0 + sx;
            const bool inmap = p < S2_ROWS && (unsigned)Y < (unsigned)a.H1 && (unsigned)X < (unsigned)a.W1;
            inmask |= inmap ? (1u << gi) : 0u;
            const int hc = 2 * Y, wc = 2 * X;
            const int base = ((b * a.Hi + hc) * a.Wi + wc) * 8;
            const bool ok0 = inmap && ((hc + dh0) | (wc + dw0)) >= 0, ok1 = inmap && ((hc + dh1) | (wc + dw1)) >= 0;
            q0[gi] = load_b64(rsi, ok0 ? (unsigned)(base + d0) : CY_OOB);
            q1[gi] = load_b64(rsi, ok1 ? (unsigned)(base + d1) : CY_OOB);
            q2[gi] = load_b64(rsi, (inmap && fq == 0) ? (unsigned)(base + d2) : CY_OOB);
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = gi * 4 + w4, p = g * 16 + fr;
            const bool inmap = (inmask >> gi) & 1u;
            // the fourth NHWC channel is padding: its weights are zero, and masking it keeps a stray NaN out of the sum
            const u32x4 u0 = {q0[gi].x, q0[gi].y & 0xFFFFu, q1[gi].x, q1[gi].y & 0xFFFFu};
            const u32x4 u1 = {q2[gi].x, q2[gi].y & 0xFFFFu, 0u, 0u};
            const f16x8 x0 = __builtin_bit_cast(f16x8, u0), x1 = __builtin_bit_cast(f16x8, u1);
            f32x4 acc[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw0[ni], x0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw1[ni], x1, acc[ni], 0, 0, 0);
            }
            f16x8 o0, o1;
            float v16[16];
            bias_act16(acc[0], acc[1], acc[2], acc[3], bv0, true, v16);
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (f16)v16[j]; o1[j] = (f16)v16[8 + j]; }
            const unsigned keep = inmap ? 0xFFFFFFFFu : 0u;
            const u32x4 k4 = {keep, keep, keep, keep};
            char* row = buf + p * 128;
            *reinterpret_cast<u32x4*>(row + (((2 * fq) ^ (p & 7)) << 4)) = __builtin_bit_cast(u32x4, o0) & k4;
            *reinterpret_cast<u32x4*>(row + (((2 * fq + 1) ^ (p & 7)) << 4)) = __builtin_bit_cast(u32x4, o1) & k4;
        }
        // the first weight fragments of the convolution that follows the barrier: in flight across it
#pragma unroll
        for (int st = 0; st < RING - 1; ++st) load_wa(wa[st], st);
    };

    // ---- convolve + store: 3x3 stride 2 over this group's patch (wave: 4 output rows x 16 columns x 64 channels)
    auto conv_store = [&](int n) {
        int b, oy0, ox0;
        coords(n, b, oy0, ox0);
        f32x4 acc[2][8];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[ni][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned xrow[8];                                     // lane part of a pixel-fragment address for (row offset & 7) = c
#pragma unroll
        for (int c = 0; c < 8; ++c) xrow[c] = (unsigned)(fr * 128 + ((fq ^ ((fr + c) & 7)) << 4));
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            const int tap = step >> 1, h = step & 1, kh = tap / 3, kw = tap % 3;
            if (step + RING - 1 < 18) load_wa(wa[(step + RING - 1) % RING], step + RING - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {                  // two half-steps of four pixel fragments (16 registers)
                f16x8 xb[4];
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4) {
                    const int m = mh * 4 + m4;
                    const int cm = (2 * m + kh) * S2_PW + (kw == 1 ? S2_EVEN : (kw == 2 ? 1 : 0));
                    // row p = fr + cm, chunk (h*4 + fq) ^ (p & 7): one of eight lane bases (by cm & 7), the second K half = bit 6 flipped
                    xb[m4] = *reinterpret_cast<const f16x8*>(buf + ((xrow[cm & 7] ^ (unsigned)(h * 64)) + cm * 128));
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int m4 = 0; m4 < 4; ++m4)
                        acc[ni][mh * 4 + m4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[step % RING][ni], xb[m4], acc[ni][mh * 4 + m4], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // packed rows w4*32 + ni*16 + (4 fq + j) hold channels 64 (w4 >> 1) + 16 fq + 4 (2 (w4 & 1) + ni) + j: 8 contiguous channels per lane
        const int cbase = (w4 >> 1) * 64 + fq * 16 + (w4 & 1) * 8;
        f32x2 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = f32x2{a.bias1[cbase + 2 * j], a.bias1[cbase + 2 * j + 1]};
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int oy = oy0 + m, ox = ox0 + fr;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const long pix = ((long)b * a.Ho + oy) * a.Wo + ox;
            f16x8 o;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {          // bias + SiLU, two values per instruction (as bias_act16)
                    f32x2 t = f32x2{acc[ni][m][2 * hh], acc[ni][m][2 * hh + 1]} + bv[ni * 2 + hh];
                    f32x2 e = t * f32x2{-1.44269504088896341f, -1.44269504088896341f};
                    e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + f32x2{1.0f, 1.0f};
                    t = t * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                    o[ni * 4 + 2 * hh] = (f16)t[0]; o[ni * 4 + 2 * hh + 1] = (f16)t[1];
                }
            *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(a.out) + pix * a.out_ct + a.out_coff + cbase) = o;
        }
    };

    // Patch n of the workgroup belongs to group n & 1; phase n: its group fills it, phase n + 1: the same group convolves it, so in
    // every phase one group fills and the other convolves.  Phases 0 .. NP, one barrier each; every wave executes NP + 1 of them
    // (group 1 sits out phase 0; the group that does not own the last patch sits out the last phase).  One straight-line loop
    // body for both groups: with a per-phase branch on the role the two instruction streams cost 388 B of scratch.
    auto phase_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    if (grp) phase_barrier();
#pragma unroll 1
    for (int n = grp; n < NP; n += 2) {
        fill(n);
        phase_barrier();
        conv_store(n);
        phase_barrier();
    }
    if (((NP + grp) & 1) == 0) phase_barrier();
}

long stem_down_blocks(const StemDownArgs& a) {
    return (long)a.B * ((a.Ho + SD_TH - 1) / SD_TH) * ((a.Wo + SD_TW - 1) / SD_TW);
}

hipError_t launch_stem_down(const StemDownArgs& a, hipStream_t s) {
    if (a.Hi != 2 * a.H1 || a.Wi != 2 * a.W1) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(stem_down_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SD_LDS);
        attr_set = true;
    }
    StemDownArgs b2 = a; b2.dbg = dev_knob("CY_SD_DBG", 0);
    // the two-group persistent form once every workgroup gets at least two patches (both wave groups busy), else the one-group
    // form; their outputs are bit for bit the same, so the choice may depend on the launch size.  CY_STEM_V = 1 / 2 forces one
    // (read per call: tests).  256 tiles of 512^2: 1.44-1.47 -> 1.18-1.20 ms alone, 2.38 -> 1.58 ms inside the pipelined pass.
    const int v = env_knob("CY_STEM_V", 0);
    const long np = (long)a.B * ((a.Ho + S2_TH - 1) / S2_TH) * ((a.Wo + S2_TW - 1) / S2_TW);
    if (v == 2 || (v == 0 && np >= 512)) {
        static bool set2 = false;
        if (!set2) { hipFuncSetAttribute(reinterpret_cast<const void*>(stem_down2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS); set2 = true; }
        hipLaunchKernelGGL(stem_down2_kernel, dim3((unsigned)(np < 256 ? np : 256)), dim3(512), S2_LDS, s, b2);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(stem_down_kernel, dim3((unsigned)stem_down_blocks(a)), dim3(512), SD_LDS, s, b2);
    return hipGetLastError();
}

// stem panel of stem_down_kernel: [64 rows, permuted like pack_weights][64] fp16; k = tap*4 + c for taps 0..7, 32 + c for
// tap 8 (c = NHWC4 channel, the fourth is zero)
void pack_stem_weights2(const float* W, int cout, void* dst) {
    f16* o = reinterpret_cast<f16*>(dst);
    for (int row = 0; row < 64; ++row) {
        const int ni = (row >> 4) & 3, rr = row & 15;
        const int n = (rr >> 2) * 16 + ni * 4 + (rr & 3);
        for (int k = 0; k < 64; ++k) {
            const int tap = k < 32 ? k / 4 : 8, c = k < 32 ? k % 4 : k - 32;
            float v = 0.0f;
            if (n < cout && c < 3 && k < 36) v = W[((size_t)n * 3 + c) * 9 + tap];
            o[row * 64 + k] = (f16)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ SPPF pool
// MaxPool2d(kernel 5, stride 1, padding 2) with implicit -inf padding, slice -> slice of one NHWC buffer.
template <typename T>
__global__ __launch_bounds__(256) void pool5_kernel(const PoolArgs a) {
    constexpr int V = 16 / sizeof(T);
    typedef T vec __attribute__((ext_vector_type(V)));
    const int cv = a.C / V;
    const long total = (long)a.B * a.H * a.W * cv;
    // (single-pass launch; workgroups in XCD-contiguous order: the 5 x 5 windows of neighbouring rows meet in one L2)
    for (long idx = (long)xcd_contiguous(blockIdx.x, gridDim.x) * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * V;
        const long pix = idx / cv;
        const int w = (int)(pix % a.W), h = (int)((pix / a.W) % a.H), b = (int)(pix / ((long)a.W * a.H));
        // window positions clamped into the image instead of skipped: a clamped position is another pixel of the same window, so the
        // maximum is unchanged, and the 25 loads are unconditional (as `if (inside) load` each one had its own branch and a full wait)
        const T* base = reinterpret_cast<const T*>(a.src) + a.src_coff + c;
        vec m = *reinterpret_cast<const vec*>(base + (((long)b * a.H + h) * a.W + w) * a.ct);
#pragma unroll
        for (int dh = -2; dh <= 2; ++dh) {
            int hh = h + dh;
            hh = hh < 0 ? 0 : (hh >= a.H ? a.H - 1 : hh);
#pragma unroll
            for (int dw = -2; dw <= 2; ++dw) {
                if (dh == 0 && dw == 0) continue;
                int ww = w + dw;
                ww = ww < 0 ? 0 : (ww >= a.W ? a.W - 1 : ww);
                const vec v = *reinterpret_cast<const vec*>(base + (((long)b * a.H + hh) * a.W + ww) * a.ct);
#pragma unroll
                for (int j = 0; j < V; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
            }
        }
        *reinterpret_cast<vec*>(reinterpret_cast<T*>(a.dst) + pix * a.ct + a.dst_coff + c) = m;
    }
}

// fp16x3 context: the maximum of hi + lo, stored as the halves of the winning pixel (a max picks one of its inputs, so no rounding)
__global__ __launch_bounds__(256) void pool5_x3_kernel(const PoolArgs a) {
    const int cv = a.C / 8;
    const long total = (long)a.B * a.H * a.W * cv;
    // (single-pass launch; workgroups in XCD-contiguous order: the 5 x 5 windows of neighbouring rows meet in one L2)
    for (long idx = (long)xcd_contiguous(blockIdx.x, gridDim.x) * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * 8;
        const long pix = idx / cv;
        const int w = (int)(pix % a.W), h = (int)((pix / a.W) % a.H), b = (int)(pix / ((long)a.W * a.H));
        float m[8]; f16x8 mh, ml;
        bool first = true;
        for (int dh = -2; dh <= 2; ++dh)
            for (int dw = -2; dw <= 2; ++dw) {
                const int hh = h + dh, ww = w + dw;
                if ((unsigned)hh >= (unsigned)a.H || (unsigned)ww >= (unsigned)a.W) continue;
                const f16* sp = reinterpret_cast<const f16*>(a.src) + (((long)b * a.H + hh) * a.W + ww) * a.ct + a.src_coff + c;
                const f16x8 vh = *reinterpret_cast<const f16x8*>(sp), vl = *reinterpret_cast<const f16x8*>(sp + a.lo);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = (float)vh[j] + (float)vl[j];
                    if (first || v > m[j]) { m[j] = v; mh[j] = vh[j]; ml[j] = vl[j]; }
                }
                first = false;
            }
        f16* dp = reinterpret_cast<f16*>(a.dst) + pix * a.ct + a.dst_coff + c;
        *reinterpret_cast<f16x8*>(dp) = mh;
        *reinterpret_cast<f16x8*>(dp + a.lo) = ml;
    }
}

hipError_t launch_pool5(Precision p, const PoolArgs& a, hipStream_t s) {
    if (p == PREC_F16X3) {
        if (a.C % 8 || a.lo <= 0) return hipErrorInvalidValue;
        const long total = (long)a.B * a.H * a.W * (a.C / 8);
        hipLaunchKernelGGL(pool5_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const int V = p == PREC_F16 ? 8 : 4;
    if (a.C % V) return hipErrorInvalidValue;
    const long total = (long)a.B * a.H * a.W * (a.C / V);
    const int grid = (int)((total + 255) / 256);
    if (p == PREC_F16) hipLaunchKernelGGL(pool5_kernel<f16>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pool5_kernel<float>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace cy
