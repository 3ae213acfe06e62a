// Per-tile preprocessing on device (gfx950): tiles never leave HBM.
//
// Replaces, for B same-shape tiles cropped from the resident mosaic:
//   utils.read_fits_crop value semantics   caesar_yolo/utils.py:373-394 (non-finite -> 0; big-endian FITS floats)
//   Analyzer.predict                        caesar_yolo/evaluation.py:146-176 (3-channel float64 cube, pipeline, None and
//                                           constant-row rejection)
//   DataPreprocessor + CLI-reachable stages caesar_yolo/preprocessing.py: BkgSubtractor :591-658, SigmaClipShifter :664-717,
//                                           SigmaClipper :723-771, ZScaleTransformer :934-971, HistEqualizer :977-1012,
//                                           Chan3Trasformer :1020-1072, MinMaxNormalizer :75-111
//   astropy ZScaleInterval / sigma_clip / sigma_clipped_stats and skimage equalize_hist as restated in SURVEY.md A.2-A.4
//   ultralytics LetterBox + channel flip + /255
//
// Every stage is "global statistics of the stage input, then a per-pixel map, then zero where the stage input was 0 or
// non-finite".  The statistics kernel therefore never materialises intermediate images: a stage's input pixel is
// recomputed from the raw fp32 pixel by replaying the already-solved maps in float64, operation for operation as numpy
// does (contraction off), so thresholds see bit-identical values.  One 1024-thread workgroup owns one (tile, channel
// program) and walks its stages.
//
// Medians are exact and cost three passes.  Every stage map is weakly monotone in its input (subtract / divide by a positive
// constant / clamp / interpolate a cdf; a reversed MINMAX range in front of a sigma-clip stage is refused on the host), and a
// pixel that hits 0 stays 0 and leaves every later set, so within a set the order of the replayed float64 values IS the
// order of the raw fp32 pixels: the k-th smallest value is the replayed k-th smallest raw pixel, found by a radix select
// over 32-bit keys (11 + 11 + 10 bits) instead of 64-bit ones.  The first level's histogram is filled by the same pass that
// accumulates the set's count and moments, and an even count selects ranks n/2-1 and n/2 together (two prefixes, two
// histograms), so one sigma-clip iteration is 3 passes over the tile where the first version of this kernel made 9.
#include "cy_kernels.h"
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <initializer_list>
#pragma clang fp contract(off)

namespace cy {

// developer diagnostics (-DCY_STAMPS_ENABLED=1 builds only): shader cycles spent in the phases of the statistics kernel, summed
// over workgroups: [0] moments passes, [1] radix-select medians, [2] bracket selections in LDS, [3] everything else of a sigma-clip run,
// [4] zscale, [5] histeq, [6] minmax
#ifndef CY_STAMPS_ENABLED
#define CY_STAMPS_ENABLED 0
#endif
__device__ unsigned long long g_pre_stamps[8];
__device__ __forceinline__ unsigned long long pre_now() { return CY_STAMPS_ENABLED ? __builtin_amdgcn_s_memtime() : 0ull; }
__device__ __forceinline__ void pre_acc(int slot, unsigned long long t0) {
    if (CY_STAMPS_ENABLED && threadIdx.x == 0) atomicAdd(&g_pre_stamps[slot], __builtin_amdgcn_s_memtime() - t0);
}
void debug_read_pre_stamps(unsigned long long* out8, bool reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pre_stamps), 8 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_pre_stamps), z, sizeof(z)); }
}

constexpr int NT = 1024;            // threads per statistics workgroup
constexpr int NWAVE = NT / 64;
constexpr int PSTRIDE = MAX_STAGES * 4;
constexpr int HEQ_STRIDE = 520;     // 256 centres + 256 cdf (+pad) doubles per (tile, channel)
constexpr int NCAND = 24576;        // capacity of the median bracket (raw keys in LDS, 96 KiB)

__device__ __forceinline__ bool cond_of(double v) { return v != 0.0 && isfinite(v); }

__device__ __forceinline__ double interp256(double x, const double* xp, const double* fp, float hinv = -1.0f) {
    // numpy.interp for 256 knots (compiled_base.c arr_interp): left/right clamps, exact knot hits, slope form
    if (x > xp[255]) return fp[255];
    if (x < xp[0]) return fp[0];
    int j;                                      // largest j with xp[j] <= x
    if (hinv >= 0.0f) {
        // the knots are bin centres, evenly spaced up to rounding: a float32 guess (hinv = 255 / (xp[255] - xp[0])) and a walk to the
        // exact answer (zero or one step) instead of eight dependent table reads
        j = (int)((float)(x - xp[0]) * hinv);
        j = j < 0 ? 0 : (j > 255 ? 255 : j);
        while (j < 255 && xp[j + 1] <= x) ++j;
        while (j > 0 && xp[j] > x) --j;
    } else {
        int lo = 0, hi = 255;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
        j = (xp[hi] <= x) ? hi : lo;
    }
    if (j == 255) return fp[255];
    if (xp[j] == x) return fp[j];
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    return slope * (x - xp[j]) + fp[j];
}

// stage k applied to its input value v, with solved parameters sp[4]
__device__ __forceinline__ double apply_stage(int op, double q0, double q1, const double* sp, const double* heq, double v, float hinv = -1.0f) {
    const bool c = cond_of(v);
    double o = v;
    switch (op) {
        case OP_BKG: o = v - sp[0]; break;
        case OP_SHIFT: o = v - sp[0]; if (o < 0.0) o = 0.0; break;
        case OP_CLIP: if (o < sp[0]) o = sp[0]; if (o > sp[1]) o = sp[1]; break;
        case OP_ZSCALE: {
            o = v - sp[0];
            const double rng = sp[1] - sp[0];
            if (rng != 0.0) o = o / rng;
            o = fmin(fmax(o, 0.0), 1.0);
            break;
        }
        case OP_HISTEQ: o = interp256(v, heq, heq + 256, hinv); break;
        case OP_MINMAX: o = (v - sp[0]) / (sp[1] - sp[0]) * (q1 - q0) + q0; break;
        default: break;
    }
    return c ? o : 0.0;
}

// ---------------------------------------------------------------------------------------- block state
struct Smem {
    double red[3 * NWAVE];
    unsigned histA[2048], histB[2048];
    unsigned scan[NT], scan2[NT];
    double zs[1024];
    unsigned char bad[1024], bad2[1024];
    double heq[512];                     // HISTEQ tables of this (tile, channel), once solved
    int op[MAX_STAGES], flag[MAX_STAGES]; // this channel's program ...
    double q0[MAX_STAGES], q1[MAX_STAGES];
    double par[MAX_STAGES * 4];          // ... and the parameters solved so far
    unsigned long long bcu[4];
    int variant;
    unsigned ncand;                      // raw keys of the set members inside the median bracket of the current pass
    unsigned cand[NCAND];
};

__device__ __forceinline__ double chain_value(const Smem& s, int upto, double raw) {
    double v = raw;
    for (int k = 0; k < upto; ++k) v = apply_stage(s.op[k], s.q0[k], s.q1[k], s.par + k * 4, s.heq, v);
    return v;
}

// Is the chain's output for this raw pixel non-zero (= does the pixel count for a MINMAX stage behind the chain)?  Every stage but
// the chain's last is evaluated exactly; of the last one only the zero test is needed, which has a closed form for the stages that end
// in a division (ZSCALE) or a subtraction (BKG, SHIFT), so the pass of a [CLIP, ZSCALE, MINMAX] program costs two clamps and three
// compares per pixel instead of a float64 division.
__device__ __forceinline__ bool chain_nonzero(const Smem& s, int upto, float rf) {
    if (!(rf != 0.0f) || !isfinite(rf)) return false;
    double v = (double)rf;
    for (int k = 0; k < upto; ++k) {
        const double* sp = s.par + k * 4;
        if (k + 1 == upto) {
            switch (s.op[k]) {
                case OP_BKG: return v != sp[0];                    // fl(v - m) == 0 iff v == m
                case OP_SHIFT: return v > sp[0];                   // v - m, negatives clamped to 0
                case OP_ZSCALE: {                                  // fmin(fmax((v - vmin) / rng, 0), 1) (no division when rng == 0)
                    const double d = v - sp[0], rng = sp[1] - sp[0];
                    if (!(d > 0.0) || rng < 0.0) return false;
                    if (rng != 0.0 && d < rng * 1e-290) return d / rng > 0.0;     // (a quotient that could underflow: decided by the division itself)
                    return true;
                }
                case OP_HISTEQ: return true;                       // interpolated cdf of a non-zero finite value: >= cdf[0] = (count of the first bin >= 1) / n > 0
                default: break;                                    // CLIP, MINMAX: evaluated
            }
        }
        v = apply_stage(s.op[k], s.q0[k], s.q1[k], sp, s.heq, v);
        if (!cond_of(v)) return false;
    }
    return true;
}

// The same test with the chain's operators and parameters in scalar registers (read from LDS once per pass instead of per pixel and
// stage: a dependent ds_read per stage made a MINMAX pass 2.5x as long as a moments pass): chains of up to three stages made of
// BKG / SHIFT / CLIP / ZSCALE, the last stage by its closed-form zero test.  ok = false: the chain is not of that kind (chain_nonzero).
struct NzChain { int n; int op[3]; double a[3], b[3]; bool ok; };
__device__ __forceinline__ double uniform_double(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xFFFFFFFFull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ NzChain nz_chain(const Smem& s, int upto) {
    NzChain c{};
    c.n = upto; c.ok = upto <= 3;
    for (int k = 0; k < 3; ++k) {
        c.op[k] = k < upto ? __builtin_amdgcn_readfirstlane(s.op[k]) : 0;
        c.a[k] = uniform_double(k < upto ? s.par[k * 4] : 0.0); c.b[k] = uniform_double(k < upto ? s.par[k * 4 + 1] : 0.0);
        if (k < upto && c.op[k] != OP_BKG && c.op[k] != OP_SHIFT && c.op[k] != OP_CLIP && c.op[k] != OP_ZSCALE &&
            !(c.op[k] == OP_HISTEQ && k + 1 == upto)) c.ok = false;       // (HISTEQ only as the last stage: its value is never zero)
        if (k + 1 < upto && c.op[k] == OP_ZSCALE) c.ok = false;      // (a ZSCALE in the middle of a chain needs its quotient: general path)
    }
    return c;
}
__device__ __forceinline__ bool nz_test(const NzChain& c, float rf) {
    if (!(rf != 0.0f) || !isfinite(rf)) return false;
    double v = (double)rf;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k >= c.n) break;
        const bool last = k + 1 == c.n;
        switch (c.op[k]) {
            case OP_BKG: if (last) return v != c.a[k]; v = v - c.a[k]; break;
            case OP_SHIFT: if (last) return v > c.a[k]; v = v - c.a[k]; if (v < 0.0) return false; break;
            case OP_CLIP: if (v < c.a[k]) v = c.a[k]; if (v > c.b[k]) v = c.b[k]; break;
            case OP_HISTEQ: return true;                           // (last stage by construction; see chain_nonzero)
            case OP_ZSCALE: {                                      // (last stage by construction)
                const double d = v - c.a[k], rng = c.b[k] - c.a[k];
                if (!(d > 0.0) || rng < 0.0) return false;
                if (rng != 0.0 && d < rng * 1e-290) return d / rng > 0.0;
                return true;
            }
            default: break;
        }
        if (!cond_of(v)) return false;
    }
    return true;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
    return v;
}
// all threads get the result; fixed tree => run-to-run deterministic
template <int MODE> __device__ __forceinline__ double block_reduce(Smem& s, double v) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = MODE == 0 ? wave_sum(v) : (MODE == 1 ? wave_min(v) : wave_max(v));
    __syncthreads();
    if (lane == 0) s.red[w] = v;
    __syncthreads();
    if (w == 0) {
        double x = lane < NWAVE ? s.red[lane] : (MODE == 0 ? 0.0 : (MODE == 1 ? INFINITY : -INFINITY));
        x = MODE == 0 ? wave_sum(x) : (MODE == 1 ? wave_min(x) : wave_max(x));
        if (lane == 0) s.red[0] = x;
    }
    __syncthreads();
    const double r = s.red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_sum(Smem& s, double v) { return block_reduce<0>(s, v); }
__device__ __forceinline__ double block_min(Smem& s, double v) { return block_reduce<1>(s, v); }
__device__ __forceinline__ double block_max(Smem& s, double v) { return block_reduce<2>(s, v); }
// three sums with one pair of barriers
__device__ __forceinline__ void block_sum3(Smem& s, double& a, double& b, double& c) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    __syncthreads();
    if (lane == 0) { s.red[w] = a; s.red[NWAVE + w] = b; s.red[2 * NWAVE + w] = c; }
    __syncthreads();
    if (w == 0) {
        double x = lane < NWAVE ? s.red[lane] : 0.0, y = lane < NWAVE ? s.red[NWAVE + lane] : 0.0, z = lane < NWAVE ? s.red[2 * NWAVE + lane] : 0.0;
        x = wave_sum(x); y = wave_sum(y); z = wave_sum(z);
        if (lane == 0) { s.red[0] = x; s.red[NWAVE] = y; s.red[2 * NWAVE] = z; }
    }
    __syncthreads();
    a = s.red[0]; b = s.red[NWAVE]; c = s.red[2 * NWAVE];
    __syncthreads();
}

__device__ __forceinline__ unsigned fkey(float f) {                     // order-preserving float -> u32
    const unsigned b = __float_as_uint(f);
    return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k >> 31) ? (k & 0x7FFFFFFFu) : ~k);
}

struct TileView {
    const float* base; int MW, tw, th, npix;
    __amdgpu_buffer_rsrc_t rs;            // the tile's rows as a raw buffer (range-checked 16-byte loads)
    __device__ __forceinline__ double raw(int i) const { const int y = i / tw, x = i - y * tw; return (double)base[(size_t)y * MW + x]; }
};

// membership of a pixel in the current sigma-clip set.  Lf / Uf: the same bounds for a stage that reads the RAW pixels (no
// earlier stage in its channel program): (double)r >= L  <=>  r >= Lf with Lf the smallest float not below L, so the test
// runs in fp32 without a conversion (the passes are instruction-bound: ~70 instructions per pixel in the generic form)
struct ClipSet { double L, U; int use_box, bx0, bx1, by0, by1; float Lf, Uf; };
__device__ __forceinline__ float f_not_below(double x) { float f = (float)x; if ((double)f < x) f = nextafterf(f, INFINITY); return f; }
__device__ __forceinline__ float f_not_above(double x) { float f = (float)x; if ((double)f > x) f = nextafterf(f, -INFINITY); return f; }
// (the fp32 bounds are clamped to +-FLT_MAX: the open initial set [-inf, +inf] then still rejects a non-finite raw pixel, as the
// reference's isfinite mask does -- cy_preproc / cy_detect_tiles accept buffers that did not go through cy_mosaic_prepare)
__device__ __forceinline__ void set_bounds(ClipSet& cs, double L, double U) {
    cs.L = L; cs.U = U; cs.Lf = fmaxf(f_not_below(L), -3.402823466e+38f); cs.Uf = fminf(f_not_above(U), 3.402823466e+38f);
}

// One pass over the tile.  A thread owns groups of FOUR consecutive pixels of a row (one 16-byte buffer load, alignment 4)
// and keeps PXG groups in flight: every statistics pass re-reads the raw tile (1-1.6 MB per workgroup, far more than an
// XCD's L2 holds for its 32 workgroups), so a pass is bound by memory latency x bytes in flight; with one dword per thread
// and load the 16 waves of a CU had 16 KB in flight (~8 GB/s per CU), now 128 KB.  The replayed values of a group are
// computed before any of them is consumed, so the program / parameter reads from LDS are shared by the group.
// f(raw fp32, value, in_box, ok) is called by every lane of the workgroup (ok = false past the row end / tile end): it may ballot.
// RAW = true: the stage reads the raw pixels (upto == 0): value = (double)raw, nothing to replay.
constexpr int PXG = 2;
#ifndef CY_PXGP
#define CY_PXGP 2
#endif
constexpr int PXGP = CY_PXGP;       // groups per thread and step of the plain raw moments pass
template <bool RAW, typename F>
__device__ __forceinline__ void for_pixels(const Smem& s, const TileView& tv, int upto, const ClipSet* box, F&& f) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int GR = (tv.tw + 3) >> 2, NG = tv.th * GR;            // groups per row, groups in the tile
    // (y, gx) of the thread's next group, advanced by NT groups per load: one integer division per pass, not per group
    int y = (int)threadIdx.x / GR, gx = (int)threadIdx.x - y * GR;
    const int dy = NT / GR, dx = NT - dy * GR;
    const bool use_box = box && box->use_box;
    for (int g0 = 0; g0 < NG; g0 += PXG * NT) {
        f32x4 r[PXG];
        int yy[PXG], xx[PXG];
#pragma unroll
        for (int u = 0; u < PXG; ++u) {
            const int g = g0 + u * NT + (int)threadIdx.x;
            yy[u] = g < NG ? y : -1; xx[u] = gx << 2;
            const unsigned off = g < NG ? (unsigned)(y * tv.MW + (gx << 2)) * 4u : 0xFFFFFF00u;   // past the end: range check -> zeros
            r[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tv.rs, off, 0, 0));
            gx += dx; y += dy;
            if (gx >= GR) { gx -= GR; ++y; }
        }
#pragma unroll
        for (int u = 0; u < PXG; ++u) {
            double v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = RAW ? (double)r[u][e] : chain_value(s, upto, (double)r[u][e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = xx[u] + e;
                const bool ok = yy[u] >= 0 && x < tv.tw;
                const bool inb = use_box && yy[u] >= box->by0 && yy[u] < box->by1 && x >= box->bx0 && x < box->bx1;
                f(r[u][e], v[e], inb, ok);
            }
        }
    }
}

template <bool RAW>
__device__ __forceinline__ bool in_set(const ClipSet& cs, float r, double v, bool in_box) {
    if (RAW) return r != 0.0f && !in_box && r >= cs.Lf && r <= cs.Uf;     // (the resident mosaic holds no non-finite values)
    return cond_of(v) && !in_box && v >= cs.L && v <= cs.U;
}

// Histogram increment with wave-level aggregation.  Radio-map pixels are mostly background noise of one sign and exponent,
// so in the leading radix level (and in the 256-bin equalisation histogram) almost every lane of a wave hits the SAME bin:
// plain LDS atomics serialise on that address (measured: 13 ms per 96-tile batch of the 3-channel 640^2 pipeline).  Up to
// three rounds of "count the lanes that share the first pending lane's bin, one atomic for all of them" take care of the
// concentrated case; whatever is still pending afterwards (uniformly spread low bits) goes through ordinary atomics.
template <int ROUNDS = 3>
__device__ __forceinline__ void hist_add(unsigned* hist, unsigned bin, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
#pragma unroll 1
    for (int it = 0; it < ROUNDS && todo != 0ull; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned lb = (unsigned)__shfl((int)bin, leader);
        const unsigned long long same = __ballot(active && bin == lb);
        if (lane == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
        if ((same >> lane) & 1ull) active = false;
        todo &= ~same;
    }
    if (active) atomicAdd(&hist[bin], 1u);
}

// buckets of ranks kA and kB (0-based) in histograms hA / hB (the same array when both ranks share a prefix) of nb <= 2048 bins:
// -> bins, and the counts of elements in lower bins
__device__ __forceinline__ void locate2(Smem& s, const unsigned* hA, const unsigned* hB, int nb, unsigned long long kA, unsigned long long kB,
                                        unsigned* binA, unsigned long long* befA, unsigned* binB, unsigned long long* befB) {
    const int t = threadIdx.x;
    const unsigned a0 = (2 * t < nb) ? hA[2 * t] : 0u, a1 = (2 * t + 1 < nb) ? hA[2 * t + 1] : 0u;
    const unsigned b0 = (2 * t < nb) ? hB[2 * t] : 0u, b1 = (2 * t + 1 < nb) ? hB[2 * t + 1] : 0u;
    s.scan[t] = a0 + a1; s.scan2[t] = b0 + b1;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        const unsigned addA = t >= off ? s.scan[t - off] : 0u, addB = t >= off ? s.scan2[t - off] : 0u;
        __syncthreads();
        s.scan[t] += addA; s.scan2[t] += addB;
        __syncthreads();
    }
    const unsigned long long inclA = s.scan[t], exclA = inclA - (a0 + a1);
    const unsigned long long inclB = s.scan2[t], exclB = inclB - (b0 + b1);
    if (kA >= exclA && kA < inclA) {
        if (kA < exclA + a0) { s.bcu[0] = (unsigned long long)(2 * t); s.bcu[1] = exclA; }
        else { s.bcu[0] = (unsigned long long)(2 * t + 1); s.bcu[1] = exclA + a0; }
    }
    if (kB >= exclB && kB < inclB) {
        if (kB < exclB + b0) { s.bcu[2] = (unsigned long long)(2 * t); s.bcu[3] = exclB; }
        else { s.bcu[2] = (unsigned long long)(2 * t + 1); s.bcu[3] = exclB + b0; }
    }
    __syncthreads();
    *binA = (unsigned)s.bcu[0]; *befA = s.bcu[1]; *binB = (unsigned)s.bcu[2]; *befB = s.bcu[3];
    __syncthreads();
}

__device__ __forceinline__ void clear_hists(Smem& s, bool both) {
    for (int i = threadIdx.x; i < 2048; i += NT) { s.histA[i] = 0u; if (both) s.histB[i] = 0u; }
}

// Median of the set (n >= 1 members) given the level-0 histogram of its raw keys in s.histA (top 11 bits): two more passes.
template <bool RAW>
__device__ __forceinline__ double set_median(Smem& s, const TileView& tv, int upto, const ClipSet& cs, unsigned long long n) {
    unsigned long long kA = (n - 1) / 2, kB = n / 2;                 // equal for odd n
    unsigned pA = 0u, pB = 0u, pmask = 0u;
#pragma unroll 1
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int shift = lvl == 0 ? 21 : (lvl == 1 ? 10 : 0), nbits = lvl == 2 ? 10 : 11;
        const unsigned nbm = (1u << nbits) - 1u;
        const bool two = pA != pB;
        if (lvl > 0) {                                               // level 0 was filled by set_moments
            __syncthreads();
            clear_hists(s, two);
            __syncthreads();
            for_pixels<RAW>(s, tv, upto, &cs, [&](float rf, double v, bool inb, bool act) {
                act = act && in_set<RAW>(cs, rf, v, inb);
                const unsigned key = fkey(rf);
                const unsigned bin = (key >> shift) & nbm;
                hist_add(s.histA, bin, act && (key & pmask) == pA);
                if (two) hist_add(s.histB, bin, act && (key & pmask) == pB);
            });
            __syncthreads();
        }
        unsigned binA, binB; unsigned long long befA, befB;
        locate2(s, s.histA, two ? s.histB : s.histA, 1 << nbits, kA, kB, &binA, &befA, &binB, &befB);
        pA |= binA << shift; pB |= binB << shift;
        kA -= befA; kB -= befB;
        pmask |= nbm << shift;
    }
    const double a = chain_value(s, upto, (double)fkey_inv(pA));
    if (pA == pB) return a;
    const double b = chain_value(s, upto, (double)fkey_inv(pB));
    return 0.5 * (a + b);
}

struct ClipStats { double lo, hi, mean, median, std; unsigned long long n; int hits, misses; };   // hits / misses of the median bracket

// One pass over the set: count and moments about K; optionally the level-0 radix histogram of the raw keys (s.histA); and,
// given a bracket [vl, vh] of values around the expected median, the number of members below it and the raw keys of the
// members inside it (s.cand, up to NCAND), from which the median is then selected without another pass over the tile.
struct Bracket { double vl, vh; bool on; float lf, hf; bool collect; };   // collect = false: only count the members inside
// The loop of a moments pass, specialised at compile time (the passes are instruction-bound: every per-pixel branch and every
// replayed stage costs).  MODE 0: count and moments only; 1: + level-0 radix histogram (s.histA); 2: + count of the members
// inside the bracket and below it; 3: + collect the raw keys of the members inside the bracket (s.cand / s.ncand).
// Moments are accumulated branch-free (a pixel outside the set adds 0); a thread appends the candidates of a 4-pixel group
// with ONE LDS atomic, and only when it has any (~15 % of the groups at the default bracket width).
// Round 4: the loop was instruction-bound (37 instructions per pixel in its cheapest form: eight v_cndmask and seven scalar mask
// operations per pixel around five float64 operations).  A group of four pixels whose members are ALL in the set -- the rule on a
// radio tile outside its zero / NaN regions: a clip removes a fraction of a percent -- takes a branch-free path per lane: one
// min3 / max3 range test for the group, no selects, the same additions in the same order (bit-identical sums).  Groups with an
// excluded pixel, partial last groups of a row and the box mask take the general path.
template <bool RAW, int MODE>
__device__ __forceinline__ void moments_loop(Smem& s, const TileView& tv, int upto, const ClipSet& cs, double K, const Bracket& br,
                                             double& s1, double& s2, unsigned& ucnt, unsigned& below, unsigned& ucand) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int GR = (tv.tw + 3) >> 2, NG = tv.th * GR;
    int y = (int)threadIdx.x / GR, gx = (int)threadIdx.x - y * GR;
    const int dy = NT / GR, dx = NT - dy * GR;
    const bool partial = (tv.tw & 3) != 0;                    // the last group of a row reaches past the tile
    // The groups of step i + 1 are requested before those of step i are consumed (register double buffer): with the arithmetic of a
    // step down to ~1 us (round 4) the ~2 us a step waited for its own loads were two thirds of a pass.
    f32x4 rn[PXG];
    int yn[PXG], xn[PXG];
    auto request = [&](int g0) {
#pragma unroll
        for (int u = 0; u < PXG; ++u) {
            const int g = g0 + u * NT + (int)threadIdx.x;
            yn[u] = y; xn[u] = gx << 2;
            const unsigned off = g < NG ? (unsigned)(y * tv.MW + (gx << 2)) * 4u : 0xFFFFFF00u;   // past the end: zeros = not in any set
            rn[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tv.rs, off, 0, 0));
            gx += dx; y += dy;
            if (gx >= GR) { gx -= GR; ++y; }
        }
    };
    request(0);
    for (int g0 = 0; g0 < NG; g0 += PXG * NT) {
        f32x4 r[PXG];
        int yy[PXG], xx[PXG];
#pragma unroll
        for (int u = 0; u < PXG; ++u) { r[u] = rn[u]; yy[u] = yn[u]; xx[u] = xn[u]; }
        if (g0 + PXG * NT < NG) request(g0 + PXG * NT);
#pragma unroll
        for (int u = 0; u < PXG; ++u) {
            unsigned inmask = 0u;                                 // MODE >= 2: which of the group's pixels are set members inside the bracket
            {
                double v[4];
                bool act[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float rf = r[u][e];
                    v[e] = RAW ? (double)rf : chain_value(s, upto, (double)rf);
                    bool a = RAW ? (rf != 0.0f && rf >= cs.Lf && rf <= cs.Uf) : (cond_of(v[e]) && v[e] >= cs.L && v[e] <= cs.U);
                    if (partial) a = a && (xx[u] + e < tv.tw);
                    if (cs.use_box) a = a && !(yy[u] >= cs.by0 && yy[u] < cs.by1 && xx[u] + e >= cs.bx0 && xx[u] + e < cs.bx1);
                    act[e] = a;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = v[e] - K, dm = act[e] ? d : 0.0;
                    s1 += dm; s2 += dm * dm; ucnt += act[e] ? 1u : 0u;
                }
                if (MODE == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) hist_add(s.histA, fkey(r[u][e]) >> 21, act[e]);
                }
                if (MODE >= 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float rf = r[u][e];
                        inmask |= (act[e] && (RAW ? (rf >= br.lf && rf <= br.hf) : (v[e] >= br.vl && v[e] <= br.vh))) ? (1u << e) : 0u;
                        below += (act[e] && (RAW ? rf < br.lf : v[e] < br.vl)) ? 1u : 0u;
                    }
                }
            }
            if (MODE == 2) ucand += (unsigned)__popc(inmask);
            if (MODE == 3) {
                // the bracket's members are appended to s.cand with ONE LDS atomic per wave and group step: a per-lane atomicAdd on
                // the single counter serialises in the LDS (15 % of the groups carry a candidate: ~15k same-address atomics per pass and
                // workgroup, which is what a pass cost -- not its arithmetic).  Slots: pixel e of all lanes, then pixel e + 1.
                if (__ballot(inmask != 0u) != 0ull) {                 // wave-uniform
                    const int lane = (int)threadIdx.x & 63;
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    unsigned long long m[4];
                    unsigned tot = 0u, my[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m[e] = __ballot(((inmask >> e) & 1u) != 0u);
                        my[e] = tot + (unsigned)__popcll(m[e] & lt);
                        tot += (unsigned)__popcll(m[e]);
                    }
                    unsigned base = 0u;
                    if (lane == 0) base = atomicAdd(&s.ncand, tot);
                    base = (unsigned)__shfl((int)base, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if ((inmask >> e) & 1u) { const unsigned pos = base + my[e]; if (pos < (unsigned)NCAND) s.cand[pos] = fkey(r[u][e]); }
                }
            }
        }
    }
}

// The same pass for the case nearly every sigma-clip pass of the benchmarks is: a stage that reads the RAW pixels of a tile whose rows are
// whole groups of four, no masked box, no histogram.  The ISA of the general loop above spends ~40 vector instructions per pixel, most
// of them on materialising booleans (row-end / box / pass-flag combinations as 0/1 registers) -- a pass of 16 waves over 410k pixels is
// bound by instruction issue (4 waves per SIMD x ~330 instructions per 8-pixel step), not by the memory path.  Here every test is a
// compare whose result IS the wave's ballot (an SGPR pair): counts (members, members below the bracket, members inside it) are
// s_bcnt1 sums in scalar registers, the moments are accumulated under the member mask with one cvt, one subtract, one add and one fma per
// pixel, and the bracket's members are appended with one LDS atomic per wave and group.  ~12 vector instructions per pixel.
// MODE 0: count + moments; 2: + members below / inside the bracket counted; 3: + the bracket's members appended to s.cand.
template <int MODE, int G>
__device__ __forceinline__ void moments_plain(Smem& s, const TileView& tv, const ClipSet& cs, double K, const Bracket& br,
                                              double& s1, double& s2, unsigned& ucnt, unsigned& below, unsigned& ucand) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int GR = tv.tw >> 2, NG = tv.th * GR;
    const int lane = (int)threadIdx.x & 63;
    const float Lf = cs.Lf, Uf = cs.Uf, blf = br.lf, bhf = br.hf;
    unsigned wcnt = 0u, wbelow = 0u, wcand = 0u;              // wave-uniform (scalar registers)
    // byte offset of the thread's next group, advanced by NT groups per load without a multiplication: dy rows and dx groups further, one
    // more row when the column wraps
    int gx, g = (int)threadIdx.x;
    unsigned off;
    { const int y0 = g / GR; gx = g - y0 * GR; off = (unsigned)(y0 * tv.MW + (gx << 2)) * 4u; }
    const int dy = NT / GR, dx = NT - dy * GR;
    const unsigned step_a = (unsigned)(dy * tv.MW + (dx << 2)) * 4u, step_b = step_a + (unsigned)(tv.MW - tv.tw) * 4u;
    auto request = [&](f32x4 (&dst)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            dst[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tv.rs, g < NG ? off : 0xFFFFFF00u, 0, 0));   // past the end: zeros = not in any set
            g += NT; gx += dx;
            const bool wrap = gx >= GR;
            gx -= wrap ? GR : 0;
            off += wrap ? step_b : step_a;
        }
    };
    auto consume = [&](const f32x4 (&r)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            bool in[4] = {false, false, false, false};
            unsigned long long mc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float rf = r[u][e];
                const bool a = rf >= Lf && rf <= Uf && rf != 0.0f;         // (NaN fails; +-inf is outside the clamped bounds)
                // (the ballot of a plain compare IS its result register; the ballot of a combined boolean costs two more vector instructions)
                const unsigned long long ma = __builtin_amdgcn_ballot_w64(rf >= Lf) & __builtin_amdgcn_ballot_w64(rf <= Uf) & __builtin_amdgcn_ballot_w64(rf != 0.0f);
                wcnt += (unsigned)__builtin_popcountll(ma);
                double d = (double)rf - K;
                d = a ? d : 0.0;
                s1 += d; s2 = __builtin_fma(d, d, s2);
                if (MODE >= 2) {
                    const bool lt = rf < blf, le = rf <= bhf;
                    const unsigned long long mlt = __builtin_amdgcn_ballot_w64(lt);
                    wbelow += (unsigned)__builtin_popcountll(ma & mlt);
                    in[e] = a && !lt && le;
                    mc[e] = ma & ~mlt & __builtin_amdgcn_ballot_w64(le);
                    if (MODE == 2) wcand += (unsigned)__builtin_popcountll(mc[e]);
                }
            }
            if (MODE == 3) {
                if ((mc[0] | mc[1] | mc[2] | mc[3]) != 0ull) {             // wave-uniform
                    unsigned tot = 0u, my[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                          // slots: pixel e of all lanes, then pixel e + 1
                        my[e] = __builtin_amdgcn_mbcnt_hi((unsigned)(mc[e] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mc[e], tot));
                        tot += (unsigned)__builtin_popcountll(mc[e]);
                    }
                    unsigned base = 0u;
                    if (lane == 0) base = atomicAdd(&s.ncand, tot);
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    if (base + tot <= (unsigned)NCAND) {                   // (an overflowing bracket is a miss anyway: s.ncand says so)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (in[e]) s.cand[base + my[e]] = fkey(r[u][e]);
                    }
                }
            }
        }
    };
    // two register sets in turn: the groups of the step after next are requested before this step's are consumed, and no set is ever copied
    f32x4 ra[G], rb[G];
    request(ra);
    for (int g0 = 0; g0 < NG; g0 += 2 * G * NT) {
        const bool more1 = g0 + G * NT < NG, more2 = g0 + 2 * G * NT < NG;
        if (more1) request(rb);
        consume(ra);
        if (more2) request(ra);
        if (more1) consume(rb);
    }
    ucnt = lane == 0 ? wcnt : 0u; below = lane == 0 ? wbelow : 0u; ucand = lane == 0 ? wcand : 0u;
}

// One pass over the set: count and moments about K; optionally the level-0 radix histogram of the raw keys (s.histA); and,
// given a bracket [vl, vh] of values around the expected median, the number of members below it and the raw keys of the
// members inside it (s.cand, up to NCAND), from which the median is then selected without another pass over the tile.
template <bool RAW>
__device__ __forceinline__ void set_moments(Smem& s, const TileView& tv, int upto, const ClipSet& cs, double K, bool want_hist,
                                            const Bracket& br, unsigned long long* n_out, double* mean_out, double* std_out,
                                            unsigned long long* below_out, unsigned* ncand_out) {
    __syncthreads();
    if (want_hist) clear_hists(s, false);
    if (threadIdx.x == 0) s.ncand = 0u;
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    unsigned below = 0u, ucnt = 0u, ucand = 0u;
    bool plain = false;
    if constexpr (RAW) plain = !want_hist && (tv.tw & 3) == 0 && !cs.use_box;
    if (plain && (s.variant & 2)) plain = false;      // developer A/B (CY_PRE_VARIANT bit 1): the general loop
    if (plain) {
        if (!br.on) moments_plain<0, PXGP>(s, tv, cs, K, br, s1, s2, ucnt, below, ucand);
        else if (!br.collect) moments_plain<2, PXGP>(s, tv, cs, K, br, s1, s2, ucnt, below, ucand);
        else moments_plain<3, PXGP>(s, tv, cs, K, br, s1, s2, ucnt, below, ucand);
    }
    else if (want_hist) moments_loop<RAW, 1>(s, tv, upto, cs, K, br, s1, s2, ucnt, below, ucand);      // (never together with a bracket)
    else if (!br.on) moments_loop<RAW, 0>(s, tv, upto, cs, K, br, s1, s2, ucnt, below, ucand);
    else if (!br.collect) moments_loop<RAW, 2>(s, tv, upto, cs, K, br, s1, s2, ucnt, below, ucand);
    else moments_loop<RAW, 3>(s, tv, upto, cs, K, br, s1, s2, ucnt, below, ucand);
    __syncthreads();
    double bl = (double)below, cnt = (double)ucnt;
    block_sum3(s, cnt, s1, s2);
    double nc = (double)ucand;
    if (br.on && !want_hist) { bl = block_sum(s, bl); if (!br.collect) nc = block_sum(s, nc); }
    const unsigned long long n = (unsigned long long)cnt;
    *n_out = n; *below_out = (unsigned long long)bl; *ncand_out = (br.on && !want_hist) ? (br.collect ? s.ncand : (unsigned)nc) : 0u;
    if (n) {
        const double m1 = s1 / cnt;                       // mean - K
        double var = s2 / cnt - m1 * m1;
        if (var < 0.0) var = 0.0;
        *mean_out = K + m1; *std_out = sqrt(var);
    } else { *mean_out = NAN; *std_out = NAN; }
}

// ranks kA <= kB (0-based) among the nc raw keys in s.cand: three radix levels over LDS
__device__ __forceinline__ void select_cand(Smem& s, unsigned nc, unsigned long long kA, unsigned long long kB, unsigned* keyA, unsigned* keyB) {
    unsigned pA = 0u, pB = 0u, pmask = 0u;
#pragma unroll 1
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int shift = lvl == 0 ? 21 : (lvl == 1 ? 10 : 0), nbits = lvl == 2 ? 10 : 11;
        const unsigned nbm = (1u << nbits) - 1u;
        const bool two = pA != pB;
        __syncthreads();
        clear_hists(s, two);
        __syncthreads();
        for (unsigned i0 = 0; i0 < nc; i0 += NT) {               // whole waves stay in the loop: hist_add ballots
            const unsigned i = i0 + threadIdx.x;
            const bool act = i < nc;
            const unsigned key = act ? s.cand[i] : 0u;
            const unsigned bin = (key >> shift) & nbm;
            // the leading level sees one or two bins (a bracket's keys share sign and exponent): one aggregation round; below it the
            // bins are the keys' middle and low bits, evenly spread: plain atomics (three rounds everywhere: clip stage +9 %, round 4)
            if (lvl == 0) {
                hist_add<1>(s.histA, bin, act && (key & pmask) == pA);
                if (two) hist_add<1>(s.histB, bin, act && (key & pmask) == pB);
            } else {
                hist_add<0>(s.histA, bin, act && (key & pmask) == pA);
                if (two) hist_add<0>(s.histB, bin, act && (key & pmask) == pB);
            }
        }
        __syncthreads();
        unsigned binA, binB; unsigned long long befA, befB;
        locate2(s, s.histA, two ? s.histB : s.histA, 1 << nbits, kA, kB, &binA, &befA, &binB, &befB);
        pA |= binA << shift; pB |= binB << shift;
        kA -= befA; kB -= befB;
        pmask |= nbm << shift;
    }
    *keyA = pA; *keyB = pB;
}

// astropy SigmaClip._sigmaclip_noaxis (maxiters 5, median centre, std spread) + statistics of the survivors
// (sigma_clipped_stats).  Passes over the tile: 1 + 2 for the initial set (moments + first radix level, two more levels for
// its exact median), 1 to accumulate its moments about that median (var = E[d^2] - E[d]^2 then loses nothing against numpy's
// two-pass form), and ONE per clipping iteration: the pass that counts the clipped set and accumulates its moments about the
// previous median also collects the raw keys inside a narrow bracket around that median (clipping moves the median by a
// tiny fraction of sigma), and the new median is selected among them in LDS.  A bracket that misses the median or overflows
// falls back to the three-pass radix select; the bracket width follows the density measured by the previous pass.
// What a run on the raw pixels learns about its INITIAL set (every non-zero finite pixel outside the box) before the first clip: shared by
// the sigma-clip stages that open two channel programs of one tile (round 4; chan3: both sigma-clip channels start from the same set)
struct InitCache { bool valid; int use_box; double fract; unsigned long long n; double mean, sd, med, density; };

template <bool RAW>
__device__ __forceinline__ ClipStats sigma_clip_run(Smem& s, const TileView& tv, int upto, double slo, double sup, int use_box, double mask_fract,
                                                    InitCache* ic = nullptr) {
    ClipSet cs{-INFINITY, INFINITY, use_box, 0, 0, 0, 0, -INFINITY, INFINITY};
    if (use_box) {
        const int xc = tv.tw / 2, yc = tv.th / 2;
        const int dy = (int)(tv.th * mask_fract / 2.0), dx = (int)(tv.tw * mask_fract / 2.0);
        cs.bx0 = xc - dx; cs.bx1 = xc + dx; cs.by0 = yc - dy; cs.by1 = yc + dy;
        if (cs.bx0 < 0) cs.bx0 = 0;          // numpy slice clamps (negative starts cannot occur for fract <= 1)
        if (cs.by0 < 0) cs.by0 = 0;
    }
    ClipStats r{NAN, NAN, NAN, NAN, NAN, 0ull, 0, 0};
    unsigned long long n = 0, nprev = ~0ull, below = 0;
    unsigned ncand = 0;
    double mean = NAN, sd = NAN, med = NAN, K = 0.0;
    double density = 0.0;                     // set members per unit of value around the median (0: not measured yet)
    // Half-width of the bracket in RANKS: a clip moves the median by about half the difference of the counts removed above
    // and below it -- up to ~1 % of n on radio tiles (sources above +k sigma against the noise tail below) -- and the bracket
    // must still hold it.  Capacity NCAND = 24576 keys = 1.5 x the full width.
    constexpr double HALF_RANKS = 8192.0;
    // ... for the first clip.  Later clips move the median less and less: the bracket of trip c + 1 is eight times the rank shift that trip c
    // observed (the median's offset from the middle of its bracket's population) plus 2048 ranks, at most the full width -- the keys to
    // select among (and to append during the pass) drop from ~16k to ~3k (round 4: clip stage -10 %; zero misses on the benchmark tiles).
    double half_ranks = HALF_RANKS;
    Bracket br{0.0, 0.0, false, 0.0f, 0.0f, true};
    // Round 4: the median of the INITIAL set used to cost three histogram passes (the level-0 radix histogram inside the first moments
    // pass, then two more radix levels: ~100 VALU instructions per pixel each, 80 % of the kernel's vector instructions).  For a stage
    // that reads the raw pixels a SAMPLE brackets it instead: the set members of every (th / 26)-th row (16k of the 410k pixels of a
    // 640^2 tile, read as whole rows) go to s.cand, their 0.48 / 0.52 quantiles (radix select in LDS) bound the true median with
    // ~5 sigma of the sampling error of a rank, and the first pass collects the set members between them like every later pass.
    // A bracket that misses (or overflows) falls back to the histogram passes, so the median stays exact.
    const bool resume = RAW && ic && ic->valid && ic->use_box == use_box && ic->fract == mask_fract;
    if (RAW && !resume) {
        const int rstep = tv.th >= 52 ? tv.th / 26 : 1;
        __syncthreads();
        if (threadIdx.x == 0) s.ncand = 0u;
        __syncthreads();
        for (int y = rstep / 2; y < tv.th; y += rstep)
            for (int x0 = 0; x0 < tv.tw; x0 += NT) {
                const int x = x0 + (int)threadIdx.x;
                float rf = 0.0f;
                if (x < tv.tw) rf = tv.base[(size_t)y * tv.MW + x];
                bool a = x < tv.tw && rf != 0.0f && isfinite(rf);
                if (cs.use_box) a = a && !(y >= cs.by0 && y < cs.by1 && x >= cs.bx0 && x < cs.bx1);
                const unsigned long long m = __ballot(a);                  // (whole waves stay in the loops)
                if (m != 0ull) {
                    const int lane = (int)threadIdx.x & 63;
                    unsigned base = 0u;
                    if (lane == 0) base = atomicAdd(&s.ncand, (unsigned)__popcll(m));
                    base = (unsigned)__shfl((int)base, 0);
                    const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    if (a && pos < (unsigned)NCAND) s.cand[pos] = fkey(rf);
                }
            }
        __syncthreads();
        const unsigned ns = s.ncand < (unsigned)NCAND ? s.ncand : (unsigned)NCAND;
        if (ns >= 4096u) {
            unsigned keyA, keyB;
            // 5 sigma of a sample rank: 2.5 / sqrt(ns) of the set (0.02 at 16k samples); at least +-1.5 %
            double w = 2.5 / sqrt((double)ns);
            if (w < 0.015) w = 0.015;
            const unsigned long long ra = (unsigned long long)((0.5 - w) * (double)ns), rb = (unsigned long long)((0.5 + w) * (double)ns);
            select_cand(s, ns, ra, rb < ns ? rb : ns - 1, &keyA, &keyB);
            br.lf = fkey_inv(keyA); br.hf = fkey_inv(keyB);
            br.vl = (double)br.lf; br.vh = (double)br.hf;
            br.on = br.hf > br.lf; br.collect = true;
        }
        __syncthreads();
    }
    int c = -2;                               // -2: first trip (about 0), -1: initial set about its median, >= 0: clips done
    // Round 4: with a sample bracket the first trip accumulates its moments about the bracket's MIDPOINT (within a few percent of sigma
    // of the median), so they are as well conditioned as moments about the median itself, and the bracket's own population gives the
    // density: the second trip over the initial set is not needed (6 passes per run instead of 7).
    // Measured and NOT kept: listing the set members next to the clip bounds ("shells") in the trip of the second or third clip so that
    // the later trips follow from that list in LDS (count, re-centred moments, rank offset of the bracket) without a pass -- on the
    // benchmark tiles (dense bright sources: the upper bound keeps moving by several sigma) the bounds left a 6144-pixel shell in 85 %
    // of the trips, and the listing cost more than the saved passes (clip stage 1.31 -> 1.48 ms per 225 tiles).
    const bool sampled = RAW && br.on && !(s.variant & 8);
    if (sampled) { K = 0.5 * (br.vl + br.vh); if (!isfinite(K)) K = 0.0; }
    bool skip_trip = resume;                   // the first trip's results come from the cache
#pragma unroll 1
    for (;;) {
      bool hit = false;
      if (skip_trip) {
        skip_trip = false;
        n = ic->n; mean = ic->mean; sd = ic->sd; med = ic->med; density = ic->density;
        if (n == 0) { mean = sd = med = NAN; break; }
        c = 0;
      } else {
        const bool radix_trip = !br.on || !br.collect;   // this trip's median (if needed) comes from the radix select
        unsigned long long ts = pre_now();
        set_moments<RAW>(s, tv, upto, cs, K, radix_trip && c != -1, br, &n, &mean, &sd, &below, &ncand);
        pre_acc(0, ts);
        if (c >= 1 && n == nprev) break;      // the clip removed nothing: converged, (mean, sd, med) describe this very set
        if (n == 0) { mean = sd = med = NAN; break; }
        if (br.on && ncand > 0) density = (double)ncand / (br.vh - br.vl);          // measured (also when the bracket overflowed or missed)
        if (c != -1) {
            const unsigned long long kA = (n - 1) / 2, kB = n / 2;
            hit = br.on && br.collect && ncand <= (unsigned)NCAND && below <= kA && kB < below + ncand;
            if (br.on && br.collect) { if (hit) ++r.hits; else ++r.misses; }
            if (hit) {
                unsigned keyA, keyB;
                if (c >= 1 && !(s.variant & 64)) {
                    const double shift = fabs((double)(kA - below) - 0.5 * (double)ncand);
                    half_ranks = fmin(HALF_RANKS, 8.0 * shift + 2048.0);
                } else half_ranks = HALF_RANKS;
                ts = pre_now();
                select_cand(s, ncand, kA - below, kB - below, &keyA, &keyB);
                pre_acc(2, ts);
                const double a = chain_value(s, upto, (double)fkey_inv(keyA));
                med = keyA == keyB ? a : 0.5 * (a + chain_value(s, upto, (double)fkey_inv(keyB)));
            } else {
                half_ranks = HALF_RANKS;
                if (!radix_trip) {            // the bracket missed (or overflowed): level-0 histogram first, then the two radix passes
                    Bracket off{0.0, 0.0, false, 0.0f, 0.0f, true};
                    unsigned long long n2, b2; unsigned c2; double m2, s2;
                    set_moments<RAW>(s, tv, upto, cs, K, true, off, &n2, &m2, &s2, &b2, &c2);
                }
                ts = pre_now();
                med = set_median<RAW>(s, tv, upto, cs, n);
                pre_acc(1, ts);
            }
        }
      }
        if (c == -2) {
            if (sampled && hit && density > 0.0) {
                c = 0;                        // moments about the sample bracket's midpoint, density from its population
                if (ic) { ic->valid = true; ic->use_box = use_box; ic->fract = mask_fract; ic->n = n; ic->mean = mean; ic->sd = sd; ic->med = med; ic->density = density; }
            } else {
                // second trip over the initial set: accurate moments about its median, and the density of members around the median
                // (count inside +-sd/32; sd of the first trip is good enough for that) to size the first bracket
                K = med; c = -1;
                const double d0 = sd / 32.0;
                br.on = sd > 0.0 && isfinite(d0); br.collect = false;
                br.vl = med - d0; br.vh = med + d0; br.lf = f_not_below(br.vl); br.hf = f_not_above(br.vh);
                continue;
            }
        }
        if (c == -1) c = 0;
        if (c == 5) break;                    // maxiters: statistics of the final survivors, bounds of the fifth iteration
        r.lo = med - sd * slo; r.hi = med + sd * sup;
        set_bounds(cs, fmax(cs.L, r.lo), fmin(cs.U, r.hi));
        nprev = n; K = med; ++c;
        // bracket for the next trip: +-delta around the current median, delta from the measured density, else from sigma
        // (a normal core has 0.4 n / sigma members per unit at its centre)
        const double rho = density > 0.0 ? density : 0.4 * (double)n / sd;
        const double delta = half_ranks / rho;
        br.on = sd > 0.0 && isfinite(delta) && delta > 0.0; br.collect = true;
        br.vl = med - delta; br.vh = med + delta;
        br.lf = f_not_below(br.vl); br.hf = f_not_above(br.vh);
    }
    r.mean = mean; r.median = med; r.std = sd; r.n = n;
    return r;
}

// astropy ZScaleInterval.get_limits (nsamples 1000, max_reject 0.5, min_npixels 5, krej 2.5, max_iterations 5)
__device__ __forceinline__ void zscale_run(Smem& s, const TileView& tv, int upto, double contrast, double* vmin_out, double* vmax_out) {
    const int t = threadIdx.x;
    int stride = (int)fmax(1.0, (double)tv.npix / 1000.0);
    int ns = (tv.npix + stride - 1) / stride;
    if (ns > 1000) ns = 1000;
    s.zs[t] = t < ns ? chain_value(s, upto, tv.raw(t * stride)) : INFINITY;
    __syncthreads();
    for (int kk = 2; kk <= 1024; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const int ixj = t ^ j;
            if (ixj > t) {
                const bool up = (t & kk) == 0;
                const double a = s.zs[t], b = s.zs[ixj];
                if ((a > b) == up) { s.zs[t] = b; s.zs[ixj] = a; }
            }
            __syncthreads();
        }
    const int npix = ns;
    double vmin = s.zs[0], vmax = s.zs[npix - 1];
    int minpix = (int)(npix * 0.5); if (minpix < 5) minpix = 5;
    int ngrow = (int)(npix * 0.01); if (ngrow < 1) ngrow = 1;
    int ngood = npix, last = npix + 1;
    s.bad[t] = 0;
    __syncthreads();
    double slope = 0.0;
    const double y = t < npix ? s.zs[t] : 0.0, x = (double)t;
    for (int it = 0; it < 5; ++it) {
        if (ngood >= last || ngood < minpix) break;
        const double w = (t < npix && !s.bad[t]) ? 1.0 : 0.0;
        // weighted straight-line least squares (np.polyfit deg 1, w in {0,1}) in centred form
        double sw = w, sx = w * x, sy = w * y;
        block_sum3(s, sw, sx, sy);
        const double xm = sx / sw, ym = sy / sw;
        double sxx = w * (x - xm) * (x - xm), sxy = w * (x - xm) * (y - ym), zero = 0.0;
        block_sum3(s, sxx, sxy, zero);
        slope = sxy / sxx;
        const double icpt = ym - slope * xm;
        const double flat = y - (slope * x + icpt);
        const double fm = block_sum(s, w * flat) / sw;
        const double var = block_sum(s, w * (flat - fm) * (flat - fm)) / sw;
        const double thr = 2.5 * sqrt(var);
        if (t < npix && (flat < -thr || flat > thr)) s.bad[t] = 1;
        __syncthreads();
        // np.convolve(badpix, ones(ngrow), 'same'): OR over j in [i + c - (ngrow-1), i + c], c = (ngrow-1)//2
        unsigned char nb = 0;
        if (t < npix) {
            const int c = (ngrow - 1) / 2;
            int j0 = t + c - (ngrow - 1), j1 = t + c;
            if (j0 < 0) j0 = 0;
            if (j1 > npix - 1) j1 = npix - 1;
            for (int j = j0; j <= j1; ++j) nb |= s.bad[j];
        }
        s.bad2[t] = nb;
        __syncthreads();
        s.bad[t] = s.bad2[t];
        __syncthreads();
        last = ngood;
        ngood = (int)block_sum(s, (t < npix && !s.bad[t]) ? 1.0 : 0.0);
    }
    if (ngood >= minpix) {
        if (contrast > 0.0) slope = slope / contrast;
        const int center = (npix - 1) / 2;
        const double median = (npix & 1) ? s.zs[(npix - 1) / 2] : 0.5 * (s.zs[npix / 2 - 1] + s.zs[npix / 2]);
        const double lo = median - (double)(center - 1) * slope, hi = median + (double)(npix - center) * slope;
        if (lo > vmin) vmin = lo;
        if (hi < vmax) vmax = hi;
    }
    *vmin_out = vmin; *vmax_out = vmax;
    __syncthreads();
}

// MINMAX behind a chain whose non-zero test is a single threshold on the RAW pixel (round 4): the chains of the benchmarks -- nothing,
// [ZSCALE], [HISTEQ], [CLIP, ZSCALE] -- keep a pixel exactly when it is non-zero, finite and (ZSCALE last) its value after the clamp
// exceeds vmin:  clamp(v, lo, hi) - vmin > 0  <=>  v > vmin  when lo <= vmin < hi (always / never on either side of that), and
// (double)r > T  <=>  r > Tf  with Tf the largest float not above T.  The quotient of the ZSCALE stage cannot underflow to zero for
// rng <= 1e200 (d is a difference of a float and a double: never below 2^-1074, so d / rng > 0), which nz_test's slow branch guards.
// -> true and the threshold, or false: the chain is not of that kind (general pass).
__device__ __forceinline__ bool raw_threshold(const Smem& s, int upto, float* tf) {
    double T = -INFINITY;
    double lo = -INFINITY, hi = INFINITY;                         // clamp in front of the last stage (at most one CLIP)
    for (int k = 0; k < upto; ++k) {
        const int op = s.op[k];
        const double* sp = s.par + k * 4;
        const bool last = k + 1 == upto;
        if (op == OP_CLIP && !last && k == 0) {
            lo = sp[0]; hi = sp[1];
            if (!(lo < 0.0 && hi > 0.0)) return false;             // (a bound at or beyond zero changes which pixels are clamped TO zero)
        } else if (op == OP_ZSCALE && last) {
            const double vmin = sp[0], rng = sp[1] - sp[0];
            if (!(rng >= 0.0) || !(rng <= 1e200) || !isfinite(vmin)) return false;
            if (vmin < lo) T = -INFINITY;                          // every clamped value is above vmin
            else if (vmin >= hi) return false;                     // none is: leave it to the general pass
            else T = vmin;
        } else if (op == OP_HISTEQ && last && k == 0) {
            // interpolated cdf of a non-zero finite value: never zero (see chain_nonzero)
        } else return false;
    }
    *tf = T == -INFINITY ? -INFINITY : f_not_above(T);
    return true;
}
// min / max of the raw pixels with r != 0, |r| <= FLT_MAX, r > tf over a tile of whole groups: seven vector instructions per pixel
// ALL = true: every pixel of the tile counts, zeros included (the range of a HISTEQ stage); a NaN fails both compares
template <bool ALL = false>
__device__ __forceinline__ void minmax_plain(const TileView& tv, float tf, float& rmn, float& rmx) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int GR = tv.tw >> 2, NG = tv.th * GR;
    int gx, g = (int)threadIdx.x;
    unsigned off;
    { const int y0 = g / GR; gx = g - y0 * GR; off = (unsigned)(y0 * tv.MW + (gx << 2)) * 4u; }
    const int dy = NT / GR, dx = NT - dy * GR;
    const unsigned step_a = (unsigned)(dy * tv.MW + (dx << 2)) * 4u, step_b = step_a + (unsigned)(tv.MW - tv.tw) * 4u;
    constexpr int G = 2;
    auto request = [&](f32x4 (&dst)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            dst[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tv.rs, g < NG ? off : 0xFFFFFF00u, 0, 0));   // past the end: zeros
            g += NT; gx += dx;
            const bool wrap = gx >= GR;
            gx -= wrap ? GR : 0;
            off += wrap ? step_b : step_a;
        }
    };
    auto consume = [&](const f32x4 (&r)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float rf = r[u][e];
                const bool a = ALL || (rf > tf && rf <= 3.402823466e+38f && rf != 0.0f);      // (NaN fails)
                rmn = (a && rf < rmn) ? rf : rmn;
                rmx = (a && rf > rmx) ? rf : rmx;
            }
    };
    f32x4 ra[G], rb[G];
    request(ra);
    for (int g0 = 0; g0 < NG; g0 += 2 * G * NT) {
        const bool more1 = g0 + G * NT < NG, more2 = g0 + 2 * G * NT < NG;
        if (more1) request(rb);
        consume(ra);
        if (more2) request(ra);
        if (more1) consume(rb);
    }
}

// Histogram pass of a HISTEQ stage that reads the raw pixels, over a tile of whole groups with no group past its end (round 4): the
// bin of a pixel is a non-decreasing function of the pixel (the float64 quotient is, and numpy's one-step correction against the bin
// edges keeps it so), hence bin(r) = number of i in 1..255 with r >= thr[i], thr[i] = the smallest float whose bin is >= i (found once
// per tile by bisection over the float keys with the exact float64 formula).  Per pixel: a float32 guess, a walk along thr (zero or one
// step) and the aggregated increment -- instead of a float64 division, two edge evaluations and their compares.
template <int ROUNDS>
__device__ __forceinline__ void hist_plain(const TileView& tv, const float* thr, float firstf, float invf, unsigned* hist) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int GR = tv.tw >> 2, NG = tv.th * GR;
    int gx, g = (int)threadIdx.x;
    unsigned off;
    { const int y0 = g / GR; gx = g - y0 * GR; off = (unsigned)(y0 * tv.MW + (gx << 2)) * 4u; }
    const int dy = NT / GR, dx = NT - dy * GR;
    const unsigned step_a = (unsigned)(dy * tv.MW + (dx << 2)) * 4u, step_b = step_a + (unsigned)(tv.MW - tv.tw) * 4u;
    constexpr int G = 2;
    auto request = [&](f32x4 (&dst)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            dst[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tv.rs, off, 0, 0));
            g += NT; gx += dx;
            const bool wrap = gx >= GR;
            gx -= wrap ? GR : 0;
            off += wrap ? step_b : step_a;
        }
    };
    auto consume = [&](const f32x4 (&r)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float rf = r[u][e];
                int j = (int)((rf - firstf) * invf);
                j = j < 0 ? 0 : (j > 255 ? 255 : j);
                while (j < 255 && rf >= thr[j + 1]) ++j;
                while (j > 0 && rf < thr[j]) --j;
                hist_add<ROUNDS>(hist, (unsigned)j, true);
            }
    };
    f32x4 ra[G], rb[G];
    request(ra);
    for (int g0 = 0; g0 < NG; g0 += 2 * G * NT) {      // (NG is a multiple of 2 G NT here)
        request(rb);
        consume(ra);
        if (g0 + 2 * G * NT < NG) request(ra);
        consume(rb);
    }
}

// skimage equalize_hist tables: np.histogram(image, 256) over [min,max] (zeros included), cdf, bin centres
__device__ __forceinline__ void histeq_run(Smem& s, const TileView& tv, int upto, double* heq_global) {
    double mn = INFINITY, mx = -INFINITY;
    const bool lean = upto == 0 && (tv.tw & 3) == 0 && (tv.th * (tv.tw >> 2)) % (4 * NT) == 0 && !(s.variant & 4);
    if (lean) {
        float rmn = INFINITY, rmx = -INFINITY;
        minmax_plain<true>(tv, 0.0f, rmn, rmx);
        mn = (double)rmn; mx = (double)rmx;
    } else {
        for_pixels<false>(s, tv, upto, nullptr, [&](float, double v, bool, bool ok) {
            if (ok) { mn = fmin(mn, v); mx = fmax(mx, v); }
        });
    }
    mn = block_min(s, mn); mx = block_max(s, mx);
    double first = mn, last = mx;
    if (first == last) { first -= 0.5; last += 0.5; }
    const double step = (last - first) / 256.0;
    auto edge = [&](int i) { return i == 256 ? last : (double)i * step + first; };       // np.linspace
    __syncthreads();
    clear_hists(s, false);
    __syncthreads();
    auto bin_of = [&](double v) {
        int idx = (int)(((v - first) / (last - first)) * 256.0);
        if (idx == 256) idx = 255;
        if (idx < 0) idx = 0;
        if (idx > 255) idx = 255;
        if (v < edge(idx)) idx -= 1;
        else if (v >= edge(idx + 1) && idx != 255) idx += 1;
        return idx;
    };
    if (lean && mn <= mx && isfinite(mn) && isfinite(mx)) {
        float* thr = reinterpret_cast<float*>(s.zs);                   // thr[1..255]; thr[0] unused
        const int i = threadIdx.x;
        if (i >= 1 && i < 256) {
            // smallest key in [key(min), key(max)] whose bin is >= i; none: +inf (no pixel exceeds the maximum)
            unsigned lo = fkey((float)mn), hi = fkey((float)mx);
            float t = INFINITY;
            if (bin_of((double)fkey_inv(hi)) >= i) {
                while (lo < hi) {
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    if (bin_of((double)fkey_inv(mid)) >= i) hi = mid; else lo = mid + 1;
                }
                t = fkey_inv(hi);
            }
            thr[i] = t;
        }
        __syncthreads();
        const double span = last - first;
        // one aggregation round: the background's bin; what is left goes through ordinary atomics (three rounds: +5 % on the chan3
        // pipeline, none: the same as one)
        hist_plain<1>(tv, thr, (float)first, (float)(256.0 / span), s.histA);
    } else
    for_pixels<false>(s, tv, upto, nullptr, [&](float, double v, bool, bool act) {
        int idx = 0;
        if (act) idx = bin_of(v);
        hist_add(s.histA, (unsigned)idx, act);
    });
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long c = 0ull;
        for (int i = 0; i < 256; ++i) { c += s.histA[i]; s.zs[i] = (double)c; }
        for (int i = 0; i < 256; ++i) {
            s.heq[256 + i] = s.zs[i] / (double)c;
            s.heq[i] = (edge(i) + edge(i + 1)) / 2.0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += NT) heq_global[i] = s.heq[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------- statistics kernel
// The argument block is read through scalar fields and two copies into LDS only (the channel's program, the tile origin):
// handing `a.prog[p]` to the routines above by reference made the compiler keep a private copy of the whole 3.3 KB block in
// scratch memory (ScratchSize 3352 B per lane, ~570 MB of scratch writes per launch) and read every stage from there per pixel.
__global__ __launch_bounds__(NT) void pre_stats_kernel(const PreArgs a) {
    __shared__ Smem s;
    // job-major: the long sigma-clip jobs first, the short HISTEQ one last.  A job is one channel program -- or, when the host found
    // that programs 0 and 1 both open with a sigma-clip stage on the same initial set (a.fuse01; chan3), those two one after the other
    // in ONE workgroup: the second run takes the row sample, the first pass and the initial median from the first (InitCache), and the
    // batch's long jobs fit the CUs in one round instead of two.
    const int b = blockIdx.x % a.B, job = blockIdx.x / a.B;
    const int pfirst = a.fuse01 ? (job == 0 ? 0 : job + 1) : job, pcount = a.fuse01 && job == 0 ? 2 : 1;
    // dynamically indexed members (prog[p].st[k], txy[2b]) are read straight from the kernel-argument segment
    typedef const __attribute__((address_space(4))) char* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    auto kint = [](kptr q) { return *(const __attribute__((address_space(4))) int*)q; };
    auto kdbl = [](kptr q) { return *(const __attribute__((address_space(4))) double*)q; };
    if (threadIdx.x == 0) s.variant = a.variant;
    const int tx0 = kint(ka + offsetof(PreArgs, txy) + (size_t)(2 * b) * 4), ty0 = kint(ka + offsetof(PreArgs, txy) + (size_t)(2 * b + 1) * 4);
    const float* tbase = a.mosaic + (size_t)ty0 * a.MW + tx0;
    // bytes addressable from the tile origin: up to the end of the mosaic (a 16-byte load of the last group of a row may
    // reach into the next row or, on the mosaic's last row, past the end: there the range check returns zeros)
    const size_t tbytes = ((size_t)(a.MH - ty0) * a.MW - tx0) * 4;
    TileView tv{tbase, a.MW, a.tw, a.th, a.tw * a.th,
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tbase), 0, (unsigned)(tbytes > 0xFFFFFF00ull ? 0xFFFFFF00ull : tbytes), 0x00020000)};
    InitCache ic{false, 0, 0.0, 0ull, 0.0, 0.0, 0.0, 0.0};
    int status = 0, hits = 0, misses = 0;
  for (int pi = 0; pi < pcount; ++pi) {
    const int p = pfirst + pi;
    const kptr kprog = ka + offsetof(PreArgs, prog) + (size_t)p * sizeof(PreProgram);
    const int nst = kint(kprog + offsetof(PreProgram, n));
    __syncthreads();
    if (threadIdx.x < MAX_STAGES) {
        const int k = threadIdx.x;
        const kptr ks = kprog + offsetof(PreProgram, st) + (size_t)k * sizeof(PreStage);
        s.op[k] = k < nst ? kint(ks + offsetof(PreStage, op)) : 0;
        s.q0[k] = kdbl(ks + offsetof(PreStage, p0)); s.q1[k] = kdbl(ks + offsetof(PreStage, p1));
        s.flag[k] = kint(ks + offsetof(PreStage, flag));
    }
    double* params = a.params + ((size_t)b * 3 + p) * PSTRIDE;
    double* heq = a.histeq + ((size_t)b * 3 + p) * HEQ_STRIDE;
    __syncthreads();
    for (int k = 0; k < nst; ++k) {
        const int op = s.op[k];
        const double p0 = s.q0[k], p1 = s.q1[k];
        const int flag = s.flag[k];
        double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
        const unsigned long long top = pre_now();
        if (op == OP_BKG || op == OP_SHIFT || op == OP_CLIP) {
            // astropy: `sigma_lower or sigma` -- a 0 falls back to the default sigma = 3
            const double slo = op == OP_CLIP ? (p0 != 0.0 ? p0 : 3.0) : p0, sup = op == OP_CLIP ? (p1 != 0.0 ? p1 : 3.0) : p0;
            const int ubox = op == OP_BKG ? flag : 0;
            const double fract = op == OP_BKG ? p1 : 0.0;
            const ClipStats r = k == 0 ? sigma_clip_run<true>(s, tv, k, slo, sup, ubox, fract, pcount > 1 ? &ic : nullptr) : sigma_clip_run<false>(s, tv, k, slo, sup, ubox, fract);
            hits += r.hits; misses += r.misses;
            if (op == OP_BKG) { o0 = r.mean; if (r.n == 0) status = 1; }
            else if (op == OP_SHIFT) { o0 = r.mean + p0 * r.std; o1 = r.mean; o2 = r.std; if (r.n == 0) status = 1; }
            else { o0 = r.lo; o1 = r.hi; if (r.n == 0 && isnan(r.lo)) status = 1; }
        } else if (op == OP_ZSCALE) {
            zscale_run(s, tv, k, p0, &o0, &o1);
        } else if (op == OP_HISTEQ) {
            histeq_run(s, tv, k, heq);
        } else if (op == OP_MINMAX) {
            // min / max over the pixels whose stage input is non-zero and finite.  Every stage map is weakly increasing in its input and a
            // pixel that hits 0 stays 0, so over those pixels the chain's output is a weakly increasing function of the RAW pixel: the
            // extremes are the chain's values at the smallest and the largest such raw pixel.  The pass therefore only finds those two
            // (fp32 min / max behind the non-zero test), and the chain is evaluated twice per tile instead of once per pixel.  A MINMAX
            // stage with a reversed range inside the chain (decreasing map) takes the general pass.
            bool increasing = true;
            for (int j = 0; j < k; ++j) if (s.op[j] == OP_MINMAX && s.q1[j] < s.q0[j]) increasing = false;
            if (increasing) {
                float rmn = INFINITY, rmx = -INFINITY;
                const NzChain nz = nz_chain(s, k);
                float tf = 0.0f;
                if ((tv.tw & 3) == 0 && !(s.variant & 4) && raw_threshold(s, k, &tf)) {
                    minmax_plain(tv, tf, rmn, rmx);
                } else if (nz.ok) {
                    for_pixels<true>(s, tv, k, nullptr, [&](float rf, double, bool, bool ok) {
                        if (ok && nz_test(nz, rf)) { rmn = fminf(rmn, rf); rmx = fmaxf(rmx, rf); }
                    });
                } else {
                    for_pixels<true>(s, tv, k, nullptr, [&](float rf, double, bool, bool ok) {
                        if (ok && chain_nonzero(s, k, rf)) { rmn = fminf(rmn, rf); rmx = fmaxf(rmx, rf); }
                    });
                }
                const double a = block_min(s, (double)rmn), b = block_max(s, (double)rmx);
                if (a <= b) { o0 = chain_value(s, k, a); o1 = chain_value(s, k, b); }
                else { o0 = INFINITY; o1 = -INFINITY; }
            } else {
                double mn = INFINITY, mx = -INFINITY;
                for_pixels<false>(s, tv, k, nullptr, [&](float, double v, bool, bool ok) {
                    if (ok && cond_of(v)) { mn = fmin(mn, v); mx = fmax(mx, v); }
                });
                o0 = block_min(s, mn); o1 = block_max(s, mx);
            }
            o2 = p0; o3 = p1;
            if (!(o0 <= o1)) status = 1;           // no non-zero finite pixel: the stage returns None (preprocessing.py:101-103)
        }
        pre_acc(op == OP_ZSCALE ? 4 : (op == OP_HISTEQ ? 5 : (op == OP_MINMAX ? 6 : 3)), top);
        __syncthreads();
        if (threadIdx.x == 0) {
            params[k * 4] = o0; params[k * 4 + 1] = o1; params[k * 4 + 2] = o2; params[k * 4 + 3] = o3;
            s.par[k * 4] = o0; s.par[k * 4 + 1] = o1; s.par[k * 4 + 2] = o2; s.par[k * 4 + 3] = o3;
        }
        __syncthreads();
    }
  }
    if (threadIdx.x == 0 && status) atomicMax(a.status + b, status);
    if (threadIdx.x == 0 && a.counters && (hits | misses)) { atomicAdd(a.counters + 2, hits); atomicAdd(a.counters + 3, misses); }
}

// ---------------------------------------------------------------------------------------- apply + letterbox + pack
__device__ __forceinline__ double chain_value(const PreProgram& pg, int upto, const double* params, const double* heq, double raw) {
    double v = raw;
    for (int k = 0; k < upto; ++k) v = apply_stage(pg.st[k].op, pg.st[k].p0, pg.st[k].p1, params + k * 4, heq, v);
    return v;
}

__device__ __forceinline__ void tile_channels(const PreArgs& a, int b, double raw, double out[3]) {
    if (a.nprog == 0) { out[0] = out[1] = out[2] = raw; return; }
    for (int c = 0; c < 3; ++c) {
        const int p = a.nprog == 1 ? 0 : c;
        if (a.nprog == 1 && c > 0) { out[c] = out[0]; continue; }
        out[c] = chain_value(a.prog[p], a.prog[p].n, a.params + ((size_t)b * 3 + p) * PSTRIDE,
                             a.histeq + ((size_t)b * 3 + p) * HEQ_STRIDE, raw);
    }
}

// The tile's channel programs and solved parameters, staged in LDS once per workgroup (round 4).  Reading them per pixel and stage
// through the argument block and the parameter arrays (loads the compiler cannot hoist over the output stores) and finding the
// HISTEQ knot by bisection in global memory made the chan3 pipeline's pack 1.32 ms per 225 tiles of 640^2 against 0.34 ms for a
// plain copy: ~470 instructions per pixel.
// Quotients by a per-tile constant.  A float64 division compiles to (ISA of `a / b` on gfx950)
//   d = div_scale(b), n = div_scale(a), r = rcp(d), two Newton steps r <- fma(r, fma(-d, r, 1), r), q0 = n * r, e = fma(-d, q0, n),
//   q = div_fmas(e, r, q0), div_fixup(q, b, a):
// eleven instructions, of which everything up to r depends on b only.  div_scale is the identity -- and div_fmas a plain fma, div_fixup the
// identity -- unless an operand is zero / denormal / non-finite or the exponents are extreme (ISA manual, V_DIV_SCALE_F64); in that
// regime the quotient is  fma(fma(-b, a * r, a), r, a * r)  with r from the same rcp + two steps: the SAME instruction sequence on the same
// values, hence the same bits as `a / b`.  fast_div takes that path for |b| in [2^-500, 2^500] and a == 0 or |a| in [2^-400, 2^400], and
// divides otherwise.  tests/test_gpu_preproc.py::test_fast_division_is_the_division compares 2^22 pairs bit for bit (cy_debug_fastdiv).
struct FastDiv { double b, nb, r; int ok; };
__device__ __forceinline__ unsigned hi_abs(double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32) & 0x7FFFFFFFu; }
__device__ __forceinline__ FastDiv fast_div_setup(double b) {
    FastDiv f; f.b = b; f.nb = -b;
    const unsigned e = hi_abs(b) >> 20;                             // biased exponent
    f.ok = e >= 1023u - 500u && e <= 1023u + 500u;
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    f.r = r;
    return f;
}
__device__ __forceinline__ double fast_div(double a, double b, double nb, double r, int ok) {
    const unsigned e = hi_abs(a) >> 20;
    const bool safe = ok && ((e >= 1023u - 400u && e <= 1023u + 400u) || a == 0.0);
    if (__builtin_expect(!safe, 0)) return a / b;
    const double q0 = a * r;
    return __builtin_fma(__builtin_fma(nb, q0, a), r, q0);
}
__global__ void fastdiv_probe_kernel(const double* a, const double* b, double* fast, double* ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FastDiv f = fast_div_setup(b[i]);
    fast[i] = fast_div(a[i], f.b, f.nb, f.r, f.ok);
    ref[i] = a[i] / b[i];
}
void debug_fastdiv(const double* d_a, const double* d_b, double* d_fast, double* d_ref, int n) {
    hipLaunchKernelGGL(fastdiv_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_a, d_b, d_fast, d_ref, n);
    hipDeviceSynchronize();
}

struct PackChain {
    int n[3]; int op[3][MAX_STAGES];
    double q0[3][MAX_STAGES], q1[3][MAX_STAGES], par[3][MAX_STAGES * 4];
    double db[3][MAX_STAGES], dr[3][MAX_STAGES]; int dok[3][MAX_STAGES];      // FastDiv of the stage's divisor (ZSCALE: vmax - vmin, MINMAX: max - min)
    float hinv[3];
    double heq[3][512];
    double hslope[3][256];                                                     // HISTEQ: slope between knots j and j + 1 (divided once per tile)
};
// apply_stage with the divisions of the per-tile constants taken from PackChain: same values, bit for bit
template <int OP = -1>
__device__ __forceinline__ double apply_stage_pack(const PackChain& pc, int p, int k, double v) {
    const int op = OP >= 0 ? OP : pc.op[p][k];
    const double* sp = pc.par[p] + k * 4;
    const bool c = cond_of(v);
    double o = v;
    switch (op) {
        case OP_BKG: o = v - sp[0]; break;
        case OP_SHIFT: o = v - sp[0]; if (o < 0.0) o = 0.0; break;
        case OP_CLIP: if (o < sp[0]) o = sp[0]; if (o > sp[1]) o = sp[1]; break;
        case OP_ZSCALE: {
            o = v - sp[0];
            const double rng = pc.db[p][k];
            if (rng != 0.0) o = fast_div(o, rng, -rng, pc.dr[p][k], pc.dok[p][k]);
            o = fmin(fmax(o, 0.0), 1.0);
            break;
        }
        case OP_HISTEQ: {
            const double* xp = pc.heq[p]; const double* fp = pc.heq[p] + 256;
            if (v > xp[255]) { o = fp[255]; break; }
            if (v < xp[0]) { o = fp[0]; break; }
            int j = (int)((float)(v - xp[0]) * pc.hinv[p]);
            j = j < 0 ? 0 : (j > 255 ? 255 : j);
            while (j < 255 && xp[j + 1] <= v) ++j;
            while (j > 0 && xp[j] > v) --j;
            if (j == 255) { o = fp[255]; break; }
            if (xp[j] == v) { o = fp[j]; break; }
            o = pc.hslope[p][j] * (v - xp[j]) + fp[j];
            break;
        }
        case OP_MINMAX: {
            const double d = pc.db[p][k];
            o = fast_div(v - sp[0], d, -d, pc.dr[p][k], pc.dok[p][k]) * (pc.q1[p][k] - pc.q0[p][k]) + pc.q0[p][k];
            break;
        }
        default: break;
    }
    return c ? o : 0.0;
}

// SIG: the shape of the channel programs, when it is one the host recognises -- 1: one program [ZSCALE, MINMAX] (the reference's default
// pipeline, the headline's).  The stages are then applied without the per-stage switch and their parameters are loop-invariant LDS reads
// the compiler hoists: the same operations on the same values as the general form (0); zscale + minmax 0.58 -> 0.46 ms per 225 tiles of 640^2.
template <typename T, int SIG = 0>
__global__ __launch_bounds__(256) void pre_pack_kernel(const PreArgs a) {
    typedef T vec4 __attribute__((ext_vector_type(4)));
    __shared__ PackChain pc;
    const int b = blockIdx.y, np = a.nprog;
    // (dynamically indexed members of the argument block are read through the kernel-argument segment: see pre_stats_kernel)
    typedef const __attribute__((address_space(4))) char* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    auto kint = [](kptr q) { return *(const __attribute__((address_space(4))) int*)q; };
    auto kdbl = [](kptr q) { return *(const __attribute__((address_space(4))) double*)q; };
    for (int t = threadIdx.x; t < np * MAX_STAGES; t += 256) {
        const int p = t / MAX_STAGES, k = t - p * MAX_STAGES;
        const kptr kprog = ka + offsetof(PreArgs, prog) + (size_t)p * sizeof(PreProgram);
        const kptr ks = kprog + offsetof(PreProgram, st) + (size_t)k * sizeof(PreStage);
        const int n = kint(kprog + offsetof(PreProgram, n));
        if (k == 0) pc.n[p] = n;
        pc.op[p][k] = k < n ? kint(ks + offsetof(PreStage, op)) : 0;
        pc.q0[p][k] = kdbl(ks + offsetof(PreStage, p0)); pc.q1[p][k] = kdbl(ks + offsetof(PreStage, p1));
        const double* sp = a.params + ((size_t)b * 3 + p) * PSTRIDE + k * 4;
        for (int j = 0; j < 4; ++j) pc.par[p][k * 4 + j] = sp[j];
        const FastDiv fd = fast_div_setup(sp[1] - sp[0]);               // (both stages with a division divide by sp[1] - sp[0])
        pc.db[p][k] = fd.b; pc.dr[p][k] = fd.r; pc.dok[p][k] = fd.ok;
    }
    __syncthreads();
    for (int p = 0; p < np; ++p) {
        bool has = false;
        for (int k = 0; k < pc.n[p]; ++k) has = has || pc.op[p][k] == OP_HISTEQ;
        if (has) {                                                     // (uniform)
            const double* hq = a.histeq + ((size_t)b * 3 + p) * HEQ_STRIDE;
            for (int t = threadIdx.x; t < 512; t += 256) pc.heq[p][t] = hq[t];
            if (threadIdx.x < 255) { const int j = threadIdx.x; pc.hslope[p][j] = (hq[256 + j + 1] - hq[256 + j]) / (hq[j + 1] - hq[j]); }
        }
        if (threadIdx.x == 0) {
            const double* hq = a.histeq + ((size_t)b * 3 + p) * HEQ_STRIDE;
            const double span = has ? hq[255] - hq[0] : 0.0;
            pc.hinv[p] = span > 0.0 && isfinite(span) ? (float)(255.0 / span) : 0.0f;
        }
    }
    __syncthreads();
    const int tx0 = kint(ka + offsetof(PreArgs, txy) + (size_t)(2 * b) * 4), ty0 = kint(ka + offsetof(PreArgs, txy) + (size_t)(2 * b + 1) * 4);
    const float* base = a.mosaic + (size_t)ty0 * a.MW + tx0;
    const int npx = a.H * a.W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int Y = i / a.W, X = i - Y * a.W;
        const int y = Y - a.top, x = X - a.left;
        float v[3] = {114.0f / 255.0f, 114.0f / 255.0f, 114.0f / 255.0f};
        if ((unsigned)y < (unsigned)a.th && (unsigned)x < (unsigned)a.tw) {
            const double raw = (double)base[(size_t)y * a.MW + x];
            double ch[3] = {raw, raw, raw};
            if constexpr (SIG == 1) {
                ch[0] = ch[1] = ch[2] = apply_stage_pack<OP_MINMAX>(pc, 0, 1, apply_stage_pack<OP_ZSCALE>(pc, 0, 0, raw));
            } else {
            for (int c = 0; c < 3 && np > 0; ++c) {
                if (np == 1 && c > 0) { ch[c] = ch[0]; continue; }
                double w = raw;
                for (int k = 0; k < pc.n[c]; ++k) w = apply_stage_pack(pc, c, k, w);
                ch[c] = w;
            }
            }
            for (int c = 0; c < 3; ++c) v[c] = (float)ch[c] / 255.0f;
        }
        // network channel c = image channel 2-c (ultralytics treats the array as BGR and flips it)
        vec4 o = {(T)v[2], (T)v[1], (T)v[0], (T)0.0f};
        reinterpret_cast<vec4*>(a.out)[(size_t)b * npx + i] = o;
    }
}

// resize path: first the preprocessed tile in float64 planes, then cv2-style bilinear (float32 coordinates/weights)
__global__ __launch_bounds__(256) void pre_plane_kernel(const PreArgs a) {
    const int b = blockIdx.y;
    const float* base = a.mosaic + (size_t)a.txy[2 * b + 1] * a.MW + a.txy[2 * b];
    const int npx = a.th * a.tw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int y = i / a.tw, x = i - y * a.tw;
        double ch[3];
        tile_channels(a, b, (double)base[(size_t)y * a.MW + x], ch);
        for (int c = 0; c < 3; ++c) a.scratch[((size_t)b * 3 + c) * npx + i] = ch[c];
    }
}

__device__ __forceinline__ void lin_axis(int d, int nsrc, int ndst, int* i0, int* i1, double* w) {
    const double scale = (double)nsrc / (double)ndst;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s0 = (int)floorf(f);
    float al = f - (float)s0;
    if (s0 < 0) { s0 = 0; al = 0.0f; }
    if (s0 >= nsrc - 1) { s0 = nsrc - 1; al = 0.0f; }
    *i0 = s0; *i1 = s0 + 1 < nsrc ? s0 + 1 : nsrc - 1; *w = (double)al;
}

template <typename T>
__global__ __launch_bounds__(256) void pre_resize_pack_kernel(const PreArgs a) {
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y;
    const int npx = a.H * a.W, spx = a.th * a.tw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int Y = i / a.W, X = i - Y * a.W;
        const int y = Y - a.top, x = X - a.left;
        float v[3] = {114.0f / 255.0f, 114.0f / 255.0f, 114.0f / 255.0f};
        if ((unsigned)y < (unsigned)a.new_h && (unsigned)x < (unsigned)a.new_w) {
            int y0, y1, x0, x1; double wy, wx;
            lin_axis(y, a.th, a.new_h, &y0, &y1, &wy);
            lin_axis(x, a.tw, a.new_w, &x0, &x1, &wx);
            for (int c = 0; c < 3; ++c) {
                const double* pl = a.scratch + ((size_t)b * 3 + c) * spx;
                const double r0 = pl[y0 * a.tw + x0] * (1.0 - wx) + pl[y0 * a.tw + x1] * wx;
                const double r1 = pl[y1 * a.tw + x0] * (1.0 - wx) + pl[y1 * a.tw + x1] * wx;
                v[c] = (float)(r0 * (1.0 - wy) + r1 * wy) / 255.0f;
            }
        }
        vec4 o = {(T)v[2], (T)v[1], (T)v[0], (T)0.0f};
        reinterpret_cast<vec4*>(a.out)[(size_t)b * npx + i] = o;
    }
}

// Analyzer.predict's "channel has constant value" test, which indexes ROWS 0..2 (evaluation.py:171-176)
__global__ __launch_bounds__(256) void pre_rowcheck_kernel(const PreArgs a) {
    __shared__ double smn[4], smx[4];
    const int b = blockIdx.x;
    const float* base = a.mosaic + (size_t)a.txy[2 * b + 1] * a.MW + a.txy[2 * b];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int bad = 0;
    for (int row = 0; row < 3 && row < a.th; ++row) {
        double mn = INFINITY, mx = -INFINITY;
        for (int x = threadIdx.x; x < a.tw; x += 256) {
            double ch[3];
            tile_channels(a, b, (double)base[(size_t)row * a.MW + x], ch);
            for (int c = 0; c < 3; ++c) { mn = fmin(mn, ch[c]); mx = fmax(mx, ch[c]); }
        }
        mn = wave_min(mn); mx = wave_max(mx);
        __syncthreads();
        if (lane == 0) { smn[w] = mn; smx[w] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k) { smn[0] = fmin(smn[0], smn[k]); smx[0] = fmax(smx[0], smx[k]); }
            if (smn[0] == smx[0]) bad = 1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && bad && a.status[b] == 0) a.status[b] = 2;
}

// program shapes the pack kernel has a specialised form for (0: none)
static int pack_signature(const PreArgs& a) {
    auto is = [&](int p, std::initializer_list<int> ops) {
        if (a.prog[p].n != (int)ops.size()) return false;
        int k = 0;
        for (int op : ops) if (a.prog[p].st[k++].op != op) return false;
        return true;
    };
    if (a.nprog == 1 && is(0, {OP_ZSCALE, OP_MINMAX})) return 1;
    // (2 = chan3 + minmax was built and measured: 2.94 vs 2.90 ms per 225 tiles for the chan3 preprocessing -- three chains' worth of hoisted
    // parameters cost more registers than the switch cost instructions; the shape takes the general form)
    return 0;
}
static int pre_variant_env() { const char* e = getenv("CY_PRE_VARIANT"); return e ? atoi(e) : 0; }
// programs 0 and 1 open with a sigma-clip stage over the same initial set (same box): one workgroup runs both (pre_stats_kernel)
static int pre_fuse01(const PreArgs& a) {
    // Opt-in (CY_PRE_VARIANT bit 4096): alone the chan3 statistics take 2.75 instead of 2.91 ms per 225 tiles of 640^2, but inside the
    // pipelined pass of config 5 the rate is the same or a hair lower (4991 / 4991 vs 4993 / 5027 tiles/s on one box): the saved CU time is
    // paid back by workgroups that hold their CU twice as long.  Bit-identical statistics either way (tests/test_gpu_preproc.py).
    if (a.nprog != 3 || a.prog[0].n < 1 || a.prog[1].n < 1 || !(pre_variant_env() & 4096)) return 0;
    const PreStage &x = a.prog[0].st[0], &y = a.prog[1].st[0];
    auto clip_kind = [](int op) { return op == OP_BKG || op == OP_SHIFT || op == OP_CLIP; };
    if (!clip_kind(x.op) || !clip_kind(y.op)) return 0;
    const int bx = x.op == OP_BKG ? x.flag : 0, by = y.op == OP_BKG ? y.flag : 0;
    const double fx = x.op == OP_BKG ? x.p1 : 0.0, fy = y.op == OP_BKG ? y.p1 : 0.0;
    return bx == by && fx == fy;
}
static int pre_variant() { const char* e = getenv("CY_PRE_VARIANT"); return e ? atoi(e) : 0; }      // (read per launch: tests/test_gpu_preproc.py switches it)

hipError_t launch_preproc(const PreArgs& a0, hipStream_t s) {
    PreArgs a = a0; a.variant = pre_variant(); a.fuse01 = pre_fuse01(a);
    hipError_t e = hipMemsetAsync(a.status, 0, a.B * sizeof(int), s);
    if (e != hipSuccess) return e;
    if (a.nprog > 0) hipLaunchKernelGGL(pre_stats_kernel, dim3(a.B * (a.nprog - a.fuse01)), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(pre_rowcheck_kernel, dim3(a.B), dim3(256), 0, s, a);
    const int npx = a.H * a.W;
    int gx = (npx + 255) / 256; if (gx > 1024) gx = 1024;
    if (a.scratch) {
        int gs = (a.th * a.tw + 255) / 256; if (gs > 1024) gs = 1024;
        hipLaunchKernelGGL(pre_plane_kernel, dim3(gs, a.B), dim3(256), 0, s, a);
        if (a.out_prec == PREC_F16) hipLaunchKernelGGL(pre_resize_pack_kernel<_Float16>, dim3(gx, a.B), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(pre_resize_pack_kernel<float>, dim3(gx, a.B), dim3(256), 0, s, a);
    } else {
        int gp = (npx + 4095) / 4096; if (gp > 1024) gp = 1024;      // 16 pixels per thread: the chain is staged in LDS once per workgroup
        const int sig = (a.variant & 8192) ? 0 : pack_signature(a);
        if (a.out_prec == PREC_F16) {
            if (sig == 1) hipLaunchKernelGGL((pre_pack_kernel<_Float16, 1>), dim3(gp, a.B), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((pre_pack_kernel<_Float16, 0>), dim3(gp, a.B), dim3(256), 0, s, a);
        } else {
            if (sig == 1) hipLaunchKernelGGL((pre_pack_kernel<float, 1>), dim3(gp, a.B), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((pre_pack_kernel<float, 0>), dim3(gp, a.B), dim3(256), 0, s, a);
        }
    }
    return hipGetLastError();
}

// statistics + rejection checks + the preprocessed image itself as float64 planes [B][3][th*tw] in a.scratch (what
// DataPreprocessor returns to Analyzer.predict, caesar_yolo/evaluation.py:157-161): parity witness and the plotting input
hipError_t launch_preproc_planes(const PreArgs& a0, hipStream_t s) {
    PreArgs a = a0; a.variant = pre_variant(); a.fuse01 = pre_fuse01(a);
    hipError_t e = hipMemsetAsync(a.status, 0, a.B * sizeof(int), s);
    if (e != hipSuccess) return e;
    if (a.nprog > 0) hipLaunchKernelGGL(pre_stats_kernel, dim3(a.B * (a.nprog - a.fuse01)), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(pre_rowcheck_kernel, dim3(a.B), dim3(256), 0, s, a);
    int gs = (a.th * a.tw + 255) / 256; if (gs > 1024) gs = 1024;
    hipLaunchKernelGGL(pre_plane_kernel, dim3(gs, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_letterbox_pack(const PreArgs& a, hipStream_t s) {
    const int npx = a.H * a.W;
    int gx = (npx + 255) / 256; if (gx > 1024) gx = 1024;
    if (a.out_prec == PREC_F16) hipLaunchKernelGGL(pre_resize_pack_kernel<_Float16>, dim3(gx, a.B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pre_resize_pack_kernel<float>, dim3(gx, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- mosaic ingest
__global__ __launch_bounds__(256) void mosaic_prepare_kernel(float* d, size_t n, int big_endian) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned u = reinterpret_cast<unsigned*>(d)[i];
        if (big_endian) u = __builtin_bswap32(u);
        float f = __uint_as_float(u);
        if (!isfinite(f)) f = 0.0f;
        d[i] = f;
    }
}

hipError_t launch_mosaic_prepare(float* data, size_t n, int big_endian, hipStream_t s) {
    size_t g = (n + 255) / 256; if (g > 16384) g = 16384;
    hipLaunchKernelGGL(mosaic_prepare_kernel, dim3((unsigned)g), dim3(256), 0, s, data, n, big_endian);
    return hipGetLastError();
}

}  // namespace cy
