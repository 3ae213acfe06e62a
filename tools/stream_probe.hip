// Developer probe (round 4): what does ONE workgroup per CU get when it streams its own 640 x 640 fp32 tile out of a 16384-wide mosaic
// (the access pattern of a pre_stats_kernel pass), and what changes it?  hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe
// Variants: threads per workgroup, 16-byte loads in flight per lane, tile layout (mosaic rows 64 KB apart / dense copy), workgroups per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int PXG>
__global__ __launch_bounds__(NT) void stream_kernel(const float* __restrict__ base, int MW, int tw, int th, int tiles_x, int tile_step,
                                                     int split, double* out) {
    const int tile = blockIdx.x / split, part = blockIdx.x % split;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const float* tb = tile_step ? base + (size_t)(ty * tile_step) * MW + tx * tile_step : base + (size_t)tile * tw * th;     // 0: dense copies, one after the other
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tb), 0, 0x7FFFFF00u, 0x00020000);
    const int GR = tw >> 2, NG = th * GR;
    const int g_begin = (int)((long)NG * part / split), g_end = (int)((long)NG * (part + 1) / split);
    double s1 = 0.0, s2 = 0.0;
    for (int g0 = g_begin; g0 < g_end; g0 += PXG * NT) {
        f32x4 r[PXG];
#pragma unroll
        for (int u = 0; u < PXG; ++u) {
            const int g = g0 + u * NT + (int)threadIdx.x;
            const int y = g / GR, gx = g - y * GR;
            const unsigned off = g < g_end ? (unsigned)(y * MW + (gx << 2)) * 4u : 0xFFFFFF00u;
            r[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < PXG; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)r[u][e]; s1 += d; s2 += d * d; }
    }
    if (s1 == 12345.678 && s2 == 1.0) out[blockIdx.x] = s1;      // keep the sums alive
}

template <int NT, int PXG>
static float run(const float* d, int MW, int tw, int th, int tiles_x, int ntiles, int step, int split, double* out, int passes) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<NT, PXG>), dim3(ntiles * split), dim3(NT), 0, 0, d, MW, tw, th, tiles_x, step, split, out);
    hipEventRecord(a);
    for (int p = 0; p < passes; ++p)
        hipLaunchKernelGGL((stream_kernel<NT, PXG>), dim3(ntiles * split), dim3(NT), 0, 0, d, MW, tw, th, tiles_x, step, split, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / passes;
}

int main() {
    const int MW = 16384, MH = 16384, tw = 640, th = 640, step = 512, tiles_x = 16, ntiles = 256;
    float* d; double* out;
    hipMalloc(&d, (size_t)MW * MH * 4);
    hipMalloc(&out, 8 * 65536);
    std::vector<float> h((size_t)MW * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) * 1e-6f - 0.03f;
    for (int k = 0; k < 16; ++k) hipMemcpy(d + (size_t)k * h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const double mb = (double)ntiles * tw * th * 4 / 1e6;
    auto report = [&](const char* name, float ms, int nt = 256) {
        const double m = mb * nt / ntiles;
        printf("%-64s %8.1f us per pass  %6.2f TB/s  %6.1f GB/s per tile\n", name, ms * 1e3, m / ms / 1e3, m / nt / ms);
    };
    report("1 WG/tile, 1024 thr, 2 loads in flight (pre_stats today)", run<1024, 2>(d, MW, tw, th, tiles_x, ntiles, step, 1, out, 20));
    report("1 WG/tile, 1024 thr, 4 loads in flight", run<1024, 4>(d, MW, tw, th, tiles_x, ntiles, step, 1, out, 20));
    report("1 WG/tile, 1024 thr, 8 loads in flight", run<1024, 8>(d, MW, tw, th, tiles_x, ntiles, step, 1, out, 20));
    report("1 WG/tile,  512 thr, 4 loads in flight", run<512, 4>(d, MW, tw, th, tiles_x, ntiles, step, 1, out, 20));
    report("1 WG/tile,  256 thr, 8 loads in flight", run<256, 8>(d, MW, tw, th, tiles_x, ntiles, step, 1, out, 20));
    report("2 WG/tile, 1024 thr, 2 loads in flight", run<1024, 2>(d, MW, tw, th, tiles_x, ntiles, step, 2, out, 20));
    report("4 WG/tile, 1024 thr, 2 loads in flight", run<1024, 2>(d, MW, tw, th, tiles_x, ntiles, step, 4, out, 20));
    report("8 WG/tile,  256 thr, 4 loads in flight", run<256, 4>(d, MW, tw, th, tiles_x, ntiles, step, 8, out, 20));
    report("dense tiles (row stride = tile width), 1 WG/tile, 1024 thr, 2", run<1024, 2>(d, tw, tw, th, 1, ntiles, 0, 1, out, 20));
    // 64 tiles only: a quarter of the CUs busy
    report("64 tiles only, 1 WG/tile, 1024 thr, 2 loads in flight", run<1024, 2>(d, MW, tw, th, tiles_x, 64, step, 1, out, 20), 64);
    return 0;
}
