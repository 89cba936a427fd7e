// tcp_rate — at what rate does a CU's vector L1 (TCP) accept 16-byte loads, (a) when every lane of a wave
// reads from a line of its own ("scattered": what a traversal step does, 7 such instructions per node) and
// (b) when 8 neighbouring lanes read the 8 pieces of one 128-byte line ("grouped")? The table is sized to
// stay in the L1 (16 KiB), the L2 (2 MiB) or the Infinity Cache (128 MiB), so the figure is the cache
// front-end's, not HBM's.
//   ./tcp_rate [table_KiB=2048] [mode: 0 scattered, 1 grouped, 2 scattered 7 x same line, 3 scattered 4-byte loads, 4 scattered 8-byte loads, 5 scattered 16-byte buffer loads, 6 = mode 2 with buffer loads] [iters=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(const float4* __restrict__ table, uint32_t lineMask, int iters, float* __restrict__ sink) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t key = MODE == 1 ? (tid >> 3) : tid;          // grouped: 8 lanes share the line sequence
    const uint32_t piece = MODE == 1 ? (tid & 7u) : (tid & 7u);
    uint32_t h = key * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 6) {
            // the traversal's pattern (7 x 16 B of one scattered line per lane) through a buffer resource
            typedef unsigned int f4_t __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, (short)0, (int)0x7fffffff, (int)0x00020000);
            h = h * 1664525u + 1013904223u;
            const uint32_t off = ((h >> 7) & lineMask) * 128u;
            f4_t v[7];
            for (int j = 0; j < 7; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u * j, 0, 0);
            for (int j = 0; j < 7; ++j) acc += __uint_as_float(v[j].x) + __uint_as_float(v[j].w);
        } else if (MODE == 5) {
            // the same scattered 16-byte loads through a buffer resource (MUBUF, offen) instead of global_load
            typedef unsigned int f4_t __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, (short)0, (int)0x7fffffff, (int)0x00020000);
            f4_t v[8];
            for (int k = 0; k < 8; ++k) {
                h = h * 1664525u + 1013904223u;
                const uint32_t off = (((h >> 7) & lineMask) * 8u + piece) * 16u;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            }
            for (int k = 0; k < 8; ++k) acc += __uint_as_float(v[k].x) + __uint_as_float(v[k].w);
        } else if (MODE == 3 || MODE == 4) {
            // narrower loads from a line of the lane's own: 4 B (mode 3) or 8 B (mode 4) per lane-load
            float2 v[8];
            for (int k = 0; k < 8; ++k) {
                h = h * 1664525u + 1013904223u;
                const float* p = (const float*)(table + (size_t)((h >> 7) & lineMask) * 8 + piece);
                if (MODE == 3) { v[k].x = *p; v[k].y = 0.0f; } else v[k] = *(const float2*)p;
            }
            for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y;
        } else if (MODE == 2) {
            h = h * 1664525u + 1013904223u;
            const float4* p = table + (size_t)((h >> 7) & lineMask) * 8;
            float4 v[7];
            for (int j = 0; j < 7; ++j) v[j] = p[j];
            for (int j = 0; j < 7; ++j) acc += v[j].x + v[j].w;
        } else {
            float4 v[8];
            for (int k = 0; k < 8; ++k) {
                h = h * 1664525u + 1013904223u;
                v[k] = table[(size_t)((h >> 7) & lineMask) * 8 + piece];
            }
            for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
        }
    }
    if (acc == 123.456f) sink[tid] = acc;
}

__global__ void k_fill(float4* t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) t[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main(int argc, char** argv) {
    const double kib = argc > 1 ? atof(argv[1]) : 2048.0;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int iters = argc > 3 ? atoi(argv[3]) : 256;
    size_t lines = 1; while ((double)(lines * 2) * 128.0 <= kib * 1024.0) lines *= 2;
    float4* table = nullptr; float* sink = nullptr;
    CHECK(hipMalloc((void**)&table, lines * 128));
    CHECK(hipMalloc((void**)&sink, 4));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, table, lines * 8);
    CHECK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    const int blocks = cus * 8;
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(t0, 0));
        if (mode == 0) hipLaunchKernelGGL((k_rate<0>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink);
        else if (mode == 1) hipLaunchKernelGGL((k_rate<1>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink);
        else if (mode == 6) hipLaunchKernelGGL((k_rate<6>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink);
        else if (mode == 5) hipLaunchKernelGGL((k_rate<5>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink);
        else if (mode == 3) hipLaunchKernelGGL((k_rate<3>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink);
        else if (mode == 4) hipLaunchKernelGGL((k_rate<4>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink);
        else hipLaunchKernelGGL((k_rate<2>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink);
        CHECK(hipEventRecord(t1, 0));
        CHECK(hipEventSynchronize(t1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, t0, t1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double laneLoads = (double)blocks * 256.0 * iters * ((mode == 2 || mode == 6) ? 7.0 : 8.0);
    const double perS = laneLoads / (best * 1e-3);
    const double bytes = mode == 3 ? 4.0 : mode == 4 ? 8.0 : 16.0;
    printf("table %.0f KiB, mode %d (%s): %.1f M lane-loads of %.0f B in %.3f ms -> %.1f G/s = %.2f per clock per CU (%d CUs at %.2f GHz), %.2f TB/s requested\n",
           lines * 128.0 / 1024.0, mode, mode == 0 ? "scattered" : mode == 1 ? "grouped by 8" : mode == 2 ? "scattered, 7 per line" : mode == 3 ? "scattered dword" : mode == 5 ? "scattered, buffer_load" : mode == 6 ? "scattered, 7 per line, buffer_load" : "scattered dwordx2",
           laneLoads / 1e6, bytes, best, perS / 1e9, perS / (cus * ghz * 1e9), cus, ghz, perS * bytes / 1e12);
    return 0;
}
