// gather_ceiling — how many scattered 128-byte lines per second does one MI355X deliver, and what do
// FETCH_SIZE / TCC counters report per line? The traversal kernels read one 128-B node line (7 x 16 B) or
// one 64-B leaf line per step at data-dependent addresses; this tool measures that access pattern without
// the traversal around it, as the calibration MI355X_MICROARCH.md asks for ("calibrate on a known byte
// count in your own access pattern").
//
//   ./gather_ceiling [table_GiB=4] [loads_per_line=7] [iters=64]      (loads_per_line = 17: 7 loads, the first one waited for)
// Every lane reads `loads_per_line` float4 from a pseudo-random line of the table per iteration, 4 lines in
// flight per lane; lines are drawn by a multiplicative hash, so a 4 GiB table (32 Mi lines) is far beyond the
// 4 MiB L2s and the 256 MiB Infinity Cache. Prints lines/s and the implied bytes/s at 64 and 128 B per line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LOADS, bool STAGED>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, uint32_t lineMask, int iters, float* __restrict__ sink) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t h = tid * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        uint32_t line[4];
        for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; line[k] = (h >> 7) & lineMask; }
        float4 v[4][LOADS];
        if (STAGED) {
            // first one load per line, wait for it, then the rest of the line (now L1 hits)
            for (int k = 0; k < 4; ++k) v[k][0] = table[(size_t)line[k] * 8];
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(v[k][0].x));
            for (int k = 0; k < 4; ++k)
                for (int j = 1; j < LOADS; ++j) v[k][j] = table[(size_t)line[k] * 8 + j];
        } else {
            for (int k = 0; k < 4; ++k)
                for (int j = 0; j < LOADS; ++j) v[k][j] = table[(size_t)line[k] * 8 + j];
        }
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < LOADS; ++j) acc += v[k][j].x + v[k][j].w;
    }
    if (acc == 123.456f) sink[tid] = acc;       // never true: keeps the loads alive
}

__global__ void k_fill(float4* t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) t[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const int loads = argc > 2 ? atoi(argv[2]) : 7;
    const int iters = argc > 3 ? atoi(argv[3]) : 64;
    size_t lines = 1; while ((double)(lines * 2) * 128.0 <= gib * 1073741824.0) lines *= 2;
    float4* table = nullptr; float* sink = nullptr;
    CHECK(hipMalloc((void**)&table, lines * 128));
    CHECK(hipMalloc((void**)&sink, 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, lines * 8);
    CHECK(hipDeviceSynchronize());
    const int blocks = 256 * 8;
    hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(t0, 0));
        switch (loads) {
        case 1: hipLaunchKernelGGL((k_gather<1, false>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink); break;
        case 3: hipLaunchKernelGGL((k_gather<3, false>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink); break;
        case 4: hipLaunchKernelGGL((k_gather<4, false>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink); break;
        case 17: hipLaunchKernelGGL((k_gather<7, true>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink); break;
        default: hipLaunchKernelGGL((k_gather<7, false>), dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(lines - 1), iters, sink); break;
        }
        CHECK(hipEventRecord(t1, 0));
        CHECK(hipEventSynchronize(t1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, t0, t1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double nLines = (double)blocks * 256.0 * iters * 4.0;
    printf("table %.1f GiB (%zu lines), %d x 16 B per line, %.0f M lines per launch, %.3f ms -> %.2f G lines/s = %.2f TB/s at 128 B/line, %.2f TB/s at 64 B/line\n",
           lines * 128.0 / 1073741824.0, lines, loads == 1 || loads == 3 || loads == 4 ? loads : 7, nLines / 1e6, best, nLines / best / 1e6,
           nLines * 128.0 / best / 1e9, nLines * 64.0 / best / 1e9);
    return 0;
}
