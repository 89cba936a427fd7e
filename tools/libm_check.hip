// libm_check — do the device's double-precision libm calls, rounded to float where the kernels round them, give the
// bits the host's glibc gives? (The framebuffer's bit-identity with the CPU oracle rests on it: rtx_shade.h calls the
// double device functions where the reference's unqualified acos / atan2 / cos / log2 resolve to the double overloads.)
// For each expression: N random float inputs, evaluated on the device and on the host, mismatching results counted.
// Built with the product's floating-point flags (no contraction, no fast-math).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include "../raytracer_amd/csrc/rtx_sincosf.h"

#define PI_F 3.1415926535f
enum { E_ACOS, E_ATAN2, E_COS2, E_LOG2, E_SQRT1, E_LAT, E_SKYU, E_SKYV, E_SINF, E_COSF, E_GTR, E_SMITH, E_COUNT };
static const char* kNames[E_COUNT] = {"(float)acos(x)", "(float)(atan2(x,y)+pi)", "(float)(cos(x)^2)", "log2 ratio (gtr)", "0.5*(-1+sqrt(1+x))",
                                      "acos(2r-1)-pi/2 (RandomDirection)", "sky u", "sky v", "sinf (restated)", "cosf (restated)", "gtr()", "smith_g1()"};

template <class M>
__host__ __device__ inline float eval(int e, float x, float y, M m) {
    switch (e) {
        case E_ACOS: return (float)m.acos_((double)x);
        case E_ATAN2: return (float)(m.atan2_((double)x, (double)y) + (double)PI_F);
        case E_COS2: { double c = m.cos_((double)x); return (float)(c * c); }
        case E_LOG2: { float a2 = x * x; float dec = a2 - 1.0f; return (float)((double)(1.0f / PI_F * dec) / (m.log2_((double)a2) * (double)(1.0f + dec * y * y))); }
        case E_SQRT1: return (float)(0.5 * (-1.0 + m.sqrt_((double)(1.0f + x))));
        case E_LAT: return (float)(m.acos_((double)(2.0f * x - 1.0f)) - (double)(PI_F * 0.5f));
        case E_SKYU: return (float)((m.atan2_((double)x, (double)y) + (double)PI_F) * 1.0 / (double)PI_F);
        case E_SKYV: return (float)(m.acos_((double)(-x)) * 1.0 / (double)PI_F);
        case E_SINF: return m.sinf_(x);
        case E_COSF: return m.cosf_(x);
        case E_GTR: { if (x >= 1.0f) return 1.0f / PI_F; float a2 = x * x, dec = a2 - 1.0f; return (float)((double)(1.0f / PI_F * dec) / (m.log2_((double)a2) * (double)(1.0f + dec * y * y))); }
        case E_SMITH: { float a2 = x * x; return (float)(2.0 / (1.0 + m.sqrt_((double)(a2 + (1.0f - a2) * y * y)))); }
    }
    return 0.0f;
}
struct DevMath {
    __device__ double acos_(double v) const { return acos(v); }
    __device__ double atan2_(double a, double b) const { return atan2(a, b); }
    __device__ double cos_(double v) const { return cos(v); }
    __device__ double log2_(double v) const { return log2(v); }
    __device__ double sqrt_(double v) const { return sqrt(v); }
    __device__ float sinf_(float v) const { return rtx_sinf(v); }
    __device__ float cosf_(float v) const { return rtx_cosf(v); }
};
struct HostMath {
    double acos_(double v) const { return ::acos(v); }
    double atan2_(double a, double b) const { return ::atan2(a, b); }
    double cos_(double v) const { return ::cos(v); }
    double log2_(double v) const { return ::log2(v); }
    double sqrt_(double v) const { return ::sqrt(v); }
    float sinf_(float v) const { return ::sinf(v); }
    float cosf_(float v) const { return ::cosf(v); }
};
__global__ void k_eval(int e, const float* x, const float* y, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = eval(e, x[i], y[i], DevMath());
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4000000;
    std::mt19937 rng(12345);
    std::vector<float> x(n), y(n), dev(n);
    float *dx, *dy, *dout;
    if (hipMalloc(&dx, n * 4) != hipSuccess || hipMalloc(&dy, n * 4) != hipSuccess || hipMalloc(&dout, n * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
    int bad = 0;
    for (int e = 0; e < E_COUNT; ++e) {
        for (int i = 0; i < n; ++i) {
            const float u = (float)(rng() >> 8) / 16777216.0f, v = (float)(rng() >> 8) / 16777216.0f;     // [0,1)
            switch (e) {
                case E_ACOS: case E_SKYV: x[i] = 2.0f * u - 1.0f; y[i] = v; break;
                case E_ATAN2: case E_SKYU: x[i] = 2.0f * u - 1.0f; y[i] = 2.0f * v - 1.0f; break;
                case E_COS2: x[i] = 2.0f * u - 1.0f; y[i] = v; break;
                case E_LOG2: case E_GTR: x[i] = 0.001f + 0.1f * u; y[i] = v; break;
                case E_SQRT1: x[i] = 100.0f * u * u; y[i] = v; break;
                case E_SINF: case E_COSF: x[i] = (i & 1) ? (6.2831853f * u) : (3.2f * u - 1.6f); y[i] = v; break;
                default: x[i] = u; y[i] = v; break;
            }
        }
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_eval, dim3((n + 255) / 256), dim3(256), 0, 0, e, dx, dy, dout, n);
        if (hipMemcpy(dev.data(), dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
        long mism = 0; int first = -1;
        for (int i = 0; i < n; ++i) {
            const float h = eval(e, x[i], y[i], HostMath());
            uint32_t a, b; memcpy(&a, &h, 4); memcpy(&b, &dev[i], 4);
            if (a != b && !(h != h && dev[i] != dev[i])) { if (first < 0) first = i; ++mism; }
        }
        printf("%-36s %ld of %d differ", kNames[e], mism, n);
        if (first >= 0) printf("   e.g. x=%.9g y=%.9g host=%.9g device=%.9g", x[first], y[first], eval(e, x[first], y[first], HostMath()), dev[first]);
        printf("\n");
        if (mism) bad++;
    }
    printf(bad ? "MISMATCHES in %d expressions\n" : "all %d expressions bit-identical\n", bad ? bad : (int)E_COUNT);
    return 0;
}
