// rtx_shade.h — material shading, texture fetch and sky lookup on the device.
// Restates src/Shading.h and src/ThreadedShader.cpp:23-132,160-171 operation by
// operation. Where the reference's unqualified libm calls resolve to the double
// overloads (acos, atan2, cos, sqrt, log2, floor, fabs) the double device
// functions are used and the result is rounded to float at the same point.
#pragma once
#include "rtx_device.h"
#include "rtx_sincosf.h"

struct DevMaterial {
    F4 albedo, tint, specular, transmissionFilter;
    float specularIntensity, transparency, indexOfRefraction, roughness;
    float metallic, sheen, tintRoughness, clearcoatThickness;
    float clearcoatRoughness, emmissionIntensity, anisotropy, anisotropyRotation;
    int textureID;
};

__device__ __forceinline__ DevMaterial load_material(const float4* __restrict__ mats, int id) {
    const float4* m = mats + (size_t)id * 8;
    DevMaterial r;
    r.albedo = from4(m[0]); r.tint = from4(m[1]); r.specular = from4(m[2]); r.transmissionFilter = from4(m[3]);
    float4 a = m[4], b = m[5], c = m[6], d = m[7];
    r.specularIntensity = a.x; r.transparency = a.y; r.indexOfRefraction = a.z; r.roughness = a.w;
    r.metallic = b.x; r.sheen = b.y; r.tintRoughness = b.z; r.clearcoatThickness = b.w;
    r.clearcoatRoughness = c.x; r.emmissionIntensity = c.y; r.anisotropy = c.z; r.anisotropyRotation = c.w;
    r.textureID = __float_as_int(d.x);
    return r;
}

// src/ThreadedShader.cpp:23-35 (macros of src/Random.h:4-5 expanded)
__device__ __forceinline__ F4 random_direction(uint32_t& seed) {
    float latitude = (float)(acos((double)(2.0f * rtx_rand(seed) - 1.0f)) - (double)(RTX_PI_F * 0.5f));
    float longitude = rtx_rand(seed) * 2.0f * RTX_PI_F;
    float cosLatitude = rtx_cosf(latitude);
    return mk4(cosLatitude * rtx_cosf(longitude), cosLatitude * rtx_sinf(longitude), rtx_sinf(latitude));
}

// src/Color.h:95-98
__device__ __forceinline__ F4 lerp_color(F4 a, F4 b, float t) {
    float scale = fmaxf(0.0f, fminf(t, 1.0f));
    return add4(a, mul4(sub4(b, a), scale));
}

namespace shading {

__device__ __forceinline__ float schlick(float value) {          // src/Shading.h:15-18
    float temp = 1.0f - value;
    return temp * temp * temp * temp * temp;
}
__device__ __forceinline__ F4 tint_of(F4 albedo) {                 // :20-24
    float luminance = albedo.x * 0.3f + albedo.y * 0.6f + albedo.z;
    float condition = luminance > 0.0f ? 1.0f : 0.0f;
    return lerp_color(mk4(1, 1, 1, 1), mul4(albedo, 1.0f / luminance), condition);
}
__device__ __forceinline__ F4 sheen(float cosLightHalf, const DevMaterial& m) {   // :26-30
    F4 tint = tint_of(m.albedo);
    F4 sh = lerp_color(mk4(1, 1, 1, 1), tint, m.tintRoughness);
    return mul4(mul4(sh, schlick(cosLightHalf)), m.sheen);
}
__device__ __forceinline__ float diffuse_brdf(float cosView, float cosLight, const DevMaterial& m) {   // :32-41
    float FL = schlick(cosLight);
    float FV = schlick(cosView);
    float R = 0.5f + 2.0f * cosLight * cosLight * m.roughness * m.roughness;
    float retro = R * (FL + FV + FL * FV * (R - 1.0f));
    return 1.0f / RTX_PI_F * ((1.0f - 0.5f * FL) * (1.0f - 0.5f * FV) + retro);
}
__device__ __forceinline__ float ggx_aniso(F4 h, float ax, float ay) {               // :43-54
    float dotHX2 = h.x * h.x;
    float dotHY2 = h.z * h.z;
    double ch = cos((double)h.y);
    float cos2Theta = (float)(ch * ch);
    float ax2 = ax * ax;
    float ay2 = ay * ay;
    float temp = (dotHX2 / ax2 + dotHY2 / ay2 + cos2Theta);
    return 1.0f / RTX_PI_F * 1.0f / (ax * ay * temp * temp);
}
// The reference writes sqrt(x) with a float x, which C++ resolves to the double overload, and stores the result in
// a float. Rounding a correctly rounded double square root to float gives the correctly rounded float square
// root (double rounding is innocuous for sqrt when the wide format has >= 2p + 2 bits: 53 >= 50), so the
// correctly rounded sqrtf (-fhip-fp32-correctly-rounded-divide-sqrt) returns the same bits without the
// double-precision sequence. Only where the double result is NOT used further in double arithmetic.
__device__ __forceinline__ float sqrt_f32_via_f64(float x) { return sqrtf(x); }

__device__ __forceinline__ float smith_g1_bsdf(F4 vec, F4 h, float ax, float ay) {   // :56-72
    float cos2Theta = h.y * h.y;
    float sin2Theta = 1.0f - cos2Theta;
    float tanTheta = sqrt_f32_via_f64(sin2Theta / cos2Theta);
    float cos2Phi = vec.x * vec.x;
    float sin2Phi = 1.0f - cos2Phi;
    float a = sqrt_f32_via_f64(cos2Phi * ax * ax + sin2Phi * ay * ay);
    float a2Tan2Theta = a * a * tanTheta * tanTheta;
    float lambda = (float)(0.5 * (-1.0 + sqrt((double)(1.0f + a2Tan2Theta))));
    return 1.0f / (1.0f + lambda);
}
__device__ __forceinline__ void ax_ay(const DevMaterial& m, float& ax, float& ay) {  // :76-81
    float aspect = sqrt_f32_via_f64(1.0f - 0.9f * m.anisotropy);
    float roughnessSqr = m.roughness * m.roughness;
    ax = fmaxf(0.001f, roughnessSqr / aspect);
    ay = fmaxf(0.001f, roughnessSqr * aspect);
}
__device__ __forceinline__ float gtr(float cosLightHalf, float alpha) {               // :117-126
    if (alpha >= 1.0f) return 1.0f / RTX_PI_F;
    float alphaSqr = alpha * alpha;
    float decAlphaSqr = alphaSqr - 1.0f;
    return (float)((double)(1.0f / RTX_PI_F * decAlphaSqr) /
                   (log2((double)alphaSqr) * (double)(1.0f + decAlphaSqr * cosLightHalf * cosLightHalf)));
}
__device__ __forceinline__ float smith_g1(float cosine, float alpha) {                // :128-131
    float a2 = alpha * alpha;
    return (float)(2.0 / (1.0 + sqrt((double)(a2 + (1.0f - a2) * cosine * cosine))));
}

__device__ __forceinline__ int to_int(float v) {
    return (v != v || v >= 2147483648.0f || v < -2147483648.0f) ? (int)0x80000000 : (int)v;
}
__device__ __forceinline__ uint32_t texel_at(const DevScene& sc, int index) {
    long long i = index;
    if (i < 0) i = 0;
    if (i >= sc.numTexelsPadded) i = sc.numTexelsPadded - 1;
    return sc.texels[i];
}
__device__ __forceinline__ F4 unpack(uint32_t c) {                                     // :152-155
    const float k = 1.0f / 255.0f;
    return mul4(mk4((float)(c & 255u), (float)((c >> 8) & 255u), (float)((c >> 16) & 255u), (float)(c >> 24)), k);
}
// :157-193 — the texture offset travels through a float (ColorSample's `const float& offset`)
__device__ __forceinline__ F4 color_sample(const DevScene& sc, float u, float v, int width, int height, int offsetInt) {
    int offset = (int)(float)offsetInt;
    float tcx = u * (float)(width - 1), tcy = v * (float)(height - 1);
    float tx = floorf(tcx), ty = floorf(tcy);      // (float)floor((double)x) == floorf(x) exactly
    float fx = tcx - tx, fy = tcy - ty;
    int itx = to_int(tx), ity = to_int(ty);
    uint32_t i00 = (uint32_t)offset + (uint32_t)(ity * width) + (uint32_t)itx;
    uint32_t i01 = (uint32_t)offset + (uint32_t)((ity + 1) * width) + (uint32_t)itx;
    F4 c00 = unpack(texel_at(sc, (int)i00));
    F4 c10 = unpack(texel_at(sc, (int)(i00 + 1u)));
    F4 c01 = unpack(texel_at(sc, (int)i01));
    F4 c11 = unpack(texel_at(sc, (int)(i01 + 1u)));
    F4 color = add4(add4(add4(mul4(mul4(c00, (1.0f - fx)), (1.0f - fy)),
                              mul4(mul4(c10, fx), (1.0f - fy))),
                         mul4(mul4(c01, (1.0f - fx)), fy)),
                    mul4(mul4(c11, fx), fy));
    return color;
}

} // namespace shading

// src/ThreadedShader.cpp:160-171
__device__ __forceinline__ F4 sky_texel(const DevScene& sc, F4 dir) {
    int4 info = sc.texInfo[sc.skyTex];
    float u = (float)((atan2((double)dir.x, (double)dir.z) + (double)RTX_PI_F) * 1.0 / (double)RTX_PI_F);
    float v = (float)(acos((double)(-dir.y)) * 1.0 / (double)RTX_PI_F);
    return shading::color_sample(sc, u, v, info.x, info.y, info.z);
}

// The primitive a ray hit, as the shading stage needs it.
struct HitPrim {
    int type;          // RTX_SPHERE / RTX_TRIANGLE
    F4 A, e1, e2;      // triangle: A, B-A, C-A ; sphere: A = centre
    float Bx, By, Cx, Cy;
    F4 n0, n1, n2;
    int materialID;
};

__device__ __forceinline__ HitPrim load_prim(const DevScene& sc, int leafIdx) {
    const float4* S = sc.shade + (size_t)leafIdx * 8;       // one 128-byte line: leaf record copy + normals + material
    float4 q0 = S[0], q1 = S[1], q2 = S[2], q3 = S[3];
    float4 s0 = S[4], s1 = S[5], s2 = S[6];
    HitPrim p;
    p.type = __float_as_int(q2.w);
    p.A = mk4(q0.x, q0.y, q0.z); p.e1 = mk4(q0.w, q1.x, q1.y); p.e2 = mk4(q1.z, q1.w, q2.x);
    p.Bx = q3.x; p.By = q3.y; p.Cx = q3.z; p.Cy = q3.w;
    p.n0 = mk4(s0.x, s0.y, s0.z); p.n1 = mk4(s0.w, s1.x, s1.y); p.n2 = mk4(s1.z, s1.w, s2.x);
    p.materialID = __float_as_int(s2.y);
    return p;
}

// src/ThreadedShader.cpp:400-428
__device__ __forceinline__ F4 hit_normal(const HitPrim& p, F4 point) {
    if (p.type == 0) return normalize4(sub4(point, p.A));
    if (p.type != 4) return mk4(0, 0, 0);
    F4 v0 = p.e1, v1 = p.e2, v2 = sub4(point, p.A);
    float dot00 = dot3(v0, v0), dot01 = dot3(v0, v1), dot02 = dot3(v0, v2);
    float dot11 = dot3(v1, v1), dot12 = dot3(v1, v2);
    float invDenom = 1.0f / (dot00 * dot11 - dot01 * dot01);
    float u = (dot11 * dot02 - dot01 * dot12) * invDenom;
    float v = (dot00 * dot12 - dot01 * dot02) * invDenom;
    float w = 1.0f - u - v;
    return normalize4(add4(add4(mul4(p.n0, w), mul4(p.n1, u)), mul4(p.n2, v)));
}

// src/Shading.h:195-232
__device__ __forceinline__ F4 get_texture_pixel(const DevScene& sc, const HitPrim& p, int4 info, F4 pt, F4 normal) {
    float tu = 0.0f, tv = 0.0f;
    if (p.type == 0) {
        float theta = (float)(atan2((double)normal.z, (double)normal.x) + (double)RTX_PI_F);
        float phi = (float)acos((double)normal.y);
        const float invPi = 1.0f / RTX_PI_F;
        tu = (theta * 1.0f) * invPi;
        tv = (phi * 1.0f) * invPi;
    } else if (p.type == 4) {
        float Ax = p.A.x, Ay = p.A.y;
        float area = ((p.Bx - Ax) * (p.Cy - Ay) - (p.By - Ay) * (p.Cx - Ax)) * 0.5f;
        float areaPBC = ((p.Bx - pt.x) * (p.Cy - pt.y) - (p.By - pt.y) * (p.Cx - pt.x)) * 0.5f;
        float areaPCA = ((p.Cx - pt.x) * (Ay - pt.y) - (p.Cy - pt.y) * (Ax - pt.x)) * 0.5f;
        tu = sse_max(sse_min(areaPBC / area, 1.0f), 0.0f);
        tv = sse_max(sse_min(areaPCA / area, 1.0f), 0.0f);
    }
    return shading::color_sample(sc, tu, tv, info.x, info.y, info.z);
}

// src/ThreadedShader.cpp:82-132. Updates ray (origin, direction) and lightSample;
// returns colorSample. `camPos` is camera.position.
__device__ __forceinline__ F4 compute_color(const DevScene& sc, F4 camPos, F4& rayO, F4& rayD, F4 point, F4 normal,
                                            const HitPrim& prim, F4& lightSample, uint32_t& seed) {
    DevMaterial m = load_material(sc.materials, prim.materialID);
    int4 info = sc.texInfo[m.textureID];

    F4 lightVector = normalize4(mul4(rayD, -1.0f));
    F4 viewVector = normalize4(sub4(camPos, point));
    F4 halfVector = normalize4(add4(lightVector, viewVector));

    // DiffuseReflect :37-43
    F4 rd = random_direction(seed);
    float cosDirection = dot3(normal, rd);
    F4 diffusionDirection = normalize4(add4(mul4(rd, cosDirection), normal));
    // Reflect :45-48
    F4 reflectionDirection = normalize4(sub4(rayD, mul4(normal, (2.0f * dot3(rayD, normal)))));
    // Refract(viewVector, normal, 1.0, ior) :50-65
    F4 refractionDirection;
    {
        float cosI = -dot3(viewVector, normal);
        float sinR2 = (1.0f - cosI * cosI);
        float eta = 1.0f / m.indexOfRefraction;
        if (eta * sinR2 > 1.0f) {
            refractionDirection = mk4(0, 0, 0);
        } else {
            float cosR2 = shading::sqrt_f32_via_f64(1.0f - sinR2 * sinR2);
            refractionDirection = normalize4(add4(mul4(viewVector, eta), mul4(normal, (eta * cosI - cosR2))));
        }
    }
    // Vector3::Lerp clamps metallic / transparency in place (src/Vector3.h:49-52)
    float metallic = clamp01_libm(m.metallic);
    float transparency = clamp01_libm(m.transparency);
    F4 outgoing = lerp4(diffusionDirection, reflectionDirection, metallic);
    rayO = point;
    rayD = normalize4(lerp4(outgoing, refractionDirection, transparency));

    float cosLight = fmaxf(1e-6f, dot3(normal, lightVector));
    float cosView = fmaxf(1e-6f, dot3(normal, viewVector));
    float cosLightHalf = fmaxf(1e-6f, dot3(lightVector, halfVector));

    F4 emission = mul4(m.albedo, m.emmissionIntensity);
    float isEmissive = (emission.x + emission.y + emission.z) > 0.0f ? 1.0f : 0.0f;

    F4 texture = get_texture_pixel(sc, prim, info, point, normal);

    F4 diffuseAlbedo = mul4(mul4(texture, m.tint), (1.0f - metallic));
    F4 specularAlbedo = lerp_color(m.specular, mk4(1, 1, 1, 1), metallic);
    float fresnel = shading::schlick(cosLightHalf);

    float ax, ay; shading::ax_ay(m, ax, ay);
    float D = shading::ggx_aniso(halfVector, ax, ay);
    float Gl = shading::smith_g1_bsdf(lightVector, halfVector, ax, ay);
    float Gv = shading::smith_g1_bsdf(viewVector, halfVector, ax, ay);

    F4 diffuseComponent = mul4(mul4(diffuseAlbedo, (1.0f - fresnel)), shading::diffuse_brdf(cosView, cosLight, m));
    // SpecularBSDF :74-91 (un-clamped dots)
    float specular = D * Gl * Gv / (4.0f * dot3(normal, lightVector) * dot3(normal, viewVector));
    F4 specularComponent = mul4(mul4(specularAlbedo, fresnel), specular);
    // SpecularTransmissionBSDF :93-115
    float transmission;
    {
        float cosViewHalf = dot3(viewVector, halfVector);
        cosViewHalf *= halfVector.y;
        float eta = 1.0f / m.indexOfRefraction;
        float F = eta + (1.0f - eta) * shading::schlick(cosViewHalf);
        transmission = D * F * Gl * Gv;
    }
    // ClearcoatBRDF :133-150
    float clearcoat;
    {
        float cosHalf = fabsf(halfVector.y), cosV = fabsf(viewVector.y), cosL = fabsf(lightVector.y);
        float cosLH = dot3(lightVector, halfVector);
        float scale = 0.1f + (0.001f - 0.1f) * m.clearcoatRoughness;
        float Dc = shading::gtr(cosHalf, scale);
        float Glc = shading::smith_g1(cosL, 0.25f);
        float Gvc = shading::smith_g1(cosV, 0.25f);
        float Fc = 0.04f + 0.96f * shading::schlick(cosLH);
        clearcoat = 0.25f * Dc * Glc * Gvc * Fc;
    }
    F4 sheenComponent = shading::sheen(cosLightHalf, m);

    // CalculateWeights :67-80 — all four lanes divided by the xyz magnitude
    F4 weights;
    {
        float dielectric = (1.0f - metallic) * (1.0f - transparency);
        weights = mk4(metallic + dielectric, (1.0f - metallic) * transparency, dielectric, m.clearcoatThickness);
        weights = normalize4(weights);
    }

    F4 colorSample = mul4(emission, isEmissive);
    colorSample = add4(colorSample, mul4(add4(diffuseComponent, sheenComponent), weights.z));
    colorSample = add4(colorSample, mul4(mk4(clearcoat, clearcoat, clearcoat, clearcoat), weights.w));
    colorSample = add4(colorSample, mul4(specularComponent, weights.x));
    colorSample = add4(colorSample, mul4(mk4(transmission, transmission, transmission, transmission), weights.y));
    colorSample = mul4(mul4(colorSample, lightSample), (cosLight > 0.0f) ? 1.0f : 0.0f);

    lightSample = mul4(mul4(mul4(mul4(lightSample, texture), m.albedo), 2.0f), cosLight);
    return colorSample;
}
