// synth.cpp — seeded synthetic geometry for the benchmark configurations of
// BASELINE.json (1 M / 10 M / 100 M triangles). Fills HostScene::objects
// directly with the arithmetic of MeshSerializer::BuildTriangles
// (src/MeshSerializer.cpp:96-137) instead of going through a multi-GB OBJ text
// file. Randomness: std::mt19937 raw 32-bit outputs mapped
// to floats by explicit arithmetic only.
#include "host_scene.h"

#include <cstdio>
#include <map>
#include <random>
#include <string>
#include <thread>

namespace rtxh {

void add_mesh(HostScene& s, const float* vertices, const float* normals, uint64_t nv,
              const uint32_t* indices, const int32_t* materialIDs, uint64_t ntri, int materialID0) {
    (void)nv;
    const RtxVec4 offset = v4(s.width / 2.0f, s.height / 2.0f, s.depth / 2.0f);
    const float scale = 100.0f * s.camera.aspectRatio;
    const size_t base = s.objects.size();
    s.objects.resize(base + ntri);
    auto work = [&](uint64_t b, uint64_t e) {
        RtxObject t = blank_object();
        t.type = RTX_TRIANGLE;
        for (uint64_t f = b; f < e; ++f) {
            for (int k = 0; k < 3; ++k) {
                const uint32_t vi = indices[3 * f + k];
                t.vertices[k] = v4(vertices[3 * (size_t)vi], vertices[3 * (size_t)vi + 1], vertices[3 * (size_t)vi + 2]) * scale + offset;
                t.normals[k] = v4(normals[3 * (size_t)vi], normals[3 * (size_t)vi + 1], normals[3 * (size_t)vi + 2]);
            }
            t.position = div_scalar(t.vertices[0] + t.vertices[1] + t.vertices[2], 3.0f);
            t.materialID = materialIDs ? materialIDs[f] : materialID0;
            s.objects[base + f] = t;
        }
    };
    const unsigned nt = ntri > (1u << 16) ? 8u : 1u;
    std::vector<std::thread> ts;
    for (unsigned i = 0; i < nt; ++i) ts.emplace_back(work, ntri * i / nt, ntri * (i + 1) / nt);
    for (auto& t : ts) t.join();
}

static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

namespace {
struct HeightField {
    std::vector<float> verts, norms;      // OBJ space, 3 floats per vertex
    std::vector<uint32_t> idx;            // 3 per triangle
    std::vector<int32_t> mid;             // material per triangle
    uint64_t nv = 0;
};
}

static HeightField gen_heightfield(uint64_t ntri, uint32_t seed, float extentU, float extentV, float amplitude,
                                   float tiltDeg, float cx, float cy, float cz, const int32_t* mats, int32_t nmats) {
    HeightField out;
    if (ntri == 0) return out;
    // cells = ceil(ntri/2); pick the factor pair of cells closest to extentU:extentV
    const uint64_t cells = (ntri + 1) / 2;
    const double want = std::sqrt((double)cells * (double)extentU / (double)extentV);
    uint64_t gu = 1, best = 1; double bestErr = 1e300;
    for (uint64_t d = 1; d * d <= cells; ++d) {
        if (cells % d) continue;
        for (uint64_t c : {d, cells / d}) { double err = std::fabs(std::log((double)c / want)); if (err < bestErr) { bestErr = err; best = c; } }
    }
    gu = best;
    const uint64_t gv = cells / gu;

    std::mt19937 rng(seed);
    auto unitf = [&]() { return (float)(rng() >> 8) * (1.0f / 16777216.0f); };
    const int kWaves = 6;
    float amp[kWaves], fu[kWaves], fv[kWaves], ph[kWaves];
    for (int k = 0; k < kWaves; ++k) {
        const float freq = (1.5f + 2.5f * (float)k) * (0.75f + 0.5f * unitf());
        const float ang = 6.2831853f * unitf();
        fu[k] = freq * cosf(ang) * 6.2831853f / extentU;
        fv[k] = freq * sinf(ang) * 6.2831853f / extentU;
        ph[k] = 6.2831853f * unitf();
        amp[k] = amplitude / (1.0f + (float)k) * (0.5f + unitf());
    }
    const float tilt = tiltDeg * 0.0174532925f;
    const float ct = cosf(tilt), st = sinf(tilt);
    const uint64_t nvu = gu + 1, nvv = gv + 1;
    std::vector<float>& verts = out.verts; std::vector<float>& norms = out.norms;
    verts.resize(3 * nvu * nvv); norms.resize(3 * nvu * nvv);
    const float du = extentU / (float)gu, dv = extentV / (float)gv;
    for (uint64_t j = 0; j < nvv; ++j)
        for (uint64_t i = 0; i < nvu; ++i) {
            // interior vertices are jittered inside their cell so triangles are not all congruent
            uint32_t hsh = mix32((uint32_t)(i * 0x9E3779B1u) ^ mix32((uint32_t)j + seed));
            float ju = (i == 0 || i == gu) ? 0.0f : (((float)(hsh & 0xFFFF) * (1.0f / 65536.0f)) - 0.5f) * 0.6f;
            float jv = (j == 0 || j == gv) ? 0.0f : (((float)(hsh >> 16) * (1.0f / 65536.0f)) - 0.5f) * 0.6f;
            float u = -0.5f * extentU + ((float)i + ju) * du;
            float v = -0.5f * extentV + ((float)j + jv) * dv;
            float h = 0, hu = 0, hv = 0;
            for (int k = 0; k < kWaves; ++k) {
                float a = fu[k] * u + fv[k] * v + ph[k];
                float sn = sinf(a), cs = cosf(a);
                h += amp[k] * sn; hu += amp[k] * cs * fu[k]; hv += amp[k] * cs * fv[k];
            }
            // local frame: u -> +x, v -> +y, height -> -z (towards the default camera), then tilt about x
            float lx = u, ly = v, lz = -h;
            float nx = hu, ny = hv, nz = -1.0f;                 // gradient of (z + h(u,v)) flipped to face -z
            float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
            nx *= inv; ny *= inv; nz *= inv;
            size_t o = 3 * (j * nvu + i);
            verts[o] = cx + lx; verts[o + 1] = cy + ly * ct - lz * st; verts[o + 2] = cz + ly * st + lz * ct;
            norms[o] = nx;      norms[o + 1] = ny * ct - nz * st;      norms[o + 2] = ny * st + nz * ct;
        }
    std::vector<uint32_t>& idx = out.idx; std::vector<int32_t>& mid = out.mid;
    idx.resize(3 * ntri); mid.resize(ntri);
    uint64_t f = 0;
    for (uint64_t j = 0; j < gv && f < ntri; ++j)
        for (uint64_t i = 0; i < gu && f < ntri; ++i) {
            uint32_t a = (uint32_t)(j * nvu + i), b = a + 1, c = (uint32_t)((j + 1) * nvu + i), d = c + 1;
            int32_t m = nmats > 0 ? mats[mix32((uint32_t)(i / 64) * 7919u + (uint32_t)(j / 64) * 104729u + seed) % (uint32_t)nmats] : 0;
            idx[3 * f] = a; idx[3 * f + 1] = c; idx[3 * f + 2] = b; mid[f++] = m;
            if (f < ntri) { idx[3 * f] = b; idx[3 * f + 1] = c; idx[3 * f + 2] = d; mid[f++] = m; }
        }
    out.nv = nvu * nvv;
    return out;
}

void add_heightfield(HostScene& s, uint64_t ntri, uint32_t seed, float extentU, float extentV, float amplitude,
                     float tiltDeg, float cx, float cy, float cz, const int32_t* mats, int32_t nmats) {
    if (ntri == 0) return;
    const HeightField hf = gen_heightfield(ntri, seed, extentU, extentV, amplitude, tiltDeg, cx, cy, cz, mats, nmats);
    add_mesh(s, hf.verts.data(), hf.norms.data(), hf.nv, hf.idx.data(), hf.mid.data(), ntri, 0);
}

// The same mesh as Wavefront OBJ text, the route the reference itself takes (mesh <file>.obj in a .scn,
// src/SceneSerializer.cpp:226-234 -> src/MeshSerializer.cpp:141-243): "v", "vn", "usemtl <name>" whenever the material
// changes, "f a//a b//b c//c". Coordinates are printed with 9 significant digits, which atof + the cast to float
// (src/MeshSerializer.cpp:150-152) map back to the very same float, so load_obj of this file yields byte-identical
// objects to add_heightfield (tests/test_host_scene.py, tests/test_gpu_configs.py: cfg3).
bool write_heightfield_obj(const HostScene& s, const std::string& path, uint64_t ntri, uint32_t seed, float extentU, float extentV,
                           float amplitude, float tiltDeg, float cx, float cy, float cz, const int32_t* mats, int32_t nmats) {
    const HeightField hf = gen_heightfield(ntri, seed, extentU, extentV, amplitude, tiltDeg, cx, cy, cz, mats, nmats);
    std::map<int32_t, std::string> nameOf;
    for (const auto& kv : s.materialNames) nameOf[(int32_t)kv.second] = kv.first;
    FILE* f = fopen(path.c_str(), "w");
    if (!f) { set_error("cannot write " + path); return false; }
    std::vector<char> buf(1 << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    fprintf(f, "# synthetic height field, %llu triangles, seed %u\n", (unsigned long long)ntri, seed);
    for (uint64_t i = 0; i < hf.nv; ++i) fprintf(f, "v %.9g %.9g %.9g\n", hf.verts[3 * i], hf.verts[3 * i + 1], hf.verts[3 * i + 2]);
    for (uint64_t i = 0; i < hf.nv; ++i) fprintf(f, "vn %.9g %.9g %.9g\n", hf.norms[3 * i], hf.norms[3 * i + 1], hf.norms[3 * i + 2]);
    int32_t cur = -1;
    bool ok = true;
    for (uint64_t t = 0; t < ntri; ++t) {
        if (hf.mid[t] != cur) {
            cur = hf.mid[t];
            auto it = nameOf.find(cur);
            if (it == nameOf.end()) { set_error("material " + std::to_string(cur) + " has no name to put behind usemtl"); ok = false; break; }
            fprintf(f, "usemtl %s\n", it->second.c_str());
        }
        const uint32_t a = hf.idx[3 * t] + 1, b = hf.idx[3 * t + 1] + 1, c = hf.idx[3 * t + 2] + 1;
        fprintf(f, "f %u//%u %u//%u %u//%u\n", a, a, b, b, c, c);
    }
    if (fclose(f) != 0) { set_error("short write to " + path); ok = false; }
    return ok;
}

} // namespace rtxh
