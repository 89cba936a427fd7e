// ingest.cpp — scene ingest for the HIP backend's host side: .scn / .mtl / .obj /
// .bmp readers and the plane / disk / cube tessellators. They fill the same
// arrays, with the same values, as the reference's serializers, so a scene
// loaded here is interchangeable with context->objects etc. in the reference.
// Reference behaviour followed (file:line relative to the reference tree):
//   .scn   src/SceneSerializer.cpp:57-238      .mtl  src/MaterialSerializer.cpp:20-240
//   .obj   src/MeshSerializer.cpp:30-243       .bmp  src/BitmapReader.cpp:69-172
//   shapes src/ObjectBuilder.h:13-184          material defaults src/MaterialBuilder.cpp:10-33
#include "host_scene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rtxh {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const std::string& get_error() { return g_error; }

// Split on single spaces like strtok(" ") (src/Serializer.h:13-34): runs of
// spaces collapse, tabs and '\r' stay inside tokens.
static std::vector<std::string> split_spaces(const std::string& line) {
    std::vector<std::string> out;
    size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && line[i] == ' ') ++i;
        size_t b = i;
        while (i < n && line[i] != ' ') ++i;
        if (i > b) out.emplace_back(line, b, i - b);
    }
    return out;
}

static std::string dir_of(const std::string& path) {
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string() : path.substr(0, p);
}

static float num(const std::string& s) { return (float)atof(s.c_str()); }

RtxObject blank_object() {
    RtxObject o;
    memset(&o, 0, sizeof(o));
    return o;
}

// ---------------------------------------------------------------------------
// Shape tessellation (src/ObjectBuilder.h). Note the tessellated triangles keep
// position = (0,0,0): the BVH median split sorts them by that.
// ---------------------------------------------------------------------------
static const float kPi = 3.1415926535f;   // ObjectBuilder.h:8 redefines M_PI as this float

void add_plane(HostScene& s, RtxVec4 position, const RtxVec4 normals[3], RtxVec4 scales, int materialID) {
    const RtxVec4 reference = v4(1, 0, 0);
    const RtxVec4 U = unit(cross3(normals[0], reference));
    const RtxVec4 W = unit(cross3(normals[0], U));
    const float delta = 2.0f * kPi / 4;
    const float offset = kPi / 4;
    RtxVec4 corner[4];
    for (int i = 0; i < 4; ++i) {
        float angle = delta * i + offset;
        float x = scales.x * cosf(angle);
        float y = scales.y * sinf(angle);
        corner[i] = position + U * x + W * y;
    }
    RtxObject p = blank_object();
    p.type = RTX_TRIANGLE;
    p.normals[0] = normals[0]; p.normals[1] = normals[1]; p.normals[2] = normals[2];
    p.materialID = materialID;
    p.vertices[0] = corner[0]; p.vertices[1] = corner[1]; p.vertices[2] = corner[2];
    s.objects.push_back(p);
    p.vertices[0] = corner[0]; p.vertices[1] = corner[2]; p.vertices[2] = corner[3];
    s.objects.push_back(p);
}

void add_disk(HostScene& s, RtxVec4 position, RtxVec4 normal, float radius, int materialID) {
    const int kFan = 32;                                   // DISK_VERTICES
    const RtxVec4 reference = v4(1, 0, 0);
    const RtxVec4 U = cross3(normal, reference);           // not normalised in the reference either
    const RtxVec4 W = cross3(normal, U);
    std::vector<RtxVec4> rim(kFan + 1, v4(0, 0, 0));
    const float delta = 2.0f * kPi / kFan;
    for (int i = 0; i < kFan; ++i) {
        float x = radius * cosf(delta * i);
        float y = radius * sinf(delta * i);
        rim[i + 1] = U * x + W * y;
    }
    RtxObject p = blank_object();
    p.type = RTX_TRIANGLE;
    p.normals[0] = p.normals[1] = p.normals[2] = normal;
    p.materialID = materialID;
    for (int i = 1; i <= kFan; ++i) {
        p.vertices[0] = position + rim[0];
        p.vertices[1] = position + rim[i];
        p.vertices[2] = position + rim[i % kFan + 1];
        s.objects.push_back(p);
    }
}

void add_cube(HostScene& s, RtxVec4 position, RtxVec4 scales, int materialID) {
    static const float sgn[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}};
    static const int tri[12][3] = {{0, 1, 6}, {0, 6, 7}, {0, 1, 2}, {0, 2, 3}, {1, 2, 5}, {1, 5, 6},
                                   {7, 6, 5}, {7, 5, 4}, {0, 3, 4}, {0, 4, 7}, {3, 2, 5}, {3, 5, 4}};
    RtxVec4 corner[8];
    for (int i = 0; i < 8; ++i) corner[i] = v4(sgn[i][0], sgn[i][1], sgn[i][2]) * scales * 0.5f;
    RtxObject p = blank_object();
    p.type = RTX_TRIANGLE;
    p.materialID = materialID;
    for (int f = 0; f < 12; ++f) {
        RtxVec4 A = corner[tri[f][0]], B = corner[tri[f][1]], C = corner[tri[f][2]];
        p.vertices[0] = position + A; p.vertices[1] = position + B; p.vertices[2] = position + C;
        RtxVec4 n = unit(cross3(B - A, C - A));
        if (dot3(n, unit(A)) < 0.0f) n = n * -1.0f;
        p.normals[0] = p.normals[1] = p.normals[2] = n;
        s.objects.push_back(p);
    }
}

// ---------------------------------------------------------------------------
// .bmp (BITMAPV5HEADER only; rows kept in file order; texel = R|G<<8|B<<16|255<<24)
// ---------------------------------------------------------------------------
static uint32_t rd32(const std::vector<char>& b, size_t off) { uint32_t v; memcpy(&v, b.data() + off, 4); return v; }
static uint16_t rd16(const std::vector<char>& b, size_t off) { uint16_t v; memcpy(&v, b.data() + off, 2); return v; }

// src/BitmapReader.cpp:139-172 with the running value started at 0 (the
// reference starts from an uninitialised member, so its de-duplication is
// accidental; texel values are unaffected either way).
static uint32_t bmp_checksum(const std::vector<uint32_t>& px) {
    uint32_t result = 0, temp = 0;
    for (uint32_t word : px) {
        uint8_t bytes[4]; memcpy(bytes, &word, 4);
        for (int k = 0; k < 4; ++k) {
            uint8_t minDiff = 255, minIdx = 0;
            for (uint8_t i = 0; i < 8; i++) {
                uint8_t d = (uint8_t)((1 << i) & bytes[k]);
                if (d < minDiff) { minDiff = d; minIdx = i; }
            }
            temp ^= bytes[k];
            temp <<= (1 << minIdx);
            temp |= bytes[k];
        }
        result ^= temp;
    }
    return result;
}

static bool read_bmp(const std::string& path, std::vector<uint32_t>& px, int32_t& w, int32_t& h) {
    std::ifstream in(path, std::ios::in | std::ios::binary);
    if (!in) { fprintf(stderr, "File can't be opened.\n"); return false; }
    std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (raw.size() < 14 + 124) return false;
    if (rd32(raw, 14) != 124) {
        fprintf(stderr, "Invalid file format. Only BITMAPV5HEADER are currently handled.\n");
        return false;
    }
    w = (int32_t)rd32(raw, 18); h = (int32_t)rd32(raw, 22);
    uint32_t bytesPerPixel = rd16(raw, 28) / 8;
    if (bytesPerPixel < 3 || bytesPerPixel > 4) { fprintf(stderr, "Invalid pixel format.\n"); return false; }
    if (w <= 0 || h <= 0) return false;
    uint32_t rowSize = ((uint32_t)w * bytesPerPixel + 3) & ~3u;
    size_t off = 14 + 124;                                 // the reader ignores dataOffset
    if (raw.size() < off + (size_t)rowSize * h - (rowSize - (uint32_t)w * bytesPerPixel)) return false;
    px.assign((size_t)w * h, 0u);
    for (int32_t y = 0; y < h; ++y) {
        for (int32_t x = 0; x < w; ++x) {
            const unsigned char* c = (const unsigned char*)raw.data() + off;
            unsigned char out[4] = {c[2], c[1], c[0], 255};
            memcpy(&px[(size_t)y * w + x], out, 4);
            off += bytesPerPixel;
        }
        off += rowSize - (uint32_t)w * bytesPerPixel;
    }
    return true;
}

int add_texture(HostScene& s, const uint32_t* texels, int32_t w, int32_t h, uint32_t checksum) {
    RtxTexture info;
    info.width = w; info.height = h;
    info.offset = (int32_t)s.textureData.size();
    info.checksum = checksum;
    int id = (int)s.textureInfo.size();
    s.textureInfo.push_back(info);
    s.textureData.insert(s.textureData.end(), texels, texels + (size_t)w * h);
    return id;
}

// ---------------------------------------------------------------------------
// .mtl
// ---------------------------------------------------------------------------
static RtxMaterial fresh_material(const HostScene& s) {
    RtxMaterial m; memset(&m, 0, sizeof(m));
    m.albedo = v4(0.5f, 0.5f, 0.5f, 1.0f);
    m.indexOfRefraction = 1.45f;
    m.roughness = 0.5f;
    m.tintRoughness = 0.5f;
    m.clearcoatThickness = s.stickyClearcoatThickness;
    m.clearcoatRoughness = s.stickyClearcoatRoughness;
    m.textureID = 0;
    return m;
}
static float clamp01f(float v) { return std::fmax(0.0f, std::fmin(v, 1.0f)); }

bool load_mtl(HostScene& s, const std::string& path) {
    std::ifstream file(path);
    if (!file.is_open()) { set_error("File " + path + " can't be opened or does not exist."); return false; }
    const std::string dir = dir_of(path);
    RtxMaterial cur = fresh_material(s);
    std::string name, line;

    auto commit = [&]() {
        // a duplicate name is reported and dropped, and the half-built material
        // keeps accumulating (src/MaterialSerializer.cpp:45-55)
        if (s.materialNames.count(name)) { fprintf(stderr, "Material %s already exists.\n", name.c_str()); return; }
        s.stickyClearcoatThickness = cur.clearcoatThickness;
        s.stickyClearcoatRoughness = cur.clearcoatRoughness;
        s.materials.push_back(cur);
        s.materialNames[name] = (uint32_t)s.materials.size() - 1;
        cur = fresh_material(s);
    };
    auto rgb = [&](const std::vector<std::string>& t, RtxVec4& dst) {
        dst.x = num(t[1]); dst.y = num(t[2]); dst.z = num(t[3]); dst.w = 1.0f;
    };

    while (std::getline(file, line, '\n')) {
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> t = split_spaces(line);
        if (t.empty()) continue;
        const std::string& k = t[0];
        const bool has1 = t.size() > 1, has3 = t.size() > 3;
        if (k == "newmtl") {
            if (!name.empty()) { commit(); name.clear(); }
            if (has1) name = t[1];
        } else if (k == "Ka") { if (!has3) return true; rgb(t, cur.albedo); }
        else if (k == "Kd") { if (!has3) return true; rgb(t, cur.tint); }
        else if (k == "Ks") { if (!has3) return true; rgb(t, cur.specular); }
        else if (k == "Tf") { if (!has3) return true; rgb(t, cur.transmissionFilter); }
        else if (k == "Pr") { if (!has1) return true; cur.roughness = std::fmax(1e-6f, std::fmin(num(t[1]), 1.0f)); }
        else if (k == "Pm") { if (!has1) return true; cur.metallic = clamp01f(num(t[1])); }
        else if (k == "Ps") { if (!has1) return true; cur.sheen = clamp01f(num(t[1])); }
        else if (k == "Pc") { if (!has1) return true; cur.clearcoatThickness = num(t[1]); }
        else if (k == "Pcr") { if (!has1) return true; cur.clearcoatRoughness = num(t[1]); }
        else if (k == "Ke") { if (!has1) return true; cur.emmissionIntensity = std::fmax(num(t[1]), 0.0f); cur.roughness = 0.0f; }
        else if (k == "aniso") { if (!has1) return true; cur.anisotropy = clamp01f(num(t[1])); }
        else if (k == "anisor") { if (!has1) return true; cur.anisotropyRotation = num(t[1]); }
        else if (k == "Ni") { if (!has1) return true; cur.indexOfRefraction = std::fmax(1e-6f, num(t[1])); }
        else if (k == "Ns") { if (!has1) return true; cur.specularIntensity = clamp01f(num(t[1])); }
        else if (k == "d" || k == "Tr") {
            if (!has1) return true;
            float v = num(t[1]);
            cur.transparency = clamp01f(k == "Tr" ? v : 1.0f - v);
        } else if (k == "map_Ka") {
            if (!has1) return true;
            std::string tp = dir + "/" + t[1];
            std::vector<uint32_t> px; int32_t w = 0, h = 0;
            if (!read_bmp(tp, px, w, h)) { fprintf(stderr, "Texture %s can't be loaded\n", tp.c_str()); continue; }
            uint32_t sum = bmp_checksum(px);
            int found = -1;
            for (size_t i = 0; i < s.textureInfo.size(); ++i) if (s.textureInfo[i].checksum == sum) { found = (int)i; break; }
            cur.textureID = found >= 0 ? found : add_texture(s, px.data(), w, h, sum);
        }
    }
    if (!name.empty()) commit();
    return true;
}

// ---------------------------------------------------------------------------
// .obj — triangles only, "f" reads corners 1..3
// ---------------------------------------------------------------------------
struct Corner { int32_t v, vt, vn; };

// src/MeshSerializer.cpp:9-72: "a", "a/b", "a//c", "a/b/c". With no normal
// field the slot after the first '/' (or the vertex field itself) is reused as
// the normal index — kept as is.
static Corner parse_corner(const std::string& tok) {
    Corner c{0, -1, -1};
    size_t s1 = tok.find('/');
    c.v = atoi(tok.substr(0, s1).c_str()) - 1;
    if (s1 == std::string::npos) { c.vn = atoi(tok.c_str()) - 1; return c; }
    if (s1 + 1 < tok.size() && tok[s1 + 1] == '/') { c.vn = atoi(tok.substr(s1 + 2).c_str()) - 1; return c; }
    size_t s2 = tok.find('/', s1 + 1);
    if (s2 == std::string::npos) { c.vn = atoi(tok.substr(s1 + 1).c_str()) - 1; return c; }
    c.vt = atoi(tok.substr(s1 + 1, s2 - s1 - 1).c_str()) - 1;
    c.vn = atoi(tok.substr(s2 + 1).c_str()) - 1;
    return c;
}

struct ObjState { uint32_t currentMaterial = 0; };
static std::map<const HostScene*, ObjState> g_objState;   // currentMaterial survives across files (MeshSerializer member)

static void place_triangles(HostScene& s, const std::vector<RtxVec4>& verts, const std::vector<RtxVec4>& normals,
                            const std::vector<Corner>& corners, const std::vector<uint32_t>& faceMat) {
    // src/MeshSerializer.cpp:96-137
    const RtxVec4 offset = v4(s.width / 2.0f, s.height / 2.0f, s.depth / 2.0f);
    const float scale = 100.0f * s.camera.aspectRatio;
    const size_t nf = faceMat.size();
    s.objects.reserve(s.objects.size() + nf);
    RtxObject t = blank_object();
    t.type = RTX_TRIANGLE;
    for (size_t f = 0; f < nf; ++f) {
        const Corner* c = &corners[3 * f];
        for (int k = 0; k < 3; ++k) t.vertices[k] = verts[(uint32_t)c[k].v] * scale + offset;
        t.position = div_scalar(t.vertices[0] + t.vertices[1] + t.vertices[2], 3.0f);
        if (c[0].vn != -1) {
            for (int k = 0; k < 3; ++k) t.normals[k] = normals[c[k].vn];
        } else {
            t.normals[0] = t.normals[1] = t.normals[2] = normals[f];
        }
        t.materialID = (int32_t)faceMat[f];
        s.objects.push_back(t);
    }
}

bool load_obj(HostScene& s, const std::string& path) {
    std::ifstream file(path);
    if (!file.is_open()) { fprintf(stderr, "File %s can't be opened.\n", path.c_str()); return true; }  // soft failure, as the reference
    ObjState& st = g_objState[&s];
    std::vector<RtxVec4> verts, normals;
    std::vector<Corner> corners;
    std::vector<uint32_t> faceMat;
    std::string line;
    RtxVec4 tmp = v4(0, 0, 0);
    while (std::getline(file, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> t = split_spaces(line);
        if (t.empty()) continue;
        if (t[0] == "v" || t[0] == "vn") {
            if (t.size() <= 3) break;
            tmp.x = num(t[1]); tmp.y = num(t[2]); tmp.z = num(t[3]);
            (t[0] == "v" ? verts : normals).push_back(tmp);
        } else if (t[0] == "vt") {
            if (t.size() <= 2) break;
        } else if (t[0] == "f") {
            if (t.size() <= 3) break;
            for (int k = 1; k <= 3; ++k) corners.push_back(parse_corner(t[k]));
            faceMat.push_back(st.currentMaterial);
        } else if (t[0] == "usemtl") {
            if (t.size() <= 1) break;
            auto it = s.materialNames.find(t[1]);
            st.currentMaterial = it == s.materialNames.end() ? 0u : it->second;
        }
    }
    // face normals are appended BEHIND the vn normals (src/MeshSerializer.cpp:74-94)
    for (size_t f = 0; f < faceMat.size(); ++f) {
        RtxVec4 A = verts[(uint32_t)corners[3 * f].v], B = verts[(uint32_t)corners[3 * f + 1].v], C = verts[(uint32_t)corners[3 * f + 2].v];
        normals.push_back(unit(cross3(B - A, C - A)));
    }
    place_triangles(s, verts, normals, corners, faceMat);
    return true;
}

// ---------------------------------------------------------------------------
// .scn
// ---------------------------------------------------------------------------
bool load_scn(HostScene& s, const std::string& path) {
    std::ifstream file(path);
    if (!file.is_open()) { set_error("File " + path + " can't be opened or does not exist."); return false; }
    const std::string dir = dir_of(path);
    static const char* kTypes[4] = {"sphere", "plane", "disk", "cube"};
    static const char* kProps[6] = {"position", "material", "radius", "normal", "scale", "rotation"};

    RtxObject cur; RtxVec4 scales; int curType;
    auto reset = [&]() {
        cur = blank_object();
        cur.normals[0] = cur.normals[1] = cur.normals[2] = v4(0, 1, 0);
        scales = v4(1, 1, 1);
        curType = RTX_INVALID;
        cur.materialID = 0;
    };
    reset();

    bool inScene = false;
    std::string line;
    while (std::getline(file, line, '\n')) {
        if (line.empty()) continue;
        if (line[0] == '!') fprintf(stdout, "%s\n", line.c_str());
        std::vector<std::string> t = split_spaces(line);
        if (t.empty()) continue;
        const std::string key = t[0].substr(0, 19);
        if (!inScene) {
            if (t[0] == "mtllib" && t.size() > 1) { if (!load_mtl(s, dir + "/" + t[1])) return false; }
            else if (t[0] == "scene") inScene = true;
            continue;
        }
        if (curType == RTX_INVALID) {
            if (t[0] == "mesh") {
                if (t.size() <= 1) { fprintf(stderr, "Invalid mesh format.\n"); return true; }
                load_obj(s, dir + "//" + t[1]);
            } else {
                for (int k = 0; k < 4; ++k) if (key == kTypes[k]) { curType = k; cur.type = k; }
            }
            continue;
        }
        bool isProp = false;
        for (int k = 0; k < 6; ++k) if (key == kProps[k]) isProp = true;
        if (isProp) {
            if (t[0] == "position") { if (t.size() > 3) cur.position = v4(num(t[1]), num(t[2]), num(t[3])); }
            else if (t[0] == "radius") { if (t.size() > 1) cur.radius = num(t[1]); }
            else if (t[0] == "scale") {
                if (t.size() > 1) {
                    RtxVec4 v = v4(0, 0, 0);
                    v.x = num(t[1]); v.y = v.x;               // z stays 0 with a single value
                    if (t.size() > 2) { v.y = num(t[2]); v.z = v.y; if (t.size() > 3) v.z = num(t[3]); }
                    scales = v;
                }
            } else if (t[0] == "material") {
                if (t.size() > 1) { auto it = s.materialNames.find(t[1]); cur.materialID = it == s.materialNames.end() ? 0 : (int32_t)it->second; }
            } else if (t[0] == "normal") { if (t.size() > 3) cur.normals[0] = v4(num(t[1]), num(t[2]), num(t[3])); }
        } else if (t[0] == "}") {
            if (curType == RTX_PLANE) add_plane(s, cur.position, cur.normals, scales, cur.materialID);
            else if (curType == RTX_DISK) add_disk(s, cur.position, cur.normals[0], cur.radius, cur.materialID);
            else if (curType == RTX_CUBE) add_cube(s, cur.position, scales, cur.materialID);
            else s.objects.push_back(cur);
            reset();
        }
    }
    return true;
}

} // namespace rtxh
