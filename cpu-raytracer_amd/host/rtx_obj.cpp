// rtx_obj.cpp — OBJ / MTL loading for librtx_host.so (the data format in front of the path).
//
// Produces exactly the triangle soup the reference's OBJLoader::load_obj builds (OBJLoader.cpp:70-187): faces in file
// order, polygons triangulated, texture v flipped (OBJLoader.cpp:139-141), per-face material id with -1 mapped to 0
// (:172-173), materials as OBJLoader.cpp:8-41 fills them (Kd / Ks / Kt|Tf / Ni / map_Kd, magenta stand-in when the MTL
// has no material).  The reference delegates parsing to the vendored tinyobjloader 2.0 header (MIT); what is restated here
// is that library's observable behaviour for the constructs these assets use:
//   * decimal number parsing — parse_double / parse_real are DERIVED from tinyobjloader's tryParseDouble / parseReal (attribution
//     and licence notice at the function): it can differ from strtod in the last bit, so its exact operation sequence is kept;
//   * index fix-up (1-based, negative = relative), "v", "v/vt", "v//vn", "v/vt/vn";
//   * polygon triangulation by ear clipping in the dominant-axis projection (triangles pass through untouched);
//   * material defaults (all colours 0, Ni 1).
#include "../../include/rtx_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits>
#include <map>
#include <string>
#include <vector>

namespace {

bool is_digit(char c) { return c >= '0' && c <= '9'; }

// ---- decimal number parsing: derived from tinyobjloader ---------------------------------------------------------------------------
// parse_double / parse_real below follow tinyobjloader 2.0's tryParseDouble / parseReal (the copy the reference vendors:
// include/tiny_obj_loader/tiny_obj_loader.h:805-940) operation for operation — digit accumulation by repeated x10, the eight-entry
// fraction table, pow(10, -k) beyond it, ldexp(mantissa * pow(5, e), e) for exponents, then a cast to float.  That sequence is NOT
// strtod (it can differ in the last bit of the float), and bit-identical vertices need exactly it, so this function is a compacted
// restatement of that third-party code rather than independent work.  tinyobjloader is MIT-licensed:
//
//   The MIT License (MIT)
//   Copyright (c) 2012-2018 Syoyo Fujita and many contributors.
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated documentation files
//   (the "Software"), to deal in the Software without restriction, including without limitation the rights to use, copy, modify, merge,
//   publish, distribute, sublicense, and/or sell copies of the Software, and to permit persons to whom the Software is furnished to do
//   so, subject to the following conditions:
//   The above copyright notice and this permission notice shall be included in all copies or substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF
//   MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE
//   FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN
//   CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
bool parse_double(const char * s, const char * end, double * out) {
    if (s >= end) return false;
    double mantissa = 0.0; int exponent = 0; char sign = '+', esign = '+';
    const char * c = s; int read = 0; bool leading_dot = false;
    if (*c == '+' || *c == '-') { sign = *c; c++; if (c != end && *c == '.') leading_dot = true; }
    else if (is_digit(*c)) {}
    else if (*c == '.') leading_dot = true;
    else return false;
    bool more = c != end;
    if (!leading_dot) {
        while (more && is_digit(*c)) { mantissa *= 10; mantissa += (int)(*c - '0'); c++; read++; more = c != end; }
        if (read == 0) return false;
    }
    if (more) {
        bool has_exp = false;
        if (*c == '.') {
            static const double lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
            c++; read = 1; more = c != end;
            while (more && is_digit(*c)) {
                mantissa += (int)(*c - '0') * (read < 8 ? lut[read] : pow(10.0, -read));
                read++; c++; more = c != end;
            }
            has_exp = more && (*c == 'e' || *c == 'E');
        } else if (*c == 'e' || *c == 'E') has_exp = true;
        if (has_exp) {
            c++; more = c != end;
            if (more && (*c == '+' || *c == '-')) { esign = *c; c++; }
            else if (more && is_digit(*c)) {}
            else return false;
            read = 0; more = c != end;
            while (more && is_digit(*c)) { exponent *= 10; exponent += (int)(*c - '0'); c++; read++; more = c != end; }
            exponent *= (esign == '+' ? 1 : -1);
            if (read == 0) return false;
        }
    }
    *out = (sign == '+' ? 1 : -1) * (exponent ? ldexp(mantissa * pow(5.0, exponent), exponent) : mantissa);
    return true;
}

float parse_real(const char *& tok, double dflt = 0.0) {
    tok += strspn(tok, " \t");
    const char * end = tok + strcspn(tok, " \t\r");
    double v = dflt;
    parse_double(tok, end, &v);
    tok = end;
    return (float)v;
}

struct VIdx { int v, vt, vn; };

bool fix_index(int idx, int n, int * out) {
    if (idx > 0) { *out = idx - 1; return true; }
    if (idx == 0) return false;
    *out = n + idx; return true;
}

bool parse_triple(const char *& tok, int nv, int nvt, int nvn, VIdx * r) {
    r->v = r->vt = r->vn = -1;
    if (!fix_index(atoi(tok), nv, &r->v)) return false;
    tok += strcspn(tok, "/ \t\r");
    if (tok[0] != '/') return true;
    tok++;
    if (tok[0] == '/') {                                    // v//vn
        tok++;
        if (!fix_index(atoi(tok), nvn, &r->vn)) return false;
        tok += strcspn(tok, "/ \t\r");
        return true;
    }
    if (!fix_index(atoi(tok), nvt, &r->vt)) return false;   // v/vt[/vn]
    tok += strcspn(tok, "/ \t\r");
    if (tok[0] != '/') return true;
    tok++;
    if (!fix_index(atoi(tok), nvn, &r->vn)) return false;
    tok += strcspn(tok, "/ \t\r");
    return true;
}

// point-in-polygon (crossing number) for the ear test
bool pnpoly3(const float * vx, const float * vy, float tx, float ty) {
    bool c = false;
    for (int i = 0, j = 2; i < 3; j = i++)
        if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
    return c;
}

void triangulate(const std::vector<VIdx> & face, const std::vector<float> & v, std::vector<VIdx> & out) {
    size_t n = face.size();
    if (n < 3) return;
    if (n == 3) { out.push_back(face[0]); out.push_back(face[1]); out.push_back(face[2]); return; }
    size_t axes[2] = { 1, 2 };
    for (size_t k = 0; k < n; ++k) {
        const size_t a = (size_t)face[k % n].v, b = (size_t)face[(k + 1) % n].v, c = (size_t)face[(k + 2) % n].v;
        if (3 * a + 2 >= v.size() || 3 * b + 2 >= v.size() || 3 * c + 2 >= v.size()) continue;
        const float e0x = v[3 * b] - v[3 * a], e0y = v[3 * b + 1] - v[3 * a + 1], e0z = v[3 * b + 2] - v[3 * a + 2];
        const float e1x = v[3 * c] - v[3 * b], e1y = v[3 * c + 1] - v[3 * b + 1], e1z = v[3 * c + 2] - v[3 * b + 2];
        const float cx = fabsf(e0y * e1z - e0z * e1y), cy = fabsf(e0z * e1x - e0x * e1z), cz = fabsf(e0x * e1y - e0y * e1x);
        const float eps = std::numeric_limits<float>::epsilon();
        if (cx > eps || cy > eps || cz > eps) {
            if (!(cx > cy && cx > cz)) { axes[0] = 0; if (cz > cx && cz > cy) axes[1] = 1; }
            break;
        }
    }
    float area = 0;
    for (size_t k = 0; k < n; ++k) {
        const size_t a = (size_t)face[k % n].v, b = (size_t)face[(k + 1) % n].v;
        if (3 * a + axes[0] >= v.size() || 3 * a + axes[1] >= v.size() || 3 * b + axes[0] >= v.size() || 3 * b + axes[1] >= v.size()) continue;
        area += (v[3 * a + axes[0]] * v[3 * b + axes[1]] - v[3 * a + axes[1]] * v[3 * b + axes[0]]) * 0.5f;
    }
    std::vector<VIdx> rem = face;
    size_t guess = 0, budget = face.size(), prev = rem.size();
    while (rem.size() > 3 && budget > 0) {
        n = rem.size();
        if (guess >= n) guess -= n;
        if (prev != n) { prev = n; budget = n; } else budget--;
        VIdx ind[3]; float vx[3], vy[3];
        for (size_t k = 0; k < 3; k++) {
            ind[k] = rem[(guess + k) % n];
            const size_t vi = (size_t)ind[k].v;
            if (3 * vi + axes[0] >= v.size() || 3 * vi + axes[1] >= v.size()) { vx[k] = vy[k] = 0.0f; }
            else { vx[k] = v[3 * vi + axes[0]]; vy[k] = v[3 * vi + axes[1]]; }
        }
        const float cross = (vx[1] - vx[0]) * (vy[2] - vy[1]) - (vy[1] - vy[0]) * (vx[2] - vx[1]);
        if (cross * area < 0.0f) { guess += 1; continue; }      // reflex corner
        bool overlap = false;
        for (size_t o = 3; o < n; ++o) {
            const size_t idx = (guess + o) % n;
            if (idx >= rem.size()) continue;
            const size_t ovi = (size_t)rem[idx].v;
            if (3 * ovi + axes[0] >= v.size() || 3 * ovi + axes[1] >= v.size()) continue;
            if (pnpoly3(vx, vy, v[3 * ovi + axes[0]], v[3 * ovi + axes[1]])) { overlap = true; break; }
        }
        if (overlap) { guess += 1; continue; }
        out.push_back(ind[0]); out.push_back(ind[1]); out.push_back(ind[2]);   // an ear
        rem.erase(rem.begin() + (long)((guess + 1) % n));
    }
    if (rem.size() == 3) { out.push_back(rem[0]); out.push_back(rem[1]); out.push_back(rem[2]); }
}

struct Mtl { std::string name; float kd[3], ks[3], kt[3], ni; std::string map_kd; };

void load_mtl(const std::string & path, std::vector<Mtl> & mats, std::map<std::string, int> & by_name) {
    FILE * f = fopen(path.c_str(), "rb");
    if (!f) return;
    char line[4096];
    Mtl cur; bool has = false;
    auto reset = [&]() { cur = Mtl(); for (int i = 0; i < 3; i++) cur.kd[i] = cur.ks[i] = cur.kt[i] = 0.0f; cur.ni = 1.0f; };
    reset();
    while (fgets(line, sizeof(line), f)) {
        size_t len = strlen(line);
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        const char * t = line + strspn(line, " \t");
        if (!*t || *t == '#') continue;
        if (!strncmp(t, "newmtl", 6) && (t[6] == ' ' || t[6] == '\t')) {
            if (has) { if (!by_name.count(cur.name)) by_name[cur.name] = (int)mats.size(); mats.push_back(cur); }
            reset(); has = true;
            t += 7; t += strspn(t, " \t"); cur.name = t;
            continue;
        }
        auto vec3 = [&](float * d) { const char * p = t + 2; d[0] = parse_real(p); d[1] = parse_real(p); d[2] = parse_real(p); };
        if (t[0] == 'K' && t[1] == 'd' && (t[2] == ' ' || t[2] == '\t')) vec3(cur.kd);
        else if (t[0] == 'K' && t[1] == 's' && (t[2] == ' ' || t[2] == '\t')) vec3(cur.ks);
        else if (((t[0] == 'K' && t[1] == 't') || (t[0] == 'T' && t[1] == 'f')) && (t[2] == ' ' || t[2] == '\t')) vec3(cur.kt);
        else if (t[0] == 'N' && t[1] == 'i' && (t[2] == ' ' || t[2] == '\t')) { const char * p = t + 2; cur.ni = parse_real(p); }
        else if (!strncmp(t, "map_Kd", 6) && (t[6] == ' ' || t[6] == '\t')) {
            const char * p = t + 7; p += strspn(p, " \t");
            // options ("-o u v w" ...) are not used by these assets; the file name is the last token
            std::string rest(p); size_t sp = rest.find_last_of(" \t");
            cur.map_kd = sp == std::string::npos ? rest : rest.substr(sp + 1);
        }
    }
    if (has) { if (!by_name.count(cur.name)) by_name[cur.name] = (int)mats.size(); mats.push_back(cur); }
    fclose(f);
}

}  // namespace

// materials, OBJLoader.cpp:8-41
static int fill_materials(rtxh_obj * out, const std::vector<Mtl> & mats, const std::string & dir) {
    rtxh_mesh & m = out->mesh;
    const int nm = mats.empty() ? 1 : (int)mats.size();
    m.material_count = nm;
    out->materials = (rtx_material *)calloc((size_t)nm, sizeof(rtx_material));
    out->texture_names = (char *)calloc((size_t)nm, RTXH_TEXNAME_MAX);
    if (!out->materials || !out->texture_names) return RTX_ERR_OOM;
    if (mats.empty()) {
        rtx_material & d = out->materials[0];
        d.diffuse[0] = 1.0f; d.diffuse[1] = 0.0f; d.diffuse[2] = 1.0f; d.texture_id = -1; d.index_of_refraction = 1.0f;   // magenta, :31-38
    } else for (int i = 0; i < nm; i++) {
        rtx_material & d = out->materials[i];
        memcpy(d.diffuse, mats[i].kd, 12); memcpy(d.reflection, mats[i].ks, 12); memcpy(d.transmittance, mats[i].kt, 12);
        d.index_of_refraction = mats[i].ni;
        d.texture_id = -1;
        if (!mats[i].map_kd.empty()) {
            d.texture_id = i;                                                       // index into texture_names; the caller de-duplicates
            snprintf(out->texture_names + (size_t)i * RTXH_TEXNAME_MAX, RTXH_TEXNAME_MAX, "%s%s", dir.c_str(), mats[i].map_kd.c_str());
        }
    }
    return RTX_OK;
}

extern "C" int rtxh_obj_load(const char * path, rtxh_obj * out) {
    if (!path || !out) return RTX_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    FILE * f = fopen(path, "rb");
    if (!f) return RTX_ERR_INVALID_ARG;                        // the reference prints and abort()s (OBJLoader.cpp:80-85)
    std::string dir(path); { size_t s = dir.find_last_of('/'); dir = s == std::string::npos ? "" : dir.substr(0, s + 1); }
    std::vector<float> v, vt, vn;
    std::vector<Mtl> mats; std::map<std::string, int> by_name;
    std::vector<VIdx> tri_idx; std::vector<int> tri_mat;
    int cur_mat = -1;
    std::vector<char> line(1 << 16);
    std::vector<VIdx> face, tris;
    while (fgets(line.data(), (int)line.size(), f)) {
        size_t len = strlen(line.data());
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        const char * t = line.data() + strspn(line.data(), " \t");
        if (!*t || *t == '#') continue;
        if (t[0] == 'v' && (t[1] == ' ' || t[1] == '\t')) { const char * p = t + 2; for (int k = 0; k < 3; k++) v.push_back(parse_real(p)); }
        else if (t[0] == 'v' && t[1] == 'n' && (t[2] == ' ' || t[2] == '\t')) { const char * p = t + 3; for (int k = 0; k < 3; k++) vn.push_back(parse_real(p)); }
        else if (t[0] == 'v' && t[1] == 't' && (t[2] == ' ' || t[2] == '\t')) { const char * p = t + 3; for (int k = 0; k < 2; k++) vt.push_back(parse_real(p)); }
        else if (t[0] == 'f' && (t[1] == ' ' || t[1] == '\t')) {
            const char * p = t + 2; p += strspn(p, " \t");
            face.clear();
            bool ok = true;
            while (*p && *p != '\r' && *p != '\n') {
                VIdx vi;
                if (!parse_triple(p, (int)v.size() / 3, (int)vt.size() / 2, (int)vn.size() / 3, &vi)) { ok = false; break; }
                face.push_back(vi);
                p += strspn(p, " \t\r");
            }
            if (!ok) continue;
            tris.clear();
            triangulate(face, v, tris);
            for (size_t k = 0; k + 2 < tris.size(); k += 3) { tri_idx.push_back(tris[k]); tri_idx.push_back(tris[k + 1]); tri_idx.push_back(tris[k + 2]); tri_mat.push_back(cur_mat); }
        }
        else if (!strncmp(t, "usemtl", 6) && (t[6] == ' ' || t[6] == '\t')) {
            const char * p = t + 7; p += strspn(p, " \t");
            auto it = by_name.find(p);
            cur_mat = it == by_name.end() ? -1 : it->second;
        }
        else if (!strncmp(t, "mtllib", 6) && (t[6] == ' ' || t[6] == '\t')) {
            const char * p = t + 7; p += strspn(p, " \t");
            load_mtl(dir + p, mats, by_name);
        }
    }
    fclose(f);

    const int n = (int)tri_mat.size();
    if (n == 0) return RTX_ERR_INVALID_ARG;
    // tinyobjloader does not range-check indices and the reference then reads past its arrays (OBJLoader.cpp:120-150); here a face that
    // names a vertex / texcoord / normal the file does not have makes the file invalid
    for (const VIdx & x : tri_idx)
        if (x.v < 0 || (size_t)x.v >= v.size() / 3 || (x.vt >= 0 && (size_t)x.vt >= vt.size() / 2) || (x.vn >= 0 && (size_t)x.vn >= vn.size() / 3)) return RTX_ERR_INVALID_ARG;
    rtxh_mesh & m = out->mesh;
    m.triangle_count = n;
    m.positions = (float *)malloc(sizeof(float) * 9 * (size_t)n);
    m.normals   = (float *)malloc(sizeof(float) * 9 * (size_t)n);
    m.texcoords = (float *)malloc(sizeof(float) * 6 * (size_t)n);
    m.material_ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!m.positions || !m.normals || !m.texcoords || !m.material_ids) return RTX_ERR_OOM;
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            const VIdx & x = tri_idx[3 * (size_t)i + k];
            for (int c = 0; c < 3; c++) m.positions[9 * (size_t)i + 3 * k + c] = v[3 * (size_t)x.v + c];
            // OBJLoader.cpp:139-141: (u, 1 - v), or (0,0) when the face has no texcoord
            m.texcoords[6 * (size_t)i + 2 * k]     = x.vt >= 0 ? vt[2 * (size_t)x.vt] : 0.0f;
            m.texcoords[6 * (size_t)i + 2 * k + 1] = x.vt >= 0 ? 1.0f - vt[2 * (size_t)x.vt + 1] : 0.0f;
            for (int c = 0; c < 3; c++) m.normals[9 * (size_t)i + 3 * k + c] = x.vn >= 0 ? vn[3 * (size_t)x.vn + c] : 0.0f;
        }
        m.material_ids[i] = tri_mat[i] < 0 ? 0 : tri_mat[i];                      // OBJLoader.cpp:172-173
    }
    return fill_materials(out, mats, dir);
}

// OBJLoader::load_mtl (OBJLoader.cpp:43-68): the material half alone, for a mesh whose geometry comes from a `.bvh` cache file
// (BottomLevelBVH.cpp:28-33).  The .mtl is the one NAMED LIKE THE OBJ (<name>.obj -> <name>.mtl), not the OBJ's mtllib line.
extern "C" int rtxh_mtl_load(const char * obj_path, rtxh_obj * out) {
    if (!obj_path || !out) return RTX_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    std::string path(obj_path);
    if (path.size() < 4) return RTX_ERR_INVALID_ARG;
    std::string dir(path); { size_t s = dir.find_last_of('/'); dir = s == std::string::npos ? "" : dir.substr(0, s + 1); }
    const std::string mtl = path.substr(0, path.size() - 4) + ".mtl";
    FILE * probe = fopen(mtl.c_str(), "rb");
    if (!probe) return RTX_ERR_STATE;                              // the reference prints a warning and registers no material at all
    fclose(probe);
    std::vector<Mtl> mats; std::map<std::string, int> by_name;
    load_mtl(mtl, mats, by_name);
    return fill_materials(out, mats, dir);
}

extern "C" int rtxh_obj_free(rtxh_obj * o) {
    if (!o) return RTX_ERR_INVALID_ARG;
    rtxh_mesh_free(&o->mesh);
    free(o->materials); free(o->texture_names);
    memset(o, 0, sizeof(*o));
    return RTX_OK;
}
