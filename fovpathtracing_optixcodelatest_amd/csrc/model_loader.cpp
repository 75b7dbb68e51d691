// model_loader.cpp -- host-side scene ingestion behind the C ABI (fovpt_model_*, include/fovpt.h): what a C++ caller
// of the reference gets from loadOBJ (PT_sv5_/Model.cpp:138-217), addVertex (:49-82) and loadTexture (:84-136), i.e.
// from the vendored tinyobjloader 2.0.0-rc (support/tinyobjloader, LoadObj with triangulate = true) and stb_image
// (stbi_load(..., STBI_rgb_alpha)).  OBJ / MTL, PNG, TGA, PPM, HDR and glTF are written from the formats' definitions and the
// behaviour of those libraries, not from their text.  The JPEG decoder necessarily restates stb_image's NUMERIC choices (inverse
// DCT, chroma filters, colour conversion: the standard leaves them to the decoder, see JpegDecoder) -- stb_image is public domain /
// MIT (Sean Barrett et al.), its transform the Independent JPEG Group's; the entropy decoding is the standard's own procedure
// (T.81 annex F, figure F.16).  Everything is pinned to the reference's own
// output by tests/test_ref_pin_cpu.py (tests/golden/ref_loaders.npz and, where oracle/_ref exists, seeded random OBJ
// files through both).  Plain host C++: no HIP, no GPU needed.
//
// What is reproduced (and tested):
//   * OBJ: v / vn / vt (one or two coordinates) / f with every corner syntax and negative (relative) indices,
//     o and g starting a new shape, usemtl, mtllib (several files), polygons triangulated by ear clipping in
//     binary32 on the projection tinyobj picks; a convex planar quad becomes the fan (0,1,2) (0,2,3)
//   * MTL: newmtl, Kd (absent: 0 0 0), Ke, map_Kd with options before the file name
//   * one TriangleMesh per (shape, material id) in ascending id order; vertices de-duplicated per SHAPE by their
//     (v, vn, vt) triple -- the reference's map lives per shape, so a corner first used under an earlier material
//     keeps the index it got in THAT mesh (a quirk of the reference, kept); textures are looked up per shape too,
//     so a file named by two shapes is loaded twice
//   * textures: PNG (all colour types and bit depths, tRNS, Adam7), Truevision TGA (the format of the reference's default
//     scene: true colour, gray, colour-mapped, run-length forms) and binary PPM, as stbi_load returns them with four
//     channels, then mirrored along y (:117-126), and JPEG (below).  Anything else counts as "could not load": id -1 (:129-131).
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <new>
#include <exception>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/fovpt.h"

void fovpt_internal_set_error(const char* text);      // fovpt_api.hip: what fovpt_last_error(NULL) returns

namespace {

struct Corner {                                        // tinyobj::index_t in the reference's comparison order (:30-44)
    int v, vn, vt;
    bool operator<(const Corner& o) const
    {
        if (v != o.v) return v < o.v;
        if (vn != o.vn) return vn < o.vn;
        return vt < o.vt;
    }
};
struct F3 { float x, y, z; };
struct F2 { float x, y; };

struct MtlRec {
    bool has_kd = false;
    float kd[3] = {0.f, 0.f, 0.f}, ke[3] = {0.f, 0.f, 0.f};
    std::string map_kd;
};

struct Image { int w = 0, h = 0; std::vector<uint32_t> px; };        // rgba8 words, row 0 first

std::vector<std::string> split_ws(const std::string& line)
{
    std::vector<std::string> t;
    const size_t end = std::min(line.find('#'), line.size());
    size_t i = 0;
    while (i < end) {
        while (i < end && isspace((unsigned char)line[i])) i++;
        size_t j = i;
        while (j < end && !isspace((unsigned char)line[j])) j++;
        if (j > i) t.emplace_back(line, i, j - i);
        i = j;
    }
    return t;
}
std::string join_from(const std::vector<std::string>& t, size_t first)
{
    std::string s;
    for (size_t k = first; k < t.size(); k++) { if (k > first) s += ' '; s += t[k]; }
    return s;
}
float to_f(const std::string& s) { return (float)strtod(s.c_str(), nullptr); }     // decimal -> binary64 -> binary32

bool read_lines(const std::string& path, std::vector<std::string>& lines)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string l;
    while (std::getline(f, l)) { if (!l.empty() && l.back() == '\r') l.pop_back(); lines.push_back(l); }
    return true;
}

void parse_mtl(const std::string& path, std::vector<std::pair<std::string, MtlRec>>& out)
{
    std::vector<std::string> lines;
    if (!read_lines(path, lines)) return;
    std::map<std::string, size_t> seen;               // a name defined twice in one file: the later record replaces the earlier
    MtlRec* cur = nullptr;
    for (const std::string& line : lines) {
        const std::vector<std::string> t = split_ws(line);
        if (t.empty()) continue;
        const std::string& k = t[0];
        if (k == "newmtl") {
            const std::string name = join_from(t, 1);
            auto it = seen.find(name);
            if (it == seen.end()) { seen[name] = out.size(); out.emplace_back(name, MtlRec()); cur = &out.back().second; }
            else { out[it->second].second = MtlRec(); cur = &out[it->second].second; }
        } else if (!cur) {
            continue;
        } else if (k == "Kd" && t.size() >= 4) {
            cur->has_kd = true;
            for (int a = 0; a < 3; a++) cur->kd[a] = to_f(t[1 + a]);
        } else if (k == "Ke" && t.size() >= 4) {
            for (int a = 0; a < 3; a++) cur->ke[a] = to_f(t[1 + a]);
        } else if (k == "map_Kd" && t.size() >= 2) {
            cur->map_kd = t.back();                   // options (-s, -o, ...) precede the file name
        }
    }
}

// 'v', 'v/vt', 'v//vn', 'v/vt/vn' -> zero-based indices, -1 for absent; negative = relative to the end so far
bool parse_corner(const std::string& tok, int nv, int nt, int nn, Corner& c)
{
    std::string part[3];
    int np = 0;
    size_t i = 0;
    for (;;) {
        const size_t j = tok.find('/', i);
        if (np < 3) part[np] = tok.substr(i, j == std::string::npos ? std::string::npos : j - i);
        np++;
        if (j == std::string::npos) break;
        i = j + 1;
    }
    auto fix = [](const std::string& s, int n, bool& ok) {
        if (s.empty()) return -1;
        char* e = nullptr;
        const long v = strtol(s.c_str(), &e, 10);
        if (e == s.c_str() || *e) { ok = false; return -1; }
        return (int)(v > 0 ? v - 1 : n + v);
    };
    bool ok = true;
    c.v = fix(part[0], nv, ok);
    c.vt = np > 1 ? fix(part[1], nt, ok) : -1;
    c.vn = np > 2 ? fix(part[2], nn, ok) : -1;
    return ok && !part[0].empty();
}

// polygon -> triangles, ear clipping in binary32 (see the header comment); appends corner triples
void triangulate(const std::vector<Corner>& cs, const std::vector<F3>& pos, std::vector<Corner>& out)
{
    const int n = (int)cs.size();
    if (n < 3) return;
    if (n == 3) { out.insert(out.end(), cs.begin(), cs.end()); return; }
    const float eps = 1.1920928955078125e-07f;
    std::vector<bool> have(n);
    std::vector<F3> P(n);
    for (int k = 0; k < n; k++) {
        have[k] = cs[k].v >= 0 && cs[k].v < (int)pos.size();
        P[k] = have[k] ? pos[cs[k].v] : F3{0.f, 0.f, 0.f};
    }
    auto comp = [](const F3& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); };
    int axes[2] = {1, 2};
    for (int k = 0; k < n; k++) {
        const int k1 = (k + 1) % n, k2 = (k + 2) % n;
        if (!have[k] || !have[k1] || !have[k2]) continue;
        const F3 &p0 = P[k], &p1 = P[k1], &p2 = P[k2];
        const float e0x = p1.x - p0.x, e0y = p1.y - p0.y, e0z = p1.z - p0.z;
        const float e1x = p2.x - p1.x, e1y = p2.y - p1.y, e1z = p2.z - p1.z;
        const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
        if (cx > eps || cy > eps || cz > eps) {
            if (!(cx > cy && cx > cz)) {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    float area = 0.f;
    for (int k = 0; k < n; k++) {
        const int k1 = (k + 1) % n;
        if (!have[k] || !have[k1]) continue;
        area = area + (comp(P[k], axes[0]) * comp(P[k1], axes[1]) - comp(P[k], axes[1]) * comp(P[k1], axes[0])) * 0.5f;
    }
    std::vector<int> rem(n);
    for (int k = 0; k < n; k++) rem[k] = k;
    int guess = 0, remaining_iter = n, prev_n = n;
    while (rem.size() > 3 && remaining_iter > 0) {
        const int m = (int)rem.size();
        if (guess >= m) guess -= m;
        if (prev_n != m) { prev_n = m; remaining_iter = m; }
        else remaining_iter--;
        int ind[3];
        float vx[3], vy[3];
        for (int k = 0; k < 3; k++) {
            ind[k] = rem[(guess + k) % m];
            vx[k] = have[ind[k]] ? comp(P[ind[k]], axes[0]) : 0.f;
            vy[k] = have[ind[k]] ? comp(P[ind[k]], axes[1]) : 0.f;
        }
        const float cross = (vx[1] - vx[0]) * (vy[2] - vy[1]) - (vy[1] - vy[0]) * (vx[2] - vx[1]);
        if (cross * area < 0.f) { guess++; continue; }
        bool overlap = false;
        for (int other = 3; other < m && !overlap; other++) {
            const int qi = rem[(guess + other) % m];
            if (!have[qi]) continue;
            const float tx = comp(P[qi], axes[0]), ty = comp(P[qi], axes[1]);
            bool c = false;
            for (int i = 0, j = 2; i < 3; j = i++)                        // pnpoly on the candidate ear
                if ((vy[i] > ty) != (vy[j] > ty) && tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]) c = !c;
            overlap = c;
        }
        if (overlap) { guess++; continue; }
        out.push_back(cs[ind[0]]); out.push_back(cs[ind[1]]); out.push_back(cs[ind[2]]);
        rem.erase(rem.begin() + (guess + 1) % m);
    }
    if (rem.size() == 3) { out.push_back(cs[rem[0]]); out.push_back(cs[rem[1]]); out.push_back(cs[rem[2]]); }
}

// ---- images ------------------------------------------------------------------------------------
bool read_file(const std::string& path, std::vector<uint8_t>& data)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    return true;
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// undo the per-scanline filters of `rows` scanlines at raw[pos]; false on truncated data or a bad filter type
bool png_unfilter(const std::vector<uint8_t>& raw, size_t& pos, int rows, size_t rowbytes, int bpp, std::vector<uint8_t>& out)
{
    out.assign((size_t)rows * rowbytes, 0);
    std::vector<uint8_t> zero(rowbytes, 0);
    for (int y = 0; y < rows; y++) {
        if (pos + 1 + rowbytes > raw.size()) return false;
        const int ft = raw[pos];
        const uint8_t* ln = raw.data() + pos + 1;
        pos += 1 + rowbytes;
        uint8_t* cur = out.data() + (size_t)y * rowbytes;
        const uint8_t* pv = y ? cur - rowbytes : zero.data();
        if (ft > 4) return false;
        for (size_t i = 0; i < rowbytes; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = pv[i], c = i >= (size_t)bpp ? pv[i - bpp] : 0;
            int pr = 0;
            if (ft == 1) pr = a;
            else if (ft == 2) pr = b;
            else if (ft == 3) pr = (a + b) >> 1;
            else if (ft == 4) {
                const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[i] = (uint8_t)(ln[i] + pr);
        }
    }
    return true;
}

// PNG -> rgba8 as stbi_load(..., 4) gives it: 16-bit samples keep their high byte, gray of 1/2/4 bits is scaled to
// 0..255, palette + tRNS alpha, a tRNS colour key gives alpha 0 (compared at the file's precision), Adam7
bool decode_png(const std::vector<uint8_t>& d, Image& img)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (d.size() < 8 || memcmp(d.data(), sig, 8)) return false;
    size_t pos = 8;
    std::vector<uint8_t> idat, plte, trns;
    bool has_hdr = false, has_trns = false, has_plte = false;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, comp = 0, filt = 0, interlace = 0;
    while (pos + 8 <= d.size()) {
        const uint32_t n = be32(d.data() + pos);
        const char* kind = (const char*)d.data() + pos + 4;
        const size_t body = pos + 8, avail = body <= d.size() ? std::min<size_t>(n, d.size() - body) : 0;
        if (!memcmp(kind, "IHDR", 4) && avail >= 13) {
            w = be32(d.data() + body); h = be32(d.data() + body + 4);
            depth = d[body + 8]; ctype = d[body + 9]; comp = d[body + 10]; filt = d[body + 11]; interlace = d[body + 12];
            has_hdr = true;
        } else if (!memcmp(kind, "PLTE", 4)) { plte.assign(d.begin() + body, d.begin() + body + avail); has_plte = true; }
        else if (!memcmp(kind, "tRNS", 4)) { trns.assign(d.begin() + body, d.begin() + body + avail); has_trns = true; }
        else if (!memcmp(kind, "IDAT", 4)) idat.insert(idat.end(), d.begin() + body, d.begin() + body + avail);
        else if (!memcmp(kind, "IEND", 4)) break;
        pos += 12 + (size_t)n;
    }
    if (!has_hdr || idat.empty()) return false;
    if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24) || comp || filt || interlace > 1) return false;
    if (!(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6)) return false;
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) return false;
    if ((ctype == 3 && depth == 16) || ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) || (ctype == 3 && !has_plte)) return false;
    if ((uint64_t)w * h > (1ull << 28)) return false;                     // (stb's own limit is 2^24 per side; this bounds the allocation)
    const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    const int bpp = std::max(1, chans * depth / 8);
    const size_t raw_max = ((size_t)w * chans * depth / 8 + 16) * ((size_t)h + 16) + 1024;   // all scanlines + filter bytes, 7 passes' padding
    // inflate
    std::vector<uint8_t> raw;
    {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) return false;
        zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size();
        uint8_t buf[1 << 16];
        int rc;
        do {
            zs.next_out = buf; zs.avail_out = sizeof(buf);
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return false; }
            raw.insert(raw.end(), buf, buf + (sizeof(buf) - zs.avail_out));
            if (raw.size() > raw_max) { inflateEnd(&zs); return false; }     // more data than the header's image can hold
            if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) { inflateEnd(&zs); return false; }   // truncated stream
        } while (rc != Z_STREAM_END);
        inflateEnd(&zs);
    }
    {   // the inflated stream must hold every scanline the header promises before buffers of the header's size are allocated
        size_t need = 0;
        if (!interlace) need = (((size_t)w * chans * depth + 7) / 8 + 1) * (size_t)h;
        else {
            static const int P7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
            for (const auto& a : P7) {
                const long pw = ((long)w - a[0] + a[2] - 1) / a[2], ph = ((long)h - a[1] + a[3] - 1) / a[3];
                if (pw > 0 && ph > 0) need += (((size_t)pw * chans * depth + 7) / 8 + 1) * (size_t)ph;
            }
        }
        if (raw.size() < need) return false;
    }
    std::vector<uint16_t> smp((size_t)w * h * chans, 0);
    auto put = [&](const std::vector<uint8_t>& rows, int pw, int ph, size_t rowbytes, int x0, int y0, int dx, int dy) {
        for (int y = 0; y < ph; y++) {
            const uint8_t* r = rows.data() + (size_t)y * rowbytes;
            for (int x = 0; x < pw; x++)
                for (int c = 0; c < chans; c++) {
                    uint16_t v;
                    if (depth == 8) v = r[x * chans + c];
                    else if (depth == 16) v = (uint16_t)((r[(x * chans + c) * 2] << 8) | r[(x * chans + c) * 2 + 1]);
                    else { const int bit = x * depth; v = (uint16_t)((r[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                    smp[((size_t)(y0 + y * dy) * w + (x0 + x * dx)) * chans + c] = v;
                }
        }
    };
    size_t off = 0;
    std::vector<uint8_t> rows;
    if (!interlace) {
        const size_t rb = ((size_t)w * chans * depth + 7) / 8;
        if (!png_unfilter(raw, off, (int)h, rb, bpp, rows)) return false;
        put(rows, (int)w, (int)h, rb, 0, 0, 1, 1);
    } else {
        static const int A7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};   // x0, y0, dx, dy
        for (const auto& a : A7) {
            const int pw = ((int)w - a[0] + a[2] - 1) / a[2], ph = ((int)h - a[1] + a[3] - 1) / a[3];
            if (pw <= 0 || ph <= 0) continue;
            const size_t rb = ((size_t)pw * chans * depth + 7) / 8;
            if (!png_unfilter(raw, off, ph, rb, bpp, rows)) return false;
            put(rows, pw, ph, rb, a[0], a[1], a[2], a[3]);
        }
    }
    img.w = (int)w; img.h = (int)h;
    img.px.assign((size_t)w * h, 0);
    if (ctype == 3) {
        uint8_t pal[256][4];
        for (int k = 0; k < 256; k++) {                                     // an index past the palette reads black
            const bool in = (size_t)k * 3 + 2 < plte.size();
            pal[k][0] = in ? plte[k * 3] : 0; pal[k][1] = in ? plte[k * 3 + 1] : 0; pal[k][2] = in ? plte[k * 3 + 2] : 0;
            pal[k][3] = (has_trns && (size_t)k < trns.size()) ? trns[k] : 255;
        }
        for (size_t p = 0; p < (size_t)w * h; p++) {
            const uint8_t* c = pal[smp[p] & 255];
            img.px[p] = c[0] | (c[1] << 8) | (c[2] << 16) | ((uint32_t)c[3] << 24);
        }
        return true;
    }
    bool keyed = false;
    uint16_t key[3] = {0, 0, 0};
    if (has_trns && (ctype == 0 || ctype == 2) && trns.size() / 2 >= (size_t)chans) {
        keyed = true;
        for (int c = 0; c < chans; c++) {
            key[c] = (uint16_t)((trns[2 * c] << 8) | trns[2 * c + 1]);
            if (depth <= 8) key[c] &= 255;                                   // stb keeps the low byte of an 8-bit key
        }
    }
    const int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    for (size_t p = 0; p < (size_t)w * h; p++) {
        const uint16_t* s = &smp[p * chans];
        uint8_t v[4];
        for (int c = 0; c < chans; c++) v[c] = depth == 16 ? (uint8_t)(s[c] >> 8) : (uint8_t)(s[c] * scale);
        uint8_t r, g, b, a;
        if (ctype == 0 || ctype == 4) { r = g = b = v[0]; a = ctype == 4 ? v[1] : 255; }
        else { r = v[0]; g = v[1]; b = v[2]; a = ctype == 6 ? v[3] : 255; }
        if (keyed) {
            bool eq = true;
            for (int c = 0; c < chans; c++) eq = eq && s[c] == key[c];
            a = eq ? 0 : 255;
        }
        img.px[p] = r | (g << 8) | (b << 16) | ((uint32_t)a << 24);
    }
    return true;
}

bool decode_ppm(const std::vector<uint8_t>& d, Image& img)       // binary 8-bit P6
{
    std::vector<std::string> toks;
    size_t pos = 0;
    while (toks.size() < 4) {
        size_t end = pos;
        while (end < d.size() && d[end] != '\n') end++;
        if (end >= d.size()) return false;
        for (const std::string& t : split_ws(std::string((const char*)d.data() + pos, end - pos))) toks.push_back(t);
        pos = end + 1;
    }
    const long w = strtol(toks[1].c_str(), nullptr, 10), h = strtol(toks[2].c_str(), nullptr, 10), mx = strtol(toks[3].c_str(), nullptr, 10);
    // sizes are bounded BEFORE any arithmetic on them (a 64-bit w * h * 3 wraps): 2^24 per side, 2^28 pixels, as for PNG
    if (toks[0] != "P6" || mx != 255 || w <= 0 || h <= 0 || w > (1l << 24) || h > (1l << 24)) return false;
    if ((uint64_t)w * (uint64_t)h > (1ull << 28) || (uint64_t)w * (uint64_t)h * 3ull > (uint64_t)(d.size() - pos)) return false;
    img.w = (int)w; img.h = (int)h;
    img.px.resize((size_t)w * h);
    for (size_t p = 0; p < (size_t)w * h; p++) {
        const uint8_t* c = d.data() + pos + p * 3;
        img.px[p] = c[0] | (c[1] << 8) | (c[2] << 16) | 0xff000000u;
    }
    return true;
}

// Truevision TGA -> rgba8 the way stb_image reads it: image types 1 / 2 / 3 and their run-length forms 9 / 10 / 11; 8 bits
// (gray or index), 15 / 16 (5-5-5, or gray + alpha for type 3), 24 and 32; rows bottom-up unless descriptor bit 5 is set;
// an index past the colour map reads entry 0.  (The texture format of the reference's default scene, main.cpp:196.)
bool decode_tga(const std::vector<uint8_t>& d, Image& img)
{
    if (d.size() < 18) return false;
    const int idlen = d[0], cmap_type = d[1], itype = d[2];
    const int cm_first = d[3] | (d[4] << 8), cm_len = d[5] | (d[6] << 8), cm_bits = d[7];
    const int w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8), bits = d[16], desc = d[17];
    const bool rle = itype >= 8;
    const int base = itype & 7;
    if (!(base == 1 || base == 2 || base == 3) || w == 0 || h == 0 || cmap_type > 1 || (base == 1) != (cmap_type == 1)) return false;
    if ((base == 1 && !(bits == 8 || bits == 16)) || (base == 3 && !(bits == 8 || bits == 16)) ||
        (base == 2 && !(bits == 15 || bits == 16 || bits == 24 || bits == 32))) return false;
    size_t pos = 18 + (size_t)idlen;
    auto expand = [](const uint8_t* px, int nbits, bool gray) -> uint32_t {       // one pixel -> rgba8 word
        uint32_t r, g, b, a = 255;
        if (nbits == 8) r = g = b = px[0];
        else if ((nbits == 15 || nbits == 16) && gray) { r = g = b = px[0]; a = px[1]; }
        else if (nbits == 15 || nbits == 16) {
            const uint32_t v = px[0] | ((uint32_t)px[1] << 8);
            r = ((v >> 10) & 31u) * 255u / 31u; g = ((v >> 5) & 31u) * 255u / 31u; b = (v & 31u) * 255u / 31u;
        } else { r = px[2]; g = px[1]; b = px[0]; if (nbits == 32) a = px[3]; }
        return r | (g << 8) | (b << 16) | (a << 24);
    };
    std::vector<uint32_t> palette;
    if (cmap_type) {
        if (!(cm_bits == 8 || cm_bits == 15 || cm_bits == 16 || cm_bits == 24 || cm_bits == 32)) return false;
        const size_t eb = (size_t)(cm_bits + 7) / 8;
        pos += (size_t)cm_first;                                   // stb skips "first entry index" BYTES, then reads cm_len entries
        if (pos + (size_t)cm_len * eb > d.size()) return false;
        if (base == 1) {
            palette.resize(cm_len);
            for (int k = 0; k < cm_len; k++) palette[k] = expand(d.data() + pos + k * eb, cm_bits, false);
        }
        pos += (size_t)cm_len * eb;
    }
    const size_t nb = (size_t)(bits + 7) / 8, n = (size_t)w * h;
    // the file must be able to hold the image before anything is allocated for it: n pixels raw, or at least one packet
    // byte per 128 pixels run-length encoded
    if (pos > d.size()) return false;
    if (!rle ? n * nb > d.size() - pos : (n + 127) / 128 > d.size() - pos) return false;
    std::vector<uint8_t> px(n * nb);
    if (!rle) {
        if (pos + n * nb > d.size()) return false;
        memcpy(px.data(), d.data() + pos, n * nb);
    } else {
        size_t got = 0;
        while (got < n) {
            if (pos >= d.size()) return false;
            const int c = d[pos++];
            const size_t cnt = (size_t)(c & 127) + 1, take = std::min(cnt, n - got);
            if (c & 128) {
                if (pos + nb > d.size()) return false;
                for (size_t k = 0; k < take; k++) memcpy(px.data() + (got + k) * nb, d.data() + pos, nb);
                pos += nb;
            } else {
                if (pos + cnt * nb > d.size()) return false;
                memcpy(px.data() + got * nb, d.data() + pos, take * nb);
                pos += cnt * nb;
            }
            got += cnt;
        }
    }
    img.w = w; img.h = h;
    img.px.resize(n);
    const bool top_down = (desc >> 5) & 1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = px.data() + ((size_t)y * w + x) * nb;
            uint32_t v;
            if (base == 1) {
                size_t idx = nb == 1 ? p[0] : (size_t)(p[0] | (p[1] << 8));
                if (idx >= (size_t)cm_len) idx = 0;
                v = palette.empty() ? 0u : palette[idx];
            } else {
                v = expand(p, bits, base == 3);
            }
            img.px[(size_t)(top_down ? y : h - 1 - y) * w + x] = v;
        }
    return true;
}

// JPEG (JFIF / Adobe; baseline, extended sequential and progressive DCT; 8 bits; 1, 3 or 4 components) -> rgba8 as
// stbi_load(..., 4) of the vendored stb_image gives it -- the JPEG format leaves the inverse DCT, the chroma interpolation and
// the colour conversion to the decoder, so "like stb_image" means ITS choices, restated here:
//   * coefficients: Huffman + EXTEND as the standard says (T.81 F.2.2, G.1.2 for the progressive refinements), dequantised in
//     16-bit arithmetic ((short)(value * q)); a baseline block as it is decoded, a progressive one when all scans are in;
//   * inverse DCT: the "slow integer" transform of the IJG library in the scaling stb_image uses (12-bit constants, columns
//     rounded to 2 extra bits, rows to the sample, +128, clamped);
//   * chroma: replicated horizontally for factors other than 2; for 2 the triangle filters 3:1 (one axis) and 9:3:3:1 (both)
//     centred between the samples, edge samples repeated; the luma rows a chroma row pair serves are chosen as stb does;
//   * Y Cb Cr -> R G B in 20.12 fixed point with the constants 1.40200, 0.71414, 0.34414 (its product masked to 16 bits),
//     1.77200; a three-component image is taken as RGB when its component ids are 'R' 'G' 'B' or when an Adobe marker says
//     "no transform" and there is no JFIF marker; four components are CMYK (transform 0) or YCCK (2), combined with K by the
//     8 x 8 -> 8 multiply of the graphics literature ((t + (t >> 8)) >> 8, t = a b + 128).
// Held against the reference's own stb_image through oracle/_ref (tests/golden/ref_jpeg.npz).
struct JpegDecoder {
    // One Huffman table in the standard's own terms (T.81 C.2 and F.2.2.3): the canonical code is given by how many codes
    // there are of every length; a symbol is found by growing the code bit by bit until it is not beyond the largest code of
    // its length (figure F.16: MINCODE / MAXCODE / VALPTR).  In front of that loop an 8-bit lookup answers the short codes.
    struct Huff {
        uint8_t value[256];           // HUFFVAL: the symbols in code order
        int32_t maxcode[17];          // largest code of each length 1..16, -1 where there is none
        uint16_t mincode[17];         // smallest code of each length
        uint16_t valptr[17];          // index of that code's symbol in value[]
        uint16_t look[256];           // the next 8 bits -> (code length << 8) | index into value[], 0: longer than 8 bits (or no code)
        int count = 0;
        bool ok = false;
    };
    struct Comp {
        int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, pred = 0;
        int x = 0, y = 0, w2 = 0, h2 = 0, bw = 0;               // valid samples, padded plane size, blocks per row
        std::vector<uint8_t> plane;
        std::vector<short> coeff;                                 // progressive only: 64 per block
    };
    const uint8_t* d; size_t n, pos = 0;
    Huff hdc[4], hac[4];
    uint16_t q[4][64];
    Comp comp[4];
    int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, jfif = false;
    int adobe = -1, rgb_ids = 0, restart = 0;
    int scan_n = 0, order[4] = {0, 0, 0, 0}, ss = 0, se = 63, ah = 0, al = 0, eobrun = 0, todo = 0;
    int marker = 0xff; bool nomore = false;

    static const uint8_t* zigzag()
    {
        static const uint8_t z[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35,
                                           42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                           63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
        return z;
    }
    int get8() { return pos < n ? d[pos++] : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }
    void skip(int k) { pos = k < 0 ? n : std::min(n, pos + (size_t)k); }

    bool build(Huff& h, const int* per_length)
    {
        // canonical assignment: the codes of one length are consecutive, the first of the next length is twice the successor of the
        // last.  A table that runs out of codes of some length (more than 2^len of them so far) is refused, like the reference's
        // reader refuses it ("bad code lengths").
        h.ok = false;
        unsigned next = 0;
        int filled = 0;
        memset(h.look, 0, sizeof(h.look));
        for (int len = 1; len <= 16; len++) {
            const int cnt = per_length[len - 1];
            h.mincode[len] = (uint16_t)next; h.valptr[len] = (uint16_t)filled; h.maxcode[len] = cnt ? (int32_t)(next + (unsigned)cnt - 1u) : -1;
            if (filled + cnt > 256) return false;
            if (cnt && next + (unsigned)cnt > (1u << len)) return false;
            if (len <= 8)
                for (int c = 0; c < cnt; c++) {
                    const unsigned first = (next + (unsigned)c) << (8 - len), span = 1u << (8 - len);
                    for (unsigned e = 0; e < span; e++) h.look[first + e] = (uint16_t)((len << 8) | (filled + c));
                }
            next = (next + (unsigned)cnt) << 1;
            filled += cnt;
        }
        h.count = filled;
        h.ok = true;
        return true;
    }
    // The entropy-coded segment as a bit stream: `acc` holds the next `have` bits left-aligned.  0xff 0x00 is the data byte 0xff;
    // 0xff followed by anything else is a marker -- it is remembered, and from there on (as past the end of the file) the stream
    // reads as zeros, which is what the reference's reader feeds its decoder too.
    uint64_t acc = 0; int have = 0;
    void need(int k)
    {
        while (have < k) {
            unsigned byte = 0;
            if (!nomore) {
                byte = (unsigned)get8();
                if (byte == 0xff) {
                    int c = get8();
                    while (c == 0xff) c = get8();
                    if (c != 0) { marker = c; nomore = true; byte = 0; }
                }
            }
            acc |= (uint64_t)byte << (56 - have);
            have += 8;
        }
    }
    unsigned peek(int k) { return (unsigned)(acc >> (64 - k)); }            // 1 <= k <= 32, after need(k)
    void drop(int k) { acc <<= k; have -= k; }
    int decode(const Huff& h)
    {
        need(32);
        const unsigned e = h.look[peek(8)];
        if (e) { drop((int)(e >> 8)); return h.value[e & 255u]; }
        const unsigned w = peek(16);
        for (int len = 9; len <= 16; len++) {
            const int32_t code = (int32_t)(w >> (16 - len));
            if (code <= h.maxcode[len]) {                                   // (maxcode -1: no code of this length)
                if (code < (int32_t)h.mincode[len]) return -1;
                drop(len);
                return h.value[h.valptr[len] + (unsigned)(code - (int32_t)h.mincode[len])];
            }
        }
        return -1;                                                          // sixteen bits that are no code
    }
    int getbits(int k)
    {
        if (k <= 0 || k > 16) return 0;
        need(k);
        const unsigned r = peek(k);
        drop(k);
        return (int)r;
    }
    int getbit() { return getbits(1); }
    int extend(int k)                                             // RECEIVE + EXTEND (T.81 F.2.2.1)
    {
        if (k <= 0 || k > 16) return 0;
        const int v = getbits(k);
        return v < (1 << (k - 1)) ? v - (1 << k) + 1 : v;
    }
    void reset()
    {
        have = 0; acc = 0; nomore = false; marker = 0xff; eobrun = 0;
        for (int i = 0; i < 4; i++) comp[i].pred = 0;
        todo = restart ? restart : 0x7fffffff;
    }

    // The inverse DCT.  JPEG does not fix it bit for bit, so matching the reference's pixels means matching ITS transform: the
    // Loeffler-Ligtenberg-Moschytz factorisation in 12-bit fixed point as the Independent JPEG Group's jidctint.c has it and as
    // stb_image (public domain / MIT, Sean Barrett et al.; the reference vendors it under support/stb) scales it -- columns first,
    // kept with 2 extra bits, then rows down to the sample, + 128, clamped; a column whose AC terms are all zero is its DC term
    // times 4.  The constants, shifts and roundings below are that transform's and are pinned by tests/golden/ref_jpeg.npz; this
    // software is based in part on the work of the Independent JPEG Group.
    static int fix(float x) { return (int)(x * 4096 + 0.5); }
    static uint8_t clamp8(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
    // one 8-point pass over in[0], in[st], ...: sample k = even[k] + odd[k], sample 7 - k = even[k] - odd[k]  (k = 0..3)
    static void lm8(const int* in, int st, int even[4], int odd[4])
    {
        const int c2 = in[2 * st], c6 = in[6 * st];
        const int z = (c2 + c6) * fix(0.5411961f);
        const int lo = z + c6 * fix(-1.847759065f), hi = z + c2 * fix(0.765366865f);
        const int sum = (in[0] + in[4 * st]) * 4096, dif = (in[0] - in[4 * st]) * 4096;
        even[0] = sum + hi; even[3] = sum - hi; even[1] = dif + lo; even[2] = dif - lo;
        const int c7 = in[7 * st], c5 = in[5 * st], c3 = in[3 * st], c1 = in[st];
        const int a = c7 + c3, b = c5 + c1, c = c7 + c1, d = c5 + c3;
        const int w = (a + b) * fix(1.175875602f);
        const int pc = w + c * fix(-0.899976223f), pd = w + d * fix(-2.562915447f), pa = a * fix(-1.961570560f), pb = b * fix(-0.390180644f);
        odd[3] = c7 * fix(0.298631336f) + (pc + pa);
        odd[2] = c5 * fix(2.053119869f) + (pd + pb);
        odd[1] = c3 * fix(3.072711026f) + (pd + pa);
        odd[0] = c1 * fix(1.501321110f) + (pc + pb);
    }
    static void idct(uint8_t* out, int stride, const short* data)
    {
        int in[64], mid[64], even[4], odd[4];
        for (int i = 0; i < 64; i++) in[i] = data[i];
        for (int x = 0; x < 8; x++) {                                     // columns -> mid, scaled by 4
            const int* c = in + x;
            bool flat = true;
            for (int k = 1; k < 8; k++) if (c[8 * k]) flat = false;
            if (flat) { for (int k = 0; k < 8; k++) mid[8 * k + x] = c[0] * 4; continue; }
            lm8(c, 8, even, odd);
            for (int k = 0; k < 4; k++) { mid[8 * k + x] = (even[k] + 512 + odd[k]) >> 10; mid[8 * (7 - k) + x] = (even[k] + 512 - odd[k]) >> 10; }
        }
        for (int y = 0; y < 8; y++) {                                     // rows -> samples
            lm8(mid + 8 * y, 1, even, odd);
            uint8_t* o = out + (size_t)y * stride;
            const int bias = 65536 + (128 << 17);
            for (int k = 0; k < 4; k++) { o[k] = clamp8((even[k] + bias + odd[k]) >> 17); o[7 - k] = clamp8((even[k] + bias - odd[k]) >> 17); }
        }
    }

    bool block_baseline(short* data, Comp& C)
    {
        const Huff &dc = hdc[C.hd], &ac = hac[C.ha];
        if (!dc.ok || !ac.ok) return false;
        const int t = decode(dc);
        if (t < 0 || t > 15) return false;
        memset(data, 0, 64 * sizeof(short));
        C.pred += t ? extend(t) : 0;
        data[0] = (short)(C.pred * q[C.tq][0]);
        int k = 1;
        do {
            const int rs = decode(ac);
            if (rs < 0) return false;
            const int sz = rs & 15, r = rs >> 4;
            if (sz == 0) { if (rs != 0xf0) break; k += 16; }
            else { k += r; const int zz = zigzag()[k++]; data[zz] = (short)(extend(sz) * q[C.tq][zz]); }
        } while (k < 64);
        return true;
    }
    bool block_prog_dc(short* data, Comp& C)
    {
        if (se != 0) return false;
        if (ah == 0) {
            if (!hdc[C.hd].ok) return false;
            memset(data, 0, 64 * sizeof(short));
            const int t = decode(hdc[C.hd]);
            if (t < 0 || t > 15) return false;
            C.pred += t ? extend(t) : 0;
            data[0] = (short)(C.pred << al);
        } else if (getbit()) data[0] += (short)(1 << al);
        return true;
    }
    bool block_prog_ac(short* data, Comp& C)
    {
        if (ss == 0 || !hac[C.ha].ok) return false;
        const Huff& ac = hac[C.ha];
        if (ah == 0) {
            if (eobrun) { eobrun--; return true; }
            int k = ss;
            do {
                const int rs = decode(ac);
                if (rs < 0) return false;
                const int sz = rs & 15, r = rs >> 4;
                if (sz == 0) {
                    if (r < 15) { eobrun = (1 << r); if (r) eobrun += getbits(r); eobrun--; break; }
                    k += 16;
                } else { k += r; const int zz = zigzag()[k++]; data[zz] = (short)(extend(sz) << al); }
            } while (k <= se);
        } else {
            const short bit = (short)(1 << al);
            auto refine = [&](short* p) { if (getbit() && (*p & bit) == 0) { if (*p > 0) *p += bit; else *p -= bit; } };
            if (eobrun) {
                eobrun--;
                for (int k = ss; k <= se; k++) { short* p = &data[zigzag()[k]]; if (*p != 0) refine(p); }
            } else {
                int k = ss;
                do {
                    const int rs = decode(ac);
                    if (rs < 0) return false;
                    int sz = rs & 15, r = rs >> 4;
                    if (sz == 0) {
                        if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); r = 64; }
                    } else {
                        if (sz != 1) return false;
                        sz = getbit() ? bit : -bit;
                    }
                    while (k <= se) {
                        short* p = &data[zigzag()[k++]];
                        if (*p != 0) refine(p);
                        else { if (r == 0) { *p = (short)sz; break; } r--; }
                    }
                } while (k <= se);
            }
        }
        return true;
    }
    // the blocks of one scan; false = corrupt.  (A missing restart marker ends the scan early and keeps what there is.)
    bool scan()
    {
        reset();
        short tmp[64];
        auto one = [&](Comp& C, int bx, int by) -> bool {
            if (!progressive) {
                if (!block_baseline(tmp, C)) return false;
                idct(C.plane.data() + (size_t)C.w2 * by * 8 + (size_t)bx * 8, C.w2, tmp);
                return true;
            }
            short* data = C.coeff.data() + 64 * ((size_t)bx + (size_t)by * C.bw);
            return ss == 0 ? block_prog_dc(data, C) : block_prog_ac(data, C);
        };
        auto counted = [&]() -> int {                            // 0 go on, 1 the scan ends here
            if (--todo <= 0) {
                need(32);                                        // past the padding of the interval's last byte: the restart marker, if there is one
                if (!(marker >= 0xd0 && marker <= 0xd7)) return 1;
                reset();
            }
            return 0;
        };
        if (scan_n == 1) {
            Comp& C = comp[order[0]];
            const int w = (C.x + 7) >> 3, h = (C.y + 7) >> 3;
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) { if (!one(C, i, j)) return false; if (counted()) return true; }
            return true;
        }
        if (progressive && ss != 0) return false;                // an interleaved scan carries DC coefficients only
        for (int j = 0; j < mcuy; j++)
            for (int i = 0; i < mcux; i++) {
                for (int k = 0; k < scan_n; k++) {
                    Comp& C = comp[order[k]];
                    for (int y = 0; y < C.v; y++)
                        for (int x = 0; x < C.h; x++) if (!one(C, i * C.h + x, j * C.v + y)) return false;
                }
                if (counted()) return true;
            }
        return true;
    }
    int next_marker()
    {
        if (marker != 0xff) { const int m = marker; marker = 0xff; return m; }
        int x = get8();
        if (x != 0xff) return 0xff;
        while (x == 0xff) x = get8();
        return x;
    }
    bool segment(int m)                                           // DRI, DQT, DHT, APPn, COM
    {
        if (m == 0xff) return false;
        if (m == 0xdd) { if (get16() != 4) return false; restart = get16(); return true; }
        if (m == 0xdb) {
            int L = get16() - 2;
            while (L > 0) {
                const int b = get8(), p = b >> 4, t = b & 15;
                if ((p != 0 && p != 1) || t > 3) return false;
                for (int i = 0; i < 64; i++) q[t][zigzag()[i]] = (uint16_t)(p ? get16() : get8());
                L -= p ? 129 : 65;
            }
            return L == 0;
        }
        if (m == 0xc4) {
            int L = get16() - 2;
            while (L > 0) {
                int sizes[16], cnt = 0;
                const int b = get8(), tc = b >> 4, th = b & 15;
                if (tc > 1 || th > 3) return false;
                for (int i = 0; i < 16; i++) { sizes[i] = get8(); cnt += sizes[i]; }
                if (cnt > 256) return false;
                L -= 17;
                Huff& h = tc == 0 ? hdc[th] : hac[th];
                if (!build(h, sizes)) return false;
                for (int i = 0; i < cnt; i++) h.value[i] = (uint8_t)get8();
                for (int i = cnt; i < 256; i++) h.value[i] = 0;
                L -= cnt;
            }
            return L == 0;
        }
        if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
            int L = get16();
            if (L < 2) return false;
            L -= 2;
            if (m == 0xe0 && L >= 5) {
                static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; i++) if (get8() != tag[i]) ok = false;
                L -= 5;
                if (ok) jfif = true;
            } else if (m == 0xee && L >= 12) {
                static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; i++) if (get8() != tag[i]) ok = false;
                L -= 6;
                if (ok) { get8(); get16(); get16(); adobe = get8(); L -= 6; }
            }
            skip(L);
            return true;
        }
        return false;
    }
    bool frame_header()
    {
        const int Lf = get16();
        if (Lf < 11 || get8() != 8) return false;
        height = get16(); width = get16();
        if (height == 0 || width == 0 || height > (1 << 24) || width > (1 << 24)) return false;
        ncomp = get8();
        if (ncomp != 3 && ncomp != 1 && ncomp != 4) return false;
        if (Lf != 8 + 3 * ncomp) return false;
        rgb_ids = 0;
        for (int i = 0; i < ncomp; i++) {
            static const uint8_t rgb[3] = {'R', 'G', 'B'};
            comp[i].id = get8();
            if (ncomp == 3 && comp[i].id == rgb[i]) rgb_ids++;
            const int b = get8();
            comp[i].h = b >> 4; comp[i].v = b & 15; comp[i].tq = get8();
            if (!comp[i].h || comp[i].h > 4 || !comp[i].v || comp[i].v > 4 || comp[i].tq > 3) return false;
        }
        if ((uint64_t)width * (uint64_t)height > (1ull << 28)) return false;
        hmax = vmax = 1;
        for (int i = 0; i < ncomp; i++) { hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v); }
        // (factors that do not divide the largest one have no integral upsampling ratio: image() would read past a plane's rows --
        // a heap overflow that the vendored stb_image has too; newer ones refuse such files with "bad H" / "bad V", so does this)
        for (int i = 0; i < ncomp; i++) if (hmax % comp[i].h != 0 || vmax % comp[i].v != 0) return false;
        mcux = (width + hmax * 8 - 1) / (hmax * 8); mcuy = (height + vmax * 8 - 1) / (vmax * 8);
        for (int i = 0; i < ncomp; i++) {
            Comp& C = comp[i];
            C.x = (width * C.h + hmax - 1) / hmax; C.y = (height * C.v + vmax - 1) / vmax;
            C.w2 = mcux * C.h * 8; C.h2 = mcuy * C.v * 8; C.bw = C.w2 / 8;
            C.plane.assign((size_t)C.w2 * C.h2, 0);
            if (progressive) C.coeff.assign((size_t)C.w2 * C.h2, 0);
        }
        return true;
    }
    bool scan_header()
    {
        const int Ls = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp || Ls != 6 + 2 * scan_n) return false;
        for (int i = 0; i < scan_n; i++) {
            const int id = get8(), b = get8();
            int which = 0;
            while (which < ncomp && comp[which].id != id) which++;
            if (which == ncomp) return false;
            comp[which].hd = b >> 4; comp[which].ha = b & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) return false;
            order[i] = which;
        }
        ss = get8(); se = get8();
        const int a = get8();
        ah = a >> 4; al = a & 15;
        if (progressive) { if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return false; }
        else { if (ss != 0 || ah != 0 || al != 0) return false; se = 63; }
        return true;
    }
    bool planes()                                                 // header, all scans, and for a progressive file the final transform
    {
        if (next_marker() != 0xd8) return false;
        int m = next_marker();
        while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
            if (!segment(m)) return false;
            m = next_marker();
            while (m == 0xff) { if (pos >= n) return false; m = next_marker(); }
        }
        progressive = m == 0xc2;
        if (!frame_header()) return false;
        m = next_marker();
        while (m != 0xd9) {
            if (m == 0xda) {
                if (!scan_header() || !scan()) return false;
                if (marker == 0xff) { while (pos < n) { if (get8() == 255) { marker = get8(); break; } } }     // padding behind the scan data
            } else if (m == 0xdc) { if (get16() != 4 || get16() != height) return false; }
            else if (!segment(m)) return false;
            m = next_marker();
        }
        if (progressive)
            for (int c = 0; c < ncomp; c++) {
                Comp& C = comp[c];
                const int w = (C.x + 7) >> 3, h = (C.y + 7) >> 3;
                for (int j = 0; j < h; j++)
                    for (int i = 0; i < w; i++) {
                        short* data = C.coeff.data() + 64 * ((size_t)i + (size_t)j * C.bw);
                        for (int k = 0; k < 64; k++) data[k] = (short)(data[k] * q[C.tq][k]);
                        idct(C.plane.data() + (size_t)C.w2 * j * 8 + (size_t)i * 8, C.w2, data);
                    }
            }
        return true;
    }
    static uint8_t mul8(uint8_t a, uint8_t b) { const unsigned t = (unsigned)a * b + 128u; return (uint8_t)((t + (t >> 8)) >> 8); }
    bool image(Image& img)
    {
        if (!planes()) return false;
        const bool is_rgb = ncomp == 3 && (rgb_ids == 3 || (adobe == 0 && !jfif));
        img.w = width; img.h = height;
        img.px.assign((size_t)width * height, 0);
        struct Up { int hs, vs, ystep, wl, ypos; const uint8_t *l0, *l1; std::vector<uint8_t> buf; };
        Up up[4];
        for (int k = 0; k < ncomp; k++) {
            Up& r = up[k];
            r.hs = hmax / comp[k].h; r.vs = vmax / comp[k].v; r.ystep = r.vs >> 1; r.wl = (width + r.hs - 1) / r.hs; r.ypos = 0;
            r.l0 = r.l1 = comp[k].plane.data();
            r.buf.assign((size_t)width + 3 + 8, 0);
        }
        const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < height; j++) {
            for (int k = 0; k < ncomp; k++) {
                Up& r = up[k];
                const bool bot = r.ystep >= (r.vs >> 1);
                const uint8_t *a = bot ? r.l1 : r.l0, *b = bot ? r.l0 : r.l1;     // near, far
                uint8_t* o = r.buf.data();
                const int w = r.wl;
                if (r.hs == 1 && r.vs == 1) row[k] = a;
                else {
                    if (r.hs == 1 && r.vs == 2) for (int i = 0; i < w; i++) o[i] = (uint8_t)((3 * a[i] + b[i] + 2) >> 2);
                    else if (r.hs == 2 && r.vs == 1) {
                        if (w == 1) o[0] = o[1] = a[0];
                        else {
                            o[0] = a[0]; o[1] = (uint8_t)((a[0] * 3 + a[1] + 2) >> 2);
                            int i;
                            for (i = 1; i < w - 1; i++) { const int t = 3 * a[i] + 2; o[i * 2] = (uint8_t)((t + a[i - 1]) >> 2); o[i * 2 + 1] = (uint8_t)((t + a[i + 1]) >> 2); }
                            o[i * 2] = (uint8_t)((a[w - 2] * 3 + a[w - 1] + 2) >> 2); o[i * 2 + 1] = a[w - 1];
                        }
                    } else if (r.hs == 2 && r.vs == 2) {
                        if (w == 1) o[0] = o[1] = (uint8_t)((3 * a[0] + b[0] + 2) >> 2);
                        else {
                            int t1 = 3 * a[0] + b[0];
                            o[0] = (uint8_t)((t1 + 2) >> 2);
                            for (int i = 1; i < w; i++) {
                                const int t0 = t1;
                                t1 = 3 * a[i] + b[i];
                                o[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); o[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
                            }
                            o[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
                        }
                    } else {
                        r.buf.resize(std::max(r.buf.size(), (size_t)w * r.hs + 8)); o = r.buf.data();
                        for (int i = 0; i < w; i++) for (int s2 = 0; s2 < r.hs; s2++) o[i * r.hs + s2] = a[i];
                    }
                    row[k] = o;
                }
                if (++r.ystep >= r.vs) { r.ystep = 0; r.l0 = r.l1; if (++r.ypos < comp[k].y) r.l1 += comp[k].w2; }
            }
            uint32_t* out = img.px.data() + (size_t)j * width;
            auto ycc = [&](int i, int& R, int& G, int& B) {
                const int yf = (row[0][i] << 20) + (1 << 19), cr = row[2][i] - 128, cb = row[1][i] - 128;
                const int c1 = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, c2 = ((int)(0.71414f * 4096.0f + 0.5f)) << 8;
                const int c3 = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, c4 = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
                R = (yf + cr * c1) >> 20;
                G = (int)((unsigned)yf + (unsigned)(cr * -c2) + ((unsigned)(cb * -c3) & 0xffff0000u)) >> 20;
                B = (yf + cb * c4) >> 20;
                R = clamp8(R); G = clamp8(G); B = clamp8(B);
            };
            for (int i = 0; i < width; i++) {
                int R, G, B;
                if (ncomp == 1) R = G = B = row[0][i];
                else if (ncomp == 3) { if (is_rgb) { R = row[0][i]; G = row[1][i]; B = row[2][i]; } else ycc(i, R, G, B); }
                else if (adobe == 0) { const uint8_t kk = row[3][i]; R = mul8(row[0][i], kk); G = mul8(row[1][i], kk); B = mul8(row[2][i], kk); }
                else if (adobe == 2) { const uint8_t kk = row[3][i]; ycc(i, R, G, B); R = mul8((uint8_t)(255 - R), kk); G = mul8((uint8_t)(255 - G), kk); B = mul8((uint8_t)(255 - B), kk); }
                else ycc(i, R, G, B);
                out[i] = (uint32_t)R | ((uint32_t)G << 8) | ((uint32_t)B << 16) | 0xff000000u;
            }
        }
        return true;
    }
};

bool decode_jpeg(const std::vector<uint8_t>& d, Image& img)
{
    JpegDecoder J{d.data(), d.size()};
    memset(J.q, 0, sizeof(J.q));
    return J.image(img);
}

bool ends_with_ci(const std::string& s, const char* suffix)
{
    const size_t n = strlen(suffix);
    if (s.size() < n) return false;
    for (size_t k = 0; k < n; k++) if (tolower((unsigned char)s[s.size() - n + k]) != suffix[k]) return false;
    return true;
}

// an image file's bytes -> rgba8, mirrored along y; `name`: the file name (Truevision TGA has no signature: by extension, as the
// python loader does) or empty for an image embedded in a glTF buffer / data URI
bool decode_texture_bytes(const std::vector<uint8_t>& d, const std::string& path, Image& img)
{
    bool ok = false;
    if (d.size() >= 8 && d[0] == 0x89 && d[1] == 'P') ok = decode_png(d, img);
    else if (d.size() >= 3 && d[0] == 0xff && d[1] == 0xd8 && d[2] == 0xff) ok = decode_jpeg(d, img);
    else if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') ok = decode_ppm(d, img);
    else if (ends_with_ci(path, ".tga")) ok = decode_tga(d, img);
    if (!ok) return false;
    for (int y = 0; y < img.h / 2; y++)                                       // mirrored along y, Model.cpp:117-126
        for (int x = 0; x < img.w; x++) std::swap(img.px[(size_t)y * img.w + x], img.px[(size_t)(img.h - 1 - y) * img.w + x]);
    return true;
}
bool load_texture_file(const std::string& path, Image& img)
{
    std::vector<uint8_t> d;
    return read_file(path, d) && decode_texture_bytes(d, path, img);
}

// ---- float4 texels as stbi_loadf(file, &w, &h, &n, 4) gives them (loadProbe, PT_sv5_/main.cpp:160-171) -------------
// Radiance .hdr: "#?RADIANCE" / "#?RGBE", header lines up to the first empty one with FORMAT=32-bit_rle_rgbe among them,
// "-Y h +X w"; new-style RLE scanlines when 8 <= w < 32768 and the first scanline starts with the marker, flat RGBE
// quadruples otherwise; texel = (r, g, b) * 2^(e - 136), alpha 1, e == 0 -> (0, 0, 0, 1).  8-bit images (PNG, PPM here) go
// through stb's ldr-to-hdr conversion: colour (float)pow(c / 255.0f, 2.2f), alpha c / 255.0f.
bool decode_hdr(const std::vector<uint8_t>& d, int& w, int& h, std::vector<float>& out, std::string& why)
{
    size_t pos = 0;
    auto token = [&]() {
        size_t end = pos;
        while (end < d.size() && d[end] != '\n') end++;
        std::string line((const char*)d.data() + pos, end - pos);
        pos = std::min(d.size(), end + 1);
        return line;
    };
    const std::string first = token();
    if (first != "#?RADIANCE" && first != "#?RGBE") { why = "not a Radiance HDR file"; return false; }
    bool valid = false;
    for (;;) {
        const std::string line = token();
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") valid = true;
        if (pos >= d.size()) break;
    }
    if (!valid) { why = "unsupported HDR format (FORMAT=32-bit_rle_rgbe expected)"; return false; }
    const std::vector<std::string> parts = split_ws(token());
    if (parts.size() != 4 || parts[0] != "-Y" || parts[2] != "+X") { why = "unsupported HDR data layout (-Y h +X w expected)"; return false; }
    const long hh = strtol(parts[1].c_str(), nullptr, 10), ww = strtol(parts[3].c_str(), nullptr, 10);
    if (ww <= 0 || hh <= 0 || (uint64_t)ww * (uint64_t)hh > (1ull << 28)) { why = "bad HDR size"; return false; }
    w = (int)ww; h = (int)hh;
    const uint8_t* data = d.data() + pos;
    const size_t size = d.size() - pos, npx = (size_t)w * h;
    std::vector<uint8_t> rows;
    const uint8_t* rgbe = nullptr;
    const bool rle = !(w < 8 || w >= 32768) && !(size >= 4 && !(data[0] == 2 && data[1] == 2 && !(data[2] & 0x80)));
    if (!rle) {
        if (size < npx * 4) { why = "truncated HDR data"; return false; }
        rgbe = data;
    } else {
        rows.assign(npx * 4, 0);
        size_t p = 0;
        for (int j = 0; j < h; j++) {
            if (p + 4 > size) { why = "truncated HDR data"; return false; }
            if (data[p] != 2 || data[p + 1] != 2 || (data[p + 2] & 0x80)) { why = "a scanline is not run-length encoded"; return false; }
            if (((data[p + 2] << 8) | data[p + 3]) != w) { why = "invalid decoded scanline length"; return false; }
            p += 4;
            for (int k = 0; k < 4; k++) {
                int i = 0;
                while (i < w) {
                    if (p >= size) { why = "truncated HDR data"; return false; }
                    int count = data[p++];
                    if (count > 128) {
                        count -= 128;
                        if (count > w - i || p >= size) { why = "bad RLE data in HDR"; return false; }
                        const uint8_t v = data[p++];
                        for (int c = 0; c < count; c++) rows[((size_t)j * w + i + c) * 4 + k] = v;
                    } else {
                        if (count == 0 || count > w - i || p + count > size) { why = "bad RLE data in HDR"; return false; }
                        for (int c = 0; c < count; c++) rows[((size_t)j * w + i + c) * 4 + k] = data[p + c];
                        p += count;
                    }
                    i += count;
                }
            }
        }
        rgbe = rows.data();
    }
    out.resize(npx * 4);
    for (size_t q = 0; q < npx; q++) {
        const int e = rgbe[q * 4 + 3];
        const float f1 = e ? std::ldexp(1.0f, e - 136) : 0.0f;
        out[q * 4 + 0] = (float)rgbe[q * 4 + 0] * f1; out[q * 4 + 1] = (float)rgbe[q * 4 + 1] * f1; out[q * 4 + 2] = (float)rgbe[q * 4 + 2] * f1;
        out[q * 4 + 3] = 1.0f;
    }
    return true;
}

struct Mesh {
    std::vector<F3> vertex, normal;
    std::vector<F2> texcoord;
    std::vector<uint32_t> index;        // 3 per triangle
    fovpt_material material;
    int diffuse_texture_id = -1;
};

fovpt_material reference_default_material()     // Material.h:48-69, the values every loadOBJ mesh keeps (Model.cpp:190-191)
{
    fovpt_material m;
    memset(&m, 0, sizeof(m));
    m.emission = {1.f, 1.f, 1.f}; m.color = {1.f, 0.f, 0.f}; m.absorption = {1.f, 1.f, 1.f};
    m.eta = 1.4f; m.metallic = 0.5f; m.subsurface = 0.f; m.specular = 1.f; m.roughness = 1.f; m.specularTint = 1.f;
    m.anisotropic = 0.f; m.sheen = 0.f; m.sheenTint = 0.f; m.clearcoat = 0.f; m.clearcoatGloss = 1.f; m.transmission = 0.4f;
    m.bump = 0.f; m.bumpTile = {1.f, 1.f, 1.f}; m.flags = 0;
    return m;
}

// ---- glTF 2.0 (.gltf with external or data-URI buffers, .glb) -> meshes in world space with the traversal rules of
// sutil::Scene (sutil/Scene.cpp:109-442): roots are the nodes without a parent; a node's transform is
// parent * matrix * T * R * S in binary32 (:148, sutil::Matrix<4,4>::operator* sums k = 0..3 from 0, Matrix.h:339-355; the
// quaternion as given, sutil/Quaternion.h:239-269); a camera node is skipped with its subtree (:150-180); below a mesh node
// nothing is visited (:181-196); triangle primitives only (:327-331); base colour / roughness / metallic factors and the base
// colour texture (:281-317).  The reference keeps object-space vertices and instances them; here every primitive becomes one
// mesh in world space: p' = ((m0 x + m1 y) + m2 z) + m3 (Matrix.h:467-485 with w = 1).  Same arithmetic, in the same order, as
// loaders.load_gltf (python), which tests/test_loaders_cpu.py holds this against bit for bit.
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json* get(const char* key) const
    {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    double number(const char* key, double dflt) const { const Json* v = get(key); return v && v->kind == Num ? v->num : dflt; }
    // (a number that is not finite or not within +-2^53 has no integer value: the default, which every caller treats as absent / invalid)
    long integer(const char* key, long dflt) const
    {
        const Json* v = get(key);
        if (!v || v->kind != Num || !(v->num >= -9007199254740992.0 && v->num <= 9007199254740992.0)) return dflt;
        return (long)v->num;
    }
};
struct JsonParser {
    const char* p; const char* end; int depth = 0;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool lit(const char* t) { const size_t n = strlen(t); if ((size_t)(end - p) >= n && !memcmp(p, t, n)) { p += n; return true; } return false; }
    static void utf8(std::string& o, unsigned c)
    {
        if (c < 0x80) o += (char)c;
        else if (c < 0x800) { o += (char)(0xC0 | (c >> 6)); o += (char)(0x80 | (c & 63)); }
        else if (c < 0x10000) { o += (char)(0xE0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 63)); o += (char)(0x80 | (c & 63)); }
        else { o += (char)(0xF0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 63)); o += (char)(0x80 | ((c >> 6) & 63)); o += (char)(0x80 | (c & 63)); }
    }
    bool string(std::string& o)
    {
        if (p >= end || *p != '"') return false;
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return false;
                const char c = *p++;
                if (c == 'n') o += '\n'; else if (c == 't') o += '\t'; else if (c == 'r') o += '\r'; else if (c == 'b') o += '\b'; else if (c == 'f') o += '\f';
                else if (c == 'u') {
                    auto hex4 = [&](unsigned& v) { if (end - p < 4) return false; v = 0; for (int k = 0; k < 4; k++) { const char h = *p++; v = v * 16 + (unsigned)(h >= '0' && h <= '9' ? h - '0' : (h | 32) >= 'a' && (h | 32) <= 'f' ? (h | 32) - 'a' + 10 : 0); } return true; };
                    unsigned cp = 0;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') { p += 2; unsigned lo = 0; if (!hex4(lo)) return false; cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); }
                    utf8(o, cp);
                } else o += c;                                   // \" \\ \/
            } else o += *p++;
        }
        if (p >= end) return false;
        p++;
        return true;
    }
    bool value(Json& v)
    {
        if (++depth > 200) return false;
        ws();
        if (p >= end) return false;
        bool ok = true;
        if (*p == '{') {
            v.kind = Json::Obj; p++; ws();
            if (p < end && *p == '}') p++;
            else for (;;) {
                ws();
                std::string k;
                if (!string(k)) { ok = false; break; }
                ws();
                if (p >= end || *p++ != ':') { ok = false; break; }
                v.obj.emplace_back(std::move(k), Json());
                if (!value(v.obj.back().second)) { ok = false; break; }
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; break; }
                ok = false; break;
            }
        } else if (*p == '[') {
            v.kind = Json::Arr; p++; ws();
            if (p < end && *p == ']') p++;
            else for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back())) { ok = false; break; }
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; break; }
                ok = false; break;
            }
        } else if (*p == '"') { v.kind = Json::Str; ok = string(v.str); }
        else if (lit("true")) { v.kind = Json::Bool; v.b = true; }
        else if (lit("false")) { v.kind = Json::Bool; v.b = false; }
        else if (lit("null")) v.kind = Json::Null;
        else {
            const char* q = p;
            while (q < end && (isdigit((unsigned char)*q) || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E')) q++;
            if (q == p) ok = false;
            else { const std::string t(p, q); char* e = nullptr; v.kind = Json::Num; v.num = strtod(t.c_str(), &e); ok = e && *e == 0; p = q; }
        }
        depth--;
        return ok;
    }
};

struct M4 { float m[16]; };            // row-major, as sutil::Matrix<4,4>
M4 m4_identity() { M4 r; for (int k = 0; k < 16; k++) r.m[k] = (k % 5 == 0) ? 1.f : 0.f; return r; }
M4 m4_mul(const M4& a, const M4& b)    // Matrix.h:339-355
{
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float sum = 0.0f;
            for (int k = 0; k < 4; k++) { const float ik = a.m[i * 4 + k], kj = b.m[k * 4 + j]; sum += ik * kj; }
            r.m[i * 4 + j] = sum;
        }
    return r;
}

bool b64_decode(const std::string& in, std::vector<uint8_t>& out)
{
    unsigned acc = 0; int bits = 0;
    for (const char ch : in) {
        int v;
        if (ch >= 'A' && ch <= 'Z') v = ch - 'A'; else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26; else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
        else if (ch == '+') v = 62; else if (ch == '/') v = 63; else continue;          // '=' padding and anything else is skipped
        acc = (acc << 6) | (unsigned)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 255u)); }
    }
    return true;
}

}  // namespace

struct fovpt_model {
    std::vector<Mesh> meshes;
    std::vector<Image> textures;
};

// No C++ exception leaves the C ABI: a failed allocation (bad_alloc, length_error) or anything else thrown while a file is
// read becomes an error code and fovpt_last_error(NULL), not std::terminate in the host application.
template <typename F> static int guarded(const char* what, F&& body)
{
    try { return body(); }
    catch (const std::bad_alloc&) { fovpt_internal_set_error((std::string(what) + ": out of memory").c_str()); return FOVPT_E_NOMEM; }
    catch (const std::exception& e) { fovpt_internal_set_error((std::string(what) + ": " + e.what()).c_str()); return FOVPT_E_INVALID; }
    catch (...) { fovpt_internal_set_error((std::string(what) + ": unknown failure").c_str()); return FOVPT_E_INVALID; }
}

extern "C" {

static int model_load_obj_impl(const char* obj_file, fovpt_model** out);
int fovpt_model_load_obj(const char* obj_file, fovpt_model** out)
{
    if (out) *out = nullptr;
    return guarded("fovpt_model_load_obj", [&]() { return model_load_obj_impl(obj_file, out); });
}
static int model_load_obj_impl(const char* obj_file, fovpt_model** out)
{
    if (!obj_file || !out) { fovpt_internal_set_error("fovpt_model_load_obj: null argument"); return FOVPT_E_INVALID; }
    *out = nullptr;
    const std::string path(obj_file);
    const size_t slash = path.rfind('/');
    const std::string dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);       // :142-143
    std::vector<std::string> lines;
    if (!read_lines(path, lines)) { fovpt_internal_set_error(("Could not read OBJ model from " + path + " : cannot open file").c_str()); return FOVPT_E_INVALID; }

    std::vector<F3> pos, nrm;
    std::vector<F2> tex;
    std::vector<MtlRec> materials;
    std::map<std::string, int> mat_ids;
    struct Shape { std::vector<Corner> faces; std::vector<int> mats; };       // 3 corners and one material id per triangle
    std::vector<Shape> shapes;
    Shape cur;
    int cur_mat = -1;
    auto flush = [&]() { if (!cur.mats.empty()) shapes.push_back(std::move(cur)); cur = Shape(); };
    std::vector<Corner> cs, tris;
    for (const std::string& line : lines) {
        const std::vector<std::string> t = split_ws(line);
        if (t.empty()) continue;
        const std::string& k = t[0];
        if (k == "v" && t.size() >= 4) pos.push_back({to_f(t[1]), to_f(t[2]), to_f(t[3])});
        else if (k == "vn" && t.size() >= 4) nrm.push_back({to_f(t[1]), to_f(t[2]), to_f(t[3])});
        else if (k == "vt" && t.size() >= 2) tex.push_back({to_f(t[1]), t.size() > 2 ? to_f(t[2]) : 0.f});
        else if (k == "f") {
            cs.clear(); tris.clear();
            bool ok = true;
            for (size_t a = 1; a < t.size(); a++) {
                Corner c;
                ok = ok && parse_corner(t[a], (int)pos.size(), (int)tex.size(), (int)nrm.size(), c);
                cs.push_back(c);
            }
            if (!ok) { fovpt_internal_set_error(("Could not read OBJ model from " + path + " : bad face corner").c_str()); return FOVPT_E_INVALID; }
            triangulate(cs, pos, tris);
            for (size_t a = 0; a + 2 < tris.size(); a += 3) {
                cur.faces.push_back(tris[a]); cur.faces.push_back(tris[a + 1]); cur.faces.push_back(tris[a + 2]);
                cur.mats.push_back(cur_mat);
            }
        } else if (k == "o" || k == "g") flush();
        else if (k == "usemtl") {
            auto it = mat_ids.find(join_from(t, 1));
            cur_mat = it == mat_ids.end() ? -1 : it->second;
        } else if (k == "mtllib") {
            for (size_t a = 1; a < t.size(); a++) {
                std::vector<std::pair<std::string, MtlRec>> recs;
                parse_mtl(dir + t[a], recs);
                for (auto& r : recs)
                    if (!mat_ids.count(r.first)) { mat_ids[r.first] = (int)materials.size(); materials.push_back(r.second); }
            }
        }
    }
    flush();
    if (pos.empty()) { fovpt_internal_set_error(("Could not read OBJ model from " + path + " : no vertices").c_str()); return FOVPT_E_INVALID; }   // :160-162

    std::unique_ptr<fovpt_model> model(new fovpt_model);
    for (const Shape& shape : shapes) {
        std::map<Corner, int> known;                       // per shape, :174
        std::map<std::string, int> known_textures;         // per shape, :175
        const std::set<int> ids(shape.mats.begin(), shape.mats.end());
        for (int mid : ids) {
            Mesh mesh;
            mesh.material = reference_default_material();
            auto add_vertex = [&](const Corner& c) -> int {                     // addVertex, :49-82
                auto it = known.find(c);
                if (it != known.end()) return it->second;
                const int id = (int)mesh.vertex.size();
                known[c] = id;
                if (c.v < 0 || c.v >= (int)pos.size()) { mesh.vertex.push_back({0.f, 0.f, 0.f}); }      // (the reference reads out of bounds)
                else mesh.vertex.push_back(pos[c.v]);
                if (c.vn >= 0 && c.vn < (int)nrm.size()) while (mesh.normal.size() < mesh.vertex.size()) mesh.normal.push_back(nrm[c.vn]);
                if (c.vt >= 0 && c.vt < (int)tex.size()) while (mesh.texcoord.size() < mesh.vertex.size()) mesh.texcoord.push_back(tex[c.vt]);
                if (!mesh.texcoord.empty()) mesh.texcoord.resize(mesh.vertex.size(), F2{0.f, 0.f});
                if (!mesh.normal.empty()) mesh.normal.resize(mesh.vertex.size(), F3{0.f, 0.f, 0.f});
                return id;
            };
            for (size_t f = 0; f < shape.mats.size(); f++) {
                if (shape.mats[f] != mid) continue;
                for (int a = 0; a < 3; a++) mesh.index.push_back((uint32_t)add_vertex(shape.faces[3 * f + a]));
            }
            // (the reference sets material and texture inside the face loop, :190-201: the texture of a mesh that ends up
            // empty -- every corner already known from an earlier material of the shape -- is still loaded and keeps its id)
            mesh.diffuse_texture_id = -1;
            if (mid >= 0) {
                const MtlRec& m = materials[mid];
                mesh.material.color = {m.kd[0], m.kd[1], m.kd[2]};
                mesh.material.emission = {m.ke[0], m.ke[1], m.ke[2]};
                if (!m.map_kd.empty()) {                                     // loadTexture, :84-136
                    auto it = known_textures.find(m.map_kd);
                    if (it == known_textures.end()) {
                        std::string file = m.map_kd;
                        for (char& ch : file) if (ch == '\\') ch = '/';
                        Image img;
                        int id = -1;
                        if (load_texture_file(dir + file, img)) { id = (int)model->textures.size(); model->textures.push_back(std::move(img)); }
                        it = known_textures.emplace(m.map_kd, id).first;
                    }
                    mesh.diffuse_texture_id = it->second;
                }
            }
            if (mesh.vertex.empty()) continue;                              // :204-205
            model->meshes.push_back(std::move(mesh));
        }
    }
    *out = model.release();
    return FOVPT_OK;
}

static int model_load_gltf_impl(const char* file, fovpt_model** out);
int fovpt_model_load_gltf(const char* file, fovpt_model** out)
{
    if (out) *out = nullptr;
    return guarded("fovpt_model_load_gltf", [&]() { return model_load_gltf_impl(file, out); });
}
static int model_load_gltf_impl(const char* file, fovpt_model** out)
{
    if (!file || !out) { fovpt_internal_set_error("fovpt_model_load_gltf: null argument"); return FOVPT_E_INVALID; }
    const std::string path = file;
    auto bad = [&](const std::string& why) { fovpt_internal_set_error(("Could not read glTF scene from " + path + " : " + why).c_str()); return FOVPT_E_INVALID; };
    const size_t slash = path.find_last_of("/\\");
    const std::string dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    std::vector<uint8_t> raw;
    if (!read_file(path, raw)) return bad("cannot open file");
    std::vector<uint8_t> glb_bin;
    bool have_glb_bin = false;
    std::string text;
    if (raw.size() >= 12 && !memcmp(raw.data(), "glTF", 4)) {              // binary container: header, JSON chunk, optional BIN chunk
        auto le32 = [&](size_t o) { return (uint32_t)raw[o] | ((uint32_t)raw[o + 1] << 8) | ((uint32_t)raw[o + 2] << 16) | ((uint32_t)raw[o + 3] << 24); };
        const uint32_t version = le32(4), total = le32(8);
        if (version != 2 || total > raw.size()) return bad("glb: unsupported version or truncated file");
        size_t pos = 12;
        bool first = true;
        while (pos + 8 <= total) {
            const uint32_t clen = le32(pos), ctype = le32(pos + 4);
            const size_t body = pos + 8, avail = body <= raw.size() ? std::min<size_t>(clen, raw.size() - body) : 0;
            if (first) {
                if (ctype != 0x4E4F534Au) return bad("glb: the first chunk is not JSON");
                text.assign((const char*)raw.data() + body, avail);
                first = false;
            } else if (ctype == 0x004E4942u && !have_glb_bin) { glb_bin.assign(raw.begin() + body, raw.begin() + body + avail); have_glb_bin = true; }
            pos += 8 + (size_t)clen + ((4u - (clen & 3u)) & 3u);
        }
        if (first) return bad("glb: the first chunk is not JSON");
    } else text.assign((const char*)raw.data(), raw.size());
    Json g;
    { JsonParser P{text.data(), text.data() + text.size()}; if (!P.value(g) || g.kind != Json::Obj) return bad("not a JSON object"); }
    static const Json empty_arr = [] { Json j; j.kind = Json::Arr; return j; }();
    auto list = [&](const Json& o, const char* key) -> const Json& { const Json* v = o.get(key); return v && v->kind == Json::Arr ? *v : empty_arr; };

    // buffers
    std::vector<std::vector<uint8_t>> buffers;
    const Json& jbuf = list(g, "buffers");
    for (size_t k = 0; k < jbuf.arr.size(); k++) {
        const Json* uri = jbuf.arr[k].get("uri");
        std::vector<uint8_t> data;
        if (uri && uri->kind == Json::Str) {
            if (uri->str.compare(0, 5, "data:") == 0) {
                const size_t comma = uri->str.find(',');
                if (comma == std::string::npos) return bad("malformed data uri");
                b64_decode(uri->str.substr(comma + 1), data);
            } else if (!read_file(dir + uri->str, data)) return bad("cannot open buffer " + uri->str);
        } else if (k == 0 && have_glb_bin) data = glb_bin;
        else return bad("a buffer has no uri");
        buffers.push_back(std::move(data));
    }
    const Json &jacc = list(g, "accessors"), &jviews = list(g, "bufferViews"), &jmeshes = list(g, "meshes"), &jnodes = list(g, "nodes"),
               &jmats = list(g, "materials"), &jtex = list(g, "textures"), &jimg = list(g, "images");

    // accessor -> rows of `nc` components as float (positions, texcoords) or uint32 (indices)
    struct Acc { size_t count = 0; int nc = 0; std::vector<float> f; std::vector<uint32_t> u; };
    auto accessor = [&](long idx, bool as_index, Acc& A) -> bool {
        if (idx < 0 || (size_t)idx >= jacc.arr.size()) return false;
        const Json& a = jacc.arr[idx];
        const long ct = a.integer("componentType", 0);
        const Json* ty = a.get("type");
        if (!ty || ty->kind != Json::Str) return false;
        const int nc = ty->str == "SCALAR" ? 1 : ty->str == "VEC2" ? 2 : ty->str == "VEC3" ? 3 : ty->str == "VEC4" ? 4 : ty->str == "MAT4" ? 16 : 0;
        const size_t esz = ct == 5120 || ct == 5121 ? 1 : ct == 5122 || ct == 5123 ? 2 : ct == 5125 || ct == 5126 ? 4 : 0;
        const long cnt = a.integer("count", -1);
        if (!nc || !esz || cnt < 0 || cnt > (1l << 28)) return false;
        A.count = (size_t)cnt; A.nc = nc;
        const size_t n = A.count * (size_t)nc;
        if (as_index) A.u.assign(n, 0u); else A.f.assign(n, 0.0f);
        const Json* bvi = a.get("bufferView");
        if (!bvi || bvi->kind != Json::Num) return true;                  // no bufferView: zeros
        const long bi = (long)bvi->num;
        if (bi < 0 || (size_t)bi >= jviews.arr.size()) return false;
        const Json& bv = jviews.arr[bi];
        const long buf = bv.integer("buffer", -1);
        if (buf < 0 || (size_t)buf >= buffers.size()) return false;
        // offsets and stride come from the file: non-negative, bounded BEFORE any arithmetic, and the last element's end is checked
        // by division (a product count * stride can wrap 64 bits)
        auto field = [](const Json& o, const char* key, long maxv, long& v) {      // absent: 0; present: a whole number in [0, maxv], or the file is refused
            const Json* j = o.get(key);
            v = 0;
            if (!j) return true;
            if (j->kind != Json::Num || !(j->num >= 0.0 && j->num <= (double)maxv) || j->num != std::floor(j->num)) return false;
            v = (long)j->num;
            return true;
        };
        long off_v, off_a, stride_v;
        if (!field(bv, "byteOffset", 1l << 40, off_v) || !field(a, "byteOffset", 1l << 40, off_a) || !field(bv, "byteStride", 1l << 31, stride_v)) return false;
        const size_t off = (size_t)off_v + (size_t)off_a, elem = esz * (size_t)nc;
        const size_t stride = stride_v ? (size_t)stride_v : elem;
        const std::vector<uint8_t>& B = buffers[buf];
        if (A.count && (off > B.size() || elem > B.size() - off || (A.count - 1) > (B.size() - off - elem) / stride)) return false;
        const Json* nrm = a.get("normalized");
        const bool normalized = nrm && nrm->kind == Json::Bool && nrm->b && ct != 5126;
        for (size_t i = 0; i < A.count; i++)
            for (int c = 0; c < nc; c++) {
                const uint8_t* p = B.data() + off + i * stride + (size_t)c * esz;
                double v; uint32_t uv = 0; float fv = 0.f;
                switch (ct) {
                case 5120: v = (double)(int8_t)p[0]; uv = (uint32_t)(int32_t)(int8_t)p[0]; break;
                case 5121: v = (double)p[0]; uv = p[0]; break;
                case 5122: { int16_t x; memcpy(&x, p, 2); v = (double)x; uv = (uint32_t)(int32_t)x; } break;
                case 5123: { uint16_t x; memcpy(&x, p, 2); v = (double)x; uv = x; } break;
                case 5125: { uint32_t x; memcpy(&x, p, 4); v = (double)x; uv = x; } break;
                default: { memcpy(&fv, p, 4); v = (double)fv; uv = (uint32_t)fv; } break;
                }
                if (as_index) A.u[i * nc + c] = uv;
                else if (ct == 5126) A.f[i * nc + c] = fv;
                else if (normalized) {                                     // max(float(v) / float(max of the type), -1)
                    const float scale = ct == 5120 ? 127.f : ct == 5121 ? 255.f : ct == 5122 ? 32767.f : ct == 5123 ? 65535.f : 4294967295.f;
                    const float q = (float)v / scale;
                    A.f[i * nc + c] = q < -1.0f ? -1.0f : q;
                } else A.f[i * nc + c] = (float)v;
            }
        return true;
    };

    std::unique_ptr<fovpt_model> model(new fovpt_model);
    std::map<long, int> tex_cache;
    auto texture_id = [&](const Json* index) -> int {
        if (!index || index->kind != Json::Num) return -1;
        const long ti = (long)index->num;
        auto it = tex_cache.find(ti);
        if (it != tex_cache.end()) return it->second;
        int tid = -1;
        if (ti >= 0 && (size_t)ti < jtex.arr.size()) {
            const long src = jtex.arr[ti].integer("source", -1);
            if (src >= 0 && (size_t)src < jimg.arr.size()) {
                // an image is a file beside the scene, a data: URI, or a range of a buffer (the usual case in a .glb) -- tinygltf
                // hands all three to stb_image; so do the decoders here
                const Json* uri = jimg.arr[src].get("uri");
                const Json* view = jimg.arr[src].get("bufferView");
                Image img;
                bool ok = false;
                if (uri && uri->kind == Json::Str) {
                    if (uri->str.compare(0, 5, "data:") != 0) ok = load_texture_file(dir + uri->str, img);
                    else {
                        const size_t comma = uri->str.find(',');
                        std::vector<uint8_t> blob;
                        if (comma != std::string::npos && b64_decode(uri->str.substr(comma + 1), blob)) ok = decode_texture_bytes(blob, std::string(), img);
                    }
                } else if (view && view->kind == Json::Num && view->num >= 0 && (size_t)view->num < jviews.arr.size()) {
                    const Json& bv = jviews.arr[(size_t)view->num];
                    const long buf = bv.integer("buffer", -1), off = bv.integer("byteOffset", 0), len = bv.integer("byteLength", -1);
                    if (buf >= 0 && (size_t)buf < buffers.size() && off >= 0 && len > 0 && (size_t)off <= buffers[buf].size()
                        && (size_t)len <= buffers[buf].size() - (size_t)off) {
                        const std::vector<uint8_t> blob(buffers[buf].begin() + off, buffers[buf].begin() + off + len);
                        ok = decode_texture_bytes(blob, std::string(), img);
                    }
                }
                if (ok) { model->textures.push_back(std::move(img)); tid = (int)model->textures.size() - 1; }
            }
        }
        tex_cache[ti] = tid;
        return tid;
    };
    auto vec = [&](const Json& o, const char* key, int n, const double* dflt, double* dst) {
        const Json* v = o.get(key);
        for (int k = 0; k < n; k++) dst[k] = (v && v->kind == Json::Arr && (size_t)k < v->arr.size() && v->arr[k].kind == Json::Num) ? v->arr[k].num : dflt[k];
        return v && v->kind == Json::Arr;
    };
    auto material = [&](const Json* idx, fovpt_material& m, int& tid) {
        m = reference_default_material();
        m.emission = {0.f, 0.f, 0.f};
        tid = -1;
        if (!idx || idx->kind != Json::Num || idx->num < 0 || (size_t)idx->num >= jmats.arr.size()) return;
        const Json& gm = jmats.arr[(size_t)idx->num];
        static const Json empty_obj = [] { Json j; j.kind = Json::Obj; return j; }();
        const Json* pbrp = gm.get("pbrMetallicRoughness");
        const Json& pbr = pbrp && pbrp->kind == Json::Obj ? *pbrp : empty_obj;
        const double one4[4] = {1, 1, 1, 1}, zero3[3] = {0, 0, 0};
        double c[4], e[3];
        vec(pbr, "baseColorFactor", 4, one4, c);
        m.color = {(float)c[0], (float)c[1], (float)c[2]};
        m.roughness = (float)pbr.number("roughnessFactor", 1.0);
        m.metallic = (float)pbr.number("metallicFactor", 1.0);
        vec(gm, "emissiveFactor", 3, zero3, e);
        m.emission = {(float)e[0], (float)e[1], (float)e[2]};
        const Json* bct = pbr.get("baseColorTexture");
        if (bct && bct->kind == Json::Obj) tid = texture_id(bct->get("index"));
    };

    std::vector<char> is_root(jnodes.arr.size(), 1);
    for (const Json& n : jnodes.arr)
        for (const Json& ch : list(n, "children").arr)
            if (ch.kind == Json::Num && ch.num >= 0 && (size_t)ch.num < is_root.size()) is_root[(size_t)ch.num] = 0;
    std::string failure;
    std::function<void(const Json&, const M4&, int)> visit = [&](const Json& node, const M4& parent, int depth) {
        if (depth > 512 || !failure.empty()) return;
        const double zero3[3] = {0, 0, 0}, one3[3] = {1, 1, 1}, ident_q[4] = {0, 0, 0, 1};
        double t[3], r[4], sc[3];
        M4 T = m4_identity(), R = m4_identity(), S = m4_identity(), M = m4_identity();
        if (vec(node, "translation", 3, zero3, t)) { T.m[3] = (float)t[0]; T.m[7] = (float)t[1]; T.m[11] = (float)t[2]; }
        if (vec(node, "rotation", 4, ident_q, r)) {                       // sutil::Quaternion(w, x, y, z).rotationMatrix()
            const float qx = (float)r[0], qy = (float)r[1], qz = (float)r[2], qw = (float)r[3], one = 1.0f, two = 2.0f;
            R.m[0] = one - two * qy * qy - two * qz * qz; R.m[1] = two * qx * qy - two * qz * qw; R.m[2] = two * qx * qz + two * qy * qw;
            R.m[4] = two * qx * qy + two * qz * qw; R.m[5] = one - two * qx * qx - two * qz * qz; R.m[6] = two * qy * qz - two * qx * qw;
            R.m[8] = two * qx * qz - two * qy * qw; R.m[9] = two * qy * qz + two * qx * qw; R.m[10] = one - two * qx * qx - two * qy * qy;
        }
        if (vec(node, "scale", 3, one3, sc)) { S.m[0] = (float)sc[0]; S.m[5] = (float)sc[1]; S.m[10] = (float)sc[2]; }
        const Json* mm = node.get("matrix");
        if (mm && mm->kind == Json::Arr && mm->arr.size() == 16)
            for (int c = 0; c < 4; c++) for (int rr = 0; rr < 4; rr++) M.m[rr * 4 + c] = (float)mm->arr[c * 4 + rr].num;   // column-major in the file
        const M4 xf = m4_mul(m4_mul(m4_mul(m4_mul(parent, M), T), R), S);
        if (node.get("camera")) return;
        const Json* mi = node.get("mesh");
        if (mi && mi->kind == Json::Num) {
            if (mi->num < 0 || (size_t)mi->num >= jmeshes.arr.size()) { failure = "a node references a mesh that does not exist"; return; }
            for (const Json& prim : list(jmeshes.arr[(size_t)mi->num], "primitives").arr) {
                if (prim.integer("mode", 4) != 4) continue;
                const Json* attrs = prim.get("attributes");
                const Json* jp = attrs ? attrs->get("POSITION") : nullptr;
                if (!jp || jp->kind != Json::Num) { failure = "a primitive has no POSITION"; return; }
                Acc pos;
                if (!accessor((long)jp->num, false, pos) || pos.nc < 3) { failure = "bad POSITION accessor"; return; }
                Mesh mesh;
                mesh.vertex.resize(pos.count);
                for (size_t i = 0; i < pos.count; i++) {
                    const float x = pos.f[i * pos.nc], y = pos.f[i * pos.nc + 1], z = pos.f[i * pos.nc + 2];
                    mesh.vertex[i] = {((xf.m[0] * x + xf.m[1] * y) + xf.m[2] * z) + xf.m[3],
                                      ((xf.m[4] * x + xf.m[5] * y) + xf.m[6] * z) + xf.m[7],
                                      ((xf.m[8] * x + xf.m[9] * y) + xf.m[10] * z) + xf.m[11]};
                }
                const Json* ji = prim.get("indices");
                if (ji && ji->kind == Json::Num) {
                    Acc ia;
                    if (!accessor((long)ji->num, true, ia)) { failure = "bad index accessor"; return; }
                    mesh.index = std::move(ia.u);
                } else { mesh.index.resize(pos.count); for (size_t i = 0; i < pos.count; i++) mesh.index[i] = (uint32_t)i; }
                mesh.index.resize(mesh.index.size() / 3 * 3);
                const Json* jt = attrs->get("TEXCOORD_0");
                bool has_tc = false;
                if (jt && jt->kind == Json::Num) {
                    Acc tc;
                    if (!accessor((long)jt->num, false, tc) || tc.nc < 2) { failure = "bad TEXCOORD_0 accessor"; return; }
                    // (one texcoord per vertex, or none: a shorter array would be read past its end by whoever indexes it with a
                    // vertex index -- fovpt_set_scene does)
                    if (tc.count == pos.count) {
                        mesh.texcoord.resize(tc.count);
                        for (size_t i = 0; i < tc.count; i++) mesh.texcoord[i] = {tc.f[i * tc.nc], tc.f[i * tc.nc + 1]};
                        has_tc = true;
                    }
                }
                int tid = -1;
                material(prim.get("material"), mesh.material, tid);
                mesh.diffuse_texture_id = has_tc ? tid : -1;
                model->meshes.push_back(std::move(mesh));
            }
            return;                                                       // sutil does not descend below a mesh node
        }
        for (const Json& ch : list(node, "children").arr)
            if (ch.kind == Json::Num && ch.num >= 0 && (size_t)ch.num < jnodes.arr.size()) visit(jnodes.arr[(size_t)ch.num], xf, depth + 1);
    };
    for (size_t i = 0; i < jnodes.arr.size(); i++)
        if (is_root[i]) visit(jnodes.arr[i], m4_identity(), 0);
    if (!failure.empty()) return bad(failure);
    *out = model.release();
    return FOVPT_OK;
}

void fovpt_model_destroy(fovpt_model* m) { delete m; }

int fovpt_model_counts(const fovpt_model* m, int* num_meshes, int* num_textures)
{
    if (!m) return FOVPT_E_INVALID;
    if (num_meshes) *num_meshes = (int)m->meshes.size();
    if (num_textures) *num_textures = (int)m->textures.size();
    return FOVPT_OK;
}

int fovpt_model_get_mesh(const fovpt_model* m, int i, fovpt_model_mesh* out)
{
    if (!m || !out || i < 0 || i >= (int)m->meshes.size()) return FOVPT_E_INVALID;
    const Mesh& s = m->meshes[i];
    out->vertex = (const fovpt_float3*)s.vertex.data();
    out->normal = s.normal.empty() ? nullptr : (const fovpt_float3*)s.normal.data();
    out->texcoord = s.texcoord.empty() ? nullptr : (const float*)s.texcoord.data();
    out->index = (const fovpt_uint3*)s.index.data();
    out->num_vertices = (uint32_t)s.vertex.size(); out->num_normals = (uint32_t)s.normal.size();
    out->num_texcoords = (uint32_t)s.texcoord.size(); out->num_triangles = (uint32_t)(s.index.size() / 3);
    out->material = s.material;
    out->diffuse_texture_id = s.diffuse_texture_id;
    return FOVPT_OK;
}

int fovpt_model_get_texture(const fovpt_model* m, int i, const uint32_t** pixels, int* width, int* height)
{
    if (!m || i < 0 || i >= (int)m->textures.size()) return FOVPT_E_INVALID;
    if (pixels) *pixels = m->textures[i].px.data();
    if (width) *width = m->textures[i].w;
    if (height) *height = m->textures[i].h;
    return FOVPT_OK;
}

static int image_load_float4_impl(const char* file, int* width, int* height, fovpt_float4** texels);
int fovpt_image_load_float4(const char* file, int* width, int* height, fovpt_float4** texels)
{
    if (texels) *texels = nullptr;
    return guarded("fovpt_image_load_float4", [&]() { return image_load_float4_impl(file, width, height, texels); });
}
static int image_load_float4_impl(const char* file, int* width, int* height, fovpt_float4** texels)
{
    if (!file || !width || !height || !texels) { fovpt_internal_set_error("fovpt_image_load_float4: null argument"); return FOVPT_E_INVALID; }
    *texels = nullptr;
    std::vector<uint8_t> d;
    if (!read_file(file, d)) { fovpt_internal_set_error((std::string("Could not read image ") + file).c_str()); return FOVPT_E_INVALID; }
    std::vector<float> px;
    int w = 0, h = 0;
    if (d.size() >= 6 && (!memcmp(d.data(), "#?RADIANCE", std::min<size_t>(10, d.size())) || !memcmp(d.data(), "#?RGBE", 6))) {
        std::string why;
        if (!decode_hdr(d, w, h, px, why)) { fovpt_internal_set_error((std::string(file) + ": " + why).c_str()); return FOVPT_E_INVALID; }
    } else {
        Image img;
        bool ok = false;
        if (d.size() >= 8 && d[0] == 0x89 && d[1] == 'P') ok = decode_png(d, img);
        else if (d.size() >= 3 && d[0] == 0xff && d[1] == 0xd8 && d[2] == 0xff) ok = decode_jpeg(d, img);
        else if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') ok = decode_ppm(d, img);
        else if (ends_with_ci(file, ".tga")) ok = decode_tga(d, img);
        if (!ok) { fovpt_internal_set_error((std::string(file) + ": not an image this loader reads (.hdr, .png, .tga, binary .ppm)").c_str()); return FOVPT_E_INVALID; }
        w = img.w; h = img.h;
        px.resize((size_t)w * h * 4);
        for (size_t q = 0; q < (size_t)w * h; q++) {                         // stbi__ldr_to_hdr: gamma 2.2, scale 1
            const uint32_t c = img.px[q];
            for (int k = 0; k < 3; k++) px[q * 4 + k] = (float)(std::pow((double)((float)((c >> (8 * k)) & 255u) / 255.0f), (double)2.2f) * 1.0f);
            px[q * 4 + 3] = (float)(c >> 24) / 255.0f;
        }
    }
    fovpt_float4* out = (fovpt_float4*)malloc(sizeof(fovpt_float4) * (size_t)w * h);
    if (!out) { fovpt_internal_set_error("fovpt_image_load_float4: out of memory"); return FOVPT_E_NOMEM; }
    memcpy(out, px.data(), sizeof(float) * px.size());
    *width = w; *height = h; *texels = out;
    return FOVPT_OK;
}

// stbi_load(file, &w, &h, &n, STBI_rgb_alpha) for the formats this library reads (PNG, JPEG, TGA, binary PPM): rgba8, row 0 first
// (NOT mirrored: loadTexture does that itself, Model.cpp:117-126).  *pixels is malloc'ed; release it with fovpt_image_free_rgba8.
int fovpt_image_load_rgba8(const char* file, int* width, int* height, uint32_t** pixels)
{
    if (pixels) *pixels = nullptr;
    return guarded("fovpt_image_load_rgba8", [&]() -> int {
        if (!file || !width || !height || !pixels) { fovpt_internal_set_error("fovpt_image_load_rgba8: null argument"); return FOVPT_E_INVALID; }
        Image img;
        if (!load_texture_file(file, img)) {
            fovpt_internal_set_error((std::string(file) + ": cannot open, or not an image this loader reads (.png, .jpg, .tga, binary .ppm)").c_str());
            return FOVPT_E_INVALID;
        }
        uint32_t* out = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)img.w * img.h);
        if (!out) { fovpt_internal_set_error("fovpt_image_load_rgba8: out of memory"); return FOVPT_E_NOMEM; }
        for (int y = 0; y < img.h; y++)                              // load_texture_file mirrors along y: undo it
            memcpy(out + (size_t)y * img.w, img.px.data() + (size_t)(img.h - 1 - y) * img.w, sizeof(uint32_t) * (size_t)img.w);
        *width = img.w; *height = img.h; *pixels = out;
        return FOVPT_OK;
    });
}
void fovpt_image_free_rgba8(uint32_t* pixels) { free(pixels); }

void fovpt_image_free(fovpt_float4* texels) { free(texels); }

}  // extern "C"
