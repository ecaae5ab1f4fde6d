// obj_model.cpp -- see obj_model.h.  Behaviour follows tinyobjloader as vendored upstream
// (third/tiny_obj_loader.h: LoadObj :2584-3155, exportGroupsToShape :1482-2025, LoadMtl :2069-2467,
// tryParseDouble :897-1028, parseTriple :1188-1239, fixIndex :819-852), restated for the fields the scene
// loader reads.
#include "obj_model.h"

#include "file_util.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <set>
#include <sstream>

namespace rth {

namespace {

inline bool is_blank(char c) { return c == ' ' || c == '\t'; }
inline bool is_digit(char c) { return static_cast<unsigned>(c - '0') < 10u; }
inline bool at_line_end(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// One line of input with a read position.  "Field" = a run of characters up to blank or '\r'.
struct Cursor {
    const char* p;
    explicit Cursor(const char* s) : p(s) {}
    void skip_blanks() { p += std::strspn(p, " \t"); }
    void skip_blanks_cr() { p += std::strspn(p, " \t\r"); }
    const char* field_end() const { return p + std::strcspn(p, " \t\r"); }
    std::string word() {
        skip_blanks();
        const char* e = field_end();
        std::string s(p, e);
        p = e;
        return s;
    }
    int integer() {
        skip_blanks();
        const int i = std::atoi(p);
        p = field_end();
        return i;
    }
    float real(double fallback = 0.0) {
        skip_blanks();
        const char* e = field_end();
        double v = fallback;
        parse_decimal(p, e, &v);
        p = e;
        return static_cast<float>(v);
    }
};

// Reads one line, accepting LF, CRLF and lone CR endings; the last line may lack one.
bool next_line(std::istream& in, std::string& line) {
    line.clear();
    std::streambuf* sb = in.rdbuf();
    if (sb->sgetc() == std::char_traits<char>::eof()) return false;
    for (;;) {
        const int c = sb->sbumpc();
        if (c == '\n') return true;
        if (c == '\r') {
            if (sb->sgetc() == '\n') sb->sbumpc();
            return true;
        }
        if (c == std::char_traits<char>::eof()) return true;
        line += static_cast<char>(c);
    }
}

struct Corner { int v, vt, vn; };

struct Face { std::vector<Corner> corners; };

// What has been read since the last flush into a shape.
struct Pending {
    std::vector<Face> faces;
    std::vector<size_t> line_lengths;     // only their presence matters to the scene loader
    size_t n_point_indices = 0;
    bool empty() const { return faces.empty() && line_lengths.empty() && n_point_indices == 0; }
    void clear() { faces.clear(); line_lengths.clear(); n_point_indices = 0; }
};

struct ShapeBuild {
    ObjShape shape;
    size_t n_line_indices = 0, n_point_indices = 0;
};

// OBJ index -> zero-based.  Zero is tolerated (as -1) only for normals and texture coordinates.
bool resolve_index(int idx, int count, bool zero_ok, int* out, std::string& warn, size_t line_no) {
    if (idx > 0) { *out = idx - 1; return true; }
    if (idx == 0) {
        warn += "A zero value index found (will have a value of -1 for normal and tex indices. Line " + std::to_string(line_no) + ").\n";
        *out = -1;
        return zero_ok;
    }
    *out = count + idx;
    return *out >= 0;
}

// i, i/j, i//k, i/j/k
bool parse_corner(Cursor& c, int nv, int nvn, int nvt, Corner* out, std::string& warn, size_t line_no) {
    Corner k = {-1, -1, -1};
    if (!resolve_index(std::atoi(c.p), nv, false, &k.v, warn, line_no)) return false;
    c.p += std::strcspn(c.p, "/ \t\r");
    if (*c.p != '/') { *out = k; return true; }
    c.p++;
    if (*c.p == '/') {
        c.p++;
        if (!resolve_index(std::atoi(c.p), nvn, true, &k.vn, warn, line_no)) return false;
        c.p += std::strcspn(c.p, "/ \t\r");
        *out = k;
        return true;
    }
    if (!resolve_index(std::atoi(c.p), nvt, true, &k.vt, warn, line_no)) return false;
    c.p += std::strcspn(c.p, "/ \t\r");
    if (*c.p != '/') { *out = k; return true; }
    c.p++;
    if (!resolve_index(std::atoi(c.p), nvn, true, &k.vn, warn, line_no)) return false;
    c.p += std::strcspn(c.p, "/ \t\r");
    *out = k;
    return true;
}

void emit_triangle(ObjShape& s, const Corner& a, const Corner& b, const Corner& c, int material) {
    s.indices.push_back({a.v, a.vn, a.vt});
    s.indices.push_back({b.v, b.vn, b.vt});
    s.indices.push_back({c.v, c.vn, c.vt});
    s.material_ids.push_back(material);
}

// Even-odd point-in-polygon test on the triangle (x[], y[]).
bool inside_triangle(const float* x, const float* y, float px, float py) {
    bool in = false;
    for (int i = 0, j = 2; i < 3; j = i++) {
        if (((y[i] > py) != (y[j] > py)) && (px < (x[j] - x[i]) * (py - y[i]) / (y[j] - y[i]) + x[i])) in = !in;
    }
    return in;
}

// Ear clipping of a polygon with five or more corners, in the plane of its two dominant axes.
void clip_polygon(ObjShape& s, const Face& face, int material, const std::vector<float>& v) {
    size_t n = face.corners.size();
    size_t ax0 = 1, ax1 = 2;
    for (size_t k = 0; k < n; ++k) {
        const size_t i0 = size_t(face.corners[k % n].v), i1 = size_t(face.corners[(k + 1) % n].v), i2 = size_t(face.corners[(k + 2) % n].v);
        if (3 * i0 + 2 >= v.size() || 3 * i1 + 2 >= v.size() || 3 * i2 + 2 >= v.size()) continue;
        const float e0x = v[3 * i1] - v[3 * i0], e0y = v[3 * i1 + 1] - v[3 * i0 + 1], e0z = v[3 * i1 + 2] - v[3 * i0 + 2];
        const float e1x = v[3 * i2] - v[3 * i1], e1y = v[3 * i2 + 1] - v[3 * i1 + 1], e1z = v[3 * i2 + 2] - v[3 * i1 + 2];
        const float cx = std::fabs(e0y * e1z - e0z * e1y);
        const float cy = std::fabs(e0z * e1x - e0x * e1z);
        const float cz = std::fabs(e0x * e1y - e0y * e1x);
        const float eps = std::numeric_limits<float>::epsilon();
        if (cx > eps || cy > eps || cz > eps) {          // first real corner decides the projection
            if (!(cx > cy && cx > cz)) {
                ax0 = 0;
                if (cz > cx && cz > cy) ax1 = 1;
            }
            break;
        }
    }
    std::vector<Corner> rest = face.corners;
    size_t guess = 0, tries_left = n, last_size = n;
    while (rest.size() > 3 && tries_left > 0) {
        n = rest.size();
        if (guess >= n) guess -= n;
        if (last_size != n) { last_size = n; tries_left = n; }
        else tries_left--;
        Corner tri[3];
        float x[3], y[3];
        for (size_t k = 0; k < 3; k++) {
            tri[k] = rest[(guess + k) % n];
            const size_t vi = size_t(tri[k].v);
            if (3 * vi + ax0 >= v.size() || 3 * vi + ax1 >= v.size()) { x[k] = 0.0f; y[k] = 0.0f; }
            else { x[k] = v[3 * vi + ax0]; y[k] = v[3 * vi + ax1]; }
        }
        const float e0x = x[1] - x[0], e0y = y[1] - y[0], e1x = x[2] - x[1], e1y = y[2] - y[1];
        const float turn = e0x * e1y - e0y * e1x;
        const float area = (x[0] * y[1] - y[0] * x[1]) * 0.5f;
        if (turn * area < 0.0f) { guess += 1; continue; }     // reflex corner
        bool blocked = false;
        for (size_t other = 3; other < n; ++other) {
            const size_t idx = (guess + other) % n;
            if (idx >= rest.size()) continue;
            const size_t ovi = size_t(rest[idx].v);
            if (3 * ovi + ax0 >= v.size() || 3 * ovi + ax1 >= v.size()) continue;
            if (inside_triangle(x, y, v[3 * ovi + ax0], v[3 * ovi + ax1])) { blocked = true; break; }
        }
        if (blocked) { guess += 1; continue; }
        emit_triangle(s, tri[0], tri[1], tri[2], material);
        rest.erase(rest.begin() + long((guess + 1) % n));
    }
    if (rest.size() == 3) emit_triangle(s, rest[0], rest[1], rest[2], material);
}

// Moves what is pending into the shape under construction (triangulating); false if nothing was pending.
bool flush_pending(ShapeBuild& sb, const Pending& pend, int material, const std::string& name, const std::vector<float>& v,
                   std::string& warn) {
    if (pend.empty()) return false;
    sb.shape.name = name;
    for (const Face& f : pend.faces) {
        const size_t n = f.corners.size();
        if (n < 3) { warn += "Degenerated face found\n."; continue; }
        if (n == 3) {
            emit_triangle(sb.shape, f.corners[0], f.corners[1], f.corners[2], material);
        } else if (n == 4) {
            const Corner &c0 = f.corners[0], &c1 = f.corners[1], &c2 = f.corners[2], &c3 = f.corners[3];
            const size_t i0 = size_t(c0.v), i1 = size_t(c1.v), i2 = size_t(c2.v), i3 = size_t(c3.v);
            if (3 * i0 + 2 >= v.size() || 3 * i1 + 2 >= v.size() || 3 * i2 + 2 >= v.size() || 3 * i3 + 2 >= v.size()) {
                warn += "Face with invalid vertex index found.\n";
                continue;
            }
            // split along the shorter diagonal
            const float ax = v[3 * i2] - v[3 * i0], ay = v[3 * i2 + 1] - v[3 * i0 + 1], az = v[3 * i2 + 2] - v[3 * i0 + 2];
            const float bx = v[3 * i3] - v[3 * i1], by = v[3 * i3 + 1] - v[3 * i1 + 1], bz = v[3 * i3 + 2] - v[3 * i1 + 2];
            const float d02 = ax * ax + ay * ay + az * az, d13 = bx * bx + by * by + bz * bz;
            if (d02 < d13) {
                emit_triangle(sb.shape, c0, c1, c2, material);
                emit_triangle(sb.shape, c0, c2, c3, material);
            } else {
                emit_triangle(sb.shape, c0, c1, c3, material);
                emit_triangle(sb.shape, c1, c2, c3, material);
            }
        } else {
            clip_polygon(sb.shape, f, material, v);
        }
    }
    for (size_t len : pend.line_lengths) sb.n_line_indices += len;
    sb.n_point_indices += pend.n_point_indices;
    return true;
}

void default_material(ObjMaterial& m) {
    m.name.clear();
    m.diffuse_texname.clear();
    for (int i = 0; i < 3; i++) { m.diffuse[i] = 0.0f; m.emission[i] = 0.0f; }
    m.illum = 0;
    m.dissolve = 1.0f;
    m.ior = 1.0f;
}

bool starts(const char* p, const char* word) { return std::strncmp(p, word, std::strlen(word)) == 0; }

// "map_Kd [-option args]... file name to the end of the line"
void parse_texture_line(const char* text, std::string* texname) {
    Cursor c(text);
    bool found = false;
    std::string name;
    while (!at_line_end(*c.p)) {
        c.skip_blanks();
        if ((starts(c.p, "-blendu") || starts(c.p, "-blendv")) && is_blank(c.p[7])) { c.p += 8; c.word(); }
        else if (starts(c.p, "-clamp") && is_blank(c.p[6])) { c.p += 7; c.word(); }
        else if (starts(c.p, "-boost") && is_blank(c.p[6])) { c.p += 7; c.real(); }
        else if (starts(c.p, "-bm") && is_blank(c.p[3])) { c.p += 4; c.real(); }
        else if ((starts(c.p, "-o") || starts(c.p, "-s") || starts(c.p, "-t")) && is_blank(c.p[2])) { c.p += 3; c.real(); c.real(); c.real(); }
        else if (starts(c.p, "-type") && is_blank(c.p[5])) { c.p += 5; c.word(); }
        else if (starts(c.p, "-texres") && is_blank(c.p[7])) { c.p += 7; c.integer(); }
        else if (starts(c.p, "-imfchan") && is_blank(c.p[8])) { c.p += 9; c.word(); }
        else if (starts(c.p, "-mm") && is_blank(c.p[3])) { c.p += 4; c.real(); c.real(); }
        else if (starts(c.p, "-colorspace") && is_blank(c.p[11])) { c.p += 12; c.word(); }
        else {
            name = std::string(c.p);          // file names may contain blanks: everything up to the end
            c.p += name.size();
            found = true;
        }
    }
    if (found) *texname = name;
}

void load_mtl(std::istream& in, std::vector<ObjMaterial>& materials, std::map<std::string, int>& by_name, std::string& warn) {
    ObjMaterial m;
    default_material(m);
    bool has_d = false, has_tr = false;
    bool has_kd = false;          // never reset: once any material of the file had Kd, map_Kd stops supplying the 0.6 default
    std::string line;
    size_t line_no = 0;
    while (next_line(in, line)) {
        line_no++;
        if (!line.empty()) line = line.substr(0, line.find_last_not_of(" \t") + 1);
        if (line.empty()) continue;
        Cursor c(line.c_str());
        c.skip_blanks();
        if (*c.p == '\0' || *c.p == '#') continue;
        const char* t = c.p;
        if (starts(t, "newmtl") && is_blank(t[6])) {
            if (!m.name.empty()) {
                by_name.insert(std::make_pair(m.name, int(materials.size())));
                materials.push_back(m);
            }
            default_material(m);
            has_d = has_tr = false;
            c.p += 7;
            m.name = c.word();
            if (m.name.empty()) warn += "empty material name in `newmtl`\n";
        } else if (t[0] == 'K' && t[1] == 'd' && is_blank(t[2])) {
            c.p += 2;
            m.diffuse[0] = c.real(); m.diffuse[1] = c.real(); m.diffuse[2] = c.real();
            has_kd = true;
        } else if (t[0] == 'N' && t[1] == 'i' && is_blank(t[2])) {
            c.p += 2;
            m.ior = c.real();
        } else if (t[0] == 'K' && t[1] == 'e' && is_blank(t[2])) {
            c.p += 2;
            m.emission[0] = c.real(); m.emission[1] = c.real(); m.emission[2] = c.real();
        } else if (starts(t, "illum") && is_blank(t[5])) {
            c.p += 6;
            m.illum = c.integer();
        } else if (t[0] == 'd' && is_blank(t[1])) {
            c.p += 1;
            m.dissolve = c.real();
            if (has_tr) warn += "Both `d` and `Tr` parameters defined for \"" + m.name + "\". Use the value of `d` for dissolve (line " + std::to_string(line_no) + " in .mtl.)\n";
            has_d = true;
        } else if (t[0] == 'T' && t[1] == 'r' && is_blank(t[2])) {
            c.p += 2;
            if (has_d) warn += "Both `d` and `Tr` parameters defined for \"" + m.name + "\". Use the value of `d` for dissolve (line " + std::to_string(line_no) + " in .mtl.)\n";
            else m.dissolve = 1.0f - c.real();         // `d` wins when both are given
            has_tr = true;
        } else if (starts(t, "map_Kd") && is_blank(t[6])) {
            parse_texture_line(t + 7, &m.diffuse_texname);
            if (!has_kd) m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 0.6f;
        }
        // every other statement (Ka, Ks, Ns, other maps, PBR extensions ...) does not reach the scene loader
    }
    // the last material is stored even when it has no name (an empty file yields one default material)
    by_name.insert(std::make_pair(m.name, int(materials.size())));
    materials.push_back(m);
}

bool load_mtl_file(const std::string& name, const std::string& search, std::vector<ObjMaterial>& materials,
                   std::map<std::string, int>& by_name, std::string& warn) {
    std::string mtl_warn;
    if (!search.empty()) {
        std::istringstream dirs(search);
        std::string dir;
        while (std::getline(dirs, dir, ':')) {
            const std::string path = dir.empty() ? name : (dir.back() == '/' ? dir + name : dir + "/" + name);
            if (!is_plain_file(path)) continue;
            std::ifstream f(path.c_str());
            if (f) { load_mtl(f, materials, by_name, mtl_warn); warn += mtl_warn; return true; }
        }
    } else {
        std::ifstream f(name.c_str());
        if (is_plain_file(name) && f) { load_mtl(f, materials, by_name, mtl_warn); warn += mtl_warn; return true; }
    }
    warn += "Material file [ " + name + " ] not found in a path : " + search + "\n";
    return false;
}

// split on blanks, backslash escapes the next character
std::vector<std::string> split_escaped(const std::string& s) {
    std::vector<std::string> out;
    std::string tok;
    bool esc = false;
    for (char ch : s) {
        if (esc) { tok += ch; esc = false; }
        else if (ch == '\\') esc = true;
        else if (ch == ' ') { if (!tok.empty()) out.push_back(tok); tok.clear(); }
        else tok += ch;
    }
    out.push_back(tok);          // the last piece counts even when it is empty (a trailing blank)
    return out;
}

}  // namespace

bool parse_decimal(const char* s, const char* s_end, double* out) {
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    bool negative = false, leading_dot = false;
    const char* p = s;
    if (*p == '+' || *p == '-') {
        negative = *p == '-';
        p++;
        if (p != s_end && *p == '.') leading_dot = true;
    } else if (*p == '.') {
        leading_dot = true;
    } else if (!is_digit(*p)) {
        return false;
    }
    if (!leading_dot) {
        int n = 0;
        while (p != s_end && is_digit(*p)) { mantissa = mantissa * 10 + (*p - '0'); p++; n++; }
        if (n == 0) return false;
    }
    bool more = p != s_end;
    if (more && *p == '.') {
        p++;
        int place = 1;
        static const double tenth[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
        while (p != s_end && is_digit(*p)) {
            mantissa += (*p - '0') * (place < 8 ? tenth[place] : std::pow(10.0, -place));
            place++;
            p++;
        }
        more = p != s_end;
    } else if (more && !(*p == 'e' || *p == 'E')) {
        more = false;
    }
    if (more && (*p == 'e' || *p == 'E')) {
        p++;
        bool exp_negative = false;
        if (p != s_end && (*p == '+' || *p == '-')) { exp_negative = *p == '-'; p++; }
        else if (!is_digit(*p)) return false;                 // a bare "e" is not a number
        int n = 0;
        while (p != s_end && is_digit(*p)) {
            if (exponent > 2147483647 / 10) return false;
            exponent = exponent * 10 + (*p - '0');
            p++; n++;
        }
        if (exp_negative) exponent = -exponent;
        if (n == 0) return false;
    }
    // 10^e as 5^e * 2^e, the power of two applied exactly
    const double magnitude = exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa;
    *out = negative ? -magnitude : magnitude;
    return true;
}

bool load_obj(const std::string& filename, const std::string& mtl_search_path, ObjModel& model) {
    model = ObjModel();
    std::ifstream in(filename.c_str());
    if (!in || !is_plain_file(filename)) {
        model.error = "Cannot open file [" + filename + "]\n";
        return false;
    }
    std::string search = mtl_search_path;
    if (search.empty()) {
        const size_t pos = filename.find_last_of("/\\");
        if (pos != std::string::npos) search = filename.substr(0, pos);
    }
    if (!search.empty() && search.back() != '/') search += '/';

    std::vector<float>& v = model.vertices;
    std::vector<float>& vn = model.normals;
    std::vector<float>& vt = model.texcoords;
    std::map<std::string, int> material_by_name;
    std::set<std::string> loaded_mtl;
    Pending pend;
    ShapeBuild cur;
    std::string name;
    int material = -1;
    int top_v = -1, top_vn = -1, top_vt = -1;
    std::string line;
    size_t line_no = 0;

    auto shape_has_content = [](const ShapeBuild& s, bool count_lines_and_points) {
        return !s.shape.indices.empty() || (count_lines_and_points && (s.n_line_indices > 0 || s.n_point_indices > 0));
    };

    while (next_line(in, line)) {
        line_no++;
        if (line.empty()) continue;
        Cursor c(line.c_str());
        c.skip_blanks();
        const char* t = c.p;
        if (t[0] == '\0' || t[0] == '#') continue;

        if (t[0] == 'v' && is_blank(t[1])) {
            c.p += 2;
            const float x = c.real(), y = c.real(), z = c.real();
            v.push_back(x); v.push_back(y); v.push_back(z);
            continue;
        }
        if (t[0] == 'v' && t[1] == 'n' && is_blank(t[2])) {
            c.p += 3;
            const float x = c.real(), y = c.real(), z = c.real();
            vn.push_back(x); vn.push_back(y); vn.push_back(z);
            continue;
        }
        if (t[0] == 'v' && t[1] == 't' && is_blank(t[2])) {
            c.p += 3;
            const float x = c.real(), y = c.real();
            vt.push_back(x); vt.push_back(y);
            continue;
        }
        if ((t[0] == 'l' || t[0] == 'p' || t[0] == 'f') && is_blank(t[1])) {
            const char kind = t[0];
            c.p += 2;
            if (kind == 'f') c.skip_blanks();
            Face face;
            while (!at_line_end(*c.p)) {
                Corner k;
                if (!parse_corner(c, int(v.size() / 3), int(vn.size() / 3), int(vt.size() / 2), &k, model.warning, line_no)) {
                    model.error += std::string("Failed to parse `") + kind + "' line (e.g. a zero value for vertex index" +
                                   (kind == 'f' ? " or invalid relative vertex index). Line " : ". Line ") + std::to_string(line_no) + ").\n";
                    return false;
                }
                if (kind == 'f') {
                    top_v = std::max(top_v, k.v); top_vn = std::max(top_vn, k.vn); top_vt = std::max(top_vt, k.vt);
                }
                face.corners.push_back(k);
                c.skip_blanks_cr();
            }
            if (kind == 'f') pend.faces.push_back(face);
            else if (kind == 'l') pend.line_lengths.push_back(face.corners.size());
            else pend.n_point_indices += face.corners.size();
            continue;
        }
        if (starts(t, "usemtl")) {
            c.p += 6;
            const std::string mname = c.word();
            int id = -1;
            auto it = material_by_name.find(mname);
            if (it != material_by_name.end()) id = it->second;
            else model.warning += "material [ '" + mname + "' ] not found in .mtl\n";
            if (id != material) {
                // faces read so far keep the old material; the shape itself goes on
                flush_pending(cur, pend, material, name, v, model.warning);
                pend.faces.clear();
                material = id;
            }
            continue;
        }
        if (starts(t, "mtllib") && is_blank(t[6])) {
            const std::vector<std::string> files = split_escaped(std::string(t + 7));
            if (files.empty()) {
                model.warning += "Looks like empty filename for mtllib. Use default material (line " + std::to_string(line_no) + ".)\n";
            } else {
                bool found = false;
                for (const std::string& f : files) {
                    if (loaded_mtl.count(f)) { found = true; continue; }
                    if (load_mtl_file(f, search, model.materials, material_by_name, model.warning)) {
                        found = true;
                        loaded_mtl.insert(f);
                        break;
                    }
                }
                if (!found) model.warning += "Failed to load material file(s). Use default material.\n";
            }
            continue;
        }
        if (t[0] == 'g' && is_blank(t[1])) {
            flush_pending(cur, pend, material, name, v, model.warning);
            if (shape_has_content(cur, false)) model.shapes.push_back(cur.shape);
            cur = ShapeBuild();
            pend.clear();
            std::vector<std::string> words;          // the first word is the "g" itself
            while (!at_line_end(*c.p)) {
                words.push_back(c.word());
                c.skip_blanks_cr();
            }
            if (words.size() < 2) {
                model.warning += "Empty group name. line: " + std::to_string(line_no) + "\n";
                name.clear();
            } else {
                name = words[1];
                for (size_t i = 2; i < words.size(); i++) name += " " + words[i];
            }
            continue;
        }
        if (t[0] == 'o' && is_blank(t[1])) {
            flush_pending(cur, pend, material, name, v, model.warning);
            if (shape_has_content(cur, true)) model.shapes.push_back(cur.shape);
            pend.clear();
            cur = ShapeBuild();
            name = std::string(t + 2);
            continue;
        }
        // vw, t, s and unknown statements carry nothing the scene loader reads
    }

    if (top_v >= int(v.size() / 3)) model.warning += "Vertex indices out of bounds (line " + std::to_string(line_no) + ".)\n\n";
    if (top_vn >= int(vn.size() / 3)) model.warning += "Vertex normal indices out of bounds (line " + std::to_string(line_no) + ".)\n\n";
    if (top_vt >= int(vt.size() / 2)) model.warning += "Vertex texcoord indices out of bounds (line " + std::to_string(line_no) + ".)\n\n";

    const bool flushed = flush_pending(cur, pend, material, name, v, model.warning);
    if (flushed || !cur.shape.indices.empty()) model.shapes.push_back(cur.shape);
    return true;
}

std::string dump_obj(const ObjModel& m) {
    std::ostringstream o;
    char buf[160];
    auto hexf = [&](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        std::snprintf(buf, sizeof buf, "%08x", u);
        return std::string(buf);
    };
    o << "vertices " << m.vertices.size() / 3 << "\n";
    for (size_t i = 0; i + 2 < m.vertices.size(); i += 3) o << hexf(m.vertices[i]) << " " << hexf(m.vertices[i + 1]) << " " << hexf(m.vertices[i + 2]) << "\n";
    o << "normals " << m.normals.size() / 3 << "\n";
    for (size_t i = 0; i + 2 < m.normals.size(); i += 3) o << hexf(m.normals[i]) << " " << hexf(m.normals[i + 1]) << " " << hexf(m.normals[i + 2]) << "\n";
    o << "texcoords " << m.texcoords.size() / 2 << "\n";
    for (size_t i = 0; i + 1 < m.texcoords.size(); i += 2) o << hexf(m.texcoords[i]) << " " << hexf(m.texcoords[i + 1]) << "\n";
    o << "shapes " << m.shapes.size() << "\n";
    for (const ObjShape& s : m.shapes) {
        o << "shape [" << s.name << "] triangles " << s.indices.size() / 3 << "\n";
        for (size_t i = 0; i + 2 < s.indices.size(); i += 3) {
            for (int k = 0; k < 3; k++) o << s.indices[i + k].v << "/" << s.indices[i + k].vt << "/" << s.indices[i + k].vn << " ";
            o << "m " << s.material_ids[i / 3] << "\n";
        }
    }
    o << "materials " << m.materials.size() << "\n";
    for (const ObjMaterial& k : m.materials) {
        o << "material [" << k.name << "] illum " << k.illum << " Kd " << hexf(k.diffuse[0]) << " " << hexf(k.diffuse[1]) << " " << hexf(k.diffuse[2])
          << " Ke " << hexf(k.emission[0]) << " " << hexf(k.emission[1]) << " " << hexf(k.emission[2]) << " Ni " << hexf(k.ior) << " d "
          << hexf(k.dissolve) << " map_Kd [" << k.diffuse_texname << "]\n";
    }
    return o.str();
}

}  // namespace rth
