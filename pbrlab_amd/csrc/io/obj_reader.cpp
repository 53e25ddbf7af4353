// obj_reader.cpp -- see obj_reader.h.  Behavioural reference: the vendored tinyobjloader 2.0.0 that pbrlab calls
// (src/io/tiny_obj_loader.h; line numbers below refer to that file).  Written from scratch around a line cursor;
// the statement order, index rules and float arithmetic are kept so that results are identical.
#include "obj_reader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <set>
#include <sstream>

namespace pbio {
namespace {

inline bool blank(char c) { return c == ' ' || c == '\t'; }
inline bool digit(char c) { return static_cast<unsigned>(c - '0') < 10u; }
inline bool eol(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// Splits a buffer into lines at "\n", "\r\n" and lone "\r" (safeGetline, :762-797).
struct LineCursor {
  const std::string& buf;
  size_t pos = 0;
  explicit LineCursor(const std::string& b) : buf(b) {}
  bool next(std::string* line) {
    if (pos >= buf.size()) return false;
    line->clear();
    while (pos < buf.size()) {
      const char c = buf[pos++];
      if (c == '\n') return true;
      if (c == '\r') {
        if (pos < buf.size() && buf[pos] == '\n') pos++;
        return true;
      }
      line->push_back(c);
    }
    return true;
  }
};

bool slurp(const std::string& path, std::string* out) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

// one whitespace-delimited word (parseString, :844-851)
std::string word(const char** p) {
  *p += strspn(*p, " \t");
  const size_t n = strcspn(*p, " \t\r");
  std::string s(*p, *p + n);
  *p += n;
  return s;
}

int integer(const char** p) {  // parseInt, :853-858
  *p += strspn(*p, " \t");
  const int i = atoi(*p);
  *p += strcspn(*p, " \t\r");
  return i;
}

float real(const char** p, double dflt = 0.0) {  // parseReal, :1019-1027
  *p += strspn(*p, " \t");
  const char* end = *p + strcspn(*p, " \t\r");
  double val = dflt;
  ScanReal(*p, end, &val);
  *p = end;
  return static_cast<float>(val);
}

bool real_checked(const char** p, float* out) {  // parseReal(token, out), :1029-1040
  *p += strspn(*p, " \t");
  const char* end = *p + strcspn(*p, " \t\r");
  double val;
  const bool ok = ScanReal(*p, end, &val);
  if (ok) *out = static_cast<float>(val);
  *p = end;
  return ok;
}

struct Corner {
  int v, vn, vt;
};

// OBJ index -> zero based; negative = relative to the count so far; zero only tolerated for vn/vt (fixIndex, :815-842)
bool fix_index(int idx, int n, int* out, bool allow_zero, std::string* warn, size_t line_no) {
  if (idx > 0) {
    *out = idx - 1;
    return true;
  }
  if (idx == 0) {
    *warn += "A zero value index found (will have a value of -1 for normal and tex indices. Line " +
             std::to_string(line_no) + ").\n";
    *out = -1;
    return allow_zero;
  }
  *out = n + idx;
  return true;
}

// "i", "i/j", "i//k", "i/j/k" (parseTriple, :1157-1208).  atoi() is applied where the cursor stands, as there.
bool corner(const char** p, int nv, int nvn, int nvt, Corner* out, std::string* warn, size_t line_no) {
  Corner c = {-1, -1, -1};
  if (!fix_index(atoi(*p), nv, &c.v, false, warn, line_no)) return false;
  *p += strcspn(*p, "/ \t\r");
  if ((*p)[0] != '/') {
    *out = c;
    return true;
  }
  (*p)++;
  if ((*p)[0] == '/') {  // i//k
    (*p)++;
    if (!fix_index(atoi(*p), nvn, &c.vn, true, warn, line_no)) return false;
    *p += strcspn(*p, "/ \t\r");
    *out = c;
    return true;
  }
  if (!fix_index(atoi(*p), nvt, &c.vt, true, warn, line_no)) return false;
  *p += strcspn(*p, "/ \t\r");
  if ((*p)[0] != '/') {
    *out = c;
    return true;
  }
  (*p)++;
  if (!fix_index(atoi(*p), nvn, &c.vn, true, warn, line_no)) return false;
  *p += strcspn(*p, "/ \t\r");
  *out = c;
  return true;
}

struct Pending {  // statements collected since the last flush (PrimGroup)
  std::vector<std::vector<Corner>> faces;
  size_t lines = 0, points = 0;          // `l` / `p` statements
  size_t line_refs = 0, point_refs = 0;  // vertex references inside them
  bool empty() const { return faces.empty() && lines == 0 && points == 0; }
  void clear() {
    faces.clear();
    lines = points = line_refs = point_refs = 0;
  }
};

void emit(ObjShape* s, const Corner& a, const Corner& b, const Corner& c, int material) {
  const Corner t[3] = {a, b, c};
  for (const Corner& k : t) {
    ObjCorner o;
    o.v = k.v, o.vn = k.vn, o.vt = k.vt;
    s->corners.push_back(o);
  }
  s->material_ids.push_back(material);
}

// crossing-number test of (tx,ty) against a triangle (pnpoly, :1408-1419), single precision
int inside3(const float* vx, const float* vy, float tx, float ty) {
  int c = 0;
  for (int i = 0, j = 2; i < 3; j = i++) {
    if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
  }
  return c;
}

// Polygon with more than 4 corners: ear clipping in the coordinate plane chosen from the first non-degenerate
// corner (:1709-1933, the build without TINYOBJLOADER_USE_MAPBOX_EARCUT, which is how pbrlab compiles it).
void clip_ears(ObjShape* s, const std::vector<Corner>& face, const std::vector<float>& v, int material) {
  size_t n = face.size();
  size_t ax[2] = {1, 2};
  for (size_t k = 0; k < n; ++k) {
    const size_t a = size_t(face[k % n].v), b = size_t(face[(k + 1) % n].v), c = size_t(face[(k + 2) % n].v);
    if ((3 * a + 2) >= v.size() || (3 * b + 2) >= v.size() || (3 * c + 2) >= v.size()) continue;
    const float e0x = v[b * 3 + 0] - v[a * 3 + 0], e0y = v[b * 3 + 1] - v[a * 3 + 1], e0z = v[b * 3 + 2] - v[a * 3 + 2];
    const float e1x = v[c * 3 + 0] - v[b * 3 + 0], e1y = v[c * 3 + 1] - v[b * 3 + 1], e1z = v[c * 3 + 2] - v[b * 3 + 2];
    const float cx = std::fabs(e0y * e1z - e0z * e1y);
    const float cy = std::fabs(e0z * e1x - e0x * e1z);
    const float cz = std::fabs(e0x * e1y - e0y * e1x);
    const float eps = std::numeric_limits<float>::epsilon();
    if (cx > eps || cy > eps || cz > eps) {
      if (!(cx > cy && cx > cz)) {
        ax[0] = 0;
        if (cz > cx && cz > cy) ax[1] = 1;
      }
      break;
    }
  }

  std::vector<Corner> rest = face;
  size_t guess = 0;
  size_t budget = face.size();  // iterations allowed without removing a vertex
  size_t last_count = rest.size();
  Corner ind[3];
  float vx[3], vy[3];
  while (rest.size() > 3 && budget > 0) {
    n = rest.size();
    if (guess >= n) guess -= n;
    if (last_count != n) {
      last_count = n;
      budget = n;
    } else {
      budget--;
    }
    for (size_t k = 0; k < 3; ++k) {
      ind[k] = rest[(guess + k) % n];
      const size_t vi = size_t(ind[k].v);
      if ((vi * 3 + ax[0]) >= v.size() || (vi * 3 + ax[1]) >= v.size()) {
        vx[k] = 0.0f, vy[k] = 0.0f;
      } else {
        vx[k] = v[vi * 3 + ax[0]], vy[k] = v[vi * 3 + ax[1]];
      }
    }
    const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0];
    const float e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
    const float cross = e0x * e1y - e0y * e1x;
    const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
    if (cross * area < 0.0f) {  // reflex corner
      guess += 1;
      continue;
    }
    bool overlap = false;
    for (size_t other = 3; other < n; ++other) {
      const size_t idx = (guess + other) % n;
      const size_t ovi = size_t(rest[idx].v);
      if ((ovi * 3 + ax[0]) >= v.size() || (ovi * 3 + ax[1]) >= v.size()) continue;
      if (inside3(vx, vy, v[ovi * 3 + ax[0]], v[ovi * 3 + ax[1]])) {
        overlap = true;
        break;
      }
    }
    if (overlap) {
      guess += 1;
      continue;
    }
    emit(s, ind[0], ind[1], ind[2], material);
    rest.erase(rest.begin() + long((guess + 1) % n));
  }
  if (rest.size() == 3) emit(s, rest[0], rest[1], rest[2], material);
}

// PrimGroup -> triangles of the current shape (exportGroupsToShape, :1457-1996, triangulate = true).
// `v` is the vertex array as read SO FAR, like there.
bool flush(ObjShape* s, const Pending& pg, int material, const std::string& name, const std::vector<float>& v,
           std::string* warn) {
  if (pg.empty()) return false;
  s->name = name;
  for (const std::vector<Corner>& f : pg.faces) {
    const size_t n = f.size();
    if (n < 3) {
      *warn += "Degenerated face found\n.";
      continue;
    }
    if (n == 3) {
      emit(s, f[0], f[1], f[2], material);
    } else if (n == 4) {
      const size_t a = size_t(f[0].v), b = size_t(f[1].v), c = size_t(f[2].v), d = size_t(f[3].v);
      if ((3 * a + 2) >= v.size() || (3 * b + 2) >= v.size() || (3 * c + 2) >= v.size() || (3 * d + 2) >= v.size()) {
        *warn += "Face with invalid vertex index found.\n";
        continue;
      }
      // split along the shorter diagonal (:1518-1575)
      const float e02x = v[c * 3 + 0] - v[a * 3 + 0], e02y = v[c * 3 + 1] - v[a * 3 + 1], e02z = v[c * 3 + 2] - v[a * 3 + 2];
      const float e13x = v[d * 3 + 0] - v[b * 3 + 0], e13y = v[d * 3 + 1] - v[b * 3 + 1], e13z = v[d * 3 + 2] - v[b * 3 + 2];
      const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
      const float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
      if (sqr02 < sqr13) {
        emit(s, f[0], f[1], f[2], material);
        emit(s, f[0], f[2], f[3], material);
      } else {
        emit(s, f[0], f[1], f[3], material);
        emit(s, f[1], f[2], f[3], material);
      }
    } else {
      clip_ears(s, f, v, material);
    }
  }
  if (pg.line_refs) s->has_lines = true;
  if (pg.point_refs) s->has_points = true;
  return true;
}

// `mtllib a.mtl b\ c.mtl`: split at unescaped spaces; the last piece is kept even when empty (SplitString, :1999-2022)
std::vector<std::string> split_names(const std::string& s) {
  std::vector<std::string> out;
  std::string cur;
  bool esc = false;
  for (char ch : s) {
    if (esc) {
      esc = false;
    } else if (ch == '\\') {
      esc = true;
      continue;
    } else if (ch == ' ') {
      if (!cur.empty()) out.push_back(cur);
      cur.clear();
      continue;
    }
    cur += ch;
  }
  out.push_back(cur);
  return out;
}

// true when the MTL grammar of tinyobj consumes this statement (LoadMtl, :2115-2413); such keys never reach the
// unknown-parameter map that pbrlab reads.
bool known_mtl_key(const char* t) {
  auto two = [&](char a, char b) { return t[0] == a && t[1] == b && blank(t[2]); };
  auto kw = [&](const char* k) {
    const size_t n = strlen(k);
    return strncmp(t, k, n) == 0 && blank(t[n]);
  };
  if (two('K', 'a') || two('K', 'd') || two('K', 's') || two('K', 't') || two('T', 'f') || two('N', 'i') ||
      two('K', 'e') || two('N', 's') || two('T', 'r') || two('P', 'r') || two('P', 'm') || two('P', 's') ||
      two('P', 'c'))
    return true;
  if (t[0] == 'd' && blank(t[1])) return true;
  static const char* kws[] = {"illum",    "Pcr",      "aniso",    "anisor", "map_Ka", "map_Kd", "map_Ks", "map_Ns",
                              "map_bump", "map_Bump", "bump",     "map_d",  "map_disp", "map_Disp", "disp", "refl",
                              "map_Pr",   "map_Pm",   "map_Ps",   "map_Ke", "norm"};
  for (const char* k : kws)
    if (kw(k)) return true;
  return false;
}

// LoadMtl (:2039-2437) reduced to what pbrlab reads: names and unknown parameters.
void read_mtl(const std::string& text, std::map<std::string, int>* name_to_id, std::vector<ObjMaterial>* materials,
              std::string* warn) {
  ObjMaterial cur;
  LineCursor lc(text);
  std::string line;
  while (lc.next(&line)) {
    if (!line.empty()) line = line.substr(0, line.find_last_not_of(" \t") + 1);  // trailing blanks
    if (line.empty()) continue;
    const char* t = line.c_str();
    t += strspn(t, " \t");
    if (t[0] == '\0' || t[0] == '#') continue;
    if (strncmp(t, "newmtl", 6) == 0 && blank(t[6])) {
      if (!cur.name.empty()) {
        name_to_id->insert(std::make_pair(cur.name, int(materials->size())));  // first definition wins
        materials->push_back(cur);
      }
      cur = ObjMaterial();
      t += 7;
      cur.name = word(&t);
      if (cur.name.empty()) *warn += "empty material name in `newmtl`\n";
      continue;
    }
    if (known_mtl_key(t)) continue;
    const char* sp = strchr(t, ' ');
    if (!sp) sp = strchr(t, '\t');
    if (sp) cur.params.insert(std::make_pair(std::string(t, size_t(sp - t)), std::string(sp + 1)));
  }
  name_to_id->insert(std::make_pair(cur.name, int(materials->size())));
  materials->push_back(cur);
}

// MaterialFileReader (:2439-2496)
bool load_mtl_file(const std::string& mtl_dir, const std::string& name, std::map<std::string, int>* name_to_id,
                   std::vector<ObjMaterial>* materials, std::string* warn) {
  std::string text;
  if (!mtl_dir.empty()) {
    std::istringstream dirs(mtl_dir);
    std::string d;
    while (std::getline(dirs, d, ':')) {
      std::string path = d.empty() ? name : (d.back() != '/' ? d + "/" + name : d + name);
      if (slurp(path, &text)) {
        read_mtl(text, name_to_id, materials, warn);
        return true;
      }
    }
  } else if (slurp(name, &text)) {
    read_mtl(text, name_to_id, materials, warn);
    return true;
  }
  *warn += "Material file [ " + name + " ] not found in a path : " + mtl_dir + "\n";
  return false;
}

}  // namespace

bool ScanReal(const char* s, const char* s_end, double* result) {
  if (s >= s_end) return false;
  double mantissa = 0.0;
  int exponent = 0;
  char sign = '+', exp_sign = '+';
  const char* p = s;
  bool lead_dot = false;
  if (*p == '+' || *p == '-') {
    sign = *p++;
    if (p != s_end && *p == '.') lead_dot = true;
  } else if (digit(*p)) {
  } else if (*p == '.') {
    lead_dot = true;
  } else {
    return false;
  }
  if (!lead_dot) {
    int nread = 0;
    while (p != s_end && digit(*p)) {
      mantissa *= 10;
      mantissa += static_cast<int>(*p - '0');
      p++, nread++;
    }
    if (nread == 0) return false;
  }
  bool has_exp = false;
  if (p != s_end) {
    if (*p == '.') {
      p++;
      int k = 1;  // k-th digit after the point weighs 10^-k: a literal for k < 8, pow() beyond
      static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
      while (p != s_end && digit(*p)) {
        mantissa += static_cast<int>(*p - '0') * (k < 8 ? lut[k] : std::pow(10.0, -k));
        k++, p++;
      }
      has_exp = (p != s_end) && (*p == 'e' || *p == 'E');
    } else if (*p == 'e' || *p == 'E') {
      has_exp = true;
    }
  }
  if (has_exp) {
    p++;
    if (p != s_end && (*p == '+' || *p == '-')) {
      exp_sign = *p++;
    } else if (digit(*p)) {
    } else {
      return false;
    }
    int nread = 0;
    while (p != s_end && digit(*p)) {
      if (exponent > (2147483647 / 10)) return false;
      exponent *= 10;
      exponent += static_cast<int>(*p - '0');
      p++, nread++;
    }
    exponent *= (exp_sign == '+' ? 1 : -1);
    if (nread == 0) return false;
  }
  *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
  return true;
}

bool ParseTextureStatement(const char* value, std::string* texname, std::string* colorspace) {
  bool found = false;
  std::string name;
  colorspace->clear();
  const char* t = value;
  auto opt = [&](const char* k) {
    const size_t n = strlen(k);
    return strncmp(t, k, n) == 0 && blank(t[n]);
  };
  auto skip_word = [&]() {
    t += strspn(t, " \t");
    t += strcspn(t, " \t\r");
  };
  while (!eol(*t)) {
    t += strspn(t, " \t");
    if (opt("-blendu") || opt("-blendv")) {
      t += 8, skip_word();
    } else if (opt("-clamp") || opt("-boost")) {
      t += 7, skip_word();
    } else if (opt("-bm")) {
      t += 4, skip_word();
    } else if (opt("-o") || opt("-s") || opt("-t")) {
      t += 3, skip_word(), skip_word(), skip_word();
    } else if (opt("-type")) {
      t += 5, skip_word();
    } else if (opt("-texres")) {
      t += 7, skip_word();
    } else if (opt("-imfchan")) {
      t += 9, skip_word();
    } else if (opt("-mm")) {
      t += 4, skip_word(), skip_word();
    } else if (opt("-colorspace")) {
      t += 12;
      *colorspace = word(&t);
    } else {
      name = std::string(t);  // the rest of the line, blanks included
      t += name.size();
      found = true;
    }
  }
  if (found) *texname = name;
  return found;
}

bool ReadObj(const std::string& filename, const std::string& mtl_dir_in, ObjFile* out) {
  *out = ObjFile();
  std::string text;
  if (!slurp(filename, &text)) {
    out->err = "Cannot open file [" + filename + "]\n";
    return false;
  }
  std::string mtl_dir = mtl_dir_in;
  if (!mtl_dir.empty() && mtl_dir.back() != '/') mtl_dir += '/';

  std::vector<float>&v = out->vertices, &vn = out->normals, &vt = out->texcoords;
  Pending pg;
  std::string name;
  std::set<std::string> loaded_mtl;
  std::map<std::string, int> name_to_id;
  int material = -1;
  int max_v = -1, max_vn = -1, max_vt = -1;
  ObjShape shape;

  LineCursor lc(text);
  std::string line;
  size_t line_no = 0;
  while (lc.next(&line)) {
    line_no++;
    if (line.empty()) continue;
    const char* t = line.c_str();
    t += strspn(t, " \t");
    if (t[0] == '\0' || t[0] == '#') continue;

    if (t[0] == 'v' && blank(t[1])) {  // position (+ optional colour, ignored by pbrlab)
      t += 2;
      const float x = real(&t), y = real(&t), z = real(&t);
      float r, g, b;
      (void)(real_checked(&t, &r) && real_checked(&t, &g) && real_checked(&t, &b));
      v.push_back(x), v.push_back(y), v.push_back(z);
      continue;
    }
    if (t[0] == 'v' && t[1] == 'n' && blank(t[2])) {
      t += 3;
      const float x = real(&t), y = real(&t), z = real(&t);
      vn.push_back(x), vn.push_back(y), vn.push_back(z);
      continue;
    }
    if (t[0] == 'v' && t[1] == 't' && blank(t[2])) {
      t += 3;
      const float x = real(&t), y = real(&t);
      vt.push_back(x), vt.push_back(y);
      continue;
    }
    if (t[0] == 'v' && t[1] == 'w' && blank(t[2])) {  // skin weights: only their failure mode matters here
      t += 3;
      (void)integer(&t);
      while (!eol(t[0])) {
        const float j = real(&t, -1.0);
        (void)real(&t, -1.0);
        if (j < 0.0f) {
          out->err += "Failed parse `vw' line. joint_id is negative. line " + std::to_string(line_no) + ".)\n";
          return false;
        }
        t += strspn(t, " \t\r");
      }
      continue;  // no later statement matches a line starting with "vw"
    }
    if ((t[0] == 'l' || t[0] == 'p') && blank(t[1])) {  // polylines / points: indices are validated, geometry unused
      const bool is_line = t[0] == 'l';
      t += 2;
      while (!eol(t[0])) {
        Corner c;
        if (!corner(&t, int(v.size() / 3), int(vn.size() / 3), int(vt.size() / 2), &c, &out->warn, line_no)) {
          out->err += std::string("Failed to parse `") + (is_line ? "l" : "p") +
                      "' line (e.g. a zero value for vertex index. Line " + std::to_string(line_no) + ").\n";
          return false;
        }
        (is_line ? pg.line_refs : pg.point_refs)++;
        t += strspn(t, " \t\r");
      }
      (is_line ? pg.lines : pg.points)++;
      continue;
    }
    if (t[0] == 'f' && blank(t[1])) {
      t += 2;
      t += strspn(t, " \t");
      std::vector<Corner> face;
      face.reserve(4);
      while (!eol(t[0])) {
        Corner c;
        if (!corner(&t, int(v.size() / 3), int(vn.size() / 3), int(vt.size() / 2), &c, &out->warn, line_no)) {
          out->err += "Failed to parse `f' line (e.g. a zero value for vertex index. Line " + std::to_string(line_no) + ").\n";
          return false;
        }
        max_v = max_v > c.v ? max_v : c.v;
        max_vn = max_vn > c.vn ? max_vn : c.vn;
        max_vt = max_vt > c.vt ? max_vt : c.vt;
        face.push_back(c);
        t += strspn(t, " \t\r");
      }
      pg.faces.push_back(face);
      continue;
    }
    if (strncmp(t, "usemtl", 6) == 0) {  // no blank required after the keyword (:2805)
      t += 6;
      const std::string mname = word(&t);
      int id = -1;
      auto it = name_to_id.find(mname);
      if (it != name_to_id.end()) {
        id = it->second;
      } else {
        out->warn += "material [ '" + mname + "' ] not found in .mtl\n";
      }
      if (id != material) {  // faces so far keep the old material; the shape stays open
        flush(&shape, pg, material, name, v, &out->warn);
        pg.faces.clear();
        material = id;
      }
      continue;
    }
    if (strncmp(t, "mtllib", 6) == 0 && blank(t[6])) {
      t += 7;
      const std::vector<std::string> names = split_names(std::string(t));
      bool found = false;
      for (const std::string& n : names) {
        if (loaded_mtl.count(n)) {
          found = true;
          continue;
        }
        if (load_mtl_file(mtl_dir, n, &name_to_id, &out->materials, &out->warn)) {
          found = true;
          loaded_mtl.insert(n);
          break;
        }
      }
      if (!found) out->warn += "Failed to load material file(s). Use default material.\n";
      continue;
    }
    if (t[0] == 'g' && blank(t[1])) {
      flush(&shape, pg, material, name, v, &out->warn);
      if (!shape.corners.empty()) out->shapes.push_back(shape);
      shape = ObjShape();
      pg.clear();
      std::vector<std::string> names;  // names[0] is the "g" itself
      while (!eol(t[0])) {
        names.push_back(word(&t));
        t += strspn(t, " \t\r");
      }
      if (names.size() < 2) {
        out->warn += "Empty group name. line: " + std::to_string(line_no) + "\n";
        name = "";
      } else {
        name = names[1];
        for (size_t i = 2; i < names.size(); ++i) name += " " + names[i];
      }
      continue;
    }
    if (t[0] == 'o' && blank(t[1])) {
      flush(&shape, pg, material, name, v, &out->warn);
      if (!shape.corners.empty() || shape.has_lines || shape.has_points) out->shapes.push_back(shape);
      pg.clear();
      shape = ObjShape();
      name = std::string(t + 2);
      continue;
    }
    // t (tags), s (smoothing groups) and everything else: nothing pbrlab reads
  }

  if (max_v >= int(v.size() / 3)) out->warn += "Vertex indices out of bounds (line " + std::to_string(line_no) + ".)\n\n";
  if (max_vn >= int(vn.size() / 3)) out->warn += "Vertex normal indices out of bounds (line " + std::to_string(line_no) + ".)\n\n";
  if (max_vt >= int(vt.size() / 2)) out->warn += "Vertex texcoord indices out of bounds (line " + std::to_string(line_no) + ".)\n\n";

  const bool any = flush(&shape, pg, material, name, v, &out->warn);
  if (any || !shape.corners.empty()) out->shapes.push_back(shape);
  return true;
}

}  // namespace pbio
