// obj_reader.h -- Wavefront OBJ + MTL ingestion for the pbrlab scene path.
//
// pbrlab reads .obj through its vendored tinyobjloader 2.0.0 (src/io/triangle-mesh-io.cc:216-236,
// tinyobj::LoadObj(..., triangulate = true)) and then only looks at: the shared attribute arrays, per shape the
// corner index triples and per-face material ids, and per material its name and the map of keys tinyobj does not
// know (the PBR/SSS extension: base_color, subsurface, ..., map_base_color; triangle-mesh-io.cc:34-212).
// This reader produces exactly that, with the same results on the same files (shape splitting at g/o/usemtl,
// quad split along the shorter diagonal, ear clipping of larger polygons, relative indices, the first-wins
// material-name and key maps, number scanning that rounds like tinyobj's own scanner).  Checked against the
// reference's loader compiled unmodified (see tests/test_io_cpu.py).
#ifndef PBRLAB_AMD_IO_OBJ_READER_H_
#define PBRLAB_AMD_IO_OBJ_READER_H_

#include <map>
#include <string>
#include <vector>

namespace pbio {

struct ObjCorner {
  int v = -1, vn = -1, vt = -1;  // zero-based; -1 = absent
};

struct ObjShape {
  std::string name;
  std::vector<ObjCorner> corners;  // 3 per triangle
  std::vector<int> material_ids;   // per triangle; -1 = none
  bool has_lines = false, has_points = false;
};

struct ObjMaterial {
  std::string name;
  std::map<std::string, std::string> params;  // keys the MTL grammar does not define; first occurrence wins
};

struct ObjFile {
  std::vector<float> vertices;   // xyz
  std::vector<float> normals;    // xyz
  std::vector<float> texcoords;  // uv as written in the file (no flip)
  std::vector<ObjShape> shapes;
  std::vector<ObjMaterial> materials;
  std::string warn, err;
};

// mtl_dir: directory searched for `mtllib` files ("" = as written, relative to the cwd); may hold several
// directories separated by ':'.  Returns false when the file cannot be opened or a face/line/point statement holds
// an invalid (zero) vertex index -- the cases in which tinyobj::LoadObj returns false.
bool ReadObj(const std::string& filename, const std::string& mtl_dir, ObjFile* out);

// Decimal scanner with tinyobj's rounding (tryParseDouble, tiny_obj_loader.h:887-1017): not strtod.
bool ScanReal(const char* s, const char* s_end, double* result);

// texture statement of a map_* key: file name (rest of the line after the options) and -colorspace value
// (tinyobj::ParseTextureNameAndOption, tiny_obj_loader.h:1243-1327, as used by triangle-mesh-io.cc:121-135)
bool ParseTextureStatement(const char* value, std::string* texname, std::string* colorspace);

}  // namespace pbio
#endif
