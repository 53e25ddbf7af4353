/* pbrhip_io.h -- C ABI of libpbrhip_io: scene ingestion and image output around the MI355X path tracer
 * (SURVEY.md §8f rows N1, N2).  Host code only (no GPU work happens here except through libpbrhip's scene calls).
 *
 * Replaces, with the same results on the same files:
 *   pbrlab::io::LoadTriangleMeshFromObj        src/io/triangle-mesh-io.cc:214-325 (+ vendored tinyobjloader 2.0.0)
 *   pbrlab::io::LoadCurveMeshAsCubicBezierCurve src/io/curve-mesh-io.cc:32-138 (+ io/cyhair.cc, curve-util.cc)
 *   CreateScene / CreateSceneFromObj / CreateSceneFromCubicBezierCurve   pc/pc-common.cc:100-270
 *   pbrlab::io::LoadImageFromFile / WritePNG   src/io/image-io.cc:98-224
 *   the output stage of pbrlab-cli             pc/pbrlab-cli.cc:47-57 (rgba/count -> sRGB -> 8-bit PNG)
 * Image files read: PNG, JPEG (baseline and progressive), BMP, TGA, GIF, PSD, PNM, Radiance .hdr, scanline OpenEXR
 * (NONE/RLE/ZIPS/ZIP/PIZ) -- pixel for pixel what the reference's stb_image / tinyexr return.  Not decoded by this build:
 * Softimage PIC and tiled or PXR24/B44/DWA OpenEXR textures (the call fails and says so). */
#ifndef PBRHIP_IO_H_
#define PBRHIP_IO_H_

#include <stddef.h>
#include <stdint.h>

#include "pbrhip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* pbrio_last_error(void);
void pbrio_free(void* p); /* buffers returned through `**` arguments */

/* ---- OBJ + MTL: io::LoadTriangleMeshFromObj(filename, &meshes, &material_params, &textures) ---- */
typedef struct pbrio_obj pbrio_obj;
int pbrio_obj_load(const char* filename, pbrio_obj** out);
void pbrio_obj_free(pbrio_obj*);
/* shared attribute buffers (mesh/attribute.h:8-12): which = 0 vertices xyzw (w = 1), 1 normals xyzw (w = 1),
 * 2 texcoords uv (v stored as 1 - v, triangle-mesh-io.cc:274-277); returns the float count */
size_t pbrio_obj_attribute(const pbrio_obj*, int which, const float** data);
uint32_t pbrio_obj_num_shapes(const pbrio_obj*);
const char* pbrio_obj_shape_name(const pbrio_obj*, uint32_t shape);
/* which = 0 vertex ids, 1 normal ids, 2 texcoord ids (3 per face; absent = 0xFFFFFFFF), 3 material ids (1 per face,
 * index into this file's materials; no material = 0xFFFFFFFF); returns the element count */
size_t pbrio_obj_shape_ids(const pbrio_obj*, uint32_t shape, int which, const uint32_t** data);
uint32_t pbrio_obj_num_materials(const pbrio_obj*);
/* ParseTinyObjMaterial (triangle-mesh-io.cc:139-212): texture ids index this file's textures */
int pbrio_obj_material(const pbrio_obj*, uint32_t material, pbrhip_principled_param* out, const char** name);
uint32_t pbrio_obj_num_textures(const pbrio_obj*);
int pbrio_obj_texture(const pbrio_obj*, uint32_t texture, const float** pixels, uint32_t* width, uint32_t* height,
                      uint32_t* channels, const char** name);
/* parse-level dump for tests: "shape\t<name>\n" per shape, then per material "material\t<name>\n" followed by
 * "param\t<key>\t<value>\n" for every key the MTL grammar does not define (sorted by key) */
const char* pbrio_obj_text(const pbrio_obj*);
const char* pbrio_obj_warnings(const pbrio_obj*);
/* the texture statement of a map_* key (tinyobj::ParseTextureNameAndOption as triangle-mesh-io.cc:121-135 uses it):
 * file name = the rest of the line after the options, colorspace = value of -colorspace ("" if absent).
 * Returns 1 when a file name was found, else 0. */
int pbrio_parse_texture_statement(const char* value, char* texname, size_t texname_cap, char* colorspace,
                                  size_t colorspace_cap);

/* ---- CyHair: io::LoadCurveMeshAsCubicBezierCurve(filepath, memory_saving_mode, &vertices_thickness, &indices) ---- */
typedef struct pbrio_curves pbrio_curves;
int pbrio_curves_load(const char* filepath, int memory_saving_mode, pbrio_curves** out);
void pbrio_curves_free(pbrio_curves*);
size_t pbrio_curves_vertices(const pbrio_curves*, const float** xyz_thickness); /* float count */
size_t pbrio_curves_indices(const pbrio_curves*, const uint32_t** first_control_point);

/* ---- scene assembly (pc/pc-common.cc) ---- */
/* CreateSceneFromObj (:100-190): materials, textures, then per shape mesh -> local scene -> identity instance;
 * shapes whose name starts with "light" emit (3,3,3) */
int pbrio_scene_add_obj(pbrhip_scene*, const char* obj_filename);
/* CreateSceneFromCubicBezierCurve (:192-236): default HairBsdfParameter for all segments */
int pbrio_scene_add_hair(pbrhip_scene*, const char* hair_filename);
/* CreateScene (:238-270): every argv[1..] ending in .obj / .hair, then CommitScene and the bounds printout */
int pbrio_create_scene(int argc, const char* const* argv, pbrhip_scene*);

/* ---- images ---- */
int pbrio_image_load(const char* filename, const char* asset_path, float** pixels, size_t* width, size_t* height,
                     size_t* channels);
int pbrio_write_png_f32(const char* filename, const char* asset_path, const float* pixels, size_t width, size_t height,
                        size_t channels);
int pbrio_write_png_u8(const char* filename, const char* asset_path, const uint8_t* pixels, size_t width, size_t height,
                       size_t channels);
int pbrio_png_decode(const uint8_t* file, size_t n, uint8_t** pixels, size_t* width, size_t* height, size_t* channels);
/* pbrlab-cli.cc:47-57: out[i] = byte(sRGB(rgba[i] / count[i / 4])) (alpha: no transfer function), RGBA8 */
int pbrio_layer_to_srgb8(const float* rgba, const uint32_t* count, size_t width, size_t height, uint8_t* out);
/* the whole output stage: resolve + WritePNG(filename, asset_path) */
int pbrio_write_layer_png(const char* filename, const char* asset_path, const float* rgba, const uint32_t* count,
                          size_t width, size_t height);

#ifdef __cplusplus
}
#endif
#endif /* PBRHIP_IO_H_ */
