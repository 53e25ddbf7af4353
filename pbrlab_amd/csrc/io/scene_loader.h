// scene_loader.h -- pbrlab's scene ingestion on top of the readers of this directory.
//   LoadTriangleMeshFromObj  <- src/io/triangle-mesh-io.cc:214-325  (material conversion :34-212)
//   AddObjToScene / AddHairToScene / CreateScene  <- pc/pc-common.cc:100-270
#ifndef PBRLAB_AMD_IO_SCENE_LOADER_H_
#define PBRLAB_AMD_IO_SCENE_LOADER_H_

#include <cstdint>
#include <string>
#include <vector>

#include "obj_reader.h"
#include "pbrhip.h"

namespace pbio {

struct LoadedTexture {  // pbrlab::Texture (src/texture.h:13-44)
  std::vector<float> pixels;
  uint32_t width = 0, height = 0, channels = 0;
  std::string name;
};

struct LoadedMesh {  // pbrlab::TriangleMesh minus the shared attribute pointer (src/mesh/triangle-mesh.h:14-57)
  std::string name;
  std::vector<uint32_t> vertex_ids, normal_ids, texcoord_ids;  // 3 per face
  std::vector<uint32_t> material_ids;                          // 1 per face, index into ObjScene::materials
};

struct ObjScene {
  std::vector<float> vertices_xyzw, normals_xyzw, texcoords_uv;  // pbrlab::Attribute (src/mesh/attribute.h:8-12)
  std::vector<LoadedMesh> meshes;
  std::vector<pbrhip_principled_param> materials;
  std::vector<std::string> material_names;
  std::vector<LoadedTexture> textures;
  ObjFile parsed;  // what the OBJ/MTL statements said (tests, diagnostics)
};

// CyclesPrincipledBsdfParameter{} (src/material-param.h:24-49)
pbrhip_principled_param DefaultPrincipledParam();
// HairBsdfParameter{} (src/material-param.h:51-72)
pbrhip_hair_param DefaultHairParam();

bool LoadTriangleMeshFromObj(const std::string& filename, ObjScene* out);

// false + message on failure; the scene keeps what was added before the failure
bool AddObjToScene(pbrhip_scene* scene, const std::string& obj_filename, std::string* err);
bool AddHairToScene(pbrhip_scene* scene, const std::string& hair_filename, std::string* err);
bool CreateScene(int argc, const char* const* argv, pbrhip_scene* scene, std::string* err);

}  // namespace pbio
#endif
