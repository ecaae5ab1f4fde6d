// obj_model.h -- Wavefront OBJ/MTL reader for the scene loader.
//
// Upstream reads models through tinyobjloader (third/tiny_obj_loader.h, ObjReader with
// triangulate = true; csrc/sceneloader.cpp:110-125).  This is an independent reader for the subset the
// scene loader consumes -- positions, normals, texture coordinates, triangulated faces grouped into
// shapes, per-face material ids, and the MTL fields Kd, Ke, Ni, d/Tr, illum, map_Kd -- that reproduces
// tinyobjloader's observable results for that subset: same shapes in the same order, same index
// triples, same triangulation of quads and polygons, same float values (its number grammar and
// accumulation order are restated), same material table.
#ifndef RT_OBJ_MODEL_H
#define RT_OBJ_MODEL_H

#include <string>
#include <vector>

namespace rth {

struct ObjIndex {
    int v, vn, vt;       // zero-based; -1 = absent
};

struct ObjShape {
    std::string name;
    std::vector<ObjIndex> indices;       // three per triangle
    std::vector<int> material_ids;       // one per triangle, -1 = none
};

struct ObjMaterial {
    std::string name;
    float diffuse[3];
    float emission[3];
    float ior;
    float dissolve;
    int illum;
    std::string diffuse_texname;
};

struct ObjModel {
    std::vector<float> vertices;     // xyz
    std::vector<float> normals;      // xyz
    std::vector<float> texcoords;    // uv
    std::vector<ObjShape> shapes;
    std::vector<ObjMaterial> materials;
    std::string warning;
    std::string error;
};

// Reads `filename`; MTL files named by `mtllib` are looked up in `mtl_search_path` (':'-separated list), or
// next to the OBJ file when that is empty.  Returns false (and sets model.error) when the file cannot be
// opened or a face line cannot be parsed.
bool load_obj(const std::string& filename, const std::string& mtl_search_path, ObjModel& model);

// Canonical text dump of a model (used to compare readers).
std::string dump_obj(const ObjModel& model);

// The number grammar of the OBJ/MTL readers: [sign] digits ['.' digits] [('e'|'E') [sign] digits], also
// ".5" and "-.5".  Returns false and leaves *out alone when [s, s_end) does not start with a number.
bool parse_decimal(const char* s, const char* s_end, double* out);

}  // namespace rth
#endif
