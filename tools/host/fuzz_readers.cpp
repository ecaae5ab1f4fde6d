// fuzz_readers.cpp -- mutation fuzzer for the three text readers that parse files a user hands to bin/main and
// bin/server: the scene file (SceneLoader::load_scene), Wavefront OBJ and MTL (rth::load_obj).  Built with
// -fsanitize=address,undefined (make bin/fuzz_readers), so an out-of-bounds read, a signed overflow or a use after free
// in a reader ends the run with a report; a reader that rejects a damaged file is doing its job.
//
//   fuzz_readers <kind: obj|scene> <seed file> <work dir> <iterations> <rng seed>
//
// obj:   the seed is an OBJ file whose `mtllib` (if any) lies beside it; every iteration writes a damaged copy of the OBJ
//        -- every fourth one a damaged copy of the MTL instead -- into the work dir and reads it.
// scene: the seed is a scene file with absolute `path` entries; the damaged copy is loaded whole (scene text, the models
//        it still manages to name, transform, BVH build -- the host half of the library, no device).
// Mutations: byte flips, cuts, duplications, line swaps, and tokens swapped for the values text readers trip over
// (huge and negative indices, "1e999", "nan", lone signs, slashes, very long words, NUL bytes).
#include <sys/stat.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "obj_model.h"
#include "scene_loader.h"

namespace {

struct Rng {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return uint32_t(s >> 33); }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

const char* const kNasty[] = {"0", "-1", "-0", "2147483647", "2147483648", "-2147483649", "99999999999999999999", "4294967295", "1e999", "-1e999",
                              "1e-999", "nan", "inf", "-inf", "+", "-", ".", "e", "1e", "1e+", "-.e5", "/", "//", "1/", "/1", "1//", "1/2/3/4", "0/0/0",
                              "-1/-1/-1", "1e308", "0x10", "#", "end", "object", "path", "mtldir", "usemtl", "mtllib", "newmtl", "f", "v", "vt", "vn",
                              "size", "sample_per_pixel", "", " ", "\t", "\r", "\\", "\"", "%s%s%n", "../../../../etc/passwd"};

std::string slurp(const std::string& path) {
    std::ifstream in(path.c_str(), std::ios::binary);
    return std::string(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
}

void spit(const std::string& path, const std::string& data) {
    std::ofstream out(path.c_str(), std::ios::binary | std::ios::trunc);
    out.write(data.data(), std::streamsize(data.size()));
}

// [begin, end) of the blank-separated word that covers or follows position `at`
bool word_at(const std::string& s, size_t at, size_t& begin, size_t& end) {
    while (at < s.size() && (s[at] == ' ' || s[at] == '\n' || s[at] == '\t' || s[at] == '\r')) at++;
    if (at >= s.size()) return false;
    begin = at;
    while (begin > 0 && !(s[begin - 1] == ' ' || s[begin - 1] == '\n' || s[begin - 1] == '\t' || s[begin - 1] == '\r')) begin--;
    end = at;
    while (end < s.size() && !(s[end] == ' ' || s[end] == '\n' || s[end] == '\t' || s[end] == '\r')) end++;
    return true;
}

std::string mutate(const std::string& seed, Rng& rng) {
    std::string s = seed;
    const int rounds = 1 + int(rng.below(6));
    for (int r = 0; r < rounds && !s.empty(); r++) {
        const size_t at = rng.below(uint32_t(s.size()));
        switch (rng.below(9)) {
        case 0: s[at] = char(rng.below(256)); break;
        case 1: s.erase(at, 1 + rng.below(32)); break;
        case 2: s.insert(at, s.substr(rng.below(uint32_t(s.size())), 1 + rng.below(64))); break;
        case 3: case 4: case 5: {                  // a word replaced by a value readers trip over
            size_t b, e;
            if (word_at(s, at, b, e)) s.replace(b, e - b, kNasty[rng.below(sizeof kNasty / sizeof kNasty[0])]);
            break;
        }
        case 6: s.resize(at); break;               // the file ends here
        case 7: s.insert(at, std::string(1 + rng.below(5000), char('0' + rng.below(10)))); break;       // a very long number
        default: s[at] = '\0'; break;
        }
    }
    return s;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 6) {
        std::cerr << "usage: fuzz_readers <obj|scene> <seed file> <work dir> <iterations> <rng seed>\n";
        return 2;
    }
    const std::string kind = argv[1], seed_path = argv[2], work = argv[3];
    const long iterations = std::atol(argv[4]);
    Rng rng{uint64_t(std::strtoull(argv[5], nullptr, 0)) * 2654435761ull + 1};
    const std::string seed = slurp(seed_path);
    if (seed.empty()) { std::cerr << "empty seed " << seed_path << "\n"; return 2; }
    mkdir(work.c_str(), 0777);
    long accepted = 0, rejected = 0;
    if (kind == "obj") {
        // the MTL file the seed names, if any, is copied beside the damaged OBJ (same name), or damaged itself
        std::string mtl_name, mtl_seed;
        const size_t lib = seed.find("mtllib ");
        if (lib != std::string::npos) {
            const size_t eol = seed.find('\n', lib);
            mtl_name = seed.substr(lib + 7, eol == std::string::npos ? std::string::npos : eol - lib - 7);
            while (!mtl_name.empty() && (mtl_name.back() == '\r' || mtl_name.back() == ' ')) mtl_name.pop_back();
            const size_t slash = seed_path.rfind('/');
            mtl_seed = slurp((slash == std::string::npos ? std::string() : seed_path.substr(0, slash + 1)) + mtl_name);
        }
        const std::string obj_out = work + "/fuzz.obj";
        for (long i = 0; i < iterations; i++) {
            const bool damage_mtl = !mtl_seed.empty() && (i % 4 == 3);
            spit(obj_out, damage_mtl ? seed : mutate(seed, rng));
            if (!mtl_name.empty() && mtl_name.find('/') == std::string::npos) spit(work + "/" + mtl_name, damage_mtl ? mutate(mtl_seed, rng) : mtl_seed);
            rth::ObjModel model;
            if (rth::load_obj(obj_out, i % 2 ? work : std::string(), model)) {
                accepted++;
                // attribute indices are handed on as written (tinyobjloader does not range-check them either; the scene
                // loader does, where it uses them); what the reader itself guarantees is checked here
                for (const rth::ObjShape& shp : model.shapes) {
                    if (shp.indices.size() != shp.material_ids.size() * 3) { std::cerr << "shape arrays out of step\n"; return 1; }
                    for (int m : shp.material_ids)
                        if (m < -1 || m >= int(model.materials.size())) { std::cerr << "material id out of range accepted\n"; return 1; }
                }
                (void)rth::dump_obj(model);
            } else {
                rejected++;
            }
        }
    } else if (kind == "scene") {
        const std::string scene_out = work + "/fuzz_scene.txt";
        // the loader reports what it rejects on stderr, thousands of times over: keep the sanitizer's reports readable
        FILE* quiet = std::freopen((work + "/loader_messages.txt").c_str(), "w", stdout);
        (void)quiet;
        for (long i = 0; i < iterations; i++) {
            spit(scene_out, mutate(seed, rng));
            rth::SceneLoader loader;
            std::streambuf* keep = std::cerr.rdbuf(std::cout.rdbuf());
            const int rc = loader.load_scene(scene_out);
            std::cerr.rdbuf(keep);
            if (rc == 0) {
                accepted++;
                for (const rth::HostObject& o : loader.objects())
                    for (const rt_triangle& t : o.triangles)
                        if (t.material_id < 0 || t.material_id >= loader.num_material()) { std::cerr << "material id out of range accepted\n"; return 1; }
                for (const rt_material& m : loader.materials())
                    if (m.texture >= int(loader.textures().size())) { std::cerr << "texture index out of range accepted\n"; return 1; }
            } else {
                rejected++;
            }
        }
    } else {
        std::cerr << "unknown kind " << kind << "\n";
        return 2;
    }
    std::cerr << kind << ": " << iterations << " damaged files, " << accepted << " read, " << rejected << " rejected, no sanitizer report\n";
    return 0;
}
