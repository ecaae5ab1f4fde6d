// obj_dump.cpp -- prints the product's OBJ reader's view of a file (same text form as oracle/ref_tinyobj_dump.cpp).
#include <iostream>

#include "obj_model.h"

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: obj_dump file.obj [mtl_search_path]\n"; return 2; }
    rth::ObjModel m;
    if (!rth::load_obj(argv[1], argc > 2 ? argv[2] : "", m)) {
        std::cout << "error\n";
        return 1;
    }
    std::cout << rth::dump_obj(m);
    return 0;
}
