// main.cpp -- bin/main <scene_file> <output_dir> <batch_size>: one frame, then exit (csrc/main.cpp:9-20).
#include <cstdlib>
#include <iostream>

#include "worker.h"

int main(int argc, char** argv) {
    if (argc != 4) {
        std::cerr << "Usage: " << argv[0] << " <scene_file> <output_dir> <batch_size>" << std::endl;
        return 0;
    }
    const int batch_size = std::atoi(argv[3]);
    rth::Worker worker(argv[1], argv[2], batch_size);
    worker.run();
    worker.force_render();
    worker.stop();
    return 0;
}
