// png_tool.cpp -- exercises image_io from the tests.
//   png_tool decode <image> <out.rgb>      prints "width height", writes 3 bytes per pixel
//   png_tool encode <in.rgb> <w> <h> <out.png>
//   png_tool frame <in.f32> <w> <h> <out.png>   float RGB frame -> PNG the way the worker writes it
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>

#include "image_io.h"

int main(int argc, char** argv) {
    std::string error;
    if (argc == 4 && !std::strcmp(argv[1], "decode")) {
        rth::Image8 img;
        if (!rth::read_image(argv[2], img, error)) { std::cerr << error << "\n"; return 1; }
        std::ofstream f(argv[3], std::ios::binary | std::ios::trunc);
        f.write(reinterpret_cast<const char*>(img.rgb.data()), std::streamsize(img.rgb.size()));
        std::printf("%d %d\n", img.width, img.height);
        return 0;
    }
    if (argc == 6 && (!std::strcmp(argv[1], "encode") || !std::strcmp(argv[1], "frame"))) {
        std::ifstream f(argv[2], std::ios::binary);
        const std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
        std::vector<uint8_t> rgb;
        if (!std::strcmp(argv[1], "frame")) {
            if (bytes.size() != size_t(w) * size_t(h) * 12) { std::cerr << "size mismatch\n"; return 1; }
            rth::frame_to_rgb8(reinterpret_cast<const float*>(bytes.data()), w, h, rgb);
        } else {
            if (bytes.size() != size_t(w) * size_t(h) * 3) { std::cerr << "size mismatch\n"; return 1; }
            rgb.assign(bytes.begin(), bytes.end());
        }
        if (!rth::write_png_rgb8(argv[5], w, h, rgb.data(), error)) { std::cerr << error << "\n"; return 1; }
        return 0;
    }
    std::cerr << "usage: png_tool decode|encode|frame ...\n";
    return 2;
}
