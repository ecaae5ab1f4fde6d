// image_io.h -- image files without OpenCV: PNG (zlib), JPEG and binary PPM/PGM.
//
// Upstream decodes textures with cv::imread(IMREAD_COLOR) and writes frames with cv::imwrite
// (csrc/sceneloader.cpp:224, csrc/worker.cpp:42).  Here: a PNG reader for every colour type and bit
// depth, interlaced or not (reduced to 8-bit RGB the way IMREAD_COLOR does: alpha dropped, grey replicated,
// 16-bit samples keep their high byte), a JPEG reader (jpeg_decode.cpp), a binary PPM/PGM reader, and an
// 8-bit RGB PNG writer.
#ifndef RT_IMAGE_IO_H
#define RT_IMAGE_IO_H

#include <cstdint>
#include <string>
#include <vector>

namespace rth {

struct Image8 {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;      // 3 bytes per pixel, rows top to bottom
};

bool read_image(const std::string& path, Image8& out, std::string& error);       // PNG, JPEG, PPM (P6) or PGM (P5) by signature
bool decode_jpeg(const std::vector<uint8_t>& file, Image8& out, std::string& error);
bool write_png_rgb8(const std::string& path, int width, int height, const uint8_t* rgb, std::string& error);

// cv::Mat::convertTo(CV_8UC3, 255.0f) on one float: v * 255 in binary32, round half to even, saturate to 0..255;
// NaN becomes 0 (csrc/worker.cpp:26).
uint8_t float_to_u8(float v);

// The frame as upstream writes it (csrc/worker.cpp:20-42): rows flipped top to bottom, each channel through
// float_to_u8, channel order R, G, B in the file.
void frame_to_rgb8(const float* rgb_float, int width, int height, std::vector<uint8_t>& out);

}  // namespace rth
#endif
