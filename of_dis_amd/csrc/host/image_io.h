// image_io.h -- minimal 8-bit image reading (PGM/PPM binary, PNG via zlib) and Middlebury .flo writing for
// the run_OF_* executables.  The reference uses cv::imread (run_dense.cpp:208-209); OpenCV is not a
// dependency here.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace ofdis_host {

struct Image8 {
  int width = 0, height = 0, channels = 0;  // channels: 1 (gray) or 3 (B,G,R interleaved like cv::imread)
  std::vector<uint8_t> data;
};

// Reads P5/P6 PNM or PNG (8-bit gray / gray+alpha / RGB / RGBA / palette, non-interlaced).
// want_channels = 1: colour input is converted with OpenCV's fixed-point BGR2GRAY
// (Y = (4899 R + 9617 G + 1868 B + 8192) >> 14); = 3: gray input is replicated.
bool read_image(const std::string& path, int want_channels, Image8* out, std::string* err);

// SaveFlowFile (run_dense.cpp:16-57): "PIEH", int32 width, int32 height, then rows of (float u, float v).
bool write_flo(const std::string& path, const float* flow_uv, int width, int height, std::string* err);
// stereo-depth result (run_dense.cpp:60-81): one channel, PFM
bool write_pfm(const std::string& path, const float* disp, int width, int height, std::string* err);

}  // namespace ofdis_host
