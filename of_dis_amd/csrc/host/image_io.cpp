#include "image_io.h"

#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <fstream>

namespace ofdis_host {
namespace {

bool slurp(const std::string& path, std::vector<uint8_t>* buf) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  const std::streamoff n = f.tellg();
  f.seekg(0);
  buf->resize((size_t)n);
  f.read((char*)buf->data(), n);
  return (bool)f;
}

int pnm_int(const std::vector<uint8_t>& b, size_t* pos) {
  size_t p = *pos;
  for (;;) {
    while (p < b.size() && (b[p] == ' ' || b[p] == '\n' || b[p] == '\r' || b[p] == '\t')) ++p;
    if (p < b.size() && b[p] == '#') {
      while (p < b.size() && b[p] != '\n') ++p;
      continue;
    }
    break;
  }
  int v = 0;
  bool any = false;
  while (p < b.size() && b[p] >= '0' && b[p] <= '9') {
    v = v * 10 + (b[p] - '0');
    ++p;
    any = true;
  }
  *pos = p;
  return any ? v : -1;
}

bool read_pnm(const std::vector<uint8_t>& b, Image8* im, std::string* err) {
  const int ch = (b[1] == '5') ? 1 : 3;
  size_t pos = 2;
  const int w = pnm_int(b, &pos), h = pnm_int(b, &pos), maxv = pnm_int(b, &pos);
  if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 255) { *err = "unsupported PNM header (need 8-bit P5/P6)"; return false; }
  ++pos;  // single whitespace after maxval
  if (pos + (size_t)w * h * ch > b.size()) { *err = "PNM file is truncated"; return false; }
  im->width = w; im->height = h; im->channels = ch;
  im->data.assign(b.begin() + pos, b.begin() + pos + (size_t)w * h * ch);
  if (ch == 3)  // stored R,G,B -> B,G,R (cv::imread order)
    for (size_t i = 0; i < (size_t)w * h; ++i) std::swap(im->data[3 * i], im->data[3 * i + 2]);
  return true;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool read_png(const std::vector<uint8_t>& b, Image8* im, std::string* err) {
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte;
  while (pos + 12 <= b.size()) {
    const uint32_t len = be32(&b[pos]);
    const char* type = (const char*)&b[pos + 4];
    const uint8_t* d = &b[pos + 8];
    if (pos + 12 + len > b.size()) { *err = "PNG chunk overruns file"; return false; }
    if (!memcmp(type, "IHDR", 4)) {
      w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(d, d + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), d, d + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) { *err = "unsupported PNG (need 8-bit, non-interlaced)"; return false; }
  int spp;
  switch (ctype) {
    case 0: spp = 1; break;
    case 2: spp = 3; break;
    case 3: spp = 1; break;
    case 4: spp = 2; break;
    case 6: spp = 4; break;
    default: *err = "unsupported PNG colour type"; return false;
  }
  const size_t stride = (size_t)w * spp;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawlen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) {
    *err = "PNG inflate failed";
    return false;
  }
  std::vector<uint8_t> pix(stride * h);
  for (int y = 0; y < h; ++y) {  // undo the scanline filters
    const int ft = raw[(stride + 1) * y];
    const uint8_t* in = &raw[(stride + 1) * y + 1];
    uint8_t* cur = &pix[stride * y];
    const uint8_t* up = y ? &pix[stride * (y - 1)] : nullptr;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)spp ? cur[i - spp] : 0, bb = up ? up[i] : 0, c = (up && i >= (size_t)spp) ? up[i - spp] : 0;
      int v = in[i];
      switch (ft) {
        case 1: v += a; break;
        case 2: v += bb; break;
        case 3: v += (a + bb) >> 1; break;
        case 4: {
          const int p = a + bb - c, pa = abs(p - a), pb = abs(p - bb), pc = abs(p - c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
          break;
        }
        default: break;
      }
      cur[i] = (uint8_t)v;
    }
  }
  const bool color = (ctype == 2 || ctype == 6 || ctype == 3);
  im->width = w; im->height = h; im->channels = color ? 3 : 1;
  im->data.resize((size_t)w * h * im->channels);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    if (ctype == 0 || ctype == 4) {
      im->data[i] = pix[i * spp];
    } else {
      uint8_t r, g, bl;
      if (ctype == 3) {
        const size_t k = (size_t)pix[i] * 3;
        if (k + 2 >= plte.size()) { r = g = bl = 0; }
        else { r = plte[k]; g = plte[k + 1]; bl = plte[k + 2]; }
      } else {
        r = pix[i * spp]; g = pix[i * spp + 1]; bl = pix[i * spp + 2];
      }
      im->data[3 * i] = bl; im->data[3 * i + 1] = g; im->data[3 * i + 2] = r;
    }
  }
  return true;
}

}  // namespace

bool read_image(const std::string& path, int want_channels, Image8* out, std::string* err) {
  std::vector<uint8_t> b;
  if (!slurp(path, &b) || b.size() < 16) { *err = "cannot read " + path; return false; }
  Image8 im;
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  bool ok;
  if (b[0] == 'P' && (b[1] == '5' || b[1] == '6')) ok = read_pnm(b, &im, err);
  else if (!memcmp(b.data(), png_sig, 8)) ok = read_png(b, &im, err);
  else { *err = path + ": unsupported format (PGM/PPM binary or PNG)"; return false; }
  if (!ok) { *err = path + ": " + *err; return false; }
  if (im.channels == want_channels) { *out = std::move(im); return true; }
  Image8 cv;
  cv.width = im.width; cv.height = im.height; cv.channels = want_channels;
  const size_t n = (size_t)im.width * im.height;
  cv.data.resize(n * want_channels);
  if (want_channels == 1) {
    for (size_t i = 0; i < n; ++i) {
      const int bl = im.data[3 * i], g = im.data[3 * i + 1], r = im.data[3 * i + 2];
      cv.data[i] = (uint8_t)((r * 4899 + g * 9617 + bl * 1868 + 8192) >> 14);
    }
  } else {
    for (size_t i = 0; i < n; ++i) cv.data[3 * i] = cv.data[3 * i + 1] = cv.data[3 * i + 2] = im.data[i];
  }
  *out = std::move(cv);
  return true;
}

bool write_flo(const std::string& path, const float* flow_uv, int width, int height, std::string* err) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { *err = "WriteFile: could not open file " + path; return false; }
  bool ok = fwrite("PIEH", 1, 4, f) == 4;
  const int32_t w = width, h = height;
  ok = ok && fwrite(&w, sizeof(w), 1, f) == 1 && fwrite(&h, sizeof(h), 1, f) == 1;
  ok = ok && fwrite(flow_uv, sizeof(float), (size_t)2 * width * height, f) == (size_t)2 * width * height;
  fclose(f);
  if (!ok) *err = "WriteFile: problem writing data to " + path;
  return ok;
}

// SavePFMFile (run_dense.cpp:60-81): "Pf", size, scale -1.000000 (little endian), rows bottom-up, values negated
bool write_pfm(const std::string& path, const float* disp, int width, int height, std::string* err) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { *err = "WriteFile: could not open file " + path; return false; }
  fprintf(f, "Pf\n%d %d\n%f\n", width, height, (float)-1.0f);
  bool ok = true;
  for (int y = height - 1; y >= 0 && ok; --y)
    for (int x = 0; x < width; ++x) {
      const float t = -disp[(size_t)y * width + x];
      if (fwrite(&t, sizeof(float), 1, f) != 1) { ok = false; break; }
    }
  fclose(f);
  if (!ok) *err = "WriteFile: problem writing data to " + path;
  return ok;
}

}  // namespace ofdis_host
