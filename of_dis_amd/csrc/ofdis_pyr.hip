// ofdis_pyr.hip -- on-device image pyramid from raw 8-bit frames, i.e. the part of
// run_dense.cpp that feeds OFClass: replicate-pad to a multiple of 2^sc_f (run_dense.cpp:298-311),
// convertTo float (326-327), ConstructImgPyramide (130-178): cv::resize x0.5 INTER_LINEAR per level,
// cv::Sobel(ksize 3, scale 1/8, BORDER_DEFAULT), copyMakeBorder(REPLICATE for images, CONSTANT 0 for
// gradients) by imgpadding.
//
// Exactness.  For 8-bit input every level-l value is the mean of a 2^l x 2^l block of integers: a
// dyadic rational with at most 8+2l significant bits, and every Sobel partial sum (two differences, one doubled) one
// with at most 10+2l: exactly representable in fp32 for l <= 7 -- the launcher's limit (at l = 8 the 26-bit partial
// sums would round, and OpenCV's f1*(S0+S2) + f0*S1 and this file's (d0 + 2*d1) + d2 could differ in the last bit).
// All fp32 sums involved are therefore exact, so the
// hierarchical 2x2 means OpenCV computes, an integer block sum scaled by 4^-l, and any summation
// order give bit-identical results.  Only levels sc_l..sc_f are materialised (the reference builds
// all of 0..sc_f, run_dense.cpp:132, and never reads the rest), and image B gets no gradients (never
// read when usefbcon == 0).
#include "ofdis_kernels.h"

namespace ofdis {

// level `l` image (unpadded, [B][h][w][noc] float) straight from the u8 frames
__global__ __launch_bounds__(256) void pyr_base_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                       int nframes, int wo, int ho, int W, int H, int noc, int l) {
  const int w = W >> l, h = H >> l;
  const int left = (W - wo) / 2, top = (H - ho) / 2;  // floor(pad/2) on the left/top (run_dense.cpp:308)
  const long long total = (long long)nframes * h * w * noc;
  const int bs = 1 << l;
  const float scale = 1.0f / (float)(1 << (2 * l));
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % noc);
    long long r = idx / noc;
    const int x = (int)(r % w);
    r /= w;
    const int y = (int)(r % h);
    const int f = (int)(r / h);
    const uint8_t* s = src + (size_t)f * wo * ho * noc;
    unsigned sum = 0;
    // fast path (gray, block of at least 4 columns entirely inside the frame, 4-byte aligned rows): whole words,
    // v_sad_u8 against 0 adds the four bytes of a word in one instruction; neighbouring threads read neighbouring words
    const int bx0 = x * bs - left, by0 = y * bs - top;
    if (noc == 1 && bs >= 4 && ((wo | bx0) & 3) == 0 && bx0 >= 0 && bx0 + bs <= wo && by0 >= 0 && by0 + bs <= ho &&
        (((size_t)f * wo * ho) & 3) == 0 && ((uintptr_t)src & 3) == 0) {
      const unsigned* s32 = reinterpret_cast<const unsigned*>(s + (size_t)by0 * wo + bx0);
      const int wpr = wo >> 2;
      for (int yy = 0; yy < bs; ++yy)
        for (int q = 0; q < (bs >> 2); ++q) sum = __builtin_amdgcn_sad_u8(s32[(size_t)yy * wpr + q], 0u, sum);
      dst[idx] = (float)sum * scale;
      continue;
    }
    for (int yy = 0; yy < bs; ++yy) {
      const int sy = clampi(y * bs + yy - top, 0, ho - 1);
      for (int xx = 0; xx < bs; ++xx) {
        const int sx = clampi(x * bs + xx - left, 0, wo - 1);
        sum += s[((size_t)sy * wo + sx) * noc + c];
      }
    }
    dst[idx] = (float)sum * scale;
  }
}

// Streaming variant of pyr_base_kernel for gray frames whose rows and left padding are multiples of 16 bytes: a lane reads
// 16 bytes per source row (one wavefront = 1 KB of a row per instruction), sums them per output block with v_sad_u8
// (four bytes per instruction) over the BS rows of the block -- rows are clamped individually, so the replicate padding
// at the top and bottom costs nothing -- and writes 16/BS adjacent outputs.  Same integers, same power-of-two scale:
// bit-identical to the generic kernel.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int BS>
__global__ __launch_bounds__(256) void pyr_base16_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                         int nframes, int wo, int ho, int W, int H, int l) {
  constexpr int NOUT = 16 / BS;  // outputs per lane
  const int w = W >> l, h = H >> l;
  const int left = (W - wo) / 2, top = (H - ho) / 2;
  const int chunks = W / 16;  // lanes per output row
  const long long total = (long long)nframes * h * chunks;
  const float scale = 1.0f / (float)(BS * BS);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % chunks);
    long long r = idx / chunks;
    const int y = (int)(r % h);
    const int f = (int)(r / h);
    // padded columns [16 ch, 16 ch + 16) = source columns clamped to the frame (whole chunk inside or outside)
    int sx = ch * 16 - left;
    const bool inside = sx >= 0 && sx + 16 <= wo;
    const uint8_t* s = src + (size_t)f * wo * ho;
    unsigned sum[NOUT];
#pragma unroll
    for (int q = 0; q < NOUT; ++q) sum[q] = 0;
    if (inside) {
#pragma unroll
      for (int yy = 0; yy < BS; ++yy) {
        const int sy = clampi(y * BS + yy - top, 0, ho - 1);
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(s + (size_t)sy * wo + sx));
        const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) sum[q * 4 / BS] = __builtin_amdgcn_sad_u8(wd[q], 0u, sum[q * 4 / BS]);
      }
    } else {  // chunk in the replicated left / right border: every column is the frame's first / last column
      sx = sx < 0 ? 0 : wo - 1;
      unsigned c = 0;
      for (int yy = 0; yy < BS; ++yy) c += s[(size_t)clampi(y * BS + yy - top, 0, ho - 1) * wo + sx];
#pragma unroll
      for (int q = 0; q < NOUT; ++q) sum[q] = c * BS;
    }
    float* o = dst + ((size_t)f * h + y) * w + (size_t)ch * NOUT;
#pragma unroll
    for (int q = 0; q < NOUT; ++q) o[q] = (float)sum[q] * scale;
  }
}

// next coarser level: 2x2 mean (cv::resize(.5,.5,INTER_LINEAR))
__global__ __launch_bounds__(256) void pyr_down_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       int nframes, int w, int h, int noc) {
  const int w2 = w / 2, h2 = h / 2;
  const long long total = (long long)nframes * h2 * w2 * noc;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % noc);
    long long r = idx / noc;
    const int x = (int)(r % w2);
    r /= w2;
    const int y = (int)(r % h2);
    const int f = (int)(r / h2);
    const float* s = src + (size_t)f * w * h * noc;
    const float a = s[((size_t)(2 * y) * w + 2 * x) * noc + c], b = s[((size_t)(2 * y) * w + 2 * x + 1) * noc + c];
    const float cc = s[((size_t)(2 * y + 1) * w + 2 * x) * noc + c], d = s[((size_t)(2 * y + 1) * w + 2 * x + 1) * noc + c];
    dst[idx] = ((a + b) + (cc + d)) * 0.25f;
  }
}

// padded planes of one level: image (replicate border) and, if dx != nullptr, Sobel/8 gradients
// (zero border).  grid = (chunks of 256 plane elements, frame): all index arithmetic is 32-bit.
__global__ __launch_bounds__(256) void pyr_planes_kernel(const float* __restrict__ src, float* __restrict__ img,
                                                         float* __restrict__ dx, float* __restrict__ dy, int nframes,
                                                         int w, int h, int noc, int pad) {
  const int tw = w + 2 * pad, th = h + 2 * pad;
  const unsigned per_frame = (unsigned)(th * tw * noc);
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  const int f = blockIdx.y;
  if (e >= per_frame) return;
  const unsigned pe = e / (unsigned)noc;  // noc is 1 or 3
  const int c = (int)(e - pe * (unsigned)noc);
  const int Y = (int)(pe / (unsigned)tw), X = (int)(pe - (unsigned)Y * (unsigned)tw);
  const size_t idx = (size_t)f * per_frame + e;
  const float* s = src + (size_t)f * w * h * noc;
  const int x = X - pad, y = Y - pad;
  img[idx] = s[(clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)) * noc + c];
  if (dx) {
    float gx = 0.0f, gy = 0.0f;
    if (x >= 0 && x < w && y >= 0 && y < h) {
      float v[3][3];
      // BORDER_REFLECT_101 of a 3-tap window: one reflection suffices (h, w >= 2; a one-pixel image repeats its pixel)
      const int ym = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yp = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
      const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
      const int ys[3] = {ym, y, yp}, xs[3] = {xm, x, xp};
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) v[j][i] = s[(ys[j] * w + xs[i]) * noc + c];
      gx = ((v[0][2] - v[0][0]) + 2.0f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0])) * 0.125f;
      gy = ((v[2][0] - v[0][0]) + 2.0f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2])) * 0.125f;
    }
    dx[idx] = gx;
    dy[idx] = gy;
  }
}

// Gray planes, four padded-plane columns per thread (round 4): the kernel above issues ten 4-byte loads and three 4-byte
// stores per element -- it is bound by its memory instructions, not its bytes.  Here a thread owns the quad X0 .. X0+3 of a
// padded row (X0 a multiple of 4; needs w and pad multiples of 4): the three source rows of the Sobel window are read once
// per quad (one 16-byte load + two scalars each where the quad lies inside the image, clamped / reflected scalars at its
// border), the three planes are written with 16-byte stores, and -- `down` -- the quad's two pixels of the NEXT level's
// image (2x2 means, cv::resize(.5,.5,INTER_LINEAR), the expression of pyr_down_kernel) come from the same rows, so that
// level is read once instead of twice.  Same expressions per element as pyr_planes_kernel / pyr_down_kernel: same bits.
typedef float f4p __attribute__((ext_vector_type(4)));
typedef float f4pu __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void pyr_planes_gray4_kernel(const float* __restrict__ src, float* __restrict__ img,
                                                               float* __restrict__ dx, float* __restrict__ dy,
                                                               float* __restrict__ down, int nframes, int w, int h, int pad) {
  const int tw = w + 2 * pad, th = h + 2 * pad, qpr = tw >> 2;  // quads per padded row
  const unsigned q = blockIdx.x * 256u + threadIdx.x;
  const int f = blockIdx.y;
  if (q >= (unsigned)(qpr * th)) return;
  const int Y = (int)(q / (unsigned)qpr), X0 = (int)(q - (unsigned)Y * (unsigned)qpr) * 4;
  const int x0 = X0 - pad, y = Y - pad;
  const float* s = src + (size_t)f * w * h;
  const size_t o = (size_t)f * tw * th + (size_t)Y * tw + X0;
  const bool yin = (y >= 0) & (y < h), xin = (x0 >= 0) & (x0 + 3 < w);  // (w, pad multiples of 4: a quad is inside or outside)
  const bool xinner = (x0 >= 1) & (x0 + 4 <= w - 1);
  // the six columns x0-1 .. x0+4 of source row r, replicate-clamped for the image, reflect-101 for the Sobel window: inside
  // the image the two rules only differ at columns -1 and w, which the image plane never uses
  auto row6 = [&](int r, bool reflect, float (&v)[6]) {
    const float* p = s + (size_t)r * w;
    if (xinner) {
      const f4pu t = *reinterpret_cast<const f4pu*>(p + x0);
      v[0] = p[x0 - 1]; v[1] = t.x; v[2] = t.y; v[3] = t.z; v[4] = t.w; v[5] = p[x0 + 4];
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        int x = x0 - 1 + i;
        if (reflect) x = x < 0 ? (w > 1 ? -x : 0) : (x > w - 1 ? (w > 1 ? 2 * (w - 1) - x : 0) : x);
        v[i] = p[clampi(x, 0, w - 1)];
      }
    }
  };
  float mid[6];
  row6(clampi(y, 0, h - 1), yin & xin, mid);
  // (outside the image: row and columns clamped = the replicate border of the image plane)
  *reinterpret_cast<f4p*>(img + o) = f4p{mid[1], mid[2], mid[3], mid[4]};
  float up[6], lo[6];
  const bool need_lo = yin & xin & ((dx != nullptr) | ((down != nullptr) & ((y & 1) == 0)));
  if (need_lo) row6(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0), true, lo);
  if (dx) {
    f4p gx = {0.0f, 0.0f, 0.0f, 0.0f}, gy = {0.0f, 0.0f, 0.0f, 0.0f};
    if (yin & xin) {  // Sobel / 8 with BORDER_REFLECT_101 (the expression of pyr_planes_kernel)
      row6(y > 0 ? y - 1 : (h > 1 ? 1 : 0), true, up);
      float gxs[4], gys[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gxs[k] = ((up[k + 2] - up[k]) + 2.0f * (mid[k + 2] - mid[k]) + (lo[k + 2] - lo[k])) * 0.125f;
        gys[k] = ((lo[k] - up[k]) + 2.0f * (lo[k + 1] - up[k + 1]) + (lo[k + 2] - up[k + 2])) * 0.125f;
      }
      gx = f4p{gxs[0], gxs[1], gxs[2], gxs[3]};
      gy = f4p{gys[0], gys[1], gys[2], gys[3]};
    }
    *reinterpret_cast<f4p*>(dx + o) = gx;
    *reinterpret_cast<f4p*>(dy + o) = gy;
  }
  if (down && yin && xin && (y & 1) == 0) {  // (h, w even here: the launcher only passes `down` then; row y + 1 exists)
    const int w2 = w >> 1;
    float2* d = reinterpret_cast<float2*>(down + (size_t)f * w2 * (h >> 1) + (size_t)(y >> 1) * w2 + (x0 >> 1));
    *d = make_float2(((mid[1] + mid[2]) + (lo[1] + lo[2])) * 0.25f, ((mid[3] + mid[4]) + (lo[3] + lo[4])) * 0.25f);
  }
}

static unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > (1 << 20)) b = 1 << 20;
  if (b < 1) b = 1;
  return (unsigned)b;
}

hipError_t launch_pyr_base(const uint8_t* src, float* dst, int nframes, int wo, int ho, int W, int H, int noc, int l,
                           hipStream_t s) {
  const long long total = (long long)nframes * (H >> l) * (W >> l) * noc;
  const int left = (W - wo) / 2;
  // 16-byte streaming variant: gray, rows / left padding / frame size / base address multiples of 16 bytes
  if (noc == 1 && (l == 2 || l == 3 || l == 4) && (wo & 15) == 0 && (left & 15) == 0 && (W & 15) == 0 &&
      (((size_t)wo * ho) & 15) == 0 && ((uintptr_t)src & 15) == 0) {
    const long long lanes = (long long)nframes * (H >> l) * (W / 16);
    if (l == 2) hipLaunchKernelGGL(pyr_base16_kernel<4>, dim3(grid_for(lanes)), dim3(256), 0, s, src, dst, nframes, wo, ho, W, H, l);
    else if (l == 3) hipLaunchKernelGGL(pyr_base16_kernel<8>, dim3(grid_for(lanes)), dim3(256), 0, s, src, dst, nframes, wo, ho, W, H, l);
    else hipLaunchKernelGGL(pyr_base16_kernel<16>, dim3(grid_for(lanes)), dim3(256), 0, s, src, dst, nframes, wo, ho, W, H, l);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(pyr_base_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, dst, nframes, wo, ho, W, H, noc, l);
  return hipGetLastError();
}
// the quad kernel also produces the next level's image (even sizes, gray, quads inside or outside the image)
bool pyr_planes_fuses_down(int w, int h, int noc, int pad) {
  return noc == 1 && (w & 3) == 0 && (pad & 3) == 0 && (h & 1) == 0 && w >= 8 && h >= 2;
}
hipError_t launch_pyr_down(const float* src, float* dst, int nframes, int w, int h, int noc, hipStream_t s) {
  const long long total = (long long)nframes * (h / 2) * (w / 2) * noc;
  hipLaunchKernelGGL(pyr_down_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, dst, nframes, w, h, noc);
  return hipGetLastError();
}
hipError_t launch_pyr_planes(const float* src, float* img, float* dx, float* dy, int nframes, int w, int h, int noc,
                             int pad, hipStream_t s, float* down) {
  const long long per_frame = (long long)(h + 2 * pad) * (w + 2 * pad) * noc;
  if (nframes > 65535 || per_frame >= (1ll << 31)) return hipErrorInvalidValue;
  if (pyr_planes_fuses_down(w, h, noc, pad) || (!down && noc == 1 && (w & 3) == 0 && (pad & 3) == 0 && w >= 8 && h >= 2)) {
    // gray, quads: one launch for the three planes and (down != nullptr) the next level's image
    hipLaunchKernelGGL(pyr_planes_gray4_kernel, dim3((unsigned)((per_frame / 4 + 255) / 256), (unsigned)nframes), dim3(256), 0, s,
                       src, img, dx, dy, down, nframes, w, h, pad);
    return hipGetLastError();
  }
  if (down) {  // no fused form for this geometry: the caller asked for the next level too
    hipError_t e = launch_pyr_down(src, down, nframes, w, h, noc, s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(pyr_planes_kernel, dim3((unsigned)((per_frame + 255) / 256), (unsigned)nframes), dim3(256), 0, s, src, img,
                     dx, dy, nframes, w, h, noc, pad);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ result to full resolution
// run_dense.cpp:406-414: flowout *= 2^lv_l; cv::resize(flowout, x 2^lv_l, INTER_LINEAR); crop the padding.
// cv::resize bilinear for CV_32FC2: half-pixel centres, source index clamped with the fraction forced to 0 at
// the borders, horizontal interpolation first.  2^lv_l is a power of two, so (X + 0.5) / s - 0.5 is exact in fp32.
// One thread per output pixel, 8-byte stores; the source (57 KB per frame at op-point 2) is L2 resident.
// grid = (x chunks of 512 pixels, groups of 2^sc_l output rows that share their two source rows, frame); a thread owns
// two adjacent output columns (16 bytes per store): it interpolates them horizontally on the two source rows ONCE and then
// writes the <= 2^sc_l rows of the group, which differ only in the vertical weight.  The output is written once
// and never read by this library: non-temporal stores.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void upsample_h(const float2* __restrict__ fl, int sw, int sy, int sy1, int X, float inv,
                                           float scf, bool scale, float2& r0, float2& r1) {
  float fx = ((float)X + 0.5f) * inv - 0.5f;
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { sx = 0; fx = 0.0f; }
  if (sx >= sw - 1) { sx = sw - 1; fx = 0.0f; }
  const int sx1 = min(sx + 1, sw - 1);
  float2 v00 = fl[sy * sw + sx], v01 = fl[sy * sw + sx1], v10 = fl[sy1 * sw + sx], v11 = fl[sy1 * sw + sx1];
  if (scale) {
    v00.x *= scf; v00.y *= scf; v01.x *= scf; v01.y *= scf;
    v10.x *= scf; v10.y *= scf; v11.x *= scf; v11.y *= scf;
  }
  const float ax = 1.0f - fx;
  r0 = make_float2(v00.x * ax + v01.x * fx, v00.y * ax + v01.y * fx);
  r1 = make_float2(v10.x * ax + v11.x * fx, v10.y * ax + v11.y * fx);
}

__global__ __launch_bounds__(256) void upsample_crop_kernel(const float2* __restrict__ flow, float2* __restrict__ out,
                                                            int sw, int sh, int sc_l, int left, int top, int wo, int ho) {
  const int f = blockIdx.z;
  const int x = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (x >= wo) return;
  const int s = 1 << sc_l;
  const float scf = (float)s, inv = 1.0f / scf;
  // the padded rows with floor((Y + 0.5) / s - 0.5) = k are [k*s + s/2, k*s + s/2 + s); k = -1 (s > 1 only) and k = sh-1
  // are the half groups at the borders, where the source row is clamped and the weight forced to 0
  const int k = (int)blockIdx.y - (s > 1 ? 1 : 0);
  const int Y0 = max(k * s + s / 2, top), Y1 = min(k * s + s / 2 + s, top + ho);  // rows of the group inside the crop
  if (Y0 >= Y1) return;
  float fy0 = ((float)Y0 + 0.5f) * inv - 0.5f;
  int sy = (int)floorf(fy0);
  const bool clamp_lo = sy < 0, clamp_hi = sy >= sh - 1;
  if (clamp_lo) sy = 0;
  if (clamp_hi) sy = sh - 1;
  const int sy1 = min(sy + 1, sh - 1);
  const float2* fl = flow + (size_t)f * sw * sh;
  float2 a0, a1, b0, b1;
  upsample_h(fl, sw, sy, sy1, x + left, inv, scf, sc_l > 0, a0, a1);
  const bool two = x + 1 < wo;
  if (two) upsample_h(fl, sw, sy, sy1, x + 1 + left, inv, scf, sc_l > 0, b0, b1);
  const bool vec = two && (wo & 1) == 0;  // rows are 16-byte aligned when wo is even
  for (int Y = Y0; Y < Y1; ++Y) {
    float fy = ((float)Y + 0.5f) * inv - 0.5f;
    fy -= floorf(fy);
    if (clamp_lo || clamp_hi) fy = 0.0f;
    const float ay = 1.0f - fy;
    float2* o = out + ((size_t)f * ho + (Y - top)) * wo + x;
    const float2 a = make_float2(a0.x * ay + a1.x * fy, a0.y * ay + a1.y * fy);
    if (two) {
      const float2 b = make_float2(b0.x * ay + b1.x * fy, b0.y * ay + b1.y * fy);
      if (vec) {
        __builtin_nontemporal_store((f4v){a.x, a.y, b.x, b.y}, reinterpret_cast<f4v*>(o));
      } else {
        o[0] = a;
        o[1] = b;
      }
    } else {
      o[0] = a;
    }
  }
}

// one-channel result of the stereo-depth mode (CV_32FC1, same resize): one pixel per thread
__global__ __launch_bounds__(256) void upsample_crop1_kernel(const float* __restrict__ flow, float* __restrict__ out,
                                                             int sw, int sh, int sc_l, int left, int top, int wo, int ho) {
  const int f = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= wo) return;
  const float scf = (float)(1 << sc_l), inv = 1.0f / scf;
  float fy = ((float)(y + top) + 0.5f) * inv - 0.5f;
  int sy = (int)floorf(fy);
  fy -= (float)sy;
  if (sy < 0) { sy = 0; fy = 0.0f; }
  if (sy >= sh - 1) { sy = sh - 1; fy = 0.0f; }
  const int sy1 = min(sy + 1, sh - 1);
  float fx = ((float)(x + left) + 0.5f) * inv - 0.5f;
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { sx = 0; fx = 0.0f; }
  if (sx >= sw - 1) { sx = sw - 1; fx = 0.0f; }
  const int sx1 = min(sx + 1, sw - 1);
  const float* fl = flow + (size_t)f * sw * sh;
  float v00 = fl[sy * sw + sx], v01 = fl[sy * sw + sx1], v10 = fl[sy1 * sw + sx], v11 = fl[sy1 * sw + sx1];
  if (sc_l > 0) { v00 *= scf; v01 *= scf; v10 *= scf; v11 *= scf; }
  const float ax = 1.0f - fx, ay = 1.0f - fy;
  const float r0 = v00 * ax + v01 * fx, r1 = v10 * ax + v11 * fx;
  out[((size_t)f * ho + y) * wo + x] = r0 * ay + r1 * fy;
}

hipError_t launch_upsample_crop(const float* flow, float* out, int nframes, int sw, int sh, int sc_l, int left, int top,
                                int wo, int ho, int channels, hipStream_t s) {
  if (ho > 65535 || nframes > 65535) return hipErrorInvalidValue;
  if (channels == 1) {
    hipLaunchKernelGGL(upsample_crop1_kernel, dim3((wo + 255) / 256, ho, nframes), dim3(256), 0, s, flow, out, sw, sh,
                       sc_l, left, top, wo, ho);
    return hipGetLastError();
  }
  // row groups k = -1 .. sh-1 (k = 0 .. sh for sc_l = 0, the last one empty)
  hipLaunchKernelGGL(upsample_crop_kernel, dim3((wo + 511) / 512, sh + 1, nframes), dim3(256), 0, s, (const float2*)flow,
                     (float2*)out, sw, sh, sc_l, left, top, wo, ho);
  return hipGetLastError();
}

}  // namespace ofdis
