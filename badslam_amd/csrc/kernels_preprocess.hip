// kernels_preprocess.hip -- what Keyframe's convenience constructor runs on its inputs
// (B/keyframe.cc:96-144, B/ = applications/badslam/src/badslam/): luma, normals, point radii,
// min/max depth.  Pixel kernels, 64x4 blocks (one wave64 per image row segment -> coalesced
// 128-byte u16 rows).
#include <hip/hip_fp16.h>

#include "ba_device.h"

namespace bahip {

constexpr int kPxBlockX = 64, kPxBlockY = 4;

// B/cuda_image_processing.cu:165-175
__global__ void __launch_bounds__(kPxBlockX* kPxBlockY)
brightness_kernel(const uint8_t* __restrict__ rgb, uint32_t rgb_pitch, uint8_t* __restrict__ rgba, uint32_t rgba_pitch,
                  int width, int height) {
  const int x = blockIdx.x * kPxBlockX + threadIdx.x, y = blockIdx.y * kPxBlockY + threadIdx.y;
  if (x >= width || y >= height) return;
  const uint8_t* p = rgb + (size_t)y * rgb_pitch + 3 * x;
  const uint8_t r = p[0], g = p[1], b = p[2];
  const uint8_t intensity = (uint8_t)(__builtin_fmaf(0.114f, (float)b, __builtin_fmaf(0.587f, (float)g, 0.299f * (float)r)) + 0.5f);
  *reinterpret_cast<uchar4*>(rgba + (size_t)y * rgba_pitch + 4 * x) = make_uchar4(r, g, b, intensity);
}

// B/cuda_depth_processing.cu:42-128 (BadSlam::PreprocessFrame runs it on every incoming depth image, B/bad_slam.cc:697-706).
// A 64x4 block stages its (64 + 2r) x (4 + 2r) neighbourhood in LDS once; every pixel then reads its (2r+1)^2 window
// from there instead of issuing up to 49 global halfword loads.
constexpr int kBilateralMaxRadius = 8;
__global__ void __launch_bounds__(kPxBlockX* kPxBlockY)
bilateral_filter_kernel(float denom_xy, float denom_value, int radius, int radius_squared, uint16_t max_depth, float raw_to_float_depth,
                        const uint16_t* __restrict__ in, uint32_t in_pitch, uint16_t* __restrict__ out, uint32_t out_pitch, int width,
                        int height) {
  __shared__ uint16_t tile[(kPxBlockY + 2 * kBilateralMaxRadius) * (kPxBlockX + 2 * kBilateralMaxRadius)];
  const int tw = kPxBlockX + 2 * radius, th = kPxBlockY + 2 * radius;
  const int x0 = blockIdx.x * kPxBlockX - radius, y0 = blockIdx.y * kPxBlockY - radius;
  for (int t = threadIdx.y * kPxBlockX + threadIdx.x; t < tw * th; t += kPxBlockX * kPxBlockY) {
    const int tx = t % tw, ty = t / tw, gx = x0 + tx, gy = y0 + ty;
    tile[t] = (gx >= 0 && gy >= 0 && gx < width && gy < height) ? pitched_load(in, in_pitch, gy, gx) : (uint16_t)0;   // 0 = "no sample"
  }
  __syncthreads();
  const int x = blockIdx.x * kPxBlockX + threadIdx.x, y = blockIdx.y * kPxBlockY + threadIdx.y;
  if (x >= width || y >= height) return;
  uint16_t* o = pitched_ptr(out, out_pitch, y, x);
  const uint16_t center_value = tile[(threadIdx.y + radius) * tw + threadIdx.x + radius];
  if (center_value == 0 || center_value > max_depth) { *o = kUnknownDepth; return; }
  const float inv_center_value = 1.0f / (raw_to_float_depth * center_value);
  float sum = 0, weight = 0;
  // the reference clips the window to the image; outside pixels are zeros in the tile and are skipped like missing samples
  for (int dy = -radius; dy <= radius; ++dy)
    for (int dx = -radius; dx <= radius; ++dx) {
      const int grid_distance_squared = dx * dx + dy * dy;
      if (grid_distance_squared > radius_squared) continue;
      const uint16_t sample = tile[(threadIdx.y + radius + dy) * tw + threadIdx.x + radius + dx];
      if (sample == 0) continue;
      const float inv_sample = 1.0f / (raw_to_float_depth * sample);
      float value_distance_squared = inv_center_value - inv_sample;
      value_distance_squared *= value_distance_squared;
      const float w = exp_det(-grid_distance_squared / denom_xy + -value_distance_squared / denom_value);
      sum += w * inv_sample;
      weight += w;
    }
  *o = (weight == 0) ? kUnknownDepth : (uint16_t)(1.0f / (raw_to_float_depth * sum / weight));
}

// B/cuda_depth_processing.cu:134-264
__global__ void __launch_bounds__(kPxBlockX* kPxBlockY)
normals_from_depth_kernel(Intrinsics in, const uint16_t* __restrict__ in_depth, uint32_t in_pitch,
                          uint16_t* __restrict__ out_depth, uint32_t out_pitch, uint16_t* __restrict__ out_normals,
                          uint32_t normals_pitch) {
  const int x = blockIdx.x * kPxBlockX + threadIdx.x, y = blockIdx.y * kPxBlockY + threadIdx.y;
  const int W = in.width, H = in.height;
  if (x >= W || y >= H) return;
  uint16_t* od = pitched_ptr(out_depth, out_pitch, y, x);
  uint16_t* on = pitched_ptr(out_normals, normals_pitch, y, x);
  const uint16_t zero_normal = pack_normal8(0, 0);
  if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) { *od = kUnknownDepth; *on = zero_normal; return; }
  const uint16_t center_raw = pitched_load(in_depth, in_pitch, y, x);
  if (center_raw & kInvalidDepthBit) { *od = kUnknownDepth; *on = zero_normal; return; }
  const uint16_t right_raw = pitched_load(in_depth, in_pitch, y, x + 1);
  const uint16_t left_raw = pitched_load(in_depth, in_pitch, y, x - 1);
  const uint16_t bottom_raw = pitched_load(in_depth, in_pitch, y + 1, x);
  const uint16_t top_raw = pitched_load(in_depth, in_pitch, y - 1, x);
  if ((right_raw | left_raw | bottom_raw | top_raw) & kInvalidDepthBit) { *od = kUnknownDepth; *on = zero_normal; return; }

  const float cd = raw_to_calibrated_depth(in.a, cfactor_at(in, x, y), in.raw_to_float_depth, center_raw);
  const float ld = raw_to_calibrated_depth(in.a, cfactor_at(in, x - 1, y), in.raw_to_float_depth, left_raw);
  const float td = raw_to_calibrated_depth(in.a, cfactor_at(in, x, y - 1), in.raw_to_float_depth, top_raw);
  const float rd = raw_to_calibrated_depth(in.a, cfactor_at(in, x + 1, y), in.raw_to_float_depth, right_raw);
  const float bd = raw_to_calibrated_depth(in.a, cfactor_at(in, x, y + 1), in.raw_to_float_depth, bottom_raw);
  const Vec3 lp = unproject(in, x - 1, y, ld), tp = unproject(in, x, y - 1, td), rp = unproject(in, x + 1, y, rd);
  const Vec3 bp = unproject(in, x, y + 1, bd), cp = unproject(in, x, y, cd);

  constexpr float kRatioThresholdSquared = 2.f * 2.f;
  const float left_sq = sqlen3(lp - cp), right_sq = sqlen3(rp - cp);
  const float lr_ratio = left_sq / right_sq;
  Vec3 left_to_right;
  if (lr_ratio < kRatioThresholdSquared && lr_ratio > 1.f / kRatioThresholdSquared) left_to_right = rp - lp;
  else if (left_sq < right_sq) left_to_right = cp - lp;
  else left_to_right = rp - cp;
  const float bottom_sq = sqlen3(bp - cp), top_sq = sqlen3(tp - cp);
  const float bt_ratio = bottom_sq / top_sq;
  Vec3 bottom_to_top;
  if (bt_ratio < kRatioThresholdSquared && bt_ratio > 1.f / kRatioThresholdSquared) bottom_to_top = tp - bp;
  else if (bottom_sq < top_sq) bottom_to_top = cp - bp;
  else bottom_to_top = tp - cp;

  Vec3 normal = cross3(left_to_right, bottom_to_top);
  const float length = norm3(normal);
  if (!(length > 1e-6f)) {
    normal = mk3(0, 0, -1);
  } else {
    const float inv_length = ((in.fy_inv < 0) ? -1.0f : 1.0f) / length;
    normal.x *= inv_length;
    normal.y *= inv_length;
  }
  *on = pack_normal8(normal.x, normal.y);
  *od = center_raw;
}

// B/cuda_depth_processing.cu:289-360 with min_neighbors_for_radius_computation = 4
__global__ void __launch_bounds__(kPxBlockX* kPxBlockY)
point_radii_kernel(Intrinsics in, float raw_to_float_depth, const uint16_t* __restrict__ depth, uint32_t depth_pitch,
                   uint16_t* __restrict__ radius, uint32_t radius_pitch, uint16_t* __restrict__ out_depth, uint32_t out_pitch) {
  const int x = blockIdx.x * kPxBlockX + threadIdx.x, y = blockIdx.y * kPxBlockY + threadIdx.y;
  const int W = in.width, H = in.height;
  if (x >= W || y >= H) return;
  const uint16_t d16 = pitched_load(depth, depth_pitch, y, x);
  if (d16 & kInvalidDepthBit) { *pitched_ptr(out_depth, out_pitch, y, x) = kUnknownDepth; return; }
  const float d = raw_to_float_depth * d16;
  const Vec3 local = mk3(d * (in.fx_inv * x + in.cx_inv), d * (in.fy_inv * y + in.cy_inv), d);
  int neighbor_count = 0;
  float min_sq = __builtin_huge_valf();
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    // same visiting order as the reference's dy/dx double loop: top, left, right, bottom
    const int dx = x + ((n == 1) ? -1 : (n == 2) ? 1 : 0);
    const int dy = y + ((n == 0) ? -1 : (n == 3) ? 1 : 0);
    if (dx < 0 || dy < 0 || dx >= W || dy >= H) continue;
    const uint16_t nd16 = pitched_load(depth, depth_pitch, dy, dx);
    if (nd16 & kInvalidDepthBit) continue;
    ++neighbor_count;
    const float nd = raw_to_float_depth * nd16;
    const Vec3 other = mk3(nd * (in.fx_inv * dx + in.cx_inv), nd * (in.fy_inv * dy + in.cy_inv), nd);
    const float dist_sq = sqlen3(other - local);
    if (dist_sq < min_sq) min_sq = dist_sq;
  }
  const bool valid = neighbor_count >= 4;
  *pitched_ptr(radius, radius_pitch, y, x) = __half_as_ushort(__float2half_rn(valid ? min_sq : 0.f));
  *pitched_ptr(out_depth, out_pitch, y, x) = valid ? d16 : kUnknownDepth;
}

// B/cuda_depth_processing.cu:391-428: positive floats order like their bit patterns.
// A fixed grid of 64 x 256 threads strides over the image (row segments of 64 pixels per wavefront), so only 256
// wavefronts contend for the two result words.
constexpr int kMinMaxBlocks = 64;
__global__ void __launch_bounds__(256)
min_max_depth_kernel(const uint16_t* __restrict__ depth, uint32_t depth_pitch, int width, int height,
                     float raw_to_float_depth, int* __restrict__ result /* [0]=min bits, [1]=max bits */) {
  float mn = __builtin_huge_valf(), mx = 0.f;
  const int segs_per_row = (width + 63) / 64;
  const int total_segs = segs_per_row * height;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, num_waves = kMinMaxBlocks * 4, lane = threadIdx.x & 63;
  for (int seg = wave; seg < total_segs; seg += num_waves) {
    const int y = seg / segs_per_row, x = (seg % segs_per_row) * 64 + lane;
    if (x < width) {
      const uint16_t d16 = pitched_load(depth, depth_pitch, y, x);
      if (!(d16 & kInvalidDepthBit)) {
        const float d = raw_to_float_depth * d16;
        mn = fminf(mn, d);
        mx = fmaxf(mx, d);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off));
    mx = fmaxf(mx, __shfl_xor(mx, off));
  }
  if (lane == 0) {
    atomicMin(&result[0], __float_as_int(mn));
    atomicMax(&result[1], __float_as_int(mx));
  }
}

static inline dim3 px_grid(int w, int h) { return dim3((w + kPxBlockX - 1) / kPxBlockX, (h + kPxBlockY - 1) / kPxBlockY); }

int launch_bilateral_filter(hipStream_t stream, float sigma_xy, float sigma_value, float radius_factor, uint16_t max_depth,
                            float raw_to_float_depth, const uint16_t* in, uint32_t in_pitch, uint16_t* out, uint32_t out_pitch, int w, int h) {
  const int radius = (int)(radius_factor * sigma_xy + 0.5f);
  if (radius < 0 || radius > kBilateralMaxRadius) return 1;
  hipLaunchKernelGGL(bilateral_filter_kernel, px_grid(w, h), dim3(kPxBlockX, kPxBlockY), 0, stream, 2.0f * sigma_xy * sigma_xy,
                     2.0f * sigma_value * sigma_value, radius, radius * radius, max_depth, raw_to_float_depth, in, in_pitch, out, out_pitch, w, h);
  return 0;
}
void launch_brightness(hipStream_t stream, const uint8_t* rgb, uint32_t rgb_pitch, uint8_t* rgba, uint32_t rgba_pitch, int w, int h) {
  hipLaunchKernelGGL(brightness_kernel, px_grid(w, h), dim3(kPxBlockX, kPxBlockY), 0, stream, rgb, rgb_pitch, rgba, rgba_pitch, w, h);
}
void launch_normals_from_depth(hipStream_t stream, const Intrinsics& in, const uint16_t* in_depth, uint32_t in_pitch,
                               uint16_t* out_depth, uint32_t out_pitch, uint16_t* out_normals, uint32_t normals_pitch) {
  hipLaunchKernelGGL(normals_from_depth_kernel, px_grid(in.width, in.height), dim3(kPxBlockX, kPxBlockY), 0, stream, in,
                     in_depth, in_pitch, out_depth, out_pitch, out_normals, normals_pitch);
}
void launch_point_radii(hipStream_t stream, const Intrinsics& in, float raw_to_float_depth, const uint16_t* depth,
                        uint32_t depth_pitch, uint16_t* radius, uint32_t radius_pitch, uint16_t* out_depth, uint32_t out_pitch) {
  hipLaunchKernelGGL(point_radii_kernel, px_grid(in.width, in.height), dim3(kPxBlockX, kPxBlockY), 0, stream, in,
                     raw_to_float_depth, depth, depth_pitch, radius, radius_pitch, out_depth, out_pitch);
}
void launch_min_max_depth(hipStream_t stream, const uint16_t* depth, uint32_t depth_pitch, int w, int h,
                          float raw_to_float_depth, int* result) {
  hipLaunchKernelGGL(min_max_depth_kernel, dim3(kMinMaxBlocks), dim3(256), 0, stream, depth, depth_pitch, w, h,
                     raw_to_float_depth, result);
}

// ---- BA planes (ba_device.h): tiled, packed copies of a frame's images for the surfel sweeps ------------------------
// One thread per output word, in the planes' memory order (ba_device.h: column strips of 8 pixels, rows of a strip contiguous): a
// wavefront writes two whole 128-byte tiles.
__global__ void __launch_bounds__(256)
pack_geom_kernel(const uint16_t* __restrict__ depth, uint32_t depth_pitch, const uint16_t* __restrict__ normals,
                 uint32_t normals_pitch, int width, int height, uint32_t strip_words, uint32_t words,
                 uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= words) return;
  const uint32_t strip = t / strip_words, in_strip = t - strip * strip_words;
  const int x = (int)(strip * kPlaneTileW + (in_strip & 7u));
  const int y = (int)(in_strip >> 3);
  uint32_t word = kInvalidDepthBit;   // padding pixels are never addressed; keep them "invalid depth"
  if (x < width && y < height)
    word = (uint32_t)pitched_load(depth, depth_pitch, y, x) | ((uint32_t)pitched_load(normals, normals_pitch, y, x) << 16);
  out[t] = word;
}

__global__ void __launch_bounds__(256)
pack_luma_footprint_kernel(const uint8_t* __restrict__ rgba, uint32_t rgba_pitch, int width, int height, uint32_t strip_words,
                           uint32_t words, uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= words) return;
  const uint32_t strip = t / strip_words, in_strip = t - strip * strip_words;
  const int ix = (int)(strip * kPlaneTileW + (in_strip & 7u)) - 1;     // top-left texel of the footprint
  const int iy = (int)(in_strip >> 3) - 1;
  uint32_t word = 0;
  if (ix <= width && iy <= height) {
    const int x0 = max(0, min(ix, width - 1)), x1 = max(0, min(ix + 1, width - 1));
    const int y0 = max(0, min(iy, height - 1)), y1 = max(0, min(iy + 1, height - 1));
    const uint8_t* r0 = rgba + (size_t)y0 * rgba_pitch;
    const uint8_t* r1 = rgba + (size_t)y1 * rgba_pitch;
    word = (uint32_t)r0[4 * x0 + 3] | ((uint32_t)r0[4 * x1 + 3] << 8) | ((uint32_t)r1[4 * x0 + 3] << 16) | ((uint32_t)r1[4 * x1 + 3] << 24);
  }
  out[t] = word;
}

void launch_pack_planes(hipStream_t stream, const KfEntry& frame, int width, int height, int cwidth, int cheight, uint32_t* geom,
                        uint32_t* lumafp) {
  const uint32_t gstrip = plane_tiles_y(height) * kPlaneTileH * 8u, gwords = plane_tiles_x(width) * gstrip;   // words per strip, per plane
  hipLaunchKernelGGL(pack_geom_kernel, dim3((gwords + 255) / 256), dim3(256), 0, stream, frame.depth, frame.depth_pitch, frame.normals,
                     frame.normals_pitch, width, height, gstrip, gwords, geom);
  // a frame handed over without a colour image (the supporting-surfel entry point of B/kernels.h:94-119 gets depth and
  // normals only) has no luma plane to pack; the calls that take such a frame never sample colour
  if (frame.color == nullptr) return;
  const uint32_t fstrip = plane_tiles_y(cheight + 2) * kPlaneTileH * 8u, fwords = plane_tiles_x(cwidth + 2) * fstrip;
  hipLaunchKernelGGL(pack_luma_footprint_kernel, dim3((fwords + 255) / 256), dim3(256), 0, stream, frame.color, frame.color_pitch,
                     cwidth, cheight, fstrip, fwords, lumafp);
}

// ---- counter calibration (tooling) -------------------------------------------------------------------------------------
// Reads of KNOWN size with the instruction width the sweeps gather with (global_load_dword, 4 bytes per lane), so that the
// FETCH_SIZE counter of rocprofv3 can be calibrated on this access width (MI355X_MICROARCH.md calibrates its x2 correction
// for wide coalesced reads only).  pattern 0: every dword of the buffer once, consecutive lanes on consecutive dwords;
// pattern 1: one dword per 128-byte line (a gather that uses 4 of every 128 bytes).
__global__ void read_pattern_kernel(const uint32_t* __restrict__ data, size_t words, int pattern, uint32_t* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  if (pattern == 0) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) acc += data[i];
  } else {
    const size_t lines = words / 32;
    for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < lines; l += stride) acc += data[l * 32 + (l & 31)];
  }
  if (acc == 0x12345678u) *sink = acc;   // never true for a zero-filled buffer: keeps the loads alive
}
void launch_read_pattern(hipStream_t stream, const uint32_t* data, size_t words, int pattern, uint32_t* sink) {
  hipLaunchKernelGGL(read_pattern_kernel, dim3(8192), dim3(256), 0, stream, data, words, pattern, sink);
}

}  // namespace bahip
