// ref_entry.cc -- C entry points onto the REFERENCE's own device-math functions, compiled for the host.
// TEST INFRASTRUCTURE (see oracle/oracle.h).  The functions called here are the reference's, included from where they lie under
// /root/reference (oracle/Makefile passes the include paths); this file only marshals arguments:
//   B/util.cuh                          RawToCalibratedDepth, ProjectSurfelToImage, ImageSpaceNormalToU16, U16ToImageSpaceNormal
//   B/util_nvcc_only.cuh                SurfelGetPosition, SurfelGetNormal, SmallFloatToTenBitSigned, TenBitSignedToSmallFloat
//   B/robust_weighting.cuh              TukeyWeight, TukeyResidual, HuberWeight, HuberResidual
//   B/cost_function.cuh                 ComputeRawDepthResidual, ComputeDepthResidual(Inv)StddevEstimate, ComputeTangentProjections,
//                                       ComputeRawDescriptorResidual, DescriptorJacobianWrtProjectedPosition, the weights
//   B/surfel_projection.cuh             the projector PODs, TransformDepthToColorPixelCorner
//   B/surfel_projection_nvcc_only.cuh   SurfelProjectsToAssociatedPixel (-> IsAssociatedWithPixel)
// (B/ = applications/badslam/src/badslam/).  The reference builds these with nvcc -use_fast_math; here they are evaluated in
// IEEE binary32 by g++ without contraction, i.e. as written.
#include <cstring>

#include "badslam/cost_function.cuh"
#include "badslam/kernels.cuh"
#include "badslam/robust_weighting.cuh"
#include "badslam/surfel_projection.cuh"
#include "badslam/surfel_projection_nvcc_only.cuh"
#include "badslam/util.cuh"
#include "badslam/util_nvcc_only.cuh"

using namespace vis;

// CUDA's float -> int conversion SATURATES (cvt.rzi.s32.f32); the host's is undefined beyond the int range and yields INT_MIN
// on x86.  ProjectSurfelToImage (B/util.cuh:83-118) relies on the former: a surfel a hair in front of the camera plane projects
// to pixel +1e10, px saturates to INT_MAX and fails `px >= width`; compiled for the host the same surfel gets px = INT_MIN,
// passes every test and reads depth_buffer(py, INT_MIN) (found as a crash on a 200-keyframe scene).  This guard gives the
// reference's function the outcome it has on its own platform: such a pixel is outside the image.
static inline bool pixel_outside_int_range(const SurfelProjectionParameters& proj, unsigned int surfel_index) {
  if (surfel_index >= proj.surfels_size) return false;
  const float3 global_position = SurfelGetPosition(proj.surfels, surfel_index);
  float3 local_position;
  if (!proj.frame_T_global.MultiplyIfResultZIsPositive(global_position, &local_position)) return false;   // rejected there anyway
  const float2 p = proj.projector.Project(local_position);
  return !(p.x < 2147483648.f && p.y < 2147483648.f);   // (also true for NaN)
}
extern "C" {

// mirrors orc_pair_eval (oracle/oracle.h) field for field; pose / surfel Jacobians are not in the reference's headers (they
// live in its .cu files) and stay zero
struct ref_pair_eval {
  int32_t associated, px, py, color_valid;
  float calibrated_depth;
  float depth_residual, depth_weight, depth_inv_stddev;
  float depth_jac_pose[6];
  float depth_jac_surfel;
  float desc_residual[2], desc_weight[2];
  float desc_jac_pose[2][6];
  float desc_jac_surfel[2];
  float grad[4];
};

struct ref_scene {
  // cameras: fx, fy, cx, cy in the pixel-corner convention
  float depth_cam[4], color_cam[4];
  int width, height, color_width, color_height;
  // depth parameters
  float a, raw_to_float_depth, baseline_fx;
  int cell;
  float* cfactor; int cf_width, cf_height;
  // keyframe images (dense, row-major) and pose
  uint16_t* depth; uint16_t* normals; uint8_t* rgba;
  float frame_T_global[12];
  // surfels: 17 rows of `capacity` floats
  float* surfel_rows; uint32_t capacity, surfels_size;
  int quantize_texture_weights;
};

float ref_raw_to_calibrated_depth(float a, float cfactor, float raw_to_float_depth, uint16_t raw) {
  return RawToCalibratedDepth(a, cfactor, raw_to_float_depth, raw);
}
float ref_tukey_weight(float r, float k) { return TukeyWeight(r, k); }
float ref_tukey_residual(float r, float k) { return TukeyResidual(r, k); }
float ref_huber_weight(float r, float k) { return HuberWeight(r, k); }
float ref_huber_residual(float r, float k) { return HuberResidual(r, k); }
uint16_t ref_image_space_normal_to_u16(float x, float y) { return ImageSpaceNormalToU16(x, y); }
void ref_u16_to_image_space_normal(uint16_t v, float out[3]) {
  const float3 n = U16ToImageSpaceNormal(v);
  out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
uint32_t ref_pack_surfel_normal(float x, float y, float z) {
  float row[32] = {0};
  CUDABuffer_<float> one(row, kSurfelAttributeCount, 1, sizeof(float));
  SurfelSetNormal(&one, 0, make_float3(x, y, z));
  uint32_t bits;
  memcpy(&bits, &row[kSurfelNormal], sizeof(bits));
  return bits;
}
void ref_unpack_surfel_normal(uint32_t bits, float out[3]) {
  float row[32] = {0};
  memcpy(&row[kSurfelNormal], &bits, sizeof(bits));
  CUDABuffer_<float> one(row, kSurfelAttributeCount, 1, sizeof(float));
  const float3 n = SurfelGetNormal(one, 0);
  out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
float ref_sample_luma(const uint8_t* rgba, int width, int height, float x, float y, int quantize) {
  RefTexture t = {reinterpret_cast<const uchar4*>(rgba), width, height, (size_t)width * 4, quantize};
  return tex2D<float4>(reinterpret_cast<cudaTextureObject_t>(&t), x, y).w;
}

// Association + residuals + weights + gradients of `count` (surfel, keyframe) pairs, by the reference's functions in the
// order its kernels call them (B/kernel_opt_pose.cu:251-353: SurfelProjectsToAssociatedPixel, depth residual,
// TransformDepthToColorPixelCorner, tangent projections, descriptor residual, descriptor gradient).
void ref_evaluate_pairs(const ref_scene* sc, const uint32_t* surfel_indices, int count, ref_pair_eval* out) {
  CUDABuffer_<float> surfels(sc->surfel_rows, kSurfelAttributeCount, (int)sc->capacity, (size_t)sc->capacity * sizeof(float));
  CUDABuffer_<u16> depth_buffer(sc->depth, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  CUDABuffer_<u16> normals_buffer(sc->normals, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  DepthParameters dp;
  dp.cfactor_buffer = CUDABuffer_<float>(sc->cfactor, sc->cf_height, sc->cf_width, (size_t)sc->cf_width * sizeof(float));
  dp.a = sc->a; dp.raw_to_float_depth = sc->raw_to_float_depth; dp.baseline_fx = sc->baseline_fx; dp.sparse_surfel_cell_size = sc->cell;
  const PixelCornerProjector depth_projector(sc->depth_cam[0], sc->depth_cam[1], sc->depth_cam[2], sc->depth_cam[3]);
  const PixelCenterUnprojector unprojector(depth_projector);                       // B/surfel_projection.h:54-71
  const PixelCornerProjector color_projector(sc->color_cam[0], sc->color_cam[1], sc->color_cam[2], sc->color_cam[3]);
  CUDAMatrix3x4 F;
  F.row0 = make_float4(sc->frame_T_global[0], sc->frame_T_global[1], sc->frame_T_global[2], sc->frame_T_global[3]);
  F.row1 = make_float4(sc->frame_T_global[4], sc->frame_T_global[5], sc->frame_T_global[6], sc->frame_T_global[7]);
  F.row2 = make_float4(sc->frame_T_global[8], sc->frame_T_global[9], sc->frame_T_global[10], sc->frame_T_global[11]);
  const SurfelProjectionParameters proj(surfels, depth_buffer, normals_buffer, dp, depth_projector, unprojector, F, sc->surfels_size);
  // depth -> colour pixel (B/surfel_projection.h:100-124)
  DepthToColorPixelCorner d2c;
  d2c.fx = sc->color_cam[0] / sc->depth_cam[0];
  d2c.fy = sc->color_cam[1] / sc->depth_cam[1];
  d2c.cx = -1 * sc->color_cam[0] * sc->depth_cam[2] / sc->depth_cam[0] + sc->color_cam[2];
  d2c.cy = -1 * sc->color_cam[1] * sc->depth_cam[3] / sc->depth_cam[1] + sc->color_cam[3];
  d2c.width = sc->color_width; d2c.height = sc->color_height;
  RefTexture tex = {reinterpret_cast<const uchar4*>(sc->rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
  const cudaTextureObject_t color_texture = reinterpret_cast<cudaTextureObject_t>(&tex);

  for (int t = 0; t < count; ++t) {
    ref_pair_eval& o = out[t];
    memset(&o, 0, sizeof(o));
    SurfelProjectionResult6 r;
    if (pixel_outside_int_range(proj, surfel_indices[t]) || !SurfelProjectsToAssociatedPixel(surfel_indices[t], proj, &r)) continue;
    o.associated = 1; o.px = r.px; o.py = r.py; o.calibrated_depth = r.pixel_calibrated_depth;
    const float3 local_normal = F.Rotate(r.surfel_normal);
    const float inv_std = ComputeDepthResidualInvStddevEstimate(unprojector.nx(r.px), unprojector.ny(r.py), r.pixel_calibrated_depth, local_normal, dp.baseline_fx);
    float3 local_unproj;
    float raw;
    ComputeRawDepthResidual(unprojector, r.px, r.py, r.pixel_calibrated_depth, inv_std, r.surfel_local_position, local_normal, &local_unproj, &raw);
    o.depth_inv_stddev = inv_std;
    o.depth_residual = raw;
    o.depth_weight = ComputeDepthResidualWeight(raw);
    float2 color_pxy;
    if (TransformDepthToColorPixelCorner(r.pxy, d2c, &color_pxy)) {
      o.color_valid = 1;
      float2 t1, t2;
      ComputeTangentProjections(r.surfel_global_position, r.surfel_normal, SurfelGetRadiusSquared(surfels, surfel_indices[t]), F, color_projector, &t1, &t2);
      ComputeRawDescriptorResidual(color_texture, color_pxy, t1, t2, surfels(kSurfelDescriptor1, surfel_indices[t]),
                                   surfels(kSurfelDescriptor2, surfel_indices[t]), &o.desc_residual[0], &o.desc_residual[1]);
      o.desc_weight[0] = ComputeDescriptorResidualWeight(o.desc_residual[0]);
      o.desc_weight[1] = ComputeDescriptorResidualWeight(o.desc_residual[1]);
      DescriptorJacobianWrtProjectedPosition(color_texture, color_pxy, t1, t2, &o.grad[0], &o.grad[1], &o.grad[2], &o.grad[3]);
    }
  }
}

// One full cost evaluation of the surfels against ONE keyframe by the reference's functions: the sum of the robust depth cost
// and of the two robust descriptor costs of every associated pair (the terms the reference's kernels sum,
// B/kernel_opt_pose.cu:311-318,372-379), OpenMP over the surfels.  This is the reference's CPU-runnable cost path as far as
// one exists: bench.py times it on the GPU box's host cores as cpu_baseline (kind "reference").
double ref_evaluate_cost(const ref_scene* sc, unsigned long long* num_residuals, int use_depth, int use_desc) {
  CUDABuffer_<float> surfels(sc->surfel_rows, kSurfelAttributeCount, (int)sc->capacity, (size_t)sc->capacity * sizeof(float));
  CUDABuffer_<u16> depth_buffer(sc->depth, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  CUDABuffer_<u16> normals_buffer(sc->normals, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  DepthParameters dp;
  dp.cfactor_buffer = CUDABuffer_<float>(sc->cfactor, sc->cf_height, sc->cf_width, (size_t)sc->cf_width * sizeof(float));
  dp.a = sc->a; dp.raw_to_float_depth = sc->raw_to_float_depth; dp.baseline_fx = sc->baseline_fx; dp.sparse_surfel_cell_size = sc->cell;
  const PixelCornerProjector depth_projector(sc->depth_cam[0], sc->depth_cam[1], sc->depth_cam[2], sc->depth_cam[3]);
  const PixelCenterUnprojector unprojector(depth_projector);
  const PixelCornerProjector color_projector(sc->color_cam[0], sc->color_cam[1], sc->color_cam[2], sc->color_cam[3]);
  CUDAMatrix3x4 F;
  F.row0 = make_float4(sc->frame_T_global[0], sc->frame_T_global[1], sc->frame_T_global[2], sc->frame_T_global[3]);
  F.row1 = make_float4(sc->frame_T_global[4], sc->frame_T_global[5], sc->frame_T_global[6], sc->frame_T_global[7]);
  F.row2 = make_float4(sc->frame_T_global[8], sc->frame_T_global[9], sc->frame_T_global[10], sc->frame_T_global[11]);
  const SurfelProjectionParameters proj(surfels, depth_buffer, normals_buffer, dp, depth_projector, unprojector, F, sc->surfels_size);
  DepthToColorPixelCorner d2c;
  d2c.fx = sc->color_cam[0] / sc->depth_cam[0];
  d2c.fy = sc->color_cam[1] / sc->depth_cam[1];
  d2c.cx = -1 * sc->color_cam[0] * sc->depth_cam[2] / sc->depth_cam[0] + sc->color_cam[2];
  d2c.cy = -1 * sc->color_cam[1] * sc->depth_cam[3] / sc->depth_cam[1] + sc->color_cam[3];
  d2c.width = sc->color_width; d2c.height = sc->color_height;
  RefTexture tex = {reinterpret_cast<const uchar4*>(sc->rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
  const cudaTextureObject_t color_texture = reinterpret_cast<cudaTextureObject_t>(&tex);
  double total = 0;
  unsigned long long count = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : total, count)
  for (long long i = 0; i < (long long)sc->surfels_size; ++i) {
    SurfelProjectionResult6 r;
    if (pixel_outside_int_range(proj, (unsigned int)i) || !SurfelProjectsToAssociatedPixel((unsigned int)i, proj, &r)) continue;
    if (use_depth) {
      const float3 local_normal = F.Rotate(r.surfel_normal);
      const float inv_std = ComputeDepthResidualInvStddevEstimate(unprojector.nx(r.px), unprojector.ny(r.py), r.pixel_calibrated_depth, local_normal, dp.baseline_fx);
      float3 local_unproj;
      float raw;
      ComputeRawDepthResidual(unprojector, r.px, r.py, r.pixel_calibrated_depth, inv_std, r.surfel_local_position, local_normal, &local_unproj, &raw);
      total += ComputeWeightedDepthResidual(raw);
      count += 1;
    }
    float2 color_pxy;
    if (use_desc && TransformDepthToColorPixelCorner(r.pxy, d2c, &color_pxy)) {
      float2 t1, t2;
      float raw1, raw2;
      ComputeTangentProjections(r.surfel_global_position, r.surfel_normal, SurfelGetRadiusSquared(surfels, (u32)i), F, color_projector, &t1, &t2);
      ComputeRawDescriptorResidual(color_texture, color_pxy, t1, t2, surfels(kSurfelDescriptor1, (u32)i), surfels(kSurfelDescriptor2, (u32)i), &raw1, &raw2);
      total += ComputeWeightedDescriptorResidual(raw1) + ComputeWeightedDescriptorResidual(raw2);
      count += 2;
    }
  }
  if (num_residuals) *num_residuals = count;
  return total;
}

}  // extern "C"
