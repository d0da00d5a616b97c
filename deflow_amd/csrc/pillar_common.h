// Device helpers shared by the pillarisation kernels (pillarize.hip, pillar_bands.hip): the bit-exact voxel coordinate
// (mmcv dynamic_voxelize semantics) and the per-point pieces of DynamicPillarFeatureNet.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ bool voxelize(const df_pillar_geom& g, float px, float py, float pz, int& cx, int& cy) {
  if (isnan(px) || isnan(py) || isnan(pz)) return false;
  const float fx = floorf(__fdiv_rn(__fsub_rn(px, g.minx), g.vx));
  const float fy = floorf(__fdiv_rn(__fsub_rn(py, g.miny), g.vy));
  const float fz = floorf(__fdiv_rn(__fsub_rn(pz, g.minz), g.vz));
  if (!(fx >= 0.f && fx < (float)g.gx)) return false;
  if (!(fy >= 0.f && fy < (float)g.gy)) return false;
  if (!(fz >= 0.f && fz < (float)g.gz)) return false;
  cx = (int)fx;
  cy = (int)fy;
  return true;
}

// ---- pillar feature net ------------------------------------------------------------------
// 8 lanes per cell; lane `sub` owns output channels 4*sub .. 4*sub+3 of Linear(9->32).
struct PfnCtx {
  float w[4][9];
  float sc[4], sh[4], mu[4], is[4];
};

__device__ __forceinline__ void pfn_load_w(PfnCtx& c, const float* __restrict__ w_pfn, int sub) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 9; ++j) c.w[k][j] = w_pfn[(4 * sub + k) * 9 + j];
}
__device__ __forceinline__ void pfn_load_bn(PfnCtx& c, const float* __restrict__ ss /*[4][32]*/, int sub) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c.sc[k] = ss[0 * 32 + 4 * sub + k];
    c.sh[k] = ss[1 * 32 + 4 * sub + k];
    c.mu[k] = ss[2 * 32 + 4 * sub + k];
    c.is[k] = ss[3 * 32 + 4 * sub + k];
  }
}
// pillar centre as DynamicPillarFeatureNet computes it: c * v + (v/2 + min), two roundings
__device__ __forceinline__ void pfn_centre(const df_pillar_geom& g, int cell, float& ctx, float& cty, float& ctz) {
  const int cy = cell / g.gx, cx = cell - cy * g.gx;
  ctx = __fadd_rn(__fmul_rn((float)cx, g.vx), g.offx);
  cty = __fadd_rn(__fmul_rn((float)cy, g.vy), g.offy);
  ctz = __fadd_rn(0.f, g.offz);
}
__device__ __forceinline__ void pfn_mean(const float* __restrict__ pts, int s, int e, float& mx, float& my, float& mz) {
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int i = s; i < e; ++i) {
    const float* p = pts + (int64_t)i * 3;
    sx += p[0];
    sy += p[1];
    sz += p[2];
  }
  const float inv = (float)(e - s);
  mx = sx / inv;
  my = sy / inv;
  mz = sz / inv;
}
__device__ __forceinline__ void pfn_feat(const float* p, float mx, float my, float mz, float ctx, float cty, float ctz,
                                         float (&f)[9]) {
  f[0] = p[0]; f[1] = p[1]; f[2] = p[2];
  f[3] = p[0] - mx; f[4] = p[1] - my; f[5] = p[2] - mz;
  f[6] = p[0] - ctx; f[7] = p[1] - cty; f[8] = p[2] - ctz;
}
__device__ __forceinline__ void pfn_linear(const PfnCtx& c, const float (&f)[9], float (&u)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) a = fmaf(c.w[k][j], f[j], a);
    u[k] = a;
  }
}


// mode 1 (DynamicScatter 'max'): the pillar's gradient goes, per channel, to the FIRST point (lowest input index; the
// stable sort keeps input order inside a pillar) whose post-ReLU feature equals the pillar maximum -- the traceback rule
// of mmcv's dynamic_point_to_voxel_backward.  Returns that sorted position per channel.
__device__ __forceinline__ void pfn_argmax(const PfnCtx& c, const float* __restrict__ pts, int s, int e, float mx, float my,
                                           float mz, float ctx, float cty, float ctz, int (&am)[4]) {
  float best[4];
  for (int i = s; i < e; ++i) {
    float f[9], u[4];
    pfn_feat(pts + (int64_t)i * 3, mx, my, mz, ctx, cty, ctz, f);
    pfn_linear(c, f, u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = fmaxf(fmaf(u[k], c.sc[k], c.sh[k]), 0.f);
      if (i == s || v > best[k]) { best[k] = v; am[k] = i; }
    }
  }
}

}  // namespace
