// ConvGRU decoder forward, WEIGHT-STATIONARY form (round 5): an EXPERIMENT, off by default (DF_GRU_WS=1; bf16x2 mode, mfma_bf16 == 3) --
// see STATUS at the MFMA helpers below: exact, and with the drains it needs no faster than the streaming kernel.
// [REF decoder.py:123-183]; same arithmetic as gru_fwd4_kernel<., 3> (csrc/decoder4.hip) -- the x contribution from the [416][4]
// table, bf16x2 products (hi + lo planes of both operands, three MFMAs), fp32 state and gates -- in another decomposition:
//
//   gru_fwd4: a wave owns 16 POINTS x all 128 hidden columns; the 3 x 128 x 128 weights stream L2 -> LDS in 32-deep chunks shared by
//             the workgroup's four waves: 12 workgroup barriers + 12 DMA waits per GRU iteration.  profiles/r05_pmc_gru4.txt: the
//             matrix pipe is 27 % busy, 44 % of the wave cycles sit in s_waitcnt / s_barrier.
//   here:     a wave owns 16 hidden COLUMNS x 64 points; its slice of the three gate matrices (3 x 16 rows x 128 k, hi + lo planes)
//             lives in 96 REGISTERS for the whole kernel (persistent workgroups: one per CU, eight waves, tiles of 64 points in a
//             grid-stride loop); h and r * h go through LDS as pre-split bf16 planes that all eight waves read as the MFMA B
//             operand.  No weight traffic after the prologue, TWO barriers per iteration (r * h complete, h' complete).
//
// MFMA shape: D[o][pt] += W[o][k] h[pt][k] with v_mfma_f32_16x16x32_bf16, A = weights (lane (o = lane & 15, kg = lane >> 4) holds
// k = 32 s + 8 kg ..+7), B = activations (lane (pt, kg): the same 8 k of point pt -- one ds_read_b128 per plane), D: lane (pt = lane
// & 15, og = lane >> 4) holds o = 4 og .. 4 og + 3 -- four CONSECUTIVE k of the next product, so h' and r * h go back to LDS as
// one 8-byte store per plane.  LDS rows are 256 B (128 bf16); the 16-byte slot of a row is XOR-swizzled with the point index,
// which is conflict-free for ds_read_b128's lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md): the sets S = {0-3, 12-15}
// and T = {4-11} of points a group combines are closed under ^1, ^2, ^3, so (slot ^ pt) covers 16 distinct bank quads.
#include <cstdlib>

#include "common.h"
#include "gemm_dma.h"

namespace {

using namespace gd;

constexpr int XT_ROWS5 = 416;
constexpr int P5 = 64;                         // points per tile
constexpr int ROWB = 256;                      // bytes per LDS activation row (128 bf16)

struct Gru5Params {
  df_img before, after;
  const int32_t* coords;
  const float* offs;
  const int32_t* counts;
  int B, N, T;
  df_gru_weights w;      // w_zr [256][hi 192 | lo 192], w_q [128][..], w_1 [32][..] bf16x2 rows; w_2, b_2 fp32
  const float* xtab;
  float* flow;
  float* hsave;          // [T + 1][B*N][128] fp32 or null
  int64_t iter_stride;
  int tiles_per_sample;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const f32x4 v, s16x4& hi, s16x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = __builtin_bit_cast(short, h);
    lo[i] = __builtin_bit_cast(short, (__bf16)(v[i] - (float)h));
  }
}

template <bool SAVE>
__global__ __launch_bounds__(512, 1) void gru_fwd5_kernel(Gru5Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  // activations as bf16 planes: [plane hi | lo][64 points][256 B], slot-swizzled
  __shared__ __attribute__((aligned(16))) unsigned char Hs[2 * P5 * ROWB];     // h       32 KB
  __shared__ __attribute__((aligned(16))) unsigned char Rs[2 * P5 * ROWB];     // r * h   32 KB
  __shared__ __attribute__((aligned(16))) float Xt[XT_ROWS5 * 4];              // 6.6 KB
  __shared__ __attribute__((aligned(16))) float Os[P5 * 4];                    // offsets of the tile's points
  __shared__ __attribute__((aligned(16))) float Hid[P5 * 36];                  // head: GELU(hidden) [pt][32] (+ pad)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane & 15, og = lane >> 4;          // D layout: point pl (+ 16 tile), rows 4 og ..
  const int o0 = 16 * wave + 4 * og;                 // first of this lane's four hidden columns
  // ---- this wave's weight slices -> registers (A operand: lane (o = pl, kg = og)) ------------------------------------
  bf16x8_t wh[3][4], wl[3][4];
  {
    const int o = 16 * wave + pl;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float* base = g < 2 ? p.w.w_zr + (int64_t)(g * 128 + o) * 192 : p.w.w_q + (int64_t)o * 192;   // row pitch: 192 floats
      const __bf16* rowp = reinterpret_cast<const __bf16*>(base);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wh[g][s] = *reinterpret_cast<const bf16x8_t*>(rowp + 32 * s + 8 * og);
        wl[g][s] = *reinterpret_cast<const bf16x8_t*>(rowp + 192 + 32 * s + 8 * og);
      }
    }
  }
  for (int i = tid; i < XT_ROWS5; i += 512) st4(Xt + 4 * i, ld4(p.xtab + 4 * i));
  const unsigned wr_slot = (unsigned)(2 * wave + (og >> 1)), wr_in = (unsigned)((og & 1) * 8);   // this lane's 8 bytes of a row
  constexpr unsigned swz_mask = 15u;

  const int ntiles = p.B * p.tiles_per_sample;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / p.tiles_per_sample;
    const int p0 = (tile - b * p.tiles_per_sample) * P5;
    const int cnt = p.counts[b];
    if (p0 >= cnt) continue;                          // (workgroup-uniform)
    const int64_t grow0 = (int64_t)b * p.N + p0;
    __syncthreads();                                  // the previous tile's readers of Os / Hs / Hid are done
    if (tid < P5) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p0 + tid < cnt) {
        const float* o = p.offs + (grow0 + tid) * 3;
        v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
      }
      st4(Os + 4 * tid, v);
    }
    // ---- gather h0 = [before | after] straight into the D layout -----------------------------------------------------
    f32x4 h[4];
    {
      const df_img& im = wave < 4 ? p.before : p.after;
      const float* ip = reinterpret_cast<const float*>(im.ptr) + df_img_base(im, b);
      const int c0 = o0 & 63;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pt = 16 * t + pl;
        h[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p0 + pt < cnt) {
          const int32_t* cc = p.coords + (grow0 + pt) * 3;
          h[t] = ld4(ip + ((int64_t)cc[1] * im.w + cc[2]) * im.ld + c0);
        }
      }
    }
    auto put_rows = [&](unsigned char* S, const f32x4 (&v)[4]) {     // D-layout values -> the bf16 planes (hi | lo) of S
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pt = 16 * t + pl;
        s16x4 hi, lo;
        split4(v[t], hi, lo);
        const unsigned off = (unsigned)pt * ROWB + ((wr_slot ^ ((unsigned)pt & swz_mask)) * 16) + wr_in;
        *reinterpret_cast<s16x4*>(S + off) = hi;
        *reinterpret_cast<s16x4*>(S + P5 * ROWB + off) = lo;
      }
    };
    auto save_plane = [&](int it) {                                   // fp32 h -> hsave plane `it` (rows past the count dropped)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pt = 16 * t + pl;
        if (p0 + pt < cnt) st4(p.hsave + it * p.iter_stride + (grow0 + pt) * 128 + o0, h[t]);
      }
    };
    put_rows(Hs, h);
    __syncthreads();
    auto xinit = [&](f32x4 (&acc)[4], int g) {                        // x contribution + bias of gate g at (o0 .. o0 + 3, points)
      f32x4 of[4];                                                    // (re-read per use: 16 registers the weight slices need)
#pragma unroll
      for (int t = 0; t < 4; ++t) of[t] = ld4(Os + 4 * (16 * t + pl));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 tb = ld4(Xt + (g * 128 + o0 + i) * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][i] = fmaf(tb[0], of[t][0], fmaf(tb[1], of[t][1], fmaf(tb[2], of[t][2], tb[3])));
      }
    };
    // B fragment of point tile t, k step s from the planes of S: lane (pt = pl, kg = og)
    auto bfrag = [&](const unsigned char* S, int t, int s, bf16x8_t& bh, bf16x8_t& bl) {
      const int pt = 16 * t + pl;
      const unsigned off = (unsigned)pt * ROWB + (((unsigned)(4 * s + og) ^ ((unsigned)pt & swz_mask)) * 16);
      bh = *reinterpret_cast<const bf16x8_t*>(S + off);
      bl = *reinterpret_cast<const bf16x8_t*>(S + P5 * ROWB + off);
    };
    // STATUS (round 5): an experiment, OFF by default (DF_GRU_WS=1 selects it; tests/test_gpu_kernels.py runs it against the goldens).
    // Without the drains below it measured 2.09 ms against the streaming kernel's 2.32 at the bench shape -- and returned a wrong
    // 16 x 16 tile in one wave about once per launch (tools/gru_ws_check2.py: every other tile exact to 1e-7; the wrong one moves with
    // the register allocation; extra barriers, sleeps and the LDS swizzle change nothing; a drain of the matrix pipe after each tile's
    // products removes it).  What the evidence points at: an MFMA this wave has ISSUED has not necessarily READ its A / B registers
    // when the next instructions overwrite them -- here the next tile's ds_read_b128 into the fragment registers the allocator just
    // freed -- once a second wave of the SIMD keeps the pipe busy.  The same happened with the builtin form (where the allocator also
    // produced MFMAs whose destination overlaps a source, whole or in part: 20-26 of 384) and with accumulators tied in inline
    // assembly (this form).  With the drains the kernel is exact and exactly as fast as the streaming kernel; a version that
    // rotates three explicit fragment register sets (reuse distance six MFMAs) would keep the gain without them -- not built.
    auto mfma = [&](f32x4& acc, const bf16x8_t a, const bf16x8_t b) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    };
    auto mfma_drain = [&]() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); };
    // two accumulators, three products each, interleaved (a dependent product follows its predecessor by one other MFMA)
    auto mm3x2 = [&](f32x4& acc0, const bf16x8_t a0h, const bf16x8_t a0l, f32x4& acc1, const bf16x8_t a1h, const bf16x8_t a1l,
                     const bf16x8_t bh, const bf16x8_t bl) {
      mfma(acc0, a0l, bh);   // small terms first
      mfma(acc1, a1l, bh);
      mfma(acc0, a0h, bl);
      mfma(acc1, a1h, bl);
      mfma(acc0, a0h, bh);
      mfma(acc1, a1h, bh);
    };

    for (int it = 0; it < p.T; ++it) {
      if (SAVE) save_plane(it);
      f32x4 z[4], r[4];
      xinit(z, 0);
      xinit(r, 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bf16x8_t bh, bl;
          bfrag(Hs, t, s, bh, bl);
          mm3x2(z[t], wh[0][s], wl[0][s], r[t], wh[1][s], wl[1][s], bh, bl);
          mfma_drain();     // (see the note at mfma(): without it a tile of a wave came out wrong about once per launch)
        }
      mfma_drain();
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          z[t][i] = df_sigmoid_fast(z[t][i]);
          r[t][i] = df_sigmoid_fast(r[t][i]) * h[t][i];     // r * h
        }
      put_rows(Rs, r);
      __syncthreads();                                        // r * h complete (all 128 columns)
      xinit(r, 2);                                            // r <- q accumulators
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; t += 2) {    // two point tiles per pass: two independent accumulators
          bf16x8_t bh0, bl0, bh1, bl1;
          bfrag(Rs, t, s, bh0, bl0);
          bfrag(Rs, t + 1, s, bh1, bl1);
          mfma(r[t], wl[2][s], bh0);
          mfma(r[t + 1], wl[2][s], bh1);
          mfma(r[t], wh[2][s], bl0);
          mfma(r[t + 1], wh[2][s], bl1);
          mfma(r[t], wh[2][s], bh0);
          mfma(r[t + 1], wh[2][s], bh1);
          mfma_drain();     // (see the note at mfma(): without it a tile of a wave came out wrong about once per launch)
        }
      mfma_drain();
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float q = df_tanh_fast(r[t][i]);
          h[t][i] = (1.f - z[t][i]) * h[t][i] + z[t][i] * q;
        }
      // every wave has passed the barrier above, i.e. finished its z / r products on Hs: overwrite it with h'
      put_rows(Hs, h);
      __syncthreads();                                        // h' complete
    }
    if (SAVE) save_plane(p.T);
    // ---- MLP head: hid = W1[:, :128] h_T + table rows 384 ..; waves 0, 1 own 16 of the 32 hidden units each ---------------
    if (wave < 2) {
      bf16x8_t a1h[4], a1l[4];
      {
        const __bf16* rowp = reinterpret_cast<const __bf16*>(p.w.w_1 + (int64_t)(16 * wave + pl) * 192);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          a1h[s] = *reinterpret_cast<const bf16x8_t*>(rowp + 32 * s + 8 * og);
          a1l[s] = *reinterpret_cast<const bf16x8_t*>(rowp + 192 + 32 * s + 8 * og);
        }
      }
      f32x4 hid[4], of[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) of[t] = ld4(Os + 4 * (16 * t + pl));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 tb = ld4(Xt + (384 + o0 + i) * 4);      // (o0 < 32 for waves 0, 1)
#pragma unroll
        for (int t = 0; t < 4; ++t) hid[t][i] = fmaf(tb[0], of[t][0], fmaf(tb[1], of[t][1], fmaf(tb[2], of[t][2], tb[3])));
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          bf16x8_t bh0, bl0, bh1, bl1;
          bfrag(Hs, t, s, bh0, bl0);
          bfrag(Hs, t + 1, s, bh1, bl1);
          mfma(hid[t], a1l[s], bh0);
          mfma(hid[t + 1], a1l[s], bh1);
          mfma(hid[t], a1h[s], bl0);
          mfma(hid[t + 1], a1h[s], bl1);
          mfma(hid[t], a1h[s], bh0);
          mfma(hid[t + 1], a1h[s], bh1);
        }
      mfma_drain();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 g4;
#pragma unroll
        for (int i = 0; i < 4; ++i) g4[i] = df_gelu(hid[t][i]);
        st4(Hid + (16 * t + pl) * 36 + o0, g4);
      }
    }
    __syncthreads();
    if (tid < P5 * 3) {
      const int pt = tid / 3, o = tid - pt * 3;
      if (p0 + pt < cnt) {
        float a = p.w.b_2[o];
        for (int c = 0; c < 32; ++c) a = fmaf(p.w.w_2[o * 32 + c], Hid[pt * 36 + c], a);
        p.flow[(grow0 + pt) * 3 + o] = a;
      }
    }
  }
#endif
}

}  // namespace

// Arguments validated by df_gru_lean_fwd (decoder4.hip), which dispatches here for mfma_bf16 == 3 unless DF_GRU_WS=0.
int df_launch_gru_fwd5(df_img before, df_img after, const int32_t* coords, const float* offs, const int32_t* counts, int B, int N,
                       int num_iters, df_gru_weights wts, const float* xtab, float* flow, float* hsave, void* stream) {
  Gru5Params p;
  p.before = before; p.after = after; p.coords = coords; p.offs = offs; p.counts = counts;
  p.B = B; p.N = N; p.T = num_iters; p.w = wts; p.xtab = xtab; p.flow = flow; p.hsave = hsave;
  p.iter_stride = (int64_t)B * N * 128;
  p.tiles_per_sample = (N + P5 - 1) / P5;
  const int64_t ntiles = (int64_t)B * p.tiles_per_sample;
  const int grid = (int)(ntiles < 256 ? ntiles : 256);     // persistent: one eight-wave workgroup per CU
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hsave) hipLaunchKernelGGL(gru_fwd5_kernel<true>, dim3(grid), dim3(512), 0, s, p);
  else hipLaunchKernelGGL(gru_fwd5_kernel<false>, dim3(grid), dim3(512), 0, s, p);
  DF_CHECK_LAUNCH();
  return DF_OK;
}
