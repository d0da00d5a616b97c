// The PERSISTENT pre-split 3x3 stride-1 convolution (forward and data gradient) of the BEV UNet ([REF decoder.py:202-220]'s Conv2d at the
// layers between which activations / gradients are stored as fp16x2 planes): split from conv.hip in round 6.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace {

// ---- PERSISTENT form of conv_halo_x3_kernel<.., NP = 2, XP> (round 4; DF_CONV_PERS=0 restores the one-tile-per-workgroup form) -------
// One workgroup per CU walks its tiles (virtual block ids blockIdx.x, + gridDim.x, ...: the same tile -> XCD placement and the same
// side-by-side rows as the plain launch).  Per tile the plain form pays its prologue (halo + first weight stages: ~2-3 us of DMA latency
// with an empty matrix pipe) and its epilogue (~3 us of stores) on top of 18-36 stages of ~1.2 us: ~20 % of a 64-channel layer's
// workgroup, ~12 % of a 128-channel one.  Here the NEXT tile's first halo group and first weight stage are issued during the last
// group of the current tile (the halo buffer and ring slot they land in are free by then), the epilogue's stores are left in flight,
// and the remaining PD - 1 weight stages of the next tile are issued right after the epilogue -- so that every store is OLDER than
// every load a counted wait later reasons about (outstanding <= N then bounds the loads among them whatever order stores and loads
// retire in).  Same arithmetic per tile in the same order: bit-identical to the plain form.
template <int BM, int BN, int WM, int WN, int SEG, int DB, bool BWS = false, int SCHED = 1, bool EPI = true>
__global__ __launch_bounds__(64 * WM * WN) void conv_halo_x3p_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NP = 2;
  constexpr int SW = BM / SEG + 2;
  constexpr int HR = (SEG * SW + 15) / 16 * 16;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NW = WM * WN;
  constexpr int NAOPS = NP * HR / 16, NAO = (NAOPS + NW - 1) / NW;
  constexpr int AP = HR * LDH, AB = NP * AP, BP = BN * LDH, BSL = NP * BP;
  constexpr int PD = DB - 1;
  constexpr int NBW = NP, NFA = NAO;
  static_assert(NW == 8 && (PD == 2 || PD == 3) && (BN == 128 || BN == 64), "persistent form: the pre-split tile shapes only");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][NP][HR][LDH]
  float* Bs = lds + 2 * AB;             // [DB][NP][BN][LDH]
  const float sx = df_h2_scale(*p.amax_x), sw = df_h2_scale(*p.amax_w);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int hx = p.x.h, wx = p.x.w, ldx = p.x.ld;
  const int KC = p.K / BK;
  const bool fwd = p.mode == DF_CONV_FWD;
  const int ntiles = p.tiles_m * p.tiles_n;
  RowDecode dec;
  dec.hw = p.hw_y; dec.w = p.y.w; dec.cls_mode = 0; dec.py = dec.px = 0; dec.hh = dec.wh = 0;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(p.x.ptr) - p.dshift), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const int brow = ((wave * 16) % BN) + (lane >> 2), bslot = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned plane_bytes = (unsigned)((int64_t)p.N * 9 * p.K * 2);
  // halo ops of this wave (as conv_halo_x3_kernel<XP>): op e = halo op (wave + e NW) mod NAOPS = plane pl, rows rb 16 .. + 15 -- wave-uniform,
  // recomputed where needed (scalar); per lane only the source offset of the tile being fetched is kept (poff)
  auto op_rb = [&](int e, int& pl, int& rb) {
    const int jo = (wave + e * NW) % NAOPS;
    pl = jo / (HR / 16);
    rb = jo - pl * (HR / 16);
  };
  auto op_seg = [&](int j) { return SEG == 1 ? 0 : min(j / SW, SEG - 1); };
  // ... and the cursors of the tile being FETCHED (A side: halo; B side: weights), which run ahead of the tile being multiplied
  unsigned poff[NAO];
  int a_oy = 0, ty_next = 0, kc_next = 0;
  unsigned sa_next = 0;
  unsigned boff = 0;
  int swb_next = 0, btx = 0, bkc = 0, bty = 0;
  const unsigned a_row = (unsigned)(wx * ldx * 4);
  const unsigned a_row_step = a_row - (unsigned)(KC * BK * 4);
  const int dtap = fwd ? p.K * 2 : -p.K * 2;
  const int NG = 3 * KC, NS = 3 * NG;
  auto tile_of = [&](int b, int& tm, int& tn) {
    const int swz = df_xcd_swizzle(b, ntiles);
    tn = swz % p.tiles_n;
    tm = swz / p.tiles_n;
  };
  // the NEXT tile's offsets are computed at the top of a tile -- while no accumulator is live (computed inside the last group, beside
  // 128 accumulator registers, they pushed the allocator into spilling in the main loop) -- and taken over where the cursors switch
  unsigned poffN[NAO], boffN = 0;
  int a_oyN = 0, rotN = 0;
  auto prep = [&](int b) {
    int tm, tn, n, oy, ox0;
    tile_of(b, tm, tn);
    dec(tm * BM, n, oy, ox0);
    const int64_t base = df_img_base(p.x, n);
#pragma unroll
    for (int e = 0; e < NAO; ++e) {
      int pl, rb;
      op_rb(e, pl, rb);
      const int j = rb * 16 + (lane >> 2), sg = op_seg(j);
      const int ix = ox0 - 1 + j - sg * SW;
      poffN[e] = (j < SEG * SW && ix >= 0 && ix < wx)
                     ? (unsigned)((base + ((int64_t)(oy + sg - 1) * wx + ix) * ldx) * 4 + pl * 64 + ((lane & 3) ^ ((j >> 2) & 3)) * 16) + p.dshift : DMA_BAD;
    }
    boffN = (unsigned)(((int64_t)(tn * BN + brow) * 9 * p.K + bslot * 8) * 2);
    a_oyN = oy;
    rotN = (p.rot && SEG <= 2) ? (1 + 2 * oy) % 3 : 0;
  };
  auto take_a = [&]() {
#pragma unroll
    for (int e = 0; e < NAO; ++e) poff[e] = poffN[e];
    a_oy = a_oyN;
    sa_next = rotN * a_row;
    ty_next = rotN;
    kc_next = 0;
  };
  auto take_b = [&]() {
    boff = boffN;
    swb_next = (fwd ? 0 : 8 * p.K * 2) + rotN * 3 * dtap;
    btx = 0;
    bkc = 0;
    bty = rotN;
  };
  auto fetch_a = [&](int abuf) {
    float* a = As + abuf * AB;
#pragma unroll
    for (int e = 0; e < NAO; ++e) {
      int pl, rb;
      op_rb(e, pl, rb);
      const int sg = op_seg(rb * 16 + (lane >> 2));
      const unsigned v = (unsigned)(a_oy + sg - 1 + ty_next) < (unsigned)hx ? poff[e] : DMA_BAD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(a + pl * AP + rb * 16 * LDH), 16, v, sa_next, 0, 0);
    }
    if (++kc_next == KC) {
      kc_next = 0;
      sa_next += a_row_step + BK * 4;
      if (++ty_next == 3) {
        ty_next = 0;
        sa_next -= 3 * a_row;
      }
    } else {
      sa_next += BK * 4;
    }
  };
  int bslot_ring = 0;
  auto issue_b = [&]() {
    float* b = Bs + bslot_ring * BSL + ((wave * 16) % BN) * LDH;
    const unsigned soff = (unsigned)(swb_next + btx * dtap);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(b + pl * BP), 16, boff, soff + pl * plane_bytes, 0, 0);
    if (++bslot_ring == DB) bslot_ring = 0;
    if (++btx == 3) {
      btx = 0;
      if (++bkc == KC) {
        bkc = 0;
        swb_next += 3 * dtap - KC * BK * 2 + BK * 2;
        if (++bty == 3) {
          bty = 0;
          swb_next -= 9 * dtap;
        }
      } else {
        swb_next += BK * 2;
      }
    }
  };
#define DF_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))
#define DF_STAGE_END() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  f32x16 acc[TM][TN], acc1[TM][TN];
  int cur_slot = 0;
  // the products of one tap stage: halo buffer gp, tap column tx, weight ring slot cur_slot
  auto mult = [&](auto tx_c, int gp) {
    constexpr int tx = decltype(tx_c)::value;
    const float* a = As + gp * AB + (wm * TM * 32 + li) * LDH + tx * LDH;
    const float* b0 = Bs + cur_slot * BSL + (wn * TN * 32 + li) * LDH;
    if (++cur_slot == DB) cur_slot = 0;
    const int sb = (li >> 2) & 3;
    // Round 6: ALL fragments of the 32-deep stage are read before its first product, and the scheduler is told the order (8 reads,
    // then one read behind each of the next 8 products, then the rest of the products).  Left to itself the compiler interleaved every
    // read one or two products ahead of its use with an s_waitcnt in between -- eight exposed LDS round trips per stage while the
    // SIMD's other wave, in lock step behind the same barrier, did the same.
    f16x8_t ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
    for (int q = 0; q < BK / 16; ++q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int sh = SEG == 1 ? 0 : 2 * ((wm * TM + i) * 32 / (BM / SEG));
        const int sa = ((li + tx + sh) >> 2) & 3;
        const float* ap = a + (i * 32 + sh) * LDH + (((2 * q + kh) ^ sa) * 4);
        ah[q][i] = *reinterpret_cast<const f16x8_t*>(ap);
        al[q][i] = *reinterpret_cast<const f16x8_t*>(ap + AP);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float* bp = b0 + j * 32 * LDH + (((2 * q + kh) ^ sb) * 4);
        bh[q][j] = *reinterpret_cast<const f16x8_t*>(bp);
        bl[q][j] = *reinterpret_cast<const f16x8_t*>(bp + BP);
      }
    }
#pragma unroll
    for (int q = 0; q < BK / 16; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q][i], bh[q][j], acc1[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bh[q][j], acc[i][j], 0, 0, 0);
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bl[q][j], acc1[i][j], 0, 0, 0);
        }
    static_assert(BK == 32 && TM == 2 && TN == 2, "the schedule below is written for 16 fragment reads and 24 products per stage");
    if constexpr (SCHED == 1) {
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // the first k step's 8 fragment reads
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one product ...
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // ... one read of the second k step behind it
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
  };
  // a steady-state stage: issue (tx == 0: the next group's halo; always: weight stage s + PD), multiply, counted wait (weight stage
  // s + 1, and at tx == 2 the halo, have landed: see conv_halo_x3_kernel), barrier
  auto steady = [&](auto tx_c, int gp) {
    constexpr int tx = decltype(tx_c)::value;
    // SCHED 3 (round 6): STAGGERED DMA issue.  An LDS-DMA instruction costs the issuing wave 100-200 cycles (MI355X_MICROARCH.md), a
    // stage has 3-5 of them per wave, and the two waves of a SIMD leave the stage barrier in lock step: with every wave issuing first,
    // the matrix pipe sat idle for the length of the DMA block once per stage.  The second half of the waves (the SIMD partners of
    // the first half) now multiplies first and issues afterwards.  Same ops, same counts in front of the counted wait below.
    const bool late = SCHED == 3 && wave >= NW / 2;
    if (!late) {
      if (tx == 0) fetch_a(gp ^ 1);
      issue_b();
    }
    mult(tx_c, gp);
    if (late) {
      if (tx == 0) fetch_a(gp ^ 1);
      issue_b();
    }
    constexpr int FA_IN = ((PD - 1) / 3) + (((PD - 1) % 3) > tx ? 1 : 0);
    DF_VMCNT((PD - 1) * NBW + FA_IN * NFA);
    DF_STAGE_END();
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;

  int bid = blockIdx.x;
  prep(bid);
  take_a();
  take_b();
  fetch_a(0);
#pragma unroll
  for (int d = 0; d < PD; ++d) issue_b();
  DF_VMCNT((PD - 1) * NBW);                             // the halo (issued first) and weight stage 0
  DF_STAGE_END();
  int gb = 0;                                           // halo buffer parity of the current tile's group 0
  while (true) {
    const int nbid = bid + gridDim.x;
    const bool has_next = nbid < ntiles;
    int tile_m, tile_n;
    tile_of(bid, tile_m, tile_n);
    if (has_next) prep(nbid);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acc1[i][j][e] = 0.f; }
    int g = 0;
    for (; g < NG - 1; ++g) {                           // every issue of these groups belongs to this tile (NS - PD >= 3 NG - 3)
      const int gp = (gb + g) & 1;
      steady(T0{}, gp);
      steady(T1{}, gp);
      steady(T2{}, gp);
    }
    const int gp = (gb + g) & 1;
    // last group.  With a next tile: the halo fetch is ITS group 0, and of its weight stages only stage 0 goes out before the epilogue;
    // without: nothing is left to issue but (PD == 2) this tile's last weight stage, and the waits drain.  (One copy of the products
    // for both cases: two copies cost the main loop 60 spill stores per group.)
    if (has_next) {
      take_a();
      fetch_a(gp ^ 1);
      if constexpr (PD == 3) take_b();
    }
    if (PD == 2 || has_next) issue_b();                 // PD == 2: this tile's stage NS - 1; PD == 3: the next tile's stage 0
    mult(T0{}, gp);
    if (has_next) DF_VMCNT((PD - 1) * NBW + NFA);       // (the steady pattern held up to here)
    else DF_VMCNT(0);
    DF_STAGE_END();
    if (PD == 2 && has_next) {
      take_b();
      issue_b();                                        // the next tile's stage 0
    }
    mult(T1{}, gp);
    if (has_next) DF_VMCNT(PD == 2 ? NBW : NFA + NBW);  // this tile's last weight stage is older than the next tile's halo / stage 0
    else DF_VMCNT(0);
    DF_STAGE_END();
    mult(T2{}, gp);
    DF_VMCNT(0);                                        // the next tile's halo group 0 and weight stage 0: nothing else is in flight
    DF_STAGE_END();
    // epilogue scratch = the halo buffer the last group just left (the other one is receiving the next tile's group 0)
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const float ix = 1.f / sx, iw = 1.f / sw;
    // EPI (round 6; bit-identical to the plain form): the fold of the cross-term accumulator, the two operand scales, the bias and -- for a
    // plane output -- the output scale are TWO fused multiply-adds per element here (every factor but the bias is a power of two:
    // (acc + acc1 / 2048) ix iw + b and fma(fma(acc1, 1 / 2048, acc), ix iw, b) round the same sum once; t = y s likewise as
    // fma(., ix iw s, b s)), and the shared epilogue takes the values as they are (PRE): five instructions per output element less in a
    // phase that is VALU-bound and in which no wave of the workgroup issues a product (13 / 22 % of a 256 x 128 / 512 x 64 workgroup's
    // time, profiles/r06_conv_experiments.txt).  Not for bf16 outputs and the folded BatchNorm + GELU epilogue (inference).
    if constexpr (EPI) {     // (the launcher picks this instance only for outputs PRE covers; ONE inlined epilogue per instance: two
                             //  copies cost the K loop hundreds of spill instructions)
      const float sy = p.y.elt == 2 ? df_h2_scale(*p.bound_y) : 1.f;
      const float fs = ix * iw * sy;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co = tile_n * BN + (wn * TN + j) * 32 + li;
        const float bs = (p.bias ? p.bias[co] : 0.f) * sy;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(fmaf(acc1[i][j][e], H2_LO_INV, acc[i][j][e]), fs, bs);
      }
      conv_epilogue<BM, BN, WM, WN, BWS, true>(p, acc, As + gp * AB, dec, tile_m * BM, p.M, tile_n * BN, tile_m, tid_o);
    } else {
      // fold the cross terms in and take the two power-of-two scales out (exact multiplications), then the shared epilogue
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] + acc1[i][j][e] * H2_LO_INV) * ix * iw;
      conv_epilogue<BM, BN, WM, WN, BWS>(p, acc, As + gp * AB, dec, tile_m * BM, p.M, tile_n * BN, tile_m, tid_o);
    }
    if (!has_next) break;
    DF_STAGE_END();                                     // every wave is done with the scratch before the next halo fetch may overwrite it
#pragma unroll
    for (int d = 1; d < PD; ++d) issue_b();             // next tile's weight stages 1 .. PD - 1: younger than every store of the epilogue
    bid = nbid;
    gb += NG;
  }
#undef DF_VMCNT
#undef DF_STAGE_END
#endif
}

template <int BM, int BN, int WM, int WN, int SEG, int DB, bool BWS = false>
static int launch_conv_halo_x3p(const ConvParams& p, hipStream_t s) {
  constexpr int HR = (SEG * (BM / SEG + 2) + 15) / 16 * 16;
  const size_t lds_bytes = (size_t)(2 * 2 * HR + DB * 2 * BN) * LDH * sizeof(float);
  const int ntiles = p.tiles_m * p.tiles_n;
  const int grid = ntiles < 256 ? (ntiles + 7) / 8 * 8 : 256;     // one workgroup per CU (the LDS holds one); a multiple of the 8 XCDs
  // DF_X3P_SCHED (A/B; every form multiplies the same fragments in the same order -- bit-identical):  0 = the compiler's own order of
  // fragment reads and products (rounds 4-5), 1 = all reads of the stage's first k step first, the second step's one behind each of
  // the next products, 3 = 0 + the second half of the waves issue their LDS-DMA AFTER the products (the SIMD partners of the first
  // half).  Measured in one process sequence on one box (profiles/r06_conv_experiments.txt): 1 = 0 within 0.3 %; 3 gains 3 % on the
  // 512 x 64 tiles (five DMA instructions per wave and stage) and loses 2 % on the 256 x 128 ones (three) -- default: 3 for BN = 64.
  static const int sched_env = getenv("DF_X3P_SCHED") ? atoi(getenv("DF_X3P_SCHED")) : -1;
  const int sched = sched_env >= 0 ? sched_env : (BN == 64 ? 3 : 0);
#define DF_X3P_GO(S, E)                                                                                                   \
  do {                                                                                                                   \
    DF_SET_LDS_ONCE((conv_halo_x3p_kernel<BM, BN, WM, WN, SEG, DB, BWS, S, E>), (int)lds_bytes);                         \
    hipLaunchKernelGGL((conv_halo_x3p_kernel<BM, BN, WM, WN, SEG, DB, BWS, S, E>), dim3(grid), dim3(64 * WM * WN), lds_bytes, s, p); \
  } while (0)
  static const int lean_env = getenv("DF_X3P_EPI") ? atoi(getenv("DF_X3P_EPI")) : 1;      // 0: fold, then the plain shared epilogue (A/B; bit-identical)
  // the pre-biased / pre-scaled epilogue form covers fp32 and plane outputs with the bias / statistics epilogues (training, every layer
  // of it); bf16 outputs, the folded BatchNorm + GELU epilogue (inference) and accumulation into planes take the plain instance
  const bool lean_epi = lean_env && p.y_bytes && p.y.elt != 1 && p.epi != DF_EPI_BN_GELU && !(p.y.elt == 2 && p.accumulate);
  if (!lean_epi) {
    if (sched == 3) DF_X3P_GO(3, false);
    else DF_X3P_GO(0, false);
  } else if (sched == 3) DF_X3P_GO(3, true);
  else DF_X3P_GO(0, true);
#undef DF_X3P_GO
  DF_CHECK_LAUNCH();
  return DF_OK;
}

}  // namespace

int df_launch_conv_halo_x3p(const dfconv::ConvParams& p, int bn, int seg, bool bws, hipStream_t s) {
  if (bn == 128) {
    if (bws) {
      if (seg == 1) return launch_conv_halo_x3p<256, 128, 4, 2, 1, 4, true>(p, s);
      if (seg == 2) return launch_conv_halo_x3p<256, 128, 4, 2, 2, 4, true>(p, s);
      return launch_conv_halo_x3p<256, 128, 4, 2, 4, 4, true>(p, s);
    }
    if (seg == 1) return launch_conv_halo_x3p<256, 128, 4, 2, 1, 4>(p, s);
    if (seg == 2) return launch_conv_halo_x3p<256, 128, 4, 2, 2, 4>(p, s);
    return launch_conv_halo_x3p<256, 128, 4, 2, 4, 4>(p, s);
  }
  if (bws) return seg == 1 ? launch_conv_halo_x3p<512, 64, 8, 1, 1, 3, true>(p, s) : launch_conv_halo_x3p<512, 64, 8, 1, 2, 3, true>(p, s);
  return seg == 1 ? launch_conv_halo_x3p<512, 64, 8, 1, 1, 3>(p, s) : launch_conv_halo_x3p<512, 64, 8, 1, 2, 3>(p, s);
}
