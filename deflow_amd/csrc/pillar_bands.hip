// Pillarisation, second generation: band-bucketed stable counting sort + fused feature net + canvas, 4 launches per
// cloud set in inference (6 in training), every canvas byte written exactly once, no library sort, no float atomics,
// no spin-waits.  Replaces keys/scan/compact/rocPRIM-sort/gather/cells/stats/canvas + a separate zero-fill (~15
// launches) of csrc/pillarize.hip on the DeFlow path ([REF deflow.py:27-30,82-83]; algorithm: oracle/ref_torch.py).
//
// The BEV grid of every sample is cut into BANDS of R rows (R * W <= 2048 cells).  Sorting points by cell =
//   pass 1 (global, stable): bucket by (sample, band)        -- df_pillar2_hist / _scan / _scatter
//   pass 2 (LDS, stable):    counting sort by cell in band    -- df_pillar2_band, one workgroup per (band, sample)
// and because band = row / R is monotone in the cell key, the concatenation of the sorted buckets IS the array sorted by
// (sample, cell) with input order kept inside a cell -- the order the first generation got from a stable radix sort, so
// every consumer (feature-net backward, sparse edge kernels, gather backward) reads the same layout as before.
//
//   hist     per 1024-point tile: points per band (LDS integer counters) and valid points            [S][NB+1][nblk]
//   scan     one wavefront per (sample, column): exclusive scan over the tiles (DPP-free shuffles)     offsets + totals
//   scatter  per tile: stable ranks from wave ballots (lanes with equal band = AND of per-bit ballots), writes the
//            order-preserving compaction the result dict needs (points, voxel coords, indices, centre offsets) and
//            the bucketed (key, index, xyz) arrays
//   band     the workgroup owning a band: LDS histogram of its <= 2048 cells -> scan -> one wavefront assigns stable
//            positions chunk by chunk (segmented rank = popcount of the equal-cell lane mask below the lane) ->
//            permuted copy into the sorted arrays and an LDS image of the sorted points -> after the last barrier one
//            wavefront streams zeros into the band's EMPTY canvas cells while the other three walk the occupied
//            pillars (8 lanes each, run read from LDS): 9-d feature, Linear(9->32) + BN1d + ReLU, mean / max, one
//            128-B store per pillar.
// Training needs the BatchNorm1d batch statistics before it can normalise: band<sort, stats> leaves per-band partial
// sums (fp32, finalised in fp64 by df_pfn_bn_finalize), band<canvas> then re-derives the cell table from the sorted keys.
#include <cstdlib>

#include "common.h"
#include "pillar_common.h"

namespace {

constexpr int TILE = 1024;        // points per workgroup in hist / scatter (4 rounds x 256 threads)
constexpr int MAX_BANDS = 512;    // LDS tables in hist / scatter
constexpr int BAND_CELLS = 2048;  // max cells per band (LDS cell tables of the band kernel)
constexpr int CHUNK = 1024;       // bucket elements ranked / staged per round of the band kernel (4 per thread)

struct P2Geom {
  df_pillar_geom g;
  int R, NB;  // rows per band, bands per sample
};

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total) {
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  total = __shfl(x, 63);
  return x - v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
// lanes of the wave holding the same `c` (among `valid` lanes): AND of per-bit ballots
template <int NBITS>
__device__ __forceinline__ unsigned long long match_lanes(int c, bool valid) {
  unsigned long long mask = __ballot(valid);
#pragma unroll
  for (int b = 0; b < NBITS; ++b) {
    const bool bit = (c >> b) & 1;
    const unsigned long long bal = __ballot(bit);
    mask &= bit ? bal : ~bal;
  }
  return mask;
}

// ------------------------------------------------------------------------------------------------ hist ----
__global__ __launch_bounds__(256) void p2_hist_kernel(const float* __restrict__ pts, int N, P2Geom q, int nblk,
                                                      int32_t* __restrict__ hist) {
  __shared__ int h[MAX_BANDS + 1];
  const int s = blockIdx.y, blk = blockIdx.x, ncol = q.NB + 1;
  for (int i = threadIdx.x; i < ncol; i += 256) h[i] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    const int n = blk * TILE + r * 256 + threadIdx.x;
    bool ok = false;
    int cx = 0, cy = 0;
    if (n < N) {
      const float* p = pts + ((int64_t)s * N + n) * 3;
      ok = voxelize(q.g, p[0], p[1], p[2], cx, cy);
    }
    if (ok) atomicAdd(&h[cy / q.R], 1);  // integer LDS counters: totals do not depend on the order of arrival
    const unsigned long long vb = __ballot(ok);
    if ((threadIdx.x & 63) == 0) atomicAdd(&h[q.NB], __popcll(vb));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ncol; i += 256) hist[((int64_t)s * ncol + i) * nblk + blk] = h[i];
}

// ------------------------------------------------------------------------------------------------ scan ----
// one wavefront per (sample, column): exclusive scan of its nblk tile counters
__global__ __launch_bounds__(256) void p2_scan_kernel(const int32_t* __restrict__ hist, int ncols_total, int ncol, int nblk,
                                                      int32_t* __restrict__ off, int32_t* __restrict__ tot,
                                                      int32_t* __restrict__ counts) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= ncols_total) return;
  const int32_t* h = hist + (int64_t)col * nblk;
  int32_t* o = off + (int64_t)col * nblk;
  int carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += 64) {
    const int i = b0 + lane;
    const int v = i < nblk ? h[i] : 0;
    int t;
    const int e = wave_excl_scan(v, lane, t);
    if (i < nblk) o[i] = carry + e;
    carry += t;
  }
  if (lane == 0) {
    tot[col] = carry;
    if (col % ncol == ncol - 1) counts[col / ncol] = carry;
  }
}

// --------------------------------------------------------------------------------------------- scatter ----
struct P2Scatter {
  const float* pts;
  const int32_t *off, *tot;
  float* points_c;
  int32_t* coords_c;
  int64_t* idx_c;
  float* offs_c;
  int32_t* cpos;
  uint32_t *bkey, *bidx;
  float* bpts;
  int32_t* bucket0;   // [S][NB] absolute sorted position of every bucket's first element (written by tile 0 of each sample)
  int S, N, nblk;
};

template <int NBITS>
__global__ __launch_bounds__(256) void p2_scatter_kernel(P2Scatter a, P2Geom q) {
  extern __shared__ int sm[];
  const int NB = q.NB, ncol = NB + 1;
  int* bstart = sm;                 // [NB] bucket start inside the sample's sorted segment
  int* blkoff = sm + NB;            // [NB] this tile's offset inside each bucket
  int* unit = sm + 2 * NB;          // [16][NB] per (round, wave) counts -> exclusive prefix over the 16 units
  int* vcnt = sm + 18 * NB;         // [16] valid points per unit -> exclusive prefix
  int* misc = vcnt + 16;            // [0] = first sorted position of this sample
  const int s = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const df_pillar_geom& g = q.g;
  if (wave == 0) {  // sorted segment of sample s starts after the valid points of samples 0 .. s-1
    int v = 0;
    for (int k = lane; k < s; k += 64) v += a.tot[(int64_t)k * ncol + NB];
    v = wave_sum(v);
    if (lane == 0) misc[0] = v;
    int carry = 0;   // bucket starts: exclusive scan of this sample's band totals
    for (int b0 = 0; b0 < NB; b0 += 64) {
      const int i = b0 + lane;
      const int t = i < NB ? a.tot[(int64_t)s * ncol + i] : 0;
      int tt;
      const int e = wave_excl_scan(t, lane, tt);
      if (i < NB) bstart[i] = carry + e;
      carry += tt;
    }
  }
  for (int i = threadIdx.x; i < NB; i += 256) blkoff[i] = a.off[((int64_t)s * ncol + i) * a.nblk + blk];
  for (int i = threadIdx.x; i < 16 * NB; i += 256) unit[i] = 0;
  __syncthreads();
  float px[4], py[4], pz[4];
  int band[4], cxy[4], rk[4], vrk[4];
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = blk * TILE + r * 256 + threadIdx.x;
    ok[r] = false;
    band[r] = 0; cxy[r] = 0;
    px[r] = py[r] = pz[r] = 0.f;
    if (n < a.N) {
      const float* p = a.pts + ((int64_t)s * a.N + n) * 3;
      px[r] = p[0]; py[r] = p[1]; pz[r] = p[2];
      int cx = 0, cy = 0;
      ok[r] = voxelize(g, px[r], py[r], pz[r], cx, cy);
      if (ok[r]) { band[r] = cy / q.R; cxy[r] = cy * g.gx + cx; }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long vb = __ballot(ok[r]);
    vrk[r] = __popcll(vb & lt);
    if (lane == 0) vcnt[r * 4 + wave] = __popcll(vb);
    const unsigned long long m = match_lanes<NBITS>(band[r], ok[r]);
    rk[r] = __popcll(m & lt);
    if (ok[r] && rk[r] == 0) unit[(r * 4 + wave) * NB + band[r]] = __popcll(m);   // one writer per (unit, band)
  }
  __syncthreads();
  for (int b = threadIdx.x; b < NB; b += 256) {  // exclusive prefix over the 16 units, per band
    int run = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = unit[u * NB + b];
      unit[u * NB + b] = run;
      run += c;
    }
  }
  if (threadIdx.x == 0) {
    int run = 0;
    for (int u = 0; u < 16; ++u) { const int c = vcnt[u]; vcnt[u] = run; run += c; }
  }
  __syncthreads();
  const int voff = a.off[((int64_t)s * ncol + NB) * a.nblk + blk];
  const int seg0 = misc[0];
  if (blk == 0)
    for (int i = threadIdx.x; i < NB; i += 256) a.bucket0[(int64_t)s * NB + i] = seg0 + bstart[i];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = blk * TILE + r * 256 + threadIdx.x;
    if (n >= a.N) continue;
    const int64_t flat = (int64_t)s * a.N + n;
    if (!ok[r]) { a.cpos[flat] = -1; continue; }
    const int u = r * 4 + wave;
    // order-preserving compaction (the result dict's per-sample lists are row slices of these)
    const int pos = voff + vcnt[u] + vrk[r];
    const int cy = cxy[r] / g.gx, cx = cxy[r] - cy * g.gx;
    const int64_t o = (int64_t)s * a.N + pos;
    a.points_c[o * 3 + 0] = px[r]; a.points_c[o * 3 + 1] = py[r]; a.points_c[o * 3 + 2] = pz[r];
    a.coords_c[o * 3 + 0] = 0; a.coords_c[o * 3 + 1] = cy; a.coords_c[o * 3 + 2] = cx;
    a.idx_c[o] = n;
    // centre = c * vs + min + vs / 2, three roundings, no contraction (DynamicVoxelizer._get_point_offsets)
    const float ctx = __fadd_rn(__fadd_rn(__fmul_rn((float)cx, g.vx), g.minx), g.vx * 0.5f);
    const float cty = __fadd_rn(__fadd_rn(__fmul_rn((float)cy, g.vy), g.miny), g.vy * 0.5f);
    const float ctz = __fadd_rn(__fadd_rn(0.f, g.minz), g.vz * 0.5f);
    a.offs_c[o * 3 + 0] = __fsub_rn(px[r], ctx);
    a.offs_c[o * 3 + 1] = __fsub_rn(py[r], cty);
    a.offs_c[o * 3 + 2] = __fsub_rn(pz[r], ctz);
    a.cpos[flat] = pos;
    // stable bucket position: sample segment + bucket start + tiles before + units before + lanes before
    const int64_t d = (int64_t)seg0 + bstart[band[r]] + blkoff[band[r]] + unit[u * NB + band[r]] + rk[r];
    a.bkey[d] = (uint32_t)(s * g.gy * g.gx + cxy[r]);
    a.bidx[d] = (uint32_t)flat;
    a.bpts[d * 3 + 0] = px[r]; a.bpts[d * 3 + 1] = py[r]; a.bpts[d * 3 + 2] = pz[r];
  }
}

// ------------------------------------------------------------------------------------------------ band ----
struct P2Band {
  const uint32_t *in_key, *in_idx;   // bucketed (SORT) or already sorted (!SORT) arrays
  const float* in_pts;
  const int32_t *tot, *bucket0;
  uint32_t *key_sorted, *idx_sorted;
  float* pts_sorted;
  int32_t* cell_rng;                 // optional [S*H*W][2]
  const float *w_pfn, *bn_ss;
  float* partial;                    // STATS: [S][NB][32][2]
  df_img out;
  int bn_stride, mode, S;
  int dbg;   // timing ablations (env DF_P2_DBG): bit 0 = no pillar loop, bit 1 = no zero stores, bit 2 = no sort / copy
  uint32_t* occ;   // round 5, PERSISTENT canvas (df_pillar2_band_sp): [S][NB][64] occupancy words of the previous call, updated in place
  unsigned* amax_out;   // optional (df_pillar2_band_sp): max of the canvas values this call writes (>= 0 after the ReLU), atomic max of the bit pattern
};

// BC = LDS cell-table size (cells per band <= BC): 2048, or 1024 for thin bands -- 24.7 KB of LDS instead of 32.9, i.e. 6
// instead of 4 resident workgroups per CU
template <bool SORT, bool STATS, bool CANVAS, int BC = BAND_CELLS>
__global__ __launch_bounds__(256) void p2_band_kernel(P2Band a, P2Geom q) {
  constexpr int BAND_CELLS = BC;      // (shadows the file-level maximum inside this kernel)
  __shared__ int cnt[BAND_CELLS];     // points per cell of the band
  __shared__ int pos0[BAND_CELLS];    // running / final END position of each cell's run, relative to the bucket
  __shared__ int stage[CHUNK];        // cell of a bucket element -> its position after ranking
  __shared__ float spts[CHUNK * 3];   // sorted points with bucket position < CHUNK (the rest is read back from L2)
  __shared__ int misc[12];
  __shared__ int clsw[5 * 4];      // per size class: occupied cells of each wavefront
  static_assert(CHUNK * 3 >= 256 * 8, "the statistics reduction reuses the point image");
  // the occupied-cell list reuses the ranking scratch as 16-bit entries: up to BAND_CELLS cells can be occupied, which is
  // more than CHUNK ints for the wide (2048-cell) band -- an int list overflowed into `spts` beyond 1024 occupied cells
  static_assert(BAND_CELLS <= 2 * CHUNK && BAND_CELLS <= 65536, "occupied-cell list: BAND_CELLS 16-bit entries must fit in stage[]");
  unsigned short* occ = reinterpret_cast<unsigned short*>(stage);
  float* red = spts;
  const df_pillar_geom& g = q.g;
  const int band = blockIdx.x, s = blockIdx.y, NB = q.NB, ncol = NB + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = band * q.R, rows = min(q.R, g.gy - row0), ncb = rows * g.gx;
  for (int i = threadIdx.x; i < ncb; i += 256) cnt[i] = 0;
  __syncthreads();
  // Barriers below order LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also drain every
  // outstanding global store -- the canvas zeros issued early would then be waited for at the next barrier instead of
  // streaming out underneath the sort.  Only a bucket beyond CHUNK elements reads its own global writes back and needs
  // the full fence.
  const int64_t g0 = a.bucket0[(int64_t)s * NB + band];   // wave-uniform scalar loads
  const int n = a.tot[(int64_t)s * ncol + band];
  auto bar = [&]() {
    if (n > CHUNK) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  const uint32_t base_key = (uint32_t)((s * g.gy + row0) * g.gx);
  // this thread's share of the first CHUNK bucket elements stays in registers from here to the permuted copy: one global
  // round trip feeds the histogram, the ranking and the copy (buckets beyond CHUNK elements take the slower loop below)
  constexpr int EPT = CHUNK / 256;
  int kreg[EPT];
  uint32_t ireg[EPT];
  float xreg[EPT][3];
  const int m0 = min(n, CHUNK);
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = threadIdx.x + 256 * e;
    kreg[e] = -1;
    if (i < m0) {
      kreg[e] = (int)(a.in_key[g0 + i] - base_key);
      const float* pp = a.in_pts + (g0 + i) * 3;
      xreg[e][0] = pp[0]; xreg[e][1] = pp[1]; xreg[e][2] = pp[2];
      if (SORT) ireg[e] = a.in_idx[g0 + i];
    }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e)
    if (kreg[e] >= 0) atomicAdd(&cnt[kreg[e]], 1);
  for (int i = CHUNK + threadIdx.x; i < n; i += 256) atomicAdd(&cnt[a.in_key[g0 + i] - base_key], 1);
  bar();
  // occupied cells are listed BY SIZE CLASS (1, 2, 3-4, 5-8, 9+ points): the 8 pillars a wavefront walks together in the
  // pillar loop then have similar lengths, instead of every step costing the longest of 8 random pillars
  constexpr int NCLS = 5;
  auto size_class = [](int k) { return k <= 1 ? 0 : k == 2 ? 1 : k <= 4 ? 2 : k <= 8 ? 3 : 4; };
  int cls_ex[NCLS];            // this thread's first slot inside each class
  int occ_mask = 0, cell_cls = 0;   // which of this thread's 8 cells are occupied / their classes (3 bits each)
  {  // exclusive scan of cnt -> pos0 (8 consecutive cells per thread), and of the per-class occupancy counts
    const int c0 = threadIdx.x * (BAND_CELLS / 256);
    int loc[BAND_CELLS / 256], sum = 0, ncls[NCLS] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < BAND_CELLS / 256; ++k) {
      loc[k] = (c0 + k < ncb) ? cnt[c0 + k] : 0;
      sum += loc[k];
      if (loc[k] > 0) {
        const int c = size_class(loc[k]);
        occ_mask |= 1 << k;
        cell_cls |= c << (3 * k);
#pragma unroll
        for (int q = 0; q < NCLS; ++q) ncls[q] += (c == q);
      }
    }
    int wt;
    int ex = wave_excl_scan(sum, lane, wt);
    if (lane == 63) misc[2 + wave] = wt;
#pragma unroll
    for (int q = 0; q < NCLS; ++q) {
      int wo;
      cls_ex[q] = wave_excl_scan(ncls[q], lane, wo);
      if (lane == 63) clsw[q * 4 + wave] = wo;
    }
    __syncthreads();
    for (int w = 0; w < wave; ++w) ex += misc[2 + w];
    int base = 0;   // class q starts after all cells of classes < q
#pragma unroll
    for (int q = 0; q < NCLS; ++q) {
      for (int w = 0; w < wave; ++w) cls_ex[q] += clsw[q * 4 + w];
      cls_ex[q] += base;
      base += clsw[q * 4] + clsw[q * 4 + 1] + clsw[q * 4 + 2] + clsw[q * 4 + 3];
    }
    if (threadIdx.x == 0) misc[8] = base;
#pragma unroll
    for (int k = 0; k < BAND_CELLS / 256; ++k) {
      if (c0 + k < ncb) pos0[c0 + k] = SORT ? ex : ex + loc[k];   // !SORT: the run END right away
      ex += loc[k];
    }
  }
  __syncthreads();
  const int n_occ = misc[8];
  float* __restrict__ op = CANVAS ? reinterpret_cast<float*>(a.out.ptr) + df_img_base(a.out, s) + (int64_t)row0 * g.gx * a.out.ld : nullptr;
  // zeros into the band's empty cells (128 B each, 8 lanes x 16 B) as soon as the histogram is known; every canvas byte is
  // written exactly once.  Measured at B = 16 (tools/bench_pillar.py, DF_P2_DBG ablations): this stream runs at 7.5 TB/s
  // (7.8 us per pair), the pillar loop costs 4 us (VALU-bound), the sort 1.5 us, hist + scan + scatter 6.1 us -- and the
  // parts add up rather than overlap; filling after the pillar loop in every other workgroup (so that half of a CU's waves
  // would compute while the other half stream) measured the same within noise and is not kept.  Neither is carrying the
  // zeros on the two small kernels in front (as a prologue of their workgroups: the single in-order vmcnt makes every wave
  // wait for its zero stores before it can use a loaded point; as dedicated store-only workgroups: the kernels simply get as
  // much longer as the stream takes, 5.1 TB/s there) with this kernel writing occupied cells only: 27.7-28.8 us per pair
  // against 23.4 -- kernel-trace: hist 19 -> 144 us, this kernel 278 -> 147 us, nothing overlapped.
  // SQ counters (profiles/r02_pmc_band.txt): waves wait 53 % of their cycles, VALU 34 % busy.  A wave that streams zeros is
  // stuck in its store loop -- 64 stores fill the vmcnt window, every further one waits for HBM -- so the stream is given to
  // ONE wavefront, after the last barrier, while the other three walk the pillars (see below).  That, too, measures the same
  // (22.7-23.6 us; ablations unchanged: no zero stores 14.4, no pillar loop 20.1, neither 8.9): while the chip's HBM queues
  // are full of the 58 MB-per-pair zero stream, every workgroup's two dependent global round trips at its start (bucket
  // table, then bucket contents) take several times longer, so the stream's 9 us are paid on top of the rest wherever the
  // stores are issued from.  Hiding them needs another, compute-bound kernel running beside (the UNet of the previous
  // forward on a second stream) -- not done.
  if (SORT && !(a.dbg & 4)) {
    for (int c0 = 0; c0 < n; c0 += CHUNK) {
      const int m = min(CHUNK, n - c0);
      if (c0 > 0) {   // rare: a bucket beyond CHUNK elements -- refill the registers
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const int i = threadIdx.x + 256 * e;
          kreg[e] = -1;
          if (i < m) {
            kreg[e] = (int)(a.in_key[g0 + c0 + i] - base_key);
            const float* pp = a.in_pts + (g0 + c0 + i) * 3;
            xreg[e][0] = pp[0]; xreg[e][1] = pp[1]; xreg[e][2] = pp[2];
            ireg[e] = a.in_idx[g0 + c0 + i];
          }
        }
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e)
        if (kreg[e] >= 0) stage[threadIdx.x + 256 * e] = kreg[e];
      bar();
      if (wave == 0) {
        // stable ranks, 64 bucket elements at a time: position = run end of the cell so far + equal-cell lanes below
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int j0 = 0; j0 < m; j0 += 64) {
          const int i = j0 + lane;
          const bool v = i < m;
          const int c = v ? stage[i] : 0;
          const unsigned long long mk = match_lanes<11>(c, v);
          const int basep = v ? pos0[c] : 0;
          const int below = __popcll(mk & lt);
          if (v) {
            stage[i] = basep + below;
            if ((mk >> lane) == 1ull) pos0[c] = basep + __popcll(mk);   // highest lane of the group advances the cell
          }
        }
      }
      bar();
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        if (kreg[e] < 0) continue;
        const int d = stage[threadIdx.x + 256 * e];
        a.key_sorted[g0 + d] = (uint32_t)kreg[e] + base_key;
        a.idx_sorted[g0 + d] = ireg[e];
        float* o = a.pts_sorted + (g0 + d) * 3;
        o[0] = xreg[e][0]; o[1] = xreg[e][1]; o[2] = xreg[e][2];
        if (d < CHUNK) { spts[d * 3 + 0] = xreg[e][0]; spts[d * 3 + 1] = xreg[e][1]; spts[d * 3 + 2] = xreg[e][2]; }
      }
      bar();
    }
  } else {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = threadIdx.x + 256 * e;
      if (kreg[e] >= 0) { spts[i * 3 + 0] = xreg[e][0]; spts[i * 3 + 1] = xreg[e][1]; spts[i * 3 + 2] = xreg[e][2]; }
    }
    bar();
  }
  // ---- feature net over the band's cells: 8 lanes per cell, lane `sub` owns output channels 4*sub .. 4*sub+3
  const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const float* __restrict__ gpts = SORT ? a.pts_sorted : a.in_pts;
  PfnCtx c;
  pfn_load_w(c, a.w_pfn, sub);
  if (CANVAS) pfn_load_bn(c, a.bn_ss + (int64_t)s * a.bn_stride, sub);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float cmax = 0.f;
  // list of the band's occupied cells (the ranking scratch is free now): the pillar loop below then keeps all 32 lane
  // groups busy instead of having most of them skip empty cells while one walks a pillar
#pragma unroll
  for (int k = 0; k < BAND_CELLS / 256; ++k)
    if ((occ_mask >> k) & 1) {
      const int c = (cell_cls >> (3 * k)) & 7;
#pragma unroll
      for (int q = 0; q < NCLS; ++q)
        if (c == q) occ[cls_ex[q]++] = (unsigned short)(threadIdx.x * (BAND_CELLS / 256) + k);
    }
  if (a.cell_rng)   // dense [start, end) table, 8 B per cell
    for (int cell = threadIdx.x; cell < ncb; cell += 256) {
      const int k = cnt[cell], e = pos0[cell];
      int2 v;
      v.x = k ? (int32_t)(g0 + e - k) : 0;
      v.y = k ? (int32_t)(g0 + e) : 0;
      *reinterpret_cast<int2*>(a.cell_rng + 2 * ((int64_t)base_key + cell)) = v;
    }
  bar();
  // ---- from here on no workgroup barrier (canvas kernels): wave 3 streams the zeros of the band's empty cells (128 B each,
  // 8 lanes x 16 B; every canvas byte is written exactly once) while waves 0-2 walk the pillars
  const int ngrp = CANVAS ? 24 : 32;
  if (CANVAS && wave == 3 && a.occ) {
    // PERSISTENT canvas (round 5): the buffer is zero wherever the PREVIOUS call of this (sample set, band geometry) left no pillar
    // -- the invariant its owner keeps (deflow.py) -- so only cells that were occupied then and are empty now need their zeros: the
    // band's occupancy words (64 x 32 cells, touched by this workgroup alone) are read, compared and rewritten in place.  The 128-byte
    // zero rows of a dense canvas were 87 % of the stage's bytes (537 MB per cloud at B = 16, 4.2 TB/s: profiles/r05_pmc_band.txt).
    uint32_t* ow = a.occ + ((int64_t)s * NB + band) * 64;
    const unsigned prevw = ow[lane];
    unsigned curw = 0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < ncb; c0 += 64) {
      const int c = c0 + lane;
      const bool on = c < ncb && cnt[c] > 0;
      const unsigned long long m = __ballot(on);
      const unsigned plo = (unsigned)__shfl((int)prevw, c0 >> 5), phi = (unsigned)__shfl((int)prevw, (c0 >> 5) + 1);
      const unsigned long long pv = ((unsigned long long)phi << 32) | plo;
      if (lane == (c0 >> 5)) curw = (unsigned)m;
      if (lane == (c0 >> 5) + 1) curw = (unsigned)(m >> 32);
      if (c < ncb && (((pv & ~m) >> lane) & 1ull)) {
        float* o = op + (int64_t)c * a.out.ld;
#pragma unroll
        for (int k = 0; k < 8; ++k) st4(o + 4 * k, z);
      }
    }
    ow[lane] = curw;
    return;
  }
  if (CANVAS && wave == 3) {
    if (!(a.dbg & 2)) {
      const int zg = lane >> 3;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      for (int c = zg; c < ncb; c += 8)
        if (cnt[c] == 0) st4(op + (int64_t)c * a.out.ld + 4 * sub, z);
    }
    return;
  }
  for (int oi = grp; oi < ((a.dbg & 1) ? 0 : n_occ); oi += ngrp) {
    const int cell = occ[oi];
    const int k = cnt[cell];
    const int e = pos0[cell], b = e - k;
    auto ld = [&](int i, float (&p)[3]) {
      if (i < CHUNK) { p[0] = spts[i * 3]; p[1] = spts[i * 3 + 1]; p[2] = spts[i * 3 + 2]; }
      else { const float* gq = gpts + (g0 + i) * 3; p[0] = gq[0]; p[1] = gq[1]; p[2] = gq[2]; }
    };
    float mx, my, mz, ctx, cty, ctz;
    if (k == 1) {   // ~45 % of the pillars: the mean IS the point (x / 1 = x exactly) -- no sum loop, no IEEE divisions
      float p[3];
      ld(b, p);
      mx = p[0]; my = p[1]; mz = p[2];
    } else {  // pillar mean: plain left-to-right sums in input order, as pfn_mean (csrc/pillar_common.h) does
      float sx = 0.f, sy = 0.f, sz = 0.f;
      for (int i = b; i < e; ++i) { float p[3]; ld(i, p); sx += p[0]; sy += p[1]; sz += p[2]; }
      const float inv = (float)k;
      mx = sx / inv; my = sy / inv; mz = sz / inv;
    }
    pfn_centre(g, row0 * g.gx + cell, ctx, cty, ctz);
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    for (int i = b; i < e; ++i) {
      float p[3], f[9], u[4];
      ld(i, p);
      pfn_feat(p, mx, my, mz, ctx, cty, ctz, f);
      pfn_linear(c, f, u);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (STATS) { acc[kk] += u[kk]; acc[4 + kk] += u[kk] * u[kk]; }
        if (CANVAS) {
          const float v = fmaxf(fmaf(u[kk], c.sc[kk], c.sh[kk]), 0.f);
          if (a.mode == 0) r[kk] += v;
          else r[kk] = (i == b) ? v : fmaxf(r[kk], v);
        }
      }
    }
    if (CANVAS) {
      if (a.mode == 0 && k > 1) {
        const float cntf = (float)k;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) r[kk] = r[kk] / cntf;
      }
      st4(op + (int64_t)cell * a.out.ld + 4 * sub, r);
      cmax = fmaxf(cmax, fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])));
    }
  }
  if (CANVAS && a.amax_out) {   // one atomic per wavefront: the bound an fp16x2 consumer of the canvas scales by (evaluation forwards)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
    const unsigned mb = __builtin_bit_cast(unsigned, cmax);      // (compare first: most wavefronts then skip the atomic)
    if (lane == 0 && mb > __atomic_load_n(a.amax_out, __ATOMIC_RELAXED)) atomicMax(a.amax_out, mb);
  }
  if (STATS) {
    bar();   // `red` aliases the point image the pillar loop above was still reading
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) red[kk * 256 + threadIdx.x] = acc[kk];
    bar();
    if (threadIdx.x < 8) {
      for (int gi = 1; gi < 32; ++gi)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc[kk] += red[kk * 256 + gi * 8 + sub];
      float* o = a.partial + (((int64_t)s * NB + band) * 32 + 4 * sub) * 2;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { o[kk * 2 + 0] = acc[kk]; o[kk * 2 + 1] = acc[4 + kk]; }
    }
  }
}

bool p2_geom(const df_pillar_geom& g, int rows_per_band, P2Geom& q) {
  if (!(g.gx > 0 && g.gy > 0 && g.gz == 1 && g.vx > 0.f && g.vy > 0.f && g.vz > 0.f)) return false;
  if (rows_per_band < 1 || (int64_t)rows_per_band * g.gx > BAND_CELLS) return false;
  q.g = g; q.R = rows_per_band; q.NB = (g.gy + rows_per_band - 1) / rows_per_band;
  return q.NB <= MAX_BANDS;
}

}  // namespace

extern "C" int df_pillar2_rows_per_band(int H, int W) {
  if (W <= 0 || H <= 0 || W > BAND_CELLS) return 0;
  int r = BAND_CELLS / W;
  while (r > 1 && (H + r - 1) / r < 8) r >>= 1;          // tiny grids: still a handful of bands
  while ((H + r - 1) / r > MAX_BANDS) ++r;
  return ((int64_t)r * W <= BAND_CELLS) ? r : 0;
}

extern "C" int df_pillar2_tile(void) { return TILE; }

extern "C" int df_pillar2_hist(const float* pts, int S, int N, df_pillar_geom g, int rows_per_band, int32_t* hist,
                               void* stream) {
  P2Geom q;
  DF_REQUIRE(pts && hist && S > 0 && N > 0, DF_E_ARG);
  DF_REQUIRE(p2_geom(g, rows_per_band, q), DF_E_SHAPE);
  const int nblk = (N + TILE - 1) / TILE;
  hipLaunchKernelGGL(p2_hist_kernel, dim3(nblk, S), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pts, N, q, nblk, hist);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_pillar2_scan(const int32_t* hist, int S, int ncol, int nblk, int32_t* off, int32_t* tot, int32_t* counts,
                               void* stream) {
  DF_REQUIRE(hist && off && tot && counts && S > 0 && ncol > 1 && nblk > 0, DF_E_ARG);
  const int total = S * ncol;
  hipLaunchKernelGGL(p2_scan_kernel, dim3((total + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), hist, total,
                     ncol, nblk, off, tot, counts);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

extern "C" int df_pillar2_scatter(const float* pts, int S, int N, df_pillar_geom g, int rows_per_band, const int32_t* off,
                                  const int32_t* tot, float* points_c, int32_t* coords_c, int64_t* idx_c, float* offs_c,
                                  int32_t* cpos, uint32_t* bkey, uint32_t* bidx, float* bpts, int32_t* bucket0,
                                  void* stream) {
  P2Geom q;
  DF_REQUIRE(pts && off && tot && points_c && coords_c && idx_c && offs_c && cpos && bkey && bidx && bpts && bucket0 && S > 0 && N > 0,
             DF_E_ARG);
  DF_REQUIRE(p2_geom(g, rows_per_band, q) && (int64_t)S * g.gx * g.gy < 0x7fffffffll && (int64_t)S * N < 0x7fffffffll, DF_E_SHAPE);
  P2Scatter a;
  a.pts = pts; a.off = off; a.tot = tot; a.points_c = points_c; a.coords_c = coords_c; a.idx_c = idx_c; a.offs_c = offs_c;
  a.cpos = cpos; a.bkey = bkey; a.bidx = bidx; a.bpts = bpts; a.bucket0 = bucket0; a.S = S; a.N = N; a.nblk = (N + TILE - 1) / TILE;
  const size_t lds = (size_t)(18 * q.NB + 16 + 8) * sizeof(int);
  const dim3 grid(a.nblk, S);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int bits = 1;
  while ((1 << bits) < q.NB) ++bits;
  if (bits <= 4) hipLaunchKernelGGL(p2_scatter_kernel<4>, grid, dim3(256), lds, st, a, q);
  else if (bits <= 7) hipLaunchKernelGGL(p2_scatter_kernel<7>, grid, dim3(256), lds, st, a, q);
  else hipLaunchKernelGGL(p2_scatter_kernel<9>, grid, dim3(256), lds, st, a, q);
  DF_CHECK_LAUNCH();
  return DF_OK;
}

static int pillar2_band_impl(const uint32_t* in_key, const uint32_t* in_idx, const float* in_pts, const int32_t* tot,
                             const int32_t* bucket0, int S, df_pillar_geom g, int rows_per_band, int flags, const float* w_pfn,
                             const float* bn_ss, int bn_sample_stride, int mode, df_img out, uint32_t* key_sorted,
                             uint32_t* idx_sorted, float* pts_sorted, int32_t* cell_rng, float* stats_partial, uint32_t* occ,
                             float* amax_out, void* stream);

extern "C" int df_pillar2_band(const uint32_t* in_key, const uint32_t* in_idx, const float* in_pts, const int32_t* tot,
                               const int32_t* bucket0, int S, df_pillar_geom g, int rows_per_band, int flags, const float* w_pfn,
                               const float* bn_ss, int bn_sample_stride, int mode, df_img out, uint32_t* key_sorted,
                               uint32_t* idx_sorted, float* pts_sorted, int32_t* cell_rng, float* stats_partial,
                               void* stream) {
  return pillar2_band_impl(in_key, in_idx, in_pts, tot, bucket0, S, g, rows_per_band, flags, w_pfn, bn_ss, bn_sample_stride, mode, out,
                           key_sorted, idx_sorted, pts_sorted, cell_rng, stats_partial, nullptr, nullptr, stream);
}

// PERSISTENT-canvas form of a canvas-writing band call (flags & 4): `out` must be zero wherever the previous call with the same
// (S, grid, rows_per_band) and the same `occ` left no pillar (initially: out all zero, occ all zero); only the occupied cells and the
// cells occupied last time are written, occ [S][bands][64] u32 is updated in place.  Same results as df_pillar2_band on such a buffer.
// amax_out (optional, zero before the call): receives max of the canvas values written (they are >= 0: the feature net ends in a ReLU).
extern "C" int df_pillar2_band_sp(const uint32_t* in_key, const uint32_t* in_idx, const float* in_pts, const int32_t* tot,
                                  const int32_t* bucket0, int S, df_pillar_geom g, int rows_per_band, int flags, const float* w_pfn,
                                  const float* bn_ss, int bn_sample_stride, int mode, df_img out, uint32_t* key_sorted,
                                  uint32_t* idx_sorted, float* pts_sorted, int32_t* cell_rng, float* stats_partial, uint32_t* occ,
                                  float* amax_out, void* stream) {
  DF_REQUIRE(occ && (flags & 4), DF_E_ARG);
  return pillar2_band_impl(in_key, in_idx, in_pts, tot, bucket0, S, g, rows_per_band, flags, w_pfn, bn_ss, bn_sample_stride, mode, out,
                           key_sorted, idx_sorted, pts_sorted, cell_rng, stats_partial, occ, amax_out, stream);
}

static int pillar2_band_impl(const uint32_t* in_key, const uint32_t* in_idx, const float* in_pts, const int32_t* tot,
                             const int32_t* bucket0, int S, df_pillar_geom g, int rows_per_band, int flags, const float* w_pfn,
                             const float* bn_ss, int bn_sample_stride, int mode, df_img out, uint32_t* key_sorted,
                             uint32_t* idx_sorted, float* pts_sorted, int32_t* cell_rng, float* stats_partial, uint32_t* occ,
                             float* amax_out, void* stream) {
  P2Geom q;
  const bool sort = flags & 1, stats = flags & 2, canvas = flags & 4;
  DF_REQUIRE(in_key && in_pts && tot && bucket0 && w_pfn && S > 0 && (mode == 0 || mode == 1), DF_E_ARG);
  DF_REQUIRE(p2_geom(g, rows_per_band, q), DF_E_SHAPE);
  DF_REQUIRE((sort && !canvas && stats) || (sort && canvas && !stats) || (!sort && canvas && !stats), DF_E_ARG);
  if (sort) DF_REQUIRE(in_idx && key_sorted && idx_sorted && pts_sorted, DF_E_ARG);
  if (stats) DF_REQUIRE(stats_partial, DF_E_ARG);
  if (canvas) {
    DF_REQUIRE(bn_ss && out.ptr, DF_E_ARG);
    DF_REQUIRE(out.n == S && out.h == g.gy && out.w == g.gx && out.c == 32 && (out.ld % 4) == 0 && df_aligned16(out.ptr), DF_E_SHAPE);
  }
  P2Band a;
  a.in_key = in_key; a.in_idx = in_idx; a.in_pts = in_pts; a.tot = tot; a.bucket0 = bucket0; a.key_sorted = key_sorted; a.idx_sorted = idx_sorted;
  a.pts_sorted = pts_sorted; a.cell_rng = cell_rng; a.w_pfn = w_pfn; a.bn_ss = bn_ss; a.partial = stats_partial; a.out = out;
  a.bn_stride = bn_sample_stride; a.mode = mode; a.S = S; a.occ = occ; a.amax_out = reinterpret_cast<unsigned*>(amax_out);
  static const int dbg = getenv("DF_P2_DBG") ? atoi(getenv("DF_P2_DBG")) : 0;
  a.dbg = dbg;
  const dim3 grid(q.NB, S);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool thin = (int64_t)q.R * g.gx <= 1024;
  if (thin) {
    if (sort && canvas) hipLaunchKernelGGL((p2_band_kernel<true, false, true, 1024>), grid, dim3(256), 0, st, a, q);
    else if (sort) hipLaunchKernelGGL((p2_band_kernel<true, true, false, 1024>), grid, dim3(256), 0, st, a, q);
    else hipLaunchKernelGGL((p2_band_kernel<false, false, true, 1024>), grid, dim3(256), 0, st, a, q);
  } else {
    if (sort && canvas) hipLaunchKernelGGL((p2_band_kernel<true, false, true>), grid, dim3(256), 0, st, a, q);
    else if (sort) hipLaunchKernelGGL((p2_band_kernel<true, true, false>), grid, dim3(256), 0, st, a, q);
    else hipLaunchKernelGGL((p2_band_kernel<false, false, true>), grid, dim3(256), 0, st, a, q);
  }
  DF_CHECK_LAUNCH();
  return DF_OK;
}
