// Prefill-sized bf16 GEMM for gfx950 (CDNA4), dense or grouped, with fused epilogues.
//
//   for every group g:  C[rows(g), :] = epilogue( A[arows(g), :] . W[g]^T )      W[g] is [N, K], K-contiguous
//
// rows(g) = offsets[g] .. offsets[g+1] read on the DEVICE (a mixture-of-experts layer is one launch
// whatever the routing, no host synchronisation, HIP-graph capturable); offsets == null is one group
// of max_rows rows (a dense projection).
//
// Structure: 256 x 256 output tile per workgroup, K in steps of 64, 512 threads = 8 waves as
// 2 (rows) x 4 (columns), each wave 128 x 64 of the tile in 32 accumulators of
// v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on this shape than on 32x32x16 for the
// same FLOPs).  The two wave rows run half a phase apart ("ping-pong"): the SIMD's two waves are
// one of each row, so while one multiplies 16 MFMAs the other reads its next fragments from LDS and
// issues the direct-to-LDS loads of later K tiles; every phase ends in a workgroup barrier, so the
// alternation is exact.  A K tile is four phases (one 64 x 32 quadrant of the wave's block each);
// its operand tiles are cut in four half tiles of 16 KiB (rows of the first / second quadrant row,
// columns of the first / second quadrant column) so that a half tile is free for re-use as early as
// one phase after the only phase that reads it; one half tile is requested per phase, three are in
// flight behind a COUNTED vmcnt at the end of each K tile (never 0 inside the loop), two LDS buffers
// of 64 KiB.
//
// The MFMA takes the weight rows as its A operand and the activation rows as B, so an accumulator
// holds C transposed: lane = one output row, registers = consecutive output columns.  The weight
// rows of a 32-column quadrant are ordered in LDS so that the two accumulators of a lane are 8
// consecutive columns: the epilogue writes 16 bytes per lane with no shuffle.
//
// LDS image of a half tile: 128 rows of 128 B, the eight 16-byte slots of a row XOR-swizzled with
// (row >> 1) & 7: conflict-free for the 16-row x 64-byte fragment reads (ds_read_b128).  Direct-to-
// LDS loads write LDS linearly (lane x 16 B), so the swizzle is applied to the per-lane SOURCE
// address; per-lane sources also make the row gather (arow_idx) free.
//
// Epilogues: bf16 store; SwiGLU (W = [gate rows | up rows], the first quadrant column of a wave is
// gate and the second the matching up columns: silu(gate) * up, [rows, N / 2]); bf16 store of
// accumulator + residual (the residual stream update of the o / down projections).
#include "common.h"

#define TG_BM 256
#define TG_BN 256
#define TG_BK 64
#define TG_BAND 4
#define TG_HT 16384          // half tile
#define TG_BUF (4 * TG_HT)   // X-h0, X-h1, W-h0, W-h1
#define TG_LDS (2 * TG_BUF)

#define TG_EPI_PLAIN 0
#define TG_EPI_SWIGLU 1
#define TG_EPI_RESIDUAL 2

typedef __attribute__((address_space(3))) void* tg_lds_ptr_t;

__device__ __forceinline__ void tg_glds16(const char* gptr, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (tg_lds_ptr_t)lds_dst, 16, 0, 0);
}

// DBG: timing experiments with WRONG results (bit 0: no loads inside the K loop, bit 1: no fragment
// reads inside the K loop, bit 2: every load inside the K loop re-requests K tile 0, i.e. hits in
// cache); 0 in every production launch.
template <int EPI, int DBG>
__global__ __launch_bounds__(512, 2) void tile_gemm_kernel(
    const uint16_t* __restrict__ a, long lda, const int* __restrict__ arow_idx,
    const uint16_t* __restrict__ w, long w_group_stride, const int* __restrict__ offsets, int n_groups, int max_rows,
    uint16_t* __restrict__ c, long ldc, const uint16_t* __restrict__ resid, long ldr,
    int n_total, int k_total, int n_tiles) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which tile is mine (wave-uniform scalar work).  Tile order: group by group; inside a group
  // bands of TG_BAND column tiles; inside a band row tile by row tile.  The order is cut in 8
  // contiguous runs, one per XCD (workgroups go to XCDs round robin), so the workgroups that share
  // an XCD's L2 at any time are (row tiles of one group) x (one band).
  int total = 0;
  if (offsets) {
    for (int gi = 0; gi < n_groups; ++gi) total += ((offsets[gi + 1] - offsets[gi] + TG_BM - 1) / TG_BM) * n_tiles;
  } else {
    total = ((max_rows + TG_BM - 1) / TG_BM) * n_tiles;
  }
  const int bid = blockIdx.x;
  if (bid >= total) return;
  int t;
  {
    const int xcd = bid & 7, q = total >> 3, r = total & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int g = 0, row0 = 0, row_end = 0, n_idx = 0;
  for (int first = 0; g < n_groups; ++g) {
    const int lo = offsets ? offsets[g] : 0, hi = offsets ? offsets[g + 1] : max_rows;
    const int m_tiles = (hi - lo + TG_BM - 1) / TG_BM;
    const int tiles = m_tiles * n_tiles;
    if (t < first + tiles) {
      const int tl = t - first;
      const int per_band = m_tiles * TG_BAND;
      const int band = tl / per_band, rem = tl - band * per_band;
      const int width = (band + 1) * TG_BAND <= n_tiles ? TG_BAND : n_tiles - band * TG_BAND;
      const int m_idx = rem / width;
      n_idx = band * TG_BAND + (rem - m_idx * width);
      row0 = lo + m_idx * TG_BM;
      row_end = hi;
      break;
    }
    first += tiles;
  }
  const char* const a_base = reinterpret_cast<const char*>(a);
  const char* const w_base = reinterpret_cast<const char*>(w + (long)g * w_group_stride);
  const int inter = n_total >> 1;  // SwiGLU: gate rows 0..inter-1, up rows inter..2*inter-1

  const int wr = wave >> 2, wc = wave & 3;

  // ---- per-lane source offsets (bytes) of the direct-to-LDS loads.  One wave instruction fills 8
  // LDS rows: lane -> (row = lane / 8, slot = lane % 8); instruction j (0, 1) of a half tile: wave
  // `wave` fills rows 64 j + 8 wave + lane / 8.  Slot s of LDS row r receives the 16-byte chunk
  // s ^ ((r >> 1) & 7) of the source row.
  //   X half tile h, LDS row r:  tile row 128 (r / 64) + 64 h + r % 64
  //   W half tile h, LDS row r:  wave column block r / 32, MFMA tile (r % 32) / 16, MFMA row i = r % 16
  //                              -> column 64 (r / 32) + 32 h + 8 (i / 4) + 4 tile + i % 4
  uint32_t src_off[4][2];
  {
    const int lrow = lane >> 3, lslot = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 64 * j + 8 * wave + lrow;
      const int chunk = lslot ^ ((r >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int arow = row0 + 128 * (r >> 6) + 64 * h + (r & 63);
        if (arow >= row_end) arow = row_end - 1;  // rows past the group's end repeat its last row (never stored)
        const long src_row = arow_idx ? (long)arow_idx[arow] : (long)arow;
        src_off[h][j] = (uint32_t)((src_row * lda + chunk * 8) * 2);
        const int cb = r >> 5, ti = (r >> 4) & 1, i = r & 15;
        const int nloc = 8 * (i >> 2) + 4 * ti + (i & 3);
        int wrow;
        if (EPI == TG_EPI_SWIGLU) {
          int col = n_idx * 128 + cb * 32 + nloc;
          if (col >= inter) col = inter - 1;
          wrow = h == 0 ? col : inter + col;
        } else {
          wrow = n_idx * TG_BN + cb * 64 + 32 * h + nloc;
          if (wrow >= n_total) wrow = n_total - 1;
        }
        src_off[2 + h][j] = (uint32_t)(((long)wrow * k_total + chunk * 8) * 2);
      }
    }
  }
  const int k_tiles = k_total / TG_BK;
  char* const fill_base = lds + wave * 1024;
  bool in_loop = false;
  auto stage = [&](int which, int buf, int kt) {
    if ((DBG & 1) && in_loop) return;
    int ktc = kt < k_tiles ? kt : k_tiles - 1;  // past the end: re-request the last tile into a free slot
    if ((DBG & 4) && in_loop) ktc = 0;
    const char* base = (which < 2 ? a_base : w_base) + (long)ktc * (TG_BK * 2);
    tg_glds16(base + src_off[which][0], fill_base + buf * TG_BUF + which * TG_HT);
    tg_glds16(base + src_off[which][1], fill_base + buf * TG_BUF + which * TG_HT + 8192);
  };

  // ---- fragment read offsets.  MFMA operand lane -> (row = lane % 16, k quarter = lane / 16); all
  // fragment rows start at a multiple of 16, so the row's swizzle is (lane % 16) / 2
  const int fr = lane & 15, fq = lane >> 4;
  const int sl0 = ((fq) ^ (fr >> 1)) << 4, sl1 = ((4 + fq) ^ (fr >> 1)) << 4;
  const int x_rd = (wr * 64 + fr) * 128;
  const int w_rd = 2 * TG_HT + (wc * 32 + fr) * 128;

  f32x4 acc[2][2][4][2];  // [quadrant row][quadrant column][row tile][column tile]
#pragma unroll
  for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
    for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int i3 = 0; i3 < 2; ++i3) acc[i0][i1][i2][i3] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xa[4][2];      // activation fragments of the current quadrant row: [row tile][k step]
  bf16x8 wb[2][2][2];   // weight fragments of both quadrant columns: [quadrant column][column tile][k step]

  auto read_w = [&](int buf, int nh) {
    if ((DBG & 2) && in_loop) return;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      wb[nh][ni][0] = *reinterpret_cast<const bf16x8*>(lds + buf * TG_BUF + nh * TG_HT + ni * 2048 + w_rd + sl0);
      wb[nh][ni][1] = *reinterpret_cast<const bf16x8*>(lds + buf * TG_BUF + nh * TG_HT + ni * 2048 + w_rd + sl1);
    }
  };
  auto read_x = [&](int buf, int mh) {
    if ((DBG & 2) && in_loop) return;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      xa[mi][0] = *reinterpret_cast<const bf16x8*>(lds + buf * TG_BUF + mh * TG_HT + mi * 2048 + x_rd + sl0);
      xa[mi][1] = *reinterpret_cast<const bf16x8*>(lds + buf * TG_BUF + mh * TG_HT + mi * 2048 + x_rd + sl1);
    }
  };
  auto mma = [&](int mh, int nh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mh][nh][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[nh][ni][s], xa[mi][s], acc[mh][nh][mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

#define TG_SB() __builtin_amdgcn_sched_barrier(0)
#define TG_BAR()                    \
  do {                              \
    TG_SB();                        \
    __builtin_amdgcn_s_barrier();   \
    TG_SB();                        \
  } while (0)
#define TG_LGKM0()                                        \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    TG_SB();                                              \
  } while (0)

  // ---- prologue: K tile 0 whole, then the first three half tiles of K tile 1 (the order the loop
  // keeps: W-h0, X-h0, W-h1; X-h1 follows in the first phase)
  stage(0, 0, 0);
  stage(1, 0, 0);
  stage(2, 0, 0);
  stage(3, 0, 0);
  stage(2, 1, 1);
  stage(0, 1, 1);
  stage(3, 1, 1);
  stage(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  TG_BAR();
  if (wr == 1) TG_BAR();  // the second wave row runs one barrier behind the first

  // One K tile (LDS buffer B, tile index KT).  Phase = {fragment reads, one half-tile request,
  // barrier, 16 MFMAs, barrier}.  Requests: X-h1 of tile KT+1 (its slot was last read two phases
  // ago), then W-h0, X-h0, W-h1 of tile KT+2 into THIS buffer: W-h0 is read in phase 1 only and
  // those reads are retired (lgkmcnt(8): the four W reads are issued first) before phase 1's first
  // barrier, so it may be overwritten from phase 2 on; X-h0 (read in phase 1) from phase 3; W-h1
  // (phase 2) from phase 4; X-h1 (phase 3) from the next tile's phase 1.  The counted wait of phase
  // 4 leaves the three youngest half tiles in flight: everything of tile KT+1 has landed before the
  // barrier that precedes its first read.
#define TG_TILE(B, KT)                                          \
  do {                                                          \
    read_w(B, 0);                                               \
    TG_SB();                                                    \
    read_x(B, 0);                                               \
    TG_SB();                                                    \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          \
    TG_BAR();                                                   \
    TG_LGKM0();                                                 \
    mma(0, 0);                                                  \
    TG_BAR();                                                   \
    read_w(B, 1);                                               \
    TG_SB();                                                    \
    stage(2, B, (KT) + 2);                                      \
    TG_BAR();                                                   \
    TG_LGKM0();                                                 \
    mma(0, 1);                                                  \
    TG_BAR();                                                   \
    read_x(B, 1);                                               \
    TG_SB();                                                    \
    stage(0, B, (KT) + 2);                                      \
    TG_SB();                                                    \
    TG_LGKM0();                                                 \
    TG_BAR();                                                   \
    mma(1, 1);                                                  \
    TG_BAR();                                                   \
    stage(3, B, (KT) + 2);                                      \
    stage(1, B, (KT) + 2);                                      \
    TG_SB();                                                    \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            \
    TG_BAR();                                                   \
    mma(1, 0);                                                  \
    TG_BAR();                                                   \
  } while (0)

  if (DBG) {
    read_w(0, 0); read_w(0, 1); read_x(0, 0);
    in_loop = true;
  }
  int kt = 0;
  for (; kt + 1 < k_tiles; kt += 2) {
    TG_TILE(0, kt);
    TG_TILE(1, kt + 1);
  }
  if (kt < k_tiles) TG_TILE(0, kt);
#undef TG_TILE
  if (wr == 0) TG_BAR();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the wave ends

  // ---- epilogue.  Accumulator [mh][nh][mi][ni]: lane -> output row 128 wr + 64 mh + 16 mi + lane % 16,
  // register e -> column 64 wc + 32 nh + 8 (lane / 16) + 4 ni + e.
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = row0 + 128 * wr + 64 * mh + 16 * mi + fr;
      if (EPI == TG_EPI_SWIGLU) {
        const int col = n_idx * 128 + wc * 32 + 8 * fq;
        float v[8];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gt = acc[mh][0][mi][ni][e], up = acc[mh][1][mi][ni][e];
            v[4 * ni + e] = gt / (1.f + __expf(-gt)) * up;
          }
        if (row < row_end && col < inter) {
          u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
          *reinterpret_cast<u32x4*>(c + (long)row * ldc + col) = o;
        }
      } else {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const int col = n_idx * TG_BN + wc * 64 + 32 * nh + 8 * fq;
          if (row < row_end && col < n_total) {
            float v[8];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * ni + e] = acc[mh][nh][mi][ni][e];
            if (EPI == TG_EPI_RESIDUAL) {
              const u32x4 rr = *reinterpret_cast<const u32x4*>(resid + (long)row * ldr + col);
#pragma unroll
              for (int p = 0; p < 4; ++p) {
                v[2 * p] += bf16lo(rr[p]);
                v[2 * p + 1] += bf16hi(rr[p]);
              }
            }
            u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            *reinterpret_cast<u32x4*>(c + (long)row * ldc + col) = o;
          }
        }
      }
    }
  }
}

static int g_tg_debug = 0;
VPSA_EXPORT int vpsa_tile_gemm_debug(int v) {
  int old = g_tg_debug;
  g_tg_debug = v;
  return old;
}

// a [*, K] bf16 (row stride lda elements, 16-byte aligned rows); arow_idx [rows] int32 or null (null:
// output row r reads A row r); w [G, N, K] contiguous; offsets [G + 1] int32 on the device with
// offsets[G] <= max_rows, or null (one group of max_rows rows); c [max_rows, N] (SwiGLU: N / 2
// columns), row stride ldc; epi: 0 store, 1 SwiGLU, 2 store of product + resid (row stride ldr; may
// alias c).  Every byte offset into a and into one group of w must fit 32 bits.
// Returns 0, or a negative number when the shape is outside what the kernel tiles.
VPSA_EXPORT int vpsa_tile_gemm(const void* a, long lda, const void* arow_idx, const void* w, const void* offsets,
                               int n_groups, void* c, long ldc, int max_rows, int n, int k, int epi, const void* resid,
                               long ldr, long a_rows, hipStream_t stream) {
  if (k % TG_BK != 0 || k <= 0 || n <= 0 || max_rows <= 0 || n_groups <= 0 || lda % 8 != 0 || ldc % 8 != 0) return -1;
  if (!offsets && n_groups != 1) return -1;
  if (epi == TG_EPI_SWIGLU && (n % 2 != 0 || (n / 2) % 32 != 0)) return -1;
  if (epi != TG_EPI_SWIGLU && n % 8 != 0) return -1;
  if (epi == TG_EPI_RESIDUAL && (!resid || ldr % 8 != 0)) return -1;
  if (epi < 0 || epi > 2) return -1;
  if ((double)a_rows * (double)lda * 2.0 >= 4294967296.0 || (double)n * (double)k * 2.0 >= 4294967296.0) return -1;
  const int n_tiles = epi == TG_EPI_SWIGLU ? (n / 2 + 127) / 128 : (n + TG_BN - 1) / TG_BN;
  const int m_tiles_max = offsets ? max_rows / TG_BM + n_groups : (max_rows + TG_BM - 1) / TG_BM;
  const long blocks = (long)n_tiles * m_tiles_max;
  if (blocks > 0x7fffffffL) return -1;
  const long wstride = (long)n * k;
#define TG_LAUNCH(E, D)                                                                                               \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      if (hipFuncSetAttribute((const void*)tile_gemm_kernel<E, D>, hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS) != \
          hipSuccess)                                                                                                \
        return -2;                                                                                                   \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    tile_gemm_kernel<E, D><<<dim3((unsigned)blocks), dim3(512), TG_LDS, stream>>>(                                       \
        (const uint16_t*)a, lda, (const int*)arow_idx, (const uint16_t*)w, wstride, (const int*)offsets, n_groups,   \
        max_rows, (uint16_t*)c, ldc, (const uint16_t*)resid, ldr, n, k, n_tiles);                                    \
  } while (0)
  if (epi == TG_EPI_SWIGLU) TG_LAUNCH(TG_EPI_SWIGLU, 0);
  else if (epi == TG_EPI_RESIDUAL) TG_LAUNCH(TG_EPI_RESIDUAL, 0);
  else if (g_tg_debug == 1) TG_LAUNCH(TG_EPI_PLAIN, 1);
  else if (g_tg_debug == 2) TG_LAUNCH(TG_EPI_PLAIN, 2);
  else if (g_tg_debug == 3) TG_LAUNCH(TG_EPI_PLAIN, 3);
  else if (g_tg_debug == 4) TG_LAUNCH(TG_EPI_PLAIN, 4);
  else TG_LAUNCH(TG_EPI_PLAIN, 0);
#undef TG_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
