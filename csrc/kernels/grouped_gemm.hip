// Grouped bf16 GEMM for gfx950 (CDNA4): prefill-sized matrix products on the matrix cores with the
// problem list on the DEVICE.
//
//   for every group g:  C[rows(g), :] = A[arows(g), :] . W[g]^T        W[g] is [N, K], K-contiguous
//
// rows(g) = offsets[g] .. offsets[g+1] (device memory), so a mixture-of-experts layer launches ONE
// kernel for all experts without the host ever learning how many tokens each expert got: no
// synchronisation, a launch shape that depends only on the token count, HIP-graph capturable.
// A dense projection is the same kernel with one group.
//
// Tiling (one workgroup = 512 threads = 8 waves, 2 along M x 4 along N; two waves per SIMD):
//   * 256 x 256 output tile, K consumed in steps of 64;
//   * each wave owns 128 x 64 of it: 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 registers);
//   * A and W tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers), two
//     LDS buffers of 64 KiB; the loads of step k+1 are in flight while step k is multiplied;
//   * an LDS row is 128 B (64 bf16); its eight 16-byte slots are XOR-swizzled with (row >> 1) & 7,
//     which makes every ds_read_b128 fragment read conflict-free.  A direct-to-LDS load writes LDS
//     linearly (lane x 16 B), so the swizzle is applied to the per-lane SOURCE address instead and
//     undone by the reader;
//   * per-lane source addresses also make the gather free: arow_idx[] (optional) names, per output
//     row, the row of A to read (MoE: the token of a sorted (token, expert) pair).
// Tile -> workgroup order: group by group, inside a group bands of 4 n-tiles, inside a band m-tile
// by m-tile; the linear order is cut in 8 contiguous runs, one per XCD, so the workgroups that share
// an XCD's L2 at any time share A panels and weight panels.
//
// Epilogues: plain bf16 store, or SwiGLU (PAIR): W = [gate rows | up rows] ([2 * inter, K]); a
// tile then holds 128 gate columns and the 128 matching up columns, laid out so that a lane has
// gate and up of the same element, and writes silu(gate) * up ([rows, inter]).
#include "common.h"

#define GG_BM 256
#define GG_BN 256
#define GG_BK 64
#define GG_BAND 4
#define GG_TILE_BYTES (GG_BM * GG_BK * 2)          // one operand tile: 32 KiB
#define GG_STAGE_BYTES (2 * GG_TILE_BYTES)         // A + W
#define GG_LDS_BYTES (2 * GG_STAGE_BYTES)          // two stages: 128 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void gg_glds16(const void* gptr, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (lds_ptr_t)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 gg_frag(const char* lds, int byte_off) {
  return *reinterpret_cast<const bf16x8*>(lds + byte_off);
}

// DIST: the load schedule.  0: one barrier per K-tile, the next tile's pieces go out in the first two
// k-steps.  3: a second barrier after the last fragment read of a tile frees its stage one k-step early,
// and the weight pieces of the tile after next go out there: weights (streamed from HBM when every
// group has its own) get a quarter tile more time to land.
template <bool PAIR, int DIST>
__global__ __launch_bounds__(512, 1) void grouped_gemm_kernel(
    const uint16_t* __restrict__ a, long lda, const int* __restrict__ arow_idx,
    const uint16_t* __restrict__ w, long w_group_stride, const int* __restrict__ offsets, int n_groups,
    uint16_t* __restrict__ c, long ldc, int n_total, int k_total, int n_tiles, int dbg) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which tile is mine (everything here is wave-uniform scalar work).  Tile order: group by
  // group; inside a group bands of GG_BAND n-tiles; inside a band m-tile by m-tile.  The order is
  // cut in 8 contiguous runs, one per XCD (workgroups go to XCDs round robin), so the ~32
  // workgroups that share an XCD's L2 at any time are (m-tiles of one group) x (one band): a
  // weight panel is fetched into one L2 only, once, and serves every m-tile of its group.
  int total = 0;
  for (int gi = 0; gi < n_groups; ++gi) total += ((offsets[gi + 1] - offsets[gi] + GG_BM - 1) / GG_BM) * n_tiles;
  const int bid = blockIdx.x;
  if (bid >= total) return;
  int t;
  {
    const int xcd = bid & 7, q = total >> 3, r = total & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int g = 0, row0 = 0, row_end = 0, n_idx = 0;
  for (int first = 0; g < n_groups; ++g) {
    const int lo = offsets[g], hi = offsets[g + 1];
    const int m_tiles = (hi - lo + GG_BM - 1) / GG_BM;
    const int tiles = m_tiles * n_tiles;
    if (t < first + tiles) {
      const int tl = t - first;
      const int per_band = m_tiles * GG_BAND;
      const int band = tl / per_band, rem = tl - band * per_band;
      const int width = (band + 1) * GG_BAND <= n_tiles ? GG_BAND : n_tiles - band * GG_BAND;  // last band may be narrower
      const int m_idx = rem / width;
      n_idx = band * GG_BAND + (rem - m_idx * width);
      row0 = lo + m_idx * GG_BM;
      row_end = hi;
      break;
    }
    first += tiles;
  }
  const uint16_t* wg = w + (long)g * w_group_stride;
  const int inter = n_total >> 1;  // PAIR: gate rows 0..inter-1, up rows inter..2*inter-1

  // ---- per-lane source pointers of the direct-to-LDS loads.  One wave instruction fills 8 LDS
  // rows (8 slots of 16 B each): lane -> (row = lane / 8, slot = lane % 8); wave `wave` fills rows
  // 32 * wave + 8 * i + lane / 8 (i = 0..3) of both tiles.  Slot s of LDS row r receives the
  // 16-byte chunk s ^ ((r >> 1) & 7) of the source row.
  const int lrow = lane >> 3, lslot = lane & 7;
  const uint16_t* a_src[4];
  const uint16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 32 * wave + 8 * i + lrow;  // row of the tile
    const int chunk = lslot ^ ((r >> 1) & 7);
    int arow = row0 + r;
    if (arow >= row_end) arow = row_end - 1;  // rows past the group's end repeat its last row (never stored)
    const long src_row = arow_idx ? (long)arow_idx[arow] : (long)arow;
    a_src[i] = a + src_row * lda + chunk * 8;
    int wrow;
    if (PAIR) {
      // W-tile row r belongs to the wave column block r / 64; its first 32 rows are gate rows and
      // its last 32 the matching up rows of output columns n_idx * 128 + (r / 64) * 32 + r % 32
      const int cb = r >> 6, j = r & 63;
      const int col = n_idx * 128 + cb * 32 + (j & 31);
      wrow = (j < 32 ? col : inter + col);
      if (col >= inter) wrow = (j < 32 ? inter - 1 : 2 * inter - 1);
    } else {
      wrow = n_idx * GG_BN + r;
      if (wrow >= n_total) wrow = n_total - 1;
    }
    w_src[i] = wg + (long)wrow * k_total + chunk * 8;
  }
  char* const lds_fill_a = lds + (32 * wave) * 128;                  // + stage * GG_STAGE_BYTES + i * 1024
  char* const lds_fill_w = lds + GG_TILE_BYTES + (32 * wave) * 128;

  auto issue = [&](int stage, int kt) {
    const long koff = (long)kt * GG_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gg_glds16(a_src[i] + koff, lds_fill_a + stage * GG_STAGE_BYTES + i * 1024);
      gg_glds16(w_src[i] + koff, lds_fill_w + stage * GG_STAGE_BYTES + i * 1024);
    }
  };

  // ---- fragment read offsets: wave (wr, wc) owns rows 128 * wr .. +127 of the A tile and rows
  // 64 * wc .. +63 of the W tile; MFMA operand lane -> (row = lane % 32, k half = lane / 32)
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  int a_off[4], a_sw[4], b_off[2], b_sw[2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int r = 128 * wr + 32 * mi + fr;
    a_off[mi] = r * 128;
    a_sw[mi] = (r >> 1) & 7;
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int r = 64 * wc + 32 * ni + fr;
    b_off[ni] = GG_TILE_BYTES + r * 128;
    b_sw[ni] = (r >> 1) & 7;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int k_tiles = k_total / GG_BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One K-tile = 4 k-steps of 8 MFMAs.  The stream is laid out by hand and pinned with scheduling
  // fences: every LDS read and every direct-to-LDS load is issued in the shadow of an MFMA (an
  // MFMA occupies the matrix pipe for 32 cycles and the issue port for 8), the fragments of k-step
  // s+1 are read while k-step s multiplies, and all loads of the next tile go out in the first two
  // k-steps so that they have the other two to land.
#define GG_FENCE() __builtin_amdgcn_sched_barrier(0)
#define GG_MFMA(MI, NI) \
  acc[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][MI], bfr[cur][NI], acc[MI][NI], 0, 0, 0)
  bf16x8 af[2][4], bfr[2][2];
  auto read_a = [&](int set, const char* buf, int ks, int mi) {
    af[set][mi] = gg_frag(buf, a_off[mi] + (((2 * ks + fh) ^ a_sw[mi]) << 4));
  };
  auto read_b = [&](int set, const char* buf, int ks, int ni) {
    bfr[set][ni] = gg_frag(buf, b_off[ni] + (((2 * ks + fh) ^ b_sw[ni]) << 4));
  };
  if (DIST == 3) {
    // early-free schedule: the weight pieces run two tiles ahead; tile 1's go out before the loop
    const long k1 = (long)(k_tiles > 1 ? 1 : 0) * GG_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) gg_glds16(w_src[i] + k1, lds_fill_w + GG_STAGE_BYTES + i * 1024);
  }
  {
    const char* buf0 = lds;
    read_b(0, buf0, 0, 0); read_b(0, buf0, 0, 1);
    read_a(0, buf0, 0, 0); read_a(0, buf0, 0, 1); read_a(0, buf0, 0, 2); read_a(0, buf0, 0, 3);
  }
  for (int kt = 0; kt < k_tiles; ++kt) {
    const int stage = kt & 1;
    const char* buf = lds + stage * GG_STAGE_BYTES;
    // the last tile re-requests itself into the idle stage: no branch in the stream, nothing reads it
    const long koff = (long)(kt + 1 < k_tiles ? kt + 1 : kt) * GG_BK;
    char* const fill_a = lds_fill_a + (stage ^ 1) * GG_STAGE_BYTES;
    char* const fill_w = lds_fill_w + (stage ^ 1) * GG_STAGE_BYTES;
    // Sixteen issue slots per K-tile (four per k-step, each behind an MFMA).  DIST 0 (one barrier per
    // K-tile): the eight pieces of tile kt+1 in the first eight slots.  DIST 3 (two barriers): below.
    const long koff2 = (long)(kt + 2 < k_tiles ? kt + 2 : k_tiles - 1) * GG_BK;
    auto load_at = [&](int slot) {
      if (DIST == 3) {
        // k-step 0: the four A pieces of tile kt+1 (other stage); k-step 3, after the barrier that
        // frees THIS stage: the four weight pieces of tile kt+2
        if (dbg & 1) return;
        if (slot < 4) gg_glds16(a_src[slot] + koff, fill_a + slot * 1024);
        if (slot >= 12) gg_glds16(w_src[slot - 12] + koff2, lds_fill_w + stage * GG_STAGE_BYTES + (slot - 12) * 1024);
        return;
      }
      const int j = slot < 4 ? 2 * slot + 1 : slot < 8 ? 2 * (slot - 4) : -1;  // weights first, then A
      if (j < 0 || (dbg & 1)) return;
      if (j & 1)
        gg_glds16(w_src[j >> 1] + koff, fill_w + (j >> 1) * 1024);
      else
        gg_glds16(a_src[j >> 1] + koff, fill_a + (j >> 1) * 1024);
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      GG_FENCE();
      if (DIST == 3 && ks == 3 && !(dbg & 2)) {
        // my fragment reads of this stage are done (k-step 3's were issued during k-step 2); once
        // everybody's are, the stage can take the weights of tile kt+2 while k-step 3 multiplies
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        GG_FENCE();
      }
      GG_MFMA(0, 0); GG_FENCE();
      load_at(4 * ks + 0);
      GG_FENCE();
      GG_MFMA(0, 1); GG_FENCE();
      if (ks < 3) read_b(nxt, buf, ks + 1, 0);
      GG_FENCE();
      GG_MFMA(1, 0); GG_FENCE();
      load_at(4 * ks + 1);
      GG_FENCE();
      GG_MFMA(1, 1); GG_FENCE();
      if (ks < 3) read_b(nxt, buf, ks + 1, 1);
      if (ks < 3) read_a(nxt, buf, ks + 1, 0);
      GG_FENCE();
      GG_MFMA(2, 0); GG_FENCE();
      load_at(4 * ks + 2);
      GG_FENCE();
      GG_MFMA(2, 1); GG_FENCE();
      if (ks < 3) read_a(nxt, buf, ks + 1, 1);
      GG_FENCE();
      GG_MFMA(3, 0); GG_FENCE();
      load_at(4 * ks + 3);
      if (ks < 3) read_a(nxt, buf, ks + 1, 2);
      GG_FENCE();
      if (ks < 3) {
        GG_MFMA(3, 1); GG_FENCE();
        read_a(nxt, buf, ks + 1, 3);
      } else {
        // the next stage has landed (mine: vmcnt; everyone's: the barrier) and nobody still reads
        // this one; the last MFMA of the tile covers the first fragment reads of the next
        if (!(dbg & 2)) {
          if (DIST == 3)  // tile kt+1 has landed; the four weight pieces of tile kt+2 stay in flight
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        GG_FENCE();
        const char* nbuf = lds + (stage ^ 1) * GG_STAGE_BYTES;
        read_b(nxt, nbuf, 0, 0); read_b(nxt, nbuf, 0, 1); read_a(nxt, nbuf, 0, 0); read_a(nxt, nbuf, 0, 1);
        GG_FENCE();
        GG_MFMA(3, 1); GG_FENCE();
        read_a(nxt, nbuf, 0, 2); read_a(nxt, nbuf, 0, 3);
      }
    }
  }
#undef GG_MFMA
#undef GG_FENCE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the wave ends

  // ---- epilogue.  Accumulator of tile (mi, ni): lane -> column lane % 32, register e -> row
  // 8 * (e / 4) + 4 * (lane / 32) + e % 4.
  if (PAIR) {
    // ni = 0 is gate, ni = 1 the matching up columns: silu(gate) * up, 32 output columns per wave
    const int col = n_idx * 128 + wc * 32 + fr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + 128 * wr + 32 * mi + 8 * (e >> 2) + 4 * fh + (e & 3);
        const float gt = acc[mi][0][e], up = acc[mi][1][e];
        const float v = gt / (1.f + __expf(-gt)) * up;
        if (row < row_end && col < inter) {
          bf16_t o = (bf16_t)v;
          c[(long)row * ldc + col] = __builtin_bit_cast(uint16_t, o);
        }
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = n_idx * GG_BN + 64 * wc + 32 * ni + fr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + 128 * wr + 32 * mi + 8 * (e >> 2) + 4 * fh + (e & 3);
          if (row < row_end && col < n_total) {
            bf16_t o = (bf16_t)acc[mi][ni][e];
            c[(long)row * ldc + col] = __builtin_bit_cast(uint16_t, o);
          }
        }
      }
    }
  }
}

static int g_gg_debug = 0;
// bit 2 selects the one-barrier schedule (A/B timing; results are the same).  Bits 0 and 1 are timing
// experiments with WRONG results: bit 0 skips the in-loop loads, bit 1 the in-loop waits and barriers
VPSA_EXPORT int vpsa_grouped_gemm_debug(int v) {
  int old = g_gg_debug;
  g_gg_debug = v;
  return old;
}

// a [*, K] bf16 (row stride lda elements, 16-byte aligned rows); arow_idx [rows] int32 or null
// (null: output row r reads A row r); w [G, N, K] contiguous; offsets [G + 1] int32 on the device,
// offsets[G] <= max_rows; c [max_rows, N] (pair: [max_rows, N / 2]), row stride ldc.
// Returns 0, or -1 when the shape is outside what the kernel tiles.
VPSA_EXPORT int vpsa_grouped_gemm(const void* a, long lda, const void* arow_idx, const void* w, const void* offsets,
                                  int n_groups, void* c, long ldc, int max_rows, int n, int k, int pair,
                                  hipStream_t stream) {
  if (k % GG_BK != 0 || k <= 0 || n <= 0 || max_rows <= 0 || n_groups <= 0 || lda % 8 != 0) return -1;
  if (pair && (n % 2 != 0 || (n / 2) % 32 != 0)) return -1;
  const int n_tiles = pair ? (n / 2 + 127) / 128 : (n + GG_BN - 1) / GG_BN;
  const int m_tiles_max = max_rows / GG_BM + n_groups;  // >= sum of per-group ceilings
  const long blocks = (long)n_tiles * m_tiles_max;
  if (blocks > 0x7fffffffL) return -1;
  const long wstride = (long)n * k;
  const bool simple = (g_gg_debug & 4) != 0;
#define GG_LAUNCH(P, D)                                                                                         \
  do {                                                                                                          \
    static bool attr_set = false;                                                                               \
    if (!attr_set) {                                                                                            \
      if (hipFuncSetAttribute((const void*)grouped_gemm_kernel<P, D>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              GG_LDS_BYTES) != hipSuccess)                                                      \
        return -2;                                                                                              \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    grouped_gemm_kernel<P, D><<<dim3((unsigned)blocks), dim3(512), GG_LDS_BYTES, stream>>>(                      \
        (const uint16_t*)a, lda, (const int*)arow_idx, (const uint16_t*)w, wstride, (const int*)offsets, n_groups, \
        (uint16_t*)c, ldc, n, k, n_tiles, g_gg_debug);                                                          \
  } while (0)
  if (pair) {
    if (simple) GG_LAUNCH(true, 0);
    else GG_LAUNCH(true, 3);
  } else {
    if (simple) GG_LAUNCH(false, 0);
    else GG_LAUNCH(false, 3);
  }
#undef GG_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
