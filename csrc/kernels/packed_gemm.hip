// Decode-step projections over PRE-PACKED weights (gfx950).
//
// A decode step multiplies up to 64 token rows with every weight matrix of the model: the work is
// reading the weights once.  A row-major [N, K] matrix does not suit the MFMA A-operand layout
// (16 rows x 64 bytes per load instruction: half of each of 16 lines, the other halves re-miss;
// going through LDS fixes the lines but costs a second trip for every weight byte).  Here the
// weights are packed once, at load time, into the order the matrix cores read them:
//
//     packed[n / 16][k / 32][n % 16][k % 32]          one 16 x 32 tile = 1 KiB, contiguous
//
// so a wave's load instruction (lane l: row l & 15, 8 elements from 8 * (l >> 4)) is one
// contiguous KiB that lands directly in the A operand, and a wave streaming along K for its 16
// weight rows reads one contiguous run of K * 32 bytes.  No LDS for weights.
//
//   * workgroup = 4 waves = 64 weight rows (PAIR: 64 gate rows + the 64 matching up rows, fused
//     SwiGLU), one K slice of the reduction (grid.y slices; fp32 partials summed in a fixed
//     order by stream_reduce-style kernel below: bitwise reproducible, no atomics);
//   * weights: a ring of DEPTH chunks (128 elements of K each = 4 loads per lane and tile) in
//     registers, issued DEPTH - 1 chunks ahead of the MFMAs that consume them: 12 KiB per tile and
//     wave in flight without depending on occupancy;
//   * token rows: the chunk's [rows, 128] slice goes through a double-buffered LDS tile shared by
//     the four waves (one barrier per chunk), read back as MFMA B operands (padded rows,
//     conflict-free 16-byte reads).
// The split count depends on the matrix only, never on the batch: a row's result is the same bits
// whatever shares its step.
#include "common.h"

#define PK_KC 128                 // elements of K per chunk
#define PK_RS (PK_KC * 2 + 16)    // LDS row stride in bytes

template <int TT, bool PAIR, int DEPTH>
__global__ __launch_bounds__(256) void packed_gemm_kernel(
    const uint16_t* __restrict__ x, long x_stride, const uint16_t* __restrict__ wp,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, long y_stride,
    float* __restrict__ ws, int M, int N, int K, int krange) {
  constexpr int NA = PAIR ? 2 : 1;
  __shared__ __attribute__((aligned(16))) uint8_t lds_x[2][TT * 16 * PK_RS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const int n_out = PAIR ? (N >> 1) : N;
  const int nb = blockIdx.x * 4 + wave;          // 16-row block of this wave
  const long k0 = (long)blockIdx.y * krange;
  const int n_chunks = krange / PK_KC;
  // packed tile (nb, kb) starts at ((nb * (K / 32)) + kb) * 512 elements; this lane's 8 elements
  // (row c16, reduction elements 8g..8g+7) are at c16 * 32 + 8 * g inside it
  const uint16_t* wbase[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
    wbase[a] = wp + ((long)(nb + a * (n_out >> 4)) * (K >> 5) + (k0 >> 5)) * 512 + c16 * 32 + g * 8;
  // token rows: thread covers row tid / 16 (+16 per pass), 8 elements from (tid % 16) * 8
  const int xr = tid >> 4, xc = (tid & 15) * 8;
  const uint16_t* xb[TT];
#pragma unroll
  for (int p = 0; p < TT; ++p) xb[p] = x + (long)min(xr + 16 * p, M - 1) * x_stride + k0 + xc;

  static_assert(DEPTH % 2 == 0, "the token-row registers alternate with the chunk parity");
  u32x4 wr[DEPTH][NA][4];
  u32x4 xreg[2][TT];  // token-row slices of chunk c live in xreg[c & 1] until they are stored to LDS
  auto load_w = [&](int slot, int chunk) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        wr[slot][a][u] = __builtin_nontemporal_load(
            reinterpret_cast<const u32x4*>(wbase[a] + ((long)chunk * 4 + u) * 512));
  };
  auto load_x = [&](int slot, int chunk) {
#pragma unroll
    for (int p = 0; p < TT; ++p) xreg[slot][p] = *reinterpret_cast<const u32x4*>(xb[p] + chunk * PK_KC);
  };
  auto store_x = [&](int slot) {
#pragma unroll
    for (int p = 0; p < TT; ++p)
      *reinterpret_cast<u32x4*>(&lds_x[slot][(xr + 16 * p) * PK_RS + xc * 2]) = xreg[slot][p];
  };

  f32x4 acc[NA][TT];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[a][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Vector-memory operations complete in issue order, so waiting for a token-row load also waits
  // for every weight load issued before it.  Token rows are therefore fetched TWO chunks ahead and
  // always before the iteration's weight load: the wait in front of the LDS store of chunk c + 1
  // only covers weight chunks that the next iteration needs anyway, and DEPTH - 2 chunks of
  // weights stay in flight across it.
  load_x(0, 0);
  if (1 < n_chunks) load_x(1, 1);
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (d < n_chunks) load_w(d, d);
  store_x(0);
  __syncthreads();

  for (int c0 = 0; c0 < n_chunks; c0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int c = c0 + d;
      if (c < n_chunks) {  // workgroup-uniform
        if (c + 2 < n_chunks) load_x(d & 1, c + 2);       // slot of chunk c: stored a step ago
        const int ahead = c + DEPTH - 1;
        if (ahead < n_chunks) load_w((d + DEPTH - 1) % DEPTH, ahead);
        const uint8_t* xbuf = lds_x[d & 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            const u32x4 bf = *reinterpret_cast<const u32x4*>(xbuf + (tt * 16 + c16) * PK_RS + (u * 32 + 8 * g) * 2);
#pragma unroll
            for (int a = 0; a < NA; ++a)
              acc[a][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, wr[d][a][u]), __builtin_bit_cast(bf16x8, bf), acc[a][tt], 0, 0, 0);
          }
        }
        if (c + 1 < n_chunks) store_x((d + 1) & 1);
        __syncthreads();
      }
    }
  }

  // D layout of a 16x16 tile: lane (c16, g) holds weight rows 4g..4g+3 of token c16
  const int n0 = nb * 16;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int j = tt * 16 + c16;
    if (j >= M) continue;
    if (ws != nullptr) {
      float* p = ws + ((long)blockIdx.y * M + j) * N + n0 + 4 * g;
      *reinterpret_cast<f32x4*>(p) = acc[0][tt];
      if (PAIR) *reinterpret_cast<f32x4*>(p + n_out) = acc[NA - 1][tt];
    } else {
      f32x4 s = acc[0][tt];
      if (PAIR) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = s[r] / (1.f + __expf(-s[r])) * acc[NA - 1][tt][r];
      } else if (bias != nullptr) {
        const u32x2 bb = *reinterpret_cast<const u32x2*>(bias + n0 + 4 * g);
        s[0] += bf16lo(bb[0]);
        s[1] += bf16hi(bb[0]);
        s[2] += bf16lo(bb[1]);
        s[3] += bf16hi(bb[1]);
      }
      u32x2 o;
      o[0] = pack_bf16x2(s[0], s[1]);
      o[1] = pack_bf16x2(s[2], s[3]);
      *reinterpret_cast<u32x2*>(y + (long)j * y_stride + n0 + 4 * g) = o;
    }
  }
}

// out[m, n..n+3] = epilogue(sum_s ws[s, m, n..n+3]) in slice order; one thread per 4 columns.
template <bool PAIR>
__global__ __launch_bounds__(256) void packed_reduce_kernel(const float* __restrict__ ws,
                                                            const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ y, long y_stride,
                                                            int M, int N, int S) {
  const int n_out = PAIR ? (N >> 1) : N;
  const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int m = blockIdx.y;
  if (n >= n_out) return;
  const float* p = ws + (long)m * N + n;
  f32x4 s = *reinterpret_cast<const f32x4*>(p);
  f32x4 s2 = PAIR ? *reinterpret_cast<const f32x4*>(p + n_out) : (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 1; i < S; ++i) {
    const float* q = p + (long)i * M * N;
    s += *reinterpret_cast<const f32x4*>(q);
    if (PAIR) s2 += *reinterpret_cast<const f32x4*>(q + n_out);
  }
  if (PAIR) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = s[r] / (1.f + __expf(-s[r])) * s2[r];
  } else if (bias != nullptr) {
    const u32x2 bb = *reinterpret_cast<const u32x2*>(bias + n);
    s[0] += bf16lo(bb[0]);
    s[1] += bf16hi(bb[0]);
    s[2] += bf16lo(bb[1]);
    s[3] += bf16hi(bb[1]);
  }
  u32x2 o;
  o[0] = pack_bf16x2(s[0], s[1]);
  o[1] = pack_bf16x2(s[2], s[3]);
  *reinterpret_cast<u32x2*>(y + (long)m * y_stride + n) = o;
}

// Workgroups to aim for (row blocks x slices).  A function of the matrix only.
static int g_packed_target = 512;
VPSA_EXPORT int vpsa_set_packed_target(int t) {
  const int old = g_packed_target;
  g_packed_target = t;
  return old;
}

// Reduction slices for a matrix: whole chunks per slice, at least two chunks each, enough
// workgroups to fill the chip.  0 = the kernel does not tile this shape.
VPSA_EXPORT int vpsa_packed_gemm_splits(int N, int K, int pair) {
  const int n_out = pair ? N / 2 : N;
  if (n_out <= 0 || n_out % 64 != 0 || K % PK_KC != 0 || (pair && (N & 1))) return 0;
  const int row_blocks = n_out / 64;
  int best = 1;
  for (int s = 1; s <= 32; ++s) {
    if (K % s != 0 || (K / s) % PK_KC != 0 || K / s < 2 * PK_KC) continue;
    best = s;
    if (row_blocks * s >= g_packed_target) break;
  }
  return best;
}

template <int TT, bool PAIR, int DEPTH>
static void launch_packed(const void* x, long xs, const void* wp, const void* bias, void* y, long ys,
                          float* ws, int M, int N, int K, int S, hipStream_t stream) {
  const int n_out = PAIR ? N / 2 : N;
  hipLaunchKernelGGL((packed_gemm_kernel<TT, PAIR, DEPTH>), dim3(n_out / 64, S), dim3(256), 0, stream,
                     (const uint16_t*)x, xs, (const uint16_t*)wp, (const uint16_t*)bias, (uint16_t*)y, ys,
                     S > 1 ? ws : nullptr, M, N, K, K / S);
  if (S > 1)
    hipLaunchKernelGGL((packed_reduce_kernel<PAIR>), dim3((n_out / 4 + 255) / 256, M), dim3(256), 0, stream,
                       (const float*)ws, (const uint16_t*)bias, (uint16_t*)y, ys, M, N, S);
}

// x [M, K] (row stride xs elements, 16-byte aligned rows), wp = the [N, K] matrix packed as
// described at the top, bias [N] or null, y [M, N] (pair: [M, N / 2] = silu(gate) * up with the
// gate rows first in the unpacked matrix).  ws: fp32 workspace of S * M * N elements when
// S = vpsa_packed_gemm_splits(N, K, pair) > 1.  M <= 64.  Returns -1 for shapes it does not tile.
VPSA_EXPORT int vpsa_packed_gemm(const void* x, long xs, const void* wp, const void* bias, void* y,
                                 long ys, void* ws, int M, int N, int K, int pair, int S,
                                 hipStream_t stream) {
  if (M <= 0) return 0;
  if (M > 64 || (xs & 7) || (ys & 3)) return -1;
  if (S <= 0 || S != vpsa_packed_gemm_splits(N, K, pair)) return -1;
  if (pair && bias != nullptr) return -1;
  if (S > 1 && ws == nullptr) return -1;
#define VPSA_PK(TT)                                                                                 \
  do {                                                                                              \
    if (pair) launch_packed<TT, true, 4>(x, xs, wp, bias, y, ys, (float*)ws, M, N, K, S, stream);   \
    else launch_packed<TT, false, 4>(x, xs, wp, bias, y, ys, (float*)ws, M, N, K, S, stream);       \
  } while (0)
  if (M <= 16) VPSA_PK(1);
  else if (M <= 32) VPSA_PK(2);
  else if (M <= 48) VPSA_PK(3);
  else VPSA_PK(4);
#undef VPSA_PK
  return 0;
}
