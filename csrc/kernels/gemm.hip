// Skinny (decode-batch) projections for gfx950: y[M, N] = x[M, K] . W[N, K]^T with M <= 64.
//
// A decode step multiplies a few dozen token rows with every weight matrix of the model, so
// these GEMMs are weight-streaming: the job is to pull W through HBM exactly once at full
// bandwidth; the matrix cores are nearly idle.  Tiling for that, not for FLOPs:
//   * a workgroup owns 16 weight rows (N / 16 workgroups: 256 ... 8016 for Llama-3-8B, so
//     every CU of every XCD has work and several workgroups are resident per CU),
//   * its WAVES wavefronts split the reduction dimension K in contiguous slices (each lane
//     streams one weight row segment with 16-byte loads, GK_KU loads per operand in flight),
//   * MFMA 16x16x32 bf16 with the weight tile as the A operand, consumed straight from the
//     load registers (no LDS round trip for the streamed operand), token tiles of 16 as B
//     operands (x is a few hundred KB and stays in L2; every workgroup re-reads all of it, so
//     the kernel is the right tool for up to ~16 rows and the library GEMM beyond: the
//     dispatch thresholds in ops.linear come from benchmarks/gemm_micro.py),
//   * the K slices are combined through LDS (fp32), then the epilogue adds an optional bias
//     and writes bf16.
// PAIR = true is the SwiGLU up-projection: W = [gate rows | up rows] ([2 * inter, K]); the
// workgroup streams 16 gate rows and the 16 matching up rows against the same B fragments
// and writes silu(gate) * up, so the [M, 2 * inter] intermediate never exists.
//
// D layout of the 16x16 tile: lane (c16 = lane & 15, g = lane >> 4) holds weight rows
// 4g .. 4g+3 of token column c16, so a lane stores 4 consecutive bf16 of one output row.
#include "common.h"

// GK_KU = reduction steps (of 32 elements) whose loads are issued together
template <int TT, int WAVES, bool PAIR, int GK_KU>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(
    const uint16_t* __restrict__ x, long x_stride, const uint16_t* __restrict__ w,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, long y_stride, int M, int N, int K) {
  __shared__ f32x4 red[WAVES][PAIR ? 2 : 1][TT][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int n_out = PAIR ? (N >> 1) : N;  // output columns
  const int slice = K / WAVES;            // multiple of 32 * GK_KU (checked on the host)
  const uint16_t* wa = w + (long)(n0 + c16) * K + wave * slice + 8 * g;
  const uint16_t* wb = wa + (long)n_out * K;  // up rows (PAIR only)
  const uint16_t* xp[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int j = min(tt * 16 + c16, M - 1);  // padded columns repeat the last token
    xp[tt] = x + (long)j * x_stride + wave * slice + 8 * g;
  }
  const int nt = (M + 15) >> 4;
  f32x4 acc[TT], acc2[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) acc[tt] = acc2[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < slice; k += 32 * GK_KU) {
    u32x4 a[GK_KU], a2[GK_KU];
#pragma unroll
    for (int u = 0; u < GK_KU; ++u) {
      a[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wa + k + 32 * u));
      if (PAIR) a2[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wb + k + 32 * u));
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      if (tt < nt) {
        u32x4 b[GK_KU];
#pragma unroll
        for (int u = 0; u < GK_KU; ++u) b[u] = *reinterpret_cast<const u32x4*>(xp[tt] + k + 32 * u);
#pragma unroll
        for (int u = 0; u < GK_KU; ++u) {
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, a[u]), __builtin_bit_cast(bf16x8, b[u]), acc[tt], 0, 0, 0);
          if (PAIR)
            acc2[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, a2[u]), __builtin_bit_cast(bf16x8, b[u]), acc2[tt], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
    if (tt < nt) {
      red[wave][0][tt][lane] = acc[tt];
      if (PAIR) red[wave][1][tt][lane] = acc2[tt];
    }
  __syncthreads();
  // wave w finishes token tile w, w + WAVES, ...
  for (int tt = wave; tt < nt; tt += WAVES) {
    f32x4 s = red[0][0][tt][lane];
#pragma unroll
    for (int v = 1; v < WAVES; ++v) s += red[v][0][tt][lane];
    if (PAIR) {
      f32x4 s2 = red[0][1][tt][lane];
#pragma unroll
      for (int v = 1; v < WAVES; ++v) s2 += red[v][1][tt][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = s[r] / (1.f + __expf(-s[r])) * s2[r];
    } else if (bias != nullptr) {
      const u32x2 bb = *reinterpret_cast<const u32x2*>(bias + n0 + 4 * g);
      s[0] += bf16lo(bb[0]);
      s[1] += bf16hi(bb[0]);
      s[2] += bf16lo(bb[1]);
      s[3] += bf16hi(bb[1]);
    }
    const int j = tt * 16 + c16;
    if (j < M) {
      u32x2 o;
      o[0] = pack_bf16x2(s[0], s[1]);
      o[1] = pack_bf16x2(s[2], s[3]);
      *reinterpret_cast<u32x2*>(y + (long)j * y_stride + n0 + 4 * g) = o;
    }
  }
}

static int g_gemm_variant = 0;

template <int WAVES, bool PAIR, int GK_KU>
static int launch_skinny(const void* x, long xs, const void* w, const void* bias, void* y, long ys,
                         int M, int N, int K, hipStream_t stream) {
  const int n_out = PAIR ? N / 2 : N;
  const dim3 grid(n_out / 16), block(WAVES * 64);
#define VPSA_SK(TT)                                                                              \
  hipLaunchKernelGGL((skinny_gemm_kernel<TT, WAVES, PAIR, GK_KU>)    , grid, block, 0, stream,       \
                     (const uint16_t*)x, xs, (const uint16_t*)w, (const uint16_t*)bias,          \
                     (uint16_t*)y, ys, M, N, K)
  if (M <= 16) VPSA_SK(1);
  else if (M <= 32) VPSA_SK(2);
  else if (M <= 48) VPSA_SK(3);
  else VPSA_SK(4);
#undef VPSA_SK
  return 0;
}

// x [M, K] (row stride xs elements, 16-byte aligned rows), w [N, K] contiguous, bias [N] or
// null, y [M, N] (or [M, N / 2] when pair != 0; row stride ys).  M <= 64.
// Returns -1 when the shape is outside what the kernel tiles (the caller uses the library).
VPSA_EXPORT int vpsa_skinny_gemm(const void* x, long xs, const void* w, const void* bias, void* y,
                                 long ys, int M, int N, int K, int pair, hipStream_t stream) {
  if (M <= 0) return 0;
  if (M > 64 || (xs & 7) || (ys & 3)) return -1;
  const int n_out = pair ? N / 2 : N;
  if (n_out % 16 != 0 || (pair && (N & 1))) return -1;
  if (pair && bias != nullptr) return -1;
  // K slices per wave must be whole unrolled steps
  const bool ku8 = (g_gemm_variant & 1) != 0, four = (g_gemm_variant & 2) != 0;
  if (pair) {
    if (ku8 && K % (4 * 32 * 8) == 0) return launch_skinny<4, true, 8>(x, xs, w, bias, y, ys, M, N, K, stream);
    if (K % (4 * 32 * 4) == 0) return launch_skinny<4, true, 4>(x, xs, w, bias, y, ys, M, N, K, stream);
    return -1;
  }
  if (!four) {
    if (ku8 && K % (8 * 32 * 8) == 0) return launch_skinny<8, false, 8>(x, xs, w, bias, y, ys, M, N, K, stream);
    if (K % (8 * 32 * 4) == 0) return launch_skinny<8, false, 4>(x, xs, w, bias, y, ys, M, N, K, stream);
  }
  if (ku8 && K % (4 * 32 * 8) == 0) return launch_skinny<4, false, 8>(x, xs, w, bias, y, ys, M, N, K, stream);
  if (K % (4 * 32 * 4) == 0) return launch_skinny<4, false, 4>(x, xs, w, bias, y, ys, M, N, K, stream);
  return -1;
}

// bit 0: eight instead of four reduction steps per load batch; bit 1: four waves per
// workgroup for the plain projection as well (register-direct kernel); bit 2: 256-element
// chunks in the streaming kernel.  Returns the previous value.
VPSA_EXPORT int vpsa_set_gemm_variant(int v) {
  const int old = g_gemm_variant;
  g_gemm_variant = v;
  return old;
}

// ------------------------------------------------------------------------------------
// Streaming projection for whole decode batches (up to 64 rows), built around the two things
// the register-direct kernel above cannot do:
//   * weight loads that are contiguous per row: a load instruction covers RPI rows x KC
//     contiguous elements (256 or 512 bytes per row) instead of sixteen 64-byte pieces; the
//     tile goes through LDS (padded rows, conflict-free 16-byte reads) to reach the MFMA
//     operand layout,
//   * token rows shared through LDS by the four waves of a workgroup (64 weight rows per
//     workgroup), and the reduction dimension split across workgroups (grid.y) so that a
//     workgroup only touches K / S columns of x: token-row traffic is M / 64 of the weight
//     traffic whatever the batch.
// Global loads of chunk i+1 are issued into registers before the MFMAs of chunk i (one LDS
// buffer, two barriers per chunk).  With S > 1 the workgroups write fp32 partial tiles and
// stream_reduce_kernel sums them in a fixed order (bitwise reproducible, no atomics) and
// applies the epilogue (bias, or silu(gate) * up for PAIR); S == 1 writes bf16 directly.
// ------------------------------------------------------------------------------------
template <int TT, bool PAIR, int KC>
__global__ __launch_bounds__(256) void stream_gemm_kernel(
    const uint16_t* __restrict__ x, long x_stride, const uint16_t* __restrict__ w,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, long y_stride,
    float* __restrict__ ws, int M, int N, int K, int krange) {
  constexpr int RS = KC * 2 + 16;     // LDS row stride in bytes (16 bytes of padding)
  constexpr int NA = PAIR ? 2 : 1;    // weight tiles per wave
  constexpr int LPR = KC / 8;         // lanes covering one row of a chunk
  constexpr int RPI = 64 / LPR;       // rows per wave-level load instruction
  constexpr int NI = 16 / RPI;        // instructions per 16-row tile
  constexpr int RPP = 256 / LPR;      // token rows per workgroup-level pass
  constexpr int NB = TT * 16 / RPP;   // passes over the token tile
  __shared__ __attribute__((aligned(16))) uint8_t lds_a[4 * NA * 16 * RS];
  __shared__ __attribute__((aligned(16))) uint8_t lds_b[TT * 16 * RS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const int n_out = PAIR ? (N >> 1) : N;
  const int n0 = blockIdx.x * 64 + wave * 16;
  const long k0 = (long)blockIdx.y * krange;
  const int ar = lane / LPR, ac = (lane % LPR) * 8;
  const int br = tid / LPR, bc = (tid % LPR) * 8;
  const uint16_t* wa = w + (long)(n0 + ar) * K + k0 + ac;
  const uint16_t* xb[NB];
#pragma unroll
  for (int p = 0; p < NB; ++p) xb[p] = x + (long)min(br + p * RPP, M - 1) * x_stride + k0 + bc;
  const int nt = (M + 15) >> 4;
  u32x4 ra[NA][NI], rb[NB];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        ra[a][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(
            wa + ((long)a * n_out + i * RPI) * K + kc));
#pragma unroll
    for (int p = 0; p < NB; ++p)
      if (p * RPP < nt * 16) rb[p] = *reinterpret_cast<const u32x4*>(xb[p] + kc);
  };
  f32x4 acc[NA][TT];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[a][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(0);
  for (int kc = 0; kc < krange; kc += KC) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        *reinterpret_cast<u32x4*>(lds_a + ((wave * NA + a) * 16 + ar + i * RPI) * RS + ac * 2) = ra[a][i];
#pragma unroll
    for (int p = 0; p < NB; ++p)
      if (p * RPP < nt * 16) *reinterpret_cast<u32x4*>(lds_b + (br + p * RPP) * RS + bc * 2) = rb[p];
    __syncthreads();
    if (kc + KC < krange) fetch(kc + KC);
#pragma unroll
    for (int u = 0; u < KC / 32; ++u) {
      u32x4 af[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a)
        af[a] = *reinterpret_cast<const u32x4*>(lds_a + ((wave * NA + a) * 16 + c16) * RS + (u * 32 + 8 * g) * 2);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        if (tt < nt) {
          const u32x4 bf = *reinterpret_cast<const u32x4*>(lds_b + (tt * 16 + c16) * RS + (u * 32 + 8 * g) * 2);
#pragma unroll
          for (int a = 0; a < NA; ++a)
            acc[a][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bf), acc[a][tt], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int j = tt * 16 + c16;
    if (tt >= nt || j >= M) continue;
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

// out[m, n..n+3] = epilogue(sum_s ws[s, m, n..n+3]); one thread per 4 output columns.
template <bool PAIR>
__global__ __launch_bounds__(256) void stream_reduce_kernel(const float* __restrict__ ws,
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

// Number of reduction slices for a shape: enough workgroups to fill the chip, slices of whole
// chunks (about three workgroups per CU measured best for both small and large batches;
// vpsa_set_stream_targets changes the aim for batches of at most / more than 16 rows).
// 0 = the kernel does not tile this shape.
static int g_stream_target_small = 768, g_stream_target_large = 768;

VPSA_EXPORT int vpsa_set_stream_targets(int small_batch, int large_batch) {
  g_stream_target_small = small_batch;
  g_stream_target_large = large_batch;
  return 0;
}

VPSA_EXPORT int vpsa_stream_gemm_splits(int M, int N, int K, int pair) {
  const int kc = (g_gemm_variant & 4) ? 256 : 128;
  const int n_out = pair ? N / 2 : N;
  if (n_out <= 0 || n_out % 64 != 0 || K % kc != 0 || (pair && (N & 1))) return 0;
  const int row_blocks = n_out / 64;
  int best = 1;
  for (int s = 1; s <= 32; ++s) {
    if (K % s != 0 || (K / s) % kc != 0 || K / s < 2 * kc) continue;
    best = s;
    if (row_blocks * s >= (M <= 16 ? g_stream_target_small : g_stream_target_large)) break;
  }
  return best;
}

template <int TT, bool PAIR, int KC>
static void launch_stream(const void* x, long xs, const void* w, const void* bias, void* y, long ys,
                          float* ws, int M, int N, int K, int S, hipStream_t stream) {
  const int n_out = PAIR ? N / 2 : N;
  hipLaunchKernelGGL((stream_gemm_kernel<TT, PAIR, KC>), dim3(n_out / 64, S), dim3(256), 0, stream,
                     (const uint16_t*)x, xs, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)y,
                     ys, S > 1 ? ws : nullptr, M, N, K, K / S);
  if (S > 1)
    hipLaunchKernelGGL((stream_reduce_kernel<PAIR>), dim3((n_out / 4 + 255) / 256, M), dim3(256), 0,
                       stream, (const float*)ws, (const uint16_t*)bias, (uint16_t*)y, ys, M, N, S);
}

// Same contract as vpsa_skinny_gemm; ws is an fp32 workspace of S * M * N elements
// (S = vpsa_stream_gemm_splits(M, N, K, pair); unused when S == 1).
VPSA_EXPORT int vpsa_stream_gemm(const void* x, long xs, const void* w, const void* bias, void* y,
                                 long ys, void* ws, int M, int N, int K, int pair, int S,
                                 hipStream_t stream) {
  if (M <= 0) return 0;
  if (M > 64 || (xs & 7) || (ys & 3)) return -1;
  if (S <= 0 || S != vpsa_stream_gemm_splits(M, N, K, pair)) return -1;
  if (pair && bias != nullptr) return -1;
  if (S > 1 && ws == nullptr) return -1;
  const bool kc256 = (g_gemm_variant & 4) != 0;
#define VPSA_ST(TT)                                                                                  \
  do {                                                                                               \
    if (pair) {                                                                                      \
      if (kc256) launch_stream<TT, true, 256>(x, xs, w, bias, y, ys, (float*)ws, M, N, K, S, stream); \
      else launch_stream<TT, true, 128>(x, xs, w, bias, y, ys, (float*)ws, M, N, K, S, stream);      \
    } else {                                                                                         \
      if (kc256) launch_stream<TT, false, 256>(x, xs, w, bias, y, ys, (float*)ws, M, N, K, S, stream); \
      else launch_stream<TT, false, 128>(x, xs, w, bias, y, ys, (float*)ws, M, N, K, S, stream);     \
    }                                                                                                \
  } while (0)
  if (M <= 16) VPSA_ST(1);
  else if (M <= 32) VPSA_ST(2);
  else if (M <= 48) VPSA_ST(3);
  else VPSA_ST(4);
#undef VPSA_ST
  return 0;
}

// ------------------------------------------------------------------------------------
// HBM read roof probe: every lane streams 16-byte loads over the buffer and folds them into
// one dword (written only when a never-true condition holds, so the loads cannot be elided).
// The bandwidth it reaches is the ceiling that the weight- and KV-streaming kernels are
// measured against.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const u32x4* __restrict__ p, long n16,
                                                             uint32_t* __restrict__ sink) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(p + i);
    const u32x4 b = __builtin_nontemporal_load(p + i + stride);
    const u32x4 c = __builtin_nontemporal_load(p + i + 2 * stride);
    const u32x4 d = __builtin_nontemporal_load(p + i + 3 * stride);
    acc ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3] ^ d[0] ^
           d[1] ^ d[2] ^ d[3];
  }
  for (; i < n16; i += stride) {
    const u32x4 a = __builtin_nontemporal_load(p + i);
    acc ^= a[0] ^ a[1] ^ a[2] ^ a[3];
  }
  if (acc == 0x9e3779b9u && sink != nullptr) sink[0] = acc;
}

VPSA_EXPORT int vpsa_hbm_read_probe(const void* p, long bytes, void* sink, int blocks,
                                    hipStream_t stream) {
  if (bytes < 16 || blocks <= 0) return -1;
  hipLaunchKernelGGL(hbm_read_probe_kernel, dim3(blocks), dim3(256), 0, stream, (const u32x4*)p,
                     bytes / 16, (uint32_t*)sink);
  return 0;
}
