// Single-query attention over dense (unpaged) K/V for 64-channel heads: the decoder of a
// speech model attends with ONE new query per window to its own few hundred earlier positions
// and to the 1500 encoder positions of the window (cross-attention), both held as plain
// [window][position][head * 64] tensors.
//
// The work is a GEMV pair per (window, head): 2 x T x 64 bf16 values read once, no reuse, so the
// kernel is a streaming reduction, not an MFMA kernel.  One workgroup per (head, window); eight
// lanes share a key (16 bytes of the key row each: one 128-byte line per key and group), every
// group keeps four keys in flight, and each group runs its own online softmax (running maximum,
// denominator, 8 output channels per lane).  Groups are merged at the end, inside a wave with
// cross-lane moves, between waves through 2.5 KB of LDS.  Scores, probabilities and the output
// stay in fp32 registers; nothing but Q, K, V and the 64 outputs touches memory, where the same
// step written with library calls is five launches with [window, head, T] fp32 round trips.
//
// The number of valid positions is read from device memory (position counter + 1), so a captured
// HIP graph replays the same launch for every decode step.
#include "common.h"

#define DA_KEYS 4  // keys in flight per 8-lane group

__device__ __forceinline__ void da_merge(float& m, float& l, float (&acc)[8], float m2, float l2,
                                         const float (&acc2)[8]) {
  const float mn = fmaxf(m, m2);
  const float a = __builtin_amdgcn_exp2f(m - mn), b = __builtin_amdgcn_exp2f(m2 - mn);
  l = l * a + l2 * b;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a + acc2[i] * b;
  m = mn;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void dense_decode_attn_kernel(
    const uint16_t* __restrict__ q, long q_bs, const uint16_t* __restrict__ k,
    const uint16_t* __restrict__ v, long kv_bs, long kv_ts, const long* __restrict__ pos, int T,
    uint16_t* __restrict__ out, long o_bs, float scale_log2e) {
  constexpr int SLOTS = THREADS / 8, WAVES = THREADS / 64;
  __shared__ float merge[WAVES][8][10];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int c8 = tid & 7, slot = tid >> 3;
  int n = T;
  if (pos != nullptr) {
    const long p = pos[0] + 1;
    n = p < T ? (int)p : T;
  }
  float qf[8];
  {
    const u32x4 qw = *(const u32x4*)(q + b * q_bs + h * 64 + c8 * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qf[2 * i] = bf16lo(qw[i]) * scale_log2e;
      qf[2 * i + 1] = bf16hi(qw[i]) * scale_log2e;
    }
  }
  const uint16_t* kb = k + b * kv_bs + h * 64 + c8 * 8;
  const uint16_t* vb = v + b * kv_bs + h * 64 + c8 * 8;
  float m = -1e30f, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < n; base += SLOTS * DA_KEYS) {
    u32x4 kk[DA_KEYS], vv[DA_KEYS];
    bool ok[DA_KEYS];
#pragma unroll
    for (int j = 0; j < DA_KEYS; ++j) {
      const int key = base + j * SLOTS + slot;
      ok[j] = key < n;
      const long off = (long)(ok[j] ? key : n - 1) * kv_ts;
      kk[j] = *(const u32x4*)(kb + off);
      vv[j] = *(const u32x4*)(vb + off);
    }
#pragma unroll
    for (int j = 0; j < DA_KEYS; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s = fmaf(qf[2 * i], bf16lo(kk[j][i]), s);
        s = fmaf(qf[2 * i + 1], bf16hi(kk[j][i]), s);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (ok[j]) {
        const float mn = fmaxf(m, s);
        const float a = __builtin_amdgcn_exp2f(m - mn), p = __builtin_amdgcn_exp2f(s - mn);
        l = l * a + p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] = acc[2 * i] * a + p * bf16lo(vv[j][i]);
          acc[2 * i + 1] = acc[2 * i + 1] * a + p * bf16hi(vv[j][i]);
        }
        m = mn;
      }
    }
  }
  // the 8 groups of a wave: lanes that differ in bits 3..5 hold the same channels
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    float acc2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc2[i] = __shfl_xor(acc[i], off, 64);
    da_merge(m, l, acc, m2, l2, acc2);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
    merge[wave][lane][0] = m;
    merge[wave][lane][1] = l;
#pragma unroll
    for (int i = 0; i < 8; ++i) merge[wave][lane][2 + i] = acc[i];
  }
  __syncthreads();
  if (tid < 8) {
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
      float acc2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc2[i] = merge[w][tid][2 + i];
      da_merge(m, l, acc, merge[w][tid][0], merge[w][tid][1], acc2);
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
    *(u32x4*)(out + b * o_bs + h * 64 + tid * 8) = o;
  }
}

// q [B][heads * 64] (row stride q_bs elements), k / v [B][T][heads * 64] (strides kv_bs, kv_ts),
// out [B][heads * 64] (row stride o_bs); every row start 16-byte aligned.  pos: device pointer to
// the index of the newest valid position (positions 0..pos[0] are attended), or null for all T.
// Returns -1 for arguments the kernel does not cover.
VPSA_EXPORT int vpsa_dense_decode_attention(const void* q, long q_bs, const void* k, const void* v,
                                            long kv_bs, long kv_ts, const void* pos, int B, int heads,
                                            int T, void* out, long o_bs, float scale,
                                            hipStream_t stream) {
  if (B <= 0 || heads <= 0) return 0;
  if (T <= 0 || B > 65535 || (q_bs & 7) || (kv_bs & 7) || (kv_ts & 7) || (o_bs & 7)) return -1;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return -1;
  const float sl = scale * 1.4426950408889634f;
  hipLaunchKernelGGL((dense_decode_attn_kernel<512>), dim3(heads, B), dim3(512), 0, stream,
                     (const uint16_t*)q, q_bs, (const uint16_t*)k, (const uint16_t*)v, kv_bs, kv_ts,
                     (const long*)pos, T, (uint16_t*)out, o_bs, sl);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Bidirectional attention over dense Q/K/V for 64-channel heads (speech encoder: 1500 positions
// attending to each other; decoder prompt rows attending to the encoder output).
//
// One workgroup = 4 waves x 32 query rows against 64-key tiles staged through LDS (double
// buffered: the next tile's global loads are in flight during the current tile's math, one
// barrier per tile).  Per wave and tile:
//   S^T[64 keys x 32 queries] = K_tile . Q^T        8 x v_mfma_f32_32x32x16_bf16, K rows from LDS
//                                                   (ds_read_b128), Q^T fragments kept in registers
//   online softmax down each lane's column          lane (query, half) holds 32 of the 64 keys; one
//                                                   cross-half exchange per tile for the maximum
//   O^T[64 ch x 32 queries] += V_tile^T . P^T       8 MFMAs; P^T is the packed score registers as
//                                                   they are; V stays row-major in LDS and is read
//                                                   through the transposing ds_read_b64_tr_b16, whose
//                                                   4-key blocks are exactly the accumulator's key order
// Row strides of the LDS images (K 144 B, V 192 B) make both kinds of read bank-conflict free.
// The [B, heads, Tq, Tk] score matrix never exists in memory.
#define DF_KT 64
#define DF_KSTRIDE 72  // elements: 128 B of channels + 16 B
#define DF_VSTRIDE 96  // elements: 128 B of channels + 64 B

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 lds_tr_pair(const uint16_t* lo, const uint16_t* hi) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)lo);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)hi);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256) void dense_attn_fwd_kernel(
    const uint16_t* __restrict__ q, long q_bs, long q_ts, const uint16_t* __restrict__ k,
    const uint16_t* __restrict__ v, long kv_bs, long kv_ts, uint16_t* __restrict__ out, long o_bs,
    long o_ts, int Tq, int Tk, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) uint16_t lds_k[2][DF_KT * DF_KSTRIDE];
  __shared__ __attribute__((aligned(16))) uint16_t lds_v[2][DF_KT * DF_VSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qrow = blockIdx.x * 128 + wave * 32 + r;
  const int qc = qrow < Tq ? qrow : Tq - 1;
  const uint16_t* qp = q + b * q_bs + (long)qc * q_ts + h * 64;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks + 8 * hf);

  const uint16_t* kbase_p = k + b * kv_bs + h * 64;
  const uint16_t* vbase_p = v + b * kv_bs + h * 64;
  const int srow = tid >> 3, sc = tid & 7;
  u32x4 kreg[2], vreg[2];
  auto stage_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = t * DF_KT + srow + 32 * i;
      key = key < Tk ? key : Tk - 1;
      kreg[i] = *(const u32x4*)(kbase_p + (long)key * kv_ts + 8 * sc);
      vreg[i] = *(const u32x4*)(vbase_p + (long)key * kv_ts + 8 * sc);
    }
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(u32x4*)(&lds_k[buf][(srow + 32 * i) * DF_KSTRIDE + 8 * sc]) = kreg[i];
      *(u32x4*)(&lds_v[buf][(srow + 32 * i) * DF_VSTRIDE + 8 * sc]) = vreg[i];
    }
  };
  const int n_tiles = (Tk + DF_KT - 1) / DF_KT;
  stage_load(0);
  stage_write(0);
  __syncthreads();

  f32x16 o_acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) o_acc[cb][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  // transposed read: lane 4q + p of a 16-lane group points at row q, columns 4p..4p+3 of its block
  const int tr_off = ((lane >> 2) & 3) * DF_VSTRIDE + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  for (int t = 0; t < n_tiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < n_tiles) stage_load(t + 1);
    f32x16 s_acc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s_acc[kb][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *(const bf16x8*)(&lds_k[buf][(32 * kb + r) * DF_KSTRIDE + 16 * ks + 8 * hf]);
        s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s_acc[kb], 0, 0, 0);
      }
    }
    // accumulator register i of block kb, half hf = key 64 t + 32 kb + 8 (i >> 2) + 4 hf + (i & 3)
    const int kbase = t * DF_KT;
    const bool need_mask = kbase + DF_KT > Tk;
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float sv = s_acc[kb][i] * scale_log2e;
        if (need_mask && kbase + 32 * kb + 8 * (i >> 2) + 4 * hf + (i & 3) >= Tk) sv = -1e30f;
        s_acc[kb][i] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 pk;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float p0 = __builtin_amdgcn_exp2f(s_acc[kb][8 * s + 2 * jj] - m_new);
          const float p1 = __builtin_amdgcn_exp2f(s_acc[kb][8 * s + 2 * jj + 1] - m_new);
          psum += p0 + p1;
          pk[jj] = pack_bf16x2(p0, p1);
        }
        pf[kb * 2 + s] = __builtin_bit_cast(bf16x8, pk);
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) o_acc[cb][i] *= alpha;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const uint16_t* blk = &lds_v[buf][(16 * s4 + 4 * hf) * DF_VSTRIDE + 32 * cb] + tr_off;
        const bf16x8 a = lds_tr_pair(blk, blk + 8 * DF_VSTRIDE);
        o_acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[s4], o_acc[cb], 0, 0, 0);
      }
    }
    if (t + 1 < n_tiles) stage_write(buf ^ 1);
    __syncthreads();
  }
  // O^T accumulator register i of block cb, half hf = channel 32 cb + 8 (i >> 2) + 4 hf + (i & 3)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qrow < Tq) {
    const float inv = 1.f / l_tot;
    uint16_t* op = out + b * o_bs + (long)qrow * o_ts + h * 64;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x2 w;
        w[0] = pack_bf16x2(o_acc[cb][4 * g4] * inv, o_acc[cb][4 * g4 + 1] * inv);
        w[1] = pack_bf16x2(o_acc[cb][4 * g4 + 2] * inv, o_acc[cb][4 * g4 + 3] * inv);
        *(u32x2*)(op + 32 * cb + 8 * g4 + 4 * hf) = w;
      }
  }
}

// q [B][Tq][heads * 64], k / v [B][Tk][heads * 64], out [B][Tq][heads * 64]; strides in elements
// (batch, position), unit stride over channels, every row start 16-byte aligned (out: 8-byte).
// Every query attends to every key.  Returns -1 for arguments the kernel does not cover.
VPSA_EXPORT int vpsa_dense_attention(const void* q, long q_bs, long q_ts, const void* k, const void* v,
                                     long kv_bs, long kv_ts, void* out, long o_bs, long o_ts, int B,
                                     int heads, int Tq, int Tk, float scale, hipStream_t stream) {
  if (B <= 0 || heads <= 0 || Tq <= 0) return 0;
  if (Tk <= 0 || B > 65535 || heads > 65535) return -1;
  if ((q_bs & 7) || (q_ts & 7) || (kv_bs & 7) || (kv_ts & 7) || (o_bs & 3) || (o_ts & 3)) return -1;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) || ((uintptr_t)out & 7)) return -1;
  hipLaunchKernelGGL(dense_attn_fwd_kernel, dim3((Tq + 127) / 128, heads, B), dim3(256), 0, stream,
                     (const uint16_t*)q, q_bs, q_ts, (const uint16_t*)k, (const uint16_t*)v, kv_bs, kv_ts,
                     (uint16_t*)out, o_bs, o_ts, Tq, Tk, scale * 1.4426950408889634f);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
