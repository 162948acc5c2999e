// Paged attention for gfx950 (CDNA4): MFMA flash-attention over a paged bf16 KV cache.
//
// Cache page layout (see common.h): per layer and page, K is [kv_head][4 channel blocks][64 keys][32 ch]
// and V is stored TRANSPOSED, [kv_head][2 key halves][128 ch][32 keys].  With that layout both
// matrix products take their A operand with plain 16-byte reads, and every fragment load of the
// decode kernel is one contiguous KiB:
//   S^T[key, q] = K[key, :] . Q[q, :]        A = K rows   (8 contiguous channels / lane)
//   O^T[ch,  q] = V^T[ch, :] . P^T[:, q]     A = V^T rows (8 contiguous keys / lane)
// and the S^T accumulator of the first product IS the B operand of the second (the
// query sits on the lane in both), so the probabilities never leave registers.
// Because the MFMA accumulator interleaves rows between the two half-waves, the K rows
// are fed to the first product in a permuted order chosen so that accumulator register
// (8s + j) of half-wave h holds key 16s + 8h + j: exactly the k-order of a natural
// 16-byte V^T read.
//
// Prefill kernel: one workgroup = 4 waves sharing one KV head's K/V stream through LDS
// (XOR-swizzled, conflict-free for ds_read_b128); each wave owns 32 query rows of one
// query head.  Tiles of 64 keys (= one page) are register-staged: the global loads of
// tile t+1 are issued before the MFMA work on tile t and written to LDS after it.
//
// Decode kernel: one query token per sequence; K/V go straight from HBM to registers
// (no reuse across waves, so LDS staging would only add a round trip); the GQA group's
// query heads occupy the columns of a 16x16x32 MFMA so each K/V byte is read once for
// the whole group.  Context is split across the 4 waves of a workgroup and across
// `num_splits` workgroups; a small second kernel merges the split partials.
#include "common.h"
#include <type_traits>

#define NEG_BIG (-1.0e30f)

__device__ __forceinline__ bf16x8 lds_read_frag(const char* lds, int byte_off) {
  return *reinterpret_cast<const bf16x8*>(lds + byte_off);
}

// ---------------------------------------------------------------------------------
// Prefill / chunked prefill / prefix-hit prefill.
//   q, out       [tokens, n_q * 128] bf16 with explicit row strides
//   tile_seq[i], tile_q0[i]: work list, one entry per (sequence, query tile of QT rows);
//                            tile_seq[i] < 0 marks an unused entry
//   cu_q[s]      first row of sequence s in q/out; q_len = cu_q[s+1] - cu_q[s]
//   ctx_lens[s]  keys visible to the LAST query row of s (cached prefix + new tokens; the
//                new tokens' K/V are already in the cache when this kernel runs)
// grid = (num_tiles, n_kv * max(G/4, 1)); block = 256.
// GP = min(G, 4) query heads share a workgroup; QT = 32 * (4 / GP) query rows per tile.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void paged_prefill_kernel(
    const uint16_t* __restrict__ q, long q_stride, uint16_t* __restrict__ out, long out_stride,
    const uint16_t* __restrict__ cache, long page_stride, long v_offset,
    const int* __restrict__ block_tables, int max_blocks, const int* __restrict__ cu_q,
    const int* __restrict__ ctx_lens, const int* __restrict__ tile_seq,
    const int* __restrict__ tile_q0, int n_q, int n_kv, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* lds_k = lds;           // 64 keys x 256 B
  char* lds_v = lds + 16384;   // 128 channels x 128 B

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;

  const int G = n_q / n_kv;
  const int GP = G < 4 ? G : 4;
  const int groups_per_kv = (G + GP - 1) / GP;
  const int kvh = blockIdx.y / groups_per_kv;
  const int hg = blockIdx.y % groups_per_kv;
  // a group ratio that is not a multiple of GP leaves the last workgroup of a KV head with
  // fewer query heads than waves: the spare waves recompute the last head (they still feed the
  // shared K/V tiles and meet every barrier) and write nothing
  const int head_slot = hg * GP + (wave % GP);
  const bool head_ok = head_slot < G;
  const int head = kvh * G + (head_ok ? head_slot : G - 1);
  const int qsub = wave / GP;

  const int seq = tile_seq[blockIdx.x];
  if (seq < 0) return;  // padding of a fixed-size work list (captured graphs): the whole workgroup leaves
  const int q0 = tile_q0[blockIdx.x];
  const int row0 = cu_q[seq];
  const int q_len = cu_q[seq + 1] - row0;
  const int ctx = ctx_lens[seq];
  const int pos0 = ctx - q_len;  // absolute position of the sequence's first new token
  const int QT = 32 * (4 / GP);

  // rows of this wave / block
  const int wq0 = q0 + qsub * 32;
  const int my_row = wq0 + r;
  const bool row_ok = my_row < q_len;
  const int my_pos = row_ok ? pos0 + my_row : ctx - 1;
  int w_last = wq0 + 31;
  if (w_last > q_len - 1) w_last = q_len - 1;
  const bool wave_ok = wq0 < q_len;
  const int w_max_pos = pos0 + w_last;   // last key this wave needs
  const int w_min_pos = pos0 + wq0;      // first row's position
  int b_last = q0 + QT - 1;
  if (b_last > q_len - 1) b_last = q_len - 1;
  const int n_tiles = (pos0 + b_last) / PAGE + 1;

  // ---- Q fragments (B operand): lane holds Q[row r][16*ks + 8h .. +8]
  bf16x8 qf[8];
  {
    const uint16_t* qp = q + (long)(row0 + (row_ok ? my_row : 0)) * q_stride + head * HEAD_DIM + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
      if (!row_ok) v = (u32x4){0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }

  const int* bt = block_tables + (long)seq * max_blocks;

  // staging registers: 4 K chunks + 4 V chunks of 16 B per thread
  u32x4 sk[4], sv[4];
  auto stage_load = [&](int tile) {
    const long page = bt[tile];
    const uint16_t* kp = cache + page * page_stride + (long)kvh * PAGE * HEAD_DIM;
    const uint16_t* vp = cache + page * page_stride + v_offset + (long)kvh * HEAD_DIM * PAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sk[i] = *reinterpret_cast<const u32x4*>(kp + k_stage_tid(tid) + k_stage_i(i));
      sv[i] = *reinterpret_cast<const u32x4*>(vp + v_stage_tid(tid) + v_stage_i(i));
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = k_stage_key(tid, i), c = k_stage_chunk(tid, i);
      *reinterpret_cast<u32x4*>(lds_k + key * 256 + ((c ^ (key & 15)) << 4)) = sk[i];
      const int d = v_stage_ch(tid, i), cv = v_stage_chunk(tid, i);
      *reinterpret_cast<u32x4*>(lds_v + d * 128 + ((cv ^ ((d >> 1) & 7)) << 4)) = sv[i];
    }
  };

  f32x16 o_acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o_acc[dt][i] = 0.f;
  float m_run = NEG_BIG;
  float l_run = 0.f;

  // MFMA row rho of a 32-key half-tile reads key swap_bits23(rho)
  const int krow = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);

  stage_load(0);
  stage_write();
  __syncthreads();

  for (int t = 0; t < n_tiles; ++t) {
    if (t + 1 < n_tiles) stage_load(t + 1);

    const int kbase = t * PAGE;
    if (wave_ok && kbase <= w_max_pos) {
      f32x16 s_acc[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s_acc[tt][i] = 0.f;
        const int key = tt * 32 + krow;
        const int rowoff = key * 256;
        const int sw = key & 15;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          bf16x8 a = lds_read_frag(lds_k, rowoff + (((2 * ks + h) ^ sw) << 4));
          s_acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s_acc[tt], 0, 0, 0);
        }
      }
      // ---- online softmax (base-2 domain); accumulator reg i of half h = key
      //      kbase + 32 tt + 16 (i>>3) + 8 h + (i&7)
      const bool need_mask = (kbase + PAGE - 1) > w_min_pos;
      float mx = NEG_BIG;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = s_acc[tt][i] * scale_log2e;
          if (need_mask) {
            const int key = kbase + 32 * tt + 16 * (i >> 3) + 8 * h + (i & 7);
            if (key > my_pos) v = NEG_BIG;
          }
          s_acc[tt][i] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
      bf16x8 pf[4];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          u32x4 pk;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float p0 = __builtin_amdgcn_exp2f(s_acc[tt][8 * s + 2 * jj] - m_new);
            const float p1 = __builtin_amdgcn_exp2f(s_acc[tt][8 * s + 2 * jj + 1] - m_new);
            psum += p0 + p1;
            pk[jj] = pack_bf16x2(p0, p1);
          }
          pf[tt * 2 + s] = __builtin_bit_cast(bf16x8, pk);
        }
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o_acc[dt][i] *= alpha;
        const int d = 32 * dt + r;
        const int rowoff = d * 128;
        const int sw = (d >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          bf16x8 a = lds_read_frag(lds_v, rowoff + (((2 * kk + h) ^ sw) << 4));
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[kk], o_acc[dt], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (t + 1 < n_tiles) stage_write();
    __syncthreads();
  }

  // ---- epilogue: O^T accumulator reg i of half h, tile dt = channel 32dt + 8(i>>2) + 4h + (i&3)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (row_ok && head_ok) {
    const float inv = 1.f / l_tot;
    uint16_t* op = out + (long)(row0 + my_row) * out_stride + head * HEAD_DIM;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x2 w;
        w[0] = pack_bf16x2(o_acc[dt][4 * g4] * inv, o_acc[dt][4 * g4 + 1] * inv);
        w[1] = pack_bf16x2(o_acc[dt][4 * g4 + 2] * inv, o_acc[dt][4 * g4 + 3] * inv);
        *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g4 + 4 * h) = w;
      }
  }
}

// ---------------------------------------------------------------------------------
// Second-generation prefill kernel.  Same work decomposition, operand layouts and LDS
// images as paged_prefill_kernel above; what changes is the schedule around the MFMAs:
//   * two LDS buffers (64 KiB per workgroup, two workgroups per CU): the next tile is
//     written into the other buffer at the top of an iteration, so one barrier per
//     64-key tile instead of two, and the global loads run two tiles ahead;
//   * softmax in the scaled base-2 domain straight off the raw scores: one fma + one
//     exp2 per probability, the row maximum taken on raw scores (max3-shaped);
//   * the O rescale is skipped (wave-uniform) while no row's maximum grew by more than
//     LAZY_THR in the base-2 domain: probabilities then stay <= 2^LAZY_THR, O and l
//     keep the same reference maximum, and O / l is unchanged;
//   * causal masking only on tiles that cross this wave's diagonal.
// ---------------------------------------------------------------------------------
#define LAZY_THR 6.0f

template <bool PRIO, bool HEAD_MAJOR>
__global__ __launch_bounds__(256, 2) void paged_prefill_kernel_v2(
    const uint16_t* __restrict__ q, long q_stride, uint16_t* __restrict__ out, long out_stride,
    const uint16_t* __restrict__ cache, long page_stride, long v_offset,
    const int* __restrict__ block_tables, int max_blocks, const int* __restrict__ cu_q,
    const int* __restrict__ ctx_lens, const int* __restrict__ tile_seq,
    const int* __restrict__ tile_q0, int n_q, int n_kv, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 x (16 KiB K + 16 KiB V)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;

  // Workgroups are handed to the 8 XCDs round-robin in linear launch order.  HEAD_MAJOR makes
  // the (kv head, head group) index the fastest-varying one, so that all q-tiles that read one
  // KV head's keys run on one XCD and share its L2 (8 KV heads <-> 8 XCDs for Llama-3).  It pays
  // when a sequence has many q-tiles (chunks of long prompts: +6 %); with one tile per sequence
  // there is nothing to share and the plain order is 14 % faster, so the launcher picks.
  int bx = blockIdx.x, by = blockIdx.y;
  if (HEAD_MAJOR) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    by = lin % gridDim.y;
    bx = lin / gridDim.y;
  }
  const int G = n_q / n_kv;
  const int GP = G < 4 ? G : 4;
  const int groups_per_kv = (G + GP - 1) / GP;
  const int kvh = by / groups_per_kv;
  const int hg = by % groups_per_kv;
  // a group ratio that is not a multiple of GP leaves the last workgroup of a KV head with
  // fewer query heads than waves: the spare waves recompute the last head (they still feed the
  // shared K/V tiles and meet every barrier) and write nothing
  const int head_slot = hg * GP + (wave % GP);
  const bool head_ok = head_slot < G;
  const int head = kvh * G + (head_ok ? head_slot : G - 1);
  const int qsub = wave / GP;

  const int seq = tile_seq[bx];
  if (seq < 0) return;  // padding of a fixed-size work list (captured graphs): the whole workgroup leaves
  const int q0 = tile_q0[bx];
  const int row0 = cu_q[seq];
  const int q_len = cu_q[seq + 1] - row0;
  const int ctx = ctx_lens[seq];
  const int pos0 = ctx - q_len;
  const int QT = 32 * (4 / GP);

  const int wq0 = q0 + qsub * 32;
  const int my_row = wq0 + r;
  const bool row_ok = my_row < q_len;
  const int my_pos = row_ok ? pos0 + my_row : ctx - 1;
  int w_last = wq0 + 31;
  if (w_last > q_len - 1) w_last = q_len - 1;
  const bool wave_ok = wq0 < q_len;
  const int w_max_pos = pos0 + w_last;
  const int w_min_pos = pos0 + wq0;
  int b_last = q0 + QT - 1;
  if (b_last > q_len - 1) b_last = q_len - 1;
  const int n_tiles = (pos0 + b_last) / PAGE + 1;

  bf16x8 qf[8];
  {
    const uint16_t* qp = q + (long)(row0 + (row_ok ? my_row : 0)) * q_stride + head * HEAD_DIM + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
      if (!row_ok) v = (u32x4){0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }

  const int* bt = block_tables + (long)seq * max_blocks;
  const long kv_head_off = (long)kvh * PAGE * HEAD_DIM;

  u32x4 sk[4], sv[4];
  auto stage_load = [&](int tile) {
    const long page = bt[tile];
    const uint16_t* kp = cache + page * page_stride + kv_head_off + k_stage_tid(tid);
    const uint16_t* vp = cache + page * page_stride + kv_head_off + v_offset + v_stage_tid(tid);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sk[i] = *reinterpret_cast<const u32x4*>(kp + k_stage_i(i));
      sv[i] = *reinterpret_cast<const u32x4*>(vp + v_stage_i(i));
    }
  };
  // per-thread LDS write offsets (same swizzled images as the first-generation kernel)
  int wk_off[4], wv_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = k_stage_key(tid, i), c = k_stage_chunk(tid, i);
    wk_off[i] = key * 256 + ((c ^ (key & 15)) << 4);
    const int d = v_stage_ch(tid, i), cv = v_stage_chunk(tid, i);
    wv_off[i] = 16384 + d * 128 + ((cv ^ ((d >> 1) & 7)) << 4);
  }
  auto stage_write = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(buf + wk_off[i]) = sk[i];
      *reinterpret_cast<u32x4*>(buf + wv_off[i]) = sv[i];
    }
  };

  f32x16 o_acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o_acc[dt][i] = 0.f;
  float m_run = NEG_BIG;  // reference maximum, scaled base-2 domain
  float l_run = 0.f;

  const int krow = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);
  // K read offsets: row (32 tt + krow), chunk (2 ks + h) ^ (row & 15)
  const int k_sw = krow & 15;  // (32 tt + krow) & 15 == krow & 15
  const int k_row_off = krow * 256;

  stage_load(0);
  stage_write(lds);
  if (n_tiles > 1) stage_load(1);
  __syncthreads();

  for (int t = 0; t < n_tiles; ++t) {
    char* cur = lds + ((t & 1) << 15);
    if (t + 1 < n_tiles) {
      stage_write(lds + (((t + 1) & 1) << 15));
      if (t + 2 < n_tiles) stage_load(t + 2);
    }

    const int kbase = t * PAGE;
    if (wave_ok && kbase <= w_max_pos) {
      const char* lk = cur;
      const char* lv = cur + 16384;
      f32x16 s_acc[2];
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s_acc[tt][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          bf16x8 a = lds_read_frag(lk, tt * 8192 + k_row_off + (((2 * ks + h) ^ k_sw) << 4));
          s_acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s_acc[tt], 0, 0, 0);
        }
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);

      // accumulator reg i of half h = key kbase + 32 tt + 16 (i>>3) + 8 h + (i&7)
      if ((kbase + PAGE - 1) > w_min_pos) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kbase + 32 * tt + 16 * (i >> 3) + 8 * h + (i & 7);
            if (key > my_pos) s_acc[tt][i] = NEG_BIG;
          }
      }
      float mx0 = fmaxf(s_acc[0][0], s_acc[1][0]);
      float mx1 = fmaxf(s_acc[0][1], s_acc[1][1]);
#pragma unroll
      for (int i = 2; i < 16; i += 2) {
        mx0 = fmaxf(fmaxf(mx0, s_acc[0][i]), s_acc[1][i]);
        mx1 = fmaxf(fmaxf(mx1, s_acc[0][i + 1]), s_acc[1][i + 1]);
      }
      float mx = fmaxf(mx0, mx1);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // masked entries are NEG_BIG: keep the product finite and far below any real score
      const float m_tile = fmaxf(mx * scale_log2e, NEG_BIG);
      const bool grow = m_tile > m_run + LAZY_THR;
      if (__any(grow)) {
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o_acc[dt][i] *= alpha;
      }
      float ps0 = 0.f, ps1 = 0.f;
      bf16x8 pf[4];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          u32x4 pk;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float p0 = __builtin_amdgcn_exp2f(
                __builtin_fmaf(s_acc[tt][8 * s + 2 * jj], scale_log2e, -m_run));
            const float p1 = __builtin_amdgcn_exp2f(
                __builtin_fmaf(s_acc[tt][8 * s + 2 * jj + 1], scale_log2e, -m_run));
            ps0 += p0;
            ps1 += p1;
            pk[jj] = pack_bf16x2(p0, p1);
          }
          pf[tt * 2 + s] = __builtin_bit_cast(bf16x8, pk);
        }
      l_run += ps0 + ps1;
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = 32 * dt + r;
        const int rowoff = d * 128;
        const int sw = (d >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          bf16x8 a = lds_read_frag(lv, rowoff + (((2 * kk + h) ^ sw) << 4));
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[kk], o_acc[dt], 0, 0, 0);
        }
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (row_ok && head_ok) {
    const float inv = 1.f / l_tot;
    uint16_t* op = out + (long)(row0 + my_row) * out_stride + head * HEAD_DIM;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        u32x2 w;
        w[0] = pack_bf16x2(o_acc[dt][4 * g4] * inv, o_acc[dt][4 * g4 + 1] * inv);
        w[1] = pack_bf16x2(o_acc[dt][4 * g4 + 2] * inv, o_acc[dt][4 * g4 + 3] * inv);
        *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g4 + 4 * h) = w;
      }
  }
}

// ---------------------------------------------------------------------------------
// Third-generation prefill kernel: ONE wave per SIMD with the whole 512-register file,
// 64 query rows per wave.
//   * a workgroup is still 4 waves = the 4 query heads of one GQA group against one KV head's
//     K/V stream, but every wave now owns TWO 32-row query blocks: each K / V^T fragment read
//     from LDS feeds two MFMAs (one per block), so LDS reads, LDS writes, L2 reads and barriers
//     per MFMA are all halved against the second generation;
//   * the unit of the online softmax is a 32-key half tile, software-pipelined one deep: while
//     the probabilities of half tile u are exponentiated (VALU) the scores of half tile u+1 are
//     already being computed (16 MFMAs), and the row maxima of u+1 are taken under the 16 PV
//     MFMAs of u.  Two score sets of 32 registers alternate, so nothing is copied;
//   * three 32 KiB LDS buffers in a ring (tile t multiplied into O, tile t+1 read for the next
//     scores, tile t+2 being written): one barrier per 64-key tile;
//   * tiles below the diagonal of the whole 64-row block take a mask-free body; the at most
//     two tiles that cross it take a second body with the causal mask;
//   * lazy rescale (LAZY_THR) as in the second generation.
// grid = (num_tiles of 64 rows, n_kv * G / 4); block = 256; dynamic LDS = 96 KiB.
// ---------------------------------------------------------------------------------
// O accumulators of the third-generation prefill kernel live in the accumulator half of the
// register file for the whole kernel: the MFMA is written as inline asm with "a" constraints so
// the compiler cannot place them in (or copy them through) arch VGPRs.  The compiler does not
// pad hazards of an asm statement, so the two wait states between a vector write of an operand
// (packed probabilities) and the MFMA that reads it are inside the string.
__device__ __forceinline__ void mfma_acc_agpr(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// Score MFMAs of the hand-interleaved stream: destination in arch VGPRs (the softmax reads it),
// query fragments from AGPRs (loaded once per workgroup).  The first MFMA of a chain takes the
// constant 0 as its accumulator input.  Whoever reads the result with vector instructions must be
// at least one full MFMA (12 wait states) behind the last MFMA of the chain: in the stream below
// the first reader is eight MFMAs later.
__device__ __forceinline__ void mfma_s_first(f32x16& d, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_s_next(f32x16& d, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
// Values produced by an asm MFMA are not known to the compiler's hazard recogniser: before vector
// code reads them (rescale, epilogue) let the matrix pipe drain.
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
// Keeps a value (and the instruction that made it) where the source put it.
#define VPSA_PIN(x) asm volatile("" : "+v"(x))

template <bool HEAD_MAJOR, int SCHED>
__global__ __launch_bounds__(256, 1) void paged_prefill_kernel_v3(
    const uint16_t* __restrict__ q, long q_stride, uint16_t* __restrict__ out, long out_stride,
    const uint16_t* __restrict__ cache, long page_stride, long v_offset,
    const int* __restrict__ block_tables, int max_blocks, const int* __restrict__ cu_q,
    const int* __restrict__ ctx_lens, const int* __restrict__ tile_seq,
    const int* __restrict__ tile_q0, int n_q, int n_kv, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 3 x (16 KiB K + 16 KiB V)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;

  int bx = blockIdx.x, by = blockIdx.y;
  if (HEAD_MAJOR) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    by = lin % gridDim.y;
    bx = lin / gridDim.y;
  }
  const int G = n_q / n_kv;
  const int groups_per_kv = G >> 2;
  const int kvh = by / groups_per_kv;
  const int head = kvh * G + (by % groups_per_kv) * 4 + wave;

  const int seq = tile_seq[bx];
  if (seq < 0) return;  // padding of a fixed-size work list (captured graphs): the whole workgroup leaves
  const int q0 = tile_q0[bx];
  const int row0 = cu_q[seq];
  const int q_len = cu_q[seq + 1] - row0;
  const int ctx = ctx_lens[seq];
  const int pos0 = ctx - q_len;

  int my_row[2], my_pos[2];
  bool row_ok[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    my_row[rb] = q0 + 32 * rb + r;
    row_ok[rb] = my_row[rb] < q_len;
    my_pos[rb] = row_ok[rb] ? pos0 + my_row[rb] : ctx - 1;
  }
  int b_last = q0 + 63;
  if (b_last > q_len - 1) b_last = q_len - 1;
  const int n_tiles = (pos0 + b_last) / PAGE + 1;
  const int last = n_tiles - 1;
  // tiles [0, n_full) lie entirely at or below the first row's position: no mask for any row
  const int n_full = (pos0 + q0 + 1) / PAGE;

  bf16x8 qf[2][8];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const uint16_t* qp =
        q + (long)(row0 + (row_ok[rb] ? my_row[rb] : 0)) * q_stride + head * HEAD_DIM + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
      if (!row_ok[rb]) v = (u32x4){0, 0, 0, 0};
      qf[rb][ks] = __builtin_bit_cast(bf16x8, v);
      // from here on the fragment is an AGPR value: every MFMA of the main loop reads it there
      if (SCHED == 1) asm volatile("" : "+a"(qf[rb][ks]));
    }
  }

  const int* bt = block_tables + (long)seq * max_blocks;
  const long kv_head_off = (long)kvh * PAGE * HEAD_DIM;

  u32x4 sk[4], sv[4];
  auto stage_load = [&](int tile) {
    const long page = bt[tile];
    const uint16_t* kp = cache + page * page_stride + kv_head_off + k_stage_tid(tid);
    const uint16_t* vp = cache + page * page_stride + kv_head_off + v_offset + v_stage_tid(tid);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sk[i] = *reinterpret_cast<const u32x4*>(kp + k_stage_i(i));
      sv[i] = *reinterpret_cast<const u32x4*>(vp + v_stage_i(i));
    }
  };
  int wk_off[4], wv_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = k_stage_key(tid, i), c = k_stage_chunk(tid, i);
    wk_off[i] = key * 256 + ((c ^ (key & 15)) << 4);
    const int d = v_stage_ch(tid, i), cv = v_stage_chunk(tid, i);
    wv_off[i] = 16384 + d * 128 + ((cv ^ ((d >> 1) & 7)) << 4);
  }
  auto stage_write = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(buf + wk_off[i]) = sk[i];
      *reinterpret_cast<u32x4*>(buf + wv_off[i]) = sv[i];
    }
  };

  f32x16 o_acc[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o_acc[rb][dt][i] = 0.f;
  float m_run[2] = {NEG_BIG, NEG_BIG};
  float l_run[2] = {0.f, 0.f};

  const int krow = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);
  const int k_sw = krow & 15;
  const int k_row_off = krow * 256;
  // V^T read offsets of this lane for the four channel blocks
  int v_row_off[4], v_sw[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int d = 32 * dt + r;
    v_row_off[dt] = 16384 + d * 128;
    v_sw[dt] = (d >> 1) & 7;
  }

  // S^T of the 32-key half `tt` of the tile in `buf`, both row blocks
  auto qk_half = [&](const char* buf, int tt, f32x16 (&s)[2]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[0][i] = 0.f;
      s[1][i] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 a = lds_read_frag(buf, tt * 8192 + k_row_off + (((2 * ks + h) ^ k_sw) << 4));
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[0][ks], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[1][ks], s[1], 0, 0, 0);
    }
  };
  // accumulator reg i of lane half h = key kbase + 32 tt + 16 (i>>3) + 8 h + (i&7)
  auto mask_half = [&](int tile, int tt, f32x16 (&s)[2]) {
    const int kb = tile * PAGE + 32 * tt + 8 * h;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (kb + 16 * (i >> 3) + (i & 7) > my_pos[rb]) s[rb][i] = NEG_BIG;
  };
  auto max_half = [&](const f32x16 (&s)[2], float (&mt)[2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float mx0 = fmaxf(s[rb][0], s[rb][1]);
      float mx1 = fmaxf(s[rb][2], s[rb][3]);
#pragma unroll
      for (int i = 4; i < 16; i += 4) {
        mx0 = fmaxf(fmaxf(mx0, s[rb][i]), s[rb][i + 1]);
        mx1 = fmaxf(fmaxf(mx1, s[rb][i + 2]), s[rb][i + 3]);
      }
      float mx = fmaxf(mx0, mx1);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      mt[rb] = fmaxf(mx * scale_log2e, NEG_BIG);
    }
  };
  auto rescale = [&](const float (&mt)[2]) {
    const bool grow = (mt[0] > m_run[0] + LAZY_THR) || (mt[1] > m_run[1] + LAZY_THR);
    if (__any(grow)) {
      mfma_drain();
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float m_new = fmaxf(m_run[rb], mt[rb]);
        const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
        m_run[rb] = m_new;
        l_run[rb] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o_acc[rb][dt][i] *= alpha;
      }
    }
  };
  auto exp_half = [&](const f32x16 (&s)[2], bf16x8 (&pf)[2][2]) {
#pragma unroll
    for (int sh = 0; sh < 2; ++sh)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float nm = -m_run[rb];
        float ps0 = 0.f, ps1 = 0.f;
        u32x4 pk;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float p0 =
              __builtin_amdgcn_exp2f(__builtin_fmaf(s[rb][8 * sh + 2 * jj], scale_log2e, nm));
          const float p1 =
              __builtin_amdgcn_exp2f(__builtin_fmaf(s[rb][8 * sh + 2 * jj + 1], scale_log2e, nm));
          ps0 += p0;
          ps1 += p1;
          pk[jj] = pack_bf16x2(p0, p1);
        }
        pf[rb][sh] = __builtin_bit_cast(bf16x8, pk);
        l_run[rb] += ps0 + ps1;
      }
  };
  // O^T += V^T(keys of half tt) . P^T, 16-key slice `sh` first for all channels, then the next
  auto pv_half = [&](const char* buf, int tt, const bf16x8 (&pf)[2][2]) {
#pragma unroll
    for (int sh = 0; sh < 2; ++sh)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x8 a = lds_read_frag(buf, v_row_off[dt] + (((4 * tt + 2 * sh + h) ^ v_sw[dt]) << 4));
        mfma_acc_agpr(o_acc[0][dt], a, pf[0][sh]);
        mfma_acc_agpr(o_acc[1][dt], a, pf[1][sh]);
      }
  };

  // One half-tile step with the instruction stream interleaved by hand (SCHED = 1): 32 MFMAs,
  // each followed by the slice of vector work that fits in its shadow, and a scheduling fence
  // so that the compiler keeps the slices where they are.
  //   MFMAs  0-15: scores of the NEXT half tile (kbuf, half ktt) -> sn
  //   MFMAs 16-23: O += V^T(keys 0-15 of the current half) . P,  24-31: keys 16-31
  //   vector work: the 32 exponentials of the current scores `sc` (24 under the score MFMAs,
  //   8 under the first PV MFMAs); under the last 8 MFMAs the row maxima of `sn` and one
  //   staging operation per MFMA (STAGE 1: the 8 LDS writes of the tile held in the staging
  //   registers, STAGE 2: the 8 global loads of the tile after it).
  // LDS fragments are read two MFMA pairs ahead into four rotating registers `fr`; the last two
  // pairs fetch the first two K fragments of the following step (nkbuf, half nktt), so a step
  // starts with its operands already on the way.
  bf16x8 fr[4];
  const uint16_t* ld_kp = nullptr;  // source of the tile being loaded (STAGE 2)
  auto fused_half = [&](auto stage_tag, const char* kbuf, int ktt, const char* vbuf, int vtt,
                        const char* nkbuf, int nktt, char* wbuf, const f32x16 (&sc)[2],
                        f32x16 (&sn)[2], float (&mn)[2], bool mask_next, int next_tile) {
    constexpr int STAGE = decltype(stage_tag)::value;
    const float nm[2] = {-m_run[0], -m_run[1]};
    float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float pv[32];     // probabilities between their exp2 and their pack (a few live at a time)
    u32x4 pw[2][2];   // packed probabilities [row block][16-key slice]
    // value v of 32: slice sh = v >> 4, row block rb = (v >> 3) & 1, element i = 8 sh + (v & 7).
    // `start` issues the fma + exp2, `finish` (one MFMA later, so the exp2 result is ready and
    // no wait state is needed) adds it to the row sum and packs each completed pair.
    auto start = [&](int v) {
      const int sh = v >> 4, rb = (v >> 3) & 1, i = 8 * sh + (v & 7);
      pv[v] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[rb][i], scale_log2e, nm[rb]));
      VPSA_PIN(pv[v]);
    };
    auto finish = [&](int v) {
      const int sh = v >> 4, rb = (v >> 3) & 1;
      ps[rb][v & 1] += pv[v];
      VPSA_PIN(ps[rb][v & 1]);
      if (v & 1) {
        uint32_t w = pack_bf16x2(pv[v - 1], pv[v]);
        VPSA_PIN(w);
        pw[rb][sh][(v & 7) >> 1] = w;
      }
    };
    // LDS read of item n: 0-7 = K fragments of this step, 8-15 = V^T fragments (sh = (n-8) >> 2,
    // dt = (n-8) & 3), 16-17 = K fragments 0-1 of the following step
    auto fetch = [&](int n) {
      if (n < 8)
        fr[n & 3] = lds_read_frag(kbuf, ktt * 8192 + k_row_off + (((2 * n + h) ^ k_sw) << 4));
      else if (n < 16)
        fr[n & 3] = lds_read_frag(
            vbuf, v_row_off[(n - 8) & 3] + (((4 * vtt + 2 * ((n - 8) >> 2) + h) ^ v_sw[(n - 8) & 3]) << 4));
      else
        fr[n & 3] = lds_read_frag(nkbuf, nktt * 8192 + k_row_off + (((2 * (n - 16) + h) ^ k_sw) << 4));
    };
    auto stage_op = [&](int i) {  // i = 0..7
      if (STAGE == 1) {
        if (i < 4)
          *reinterpret_cast<u32x4*>(wbuf + wk_off[i]) = sk[i];
        else
          *reinterpret_cast<u32x4*>(wbuf + wv_off[i - 4]) = sv[i - 4];
      } else if (STAGE == 2) {
        if (i < 4)
          sk[i] = *reinterpret_cast<const u32x4*>(ld_kp + k_stage_tid(tid) + k_stage_i(i));
        else
          sv[i - 4] = *reinterpret_cast<const u32x4*>(ld_kp + v_offset + v_stage_tid(tid) + v_stage_i(i - 4));
      }
    };
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks == 0)
        mfma_s_first(sn[0], fr[0], qf[0][0]);
      else
        mfma_s_next(sn[0], fr[ks & 3], qf[0][ks]);
      fetch(ks + 2);
      if (ks > 0) {
        finish(3 * ks - 2);
        finish(3 * ks - 1);
      }
      start(3 * ks);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0)
        mfma_s_first(sn[1], fr[0], qf[1][0]);
      else
        mfma_s_next(sn[1], fr[ks & 3], qf[1][ks]);
      finish(3 * ks);
      start(3 * ks + 1);
      start(3 * ks + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    float mx[2][2];
    float mfin[2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
      if (sh == 1 && mask_next) mask_half(next_tile, ktt, sn);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int n = 8 + 4 * sh + dt;
        const int g = 2 * (4 * sh + dt);  // PV MFMA index 0..15 of this and the next MFMA
        mfma_acc_agpr(o_acc[0][dt], fr[n & 3], __builtin_bit_cast(bf16x8, pw[0][sh]));
        fetch(n + 2);
        if (sh == 0) {
          if (dt == 0) finish(22);
          finish(23 + 2 * dt);
          start(24 + 2 * dt);
        } else {
          if (dt == 0) finish(31);
          stage_op(2 * dt);
        }
        if (sh == 1) {
          // row maxima of sn: block 0 under PV MFMAs 8-10, block 1 under 11-13, two chains each
#define VPSA_MAX_SLICE(gg)                                                                 \
  {                                                                                        \
    const int rb_ = (gg) >= 11 ? 1 : 0, st_ = ((gg) - 8) % 3;                               \
    if ((gg) >= 8 && (gg) <= 13) {                                                         \
      if (st_ == 0) {                                                                      \
        mx[rb_][0] = fmaxf(fmaxf(sn[rb_][0], sn[rb_][1]), sn[rb_][2]);                    \
        mx[rb_][1] = fmaxf(fmaxf(sn[rb_][3], sn[rb_][4]), sn[rb_][5]);                    \
      } else if (st_ == 1) {                                                               \
        mx[rb_][0] = fmaxf(fmaxf(mx[rb_][0], sn[rb_][6]), sn[rb_][7]);                    \
        mx[rb_][1] = fmaxf(fmaxf(mx[rb_][1], sn[rb_][8]), sn[rb_][9]);                    \
        mx[rb_][0] = fmaxf(fmaxf(mx[rb_][0], sn[rb_][10]), sn[rb_][11]);                  \
      } else {                                                                             \
        mx[rb_][1] = fmaxf(fmaxf(mx[rb_][1], sn[rb_][12]), sn[rb_][13]);                  \
        mx[rb_][0] = fmaxf(fmaxf(mx[rb_][0], sn[rb_][14]), sn[rb_][15]);                  \
        mx[rb_][0] = fmaxf(mx[rb_][0], mx[rb_][1]);                                        \
      }                                                                                    \
      VPSA_PIN(mx[rb_][0]);                                                                \
      VPSA_PIN(mx[rb_][1]);                                                                \
    } else if ((gg) >= 14) { /* cross-half maximum and scaling of block gg - 14 */          \
      const int rb2_ = (gg) - 14;                                                          \
      const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[rb2_][0]),      \
                                                        __float_as_uint(mx[rb2_][0]), false, false); \
      mfin[rb2_] = fmaxf(fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1])) * scale_log2e, NEG_BIG); \
      l_run[rb2_] += ps[rb2_][0] + ps[rb2_][1];                                            \
      VPSA_PIN(mfin[rb2_]);                                                                \
    }                                                                                      \
  }
          VPSA_MAX_SLICE(g)
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_acc_agpr(o_acc[1][dt], fr[n & 3], __builtin_bit_cast(bf16x8, pw[1][sh]));
        if (sh == 0) {
          finish(24 + 2 * dt);
          start(24 + 2 * dt + 1);
        } else {
          stage_op(2 * dt + 1);
          VPSA_MAX_SLICE(g + 1)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef VPSA_MAX_SLICE
    mn[0] = mfin[0];
    mn[1] = mfin[1];
  };

  // ring state: byte offsets of the buffers holding tile t, tile t+1, and the one to fill
  int b_cur = 0, b_nxt = 32768, b_wr = 65536;

  stage_load(0);
  stage_write(lds);
  stage_load(last < 1 ? last : 1);
  stage_write(lds + 32768);
  stage_load(last < 2 ? last : 2);
  __syncthreads();

  f32x16 sA[2], sB[2];
  float mA[2], mB[2];
  qk_half(lds, 0, sA);
  if (n_full < 1) mask_half(0, 0, sA);
  max_half(sA, mA);

  // One 64-key tile = two half-tile steps.  Entering: sA / mA = scores and row maxima of
  // (tile t, half 0).  Leaving: sA / mA = those of (tile t+1, half 0).
#define VPSA_V3_TILE(t, MASKED)                                                       \
  {                                                                                   \
    if (SCHED == 1) {                                                                 \
      {                                                                               \
        const int nt = (t) + 3;                                                       \
        const long page = bt[nt < last ? nt : last];                                  \
        ld_kp = cache + page * page_stride + kv_head_off;                             \
      }                                                                               \
      rescale(mA);                                                                    \
      fused_half(std::integral_constant<int, 1>{}, lds + b_cur, 1, lds + b_cur, 0, lds + b_nxt, 0, \
                 lds + b_wr, sA, sB, mB, MASKED && (t) >= n_full, (t));               \
      rescale(mB);                                                                    \
      fused_half(std::integral_constant<int, 2>{}, lds + b_nxt, 0, lds + b_cur, 1, lds + b_nxt, 1, \
                 lds + b_wr, sB, sA, mA, MASKED && (t) + 1 >= n_full, (t) + 1);       \
    } else {                                                                          \
      stage_write(lds + b_wr);                                                        \
      {                                                                               \
        const int nt = (t) + 3;                                                       \
        stage_load(nt < last ? nt : last);                                            \
      }                                                                               \
      rescale(mA);                                                                    \
      {                                                                               \
        bf16x8 pf[2][2];                                                              \
        exp_half(sA, pf);                                                             \
        qk_half(lds + b_cur, 1, sB);                                                  \
        pv_half(lds + b_cur, 0, pf);                                                  \
        if (MASKED && (t) >= n_full) mask_half((t), 1, sB);                           \
        max_half(sB, mB);                                                             \
      }                                                                               \
      rescale(mB);                                                                    \
      {                                                                               \
        bf16x8 pf[2][2];                                                              \
        exp_half(sB, pf);                                                             \
        qk_half(lds + b_nxt, 0, sA);                                                  \
        pv_half(lds + b_cur, 1, pf);                                                  \
        if (MASKED && (t) + 1 >= n_full) mask_half((t) + 1, 0, sA);                   \
        max_half(sA, mA);                                                             \
      }                                                                               \
    }                                                                                 \
    __syncthreads();                                                                  \
    const int tmp = b_cur;                                                            \
    b_cur = b_nxt;                                                                    \
    b_nxt = b_wr;                                                                     \
    b_wr = tmp;                                                                       \
  }

  // operands of the first hand-interleaved step: K fragments 0-1 of (tile 0, half 1)
  if (SCHED == 1) {
    fr[0] = lds_read_frag(lds, 8192 + k_row_off + (((0 + h) ^ k_sw) << 4));
    fr[1] = lds_read_frag(lds, 8192 + k_row_off + (((2 + h) ^ k_sw) << 4));
  }

  int t = 0;
  // mask-free iterations: neither tile t nor tile t+1 crosses the diagonal
  const int n_fast = n_full - 1 < last ? n_full - 1 : last;
  for (; t < n_fast; ++t) VPSA_V3_TILE(t, false)
  for (; t < last; ++t) VPSA_V3_TILE(t, true)
#undef VPSA_V3_TILE
  // last tile: nothing left to prefetch after its second half
  {
    rescale(mA);
    bf16x8 pf[2][2];
    exp_half(sA, pf);
    qk_half(lds + b_cur, 1, sB);
    pv_half(lds + b_cur, 0, pf);
    if (last >= n_full) mask_half(last, 1, sB);
    max_half(sB, mB);
    rescale(mB);
    exp_half(sB, pf);
    pv_half(lds + b_cur, 1, pf);
  }

  mfma_drain();
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float l_tot = l_run[rb] + __shfl_xor(l_run[rb], 32, 64);
    if (row_ok[rb]) {
      const float inv = 1.f / l_tot;
      uint16_t* op = out + (long)(row0 + my_row[rb]) * out_stride + head * HEAD_DIM;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          u32x2 w;
          w[0] = pack_bf16x2(o_acc[rb][dt][4 * g4] * inv, o_acc[rb][dt][4 * g4 + 1] * inv);
          w[1] = pack_bf16x2(o_acc[rb][dt][4 * g4 + 2] * inv, o_acc[rb][dt][4 * g4 + 3] * inv);
          *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g4 + 4 * h) = w;
        }
    }
  }
}

// ---------------------------------------------------------------------------------
// Fourth-generation prefill kernel: the data path of the third generation (64 query rows per
// wave, one wave per SIMD with the whole register file, O accumulators in AGPRs, ring of three
// 32 KiB tile buffers, lazy rescale) with the instruction stream re-timed so that every MFMA gap
// carries ONE exponential:
//   * a step still issues 32 MFMAs for one 32-key half tile u, but in the order
//       gaps  0- 7: O += V^T(keys 16-31 of u-1) . P      (probabilities packed in the step before)
//       gaps  8-23: scores of half tile u+1
//       gaps 24-31: O += V^T(keys  0-15 of u)   . P
//     so the 32 exponentials of u have the whole step: one per gap (the third generation needed
//     24 of them under the first 16 MFMAs), each with its fma issued at least one gap earlier
//     and its add / pack one gap later: no instruction waits for the one before it;
//   * the fmas of the last eight values are issued early (gaps 8-15), which leaves gaps 25-31 to
//     the row maxima of the new scores; staging (8 global loads in the first step of a tile,
//     8 LDS writes in the second) sits in gaps 16-23;
//   * the barrier of a tile is between its two steps: the buffer that receives tile t+2 is last
//     read (V of tile t-1, keys 48-63) at the start of the first step of tile t;
//   * a rescale (rare) also scales the packed probabilities that are still waiting for their
//     MFMAs; row maxima use v_max3 written as asm (the compiler otherwise canonicalises MFMA
//     results with an extra v_max before every fmaxf); PV MFMAs carry no wait states (their
//     operands were written several instructions earlier).
// grid = (num_tiles of 64 rows, n_kv * G / 4); block = 256; dynamic LDS = 96 KiB.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void mfma_acc_agpr_now(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float d;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// DBG: timing experiments with WRONG results (bit 0: no barrier inside the tile loop, bit 1: no
// K/V loads inside the tile loop, bit 2: a plain move in place of every exponential); 0 in
// every production launch.
#define SPLIT_REC_FLOATS (8192 + 256)

template <bool HEAD_MAJOR, bool DIRECT, int DBG = 0>
__global__ __launch_bounds__(256, 1) void paged_prefill_kernel_v4(
    const uint16_t* __restrict__ q, long q_stride, uint16_t* __restrict__ out, long out_stride,
    const uint16_t* __restrict__ cache, long page_stride, long v_offset,
    const int* __restrict__ block_tables, int max_blocks, const int* __restrict__ cu_q,
    const int* __restrict__ ctx_lens, const int* __restrict__ tile_seq,
    const int* __restrict__ tile_q0, int n_q, int n_kv, float scale_log2e,
    const float* __restrict__ rope, const int* __restrict__ tile_split, float* __restrict__ split_ws) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 3 x (16 KiB K + 16 KiB V)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;

  int bx = blockIdx.x, by = blockIdx.y;
  if (HEAD_MAJOR) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    by = lin % gridDim.y;
    bx = lin / gridDim.y;
  }
  const int G = n_q / n_kv;
  // a group ratio that is not a multiple of 4 leaves the last workgroup of a KV head with fewer
  // heads than waves: the spare waves recompute the last head (they share the K/V loads and the
  // barriers) and store nothing
  const int groups_per_kv = (G + 3) >> 2;
  const int kvh = by / groups_per_kv;
  const int head_slot = (by % groups_per_kv) * 4 + wave;
  const bool head_ok = head_slot < G;
  const int head = kvh * G + (head_ok ? head_slot : G - 1);

  const int seq = tile_seq[bx];
  if (seq < 0) return;  // padding of a fixed-size work list (captured graphs): the whole workgroup leaves
  const int q0 = tile_q0[bx];
  const int row0 = cu_q[seq];
  const int q_len = cu_q[seq + 1] - row0;
  const int ctx = ctx_lens[seq];
  const int pos0 = ctx - q_len;

  int my_row[2], my_pos[2];
  bool row_ok[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    my_row[rb] = q0 + 32 * rb + r;
    row_ok[rb] = my_row[rb] < q_len;
    my_pos[rb] = row_ok[rb] ? pos0 + my_row[rb] : ctx - 1;
  }
  int b_last = q0 + 63;
  if (b_last > q_len - 1) b_last = q_len - 1;
  const int n_tiles_all = (pos0 + b_last) / PAGE + 1;
  // tiles [0, n_full) lie entirely at or below the first row's position: no mask for any row
  const int n_full_all = (pos0 + q0 + 1) / PAGE;
  // Split work-list entries (few query tiles in front of a long context: one workgroup per tile
  // and head group would walk hundreds of key tiles alone while the chip idles): the entry covers
  // key tiles [kt0, kt1) only and leaves an unnormalised partial (O, row maxima, row sums) in
  // split_ws record `part`; prefill_combine_kernel merges the records of a tile.  Everything below
  // works on tile numbers and positions RELATIVE to kt0.  The host makes the last split at least
  // three tiles long: a 64-row tile spans at most two key tiles of the diagonal, so every row sees
  // at least one whole tile in every split it takes part in.
  int kt0 = 0, kt1 = n_tiles_all, part = -1;
  if (tile_split != nullptr) {
    kt0 = tile_split[3 * bx];
    kt1 = tile_split[3 * bx + 1] < n_tiles_all ? tile_split[3 * bx + 1] : n_tiles_all;
    part = tile_split[3 * bx + 2];
  }
  const int last = kt1 - 1 - kt0;
  const int n_full = n_full_all - kt0;

  bf16x8 qf[2][8];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const uint16_t* qp =
        q + (long)(row0 + (row_ok[rb] ? my_row[rb] : 0)) * q_stride + head * HEAD_DIM + 8 * h;
    u32x4 qv[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qv[ks] = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
      if (!row_ok[rb]) qv[ks] = (u32x4){0, 0, 0, 0};
    }
    if (rope != nullptr) {
      // rotary embedding of the query row, here instead of in a pass of its own over the
      // activations: channel c < 64 pairs with c + 64, which the same lane holds (fragment ks
      // and ks + 4); table row = [cos(0..63), sin(0..63)] of the row's position
      const float* cs = rope + (long)my_pos[rb] * HEAD_DIM + 8 * h;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 16 * ks);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(cs + 16 * ks + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + 64 + 16 * ks);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(cs + 64 + 16 * ks + 4);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float ca = w < 2 ? c0[2 * w] : c1[2 * w - 4], cb = w < 2 ? c0[2 * w + 1] : c1[2 * w - 3];
          const float sa = w < 2 ? s0[2 * w] : s1[2 * w - 4], sb = w < 2 ? s0[2 * w + 1] : s1[2 * w - 3];
          const uint32_t lo = qv[ks][w], hi = qv[ks + 4][w];
          const float x0 = bf16lo(lo), x1 = bf16hi(lo), y0 = bf16lo(hi), y1 = bf16hi(hi);
          qv[ks][w] = pack_bf16x2(x0 * ca - y0 * sa, x1 * cb - y1 * sb);
          qv[ks + 4][w] = pack_bf16x2(y0 * ca + x0 * sa, y1 * cb + x1 * sb);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qf[rb][ks] = __builtin_bit_cast(bf16x8, qv[ks]);
      asm volatile("" : "+a"(qf[rb][ks]));  // an AGPR value from here on
    }
  }

  const int* bt = block_tables + (long)seq * max_blocks + kt0;
  const long kv_head_off = (long)kvh * PAGE * HEAD_DIM;
  my_pos[0] -= kt0 * PAGE;  // (the rotary table above took the absolute positions)
  my_pos[1] -= kt0 * PAGE;

  u32x4 sk[4], sv[4];
  auto stage_load = [&](int tile) {
    const long page = bt[tile];
    const uint16_t* kp = cache + page * page_stride + kv_head_off + k_stage_tid(tid);
    const uint16_t* vp = cache + page * page_stride + kv_head_off + v_offset + v_stage_tid(tid);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sk[i] = *reinterpret_cast<const u32x4*>(kp + k_stage_i(i));
      sv[i] = *reinterpret_cast<const u32x4*>(vp + v_stage_i(i));
    }
  };
  int wk_off[4], wv_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = k_stage_key(tid, i), c = k_stage_chunk(tid, i);
    wk_off[i] = key * 256 + ((c ^ (key & 15)) << 4);
    const int d = v_stage_ch(tid, i), cv = v_stage_chunk(tid, i);
    wv_off[i] = 16384 + d * 128 + ((cv ^ ((d >> 1) & 7)) << 4);
  }
  auto stage_write = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(buf + wk_off[i]) = sk[i];
      *reinterpret_cast<u32x4*>(buf + wv_off[i]) = sv[i];
    }
  };
  // DIRECT: the tile goes from global memory into its swizzled LDS image without passing through
  // registers.  One load instruction of a wave fills one contiguous KiB of LDS, lane l the 16 bytes
  // at l * 16: four keys' rows of the K image (16 slots each) or eight channels' rows of the V image
  // (8 slots each).  The swizzle is therefore applied to the SOURCE: the lane reads the chunk whose
  // place in the image its slot is.  Instruction i of wave w fills KiB 4 i + w of the 16.
  int gk_off[4], gv_off[4];  // element offsets inside one head's K part / V part
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kib = 4 * i + wave;
    const int key = 4 * kib + (lane >> 4), c = (lane & 15) ^ (key & 15);
    gk_off[i] = (c >> 2) * (PAGE * 32) + key * 32 + (c & 3) * 8;
    const int d = 8 * kib + (lane >> 3), cv = (lane & 7) ^ ((d >> 1) & 7);
    gv_off[i] = (cv >> 2) * (HEAD_DIM * 32) + d * 32 + (cv & 3) * 8;
  }
  auto direct_op = [&](const uint16_t* head_kp, char* buf, int i) {  // i = 0..7: K KiB, then V KiB
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    if (i < 4)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(head_kp + gk_off[i]), (lds_ptr_t)(buf + (4 * i + wave) * 1024),
                                       16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(head_kp + v_offset + gv_off[i - 4]),
                                       (lds_ptr_t)(buf + 16384 + (4 * (i - 4) + wave) * 1024), 16, 0, 0);
  };
  auto direct_tile = [&](int tile, char* buf) {
    const uint16_t* head_kp = cache + (long)bt[tile] * page_stride + kv_head_off;
#pragma unroll
    for (int i = 0; i < 8; ++i) direct_op(head_kp, buf, i);
  };

  f32x16 o_acc[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o_acc[rb][dt][i] = 0.f;
  float m_run[2] = {NEG_BIG, NEG_BIG};
  float l_run[2] = {0.f, 0.f};

  const int krow = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);
  const int k_sw = krow & 15;
  const int k_row_off = krow * 256;
  int v_row_off[4], v_sw[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int d = 32 * dt + r;
    v_row_off[dt] = 16384 + d * 128;
    v_sw[dt] = (d >> 1) & 7;
  }
  auto k_frag = [&](const char* buf, int tt, int ks) {
    return lds_read_frag(buf, tt * 8192 + k_row_off + (((2 * ks + h) ^ k_sw) << 4));
  };
  auto v_frag = [&](const char* buf, int tt, int sh, int dt) {
    return lds_read_frag(buf, v_row_off[dt] + (((4 * tt + 2 * sh + h) ^ v_sw[dt]) << 4));
  };

  // ---- plain (compiler-scheduled) pieces: prologue, last tile
  auto qk_half = [&](const char* buf, int tt, f32x16 (&s)[2]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[0][i] = 0.f;
      s[1][i] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 a = k_frag(buf, tt, ks);
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[0][ks], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[1][ks], s[1], 0, 0, 0);
    }
  };
  // accumulator reg i of lane half h = key kbase + 32 tt + 16 (i>>3) + 8 h + (i&7)
  auto mask_half = [&](int tile, int tt, f32x16 (&s)[2]) {
    const int kb = tile * PAGE + 32 * tt + 8 * h;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (kb + 16 * (i >> 3) + (i & 7) > my_pos[rb]) s[rb][i] = NEG_BIG;
  };
  auto max_half = [&](const f32x16 (&s)[2], float (&mt)[2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float mx0 = fmaxf(s[rb][0], s[rb][1]);
      float mx1 = fmaxf(s[rb][2], s[rb][3]);
#pragma unroll
      for (int i = 4; i < 16; i += 4) {
        mx0 = fmaxf(fmaxf(mx0, s[rb][i]), s[rb][i + 1]);
        mx1 = fmaxf(fmaxf(mx1, s[rb][i + 2]), s[rb][i + 3]);
      }
      float mx = fmaxf(mx0, mx1);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      mt[rb] = fmaxf(mx * scale_log2e, NEG_BIG);
    }
  };
  // packed probabilities of the 16-key slice still waiting for its MFMAs (keys 16-31 of the
  // half tile exponentiated last); zero before the first step
  u32x4 pw1[2] = {(u32x4){0, 0, 0, 0}, (u32x4){0, 0, 0, 0}};
  auto rescale = [&](const float (&mt)[2]) {
    const bool grow = (mt[0] > m_run[0] + LAZY_THR) || (mt[1] > m_run[1] + LAZY_THR);
    if (__any(grow)) {
      mfma_drain();
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float m_new = fmaxf(m_run[rb], mt[rb]);
        const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
        m_run[rb] = m_new;
        l_run[rb] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o_acc[rb][dt][i] *= alpha;
        // the waiting probabilities were taken against the old maximum
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pw1[rb][j] = pack_bf16x2(bf16lo(pw1[rb][j]) * alpha, bf16hi(pw1[rb][j]) * alpha);
      }
      asm volatile("s_nop 1" ::: "memory");
    }
  };
  // probabilities of 16-key slice `sh` of the scores `s`, packed; adds their sums to l_run
  auto exp_slice = [&](const f32x16 (&s)[2], int sh, u32x4 (&pk)[2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const float nm = -m_run[rb];
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[rb][8 * sh + 2 * jj], scale_log2e, nm));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[rb][8 * sh + 2 * jj + 1], scale_log2e, nm));
        ps0 += p0;
        ps1 += p1;
        pk[rb][jj] = pack_bf16x2(p0, p1);
      }
      l_run[rb] += ps0 + ps1;
    }
  };
  // O^T += V^T(16-key slice sh of half tt) . P^T
  auto pv_slice = [&](const char* buf, int tt, int sh, const u32x4 (&pk)[2]) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x8 a = v_frag(buf, tt, sh, dt);
      mfma_acc_agpr(o_acc[0][dt], a, __builtin_bit_cast(bf16x8, pk[0]));
      mfma_acc_agpr(o_acc[1][dt], a, __builtin_bit_cast(bf16x8, pk[1]));
    }
  };

  // ---- one hand-placed step.  Entering: sc = scores of half tile u with m_run settled for them,
  // pw1 = packed probabilities of keys 16-31 of u-1 (their V^T in pbuf, half ptt), fr[0..1] = the
  // V^T fragments dt = 0, 1 of that slice already on their way.  Leaving: sn / mn = scores and
  // scaled row maxima of u+1 (K in kbuf, half ktt), pw1 = keys 16-31 of u (V^T in vbuf, half vtt),
  // fr[0..1] = the first two fragments of the next step.
  // LDS item n of a step: 0-3 V^T(pbuf, ptt, slice 1, dt n), 4-11 K(kbuf, ktt, ks n-4),
  // 12-15 V^T(vbuf, vtt, slice 0, dt n-12), 16-17 V^T(vbuf, vtt, slice 1, dt n-16) for the next step.
  bf16x8 fr[4];
  const uint16_t* ld_kp = nullptr;  // source of the tile being loaded (STAGE 2)
  auto step = [&](auto stage_tag, auto masked_tag, const char* pbuf, int ptt, const char* kbuf,
                  int ktt, const char* vbuf, int vtt, char* wbuf, const f32x16 (&sc)[2],
                  f32x16 (&sn)[2], float (&mn)[2], int next_tile) {
    constexpr int STAGE = decltype(stage_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    const float nm[2] = {-m_run[0], -m_run[1]};
    float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float pv[32];
    u32x4 pw0[2];
    // value v of 32: slice sh = v >> 4, row block rb = (v >> 3) & 1, element i = 8 sh + (v & 7)
    auto fma_ = [&](int v) {
      const int sh = v >> 4, rb = (v >> 3) & 1, i = 8 * sh + (v & 7);
      pv[v] = __builtin_fmaf(sc[rb][i], scale_log2e, nm[rb]);
      VPSA_PIN(pv[v]);
    };
    auto exp_ = [&](int v) {
      if (DBG & 4)
        asm volatile("v_mov_b32 %0, %0" : "+v"(pv[v]));
      else
        pv[v] = __builtin_amdgcn_exp2f(pv[v]);
      VPSA_PIN(pv[v]);
    };
    auto fin_ = [&](int v) {
      const int sh = v >> 4, rb = (v >> 3) & 1;
      ps[rb][v & 1] += pv[v];
      VPSA_PIN(ps[rb][v & 1]);
      if (v & 1) {
        uint32_t w = pack_bf16x2(pv[v - 1], pv[v]);
        VPSA_PIN(w);
        if (sh == 0)
          pw0[rb][(v & 7) >> 1] = w;
        else
          pw1[rb][(v & 7) >> 1] = w;
      }
    };
    auto fetch = [&](int n) {
      if (n < 4)
        fr[n & 3] = v_frag(pbuf, ptt, 1, n);
      else if (n < 12)
        fr[n & 3] = k_frag(kbuf, ktt, n - 4);
      else if (n < 16)
        fr[n & 3] = v_frag(vbuf, vtt, 0, n - 12);
      else
        fr[n & 3] = v_frag(vbuf, vtt, 1, n - 16);
    };
    auto stage_op = [&](int i) {  // i = 0..7
      if (DIRECT) {
        if (!(DBG & 2)) direct_op(ld_kp, wbuf, i);
      } else if (STAGE == 1) {
        if (i < 4)
          *reinterpret_cast<u32x4*>(wbuf + wk_off[i]) = sk[i];
        else
          *reinterpret_cast<u32x4*>(wbuf + wv_off[i - 4]) = sv[i - 4];
      } else if (STAGE == 2) {
        if (i < 4)
          sk[i] = *reinterpret_cast<const u32x4*>(ld_kp + k_stage_tid(tid) + k_stage_i(i));
        else
          sv[i - 4] = *reinterpret_cast<const u32x4*>(ld_kp + v_offset + v_stage_tid(tid) + v_stage_i(i - 4));
      }
    };
    float mx[2][2];
    float mfin[2];
    // vector work of gap g (after MFMA g): finish value g, exponentiate g+1, fma of g+2 (values
    // 24-31 get theirs in gaps 8-15), staging in 16-23, row maxima of sn in 25-31
    auto gap = [&](int g) {
      if (g + 1 < 32) exp_(g + 1);
      if (g + 2 < 24) fma_(g + 2);
      if (g >= 8 && g < 16) fma_(g + 16);
      if (DIRECT) {
        // the receiving buffer was last read in the first step of this tile (before the barrier):
        // the loads leave at the start of the second step and have until the next tile's barrier
        if (STAGE == 1 && g < 8) stage_op(g);
      } else {
        if (STAGE == 1 && g >= 16 && g < 24) stage_op(g - 16);
        if (STAGE == 2 && g < 8) stage_op(g);
      }
      fin_(g);  // last: its exponential was issued a whole gap ago, no wait state is needed
      if (!MASKED && g >= 25) {
        const int rb_ = g >= 28 ? 1 : 0, st_ = (g - 25) % 3;
        if (g <= 30) {
          if (st_ == 0) {
            mx[rb_][0] = max3_raw(sn[rb_][0], sn[rb_][1], sn[rb_][2]);
            mx[rb_][1] = max3_raw(sn[rb_][3], sn[rb_][4], sn[rb_][5]);
          } else if (st_ == 1) {
            mx[rb_][0] = max3_raw(mx[rb_][0], sn[rb_][6], sn[rb_][7]);
            mx[rb_][1] = max3_raw(mx[rb_][1], sn[rb_][8], sn[rb_][9]);
            mx[rb_][0] = max3_raw(mx[rb_][0], sn[rb_][10], sn[rb_][11]);
          } else {
            mx[rb_][1] = max3_raw(mx[rb_][1], sn[rb_][12], sn[rb_][13]);
            mx[rb_][0] = max3_raw(mx[rb_][0], sn[rb_][14], sn[rb_][15]);
            mx[rb_][0] = max3_raw(mx[rb_][0], mx[rb_][1], mx[rb_][1]);
          }
        }
        if (g == 28 || g == 31) {  // cross-half maximum and scaling of block 0 / 1
          const int rb2_ = g == 28 ? 0 : 1;
          const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[rb2_][0]),
                                                            __float_as_uint(mx[rb2_][0]), false, false);
          mfin[rb2_] = max3_raw(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]), __uint_as_float(sw_[1])) *
                       scale_log2e;
          mfin[rb2_] = fmaxf(mfin[rb2_], NEG_BIG);
          VPSA_PIN(mfin[rb2_]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // before the first MFMA: the first value's fma and exponential, the second's fma
    fma_(0);
    fma_(1);
    exp_(0);
    __builtin_amdgcn_sched_barrier(0);
    // gaps 0-7: pending slice (keys 16-31 of u-1)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      mfma_acc_agpr_now(o_acc[0][dt], fr[dt & 3], __builtin_bit_cast(bf16x8, pw1[0]));
      __builtin_amdgcn_sched_barrier(0);
      fetch(dt + 2);
      gap(2 * dt);
      mfma_acc_agpr_now(o_acc[1][dt], fr[dt & 3], __builtin_bit_cast(bf16x8, pw1[1]));
      __builtin_amdgcn_sched_barrier(0);
      gap(2 * dt + 1);
    }
    // gaps 8-23: scores of u+1
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int n = 4 + ks;
      if (ks == 0)
        mfma_s_first(sn[0], fr[n & 3], qf[0][0]);
      else
        mfma_s_next(sn[0], fr[n & 3], qf[0][ks]);
      __builtin_amdgcn_sched_barrier(0);
      fetch(n + 2);
      gap(8 + 2 * ks);
      if (ks == 0)
        mfma_s_first(sn[1], fr[n & 3], qf[1][0]);
      else
        mfma_s_next(sn[1], fr[n & 3], qf[1][ks]);
      __builtin_amdgcn_sched_barrier(0);
      gap(8 + 2 * ks + 1);
    }
    // gaps 24-31: keys 0-15 of u
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int n = 12 + dt;
      mfma_acc_agpr_now(o_acc[0][dt], fr[n & 3], __builtin_bit_cast(bf16x8, pw0[0]));
      __builtin_amdgcn_sched_barrier(0);
      fetch(n + 2);
      gap(24 + 2 * dt);
      mfma_acc_agpr_now(o_acc[1][dt], fr[n & 3], __builtin_bit_cast(bf16x8, pw0[1]));
      __builtin_amdgcn_sched_barrier(0);
      gap(24 + 2 * dt + 1);
    }
    l_run[0] += ps[0][0] + ps[0][1];
    l_run[1] += ps[1][0] + ps[1][1];
    if (MASKED) {
      // tiles on the diagonal (at most two per block): mask and maxima after the matrix pipe drained
      mfma_drain();
      if (next_tile >= n_full) mask_half(next_tile, ktt, sn);
      max_half(sn, mn);
    } else {
      mn[0] = mfin[0];
      mn[1] = mfin[1];
    }
  };

  // ring state: byte offsets of the buffers holding tile t, tile t+1, and tile t-1 (= the one
  // that receives tile t+2)
  int b_cur = 0, b_nxt = 32768, b_wr = 65536;

  if (DIRECT) {
    direct_tile(0, lds);
    direct_tile(0, lds + 65536);  // (finite contents for the empty pending slice, see below)
    direct_tile(last < 1 ? last : 1, lds + 32768);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    stage_load(0);
    stage_write(lds);
    // the first step multiplies an empty pending slice (pw1 = 0) by V^T of "tile -1": give that
    // buffer finite contents
    stage_write(lds + 65536);
    stage_load(last < 1 ? last : 1);
    stage_write(lds + 32768);
  }
  __syncthreads();

  f32x16 sA[2], sB[2];
  float mA[2], mB[2];
  qk_half(lds, 0, sA);
  if (n_full < 1) mask_half(0, 0, sA);
  max_half(sA, mA);

  // One 64-key tile = two steps.  Entering: sA / mA = scores and row maxima of (tile t, half 0).
#define VPSA_V4_TILE(t, MASKED)                                                                 \
  {                                                                                             \
    {                                                                                           \
      const int nt = (t) + 2;                                                                   \
      const long page = bt[nt < last ? nt : last];                                              \
      ld_kp = cache + page * page_stride + kv_head_off;                                         \
    }                                                                                           \
    rescale(mA);                                                                                \
    step(std::integral_constant<int, 2>{}, std::integral_constant<bool, MASKED>{}, lds + b_wr, 1, \
         lds + b_cur, 1, lds + b_cur, 0, lds + b_wr, sA, sB, mB, (t));                          \
    /* tile t+1 has landed (a ring of four with two tiles of slack measured the same: the     \
       loads cost throughput, not latency) */                                                   \
    if (DIRECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                \
    if (!(DBG & 1)) __syncthreads();                                                            \
    rescale(mB);                                                                                \
    step(std::integral_constant<int, 1>{}, std::integral_constant<bool, MASKED>{}, lds + b_cur, 0, \
         lds + b_nxt, 0, lds + b_cur, 1, lds + b_wr, sB, sA, mA, (t) + 1);                      \
    const int tmp = b_wr;                                                                       \
    b_wr = b_cur;                                                                               \
    b_cur = b_nxt;                                                                              \
    b_nxt = tmp;                                                                                \
  }

  // operands of the first step: V^T fragments of the (empty) pending slice
  fr[0] = v_frag(lds + b_wr, 1, 1, 0);
  fr[1] = v_frag(lds + b_wr, 1, 1, 1);

  int t = 0;
  const int n_fast = n_full - 1 < last ? n_full - 1 : last;
  for (; t < n_fast; ++t) VPSA_V4_TILE(t, false)
  for (; t < last; ++t) VPSA_V4_TILE(t, true)
#undef VPSA_V4_TILE
  // last tile (compiler-scheduled): the slice left pending by the steps, then both halves
  {
    // (the tail of the loop re-requests the last tile into free buffers: nothing may still be
    // on its way into LDS when this workgroup ends)
    if (DIRECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mfma_drain();
    if (last > 0) pv_slice(lds + b_wr, 1, 1, pw1);
    u32x4 pk[2];
    rescale(mA);
    exp_slice(sA, 0, pk);
    pv_slice(lds + b_cur, 0, 0, pk);
    exp_slice(sA, 1, pk);
    pv_slice(lds + b_cur, 0, 1, pk);
    qk_half(lds + b_cur, 1, sB);
    if (last >= n_full) mask_half(last, 1, sB);
    max_half(sB, mB);
    pw1[0] = (u32x4){0, 0, 0, 0};
    pw1[1] = (u32x4){0, 0, 0, 0};
    rescale(mB);
    exp_slice(sB, 0, pk);
    pv_slice(lds + b_cur, 1, 0, pk);
    exp_slice(sB, 1, pk);
    pv_slice(lds + b_cur, 1, 1, pk);
  }

  mfma_drain();
  if (!head_ok) return;  // (no barrier below)
  if (part >= 0) {
    // record = [2 row blocks][4 channel tiles][16 values][64 lanes] fp32, then [2][64] maxima and
    // [2][64] sums: lane-major, so that the combine kernel (same lane roles) reads it back whole
    float* rec = split_ws + ((long)(part * gridDim.y + by) * 4 + wave) * SPLIT_REC_FLOATS;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const float l_tot = l_run[rb] + __shfl_xor(l_run[rb], 32, 64);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) rec[((rb * 4 + dt) * 16 + i) * 64 + lane] = o_acc[rb][dt][i];
      rec[8192 + rb * 64 + lane] = m_run[rb];
      rec[8192 + 128 + rb * 64 + lane] = l_tot;
    }
    return;
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float l_tot = l_run[rb] + __shfl_xor(l_run[rb], 32, 64);
    if (row_ok[rb]) {
      const float inv = 1.f / l_tot;
      uint16_t* op = out + (long)(row0 + my_row[rb]) * out_stride + head * HEAD_DIM;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          u32x2 w;
          w[0] = pack_bf16x2(o_acc[rb][dt][4 * g4] * inv, o_acc[rb][dt][4 * g4 + 1] * inv);
          w[1] = pack_bf16x2(o_acc[rb][dt][4 * g4 + 2] * inv, o_acc[rb][dt][4 * g4 + 3] * inv);
          *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g4 + 4 * h) = w;
        }
    }
  }
}

// Merge the partial records of split work-list entries.  grid = (groups, head groups, 8): group g =
// (sequence, first row of the tile, first record, number of records); blockIdx.z picks one (row
// block, channel tile) slice of the record, a wave is a head and its lanes hold what they hold in
// the attention kernel's epilogue: 16 values each, so the merge of a tile is spread over 8 x the
// head groups workgroups and every record is read with whole-wave contiguous loads.
__global__ __launch_bounds__(256) void prefill_combine_kernel(
    const float* __restrict__ split_ws, const int* __restrict__ groups, uint16_t* __restrict__ out,
    long out_stride, const int* __restrict__ cu_q, int n_q, int n_kv) {
  const int nparts = groups[4 * blockIdx.x + 3];
  if (nparts <= 0) return;  // padding
  const int seq = groups[4 * blockIdx.x], q0 = groups[4 * blockIdx.x + 1], part0 = groups[4 * blockIdx.x + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int by = blockIdx.y, rb = blockIdx.z >> 2, dt = blockIdx.z & 3;
  const int G = n_q / n_kv, groups_per_kv = (G + 3) >> 2;
  const int head_slot = (by % groups_per_kv) * 4 + wave;
  if (head_slot >= G) return;  // a spare wave of the last head group
  const int head = (by / groups_per_kv) * G + head_slot;
  const int row0 = cu_q[seq], q_len = cu_q[seq + 1] - row0;
  const int row = q0 + 32 * rb + r;
  const float* rec0 = split_ws + ((long)(part0 * gridDim.y + by) * 4 + wave) * SPLIT_REC_FLOATS;
  const long rec_step = (long)gridDim.y * 4 * SPLIT_REC_FLOATS;
  float m = NEG_BIG;
  for (int p = 0; p < nparts; ++p) m = fmaxf(m, rec0[p * rec_step + 8192 + rb * 64 + lane]);
  float l = 0.f;
  float o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll 4
  for (int p = 0; p < nparts; ++p) {
    const float* rec = rec0 + p * rec_step;
    const float f = __builtin_amdgcn_exp2f(rec[8192 + rb * 64 + lane] - m);
    l += f * rec[8192 + 128 + rb * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] += f * rec[((rb * 4 + dt) * 16 + i) * 64 + lane];
  }
  if (row < q_len) {
    const float inv = 1.f / l;
    uint16_t* op = out + (long)(row0 + row) * out_stride + head * HEAD_DIM;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      u32x2 w;
      w[0] = pack_bf16x2(o[4 * g4] * inv, o[4 * g4 + 1] * inv);
      w[1] = pack_bf16x2(o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv);
      *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * g4 + 4 * h) = w;
    }
  }
}

// Kernel selection.  -1 = automatic: the kernel follows the tile height the host built its work
// list for (64 rows: third generation; 32 * (4 / GP) rows: second generation), head-major tile
// order when sequences have many q-tiles.  Fixed choices are for A/B timing and tests:
// 0 = first generation, 1 = generation 2 without wave priorities, 2 = generation 2,
// 3 = generation 2 head-major, 4 = generation 3, 5 = generation 3 head-major, 6 / 7 = as 4 / 5
// without the instruction-interleave hints, 8 = generation 4 with K/V tiles staged through
// registers, 9 = the same head-major, 10 / 11 = generation 4 with K/V tiles loaded straight into
// LDS (what the automatic choice takes); 12 / 13 / 14 = timing experiments on 11 with wrong results
// (no barrier / no loads / no exponentials inside the tile loop).
static int g_prefill_variant = -1;
VPSA_EXPORT int vpsa_set_prefill_variant(int v) {
  const int old = g_prefill_variant;
  g_prefill_variant = v;
  return old;
}

// Query rows per work-list entry for a GQA ratio and kernel selection (the host builds its tile
// list with this height and passes it back to vpsa_paged_prefill).
VPSA_EXPORT int vpsa_prefill_tile_rows(int n_q, int n_kv, int want64) {
  if (n_kv <= 0 || n_q % n_kv != 0) return -1;
  const int G = n_q / n_kv;
  const int GP = G < 4 ? G : 4;
  const int v = g_prefill_variant;
  const bool gen3_ok = (G % 4) == 0;
  if (gen3_ok && (v >= 4 || (v < 0 && want64))) return 64;
  // the fourth generation pads the last head group of other ratios with spare waves
  if (G >= 3 && (v >= 8 || (v < 0 && want64))) return 64;
  return 32 * (4 / GP);
}

// 1 when the kernel that vpsa_paged_prefill would pick for this head layout and tile height can
// apply the rotary embedding to its query rows itself (vpsa_paged_prefill_rope).
VPSA_EXPORT int vpsa_prefill_fuses_rope(int n_q, int n_kv, int tile_rows) {
  if (n_kv <= 0 || n_q % n_kv != 0) return 0;
  const int v = g_prefill_variant;
  return tile_rows == 64 && n_q / n_kv >= 3 && (v < 0 || v >= 8);
}

// rope: null = q is already rotated; else the [position][HEAD_DIM] fp32 table (cos of the 64
// channel pairs, then sin), and q holds UNROTATED rows at consecutive positions ending at each
// sequence's context length.  Only the fourth-generation kernels take it (-4 otherwise).
// tile_split: null, or [num_tiles][3] = (first key tile, end key tile, partial record or -1) per
// work-list entry; entries with a record leave partials in split_ws (SPLIT_REC_FLOATS floats per
// record, head group and head) and `groups` [num_groups][4] = (sequence, first row, first record,
// number of records; 0 records = padding) says which records make up a tile: they are merged by a
// second launch.  Fourth-generation kernels only (-4 otherwise).
VPSA_EXPORT int vpsa_paged_prefill_split(const void* q, long q_stride, void* out, long out_stride,
                                         const void* cache, long page_stride, long v_offset,
                                         const int* block_tables, int max_blocks, const int* cu_q,
                                         const int* ctx_lens, const int* tile_seq, const int* tile_q0,
                                         int num_tiles, int num_seqs, int tile_rows, int n_q, int n_kv,
                                         float scale, const float* rope, const int* tile_split,
                                         float* split_ws, const int* groups, int num_groups,
                                         hipStream_t stream);

VPSA_EXPORT long vpsa_prefill_split_record_bytes(int n_q, int n_kv) {
  // one record per (partial, head group of 4, wave)
  if (n_kv <= 0 || n_q % n_kv != 0) return -1;
  return (long)n_kv * ((n_q / n_kv + 3) / 4) * 4 * SPLIT_REC_FLOATS * 4;
}

VPSA_EXPORT int vpsa_paged_prefill_rope(const void* q, long q_stride, void* out, long out_stride,
                                        const void* cache, long page_stride, long v_offset,
                                        const int* block_tables, int max_blocks, const int* cu_q,
                                        const int* ctx_lens, const int* tile_seq, const int* tile_q0,
                                        int num_tiles, int num_seqs, int tile_rows, int n_q, int n_kv,
                                        float scale, const float* rope, hipStream_t stream) {
  return vpsa_paged_prefill_split(q, q_stride, out, out_stride, cache, page_stride, v_offset, block_tables,
                                  max_blocks, cu_q, ctx_lens, tile_seq, tile_q0, num_tiles, num_seqs, tile_rows,
                                  n_q, n_kv, scale, rope, nullptr, nullptr, nullptr, 0, stream);
}

VPSA_EXPORT int vpsa_paged_prefill_split(const void* q, long q_stride, void* out, long out_stride,
                                         const void* cache, long page_stride, long v_offset,
                                         const int* block_tables, int max_blocks, const int* cu_q,
                                         const int* ctx_lens, const int* tile_seq, const int* tile_q0,
                                         int num_tiles, int num_seqs, int tile_rows, int n_q, int n_kv,
                                         float scale, const float* rope, const int* tile_split,
                                         float* split_ws, const int* groups, int num_groups,
                                         hipStream_t stream) {
  if (num_tiles <= 0) return 0;
  if (tile_split != nullptr && (split_ws == nullptr || groups == nullptr)) return -1;
  if (rope != nullptr && !vpsa_prefill_fuses_rope(n_q, n_kv, tile_rows)) return -4;
  if (n_q % n_kv != 0) return -1;
  const int G = n_q / n_kv;
  const int GP = G < 4 ? G : 4;
  const float sl2 = scale * 1.4426950408889634f;
  int variant = g_prefill_variant;
  const bool many_tiles = num_seqs > 0 && num_tiles * (tile_rows / 32) >= 8 * num_seqs;
#define VPSA_PREFILL_ARGS                                                                      \
  (const uint16_t*)q, q_stride, (uint16_t*)out, out_stride, (const uint16_t*)cache,             \
      page_stride, v_offset, block_tables, max_blocks, cu_q, ctx_lens, tile_seq, tile_q0, n_q,  \
      n_kv, sl2
  if (tile_rows == 64 && G >= 3) {  // (a ratio of 2 also has 64-row tiles: two waves per head)
    if (variant >= 0 && variant < 4) return -2;
    if (G % 4 != 0 && variant >= 4 && variant < 8) return -2;  // the third generation has no spare-wave path
    static bool attr_set = false;
    if (!attr_set) {
      const void* fns[11] = {(const void*)paged_prefill_kernel_v4<true, true, 1>,
                             (const void*)paged_prefill_kernel_v4<true, true, 2>,
                             (const void*)paged_prefill_kernel_v4<true, true, 4>,
                             (const void*)paged_prefill_kernel_v3<false, 1>,
                             (const void*)paged_prefill_kernel_v3<true, 1>,
                             (const void*)paged_prefill_kernel_v3<false, 0>,
                             (const void*)paged_prefill_kernel_v3<true, 0>,
                             (const void*)paged_prefill_kernel_v4<false, false>,
                             (const void*)paged_prefill_kernel_v4<true, false>,
                             (const void*)paged_prefill_kernel_v4<false, true>,
                             (const void*)paged_prefill_kernel_v4<true, true>};
      for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 98304) != hipSuccess)
          return -3;
      attr_set = true;
    }
    dim3 grid(num_tiles, n_kv * ((G + 3) / 4)), block(256);
    if (variant < 0) variant = many_tiles ? 11 : 10;
    if (tile_split != nullptr && variant < 8) return -4;
    if (variant >= 12 && variant <= 14) {
      if (variant == 12)
        hipLaunchKernelGGL((paged_prefill_kernel_v4<true, true, 1>), grid, block, 98304, stream, VPSA_PREFILL_ARGS, rope, tile_split, split_ws);
      else if (variant == 13)
        hipLaunchKernelGGL((paged_prefill_kernel_v4<true, true, 2>), grid, block, 98304, stream, VPSA_PREFILL_ARGS, rope, tile_split, split_ws);
      else
        hipLaunchKernelGGL((paged_prefill_kernel_v4<true, true, 4>), grid, block, 98304, stream, VPSA_PREFILL_ARGS, rope, tile_split, split_ws);
    } else if (variant == 11) {
      hipLaunchKernelGGL((paged_prefill_kernel_v4<true, true>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS, rope, tile_split, split_ws);
    } else if (variant == 10) {
      hipLaunchKernelGGL((paged_prefill_kernel_v4<false, true>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS, rope, tile_split, split_ws);
    } else if (variant == 9) {
      hipLaunchKernelGGL((paged_prefill_kernel_v4<true, false>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS, rope, tile_split, split_ws);
    } else if (variant == 8) {
      hipLaunchKernelGGL((paged_prefill_kernel_v4<false, false>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS, rope, tile_split, split_ws);
    } else if (variant == 5) {
      hipLaunchKernelGGL((paged_prefill_kernel_v3<true, 1>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS);
    } else if (variant == 4) {
      hipLaunchKernelGGL((paged_prefill_kernel_v3<false, 1>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS);
    } else if (variant == 7) {
      hipLaunchKernelGGL((paged_prefill_kernel_v3<true, 0>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS);
    } else {
      hipLaunchKernelGGL((paged_prefill_kernel_v3<false, 0>), grid, block, 98304, stream,
                         VPSA_PREFILL_ARGS);
    }
    if (tile_split != nullptr && num_groups > 0)
      hipLaunchKernelGGL(prefill_combine_kernel, dim3(num_groups, n_kv * ((G + 3) / 4), 8), block, 0, stream,
                         (const float*)split_ws, groups, (uint16_t*)out, out_stride, cu_q, n_q, n_kv);
    return 0;
  }
  if (tile_split != nullptr) return -4;
  if (G == 3) return -1;  // 4 waves = GP heads x (4 / GP) row blocks: GP must divide 4 (64-row tiles take a ratio of 3)
  if (tile_rows != 32 * (4 / GP) || variant >= 4) return -2;
  dim3 grid(num_tiles, n_kv * ((G + GP - 1) / GP)), block(256);
  if (variant < 0) variant = many_tiles ? 3 : 2;
  if (variant == 0) {
    hipLaunchKernelGGL(paged_prefill_kernel, grid, block, 32768, stream, VPSA_PREFILL_ARGS);
  } else if (variant == 3) {
    hipLaunchKernelGGL((paged_prefill_kernel_v2<true, true>), grid, block, 65536, stream,
                       VPSA_PREFILL_ARGS);
  } else if (variant == 2) {
    hipLaunchKernelGGL((paged_prefill_kernel_v2<true, false>), grid, block, 65536, stream,
                       VPSA_PREFILL_ARGS);
  } else {
    hipLaunchKernelGGL((paged_prefill_kernel_v2<false, false>), grid, block, 65536, stream,
                       VPSA_PREFILL_ARGS);
  }
#undef VPSA_PREFILL_ARGS
  return 0;
}

VPSA_EXPORT int vpsa_paged_prefill(const void* q, long q_stride, void* out, long out_stride,
                                   const void* cache, long page_stride, long v_offset,
                                   const int* block_tables, int max_blocks, const int* cu_q,
                                   const int* ctx_lens, const int* tile_seq, const int* tile_q0,
                                   int num_tiles, int num_seqs, int tile_rows, int n_q, int n_kv,
                                   float scale, hipStream_t stream) {
  return vpsa_paged_prefill_rope(q, q_stride, out, out_stride, cache, page_stride, v_offset, block_tables,
                                 max_blocks, cu_q, ctx_lens, tile_seq, tile_q0, num_tiles, num_seqs,
                                 tile_rows, n_q, n_kv, scale, nullptr, stream);
}

// ---------------------------------------------------------------------------------
// Decode: one query token per sequence.
//   q, out  [num_seqs, n_q * 128] bf16
//   ws_o    [num_seqs, n_q, num_splits, 128] fp32 (unnormalised), ws_ml [.., 2] = (max, sum)
// grid = (num_seqs, n_kv, num_splits); block = 256 (4 waves, interleaved over 32-key chunks).
// Requires G = n_q / n_kv <= 16.
// ---------------------------------------------------------------------------------
// MODE 0: one register set per wave; 1: software-pipelined over two chunks four apart; 2: a
// wave takes both 32-key halves of a page together, so the two halves of every 128-byte V line
// (64 keys of one channel) are requested back to back by the same wave instead of by two waves
// that may have drifted apart by more than the caches hold.
template <bool NT>
__device__ __forceinline__ u32x4 ld16(const uint16_t* p) {
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
  return NT ? __builtin_nontemporal_load(q) : *q;
}

// NT: K/V fragments are fetched with non-temporal loads.  Every byte of a context is read once
// per step and is gone from the caches long before the next one: the streaming policy keeps it
// from displacing anything and reads the card's HBM 10-12 % faster than plain loads do
// (`benchmarks/roof_probe.py`: 6.9-7.1 against 6.2-6.3 TB/s).
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void paged_decode_kernel(
    const uint16_t* __restrict__ q, long q_stride, uint16_t* __restrict__ out, long out_stride,
    const uint16_t* __restrict__ cache, long page_stride, long v_offset,
    const int* __restrict__ block_tables, int max_blocks, const int* __restrict__ ctx_lens,
    float* __restrict__ ws_o, float* __restrict__ ws_ml, int n_q, int n_kv, int num_splits,
    float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* lds_o = reinterpret_cast<float*>(lds);               // [4][128][16]
  float* lds_m = reinterpret_cast<float*>(lds + 32768);       // [4][16]
  float* lds_l = reinterpret_cast<float*>(lds + 32768 + 256); // [4][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15;
  const int g = lane >> 4;

  const int seq = blockIdx.x;
  const int kvh = blockIdx.y;
  const int split = blockIdx.z;
  const int G = n_q / n_kv;
  const int ctx = ctx_lens[seq];

  const int nchunks = (ctx + 31) >> 5;
  int per_split = (nchunks + num_splits - 1) / num_splits;
  if (MODE == 2) per_split = (per_split + 1) & ~1;  // splits begin on page boundaries
  const int c_begin = split * per_split;
  int c_end = c_begin + per_split;
  if (c_end > nchunks) c_end = nchunks;

  // Q fragments: B[k = 32 ks + 8 g + j][col = query head c16]
  bf16x8 qf[4];
  {
    const bool ok = c16 < G;
    const uint16_t* qp =
        q + (long)seq * q_stride + (long)(kvh * G + (ok ? c16 : 0)) * HEAD_DIM + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 32 * ks);
      if (!ok) v = (u32x4){0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }

  f32x4 o_acc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o_acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  const int* bt = block_tables + (long)seq * max_blocks;
  // MFMA row rho = c16 of key sub-tile t reads key 8*(rho>>2) + 4t + (rho&3)
  const int krow_base = 8 * (c16 >> 2) + (c16 & 3);

  auto load_chunk = [&](int c, u32x4 (&kf)[2][4], u32x4 (&vf)[8]) {
    const long page = bt[c >> 1];
    const int koff = (c & 1) * 32;
    const uint16_t* kp = cache + page * page_stride + (long)kvh * PAGE * HEAD_DIM;
    const uint16_t* vp = cache + page * page_stride + v_offset + (long)kvh * HEAD_DIM * PAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        kf[t][ks] = ld16<NT>(kp + ks * (PAGE * 32) + (koff + krow_base + 4 * t) * 32 + 8 * g);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
      vf[dt] = ld16<NT>(vp + (koff >> 5) * (HEAD_DIM * 32) + (16 * dt + c16) * 32 + 8 * g);
  };
  auto compute_chunk = [&](int c, u32x4 (&kf)[2][4], u32x4 (&vf)[8]) {
    f32x4 s_acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s_acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        s_acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[t][ks]),
                                                           qf[ks], s_acc[t], 0, 0, 0);
    }
    // accumulator reg of sub-tile t, lane group g = key c*32 + 8g + 4t + reg
    float mx = NEG_BIG;
    const int key0 = c * 32 + 8 * g;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = s_acc[t][i] * scale_log2e;
        if (key0 + 4 * t + i >= ctx) v = NEG_BIG;
        s_acc[t][i] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    u32x4 pk;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float p0 = __builtin_amdgcn_exp2f(s_acc[t][2 * jj] - m_new);
        float p1 = __builtin_amdgcn_exp2f(s_acc[t][2 * jj + 1] - m_new);
        // keys past the context contribute exactly zero even when the whole chunk is masked
        if (key0 + 4 * t + 2 * jj >= ctx) p0 = 0.f;
        if (key0 + 4 * t + 2 * jj + 1 >= ctx) p1 = 0.f;
        psum += p0 + p1;
        pk[t * 2 + jj] = pack_bf16x2(p0, p1);
      }
    l_run = l_run * alpha + psum;
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      o_acc[dt] *= alpha;
      o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf[dt]), pf,
                                                         o_acc[dt], 0, 0, 0);
    }
  };

  if (MODE == 2) {
    u32x4 kfa[2][4], vfa[8], kfb[2][4], vfb[8];
    for (int c = c_begin + 2 * wave; c < c_end; c += 8) {
      const bool two = c + 1 < c_end;
      load_chunk(c, kfa, vfa);
      if (two) load_chunk(c + 1, kfb, vfb);
      compute_chunk(c, kfa, vfa);
      if (two) compute_chunk(c + 1, kfb, vfb);
    }
  } else if (MODE == 1) {
    // two register sets: the loads of the wave's next chunk are in flight while the current
    // one is multiplied (the wave count per SIMD drops, each wave keeps twice the bytes in flight)
    u32x4 kfa[2][4], vfa[8], kfb[2][4], vfb[8];
    int c = c_begin + wave;
    if (c < c_end) load_chunk(c, kfa, vfa);
    for (; c < c_end; c += 8) {
      if (c + 4 < c_end) load_chunk(c + 4, kfb, vfb);
      compute_chunk(c, kfa, vfa);
      if (c + 4 < c_end) {
        if (c + 8 < c_end) load_chunk(c + 8, kfa, vfa);
        compute_chunk(c + 4, kfb, vfb);
      }
    }
  } else {
    u32x4 kf[2][4], vf[8];
    for (int c = c_begin + wave; c < c_end; c += 4) {
      load_chunk(c, kf, vf);
      compute_chunk(c, kf, vf);
    }
  }

  // combine the four lane groups' partial sums, then the four waves through LDS
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_o[(wave * 128 + 16 * dt + 4 * g + i) * 16 + c16] = o_acc[dt][i];
  if (g == 0) {
    lds_m[wave * 16 + c16] = m_run;
    lds_l[wave * 16 + c16] = l_run;
  }
  __syncthreads();

  const int qh = tid & 15;
  const int dgrp = tid >> 4;  // 16 groups of 8 channels
  if (qh < G) {
    float M = NEG_BIG;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, lds_m[w * 16 + qh]);
    float L = 0.f;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __builtin_amdgcn_exp2f(lds_m[w * 16 + qh] - M);
      L += lds_l[w * 16 + qh] * f;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += lds_o[(w * 128 + dgrp * 8 + k) * 16 + qh] * f;
    }
    const int head = kvh * G + qh;
    if (num_splits == 1) {
      const float inv = L > 0.f ? 1.f / L : 0.f;
      u32x4 w4;
#pragma unroll
      for (int k = 0; k < 4; ++k) w4[k] = pack_bf16x2(acc[2 * k] * inv, acc[2 * k + 1] * inv);
      *reinterpret_cast<u32x4*>(out + (long)seq * out_stride + head * HEAD_DIM + dgrp * 8) = w4;
    } else {
      const long base = ((long)seq * n_q + head) * num_splits + split;
      float* wo = ws_o + base * HEAD_DIM + dgrp * 8;
      *reinterpret_cast<f32x4*>(wo) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(wo + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
      if (dgrp == 0) {
        ws_ml[base * 2] = M;
        ws_ml[base * 2 + 1] = L;
      }
    }
  }
}

// merge split partials: grid = (num_seqs, n_q), block = 128 (one channel per thread)
__global__ __launch_bounds__(128) void decode_reduce_kernel(const float* __restrict__ ws_o,
                                                            const float* __restrict__ ws_ml,
                                                            uint16_t* __restrict__ out,
                                                            long out_stride, int n_q,
                                                            int num_splits) {
  const int seq = blockIdx.x, head = blockIdx.y, d = threadIdx.x;
  const long base = ((long)seq * n_q + head) * num_splits;
  float M = NEG_BIG;
  for (int s = 0; s < num_splits; ++s) M = fmaxf(M, ws_ml[(base + s) * 2]);
  float L = 0.f, acc = 0.f;
  for (int s = 0; s < num_splits; ++s) {
    const float f = __builtin_amdgcn_exp2f(ws_ml[(base + s) * 2] - M);
    L += ws_ml[(base + s) * 2 + 1] * f;
    acc += ws_o[(base + s) * HEAD_DIM + d] * f;
  }
  const float r = L > 0.f ? acc / L : 0.f;
  bf16_t b = (bf16_t)r;
  out[(long)seq * out_stride + head * HEAD_DIM + d] = __builtin_bit_cast(uint16_t, b);
}

// 0 = one register set per wave, 1 = software-pipelined (two register sets), 2 = page-paired,
// -1 = automatic: pipelined from three context splits on (measured at 40 sequences x 21.5k keys:
// 5.95 against 5.69 TB/s at four splits; the plain kernel is faster with one split, 5.68 against 5.19).
static int g_decode_variant = -1;
VPSA_EXPORT int vpsa_set_decode_variant(int v) {
  const int old = g_decode_variant;
  g_decode_variant = v;
  return old;
}

VPSA_EXPORT int vpsa_paged_decode(const void* q, long q_stride, void* out, long out_stride,
                                  const void* cache, long page_stride, long v_offset,
                                  const int* block_tables, int max_blocks, const int* ctx_lens,
                                  void* ws_o, void* ws_ml, int num_seqs, int n_q, int n_kv,
                                  int num_splits, float scale, hipStream_t stream) {
  if (num_seqs <= 0) return 0;
  if (n_q % n_kv != 0 || n_q / n_kv > 16 || num_splits < 1) return -1;
  if (num_splits > 1 && (!ws_o || !ws_ml)) return -1;
  dim3 grid(num_seqs, n_kv, num_splits), block(256);
  // variants 0-2: MODE with plain loads (kept for A/B), 3-5: MODE 0-2 with non-temporal loads
  const int variant = g_decode_variant >= 0 ? g_decode_variant : (num_splits >= 3 ? 4 : 3);
#define VPSA_DECODE(MODE, NT)                                                                       \
  hipLaunchKernelGGL((paged_decode_kernel<MODE, NT>), grid, block, 32768 + 512, stream,             \
                     (const uint16_t*)q, q_stride, (uint16_t*)out, out_stride,                      \
                     (const uint16_t*)cache, page_stride, v_offset, block_tables, max_blocks,       \
                     ctx_lens, (float*)ws_o, (float*)ws_ml, n_q, n_kv, num_splits,                  \
                     scale * 1.4426950408889634f)
  switch (variant) {
    case 0: VPSA_DECODE(0, false); break;
    case 1: VPSA_DECODE(1, false); break;
    case 2: VPSA_DECODE(2, false); break;
    case 3: VPSA_DECODE(0, true); break;
    case 5: VPSA_DECODE(2, true); break;
    default: VPSA_DECODE(1, true); break;
  }
#undef VPSA_DECODE
  if (num_splits > 1) {
    hipLaunchKernelGGL(decode_reduce_kernel, dim3(num_seqs, n_q), dim3(128), 0, stream,
                       (const float*)ws_o, (const float*)ws_ml, (uint16_t*)out, out_stride, n_q,
                       num_splits);
  }
  return 0;
}
