// Adaptive binary range coder (LZMA-style rc) over int8 symbols decomposed with a per-context
// binary tree: 8 binary decisions per symbol, context = tree node.  Narrow alphabets (late layers
// quantised to few levels) cost proportionally fewer bits.  Host-only, no dependencies: the same
// header is compiled into libvpsa_kvio and into the sanitizer self-test.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace vrc {

struct Encoder {
  uint64_t low = 0;
  uint32_t range = 0xFFFFFFFFu;
  uint8_t cache = 0;
  uint64_t cache_size = 1;
  uint8_t* out = nullptr;
  size_t cap = 0, pos = 0;
  bool overflow = false;
  void put(uint8_t b) {
    if (pos < cap)
      out[pos++] = b;
    else
      overflow = true;
  }
  void shift_low() {
    if ((uint32_t)low < 0xFF000000u || (uint32_t)(low >> 32) != 0) {
      uint8_t temp = cache;
      do {
        put((uint8_t)(temp + (uint8_t)(low >> 32)));
        temp = 0xFF;
      } while (--cache_size != 0);
      cache = (uint8_t)((uint32_t)low >> 24);
    }
    cache_size++;
    low = (uint64_t)((uint32_t)low << 8);
  }
  void encode(uint16_t* prob, int bit) {
    uint32_t bound = (range >> 11) * *prob;
    if (!bit) {
      range = bound;
      *prob = (uint16_t)(*prob + ((2048 - *prob) >> 5));
    } else {
      low += bound;
      range -= bound;
      *prob = (uint16_t)(*prob - (*prob >> 5));
    }
    while (range < (1u << 24)) {
      range <<= 8;
      shift_low();
    }
  }
  void flush() {
    for (int i = 0; i < 5; ++i) shift_low();
  }
};

struct Decoder {
  uint32_t range = 0xFFFFFFFFu, code = 0;
  const uint8_t* in = nullptr;
  size_t len = 0, pos = 0;
  bool overrun = false;  // asked for a byte the input does not have
  uint8_t get() {
    if (pos < len) return in[pos++];
    overrun = true;
    return 0;
  }
  void init() {
    get();
    for (int i = 0; i < 4; ++i) code = (code << 8) | get();
  }
  int decode(uint16_t* prob) {
    uint32_t bound = (range >> 11) * *prob;
    int bit;
    if (code < bound) {
      range = bound;
      *prob = (uint16_t)(*prob + ((2048 - *prob) >> 5));
      bit = 0;
    } else {
      code -= bound;
      range -= bound;
      *prob = (uint16_t)(*prob - (*prob >> 5));
      bit = 1;
    }
    while (range < (1u << 24)) {
      range <<= 8;
      code = (code << 8) | get();
    }
    return bit;
  }
};

// Returns bytes written, or -1 when `cap` is too small.  n_ctx independent model sets are cycled
// every `ctx_period` symbols (e.g. one per layer) so each layer adapts separately.
inline int64_t encode(const int8_t* sym, size_t n, uint8_t* out, size_t cap, size_t ctx_period, int n_ctx) {
  if (n_ctx < 1) n_ctx = 1;
  if (n_ctx > 4096) n_ctx = 4096;
  if (ctx_period == 0) ctx_period = n ? n : 1;
  std::vector<uint16_t> probs((size_t)n_ctx * 256, 1024);
  Encoder e;
  e.out = out;
  e.cap = cap;
  for (size_t i = 0; i < n; ++i) {
    uint16_t* pr = probs.data() + ((i / ctx_period) % (size_t)n_ctx) * 256;
    unsigned v = (uint8_t)sym[i] ^ 0x80u;  // order symbols by value: -128..127 -> 0..255
    unsigned node = 1;
    for (int b = 7; b >= 0; --b) {
      int bit = (int)((v >> b) & 1u);
      e.encode(pr + node, bit);
      node = (node << 1) | (unsigned)bit;
    }
  }
  e.flush();
  return e.overflow ? -1 : (int64_t)e.pos;
}

// 0 = ok; -1 = the input ran out before `n` symbols were decoded (a truncated stream, or bytes
// that never were one).  Never reads outside [in, in+len) and never writes outside [sym, sym+n):
// whatever the bytes are, the worst outcome is wrong symbols and a -1.
inline int decode(const uint8_t* in, size_t len, int8_t* sym, size_t n, size_t ctx_period, int n_ctx) {
  if (n_ctx < 1) n_ctx = 1;
  if (n_ctx > 4096) n_ctx = 4096;
  if (ctx_period == 0) ctx_period = n ? n : 1;
  std::vector<uint16_t> probs((size_t)n_ctx * 256, 1024);
  Decoder d;
  d.in = in;
  d.len = len;
  d.init();
  for (size_t i = 0; i < n; ++i) {
    uint16_t* pr = probs.data() + ((i / ctx_period) % (size_t)n_ctx) * 256;
    unsigned node = 1;
    for (int b = 0; b < 8; ++b) node = (node << 1) | (unsigned)d.decode(pr + node);
    sym[i] = (int8_t)(uint8_t)((node & 0xFFu) ^ 0x80u);
    if (d.overrun) {  // nothing real left to decode: do not spin over the rest of a huge n
      for (size_t j = i + 1; j < n; ++j) sym[j] = 0;
      return -1;
    }
  }
  return 0;
}

}  // namespace vrc
