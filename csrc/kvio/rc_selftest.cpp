// Self-test of the host range coder, meant to be built with -fsanitize=address,undefined:
// round trips over the symbol distributions the KV codec produces, then every kind of input a
// remote cache or a disk can hand back (truncated, empty, random, the right bytes with wrong
// parameters).  The decoder may return wrong symbols for bad input; it may not touch memory
// outside its buffers, loop forever or report success for a stream that ended early.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "rangecoder.hpp"

static int failures = 0;
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                    \
    }                                                                \
  } while (0)

// buffers sized exactly, on the heap: one byte out of bounds is a sanitizer report
static std::vector<uint8_t> encode(const std::vector<int8_t>& sym, size_t period, int n_ctx) {
  std::vector<uint8_t> out(sym.size() + sym.size() / 8 + 64);
  int64_t n = vrc::encode(sym.data(), sym.size(), out.data(), out.size(), period, n_ctx);
  CHECK(n >= 0);
  out.resize(n < 0 ? 0 : (size_t)n);
  out.shrink_to_fit();
  return out;
}

int main() {
  std::mt19937 rng(1234);
  // 1. round trips: uniform, narrow (few levels), constant, all 256 values, empty, one symbol
  struct Case { size_t n; int lo, hi; size_t period; int n_ctx; };
  const Case cases[] = {{0, 0, 0, 0, 1},       {1, -128, -128, 0, 1},   {1, 127, 127, 7, 3},
                        {4096, -128, 127, 512, 8}, {65536, -15, 15, 8192, 4}, {10000, 0, 0, 1, 4096},
                        {3001, -2, 2, 1000, 0}, {50000, -128, 127, 1, 100000}};
  for (const Case& c : cases) {
    std::vector<int8_t> sym(c.n);
    std::uniform_int_distribution<int> d(c.lo, c.hi);
    for (auto& s : sym) s = (int8_t)d(rng);
    std::vector<uint8_t> coded = encode(sym, c.period, c.n_ctx);
    std::vector<int8_t> back(c.n, 99);
    int rc = vrc::decode(coded.data(), coded.size(), back.data(), back.size(), c.period, c.n_ctx);
    CHECK(rc == 0);
    CHECK(back == sym);
    if (c.hi - c.lo < 40 && c.n > 1000 && c.n_ctx <= 8) CHECK(coded.size() < c.n * 3 / 4);  // narrow alphabets do compress
    // 2. output buffer too small: -1, nothing written past `cap`
    if (coded.size() > 8) {
      std::vector<uint8_t> small(coded.size() / 2);
      CHECK(vrc::encode(sym.data(), sym.size(), small.data(), small.size(), c.period, c.n_ctx) == -1);
      CHECK(vrc::encode(sym.data(), sym.size(), nullptr, 0, c.period, c.n_ctx) == -1);
    }
    // 3. every truncation of a short stream, a sample of truncations of a long one
    for (size_t cut = 0; cut < coded.size(); cut += (coded.size() > 300 ? coded.size() / 97 + 1 : 1)) {
      std::vector<uint8_t> part(coded.begin(), coded.begin() + (long)cut);
      std::vector<int8_t> out(c.n);
      int r = vrc::decode(part.data(), part.size(), out.data(), out.size(), c.period, c.n_ctx);
      if (c.n > 16 && cut + 8 < coded.size()) CHECK(r == -1);
    }
    // 4. the right bytes decoded with the wrong geometry
    if (c.n > 0) {
      std::vector<int8_t> out(c.n * 2 + 5);
      (void)vrc::decode(coded.data(), coded.size(), out.data(), out.size(), c.period + 3, c.n_ctx + 1);
      (void)vrc::decode(coded.data(), coded.size(), out.data(), 1, 0, -5);
    }
  }
  // 5. bytes that never were a stream: random, all 0x00, all 0xFF, lengths 0..5 (shorter than the
  // decoder's own preamble)
  for (int round = 0; round < 200; ++round) {
    size_t len = round < 8 ? (size_t)round : (size_t)(rng() % 5000);
    std::vector<uint8_t> junk(len);
    for (auto& b : junk) b = round % 3 == 0 ? (uint8_t)rng() : (round % 3 == 1 ? 0x00 : 0xFF);
    size_t n = rng() % 20000;
    std::vector<int8_t> out(n);
    (void)vrc::decode(junk.data(), junk.size(), out.data(), out.size(), rng() % 4096, (int)(rng() % 64) - 2);
  }
  {  // a null input of length 0 asked for many symbols: -1 at once, output zero-filled
    std::vector<int8_t> out(1 << 20, 7);
    CHECK(vrc::decode(nullptr, 0, out.data(), out.size(), 0, 1) == -1);
    CHECK(out.back() == 0);
  }
  // 6. one flipped bit anywhere in a valid stream: decodes to something, inside its buffers
  {
    std::vector<int8_t> sym(20000);
    std::uniform_int_distribution<int> d(-31, 31);
    for (auto& s : sym) s = (int8_t)d(rng);
    std::vector<uint8_t> coded = encode(sym, 5000, 4);
    for (int k = 0; k < 64; ++k) {
      std::vector<uint8_t> bad = coded;
      bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
      std::vector<int8_t> out(sym.size());
      (void)vrc::decode(bad.data(), bad.size(), out.data(), out.size(), 5000, 4);
    }
  }
  if (failures) {
    fprintf(stderr, "%d checks failed\n", failures);
    return 1;
  }
  printf("rangecoder selftest ok\n");
  return 0;
}
