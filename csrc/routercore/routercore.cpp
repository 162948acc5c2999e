#include "routercore.h"

#include <algorithm>
#include <cstring>
#include <mutex>

namespace vpsa {

// ------------------------------------------------------------------ xxHash64
namespace {
constexpr uint64_t P1 = 11400714785074694791ULL;
constexpr uint64_t P2 = 14029467366897019727ULL;
constexpr uint64_t P3 = 1609587929392839161ULL;
constexpr uint64_t P4 = 9650029242287828579ULL;
constexpr uint64_t P5 = 2870177450012600261ULL;

inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t* p) {
  uint64_t v;
  std::memcpy(&v, p, 8);
  return v;
}
inline uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  std::memcpy(&v, p, 4);
  return v;
}
inline uint64_t round1(uint64_t acc, uint64_t in) {
  acc += in * P2;
  acc = rotl(acc, 31);
  return acc * P1;
}
inline uint64_t merge(uint64_t acc, uint64_t v) {
  acc ^= round1(0, v);
  return acc * P1 + P4;
}
}  // namespace

uint64_t xxh64(const void* data, size_t len, uint64_t seed) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  const uint8_t* end = p + len;
  uint64_t h;
  if (len >= 32) {
    const uint8_t* limit = end - 32;
    uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    do {
      v1 = round1(v1, rd64(p));
      v2 = round1(v2, rd64(p + 8));
      v3 = round1(v3, rd64(p + 16));
      v4 = round1(v4, rd64(p + 24));
      p += 32;
    } while (p <= limit);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = merge(h, v1);
    h = merge(h, v2);
    h = merge(h, v3);
    h = merge(h, v4);
  } else {
    h = seed + P5;
  }
  h += (uint64_t)len;
  while (p + 8 <= end) {
    h ^= round1(0, rd64(p));
    h = rotl(h, 27) * P1 + P4;
    p += 8;
  }
  if (p + 4 <= end) {
    h ^= (uint64_t)rd32(p) * P1;
    h = rotl(h, 23) * P2 + P3;
    p += 4;
  }
  while (p < end) {
    h ^= (*p) * P5;
    h = rotl(h, 11) * P1;
    ++p;
  }
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h;
}

// ------------------------------------------------------------------ HashTrie
// Chunks are `chunk_` UTF-8 *characters* long so that the native trie and a Python
// implementation slicing a str agree on chunk boundaries for non-ASCII prompts.
static size_t advance_chars(const char* d, size_t len, size_t pos, size_t nchars) {
  size_t c = 0;
  while (pos < len && c < nchars) {
    unsigned char b = (unsigned char)d[pos];
    size_t step = b < 0x80 ? 1 : (b >> 5) == 0x6 ? 2 : (b >> 4) == 0xE ? 3 : (b >> 3) == 0x1E ? 4 : 1;
    pos += step;
    ++c;
  }
  return pos > len ? len : pos;
}

void HashTrie::insert(const char* data, size_t len, uint32_t endpoint) {
  std::unique_lock<std::shared_mutex> lk(mu_);
  uint64_t now = ++clock_;
  Node* node = root_.get();
  node->endpoints.add(endpoint);
  size_t pos = 0;
  while (pos < len) {
    size_t nxt = advance_chars(data, len, pos, chunk_);
    uint64_t h = xxh64(data + pos, nxt - pos);
    auto it = node->children.find(h);
    if (it == node->children.end()) {
      it = node->children.emplace(h, std::make_unique<Node>()).first;
      ++nodes_;
    }
    node = it->second.get();
    node->endpoints.add(endpoint);
    node->last_used = now;
    pos = nxt;
  }
  evict_if_needed();
}

size_t HashTrie::longest_prefix_match(const char* data, size_t len, const EndpointSet& available,
                                      EndpointSet* selected) const {
  std::shared_lock<std::shared_mutex> lk(mu_);
  const Node* node = root_.get();
  EndpointSet cur = available;
  size_t matched = 0, pos = 0;
  while (pos < len) {
    size_t nxt = advance_chars(data, len, pos, chunk_);
    uint64_t h = xxh64(data + pos, nxt - pos);
    auto it = node->children.find(h);
    if (it == node->children.end()) break;
    node = it->second.get();
    EndpointSet inter = node->endpoints.intersect(cur);
    if (!inter.any()) break;
    matched += chunk_;
    cur = std::move(inter);
    pos = nxt;
  }
  if (selected) *selected = std::move(cur);
  return matched;
}

void HashTrie::remove_rec(Node* n, uint32_t endpoint) {
  n->endpoints.remove(endpoint);
  for (auto it = n->children.begin(); it != n->children.end();) {
    remove_rec(it->second.get(), endpoint);
    if (!it->second->endpoints.any()) {
      // dropping a subtree: count its nodes
      std::vector<Node*> st{it->second.get()};
      size_t cnt = 0;
      while (!st.empty()) {
        Node* x = st.back();
        st.pop_back();
        ++cnt;
        for (auto& c : x->children) st.push_back(c.second.get());
      }
      nodes_ -= cnt;
      it = n->children.erase(it);
    } else {
      ++it;
    }
  }
}

void HashTrie::remove_endpoint(uint32_t endpoint) {
  std::unique_lock<std::shared_mutex> lk(mu_);
  remove_rec(root_.get(), endpoint);
}

size_t HashTrie::prune_older(Node* n, uint64_t cutoff) {
  size_t removed = 0;
  for (auto it = n->children.begin(); it != n->children.end();) {
    Node* c = it->second.get();
    if (c->last_used < cutoff) {
      std::vector<Node*> st{c};
      while (!st.empty()) {
        Node* x = st.back();
        st.pop_back();
        ++removed;
        for (auto& cc : x->children) st.push_back(cc.second.get());
      }
      it = n->children.erase(it);
    } else {
      removed += prune_older(c, cutoff);
      ++it;
    }
  }
  return removed;
}

// Bounded memory: when the node budget is exceeded, drop every subtree not touched by the
// most recent half of the insertions (the reference trie grows without bound).
void HashTrie::evict_if_needed() {
  if (max_nodes_ == 0 || nodes_.load() <= max_nodes_) return;
  uint64_t now = clock_.load();
  uint64_t window = now / 2;
  while (nodes_.load() > max_nodes_ && window > 0) {
    size_t removed = prune_older(root_.get(), now - window);
    nodes_ -= removed;
    window /= 2;
  }
}

// ------------------------------------------------------------------ HashRing
void HashRing::add_node(const std::string& name) {
  std::unique_lock<std::shared_mutex> lk(mu_);
  if (members_.count(name)) return;
  members_[name] = vnodes_;
  for (int i = 0; i < vnodes_; ++i) {
    std::string k = name + "#" + std::to_string(i);
    uint64_t h = xxh64(k.data(), k.size());
    // on the (vanishingly rare) collision the lexicographically smaller owner wins, so the
    // ring does not depend on insertion order
    auto it = ring_.find(h);
    if (it == ring_.end() || name < it->second) ring_[h] = name;
  }
}

void HashRing::remove_node(const std::string& name) {
  std::unique_lock<std::shared_mutex> lk(mu_);
  if (!members_.erase(name)) return;
  for (auto it = ring_.begin(); it != ring_.end();) {
    if (it->second == name)
      it = ring_.erase(it);
    else
      ++it;
  }
}

bool HashRing::has_node(const std::string& name) const {
  std::shared_lock<std::shared_mutex> lk(mu_);
  return members_.count(name) > 0;
}

std::vector<std::string> HashRing::nodes() const {
  std::shared_lock<std::shared_mutex> lk(mu_);
  std::vector<std::string> out;
  for (auto& m : members_) out.push_back(m.first);
  return out;
}

std::string HashRing::get_node(const std::string& key) const {
  std::shared_lock<std::shared_mutex> lk(mu_);
  if (ring_.empty()) return std::string();
  uint64_t h = xxh64(key.data(), key.size());
  auto it = ring_.lower_bound(h);
  if (it == ring_.end()) it = ring_.begin();
  return it->second;
}

}  // namespace vpsa

// ------------------------------------------------------------------ C ABI (ctypes)
using namespace vpsa;
#define API extern "C" __attribute__((visibility("default")))

API uint64_t vpsa_xxh64(const char* data, size_t len, uint64_t seed) { return xxh64(data, len, seed); }

API void* vpsa_trie_new(size_t chunk, size_t max_nodes) { return new HashTrie(chunk, max_nodes); }
API void vpsa_trie_free(void* t) { delete static_cast<HashTrie*>(t); }
API void vpsa_trie_insert(void* t, const char* data, size_t len, uint32_t ep) {
  static_cast<HashTrie*>(t)->insert(data, len, ep);
}
API size_t vpsa_trie_nodes(void* t) { return static_cast<HashTrie*>(t)->node_count(); }
API void vpsa_trie_remove_endpoint(void* t, uint32_t ep) {
  static_cast<HashTrie*>(t)->remove_endpoint(ep);
}
// avail / out are arrays of endpoint ids; returns matched length, *n_out = selected count
API size_t vpsa_trie_match(void* t, const char* data, size_t len, const uint32_t* avail,
                           size_t n_avail, uint32_t* out, size_t* n_out) {
  EndpointSet a, sel;
  for (size_t i = 0; i < n_avail; ++i) a.add(avail[i]);
  size_t m = static_cast<HashTrie*>(t)->longest_prefix_match(data, len, a, &sel);
  auto ids = sel.ids();
  size_t n = ids.size() < n_avail ? ids.size() : n_avail;
  for (size_t i = 0; i < n; ++i) out[i] = ids[i];
  *n_out = n;
  return m;
}

API void* vpsa_ring_new(int vnodes) { return new HashRing(vnodes); }
API void vpsa_ring_free(void* r) { delete static_cast<HashRing*>(r); }
API void vpsa_ring_add(void* r, const char* name) { static_cast<HashRing*>(r)->add_node(name); }
API void vpsa_ring_remove(void* r, const char* name) { static_cast<HashRing*>(r)->remove_node(name); }
API int vpsa_ring_get(void* r, const char* key, size_t key_len, char* out, size_t cap) {
  std::string n = static_cast<HashRing*>(r)->get_node(std::string(key, key_len));
  if (n.size() + 1 > cap) return -1;
  std::memcpy(out, n.c_str(), n.size() + 1);
  return (int)n.size();
}

API void* vpsa_rr_new() { return new RoundRobin(); }
API void vpsa_rr_free(void* r) { delete static_cast<RoundRobin*>(r); }
API uint64_t vpsa_rr_next(void* r, uint64_t n) { return static_cast<RoundRobin*>(r)->next(n); }
