// Native routing core shared by the Python router (ctypes) and the standalone picker
// service: xxHash64, a chunk-hash prefix trie, a consistent-hash ring, round-robin.
// Semantics follow the reference's Python router and Go pickers
// (/root/reference/src/vllm_router/prefix/hashtrie.py:35-103,
//  /root/reference/src/vllm_router/routers/routing_logic.py:58-219,
//  /root/reference/src/gateway_inference_extension/prefix_aware_picker.go:25-213,
//  /root/reference/src/gateway_inference_extension/roundrobin_picker.go:32-58):
// 128-unit chunks hashed with xxh64, endpoint sets intersected along the trie path.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace vpsa {

uint64_t xxh64(const void* data, size_t len, uint64_t seed = 0);

// Dynamic bitset of endpoint ids.
struct EndpointSet {
  std::vector<uint64_t> w;
  void add(uint32_t id) {
    size_t i = id >> 6;
    if (w.size() <= i) w.resize(i + 1, 0);
    w[i] |= 1ull << (id & 63);
  }
  void remove(uint32_t id) {
    size_t i = id >> 6;
    if (i < w.size()) w[i] &= ~(1ull << (id & 63));
  }
  bool any() const {
    for (auto x : w)
      if (x) return true;
    return false;
  }
  EndpointSet intersect(const EndpointSet& o) const {
    EndpointSet r;
    size_t n = w.size() < o.w.size() ? w.size() : o.w.size();
    r.w.resize(n);
    for (size_t i = 0; i < n; ++i) r.w[i] = w[i] & o.w[i];
    return r;
  }
  std::vector<uint32_t> ids() const {
    std::vector<uint32_t> out;
    for (size_t i = 0; i < w.size(); ++i)
      for (int b = 0; b < 64; ++b)
        if (w[i] >> b & 1) out.push_back((uint32_t)(i * 64 + b));
    return out;
  }
};

class HashTrie {
 public:
  explicit HashTrie(size_t chunk_size = 128, size_t max_nodes = 0)
      : chunk_(chunk_size), max_nodes_(max_nodes), root_(new Node()) {}
  void insert(const char* data, size_t len, uint32_t endpoint);
  // Returns matched length in units; `selected` receives the surviving endpoint set
  // (== available when nothing matched).
  size_t longest_prefix_match(const char* data, size_t len, const EndpointSet& available,
                              EndpointSet* selected) const;
  void remove_endpoint(uint32_t endpoint);
  size_t node_count() const { return nodes_.load(); }

 private:
  struct Node {
    std::unordered_map<uint64_t, std::unique_ptr<Node>> children;
    EndpointSet endpoints;
    uint64_t last_used = 0;
  };
  void remove_rec(Node* n, uint32_t endpoint);
  void evict_if_needed();
  size_t prune_older(Node* n, uint64_t cutoff);
  size_t chunk_;
  size_t max_nodes_;
  std::unique_ptr<Node> root_;
  mutable std::shared_mutex mu_;
  std::atomic<size_t> nodes_{1};
  std::atomic<uint64_t> clock_{0};
};

// Consistent-hash ring with virtual nodes (ketama-style placement by xxh64).
class HashRing {
 public:
  explicit HashRing(int vnodes = 160) : vnodes_(vnodes) {}
  void add_node(const std::string& name);
  void remove_node(const std::string& name);
  bool has_node(const std::string& name) const;
  std::vector<std::string> nodes() const;
  // empty string when the ring is empty
  std::string get_node(const std::string& key) const;

 private:
  int vnodes_;
  std::map<uint64_t, std::string> ring_;
  std::map<std::string, int> members_;
  mutable std::shared_mutex mu_;
};

class RoundRobin {
 public:
  // index into a list of n sorted candidates
  uint64_t next(uint64_t n) { return n ? counter_.fetch_add(1) % n : 0; }

 private:
  std::atomic<uint64_t> counter_{0};
};

}  // namespace vpsa
