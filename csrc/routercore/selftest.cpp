// Concurrency self-test of the routing core, meant to be built with -fsanitize=address,undefined
// and with -fsanitize=thread: router worker threads and the picker service hit one trie / ring /
// counter concurrently, so readers (longest_prefix_match, get_node) must be safe against writers
// (insert, remove_endpoint, add_node/remove_node, eviction).  Exits non-zero on a wrong answer;
// the sanitizers abort on a data race or a memory error.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "routercore.h"

using namespace vpsa;

static int fail(const char* what) {
  fprintf(stderr, "routercore selftest FAILED: %s\n", what);
  return 1;
}

int main() {
  // ---- single-threaded semantics
  {
    HashTrie t(4, 0);
    EndpointSet avail;
    avail.add(0);
    avail.add(1);
    t.insert("aaaabbbbcccc", 12, 0);
    t.insert("aaaabbbbdddd", 12, 1);
    EndpointSet sel;
    if (t.longest_prefix_match("aaaabbbbcccczzzz", 16, avail, &sel) != 12) return fail("match length");
    if (sel.ids() != std::vector<uint32_t>{0}) return fail("match set");
    if (t.longest_prefix_match("aaaabbbbxxxx", 12, avail, &sel) != 8) return fail("shared prefix length");
    if (sel.ids().size() != 2) return fail("shared prefix set");
    if (t.longest_prefix_match("zzzz", 4, avail, &sel) != 0 || sel.ids().size() != 2)
      return fail("miss returns the available set");
    t.remove_endpoint(0);
    t.longest_prefix_match("aaaabbbbcccc", 12, avail, &sel);
    if (sel.ids() != std::vector<uint32_t>{1}) return fail("remove_endpoint");
  }
  {
    HashRing r;
    for (int i = 0; i < 8; ++i) r.add_node("http://e" + std::to_string(i));
    std::vector<std::string> before;
    for (int k = 0; k < 2000; ++k) before.push_back(r.get_node("user-" + std::to_string(k)));
    r.remove_node("http://e3");
    int moved = 0;
    for (int k = 0; k < 2000; ++k) {
      std::string now = r.get_node("user-" + std::to_string(k));
      if (now != before[k]) {
        if (before[k] != "http://e3") return fail("a key moved although its node stayed");
        ++moved;
      }
    }
    if (moved == 0 || moved > 600) return fail("remap fraction");
  }
  // ---- concurrent readers and writers
  HashTrie trie(8, 20000);  // small cap: eviction runs under load
  HashRing ring;
  RoundRobin rr;
  for (int i = 0; i < 4; ++i) ring.add_node("n" + std::to_string(i));
  std::vector<std::thread> th;
  std::atomic<long> picks{0};
  for (int w = 0; w < 4; ++w)
    th.emplace_back([&, w] {
      unsigned s = 1234 + w;
      for (int i = 0; i < 4000; ++i) {
        std::string p;
        for (int c = 0; c < 6; ++c) p += "chunk" + std::to_string(rand_r(&s) % 7) + "__";
        trie.insert(p.data(), p.size(), (uint32_t)(rand_r(&s) % 16));
        if (i % 500 == 0) trie.remove_endpoint((uint32_t)(rand_r(&s) % 16));
        if (i % 700 == 0) {
          ring.add_node("dyn" + std::to_string(w));
          ring.remove_node("dyn" + std::to_string(w));
        }
      }
    });
  for (int r = 0; r < 4; ++r)
    th.emplace_back([&, r] {
      unsigned s = 99 + r;
      EndpointSet avail;
      for (uint32_t e = 0; e < 16; ++e) avail.add(e);
      for (int i = 0; i < 6000; ++i) {
        std::string p;
        for (int c = 0; c < 6; ++c) p += "chunk" + std::to_string(rand_r(&s) % 7) + "__";
        EndpointSet sel;
        trie.longest_prefix_match(p.data(), p.size(), avail, &sel);
        if (!sel.any()) abort();
        if (ring.get_node(p).empty()) abort();
        picks += (long)rr.next(5);
      }
    });
  for (auto& t : th) t.join();
  if (trie.node_count() > 40000) return fail("trie eviction did not bound the node count");
  printf("routercore selftest ok (nodes=%zu)\n", trie.node_count());
  return 0;
}
