// vpsa-picker: standalone native endpoint picker for Kubernetes Gateway API deployments.
//
// Native counterpart of the reference's Go pickers that are patched into the
// gateway-api-inference-extension endpoint picker
// (/root/reference/src/gateway_inference_extension/roundrobin_picker.go:32-58,
//  prefix_aware_picker.go:25-213, kv_aware_picker.go:28-133, scheduler.patch:13-77):
//   roundrobin  candidates sorted by name, atomic counter modulo n
//   prefix      128-unit xxh64 chunk trie, longest match, random tie-break, insert after pick
//   kvaware     POST <controller>/lookup {model, prompt} -> {instance_id, tokens};
//               GET <controller>/query?ip=<pod address> maps pods to instances; the match is
//               honoured when tokens >= max(whitespace_fields(prompt) - threshold, 0), else
//               round robin
//   session     consistent-hash ring on the session key
// It shares the routing core (routercore.cpp) with the Python router, so both data planes
// make identical decisions for identical inputs.
//
//   POST /pick {"model","prompt","session","pods":[{"name","address"},...]}
//        -> {"pod","address","picker","matched"}
//   GET  /health
#include <algorithm>
#include <cstdio>
#include <mutex>
#include <random>
#include <sstream>

#include "../common/http.hpp"
#include "../common/json.hpp"
#include "routercore.h"

using vjson::Value;

struct Pod {
  std::string name, address;
};

struct Picker {
  std::string kind = "roundrobin";
  std::string controller;
  long threshold = 2000;
  vpsa::RoundRobin rr;
  vpsa::HashTrie trie{128, 4000000};
  vpsa::HashRing ring{160};
  std::mutex mu;
  std::map<std::string, uint32_t> ids;       // pod name -> trie endpoint id
  std::vector<std::string> names;            // id -> pod name
  std::map<std::string, std::string> inst;   // pod address -> controller instance id
  std::mt19937_64 rng{std::random_device{}()};

  uint32_t id_of(const std::string& name) {
    auto it = ids.find(name);
    if (it != ids.end()) return it->second;
    uint32_t i = (uint32_t)names.size();
    ids[name] = i;
    names.push_back(name);
    return i;
  }

  const Pod& round_robin(std::vector<Pod>& pods) {
    std::sort(pods.begin(), pods.end(), [](const Pod& a, const Pod& b) { return a.name < b.name; });
    return pods[rr.next(pods.size())];
  }

  static long fields(const std::string& s) {
    long n = 0;
    bool in = false;
    for (unsigned char c : s) {
      bool sp = isspace(c);
      if (!sp && !in) ++n;
      in = !sp;
    }
    return n;
  }

  Value pick(const Value& req) {
    std::vector<Pod> pods;
    const Value& pl = req.at("pods");
    if (pl.is_arr())
      for (auto& p : *pl.arr) pods.push_back({p.s("name"), p.s("address")});
    Value out = Value::object();
    if (pods.empty()) {
      out.set("error", "no candidate pods");
      return out;
    }
    const std::string prompt = req.s("prompt");
    const Pod* chosen = nullptr;
    long matched = 0;
    std::string used = kind;
    if (kind == "prefix") {
      std::lock_guard<std::mutex> lk(mu);
      vpsa::EndpointSet avail, sel;
      for (auto& p : pods) avail.add(id_of(p.name));
      matched = (long)trie.longest_prefix_match(prompt.data(), prompt.size(), avail, &sel);
      auto cand = sel.ids();
      if (cand.empty()) cand = avail.ids();
      uint32_t pickid = cand[rng() % cand.size()];
      for (auto& p : pods)
        if (ids[p.name] == pickid) chosen = &p;
      trie.insert(prompt.data(), prompt.size(), pickid);
    } else if (kind == "session" && !req.s("session").empty()) {
      std::lock_guard<std::mutex> lk(mu);
      std::vector<std::string> want;
      for (auto& p : pods) want.push_back(p.name);
      for (auto& n : ring.nodes())
        if (std::find(want.begin(), want.end(), n) == want.end()) ring.remove_node(n);
      for (auto& n : want) ring.add_node(n);
      std::string node = ring.get_node(req.s("session"));
      for (auto& p : pods)
        if (p.name == node) chosen = &p;
    } else if (kind == "kvaware" && !controller.empty()) {
      Value q = Value::object();
      q.set("model", req.s("model"));
      q.set("prompt", prompt);
      std::string resp;
      int st = vhttp::request("POST", controller, "/lookup", vjson::dump(q), &resp);
      if (st == 200) {
        try {
          Value r = vjson::parse(resp);
          std::string target = r.s("instance_id");
          matched = (long)r.i("tokens", 0, 0);
          long need = std::max(fields(prompt) - threshold, 0L);
          if (!target.empty() && matched >= need) {
            for (auto& p : pods) {
              std::string iid;
              {
                std::lock_guard<std::mutex> lk(mu);
                auto it = inst.find(p.address);
                if (it != inst.end()) iid = it->second;
              }
              if (iid.empty()) {
                std::string qr;
                if (vhttp::request("GET", controller, "/query?ip=" + p.address, "", &qr) == 200) {
                  try {
                    iid = vjson::parse(qr).s("instance_id");
                  } catch (...) {
                  }
                  if (!iid.empty()) {
                    std::lock_guard<std::mutex> lk(mu);
                    inst[p.address] = iid;
                  }
                }
              }
              if (iid == target) {
                chosen = &p;
                break;
              }
            }
          }
        } catch (...) {
        }
      }
      if (!chosen) used = "roundrobin(fallback)";
    }
    if (!chosen) chosen = &round_robin(pods);
    out.set("pod", chosen->name);
    out.set("address", chosen->address);
    out.set("picker", used);
    out.set("matched", (int64_t)matched);
    return out;
  }
};

int main(int argc, char** argv) {
  Picker pk;
  int port = 9002;
  std::string host = "0.0.0.0";
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--picker") pk.kind = next();
    else if (a == "--port") port = atoi(next().c_str());
    else if (a == "--host") host = next();
    else if (a == "--controller") pk.controller = next();
    else if (a == "--threshold") pk.threshold = atol(next().c_str());
    else if (a == "--help") {
      printf("vpsa-picker --picker roundrobin|prefix|kvaware|session [--port N] [--controller URL] [--threshold N]\n");
      return 0;
    }
  }
  vhttp::Server srv;
  int bound = srv.start(host, port, [&](const vhttp::Request& r) {
    vhttp::Response resp;
    if (r.path == "/health") {
      resp.body = "{\"status\":\"ok\"}";
    } else if (r.path == "/pick" && r.method == "POST") {
      try {
        Value out = pk.pick(vjson::parse(r.body));
        if (out.find("error")) resp.status = 400;
        resp.body = vjson::dump(out);
      } catch (const std::exception& e) {
        resp.status = 400;
        Value err = Value::object();
        err.set("error", e.what());
        resp.body = vjson::dump(err);
      }
    } else {
      resp.status = 404;
      resp.body = "{\"error\":\"not found\"}";
    }
    return resp;
  });
  if (!bound) {
    fprintf(stderr, "vpsa-picker: cannot listen on %s:%d\n", host.c_str(), port);
    return 1;
  }
  printf("vpsa-picker (%s) listening on %s:%d\n", pk.kind.c_str(), host.c_str(), bound);
  fflush(stdout);
  srv.wait();
  return 0;
}
