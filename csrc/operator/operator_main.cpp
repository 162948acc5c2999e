// vpsa-operator: Kubernetes operator for the production-stack CRDs
// (VLLMRuntime, VLLMRouter, CacheServer, LoraAdapter; group production-stack.vllm.ai/v1alpha1).
//
//   vpsa-operator render <VLLMRuntime|VLLMRouter|CacheServer> < cr.json
//       print the desired child objects as a JSON List (pipe into `kubectl apply -f -`)
//   vpsa-operator plan-lora < {"cr": <LoraAdapter>, "pods": [{name, ip, ready, adapters[]}], "deleting": bool}
//       print the load / unload actions the reconciler would take
//   vpsa-operator run --api http://127.0.0.1:8001 [--namespace ns] [--interval 5] [--gpu-resource amd.com/gpu]
//       level-triggered control loop against the API server (reached through `kubectl proxy`
//       or any plain-HTTP endpoint): ensure child objects, update drifted fields, write CR
//       status, load/unload LoRA adapters on engine pods, honour finalizers.
//
// Reference (Go): /root/reference/operator/cmd/main.go:205-259 wires four reconcilers with
// leader election and health probes; the reconcile semantics are cited in reconcile.hpp.
#include <unistd.h>

#include <cstdio>
#include <iostream>
#include <iterator>
#include <set>

#include "../common/http.hpp"
#include "../common/json.hpp"
#include "reconcile.hpp"

using vjson::Value;

static const char* GROUP = "production-stack.vllm.ai";
static const char* VERSION = "v1alpha1";
static const char* FINALIZER = "production-stack.vllm.ai/lora-finalizer";

struct Api {
  std::string base, token;
  int get(const std::string& path, Value* out) const {
    std::string body;
    int st = vhttp::request("GET", base, path, "", &body, "application/json", token);
    if (st == 200 && out) {
      try {
        *out = vjson::parse(body);
      } catch (...) {
        return 0;
      }
    }
    return st;
  }
  int send(const std::string& method, const std::string& path, const Value& obj, const std::string& ctype = "application/json") const {
    std::string body;
    return vhttp::request(method, base, path, vjson::dump(obj), &body, ctype, token);
  }
};

static std::string plural(const std::string& kind) {
  if (kind == "Deployment") return "deployments";
  if (kind == "Service") return "services";
  if (kind == "PersistentVolumeClaim") return "persistentvolumeclaims";
  if (kind == "ServiceAccount") return "serviceaccounts";
  if (kind == "Role") return "roles";
  if (kind == "RoleBinding") return "rolebindings";
  return "";
}

static std::string collection_path(const Value& obj) {
  std::string api = obj.s("apiVersion"), ns = obj.s("metadata.namespace", "default");
  std::string root = api.find('/') == std::string::npos ? "/api/" + api : "/apis/" + api;
  return root + "/namespaces/" + ns + "/" + plural(obj.s("kind"));
}

// Drift check on the fields the reference's needsUpdate compares (replicas, image, args, env,
// resources, ports) - a child object is replaced only when one of them differs.
static bool drifted(const Value& want, const Value& have) {
  if (want.s("kind") != "Deployment") return false;
  if (want.n("spec.replicas") != have.n("spec.replicas")) return true;
  const Value& wc = want.at("spec.template.spec.containers");
  const Value& hc = have.at("spec.template.spec.containers");
  if (!wc.is_arr() || !hc.is_arr() || wc.size() != hc.size()) return true;
  for (size_t i = 0; i < wc.size(); ++i) {
    const Value &w = (*wc.arr)[i], &h = (*hc.arr)[i];
    if (w.s("image") != h.s("image")) return true;
    if (vjson::dump(w.at("args")) != vjson::dump(h.at("args"))) return true;
    if (vjson::dump(w.at("env")) != vjson::dump(h.at("env"))) return true;
    if (vjson::dump(w.at("resources.limits")) != vjson::dump(h.at("resources.limits"))) return true;
    if (w.n("ports") != h.n("ports") && vjson::dump(w.at("ports")) != vjson::dump(h.at("ports"))) return true;
  }
  return false;
}

static std::string ensure(const Api& api, const Value& want) {
  std::string coll = collection_path(want), name = want.s("metadata.name");
  Value have;
  int st = api.get(coll + "/" + name, &have);
  if (st == 404) {
    int c = api.send("POST", coll, want);
    return c >= 200 && c < 300 ? "created" : "create-failed(" + std::to_string(c) + ")";
  }
  if (st != 200) return "get-failed(" + std::to_string(st) + ")";
  if (drifted(want, have)) {
    Value upd = want;
    upd["metadata"].set("resourceVersion", have.s("metadata.resourceVersion"));
    int c = api.send("PUT", coll + "/" + name, upd);
    return c >= 200 && c < 300 ? "updated" : "update-failed(" + std::to_string(c) + ")";
  }
  return "unchanged";
}

static void write_status(const Api& api, const Value& cr, const std::string& plural_name, const Value& status) {
  Value patch = Value::object();
  patch.set("status", status);
  std::string path = std::string("/apis/") + GROUP + "/" + VERSION + "/namespaces/" + cr.s("metadata.namespace", "default") +
                     "/" + plural_name + "/" + cr.s("metadata.name") + "/status";
  api.send("PATCH", path, patch, "application/merge-patch+json");
}

static std::string deployment_phase(const Api& api, const std::string& ns, const std::string& name) {
  Value dep;
  if (api.get("/apis/apps/v1/namespaces/" + ns + "/deployments/" + name, &dep) != 200) return "Unknown";
  double want = dep.n("spec.replicas", 1), ready = dep.n("status.readyReplicas"), upd = dep.n("status.updatedReplicas");
  if (ready >= want && upd >= want) return "Ready";
  if (upd < want || dep.n("status.unavailableReplicas") > 0) return ready > 0 ? "Updating" : "NotReady";
  return "NotReady";
}

static void reconcile_simple(const Api& api, const Value& cr, const std::string& plural_name, const std::vector<Value>& desired) {
  bool ok = true;
  for (auto& obj : desired) {
    std::string r = ensure(api, obj);
    if (r != "unchanged") fprintf(stderr, "[%s/%s] %s %s: %s\n", plural_name.c_str(), cr.s("metadata.name").c_str(),
                                  obj.s("kind").c_str(), obj.s("metadata.name").c_str(), r.c_str());
    if (r.find("failed") != std::string::npos) ok = false;
  }
  Value st = Value::object();
  st.set("status", ok ? deployment_phase(api, cr.s("metadata.namespace", "default"), cr.s("metadata.name")) : "NotReady");
  write_status(api, cr, plural_name, st);
}

// ------------------------------------------------------------------ LoraAdapter
static std::vector<vop::PodState> observe_pods(const Api& api, const std::string& ns, const std::string& base_model, int port) {
  std::vector<vop::PodState> pods;
  Value list;
  if (api.get("/api/v1/namespaces/" + ns + "/pods?labelSelector=model%3D" + base_model, &list) != 200) return pods;
  const Value& items = list.at("items");
  if (!items.is_arr()) return pods;
  for (auto& p : *items.arr) {
    vop::PodState s;
    s.name = p.s("metadata.name");
    s.ip = p.s("status.podIP");
    const Value& cs = p.at("status.containerStatuses");
    s.ready = cs.is_arr() && cs.size() > 0 && p.at("metadata.deletionTimestamp").is_null();
    if (cs.is_arr())
      for (auto& c : *cs.arr) s.ready = s.ready && c.flag("ready");
    if (s.ready && !s.ip.empty()) {
      std::string body;
      if (vhttp::request("GET", "http://" + s.ip + ":" + std::to_string(port), "/v1/models", "", &body) == 200) {
        try {
          Value m = vjson::parse(body);
          const Value& data = m.at("data");
          if (data.is_arr())
            for (auto& e : *data.arr)
              if (!e.at("parent").is_null()) s.adapters.push_back(e.s("id"));
        } catch (...) {
        }
      }
    }
    pods.push_back(s);
  }
  return pods;
}

static void reconcile_lora(const Api& api, const Value& cr) {
  const std::string ns = cr.s("metadata.namespace", "default"), name = cr.s("metadata.name");
  const std::string path = std::string("/apis/") + GROUP + "/" + VERSION + "/namespaces/" + ns + "/loraadapters/" + name;
  bool deleting = !cr.at("metadata.deletionTimestamp").is_null();
  bool has_fin = false;
  const Value& fins = cr.at("metadata.finalizers");
  if (fins.is_arr())
    for (auto& f : *fins.arr) has_fin = has_fin || f.str == FINALIZER;
  if (!deleting && !has_fin) {
    Value patch = Value::object(), m = Value::object(), arr = Value::array();
    if (fins.is_arr())
      for (auto& f : *fins.arr) arr.push(f);
    arr.push(FINALIZER);
    m.set("finalizers", arr);
    patch.set("metadata", m);
    api.send("PATCH", path, patch, "application/merge-patch+json");
  }
  int port = (int)cr.n("spec.vllmPort", 8000);
  // huggingface source: ask the download sidecar (port 30090) of each target pod first
  auto pods = observe_pods(api, ns, cr.s("spec.baseModel"), port);
  auto acts = vop::plan_lora(cr, pods, deleting);
  Value loaded = Value::array();
  std::string phase = "Ready", message;
  for (auto& a : acts) {
    std::string engine = "http://" + a.ip + ":" + std::to_string(port), body;
    std::string bearer = cr.s("spec.vllmApiKey.value");
    if (a.op == "load") {
      std::string lpath = a.path;
      if (cr.s("spec.adapterSource.type") == "huggingface") {
        Value dl = Value::object();
        dl.set("model_id", cr.s("spec.adapterSource.repository"));
        dl.set("local_dir", a.adapter);
        std::string r;
        int st = vhttp::request("POST", "http://" + a.ip + ":30090", "/model/download", vjson::dump(dl), &r);
        if (st == 200) {
          try {
            std::string p = vjson::parse(r).s("path");
            if (!p.empty()) lpath = p;
          } catch (...) {
          }
        } else {
          phase = "Failed";
          message = "adapter download failed on " + a.pod;
          continue;
        }
      }
      Value req = Value::object();
      req.set("lora_name", a.adapter);
      req.set("lora_path", lpath);
      int st = vhttp::request("POST", engine, "/v1/load_lora_adapter", vjson::dump(req), &body, "application/json", bearer);
      bool already = st == 400 && body.find("has already been loaded") != std::string::npos;
      if (st != 200 && !already) {
        phase = "Failed";
        message = "load failed on " + a.pod + " (" + std::to_string(st) + ")";
      }
    } else {
      Value req = Value::object();
      req.set("lora_name", a.adapter);
      vhttp::request("POST", engine, "/v1/unload_lora_adapter", vjson::dump(req), &body, "application/json", bearer);
    }
  }
  if (deleting) {
    if (has_fin) {
      Value patch = Value::object(), m = Value::object(), arr = Value::array();
      for (auto& f : *fins.arr)
        if (f.str != FINALIZER) arr.push(f);
      m.set("finalizers", arr);
      patch.set("metadata", m);
      api.send("PATCH", path, patch, "application/merge-patch+json");
    }
    return;
  }
  auto target = vop::place_adapter(cr, pods);
  for (auto& p : pods)
    if (std::find(target.begin(), target.end(), p.name) != target.end()) {
      Value e = Value::object();
      e.set("name", cr.s("spec.adapterSource.adapterName"));
      e.set("podName", p.name);
      e.set("status", "Loaded");
      loaded.push(e);
    }
  if (target.empty()) {
    phase = "Pending";
    message = "no ready pods with label model=" + cr.s("spec.baseModel");
  }
  Value st = Value::object();
  st.set("phase", phase);
  st.set("message", message);
  st.set("loadedAdapters", loaded);
  write_status(api, cr, "loraadapters", st);
}

static void run_once(const Api& api, const std::string& ns, const std::string& gpu_resource) {
  struct Kind {
    const char* plural;
    const char* kind;
  } kinds[] = {{"vllmruntimes", "VLLMRuntime"}, {"vllmrouters", "VLLMRouter"}, {"cacheservers", "CacheServer"}, {"loraadapters", "LoraAdapter"}};
  for (auto& k : kinds) {
    std::string path = std::string("/apis/") + GROUP + "/" + VERSION + (ns.empty() ? "" : "/namespaces/" + ns) + "/" + k.plural;
    Value list;
    if (api.get(path, &list) != 200) continue;
    const Value& items = list.at("items");
    if (!items.is_arr()) continue;
    for (auto cr : *items.arr) {
      cr.set("kind", k.kind);
      cr.set("apiVersion", std::string(GROUP) + "/" + VERSION);
      try {
        std::string kk = k.kind;
        if (kk == "VLLMRuntime") reconcile_simple(api, cr, k.plural, vop::desired_for_runtime(cr, gpu_resource));
        else if (kk == "VLLMRouter") {
          if (!vop::valid_routing_logic(cr.s("spec.routingLogic", "roundrobin"))) {
            Value st = Value::object();
            st.set("status", "Invalid");
            st.set("message", "unknown routingLogic");
            write_status(api, cr, k.plural, st);
            continue;
          }
          reconcile_simple(api, cr, k.plural, vop::desired_for_router(cr));
        } else if (kk == "CacheServer") reconcile_simple(api, cr, k.plural, vop::desired_for_cacheserver(cr));
        else reconcile_lora(api, cr);
      } catch (const std::exception& e) {
        fprintf(stderr, "reconcile %s/%s failed: %s\n", k.plural, cr.s("metadata.name").c_str(), e.what());
      }
    }
  }
}

static std::string slurp_stdin() { return std::string(std::istreambuf_iterator<char>(std::cin), std::istreambuf_iterator<char>()); }

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: vpsa-operator render <Kind> | plan-lora | run --api URL [--namespace ns] [--interval s]\n");
    return 2;
  }
  std::string mode = argv[1];
  try {
    if (mode == "render" && argc >= 3) {
      std::string kind = argv[2], gpu = argc >= 4 ? argv[3] : "amd.com/gpu";
      Value cr = vjson::parse(slurp_stdin());
      cr.set("kind", kind);
      std::vector<Value> objs = kind == "VLLMRuntime" ? vop::desired_for_runtime(cr, gpu)
                                : kind == "VLLMRouter" ? vop::desired_for_router(cr)
                                : kind == "CacheServer" ? vop::desired_for_cacheserver(cr)
                                                        : std::vector<Value>{};
      Value list = Value::object();
      list.set("apiVersion", "v1");
      list.set("kind", "List");
      Value items = Value::array();
      for (auto& o : objs) items.push(o);
      list.set("items", items);
      printf("%s\n", vjson::dump(list).c_str());
      return 0;
    }
    if (mode == "plan-lora") {
      Value in = vjson::parse(slurp_stdin());
      std::vector<vop::PodState> pods;
      if (in.at("pods").is_arr())
        for (auto& p : *in.at("pods").arr) {
          vop::PodState s;
          s.name = p.s("name");
          s.ip = p.s("ip");
          s.ready = p.flag("ready", true);
          if (p.at("adapters").is_arr())
            for (auto& a : *p.at("adapters").arr) s.adapters.push_back(a.str);
          pods.push_back(s);
        }
      Value out = Value::array();
      for (auto& a : vop::plan_lora(in.at("cr"), pods, in.flag("deleting"))) {
        Value e = Value::object();
        e.set("op", a.op);
        e.set("pod", a.pod);
        e.set("adapter", a.adapter);
        e.set("path", a.path);
        out.push(e);
      }
      printf("%s\n", vjson::dump(out).c_str());
      return 0;
    }
    if (mode == "run") {
      Api api;
      std::string ns, gpu = "amd.com/gpu";
      int interval = 5, health_port = 8081;
      bool once = false;
      for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--api") api.base = next();
        else if (a == "--token") api.token = next();
        else if (a == "--namespace") ns = next();
        else if (a == "--interval") interval = atoi(next().c_str());
        else if (a == "--gpu-resource") gpu = next();
        else if (a == "--health-port") health_port = atoi(next().c_str());
        else if (a == "--once") once = true;
      }
      if (api.base.empty()) {
        fprintf(stderr, "--api is required (e.g. http://127.0.0.1:8001 from `kubectl proxy`)\n");
        return 2;
      }
      vhttp::Server health;
      if (!once && health_port > 0)
        health.start("0.0.0.0", health_port, [](const vhttp::Request& r) {
          vhttp::Response resp;
          resp.body = (r.path == "/healthz" || r.path == "/readyz") ? "ok" : "not found";
          resp.status = (r.path == "/healthz" || r.path == "/readyz") ? 200 : 404;
          resp.content_type = "text/plain";
          return resp;
        });
      do {
        run_once(api, ns, gpu);
        if (!once) sleep((unsigned)interval);
      } while (!once);
      return 0;
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "vpsa-operator: %s\n", e.what());
    return 1;
  }
  fprintf(stderr, "unknown mode %s\n", mode.c_str());
  return 2;
}
