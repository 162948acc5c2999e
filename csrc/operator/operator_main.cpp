// vpsa-operator: Kubernetes operator for the production-stack CRDs
// (VLLMRuntime, VLLMRouter, CacheServer, LoraAdapter; group production-stack.vllm.ai/v1alpha1).
//
//   vpsa-operator render <VLLMRuntime|VLLMRouter|CacheServer> < cr.json
//       print the desired child objects as a JSON List (pipe into `kubectl apply -f -`)
//   vpsa-operator plan-lora < {"cr": <LoraAdapter>, "pods": [{name, ip, ready, adapters[]}], "deleting": bool}
//       print the load / unload actions the reconciler would take
//   vpsa-operator run [--api URL] [--namespace ns] [--interval 30] [--gpu-resource amd.com/gpu]
//                     [--leader-elect --leader-elect-namespace ns --leader-id id] [--health-port 8081]
//                     [--ca-file f] [--token-file f] [--insecure-skip-tls-verify]
//                     [--metrics-bind-address :8443] [--metrics-secure[=false]]
//                     [--metrics-cert-path dir [--metrics-cert-name tls.crt] [--metrics-cert-key tls.key]]
//       control loop against the API server: ensure child objects, update drifted fields, write
//       CR status, load/unload LoRA adapters on engine pods, honour finalizers.
//       Without --api, inside a pod: https://$KUBERNETES_SERVICE_HOST:$KUBERNETES_SERVICE_PORT,
//       verified against the service account's ca.crt and authenticated with its token (the
//       token file is read again every minute: bound tokens are rotated by the kubelet).  With
//       --api http://... any plain endpoint (a `kubectl proxy`, a test server) still works.
//       --metrics-bind-address serves /metrics on a port of its own: over TLS unless
//       --metrics-secure=false (the certificate from --metrics-cert-path, picked up again when the
//       files change; a self-signed one made at start otherwise), and, when secure, only for
//       callers whose bearer token the API server authenticates (TokenReview) and authorizes for
//       `get /metrics` (SubjectAccessReview).
//       Event-driven: one watch stream per custom resource kind plus one on the engine pods
//       (?watch=true&resourceVersion=...) wakes the reconciler as soon as something changes; the
//       interval is only the resync period.  With --leader-elect only the holder of the Lease
//       `4549d26f.vllm.ai` (coordination.k8s.io/v1) reconciles; the others stand by and take
//       over when the lease is not renewed.  The health port serves /healthz, /readyz and
//       Prometheus text on /metrics.
//
// Reference (Go): /root/reference/operator/cmd/main.go:107-259 (manager with metrics server,
// leader election id :187, health and ready probes :252-259, four reconcilers :205-233) and the
// pod-ready watch predicate of /root/reference/operator/internal/controller/loraadapter_controller.go:234-275;
// the reconcile semantics are cited in reconcile.hpp.
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <cstdio>
#include <ctime>
#include <fstream>
#include <iostream>
#include <iterator>
#include <mutex>
#include <set>
#include <thread>

#include "../common/http.hpp"
#include "../common/json.hpp"
#include "reconcile.hpp"

using vjson::Value;

static const char* GROUP = "production-stack.vllm.ai";
static const char* VERSION = "v1alpha1";
static const char* FINALIZER = "production-stack.vllm.ai/lora-finalizer";
static const char* LEASE_NAME = "4549d26f.vllm.ai";  // the reference manager's election id
static int g_sidecar_port = 30090;

// ------------------------------------------------------------------ metrics (Prometheus text)
struct Metrics {
  std::mutex mu;
  std::map<std::string, double> values;  // full series name incl. labels -> value
  void add(const std::string& series, double d = 1) {
    std::lock_guard<std::mutex> lk(mu);
    values[series] += d;
  }
  void set(const std::string& series, double v) {
    std::lock_guard<std::mutex> lk(mu);
    values[series] = v;
  }
  std::string render() {
    std::lock_guard<std::mutex> lk(mu);
    std::string out, last;
    for (auto& kv : values) {
      std::string base = kv.first.substr(0, kv.first.find('{'));
      if (base != last) {
        bool counter = base.size() > 6 && base.compare(base.size() - 6, 6, "_total") == 0;
        out += "# TYPE " + base + (counter ? " counter\n" : " gauge\n");
        last = base;
      }
      char num[64];
      snprintf(num, sizeof num, "%.17g", kv.second);
      out += kv.first + " " + num + "\n";
    }
    return out;
  }
};
static Metrics g_metrics;

static std::string read_file_trimmed(const std::string& path) {
  std::ifstream f(path);
  std::string s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
  return s;
}

// The bearer token of the API client: fixed (--token) or a file that is read again once a minute
// (a projected service-account token is replaced by the kubelet well before it expires).
struct TokenSource {
  std::mutex mu;
  std::string fixed, file, cached;
  time_t read_at = 0;
  std::string get() {
    std::lock_guard<std::mutex> lk(mu);
    if (file.empty()) return fixed;
    time_t now = time(nullptr);
    if (cached.empty() || now - read_at >= 60) {
      std::string t = read_file_trimmed(file);
      if (!t.empty()) cached = t;  // an unreadable file keeps the last good token
      read_at = now;
    }
    return cached;
  }
};
static TokenSource g_token;

struct Api {
  std::string base;
  std::string bearer() const { return g_token.get(); }
  int get(const std::string& path, Value* out) const {
    std::string body;
    int st = vhttp::request("GET", base, path, "", &body, "application/json", bearer());
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
    return vhttp::request(method, base, path, vjson::dump(obj), &body, ctype, bearer());
  }
  // POST that hands back the reply (review objects: the verdict is in the created object's status)
  int post(const std::string& path, const Value& obj, Value* out) const {
    std::string body;
    int st = vhttp::request("POST", base, path, vjson::dump(obj), &body, "application/json", bearer());
    if (st >= 200 && st < 300 && out) {
      try {
        *out = vjson::parse(body);
      } catch (...) {
        return 0;
      }
    }
    return st;
  }
};

// ------------------------------------------------------------------ who may read /metrics
// The caller's bearer token is authenticated by the API server (TokenReview) and the user it
// names must be allowed `get` on the non-resource URL /metrics (SubjectAccessReview): the same
// two questions the reference's metrics filter asks
// (/root/reference/operator/cmd/main.go:146-152).  Verdicts are kept for a short while so a
// scrape every few seconds is not two API calls each time.
struct MetricsAuth {
  struct Verdict {
    int status;
    time_t at;
  };
  std::mutex mu;
  std::map<std::string, Verdict> cache;
  static constexpr int kAllowTtl = 60, kDenyTtl = 10;

  // 200 allowed, 401 not authenticated, 403 authenticated but not authorized, 500 review failed
  int check(const Api& api, const std::string& authorization) {
    if (authorization.compare(0, 7, "Bearer ") != 0 || authorization.size() <= 7) return 401;
    std::string token = authorization.substr(7);
    time_t now = time(nullptr);
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = cache.find(token);
      if (it != cache.end() && now - it->second.at < (it->second.status == 200 ? kAllowTtl : kDenyTtl)) return it->second.status;
      if (cache.size() > 1024) cache.clear();
    }
    int status = review(api, token);
    if (status != 500) {
      std::lock_guard<std::mutex> lk(mu);
      cache[token] = Verdict{status, now};
    }
    return status;
  }

  static int review(const Api& api, const std::string& token) {
    Value tr = Value::object();
    tr.set("apiVersion", "authentication.k8s.io/v1");
    tr.set("kind", "TokenReview");
    tr["spec"].set("token", token);
    Value tr_out;
    int st = api.post("/apis/authentication.k8s.io/v1/tokenreviews", tr, &tr_out);
    if (st < 200 || st >= 300) return 500;
    if (!tr_out.flag("status.authenticated")) return 401;
    Value sar = Value::object();
    sar.set("apiVersion", "authorization.k8s.io/v1");
    sar.set("kind", "SubjectAccessReview");
    Value& spec = sar["spec"];
    spec.set("user", tr_out.s("status.user.username"));
    const Value& groups = tr_out.at("status.user.groups");
    if (groups.is_arr()) spec.set("groups", groups);
    const Value& uid = tr_out.at("status.user.uid");
    if (uid.is_str()) spec.set("uid", uid);
    spec["nonResourceAttributes"].set("path", "/metrics");
    spec["nonResourceAttributes"].set("verb", "get");
    Value sar_out;
    st = api.post("/apis/authorization.k8s.io/v1/subjectaccessreviews", sar, &sar_out);
    if (st < 200 || st >= 300) return 500;
    return sar_out.flag("status.allowed") ? 200 : 403;
  }
};

#ifdef VHTTP_TLS
// A throw-away certificate for the metrics port when none is mounted (what controller-runtime
// does for the reference, /root/reference/operator/cmd/main.go:154-156): P-256 key, one year,
// written 0600 into a private temp directory.
static bool make_self_signed(std::string* cert_file, std::string* key_file) {
  char dir[] = "/tmp/vpsa-operator-tls-XXXXXX";
  if (!mkdtemp(dir)) return false;
  EVP_PKEY* key = EVP_EC_gen("P-256");
  X509* crt = X509_new();
  bool ok = key && crt;
  if (ok) {
    X509_set_version(crt, 2);
    ASN1_INTEGER_set(X509_get_serialNumber(crt), (long)time(nullptr));
    X509_gmtime_adj(X509_getm_notBefore(crt), -300);
    X509_gmtime_adj(X509_getm_notAfter(crt), 365L * 24 * 3600);
    X509_set_pubkey(crt, key);
    X509_NAME* name = X509_get_subject_name(crt);
    X509_NAME_add_entry_by_txt(name, "CN", MBSTRING_ASC, (const unsigned char*)"vpsa-operator-metrics", -1, -1, 0);
    X509_set_issuer_name(crt, name);
    ok = X509_sign(crt, key, EVP_sha256()) > 0;
  }
  *cert_file = std::string(dir) + "/tls.crt";
  *key_file = std::string(dir) + "/tls.key";
  if (ok) {
    FILE* f = fopen(cert_file->c_str(), "w");
    ok = f && PEM_write_X509(f, crt) == 1;
    if (f) fclose(f);
  }
  if (ok) {
    mode_t old = umask(0077);
    FILE* f = fopen(key_file->c_str(), "w");
    umask(old);
    ok = f && PEM_write_PrivateKey(f, key, nullptr, nullptr, 0, nullptr, nullptr) == 1;
    if (f) fclose(f);
  }
  if (crt) X509_free(crt);
  if (key) EVP_PKEY_free(key);
  return ok;
}
#endif

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
    if (r != "unchanged") g_metrics.add("vpsa_operator_child_objects_total{kind=\"" + obj.s("kind") + "\",result=\"" + r.substr(0, r.find('(')) + "\"}");
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
  int port = (int)cr.i("spec.vllmPort", 8000, 1, 65535);
  // huggingface source: ask the download sidecar (port 30090) of each target pod first
  auto pods = observe_pods(api, ns, cr.s("spec.baseModel"), port);
  auto acts = vop::plan_lora(cr, pods, deleting);
  if (deleting) {
    // unload everywhere the adapter is known to be: what the pods list AND what this CR's status
    // recorded (an engine's model listing can lag behind its loaded adapters)
    const Value& rec = cr.at("status.loadedAdapters");
    if (rec.is_arr())
      for (auto& e : *rec.arr) {
        const std::string pod_name = e.s("podName");
        bool planned = false;
        for (auto& a : acts) planned = planned || a.pod == pod_name;
        if (planned) continue;
        for (auto& p : pods)
          if (p.name == pod_name && !p.ip.empty())
            acts.push_back({"unload", p.name, p.ip, cr.s("spec.adapterSource.adapterName"), ""});
      }
  }
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
        int st = vhttp::request("POST", "http://" + a.ip + ":" + std::to_string(g_sidecar_port), "/model/download", vjson::dump(dl), &r);
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
      g_metrics.add(std::string("vpsa_operator_lora_actions_total{op=\"load\",result=\"") + (st == 200 ? "ok" : already ? "already_loaded" : "failed") + "\"}");
    } else {
      Value req = Value::object();
      req.set("lora_name", a.adapter);
      int st = vhttp::request("POST", engine, "/v1/unload_lora_adapter", vjson::dump(req), &body, "application/json", bearer);
      g_metrics.add(std::string("vpsa_operator_lora_actions_total{op=\"unload\",result=\"") + (st == 200 ? "ok" : "failed") + "\"}");
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
        g_metrics.add(std::string("vpsa_operator_reconcile_total{kind=\"") + k.kind + "\",result=\"ok\"}");
      } catch (const std::exception& e) {
        fprintf(stderr, "reconcile %s/%s failed: %s\n", k.plural, cr.s("metadata.name").c_str(), e.what());
        g_metrics.add(std::string("vpsa_operator_reconcile_total{kind=\"") + k.kind + "\",result=\"error\"}");
      }
    }
  }
}

// ------------------------------------------------------------------ event-driven wake-up
// Reconciliation stays level-triggered (every pass looks at the whole state); the watch streams
// only decide WHEN a pass runs: at once after any change to a custom resource or to an engine
// pod (a pod turning ready is what lets a pending LoRA adapter load: the reference's watch
// predicate), otherwise at the resync interval.
struct Wakeup {
  std::mutex mu;
  std::condition_variable cv;
  bool dirty = true;  // first pass runs immediately
  void notify() {
    {
      std::lock_guard<std::mutex> lk(mu);
      dirty = true;
    }
    cv.notify_all();
  }
  // true: woken by an event, false: the interval passed
  bool wait(int seconds) {
    std::unique_lock<std::mutex> lk(mu);
    bool ev = cv.wait_for(lk, std::chrono::seconds(seconds), [&] { return dirty; });
    dirty = false;
    return ev;
  }
};
static Wakeup g_wake;
static std::atomic<bool> g_stop{false};

static void watch_loop(Api api, std::string list_path, std::string label) {
  const std::string sep = list_path.find('?') == std::string::npos ? "?" : "&";
  while (!g_stop) {
    Value list;
    std::string rv;
    if (api.get(list_path, &list) == 200) rv = list.s("metadata.resourceVersion");
    std::string path = list_path + sep + "watch=true&allowWatchBookmarks=true&timeoutSeconds=300";
    if (!rv.empty()) path += "&resourceVersion=" + rv;
    int st = vhttp::stream_lines(api.base, path, api.bearer(), [&](const std::string& line) {
      if (g_stop) return false;
      try {
        Value ev = vjson::parse(line);
        std::string type = ev.s("type");
        if (type == "ERROR") return false;  // e.g. 410 Gone: list again for a fresh resourceVersion
        if (type != "BOOKMARK") {
          g_metrics.add("vpsa_operator_watch_events_total{resource=\"" + label + "\",type=\"" + type + "\"}");
          g_wake.notify();
        }
      } catch (...) {
        return false;  // not an event line: start over rather than guess
      }
      return true;
    }, 330);
    if (st != 200) g_metrics.add("vpsa_operator_watch_restarts_total{resource=\"" + label + "\"}");
    // the stream ended (server timeout, error, 410): re-list and watch again, after a breath
    for (int i = 0; i < 5 && !g_stop; ++i) usleep(100 * 1000);
  }
}

// ------------------------------------------------------------------ leader election (Lease)
static std::string rfc3339_now() {
  timespec ts{};
  clock_gettime(CLOCK_REALTIME, &ts);
  tm t{};
  gmtime_r(&ts.tv_sec, &t);
  char buf[64];
  snprintf(buf, sizeof buf, "%04d-%02d-%02dT%02d:%02d:%02d.%06ldZ", t.tm_year + 1900, t.tm_mon + 1, t.tm_mday,
           t.tm_hour, t.tm_min, t.tm_sec, ts.tv_nsec / 1000);
  return buf;
}

static double parse_rfc3339(const std::string& s) {
  tm t{};
  if (s.size() < 19 || !strptime(s.substr(0, 19).c_str(), "%Y-%m-%dT%H:%M:%S", &t)) return 0;
  double frac = 0;
  if (s.size() > 20 && s[19] == '.') frac = atof(("0" + s.substr(19, s.find_first_not_of("0123456789", 20) - 19)).c_str());
  return (double)timegm(&t) + frac;
}

// One round of the election: create the Lease, renew our own, or take over an expired one.
// Returns true while this process holds it.  Optimistic concurrency: every write carries the
// resourceVersion it read, a 409 means somebody else wrote first and we are not the leader.
static bool elect_once(const Api& api, const std::string& ns, const std::string& id, int duration_s) {
  const std::string coll = "/apis/coordination.k8s.io/v1/namespaces/" + ns + "/leases";
  Value lease;
  int st = api.get(coll + "/" + LEASE_NAME, &lease);
  const std::string now = rfc3339_now();
  auto spec = [&](double transitions, const std::string& acquired) {
    Value sp = Value::object();
    sp.set("holderIdentity", id);
    sp.set("leaseDurationSeconds", (double)duration_s);
    sp.set("acquireTime", acquired);
    sp.set("renewTime", now);
    sp.set("leaseTransitions", transitions);
    return sp;
  };
  if (st == 404) {
    Value obj = Value::object(), meta = Value::object();
    obj.set("apiVersion", "coordination.k8s.io/v1");
    obj.set("kind", "Lease");
    meta.set("name", LEASE_NAME);
    meta.set("namespace", ns);
    obj.set("metadata", meta);
    obj.set("spec", spec(0, now));
    int c = api.send("POST", coll, obj);
    return c >= 200 && c < 300;
  }
  if (st != 200) return false;
  const std::string holder = lease.s("spec.holderIdentity");
  const double renewed = parse_rfc3339(lease.s("spec.renewTime"));
  const double dur = lease.n("spec.leaseDurationSeconds", duration_s);
  const bool mine = holder == id;
  const bool expired = holder.empty() || parse_rfc3339(now) > renewed + dur;
  if (!mine && !expired) return false;
  Value upd = lease;
  upd.set("spec", spec(lease.n("spec.leaseTransitions") + (mine ? 0 : 1), mine ? lease.s("spec.acquireTime", now) : now));
  int c = api.send("PUT", coll + "/" + LEASE_NAME, upd);
  return c >= 200 && c < 300;
}

static std::string slurp_stdin() { return std::string(std::istreambuf_iterator<char>(std::cin), std::istreambuf_iterator<char>()); }

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: vpsa-operator render <Kind> | plan-lora | run [--api URL] [--namespace ns] [--interval s] [--leader-elect] [--health-port p] [--metrics-bind-address :8443 [--metrics-secure=false] [--metrics-cert-path dir]] [--ca-file f] [--token-file f]\n");
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
      std::string ns, gpu = "amd.com/gpu", lease_ns, leader_id;
      std::string ca_file, token_file, metrics_addr, metrics_cert_path, metrics_cert_name = "tls.crt", metrics_cert_key = "tls.key";
      int interval = 30, health_port = 8081, lease_s = 15;
      bool once = false, leader_elect = false, watch = true, metrics_secure = true, skip_verify = false;
      for (int i = 2; i < argc; ++i) {
        std::string a = argv[i], inline_value;
        bool has_inline = false;  // --flag=value is the same as --flag value
        size_t eq = a.find('=');
        if (a.compare(0, 2, "--") == 0 && eq != std::string::npos && a.compare(0, 16, "--metrics-secure") != 0) {
          inline_value = a.substr(eq + 1);
          a = a.substr(0, eq);
          has_inline = true;
        }
        auto next = [&]() -> std::string { return has_inline ? inline_value : i + 1 < argc ? argv[++i] : ""; };
        if (a == "--api") api.base = next();
        else if (a == "--token") g_token.fixed = next();
        else if (a == "--token-file") token_file = next();
        else if (a == "--ca-file") ca_file = next();
        else if (a == "--insecure-skip-tls-verify") skip_verify = true;
        else if (a == "--metrics-bind-address") metrics_addr = next();
        else if (a == "--metrics-secure" || a == "--metrics-secure=true") metrics_secure = true;
        else if (a == "--metrics-secure=false") metrics_secure = false;
        else if (a == "--metrics-cert-path") metrics_cert_path = next();
        else if (a == "--metrics-cert-name") metrics_cert_name = next();
        else if (a == "--metrics-cert-key") metrics_cert_key = next();
        else if (a == "--namespace") ns = next();
        else if (a == "--interval") interval = atoi(next().c_str());
        else if (a == "--gpu-resource") gpu = next();
        else if (a == "--health-port") health_port = atoi(next().c_str());
        else if (a == "--sidecar-port") g_sidecar_port = atoi(next().c_str());
        else if (a == "--leader-elect") leader_elect = true;
        else if (a == "--leader-elect-namespace") lease_ns = next();
        else if (a == "--leader-id") leader_id = next();
        else if (a == "--lease-duration") lease_s = atoi(next().c_str());
        else if (a == "--no-watch") watch = false;
        else if (a == "--once") once = true;
      }
      // inside a pod: the API server's address comes from the environment, its CA bundle and
      // this pod's token from the service-account volume (VPSA_SA_DIR relocates it for tests)
      const char* sa_env = getenv("VPSA_SA_DIR");
      const std::string sa_dir = sa_env ? sa_env : "/var/run/secrets/kubernetes.io/serviceaccount";
      const bool have_sa = access((sa_dir + "/token").c_str(), R_OK) == 0;
      if (api.base.empty()) {
        const char* h = getenv("KUBERNETES_SERVICE_HOST");
        const char* p = getenv("KUBERNETES_SERVICE_PORT");
        if (h && *h) {
          std::string host = h;
          if (host.find(':') != std::string::npos) host = "[" + host + "]";  // IPv6 cluster
          api.base = "https://" + host + ":" + (p && *p ? p : "443");
        }
      }
      if (api.base.empty()) {
        fprintf(stderr, "no API server: pass --api URL, or run in a pod (KUBERNETES_SERVICE_HOST)\n");
        return 2;
      }
      const bool https = api.base.compare(0, 8, "https://") == 0;
      if (https) {
#ifndef VHTTP_TLS
        fprintf(stderr, "this build has no TLS: %s cannot be reached\n", api.base.c_str());
        return 2;
#endif
        if (ca_file.empty() && access((sa_dir + "/ca.crt").c_str(), R_OK) == 0) ca_file = sa_dir + "/ca.crt";
        vhttp::tls_client_config().ca_file = ca_file;
        vhttp::tls_client_config().insecure_skip_verify = skip_verify;
      }
      if (token_file.empty() && g_token.fixed.empty() && https && have_sa) token_file = sa_dir + "/token";
      g_token.file = token_file;
      if (ns.empty() && getenv("VPSA_WATCH_OWN_NAMESPACE")) ns = read_file_trimmed(sa_dir + "/namespace");
      fprintf(stderr, "[operator] API server %s (%s, %s)\n", api.base.c_str(),
              https ? (skip_verify ? "TLS, certificate NOT verified" : ca_file.empty() ? "TLS, system trust store" : "TLS, CA bundle") : "plain HTTP",
              !token_file.empty() ? "service-account token" : g_token.fixed.empty() ? "no token" : "fixed token");
      if (once) {
        run_once(api, ns, gpu);
        return 0;
      }
      if (interval < 1) interval = 1;
      if (lease_s < 2) lease_s = 2;
      if (lease_ns.empty()) lease_ns = ns.empty() ? "default" : ns;
      if (leader_id.empty()) {
        char host[256] = "operator";
        gethostname(host, sizeof host - 1);
        leader_id = std::string(host) + "_" + std::to_string((long)getpid());
      }
      signal(SIGTERM, [](int) { g_stop = true; g_wake.notify(); });
      signal(SIGINT, [](int) { g_stop = true; g_wake.notify(); });
      std::atomic<bool> leader{!leader_elect};
      std::atomic<bool> election_alive{!leader_elect};
      g_metrics.set("vpsa_operator_leader", leader ? 1 : 0);
      vhttp::Server health;
      if (health_port > 0) {
        int bound = health.start("0.0.0.0", health_port, [&](const vhttp::Request& r) {
          vhttp::Response resp;
          resp.content_type = "text/plain";
          if (r.path == "/healthz") resp.body = "ok";
          else if (r.path == "/readyz") {
            // ready = taking part in the election (leader or standby); both may be sent traffic-free probes
            resp.status = election_alive ? 200 : 503;
            resp.body = election_alive ? "ok" : "election loop not running";
          } else if (r.path == "/metrics" && metrics_addr.empty()) {
            // no metrics port of its own: the probe port carries the series (plain, unprotected)
            resp.content_type = "text/plain; version=0.0.4";
            resp.body = g_metrics.render();
          } else {
            resp.status = 404;
            resp.body = "not found";
          }
          return resp;
        });
        if (!bound) fprintf(stderr, "could not bind health port %d\n", health_port);
      }
      vhttp::Server metrics_srv;
      MetricsAuth metrics_auth;
      if (!metrics_addr.empty() && metrics_addr != "0") {
        std::string mhost = "0.0.0.0";
        size_t colon = metrics_addr.rfind(':');
        int mport = atoi(metrics_addr.substr(colon == std::string::npos ? 0 : colon + 1).c_str());
        if (colon != std::string::npos && colon > 0) mhost = metrics_addr.substr(0, colon);
        if (metrics_secure) {
#ifdef VHTTP_TLS
          std::string crt, key;
          if (!metrics_cert_path.empty()) {
            crt = metrics_cert_path + "/" + metrics_cert_name;
            key = metrics_cert_path + "/" + metrics_cert_key;
          } else if (!make_self_signed(&crt, &key)) {
            fprintf(stderr, "could not make a self-signed certificate for the metrics port\n");
            return 1;
          }
          if (!metrics_srv.use_tls(crt, key)) {
            fprintf(stderr, "could not load the metrics certificate %s / %s\n", crt.c_str(), key.c_str());
            return 1;
          }
#else
          fprintf(stderr, "this build has no TLS: use --metrics-secure=false\n");
          return 2;
#endif
        }
        int bound = metrics_srv.start(mhost, mport, [&, metrics_secure](const vhttp::Request& r) {
          vhttp::Response resp;
          resp.content_type = "text/plain";
          if (r.path != "/metrics") {
            resp.status = 404;
            resp.body = "not found";
            return resp;
          }
          if (metrics_secure) {
            auto it = r.headers.find("authorization");
            int verdict = metrics_auth.check(api, it == r.headers.end() ? "" : it->second);
            g_metrics.add("vpsa_operator_metrics_scrapes_total{code=\"" + std::to_string(verdict) + "\"}");
            if (verdict != 200) {
              resp.status = verdict;
              resp.body = verdict == 401 ? "Unauthorized" : verdict == 403 ? "Forbidden" : "could not review the token";
              return resp;
            }
          }
          resp.content_type = "text/plain; version=0.0.4";
          resp.body = g_metrics.render();
          return resp;
        });
        if (!bound) fprintf(stderr, "could not bind the metrics address %s\n", metrics_addr.c_str());
        else fprintf(stderr, "[operator] metrics on %s:%d (%s)\n", mhost.c_str(), bound, metrics_secure ? "TLS, token review" : "plain HTTP");
      }
      std::vector<std::thread> threads;
      if (leader_elect)
        threads.emplace_back([&] {
          // renew at a third of the lease duration: two missed rounds still leave a margin
          while (!g_stop) {
            bool now_leader = false;
            try {
              now_leader = elect_once(api, lease_ns, leader_id, lease_s);
            } catch (...) {
            }
            election_alive = true;
            if (now_leader != leader) {
              fprintf(stderr, "[leader-election] %s %s\n", leader_id.c_str(), now_leader ? "became leader" : "lost the lease");
              g_metrics.add("vpsa_operator_leader_transitions_total");
              leader = now_leader;
              g_metrics.set("vpsa_operator_leader", now_leader ? 1 : 0);
              if (now_leader) g_wake.notify();
            }
            for (int i = 0; i < lease_s * 10 / 3 && !g_stop; ++i) usleep(100 * 1000);
          }
        });
      if (watch) {
        const std::string crd_root = std::string("/apis/") + GROUP + "/" + VERSION + (ns.empty() ? "" : "/namespaces/" + ns) + "/";
        for (const char* plural_name : {"vllmruntimes", "vllmrouters", "cacheservers", "loraadapters"})
          threads.emplace_back(watch_loop, api, crd_root + plural_name, std::string(plural_name));
        // engine pods carry the label `model` (the chart and the runtime controller stamp it)
        threads.emplace_back(watch_loop, api, std::string("/api/v1") + (ns.empty() ? "" : "/namespaces/" + ns) + "/pods?labelSelector=model",
                             std::string("pods"));
      }
      while (!g_stop) {
        bool by_event = g_wake.wait(interval);
        if (g_stop) break;
        if (!leader) continue;
        g_metrics.add(std::string("vpsa_operator_passes_total{trigger=\"") + (by_event ? "event" : "resync") + "\"}");
        run_once(api, ns, gpu);
        g_metrics.set("vpsa_operator_last_pass_timestamp_seconds", (double)time(nullptr));
        if (by_event) usleep(200 * 1000);  // a burst of events becomes one pass, not one pass each
      }
      health.stop();
      metrics_srv.stop();
      for (auto& t : threads) t.detach();  // blocked in reads with a timeout; the process is exiting
      return 0;
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "vpsa-operator: %s\n", e.what());
    return 1;
  }
  fprintf(stderr, "unknown mode %s\n", mode.c_str());
  return 2;
}
