// Desired-state builders of the production-stack operator (API group
// production-stack.vllm.ai/v1alpha1).  Pure functions CR (JSON) -> Kubernetes objects (JSON):
// the control loop in operator_main.cpp applies them, tests call them directly.
//
// Behavioural reference (Go, kubebuilder):
//   VLLMRuntime  /root/reference/operator/internal/controller/vllmruntime_controller.go:57-734
//                (args :228-286, env :289-385, resources :388-410, sidecar :526-621)
//   VLLMRouter   /root/reference/operator/internal/controller/vllmrouter_controller.go:62-513
//   CacheServer  /root/reference/operator/internal/controller/cacheserver_controller.go:54-297
//   LoraAdapter  /root/reference/operator/internal/controller/loraadapter_controller.go:234-906
// Differences by design: the GPU resource name is configurable (default amd.com/gpu, the
// reference hard-codes nvidia.com/gpu); the router CRD accepts all five routing logics;
// LoRA placement implements the `ordered` and `equalized` algorithms, not only `default`.
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "../common/json.hpp"

namespace vop {
using vjson::Value;

inline Value env_var(const std::string& n, const std::string& v) {
  Value e = Value::object();
  e.set("name", n);
  e.set("value", v);
  return e;
}

inline Value meta(const std::string& name, const std::string& ns, const Value& owner, const std::string& app) {
  Value m = Value::object();
  m.set("name", name);
  m.set("namespace", ns);
  Value labels = Value::object();
  labels.set("app", app);
  m.set("labels", labels);
  if (owner.is_obj()) {
    Value ref = Value::object();
    ref.set("apiVersion", owner.s("apiVersion", "production-stack.vllm.ai/v1alpha1"));
    ref.set("kind", owner.s("kind"));
    ref.set("name", owner.s("metadata.name"));
    ref.set("uid", owner.s("metadata.uid"));
    ref.set("controller", true);
    ref.set("blockOwnerDeletion", true);
    Value refs = Value::array();
    refs.push(ref);
    m.set("ownerReferences", refs);
  }
  return m;
}

inline std::string num(double d) {
  char b[32];
  snprintf(b, sizeof b, "%lld", (long long)d);
  return b;
}

inline Value strategy(const std::string& kind) {
  Value s = Value::object();
  s.set("type", kind == "Recreate" ? "Recreate" : "RollingUpdate");
  return s;
}

inline Value service(const std::string& name, const std::string& ns, const Value& owner, const std::string& app,
                     int port, int target, const std::string& port_name = "http") {
  Value svc = Value::object();
  svc.set("apiVersion", "v1");
  svc.set("kind", "Service");
  svc.set("metadata", meta(name, ns, owner, app));
  Value spec = Value::object();
  Value sel = Value::object();
  sel.set("app", app);
  spec.set("selector", sel);
  Value p = Value::object();
  p.set("name", port_name);
  p.set("port", port);
  p.set("targetPort", target);
  p.set("protocol", "TCP");
  Value ports = Value::array();
  ports.push(p);
  spec.set("ports", ports);
  spec.set("type", "ClusterIP");
  svc.set("spec", spec);
  return svc;
}

// ------------------------------------------------------------------ VLLMRuntime
inline std::vector<std::string> runtime_args(const Value& cr) {
  const Value& sp = cr.at("spec");
  int port = (int)sp.i("vllmConfig.port", 8000, 1, 65535);
  std::vector<std::string> a = {sp.s("model.modelURL"), "--host", "0.0.0.0", "--port", num(port)};
  if (sp.flag("model.enableLoRA")) a.push_back("--enable-lora");
  if (sp.flag("model.enableTool")) a.push_back("--enable-auto-tool-choice");
  if (!sp.s("model.toolCallParser").empty()) {
    a.push_back("--tool-call-parser");
    a.push_back(sp.s("model.toolCallParser"));
  }
  a.push_back(sp.flag("vllmConfig.enableChunkedPrefill") ? "--enable-chunked-prefill" : "--no-enable-chunked-prefill");
  a.push_back(sp.flag("vllmConfig.enablePrefixCaching") ? "--enable-prefix-caching" : "--no-enable-prefix-caching");
  if (sp.n("model.maxModelLen") > 0) {
    a.push_back("--max-model-len");
    a.push_back(num(sp.n("model.maxModelLen")));
  }
  if (!sp.s("model.dtype").empty()) {
    a.push_back("--dtype");
    a.push_back(sp.s("model.dtype"));
  }
  if (sp.n("vllmConfig.tensorParallelSize") > 0) {
    a.push_back("--tensor-parallel-size");
    a.push_back(num(sp.n("vllmConfig.tensorParallelSize")));
  }
  if (sp.n("model.maxNumSeqs") > 0) {
    a.push_back("--max-num-seqs");
    a.push_back(num(sp.n("model.maxNumSeqs")));
  }
  if (!sp.s("vllmConfig.gpuMemoryUtilization").empty()) {
    a.push_back("--gpu_memory_utilization");
    a.push_back(sp.s("vllmConfig.gpuMemoryUtilization"));
  }
  if (sp.n("vllmConfig.maxLoras") > 0) {
    a.push_back("--max_loras");
    a.push_back(num(sp.n("vllmConfig.maxLoras")));
  }
  const Value& extra = sp.at("vllmConfig.extraArgs");
  if (extra.is_arr())
    for (auto& e : *extra.arr) a.push_back(e.str);
  if (sp.flag("lmCacheConfig.enabled")) {
    a.push_back("--kv-transfer-config");
    a.push_back(sp.flag("vllmConfig.v1") ? "{\"kv_connector\":\"LMCacheConnectorV1\",\"kv_role\":\"kv_both\"}"
                                          : "{\"kv_connector\":\"LMCacheConnector\",\"kv_role\":\"kv_both\"}");
  }
  return a;
}

inline Value runtime_env(const Value& cr) {
  const Value& sp = cr.at("spec");
  Value env = Value::array();
  env.push(env_var("VLLM_USE_V1", sp.flag("vllmConfig.v1") ? "1" : "0"));
  if (sp.flag("model.enableLoRA")) env.push(env_var("VLLM_ALLOW_RUNTIME_LORA_UPDATING", "True"));
  if (!sp.s("model.hfTokenSecret.name").empty()) {
    Value e = Value::object();
    e.set("name", "HF_TOKEN");
    Value ref = Value::object();
    ref.set("name", sp.s("model.hfTokenSecret.name"));
    ref.set("key", sp.s("model.hfTokenName", "token"));
    Value from = Value::object();
    from.set("secretKeyRef", ref);
    e.set("valueFrom", from);
    env.push(e);
  }
  if (sp.flag("lmCacheConfig.enabled")) {
    env.push(env_var("LMCACHE_LOG_LEVEL", "DEBUG"));
    env.push(env_var("LMCACHE_USE_EXPERIMENTAL", "True"));
    env.push(env_var("VLLM_RPC_TIMEOUT", "1000000"));
    if (!sp.s("lmCacheConfig.cpuOffloadingBufferSize").empty()) {
      env.push(env_var("LMCACHE_LOCAL_CPU", "True"));
      env.push(env_var("LMCACHE_MAX_LOCAL_CPU_SIZE", sp.s("lmCacheConfig.cpuOffloadingBufferSize")));
    }
    if (!sp.s("lmCacheConfig.diskOffloadingBufferSize").empty()) {
      env.push(env_var("LMCACHE_LOCAL_DISK", "True"));
      env.push(env_var("LMCACHE_MAX_LOCAL_DISK_SIZE", sp.s("lmCacheConfig.diskOffloadingBufferSize")));
    }
    if (!sp.s("lmCacheConfig.remoteUrl").empty()) {
      env.push(env_var("LMCACHE_REMOTE_URL", sp.s("lmCacheConfig.remoteUrl")));
      env.push(env_var("LMCACHE_REMOTE_SERDE", sp.s("lmCacheConfig.remoteSerde", "naive")));
    }
  }
  const Value& user = sp.at("vllmConfig.env");
  if (user.is_arr())
    for (auto& e : *user.arr) env.push(env_var(e.s("name"), e.s("value")));
  return env;
}

inline Value resources(const Value& res, const std::string& gpu_resource) {
  Value req = Value::object(), lim = Value::object();
  if (!res.s("cpu").empty()) {
    req.set("cpu", res.s("cpu"));
    lim.set("cpu", res.s("cpu"));
  }
  if (!res.s("memory").empty()) {
    req.set("memory", res.s("memory"));
    lim.set("memory", res.s("memory"));
  }
  if (!res.s("gpu").empty()) {
    req.set(gpu_resource, res.s("gpu"));
    lim.set(gpu_resource, res.s("gpu"));
  }
  Value r = Value::object();
  r.set("requests", req);
  r.set("limits", lim);
  return r;
}

inline Value probe(const std::string& path, int port, int initial, int period, int failures) {
  Value get = Value::object();
  get.set("path", path);
  get.set("port", port);
  Value p = Value::object();
  p.set("httpGet", get);
  p.set("initialDelaySeconds", initial);
  p.set("periodSeconds", period);
  p.set("failureThreshold", failures);
  return p;
}

inline std::vector<Value> desired_for_runtime(const Value& cr, const std::string& gpu_resource = "amd.com/gpu") {
  const Value& sp = cr.at("spec");
  const std::string name = cr.s("metadata.name"), ns = cr.s("metadata.namespace", "default");
  int port = (int)sp.i("vllmConfig.port", 8000, 1, 65535);
  std::vector<Value> out;
  out.push_back(service(name, ns, cr, name, 80, port));
  const Value& st = sp.at("storageConfig");
  bool pvc = st.flag("enabled");
  if (pvc) {
    Value p = Value::object();
    p.set("apiVersion", "v1");
    p.set("kind", "PersistentVolumeClaim");
    p.set("metadata", meta(name + "-pvc", ns, cr, name));
    Value ps = Value::object();
    Value modes = Value::array();
    modes.push(st.s("accessMode", "ReadWriteOnce"));
    ps.set("accessModes", modes);
    if (!st.s("storageClassName").empty()) ps.set("storageClassName", st.s("storageClassName"));
    Value rq = Value::object();
    rq.set("storage", st.s("size", "10Gi"));
    Value rr = Value::object();
    rr.set("requests", rq);
    ps.set("resources", rr);
    p.set("spec", ps);
    out.push_back(p);
  }
  Value c = Value::object();
  c.set("name", "vllm");
  c.set("image", sp.s("deploymentConfig.image.name", sp.s("deploymentConfig.image.registry") + "/" + sp.s("deploymentConfig.image.name")));
  if (!sp.s("deploymentConfig.image.registry").empty())
    c.set("image", sp.s("deploymentConfig.image.registry") + "/" + sp.s("deploymentConfig.image.name"));
  if (!sp.s("deploymentConfig.image.pullPolicy").empty()) c.set("imagePullPolicy", sp.s("deploymentConfig.image.pullPolicy"));
  Value cmd = Value::array();
  cmd.push("vllm");
  cmd.push("serve");
  c.set("command", cmd);
  Value args = Value::array();
  for (auto& a : runtime_args(cr)) args.push(a);
  c.set("args", args);
  Value env = runtime_env(cr);
  if (pvc) env.push(env_var("HF_HOME", st.s("mountPath", "/data")));
  c.set("env", env);
  Value cp = Value::object();
  cp.set("name", "http");
  cp.set("containerPort", port);
  Value cports = Value::array();
  cports.push(cp);
  c.set("ports", cports);
  c.set("resources", resources(sp.at("deploymentConfig.resources"), sp.s("deploymentConfig.gpuResourceName", gpu_resource)));
  c.set("readinessProbe", probe("/health", port, 30, 20, 10));
  c.set("livenessProbe", probe("/health", port, 300, 20, 10));
  Value mounts = Value::array(), volumes = Value::array();
  if (pvc) {
    Value m = Value::object();
    m.set("name", "model-storage");
    m.set("mountPath", st.s("mountPath", "/data"));
    mounts.push(m);
    Value v = Value::object();
    v.set("name", "model-storage");
    Value claim = Value::object();
    claim.set("claimName", name + "-pvc");
    v.set("persistentVolumeClaim", claim);
    volumes.push(v);
  }
  if (sp.n("vllmConfig.tensorParallelSize") > 1) {
    Value m = Value::object();
    m.set("name", "shm");
    m.set("mountPath", "/dev/shm");
    mounts.push(m);
    Value v = Value::object();
    v.set("name", "shm");
    Value ed = Value::object();
    ed.set("medium", "Memory");
    ed.set("sizeLimit", "20Gi");
    v.set("emptyDir", ed);
    volumes.push(v);
  }
  if (mounts.size()) c.set("volumeMounts", mounts);
  Value containers = Value::array();
  containers.push(c);
  if (sp.flag("deploymentConfig.sidecarConfig.enabled")) {
    const Value& sc = sp.at("deploymentConfig.sidecarConfig");
    Value s = Value::object();
    s.set("name", sc.s("name", "sidecar"));
    s.set("image", sc.s("image.registry").empty() ? sc.s("image.name") : sc.s("image.registry") + "/" + sc.s("image.name"));
    Value senv = Value::array();
    senv.push(env_var("PORT", "30090"));
    senv.push(env_var("LORA_DOWNLOAD_BASE_DIR", sc.s("mountPath", "/data") + "/lora-adapters"));
    s.set("env", senv);
    if (mounts.size()) s.set("volumeMounts", mounts);
    containers.push(s);
  }
  Value podspec = Value::object();
  podspec.set("containers", containers);
  if (volumes.size()) podspec.set("volumes", volumes);
  Value tmeta = Value::object();
  Value tl = Value::object();
  tl.set("app", name);
  tl.set("model", sp.s("model.modelURL").substr(sp.s("model.modelURL").rfind('/') == std::string::npos ? 0 : sp.s("model.modelURL").rfind('/') + 1));
  tmeta.set("labels", tl);
  Value tmpl = Value::object();
  tmpl.set("metadata", tmeta);
  tmpl.set("spec", podspec);
  Value dsp = Value::object();
  dsp.set("replicas", (int)sp.i("deploymentConfig.replicas", 1, 0, 100000));
  Value sel = Value::object();
  Value ml = Value::object();
  ml.set("app", name);
  sel.set("matchLabels", ml);
  dsp.set("selector", sel);
  dsp.set("strategy", strategy(sp.s("deploymentConfig.deploymentStrategy", "RollingUpdate")));
  dsp.set("template", tmpl);
  Value dep = Value::object();
  dep.set("apiVersion", "apps/v1");
  dep.set("kind", "Deployment");
  dep.set("metadata", meta(name, ns, cr, name));
  dep.set("spec", dsp);
  out.push_back(dep);
  return out;
}

// ------------------------------------------------------------------ VLLMRouter
inline std::vector<std::string> router_args(const Value& cr) {
  const Value& sp = cr.at("spec");
  const std::string ns = cr.s("metadata.namespace", "default");
  std::vector<std::string> a = {"--host", "0.0.0.0", "--port", num(sp.n("port", 80))};
  std::string sd = sp.s("serviceDiscovery", "k8s");
  a.push_back("--service-discovery");
  a.push_back(sd);
  if (sd == "k8s") {
    a.push_back("--k8s-namespace");
    a.push_back(ns);
    if (!sp.s("k8sLabelSelector").empty()) {
      a.push_back("--k8s-label-selector");
      a.push_back(sp.s("k8sLabelSelector"));
    }
  } else {
    a.push_back("--static-backends");
    a.push_back(sp.s("staticBackends"));
    a.push_back("--static-models");
    a.push_back(sp.s("staticModels"));
  }
  a.push_back("--routing-logic");
  a.push_back(sp.s("routingLogic", "roundrobin"));
  if (!sp.s("sessionKey").empty()) {
    a.push_back("--session-key");
    a.push_back(sp.s("sessionKey"));
  }
  if (sp.n("engineScrapeInterval") > 0) {
    a.push_back("--engine-stats-interval");
    a.push_back(num(sp.n("engineScrapeInterval")));
  }
  if (sp.n("requestStatsWindow") > 0) {
    a.push_back("--request-stats-window");
    a.push_back(num(sp.n("requestStatsWindow")));
  }
  const Value& extra = sp.at("extraArgs");
  if (extra.is_arr())
    for (auto& e : *extra.arr) a.push_back(e.str);
  return a;
}

inline bool valid_routing_logic(const std::string& s) {
  static const char* ok[] = {"roundrobin", "session", "kvaware", "prefixaware", "disaggregated_prefill"};
  for (auto* k : ok)
    if (s == k) return true;
  return false;
}

inline std::vector<Value> desired_for_router(const Value& cr) {
  const Value& sp = cr.at("spec");
  const std::string name = cr.s("metadata.name"), ns = cr.s("metadata.namespace", "default");
  const std::string sa = sp.s("serviceAccountName", name + "-sa");
  int port = (int)sp.i("port", 80, 1, 65535);
  std::vector<Value> out;
  Value acct = Value::object();
  acct.set("apiVersion", "v1");
  acct.set("kind", "ServiceAccount");
  acct.set("metadata", meta(sa, ns, cr, name));
  out.push_back(acct);
  Value role = Value::object();
  role.set("apiVersion", "rbac.authorization.k8s.io/v1");
  role.set("kind", "Role");
  role.set("metadata", meta("pod-viewer-role", ns, cr, name));
  Value rule = Value::object();
  Value groups = Value::array();
  groups.push("");
  rule.set("apiGroups", groups);
  Value rsrc = Value::array();
  for (auto* r : {"pods", "services", "endpoints"}) rsrc.push(r);
  rule.set("resources", rsrc);
  Value verbs = Value::array();
  for (auto* v : {"get", "list", "watch", "patch"}) verbs.push(v);
  rule.set("verbs", verbs);
  Value rules = Value::array();
  rules.push(rule);
  role.set("rules", rules);
  out.push_back(role);
  Value rb = Value::object();
  rb.set("apiVersion", "rbac.authorization.k8s.io/v1");
  rb.set("kind", "RoleBinding");
  rb.set("metadata", meta(name + "-pod-viewer-binding", ns, cr, name));
  Value subj = Value::object();
  subj.set("kind", "ServiceAccount");
  subj.set("name", sa);
  subj.set("namespace", ns);
  Value subjects = Value::array();
  subjects.push(subj);
  rb.set("subjects", subjects);
  Value rref = Value::object();
  rref.set("kind", "Role");
  rref.set("name", "pod-viewer-role");
  rref.set("apiGroup", "rbac.authorization.k8s.io");
  rb.set("roleRef", rref);
  out.push_back(rb);
  out.push_back(service(name, ns, cr, name, port, port));
  Value c = Value::object();
  c.set("name", "router");
  c.set("image", sp.s("image.registry").empty() ? sp.s("image.name") : sp.s("image.registry") + "/" + sp.s("image.name"));
  Value args = Value::array();
  for (auto& a : router_args(cr)) args.push(a);
  c.set("args", args);
  Value env = Value::array();
  const Value& user = sp.at("env");
  if (user.is_arr())
    for (auto& e : *user.arr) env.push(env_var(e.s("name"), e.s("value")));
  if (!sp.s("vllmApiKeySecret.name").empty()) {
    Value e = Value::object();
    e.set("name", "VLLM_API_KEY");
    Value ref = Value::object();
    ref.set("name", sp.s("vllmApiKeySecret.name"));
    ref.set("key", sp.s("vllmApiKeyName", "VLLM_API_KEY"));
    Value from = Value::object();
    from.set("secretKeyRef", ref);
    e.set("valueFrom", from);
    env.push(e);
  }
  c.set("env", env);
  Value cp = Value::object();
  cp.set("name", "http");
  cp.set("containerPort", port);
  Value cports = Value::array();
  cports.push(cp);
  c.set("ports", cports);
  c.set("resources", resources(sp.at("resources"), "amd.com/gpu"));
  c.set("livenessProbe", probe("/health", port, 30, 5, 3));
  Value containers = Value::array();
  containers.push(c);
  Value podspec = Value::object();
  podspec.set("serviceAccountName", sa);
  podspec.set("containers", containers);
  Value tl = Value::object();
  tl.set("app", name);
  Value tmeta = Value::object();
  tmeta.set("labels", tl);
  Value tmpl = Value::object();
  tmpl.set("metadata", tmeta);
  tmpl.set("spec", podspec);
  Value dsp = Value::object();
  dsp.set("replicas", (int)sp.i("replicas", 1, 0, 100000));
  Value ml = Value::object();
  ml.set("app", name);
  Value sel = Value::object();
  sel.set("matchLabels", ml);
  dsp.set("selector", sel);
  dsp.set("template", tmpl);
  Value dep = Value::object();
  dep.set("apiVersion", "apps/v1");
  dep.set("kind", "Deployment");
  dep.set("metadata", meta(name, ns, cr, name));
  dep.set("spec", dsp);
  out.push_back(dep);
  return out;
}

// ------------------------------------------------------------------ CacheServer
inline std::vector<Value> desired_for_cacheserver(const Value& cr) {
  const Value& sp = cr.at("spec");
  const std::string name = cr.s("metadata.name"), ns = cr.s("metadata.namespace", "default");
  int port = (int)sp.i("port", 8000, 1, 65535);
  std::vector<Value> out;
  out.push_back(service(name, ns, cr, name, 80, port));
  Value c = Value::object();
  c.set("name", "cache-server");
  c.set("image", sp.s("image.registry").empty() ? sp.s("image.name") : sp.s("image.registry") + "/" + sp.s("image.name"));
  Value cmd = Value::array();
  for (auto& s : {std::string("lmcache_experimental_server"), std::string("0.0.0.0"), num(port)}) cmd.push(s);
  c.set("command", cmd);
  Value cp = Value::object();
  cp.set("containerPort", port);
  Value cports = Value::array();
  cports.push(cp);
  c.set("ports", cports);
  c.set("resources", resources(sp.at("resources"), "amd.com/gpu"));
  Value containers = Value::array();
  containers.push(c);
  Value podspec = Value::object();
  podspec.set("containers", containers);
  Value tl = Value::object();
  tl.set("app", name);
  Value tmeta = Value::object();
  tmeta.set("labels", tl);
  Value tmpl = Value::object();
  tmpl.set("metadata", tmeta);
  tmpl.set("spec", podspec);
  Value dsp = Value::object();
  dsp.set("replicas", (int)sp.i("replicas", 1, 0, 100000));
  Value ml = Value::object();
  ml.set("app", name);
  Value sel = Value::object();
  sel.set("matchLabels", ml);
  dsp.set("selector", sel);
  dsp.set("strategy", strategy(sp.s("deploymentStrategy", "RollingUpdate")));
  dsp.set("template", tmpl);
  Value dep = Value::object();
  dep.set("apiVersion", "apps/v1");
  dep.set("kind", "Deployment");
  dep.set("metadata", meta(name, ns, cr, name));
  dep.set("spec", dsp);
  out.push_back(dep);
  return out;
}

// ------------------------------------------------------------------ LoraAdapter placement
struct PodState {
  std::string name, ip;
  bool ready = false;
  std::vector<std::string> adapters;  // currently loaded adapter names
};

struct LoraAction {
  std::string op;  // load | unload
  std::string pod, ip, adapter, path;
};

// Which pods should hold the adapter.  default: first min(replicas, ready) pods by name;
// ordered: the same but stable by pod ordinal suffix; equalized: pods with the fewest loaded
// adapters first (spreads adapters over the fleet).  replicas <= 0 means every ready pod.
inline std::vector<std::string> place_adapter(const Value& cr, std::vector<PodState> pods) {
  const std::string algo = cr.s("spec.loraAdapterDeploymentConfig.algorithm", "default");
  int replicas = (int)cr.i("spec.loraAdapterDeploymentConfig.replicas", 0, 0, 100000);
  const std::string adapter = cr.s("spec.adapterSource.adapterName");
  std::vector<PodState> ready;
  for (auto& p : pods)
    if (p.ready) ready.push_back(p);
  if (algo == "equalized") {
    std::stable_sort(ready.begin(), ready.end(), [&](const PodState& a, const PodState& b) {
      auto load = [&](const PodState& p) {
        size_t n = p.adapters.size();
        if (std::find(p.adapters.begin(), p.adapters.end(), adapter) != p.adapters.end()) --n;  // sticky
        return n;
      };
      if (load(a) != load(b)) return load(a) < load(b);
      return a.name < b.name;
    });
  } else {
    std::sort(ready.begin(), ready.end(), [](const PodState& a, const PodState& b) { return a.name < b.name; });
  }
  size_t want = replicas <= 0 ? ready.size() : std::min((size_t)replicas, ready.size());
  std::vector<std::string> out;
  for (size_t i = 0; i < want; ++i) out.push_back(ready[i].name);
  return out;
}

inline std::vector<LoraAction> plan_lora(const Value& cr, const std::vector<PodState>& pods, bool deleting) {
  const std::string adapter = cr.s("spec.adapterSource.adapterName");
  std::string path = cr.s("spec.adapterSource.adapterPath");
  std::vector<std::string> target = deleting ? std::vector<std::string>{} : place_adapter(cr, pods);
  std::vector<LoraAction> acts;
  for (auto& p : pods) {
    bool has = std::find(p.adapters.begin(), p.adapters.end(), adapter) != p.adapters.end();
    bool want = std::find(target.begin(), target.end(), p.name) != target.end();
    if (want && !has) acts.push_back({"load", p.name, p.ip, adapter, path});
    if (!want && has) acts.push_back({"unload", p.name, p.ip, adapter, path});
  }
  return acts;
}

}  // namespace vop
