"""Multi-model stack with KV-aware routing and a shared remote KV cache, end to end with real
processes: two Llama-3-8B replicas and one Mixtral-8x7B replica (all on this box's GPU when there
is only one), the cache server, and the router with ``--routing-logic kvaware`` hosting the KV
controller.  It is the single-box rehearsal of the eight-GPU deployment of
deploy/examples/values-17-kv-aware.yaml + values-06-shared-kv-cache.yaml.

What is checked and reported (one JSON line):
  * kvaware: repeats of a long prompt go to the replica whose KV the controller knows about
    (router log), for both models;
  * shared cache: the same conversation pinned to the OTHER Llama replica finds its prefix in the
    cache server (pages loaded from the remote tier, TTFT against the cold TTFT);
  * both models answer through one router; tokens/s of a short mixed load.

    python benchmarks/config5_multi_model.py [--small]     # --small: tiny CPU models, plumbing only
"""

import argparse
import json
import os
import re
import subprocess
import sys
import time
import urllib.request
import uuid
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUTE_RE = re.compile(r"Routing request (\S+) with session id (\S+) to (\S+) at ")


def http(url, body=None, headers=None, timeout=600):
    data = json.dumps(body).encode() if body is not None else None
    h = {"Content-Type": "application/json"} if data else {}
    h.update(headers or {})
    req = urllib.request.Request(url, data=data, headers=h)
    with urllib.request.urlopen(req, timeout=timeout) as r:  # noqa: S310
        raw = r.read()
    return json.loads(raw) if raw[:1] in (b"{", b"[") else raw.decode()


def wait_http(url, timeout, proc):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if proc.poll() is not None:
            raise RuntimeError(f"process behind {url} exited with code {proc.returncode}")
        try:
            urllib.request.urlopen(url, timeout=2).read()  # noqa: S310
            return
        except Exception:
            time.sleep(0.5)
    raise RuntimeError(f"{url} did not come up in {timeout}s")


def metric(text: str, name: str) -> float:
    tot = 0.0
    for ln in text.splitlines():
        if ln.startswith(name) and not ln.startswith("#"):
            try:
                tot += float(ln.rsplit(" ", 1)[1])
            except ValueError:
                pass
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true", help="tiny models on the CPU (plumbing rehearsal)")
    ap.add_argument("--prompt-tokens", type=int, default=0)
    ap.add_argument("--base-port", type=int, default=18700)
    ap.add_argument("--pages", type=int, default=0, help="KV pages per replica (0 = by model)")
    a = ap.parse_args()
    dense, moe = ("tiny-llama", "tiny-mixtral") if a.small else ("llama-3-8b", "mixtral-8x7b")
    n_words = a.prompt_tokens or (600 if a.small else 8000)
    log_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp"
    rp, cp, ctl = a.base_port, a.base_port + 9, a.base_port + 8
    engines = [("llama-a", dense, a.base_port + 1), ("llama-b", dense, a.base_port + 2),
               ("mixtral", moe, a.base_port + 3)]
    procs, logs = [], {}

    def say(msg):
        print(f"[config5 {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def spawn(name, cmd, env=None):
        logs[name] = open(os.path.join(log_dir, f"config5_{name}.log"), "w")
        p = subprocess.Popen(cmd, env=env or dict(os.environ), stdout=logs[name], stderr=subprocess.STDOUT, cwd=ROOT)
        procs.append(p)
        return p

    try:
        say("cache server")
        spawn("cache", [sys.executable, "-m", "vllm_production_stack_amd.cache_server", "127.0.0.1", str(cp)])
        say("router (kvaware, controller on %d)" % ctl)
        router = spawn("router", [
            sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1", "--port", str(rp),
            "--service-discovery", "static",
            "--static-backends", ",".join(f"http://127.0.0.1:{p}" for _, _, p in engines),
            "--static-models", ",".join(m for _, m, _ in engines),
            "--routing-logic", "kvaware", "--lmcache-controller-port", str(ctl), "--session-key", "x-user-id",
            "--kv-aware-threshold", "256", "--engine-stats-interval", "2", "--log-level", "info"])
        wait_http(f"http://127.0.0.1:{rp}/health", 60, router)
        eprocs = {}
        for i, (name, model, port) in enumerate(engines):
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LMCACHE_REMOTE_URL=f"lm://127.0.0.1:{cp}",
                       LMCACHE_REMOTE_SERDE="naive", LMCACHE_ENABLE_CONTROLLER="True",
                       LMCACHE_LMCACHE_INSTANCE_ID=name, LMCACHE_CONTROLLER_URL=f"127.0.0.1:{ctl}",
                       LMCACHE_LMCACHE_WORKER_PORT=str(a.base_port + 20 + i))
            pages = a.pages or (64 if a.small else (1500 if model == dense else 1200))
            cmd = [sys.executable, "-m", "vllm_production_stack_amd.engine.cli", "serve", model, "--host", "127.0.0.1",
                   "--port", str(port), "--max-model-len", "4096" if a.small else "32000",
                   "--enable-prefix-caching", "--num-gpu-blocks-override", str(pages), "--log-level", "info",
                   "--kv-transfer-config", json.dumps({"kv_connector": "LMCacheConnectorV1", "kv_role": "kv_both"})]
            if a.small:
                cmd += ["--device", "cpu", "--dtype", "float32"]
            say(f"engine {name} ({model}) on {port}")
            eprocs[name] = spawn(name, cmd, env)
            wait_http(f"http://127.0.0.1:{port}/health", 600, eprocs[name])  # one at a time: they share the card
        time.sleep(3)  # a stats scrape

        url = f"http://127.0.0.1:{rp}"
        by_port = {f"http://127.0.0.1:{p}": n for n, _, p in engines}
        models = http(url + "/v1/models")
        served = sorted(m["id"] for m in models["data"])
        say(f"router serves {served}")

        def ask(model, prompt, rid=None, pin=None, max_tokens=8):
            rid = rid or uuid.uuid4().hex
            t0 = time.time()
            q = f"?id={pin}" if pin else ""
            out = http(url + "/v1/completions" + q, {"model": model, "prompt": prompt, "max_tokens": max_tokens,
                                                     "temperature": 0}, {"X-Request-Id": rid})
            return rid, time.time() - t0, out

        def routed():
            logs["router"].flush()
            got = {}
            with open(os.path.join(log_dir, "config5_router.log")) as f:
                for ln in f:
                    m = ROUTE_RE.search(ln)
                    if m:
                        got[m.group(1)] = by_port.get(m.group(3), m.group(3))
            return got

        result = {"case": "config5_multi_model", "models": served, "prompt_words": n_words}
        # ---- KV-aware routing, per model
        for model, tag in ((dense, "llama"), (moe, "mixtral")):
            doc = f"{tag} shared document " * (n_words // 3)
            r1, t_cold, o1 = ask(model, doc)
            time.sleep(2.5)  # admitted chunks reach the controller asynchronously
            repeats = [ask(model, doc + f" question {k}") for k in range(3)]
            where = routed()
            holder = where[r1]
            result[f"{tag}_first_replica"] = holder
            result[f"{tag}_repeats_on_holder"] = sum(1 for r, _, _ in repeats if where[r] == holder)
            result[f"{tag}_cold_s"] = round(t_cold, 3)
            result[f"{tag}_repeat_s"] = round(min(t for _, t, _ in repeats), 3)
            result[f"{tag}_cached_tokens_on_repeat"] = (repeats[-1][2].get("usage", {}).get("prompt_tokens_details") or
                                                        {}).get("cached_tokens", 0)
        # ---- shared remote cache: pin the Llama conversation to the replica that has NOT seen it
        eng_list = http(url + "/engines")
        other = "llama-b" if result["llama_first_replica"] == "llama-a" else "llama-a"
        other_port = dict((n, p) for n, _, p in engines)[other]
        # static discovery lists the engines in backend order
        ids_in_order = [e["engine_id"] for e in eng_list]
        other_id = ids_in_order[[n for n, _, _ in engines].index(other)]
        before = http(f"http://127.0.0.1:{other_port}/metrics")
        doc = "llama shared document " * (n_words // 3)
        time.sleep(1.0)
        _, t_remote, o = ask(dense, doc + " moved", pin=other_id)
        after = http(f"http://127.0.0.1:{other_port}/metrics")
        result["moved_to"] = other
        result["moved_request_s"] = round(t_remote, 3)
        result["moved_cached_tokens"] = (o.get("usage", {}).get("prompt_tokens_details") or {}).get("cached_tokens", 0)
        result["remote_hit_tokens_on_other_replica"] = int(metric(after, "lmcache:num_hit_tokens_total") -
                                                           metric(before, "lmcache:num_hit_tokens_total"))
        cache_stats = http(f"http://127.0.0.1:{other_port}/debug/stats")
        result["kv_tiers_of_other_replica"] = cache_stats.get("kv_tiers") if isinstance(cache_stats, dict) else None
        # ---- short mixed load through the one router
        t0 = time.time()
        with ThreadPoolExecutor(12) as ex:
            futs = [ex.submit(ask, dense if k % 3 else moe, f"user {k} says hello " * 20, None, None, 32)
                    for k in range(24)]
            outs = [f.result() for f in futs]
        dt = time.time() - t0
        toks = sum(o[2]["usage"]["completion_tokens"] for o in outs)
        result["mixed_load_requests"] = len(outs)
        result["mixed_load_output_tok_per_s"] = round(toks / dt, 1)
        ok = (result["llama_repeats_on_holder"] == 3 and result["mixtral_repeats_on_holder"] == 3 and
              result["remote_hit_tokens_on_other_replica"] > 0)
        result["pass"] = bool(ok)
        print(json.dumps(result), flush=True)
        sys.exit(0 if ok else 1)
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=15)
            except Exception:
                p.kill()
        for f in logs.values():
            f.close()


if __name__ == "__main__":
    main()
