"""Whole-node serving layouts for ``bench.py --layout ...``: the multi-round QA workload against

  pd:PxD       P prefill + D decode replicas of Llama-3-8B (one GPU each), router in
               ``disaggregated_prefill`` mode over both pools, KV pages pushed prefill -> decode
               (default pd:2x6: deploy/examples/values-16-disaggregated-prefill.yaml on one node);
  multimodel   six Llama-3-8B replicas + one tensor-parallel (TP2) Mixtral-8x7B replica, the KV
               cache server, router with ``kvaware`` routing hosting the KV controller
               (deploy/examples/values-17-kv-aware.yaml + values-06-shared-kv-cache.yaml).

One launcher process (this one, which never touches a GPU) starts every engine as a process of
its own, pinned with HIP_VISIBLE_DEVICES before its first GPU call, then the router, then the
load generators (``bench.py --client``), and prints ONE JSON line.  On a machine with fewer GPUs
than the layout wants, ``VPSA_BENCH_SINGLE_DEVICE=1`` puts every engine on the first card with
capped KV pools (``--num-gpu-blocks``) and runs the tensor-parallel ranks over gloo
(``VPSA_TP_BACKEND=gloo``): a rehearsal of the launch path and of every protocol, not a measurement.

Reference counterparts: the two-pool request flow of
/root/reference/src/vllm_router/services/request_service/request.py:339-431, the engine roles of
/root/reference/helm/templates/deployment-vllm-multi.yaml:182-186,267-305, KV-aware routing of
/root/reference/src/vllm_router/routers/routing_logic.py:222-344.
"""

from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import time
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUTE_RE = re.compile(r"Routing request (\S+) with session id (\S+) to (\S+) at ")


def _http(url, body=None, timeout=60):
    data = json.dumps(body).encode() if body is not None else None
    req = urllib.request.Request(url, data=data, headers={"Content-Type": "application/json"} if data else {})
    with urllib.request.urlopen(req, timeout=timeout) as r:  # noqa: S310
        raw = r.read()
    return json.loads(raw) if raw[:1] in (b"{", b"[") else raw.decode()


def _wait_http(url, timeout, proc):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if proc.poll() is not None:
            raise RuntimeError(f"process behind {url} exited with code {proc.returncode}")
        try:
            urllib.request.urlopen(url, timeout=2).read()  # noqa: S310
            return
        except Exception:  # noqa: BLE001
            time.sleep(0.5)
    raise RuntimeError(f"{url} did not come up in {timeout}s")


def _pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))] if xs else None


class Stack:
    """Child processes of one layout run, their logs, and their orderly end."""

    def __init__(self, tag: str):
        self.tag = tag
        self.procs: list[subprocess.Popen] = []
        self.logs: dict[str, object] = {}
        d = os.path.join(ROOT, "gpurun_out")
        self.log_dir = d if os.path.isdir(d) else "/tmp"

    def say(self, msg):
        print(f"[{self.tag} {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def log_path(self, name):
        return os.path.join(self.log_dir, f"{self.tag}_{name}.log")

    def spawn(self, name, cmd, env=None, **kw):
        self.logs[name] = open(self.log_path(name), "w")
        p = subprocess.Popen(cmd, env=env or dict(os.environ), stdout=self.logs[name], stderr=subprocess.STDOUT,
                             cwd=ROOT, **kw)
        self.procs.append(p)
        return p

    def stop(self):
        for p in self.procs:
            try:
                p.terminate()
            except Exception:  # noqa: BLE001
                pass
        for p in self.procs:
            try:
                p.wait(timeout=20)
            except Exception:  # noqa: BLE001
                p.kill()
        for f in self.logs.values():
            f.close()

    def tail(self, name, n=30):
        try:
            with open(self.log_path(name)) as f:
                return f.read().splitlines()[-n:]
        except OSError:
            return []


def _gpu_plan(n_slots: int, device: str = "cuda"):
    """GPU index (as HIP_VISIBLE_DEVICES string) of each engine slot; ``single`` when they share a card."""
    single = os.environ.get("VPSA_BENCH_SINGLE_DEVICE") == "1" or device != "cuda"
    if single:
        return ["0"] * n_slots, True
    import torch

    have = torch.cuda.device_count()  # does not initialise the GPU
    if have < n_slots:
        raise SystemExit(f"this layout wants {n_slots} GPUs and the machine has {have}: "
                         "set VPSA_BENCH_SINGLE_DEVICE=1 (and --num-gpu-blocks) for a rehearsal on one card")
    return [str(i) for i in range(n_slots)], False


def _engine_cmd(a, model, port, pages, extra=()):
    cmd = [sys.executable, "-m", "vllm_production_stack_amd.engine.cli", "serve", model, "--host", "127.0.0.1",
           "--port", str(port), "--max-model-len", str(a.max_model_len), "--enable-prefix-caching",
           "--max-num-seqs", "256", "--max-num-batched-tokens", str(a.max_num_batched_tokens), "--log-level", "info"]
    if pages:
        cmd += ["--num-gpu-blocks-override", str(pages)]
    if a.device != "cuda":
        cmd += ["--device", "cpu", "--dtype", "float32"]
    return cmd + list(extra)


def _run_clients(stack: Stack, a, base_url: str, groups: list[tuple[str, list[int], list[int]]]):
    """``groups``: (model, user ids, warm-up user ids).  One load generator per group (the
    ``bench.py --client`` protocol: WARMUP -> READY -> GO -> RESULT).  Returns (rows by model, wall seconds)."""
    clients = []
    for model, users, wusers in groups:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--client", "--base-url", base_url, "--model", model,
               "--steps", str(a.steps), "--warmup", str(a.warmup), "--sys-len", str(a.sys_len),
               "--history-len", str(a.history_len), "--answer-len", str(a.answer_len),
               "--user-ids", ",".join(map(str, users)), "--warmup-user-ids", ",".join(map(str, wusers))]
        p = subprocess.Popen(cmd, cwd=ROOT, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
        stack.procs.append(p)
        clients.append((model, p))

    def wait_line(p, prefix):
        while True:
            line = p.stdout.readline()
            if not line:
                raise RuntimeError(f"load generator exited early (code {p.poll()})")
            if line.startswith(prefix):
                return line[len(prefix):].strip()

    for _, p in clients:
        p.stdin.write("WARMUP\n")
        p.stdin.flush()
    for _, p in clients:
        wait_line(p, "READY")
    t0 = time.time()
    for _, p in clients:
        p.stdin.write("GO\n")
        p.stdin.flush()
    rows: dict[str, list] = {}
    for model, p in clients:
        rows.setdefault(model, []).extend(json.loads(wait_line(p, "RESULT "))["rows"])
    return rows, time.time() - t0


def _summary(rows, wall):
    ok = [r for r in rows if r["ttft"] >= 0]
    out = sum(r["generation_tokens"] for r in rows)
    return {"requests": len(rows), "failed_requests": len(rows) - len(ok),
            "output_tok_per_s": round(out / wall, 2), "input_tok_per_s": round(sum(r["prompt_tokens"] for r in rows) / wall, 1),
            "p50_ttft_s": round(_pct([r["ttft"] for r in ok], 0.5), 4) if ok else None,
            "p90_ttft_s": round(_pct([r["ttft"] for r in ok], 0.9), 4) if ok else None,
            "p50_ttft_warm_rounds_s": round(_pct([r["ttft"] for r in ok if r["question_id"] > 1], 0.5) or 0, 4)}


def _result_head(a, metric, value, n_gpus, single, parallelism, users):
    label = "Llama-3-8B" if a.model == "llama-3-8b" else a.model
    return {"metric": metric, "value": value, "unit": "tokens/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.device == "cuda" else "fp32",
            "data": "synthetic multi-turn prompts; random-init weights",
            "rehearsal_on_one_card": bool(single),
            "config": {"model": label, "global_batch": users, "seq_len": a.sys_len + a.history_len,
                       "answer_len": a.answer_len, "rounds": a.steps, "parallelism": parallelism}}


# ---------------------------------------------------------------------------------------------- pd:PxD
def run_pd(a, n_prefill: int, n_decode: int) -> dict:
    stack = Stack(f"pd{n_prefill}x{n_decode}")
    gpus, single = _gpu_plan(n_prefill + n_decode, a.device)
    model = a.model
    engines = [(f"prefill-{i}", "prefill", a.base_port + 1 + i) for i in range(n_prefill)] + \
              [(f"decode-{i}", "decode", a.base_port + 1 + n_prefill + i) for i in range(n_decode)]
    try:
        # decode replicas first: each sizes (and exports) its pool from what is free on its card
        order = engines[n_prefill:] + engines[:n_prefill]
        for name, role, port in order:
            slot = [e[0] for e in engines].index(name)
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HIP_VISIBLE_DEVICES=gpus[slot],
                       LMCACHE_ENABLE_NIXL="True", LMCACHE_NIXL_ROLE="sender" if role == "prefill" else "receiver",
                       LMCACHE_NIXL_RECEIVER_HOST="127.0.0.1", LMCACHE_LOCAL_CPU="False")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
                env.pop(k, None)
            kv = {"kv_connector": "LMCacheConnectorV1", "kv_role": "kv_producer" if role == "prefill" else "kv_consumer"}
            stack.say(f"engine {name} on GPU {gpus[slot]} port {port}")
            p = stack.spawn(name, _engine_cmd(a, model, port, a.num_gpu_blocks, ["--kv-transfer-config", json.dumps(kv)]), env)
            _wait_http(f"http://127.0.0.1:{port}/health", 900, p)
        rp = a.base_port
        router = stack.spawn("router", [
            sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1", "--port", str(rp),
            "--service-discovery", "static",
            "--static-backends", ",".join(f"http://127.0.0.1:{p}" for _, _, p in engines),
            "--static-models", ",".join([model] * len(engines)),
            "--static-model-labels", ",".join(r for _, r, _ in engines),
            "--routing-logic", "disaggregated_prefill", "--prefill-model-labels", "prefill",
            "--decode-model-labels", "decode", "--session-key", "x-user-id", "--log-level", "info"])
        _wait_http(f"http://127.0.0.1:{rp}/health", 60, router)
        n_users = a.users_per_gpu * n_decode  # conversations live on the decode pool
        users = list(range(n_users))
        wusers = list(range(10_000_000, 10_000_000 + max(1, a.warmup_users_per_gpu) * n_decode))
        n_clients = max(1, min(4, n_decode // 2))
        groups = [(model, users[c::n_clients], wusers[c::n_clients]) for c in range(n_clients)]
        stack.say(f"{n_users} users x {a.steps} rounds through the router")
        rows_by_model, wall = _run_clients(stack, a, f"http://127.0.0.1:{rp}", groups)
        rows = rows_by_model[model]
        pools = {"prefill": [], "decode": []}
        for name, role, port in engines:
            st = _http(f"http://127.0.0.1:{port}/debug/stats")
            ent = {"replica": name, "gpu": gpus[[e[0] for e in engines].index(name)], "rows": st["prof"]["rows"],
                   "steps": st["prof"]["steps"], "busy_s": round(st["prof"]["t_execute"], 2)}
            if role == "prefill":
                push = st.get("pd_push") or {}
                ent["pushed_pages"] = push.get("pages", 0)
                ent["pushed_gb"] = round(push.get("pages", 0) * st.get("page_bytes", 0) / 1e9, 2)
                ent["pushes_by_transport"] = push.get("by_mode", {})
                ent["push_jobs"] = push.get("jobs", 0)
                ent["push_tail_ms_per_request"] = round(1e3 * push.get("tail_seconds", 0.0) / max(1, push.get("pushes", 1)), 2)
                ent["failed_pushes"] = push.get("failed", 0)
            else:
                ent["prefix_hit_tokens"] = st.get("prefix_hits")
                ent["prefix_query_tokens"] = st.get("prefix_queries")
            pools[role].append(ent)
        hops: dict[str, list[str]] = {}
        by_url = {f"http://127.0.0.1:{p}": n for n, _, p in engines}
        stack.logs["router"].flush()
        with open(stack.log_path("router")) as f:
            for ln in f:
                m = ROUTE_RE.search(ln)
                if m:
                    hops.setdefault(m.group(1), []).append(by_url.get(m.group(3), m.group(3)))
        two_hop = sum(1 for h in hops.values() if len(h) == 2 and h[0].startswith("prefill") and h[1].startswith("decode"))
        s = _summary(rows, wall)
        res = _result_head(a, f"output tok/s (+ p50 TTFT), multi-round QA, disaggregated prefill {n_prefill}P+{n_decode}D",
                           s["output_tok_per_s"], n_prefill + n_decode, single,
                           f"pd:{n_prefill}x{n_decode} (prefill pool + decode pool behind the router)", n_users)
        res.update(s)
        res.update({"elapsed_s": round(wall, 3), "requests_routed_prefill_then_decode": two_hop,
                    "prefill_pool": pools["prefill"], "decode_pool": pools["decode"],
                    "prefill_pool_input_tok_per_s": round(sum(e["rows"] for e in pools["prefill"]) / wall, 1),
                    "decode_pool_output_tok_per_s": s["output_tok_per_s"],
                    "kv_pushed_gb_by_transport": _sum_by_transport(pools["prefill"])})
        return res
    except Exception:
        for name in list(stack.logs):
            print(f"==== {name}\n" + "\n".join(stack.tail(name, 25)), file=sys.stderr)
        raise
    finally:
        stack.stop()


def _sum_by_transport(prefill_pool):
    out: dict[str, float] = {}
    for e in prefill_pool:
        modes = e.get("pushes_by_transport") or {}
        tot = sum(modes.values()) or 1
        for m, n in modes.items():
            out[m] = round(out.get(m, 0.0) + e.get("pushed_gb", 0.0) * n / tot, 2)
    return out


# ------------------------------------------------------------------------------------------ multimodel
def run_multimodel(a, n_dense: int = 6, tp: int = 2) -> dict:
    stack = Stack("multimodel")
    gpus, single = _gpu_plan(n_dense + tp, a.device)
    dense = a.model
    moe = "tiny-mixtral" if dense.startswith("tiny") else "mixtral-8x7b"
    rp, ctl, cp = a.base_port, a.base_port + 40, a.base_port + 41
    engines = [(f"llama-{i}", dense, a.base_port + 1 + i) for i in range(n_dense)] + [("mixtral", moe, a.base_port + 1 + n_dense)]
    try:
        stack.say("cache server")
        stack.spawn("cache", [sys.executable, "-m", "vllm_production_stack_amd.cache_server", "127.0.0.1", str(cp)])
        router = stack.spawn("router", [
            sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1", "--port", str(rp),
            "--service-discovery", "static",
            "--static-backends", ",".join(f"http://127.0.0.1:{p}" for _, _, p in engines),
            "--static-models", ",".join(m for _, m, _ in engines),
            "--routing-logic", "kvaware", "--lmcache-controller-port", str(ctl), "--session-key", "x-user-id",
            "--kv-aware-threshold", "256", "--kv-aware-fallback", "load", "--engine-stats-interval", "2", "--log-level", "info"])
        _wait_http(f"http://127.0.0.1:{rp}/health", 60, router)
        for i, (name, model, port) in enumerate(engines):
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LMCACHE_REMOTE_URL=f"lm://127.0.0.1:{cp}",
                       LMCACHE_REMOTE_SERDE="naive", LMCACHE_ENABLE_CONTROLLER="True", LMCACHE_LMCACHE_INSTANCE_ID=name,
                       LMCACHE_CONTROLLER_URL=f"127.0.0.1:{ctl}", LMCACHE_LMCACHE_WORKER_PORT=str(a.base_port + 50 + i))
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_PORT"):
                env.pop(k, None)
            kv = ["--kv-transfer-config", json.dumps({"kv_connector": "LMCacheConnectorV1", "kv_role": "kv_both"})]
            if model == moe:
                env["HIP_VISIBLE_DEVICES"] = ",".join(gpus[n_dense:n_dense + tp]) if not single else "0"
                if single or a.device != "cuda":
                    env["VPSA_TP_BACKEND"] = "gloo"  # several ranks on one card: RCCL would refuse
                serve = _engine_cmd(a, model, port, a.num_gpu_blocks, kv + ["--tensor-parallel-size", str(tp)])
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={tp}",
                       "--master-addr", "127.0.0.1", "--master-port", str(a.base_port + 45), "-m",
                       "vllm_production_stack_amd.engine.cli"] + serve[serve.index("serve"):]
                stack.say(f"engine {name} ({model}, TP{tp}) on GPUs {env['HIP_VISIBLE_DEVICES']} port {port}")
                p = stack.spawn(name, cmd, env)
            else:
                env["HIP_VISIBLE_DEVICES"] = gpus[i]
                stack.say(f"engine {name} ({model}) on GPU {gpus[i]} port {port}")
                p = stack.spawn(name, _engine_cmd(a, model, port, a.num_gpu_blocks, kv), env)
            _wait_http(f"http://127.0.0.1:{port}/health", 1200, p)
        time.sleep(3)  # a stats scrape and the controller registrations
        n_users = a.users_per_gpu * n_dense
        users = list(range(n_users))
        wusers = list(range(10_000_000, 10_000_000 + max(1, a.warmup_users_per_gpu) * n_dense))
        m_users = list(range(5_000_000, 5_000_000 + max(1, a.users_per_gpu // 2) * tp))
        m_wusers = [15_000_000]
        groups = [(dense, users[0::2], wusers[0::2]), (dense, users[1::2], wusers[1::2]), (moe, m_users, m_wusers)]
        stack.say(f"{n_users} {dense} users + {len(m_users)} {moe} users x {a.steps} rounds through the router")
        rows_by_model, wall = _run_clients(stack, a, f"http://127.0.0.1:{rp}", groups)
        per_engine = []
        for name, model, port in engines:
            st = _http(f"http://127.0.0.1:{port}/debug/stats")
            per_engine.append({"replica": name, "model": model, "rows": st["prof"]["rows"], "steps": st["prof"]["steps"],
                               "busy_s": round(st["prof"]["t_execute"], 2), "prefix_hit_tokens": st.get("prefix_hits"),
                               "kv_pages_loaded_from": (st.get("kv_tiers") or {}).get("pages_loaded_from")})
        # affinity: follow-up turns of a conversation went where its KV is
        stack.logs["router"].flush()
        by_user: dict[str, set] = {}
        with open(stack.log_path("router")) as f:
            for ln in f:
                m = ROUTE_RE.search(ln)
                if m and m.group(2) != "None":
                    by_user.setdefault(m.group(2), set()).add(m.group(3))
        timed_users = {str(u) for u in users + m_users}
        sticky = sum(1 for u, eps in by_user.items() if u in timed_users and len(eps) == 1)
        sd, sm = _summary(rows_by_model[dense], wall), _summary(rows_by_model[moe], wall)
        total = round(sd["output_tok_per_s"] + sm["output_tok_per_s"], 2)
        res = _result_head(a, f"output tok/s (+ p50 TTFT), multi-round QA, {n_dense} x {dense} + {moe} TP{tp}, kvaware routing, shared KV cache",
                           total, n_dense + tp, single, f"dp{n_dense} + tp{tp} (two models behind one router)",
                           n_users + len(m_users))
        res.update({"elapsed_s": round(wall, 3), dense: sd, moe: sm, "per_replica": per_engine,
                    "conversations_kept_on_one_replica": sticky, "conversations": len(timed_users),
                    "cache_server": _http(f"http://127.0.0.1:{engines[0][2]}/debug/stats").get("kv_tiers", {}).get("remote")})
        return res
    except Exception:
        for name in list(stack.logs):
            print(f"==== {name}\n" + "\n".join(stack.tail(name, 25)), file=sys.stderr)
        raise
    finally:
        stack.stop()


def main(a) -> None:
    """Entry from bench.py: ``a.layout`` is ``pd:PxD`` or ``multimodel``."""
    if a.layout.startswith("pd"):
        spec = a.layout.split(":", 1)[1] if ":" in a.layout else "2x6"
        p, d = (int(x) for x in spec.lower().split("x"))
        res = run_pd(a, p, d)
    elif a.layout == "multimodel":
        res = run_multimodel(a)
    else:
        raise SystemExit(f"unknown --layout {a.layout!r} (pd:PxD, multimodel)")
    print(json.dumps(res), flush=True)
