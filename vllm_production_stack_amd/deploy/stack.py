"""``vpsa-stack`` - deploy a values.yaml without Kubernetes, or render it for a cluster.

  vpsa-stack template -f values.yaml [--release R --namespace N]   # manifests on stdout
  vpsa-stack plan     -f values.yaml [--gpus 8]                    # local process plan (JSON)
  vpsa-stack up       -f values.yaml [--gpus 8 --router-port 30080]

``up`` is the single-node deployment of the stack on one MI355X host: every engine replica is
its own process pinned to its GPU group through ``HIP_VISIBLE_DEVICES`` (tensor-parallel
replicas are ``torch.distributed`` rank groups over xGMI), the router runs with static
discovery over the replicas, and ``cacheserverSpec`` becomes one local cache-server process.
It is the bare-metal counterpart of ``helm install`` in the reference's quick start
(/root/reference/tutorials/01-minimal-helm-installation.md) and what ``bench.py`` builds by
hand.
"""

from __future__ import annotations

import argparse
import json
import os
import signal
import subprocess
import sys
import time
from dataclasses import asdict, dataclass, field

from . import render


@dataclass
class ProcessSpec:
    name: str
    argv: list[str]
    env: dict[str, str] = field(default_factory=dict)
    gpus: list[int] = field(default_factory=list)
    port: int = 0
    role: str = "engine"


def _plain_env(env_list: list[dict]) -> dict[str, str]:
    """Literal entries only: secret/field references have no meaning outside a cluster."""
    return {e["name"]: str(e["value"]) for e in env_list if "value" in e}


def local_plan(values: dict, gpus: int = 8, base_port: int = 8100, router_port: int = 30080,
               device: str = "auto", python: str = sys.executable) -> list[ProcessSpec]:
    ses = values["servingEngineSpec"]
    plan: list[ProcessSpec] = []
    release = "local"
    cache_port = 0
    cs = values.get("cacheserverSpec")
    if cs:
        cache_port = int(cs.get("containerPort", 8080))
        argv = [python, "-m", "vllm_production_stack_amd.cache_server", "127.0.0.1", str(cache_port)]
        if cs.get("maxBytes"):
            argv += ["--max-bytes", str(cs["maxBytes"])]
        plan.append(ProcessSpec("cache-server", argv, {}, [], cache_port, "cache-server"))
    next_gpu = 0
    port = base_port
    backends, models = [], []
    rs = values["routerSpec"]
    controller_port = int(rs.get("lmcacheControllerPort") or 0)
    for ms in (ses.get("modelSpec") or []) if ses.get("enableEngine") else []:
        n_gpu = render.gpu_count(ms)
        vc = ms.get("vllmConfig") or {}
        tp = int(vc.get("tensorParallelSize") or 1)
        for r in range(int(ms.get("replicaCount", 1))):
            group = list(range(next_gpu, next_gpu + n_gpu))
            if n_gpu and group[-1] >= gpus:
                raise render.ValuesError(
                    f"modelSpec {ms['name']!r} replica {r} needs GPUs {group} but the node has {gpus}")
            next_gpu += n_gpu
            args = render.engine_args({**values, "servingEngineSpec": {**ses, "containerPort": port}}, ms)
            args[args.index("--host") + 1] = "127.0.0.1"
            if device != "auto":
                args += ["--device", device]
            env = _plain_env(render.engine_env(values, ms, release))
            env.pop("HF_HOME", None)
            if n_gpu:
                env["HIP_VISIBLE_DEVICES"] = ",".join(map(str, group))
            if "LMCACHE_REMOTE_URL" in env:
                env["LMCACHE_REMOTE_URL"] = f"lm://127.0.0.1:{cache_port}"
            if "LMCACHE_CONTROLLER_URL" in env:
                env["LMCACHE_CONTROLLER_URL"] = f"127.0.0.1:{controller_port or 9000}"
            env.setdefault("LMCACHE_LMCACHE_INSTANCE_ID", f"{ms['name']}-{r}")
            env["POD_IP"] = "127.0.0.1"
            if tp > 1:
                argv = [python, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={tp}",
                        "--master-addr", "127.0.0.1", "--master-port", str(29500 + len(plan)),
                        "-m", "vllm_production_stack_amd.engine.cli", *args]
            else:
                argv = [python, "-m", "vllm_production_stack_amd.engine.cli", *args]
            plan.append(ProcessSpec(f"{ms['name']}-{r}", argv, env, group, port, "engine"))
            backends.append(f"http://127.0.0.1:{port}")
            models.append(str(ms.get("servedModelName") or ms["modelURL"]))
            port += 1
    if rs.get("enableRouter"):
        rvals = render.deep_merge(values, {"routerSpec": {
            "serviceDiscovery": "static", "containerPort": router_port,
            "staticBackends": ",".join(backends) or rs.get("staticBackends", ""),
            "staticModels": ",".join(models) or rs.get("staticModels", "")}})
        args = render.router_args(rvals, release, "default")
        args[args.index("--host") + 1] = "127.0.0.1"
        plan.append(ProcessSpec("router", [python, "-m", "vllm_production_stack_amd.router.app", *args],
                                {}, [], router_port, "router"))
    return plan


def _wait_http(url: str, timeout: float) -> bool:
    import urllib.request

    end = time.time() + timeout
    while time.time() < end:
        try:
            with urllib.request.urlopen(url, timeout=2) as r:
                if r.status == 200:
                    return True
        except Exception:
            time.sleep(0.5)
    return False


class LocalStack:
    """Owns the child processes of one ``up``."""

    def __init__(self, plan: list[ProcessSpec], log_dir: str | None = None):
        self.plan = plan
        self.log_dir = log_dir
        self.procs: list[tuple[ProcessSpec, subprocess.Popen]] = []

    def start(self, ready_timeout: float = 600.0) -> None:
        """Cache servers and engines start together (engines of one GPU set size their pools from
        free memory, so they come up one after another there); the router starts once every
        engine answers /health, so its first scrape finds all of them."""
        def launch(spec: ProcessSpec) -> None:
            env = {**os.environ, **spec.env}
            out = None
            if self.log_dir:
                os.makedirs(self.log_dir, exist_ok=True)
                out = open(os.path.join(self.log_dir, f"{spec.name}.log"), "w")
            p = subprocess.Popen(spec.argv, env=env, stdout=out, stderr=subprocess.STDOUT if out else None,
                                 start_new_session=True)
            self.procs.append((spec, p))

        def healthy(spec: ProcessSpec) -> None:
            if not _wait_http(f"http://127.0.0.1:{spec.port}/health", ready_timeout):
                self.stop()
                raise RuntimeError(f"{spec.name} did not become healthy on port {spec.port}")

        for role in ("cache-server", "engine"):
            for spec in self.plan:
                if spec.role == role:
                    launch(spec)
        for spec in self.plan:
            if spec.role == "engine":
                healthy(spec)
        for spec in self.plan:
            if spec.role == "router":
                launch(spec)
                healthy(spec)

    def stop(self) -> None:
        for _spec, p in reversed(self.procs):
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        end = time.time() + 10
        for _spec, p in self.procs:
            try:
                p.wait(max(0.1, end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
        self.procs.clear()

    def wait(self) -> int:
        try:
            while True:
                for spec, p in self.procs:
                    rc = p.poll()
                    if rc is not None:
                        print(f"[vpsa-stack] {spec.name} exited with {rc}; stopping the stack", file=sys.stderr)
                        self.stop()
                        return rc or 1
                time.sleep(1.0)
        except KeyboardInterrupt:
            self.stop()
            return 0


def main(argv: list[str] | None = None) -> int:
    ap = argparse.ArgumentParser(prog="vpsa-stack")
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("template", "plan", "up"):
        s = sub.add_parser(name)
        s.add_argument("-f", "--values", action="append", default=[])
        s.add_argument("--release", default="vllm")
        s.add_argument("--namespace", default="default")
        s.add_argument("--gpus", type=int, default=8)
        s.add_argument("--base-port", type=int, default=8100)
        s.add_argument("--router-port", type=int, default=30080)
        s.add_argument("--device", default="auto")
        s.add_argument("--log-dir", default=None)
    a = ap.parse_args(argv)
    values = render.load_values(*a.values)
    if a.cmd == "template":
        sys.stdout.write(render.to_yaml(render.render_all(values, a.release, a.namespace)))
        return 0
    plan = local_plan(values, a.gpus, a.base_port, a.router_port, a.device)
    if a.cmd == "plan":
        print(json.dumps([asdict(p) for p in plan], indent=2))
        return 0
    stack = LocalStack(plan, a.log_dir)
    stack.start()
    print(f"[vpsa-stack] up: router on http://127.0.0.1:{a.router_port} over "
          f"{sum(1 for p in plan if p.role == 'engine')} engine replica(s)", flush=True)
    return stack.wait()


if __name__ == "__main__":
    sys.exit(main())
