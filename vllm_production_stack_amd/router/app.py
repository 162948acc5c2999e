"""Router application: state wiring, HTTP routes, background tasks, ``main``.

Endpoint table and response shapes follow /root/reference/src/vllm_router/routers/
(main_router.py:50-246, files_router.py:23-81, batches_router.py:23-113,
metrics_router.py:57-123) and the bootstrap order of /root/reference/src/vllm_router/app.py:83-300.
The server is aiohttp (C-accelerated HTTP parser, native multipart), and every background
activity (discovery watchers, scraping, stats logging, config reload) is an asyncio task
on the serving loop instead of a thread.
"""

from __future__ import annotations

import argparse
import asyncio
import json
import os
import time

import aiohttp
from aiohttp import web

from .. import __version__, tracing
from . import metrics as router_metrics
from .batches import initialize_batch_processor
from .callbacks import load_callbacks
from .config import DynamicRouterConfig, parse_args
from .discovery import create_service_discovery
from .feature_gates import get_feature_gates, initialize_feature_gates
from .files import DEFAULT_USER, initialize_storage
from .log import init_logger, set_level
from .protocols import model_card, model_list
from .proxy import (route_general_request, route_general_transcriptions,
                    route_sleep_wakeup_request)
from .rewriter import get_request_rewriter
from .routing import initialize_routing_logic
from .stats import EngineStatsScraper, RequestStatsMonitor, format_stats
from .utils import (parse_comma_separated_args, parse_static_aliases, parse_static_urls,
                    set_ulimit)

logger = init_logger(__name__)


class RouterState:
    """Everything a request handler needs; replaces the reference's process-wide singletons."""

    def __init__(self, args: argparse.Namespace):
        self.args = args
        self.client: aiohttp.ClientSession | None = None
        self.service_discovery = None
        self.engine_stats_scraper: EngineStatsScraper | None = None
        self.request_stats_monitor: RequestStatsMonitor | None = None
        self.router = None
        self.callbacks = None
        self.rewriter = None
        self.batch_storage = None
        self.batch_processor = None
        self.dynamic_config: DynamicRouterConfig | None = None
        self.semantic_cache_store = None
        self.semantic_cache_check = None
        self.pii_guard = None
        self.shared_stats = None  # under --router-workers: this worker's client of the supervisor's socket
        self.tasks: list[asyncio.Task] = []
        self.start_time = time.time()
        # request spans go to the collector named by the OTEL_* environment (off without it)
        self.tracer = tracing.Tracer.from_env("vpsa-router")


STATE_KEY = web.AppKey("state", RouterState)


# ------------------------------------------------------------------ construction helpers
def build_service_discovery(cfg, args):
    api_key = os.environ.get("VLLM_API_KEY") or None
    pl = parse_comma_separated_args(cfg.prefill_model_labels)
    dl = parse_comma_separated_args(cfg.decode_model_labels)
    if cfg.service_discovery == "static":
        return create_service_discovery(
            "static", urls=parse_static_urls(cfg.static_backends),
            models=parse_comma_separated_args(cfg.static_models),
            aliases=parse_static_aliases(cfg.static_aliases) if cfg.static_aliases else None,
            model_labels=parse_comma_separated_args(cfg.static_model_labels),
            model_types=parse_comma_separated_args(cfg.static_model_types),
            static_backend_health_checks=bool(cfg.static_backend_health_checks),
            prefill_model_labels=pl, decode_model_labels=dl,
            health_check_timeout=args.backend_health_check_timeout_seconds, api_key=api_key)
    if cfg.service_discovery == "k8s":
        kind = getattr(cfg, "k8s_service_discovery_type", None) or args.k8s_service_discovery_type
        return create_service_discovery(
            "k8s-service-name" if kind == "service-name" else "k8s-pod-ip",
            namespace=cfg.k8s_namespace or args.k8s_namespace, port=cfg.k8s_port or args.k8s_port,
            label_selector=cfg.k8s_label_selector if cfg.k8s_label_selector is not None
            else args.k8s_label_selector,
            prefill_model_labels=pl, decode_model_labels=dl,
            watcher_timeout_seconds=args.k8s_watcher_timeout_seconds,
            health_check_timeout=args.backend_health_check_timeout_seconds, api_key=api_key)
    raise ValueError(f"Invalid service discovery type: {cfg.service_discovery}")


def build_routing_logic(cfg, args):
    return initialize_routing_logic(
        cfg.routing_logic, session_key=cfg.session_key,
        lmcache_controller_port=args.lmcache_controller_port,
        prefill_model_labels=parse_comma_separated_args(cfg.prefill_model_labels),
        decode_model_labels=parse_comma_separated_args(cfg.decode_model_labels),
        kv_aware_threshold=cfg.kv_aware_threshold or args.kv_aware_threshold,
        kv_aware_fallback=getattr(args, "kv_aware_fallback", "session"),
        prefix_trie_max_nodes=args.prefix_trie_max_nodes)


async def apply_dynamic_config(state: RouterState, cfg: DynamicRouterConfig) -> None:
    """(Re)build discovery, routing logic and callbacks from ``cfg``."""
    old_sd, old_router = state.service_discovery, state.router
    sd = build_service_discovery(cfg, state.args)
    await sd.start()
    state.service_discovery = sd
    router = build_routing_logic(cfg, state.args)
    if hasattr(router, "start"):
        if old_router is not None:
            await old_router.close()
            old_router = None
        await router.start()
    state.router = router
    state.callbacks = load_callbacks(cfg.callbacks) if cfg.callbacks else None
    state.dynamic_config = cfg
    if old_sd is not None:
        await old_sd.close()
    if old_router is not None:
        await old_router.close()


async def _dynamic_config_watcher(state: RouterState, path: str, interval: float = 10.0) -> None:
    while True:
        await asyncio.sleep(interval)
        try:
            cfg = DynamicRouterConfig.from_file(path)
            base = DynamicRouterConfig.from_args(state.args).to_dict()
            merged = DynamicRouterConfig(**{**base, **{k: v for k, v in cfg.to_dict().items()
                                                       if v is not None}})
            if merged != state.dynamic_config:
                logger.info("DynamicConfigWatcher: Config changed, reconfiguring...")
                await apply_dynamic_config(state, merged)
                logger.info("DynamicConfigWatcher: Config reconfiguration complete")
        except asyncio.CancelledError:
            raise
        except Exception as e:
            logger.warning("DynamicConfigWatcher: Error loading config file: %s", e)


async def _log_stats(state: RouterState, interval: float) -> None:
    while True:
        await asyncio.sleep(interval)
        try:
            eps = state.service_discovery.get_endpoint_info()
            logger.info(format_stats(eps, state.engine_stats_scraper.get_engine_stats(),
                                     state.request_stats_monitor.get_request_stats(time.time())))
            router_metrics.refresh(state)
        except asyncio.CancelledError:
            raise
        except Exception as e:
            logger.error("log_stats failed: %s", e)


# ------------------------------------------------------------------ handlers
def _state(request: web.Request) -> RouterState:
    return request.app[STATE_KEY]


def _proxy(endpoint: str):
    async def handler(request: web.Request):
        state = _state(request)
        if state.pii_guard is not None or (endpoint == "/v1/chat/completions"
                                           and state.semantic_cache_check is not None):
            try:
                body = json.loads(await request.read())
            except Exception:
                body = None
            if isinstance(body, dict):
                if state.pii_guard is not None:
                    res = state.pii_guard.check(body)
                    if res is not None:
                        return web.json_response(
                            {"error": {"message": "Request blocked: personal data detected "
                                                  f"({', '.join(sorted(t.value for t in res.types))})",
                                       "type": "pii_detected", "code": 400}}, status=400)
                if endpoint == "/v1/chat/completions" and state.semantic_cache_check is not None:
                    hit = await state.semantic_cache_check(body)
                    if hit is not None:
                        return web.json_response(hit)
        return await route_general_request(state, request, endpoint)

    return handler


async def h_sleep(request):
    return await route_sleep_wakeup_request(_state(request), request, "/sleep")


async def h_wake(request):
    return await route_sleep_wakeup_request(_state(request), request, "/wake_up")


async def h_is_sleeping(request):
    return await route_sleep_wakeup_request(_state(request), request, "/is_sleeping")


async def h_transcriptions(request):
    return await route_general_transcriptions(_state(request), request, "/v1/audio/transcriptions")


async def h_version(request):
    return web.json_response({"version": __version__})


async def h_models(request):
    state = _state(request)
    seen: set[str] = set()
    cards = []
    for ep in state.service_discovery.get_endpoint_info():
        for mid in ep.model_names:
            if mid in seen:
                continue
            seen.add(mid)
            info = ep.model_info.get(mid)
            cards.append(model_card(mid, info.created, info.owned_by, info.root, info.parent) if info
                         else model_card(mid, ep.added_timestamp))
    aliases = getattr(state.service_discovery, "aliases", None) or {}
    for alias, target in aliases.items():
        if alias not in seen and target in seen:
            cards.append(model_card(alias, state.start_time, root=target))
    return web.json_response(model_list(cards))


async def h_engines(request):
    state = _state(request)
    seen, out = set(), []
    for ep in state.service_discovery.get_endpoint_info():
        if ep.Id in seen:
            continue
        seen.add(ep.Id)
        out.append({"engine_id": ep.Id, "serving_models": ep.model_names,
                    "created": ep.added_timestamp})
    return web.json_response(out)


async def h_health(request):
    state = _state(request)
    if not state.service_discovery.get_health():
        return web.json_response({"status": "Service discovery module is down."}, status=503)
    if not state.engine_stats_scraper.get_health():
        return web.json_response({"status": "Engine stats scraper is down."}, status=503)
    if state.args.dynamic_config_yaml or state.args.dynamic_config_json:
        return web.json_response({"status": "healthy",
                                  "dynamic_config": state.dynamic_config.to_dict()})
    return web.json_response({"status": "healthy"})


async def h_metrics(request):
    state = _state(request)
    if state.shared_stats is not None:  # one of several workers answers for the whole router
        from . import workers

        body = await router_metrics.render_all_workers(state, state.shared_stats, workers.worker_index())
    else:
        body = router_metrics.render(state)
    return web.Response(body=body, headers={"Content-Type": "text/plain; version=0.0.4; charset=utf-8"})


async def _publish_stats(state: RouterState, interval: float = 1.0) -> None:
    """One of several workers: keep this worker's metric snapshot at the supervisor fresh, so that a
    scrape answered by another worker counts this one's traffic."""
    from . import workers

    while True:
        try:
            await state.shared_stats.put_stats(workers.worker_index(), router_metrics.snapshot(state))
        except asyncio.CancelledError:
            raise
        except Exception as e:
            logger.debug("publishing the metric snapshot failed: %s", e)
        await asyncio.sleep(interval)


# ---- files ----------------------------------------------------------------------------
def _need_files(state):
    if state.batch_storage is None:
        return web.json_response({"error": "Files API is disabled (start with --enable-batch-api)"},
                                 status=501)
    return None


async def h_file_upload(request):
    state = _state(request)
    if (r := _need_files(state)) is not None:
        return r
    try:
        reader = await request.multipart()
    except Exception:
        return web.json_response({"error": "Invalid request: expected multipart/form-data"}, status=400)
    purpose, content, filename = "unknown", None, None
    async for part in reader:
        if part.name == "file":
            content = await part.read(decode=False)
            filename = part.filename
        elif part.name == "purpose":
            purpose = (await part.read(decode=False)).decode()
    if content is None:
        return web.json_response({"error": "Missing required parameter 'file'"}, status=400)
    info = await state.batch_storage.save_file(file_name=filename, content=content, purpose=purpose)
    return web.json_response(info.metadata())


async def h_file_get(request):
    state = _state(request)
    if (r := _need_files(state)) is not None:
        return r
    try:
        info = await state.batch_storage.get_file(request.match_info["file_id"])
    except FileNotFoundError:
        return web.json_response({"error": f"File {request.match_info['file_id']} not found"}, status=404)
    return web.json_response(info.metadata())


async def h_file_content(request):
    state = _state(request)
    if (r := _need_files(state)) is not None:
        return r
    try:
        data = await state.batch_storage.get_file_content(request.match_info["file_id"])
    except FileNotFoundError:
        return web.json_response({"error": f"File {request.match_info['file_id']} not found"}, status=404)
    return web.Response(body=data, content_type="application/octet-stream")


async def h_file_list(request):
    state = _state(request)
    if (r := _need_files(state)) is not None:
        return r
    ids = await state.batch_storage.list_files(DEFAULT_USER)
    data = [(await state.batch_storage.get_file(i)).metadata() for i in ids]
    return web.json_response({"object": "list", "data": data})


async def h_file_delete(request):
    state = _state(request)
    if (r := _need_files(state)) is not None:
        return r
    fid = request.match_info["file_id"]
    try:
        await state.batch_storage.delete_file(fid)
    except FileNotFoundError:
        return web.json_response({"error": f"File {fid} not found"}, status=404)
    return web.json_response({"id": fid, "object": "file", "deleted": True})


# ---- batches --------------------------------------------------------------------------
def _need_batches(state):
    if state.batch_processor is None:
        return web.json_response({"error": "Batch API is disabled (start with --enable-batch-api)"},
                                 status=501)
    return None


async def h_batch_create(request):
    state = _state(request)
    if (r := _need_batches(state)) is not None:
        return r
    try:
        body = await request.json()
    except Exception:
        return web.json_response({"error": "Invalid JSON body"}, status=400)
    fid, ep = body.get("input_file_id"), body.get("endpoint")
    if not fid:
        return web.json_response({"error": "Missing required parameter 'input_file_id'"}, status=400)
    if not ep:
        return web.json_response({"error": "Missing required parameter 'endpoint'"}, status=400)
    try:
        info = await state.batch_processor.create_batch(fid, ep, body.get("completion_window", "5s"),
                                                        body.get("metadata"))
    except FileNotFoundError:
        return web.json_response({"error": f"File {fid} not found"}, status=404)
    except ValueError as e:
        return web.json_response({"error": str(e)}, status=400)
    return web.json_response(info.to_dict())


async def h_batch_get(request):
    state = _state(request)
    if (r := _need_batches(state)) is not None:
        return r
    try:
        info = await state.batch_processor.retrieve_batch(request.match_info["batch_id"])
    except FileNotFoundError:
        return web.json_response({"error": f"Batch {request.match_info['batch_id']} not found"}, status=404)
    return web.json_response(info.to_dict())


async def h_batch_list(request):
    state = _state(request)
    if (r := _need_batches(state)) is not None:
        return r
    limit = int(request.query.get("limit", "20"))
    infos = await state.batch_processor.list_batches(limit + 1, request.query.get("after"))
    data = [i.to_dict() for i in infos[:limit]]
    return web.json_response({"object": "list", "data": data,
                              "first_id": data[0]["id"] if data else None,
                              "last_id": data[-1]["id"] if data else None,
                              "has_more": len(infos) > limit})


async def h_batch_cancel(request):
    state = _state(request)
    if (r := _need_batches(state)) is not None:
        return r
    try:
        info = await state.batch_processor.cancel_batch(request.match_info["batch_id"])
    except FileNotFoundError:
        return web.json_response({"error": f"Batch {request.match_info['batch_id']} not found"}, status=404)
    return web.json_response(info.to_dict())


def make_batch_dispatch(state: RouterState):
    """One batch line -> one routed engine request (same filtering + routing as live traffic)."""

    class _Req:
        headers: dict = {}

    async def dispatch(url: str, body: dict):
        model = body.get("model")
        sd = state.service_discovery
        aliases = getattr(sd, "aliases", None) or {}
        if model in aliases:
            model = body["model"] = aliases[model]
        eps = [e for e in sd.get_endpoint_info() if model in e.model_names and not e.sleep]
        if not eps:
            return 400, {"error": f"Model {model} not found or vLLM engine is sleeping."}
        r = state.router.route_request(eps, state.engine_stats_scraper.get_engine_stats(),
                                       state.request_stats_monitor.get_request_stats(time.time()),
                                       _Req(), body)
        if asyncio.iscoroutine(r):
            r = await r
        headers = {}
        key = os.environ.get("VLLM_API_KEY")
        if key:
            headers["Authorization"] = f"Bearer {key}"
        body = {**body, "stream": False}
        async with state.client.post(r + url, json=body, headers=headers) as resp:
            try:
                payload = await resp.json()
            except Exception:
                payload = {"raw": (await resp.read()).decode(errors="replace")}
            return resp.status, payload

    return dispatch


# ------------------------------------------------------------------ app assembly
async def initialize_all(state: RouterState) -> None:
    args = state.args
    if args.sentry_dsn:
        try:
            import sentry_sdk  # type: ignore

            sentry_sdk.init(dsn=args.sentry_dsn, send_default_pii=True, profile_lifecycle="trace",
                            traces_sample_rate=args.sentry_traces_sample_rate,
                            profile_session_sample_rate=args.sentry_profile_session_sample_rate)
        except ImportError:
            logger.warning("--sentry-dsn given but sentry_sdk is not installed; error reporting is off")
    state.client = aiohttp.ClientSession(
        connector=aiohttp.TCPConnector(limit=0, ttl_dns_cache=300, keepalive_timeout=60),
        timeout=aiohttp.ClientTimeout(total=None), auto_decompress=False)
    state.request_stats_monitor = RequestStatsMonitor(args.request_stats_window)
    state.engine_stats_scraper = EngineStatsScraper(
        args.engine_stats_interval, lambda: state.service_discovery,
        os.environ.get("VLLM_API_KEY") or None)

    def engines_gone(urls: list[str]) -> None:
        for url in urls:
            state.request_stats_monitor.forget_engine(url)
            forget = getattr(state.router, "forget_endpoint", None)
            if forget is not None:
                forget(url)
        logger.info("dropped router state of departed engines: %s", ", ".join(urls))

    state.engine_stats_scraper.on_engines_gone = engines_gone
    state.rewriter = get_request_rewriter(args.request_rewriter)
    await apply_dynamic_config(state, DynamicRouterConfig.from_args(args))
    await state.engine_stats_scraper.start()
    if args.enable_batch_api:
        state.batch_storage = initialize_storage(args.file_storage_class, args.file_storage_path)
        state.batch_processor = initialize_batch_processor(
            args.batch_processor, args.file_storage_path, state.batch_storage,
            make_batch_dispatch(state))
        from . import workers

        # the queue is a SQLite file every worker can read and write; one of them executes it
        await state.batch_processor.initialize(run_queue=workers.worker_index() == 0)
    path = args.dynamic_config_yaml or args.dynamic_config_json
    if path:
        state.tasks.append(asyncio.create_task(_dynamic_config_watcher(state, path)))
    from . import workers as _workers

    if _workers.worker_count() > 1 and _workers.shared_state_path():
        state.shared_stats = _workers.SharedTrieClient(_workers.shared_state_path())
        state.tasks.append(asyncio.create_task(_publish_stats(state)))
    if args.log_stats:
        state.tasks.append(asyncio.create_task(_log_stats(state, args.log_stats_interval)))
    gates = initialize_feature_gates(args.feature_gates)
    if gates.is_enabled("SemanticCache"):
        from . import semantic_cache as sc

        sc.initialize_semantic_cache(args.semantic_cache_model, args.semantic_cache_dir,
                                     args.semantic_cache_threshold)
        state.semantic_cache_check = sc.check_semantic_cache
        state.semantic_cache_store = sc.store_in_semantic_cache
        logger.info("Semantic cache enabled (threshold %.2f)", args.semantic_cache_threshold)
    if gates.is_enabled("PIIDetection"):
        from .pii import PIIConfig, PIIGuard

        from .pii import PIIType

        pii_cfg = json.loads(args.pii_analyzer_config) if getattr(args, "pii_analyzer_config", "") else {}
        pii_cfg.setdefault("fallback", "regex")  # a missing optional package must not stop the router
        pii_types = {PIIType(t.strip()) for t in getattr(args, "pii_types", "").split(",") if t.strip()} or None
        state.pii_guard = PIIGuard(PIIConfig(enabled=True, analyzer=args.pii_analyzer, types=pii_types,
                                             score_threshold=getattr(args, "pii_score_threshold", 0.5),
                                             analyzer_config=pii_cfg))
        logger.info("PII detection enabled (%s analyzer)", args.pii_analyzer)


async def shutdown_all(state: RouterState) -> None:
    for t in state.tasks:
        t.cancel()
    if state.batch_processor is not None:
        await state.batch_processor.close()
    if state.engine_stats_scraper is not None:
        await state.engine_stats_scraper.close()
    if state.router is not None:
        await state.router.close()
    if state.shared_stats is not None:
        await state.shared_stats.close()
    if state.service_discovery is not None:
        await state.service_discovery.close()
    if state.client is not None:
        await state.client.close()
    await asyncio.get_running_loop().run_in_executor(None, state.tracer.shutdown)


def create_app(args: argparse.Namespace) -> web.Application:
    app = web.Application(client_max_size=1024 * 1024 * 1024)
    state = RouterState(args)
    app[STATE_KEY] = state

    async def on_startup(app):
        await initialize_all(state)

    async def on_cleanup(app):
        await shutdown_all(state)

    app.on_startup.append(on_startup)
    app.on_cleanup.append(on_cleanup)
    r = app.router
    for path in ("/v1/chat/completions", "/v1/completions", "/v1/embeddings", "/tokenize",
                 "/detokenize", "/v1/rerank", "/rerank", "/v1/score", "/score"):
        r.add_post(path, _proxy(path))
    r.add_post("/sleep", h_sleep)
    r.add_post("/wake_up", h_wake)
    r.add_get("/is_sleeping", h_is_sleeping)
    r.add_post("/v1/audio/transcriptions", h_transcriptions)
    r.add_get("/version", h_version)
    r.add_get("/v1/models", h_models)
    r.add_get("/engines", h_engines)
    r.add_get("/health", h_health)
    r.add_get("/metrics", h_metrics)
    r.add_post("/v1/files", h_file_upload)
    r.add_get("/v1/files", h_file_list)
    r.add_get("/v1/files/{file_id}", h_file_get)
    r.add_delete("/v1/files/{file_id}", h_file_delete)
    r.add_get("/v1/files/{file_id}/content", h_file_content)
    r.add_post("/v1/batches", h_batch_create)
    r.add_get("/v1/batches", h_batch_list)
    r.add_get("/v1/batches/{batch_id}", h_batch_get)
    r.add_post("/v1/batches/{batch_id}/cancel", h_batch_cancel)
    return app


def main(argv: list[str] | None = None) -> None:
    args = parse_args(argv)
    set_level(args.log_level)
    set_ulimit()
    get_feature_gates()
    from . import workers

    n_workers = workers.effective_workers(args)
    if n_workers > 1:
        workers.run_workers(args, n_workers)
        return
    app = create_app(args)
    web.run_app(app, host=args.host, port=args.port, access_log=None, backlog=8192,
                print=lambda *a: logger.info("router listening on %s:%d", args.host, args.port))


if __name__ == "__main__":
    main()
