"""Router configuration: CLI flags, config-file defaults, validation, hot-reloadable subset.

Flag names and defaults follow /root/reference/src/vllm_router/parsers/parser.py:118-386;
the nested YAML form (``static_models: {name: {static_backends: [...]}}``) follows
/root/reference/src/vllm_router/parsers/yaml_utils.py:9-56; the reloadable subset follows
/root/reference/src/vllm_router/dynamic_config.py:43-117 (plus the label / health-check
fields the reference forgets to carry over from the command line).
"""

from __future__ import annotations

import argparse
import json
from dataclasses import asdict, dataclass
from typing import Any

import yaml

from .. import __version__
from .utils import ModelType, parse_comma_separated_args

ROUTING_LOGICS = ["roundrobin", "session", "kvaware", "prefixaware", "disaggregated_prefill"]


# ------------------------------------------------------------------ config files
def flatten_config(cfg: dict[str, Any] | None) -> dict[str, Any]:
    """Nested ``static_models`` / ``static_aliases`` mappings -> the flat comma-separated
    strings the flags use."""
    if not cfg:
        return {}
    cfg = dict(cfg)
    models = cfg.pop("static_models", None)
    aliases = cfg.pop("static_aliases", None)
    if isinstance(models, dict):
        backends, names, types, labels = [], [], [], []
        for name, det in models.items():
            urls = (det or {}).get("static_backends") or []
            backends += urls
            names += [name] * len(urls)
            if "static_model_type" in det:
                types += [det["static_model_type"]] * len(urls)
            if "static_model_label" in det:
                labels += [det["static_model_label"]] * len(urls)
        cfg["static_backends"] = ",".join(backends)
        cfg["static_models"] = ",".join(names)
        cfg["static_model_types"] = ",".join(types)
        if labels:
            cfg["static_model_labels"] = ",".join(labels)
    elif models is not None:
        cfg["static_models"] = models
    if isinstance(aliases, dict):
        cfg["static_aliases"] = ",".join(f"{a}:{m}" for a, m in aliases.items())
    elif aliases is not None:
        cfg["static_aliases"] = aliases
    return cfg


def read_config_file(path: str) -> dict[str, Any]:
    with open(path, encoding="utf-8") as f:
        if path.endswith(".json"):
            return flatten_config(json.load(f))
        return flatten_config(yaml.safe_load(f))


# ------------------------------------------------------------------ CLI
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="vpsa-router", description="OpenAI-compatible request router")
    a = p.add_argument
    a("--host", default="0.0.0.0")
    a("--port", type=int, default=8001)
    a("--service-discovery", choices=["static", "k8s"])
    a("--k8s-service-discovery-type", choices=["pod-ip", "service-name"], default="pod-ip")
    a("--static-backends", default=None)
    a("--static-models", default=None)
    a("--static-aliases", default=None)
    a("--static-model-types", default=None)
    a("--static-model-labels", default=None)
    a("--static-backend-health-checks", action="store_true")
    a("--k8s-port", type=int, default=8000)
    a("--k8s-namespace", default="default")
    a("--k8s-label-selector", default="")
    a("--k8s-watcher-timeout-seconds", type=int, default=0)
    a("--backend-health-check-timeout-seconds", type=int, default=10)
    a("--routing-logic", choices=ROUTING_LOGICS)
    a("--lmcache-controller-port", type=int, default=9000)
    a("--session-key", default=None)
    a("--callbacks", default=None)
    a("--request-rewriter", default="noop", choices=["noop"])
    a("--enable-batch-api", action="store_true")
    a("--file-storage-class", default="local_file", choices=["local_file"])
    a("--file-storage-path", default="/tmp/vllm_files")
    a("--batch-processor", default="local", choices=["local"])
    a("--engine-stats-interval", type=int, default=30)
    a("--request-stats-window", type=int, default=60)
    a("--log-stats", action="store_true")
    a("--log-stats-interval", type=int, default=10)
    g = p.add_mutually_exclusive_group()
    g.add_argument("--dynamic-config-yaml", default=None)
    g.add_argument("--dynamic-config-json", default=None)
    a("--version", action="version", version=f"%(prog)s {__version__}")
    a("--semantic-cache-model", default="all-MiniLM-L6-v2")
    a("--semantic-cache-dir", default="semantic_cache")
    a("--semantic-cache-threshold", type=float, default=0.95)
    a("--feature-gates", default="")
    a("--log-level", default="info",
      choices=["critical", "error", "warning", "info", "debug", "trace"])
    a("--sentry-dsn", default=None)
    a("--sentry-traces-sample-rate", type=float, default=0.1)
    a("--sentry-profile-session-sample-rate", type=float, default=1.0)
    a("--prefill-model-labels", default=None)
    a("--decode-model-labels", default=None)
    a("--kv-aware-threshold", type=int, default=2000)
    a("--kv-aware-fallback", choices=["session", "load"], default="session",
      help="kvaware routing, when no replica holds enough of the prompt: 'session' = consistent hash of the "
           "session header (the reference's rule: replicas get a binomial share of new conversations), "
           "'load' = fewest in-flight requests, equals in rotation (an even share)")
    # extensions beyond the reference flag set
    a("--max-failover-attempts", type=int, default=0,
      help="retry a request on another endpoint when the chosen one refuses the connection")
    a("--prefix-trie-max-nodes", type=int, default=4_000_000,
      help="node budget of the prefix-aware trie (0 = unbounded)")
    a("--pii-analyzer", default="regex",
      help="regex | presidio | package.module:Class (a PIIAnalyzer subclass)")
    a("--pii-analyzer-config", default="",
      help="JSON object handed to the analyzer, e.g. '{\"spacy_model\": \"en_core_web_lg\", \"fallback\": \"regex\"}'")
    a("--pii-types", default="", help="comma-separated types to look for (default: all the analyzer knows)")
    a("--pii-score-threshold", type=float, default=0.5)
    a("--router-workers", type=int, default=1,
      help="router processes sharing the listening port (SO_REUSEPORT); the prefix trie and the "
           "round-robin position are shared between them")
    return p


def validate_args(args: argparse.Namespace) -> None:
    if not args.routing_logic:
        raise ValueError("--routing-logic must be provided.")
    if not args.service_discovery:
        raise ValueError("--service-discovery must be provided.")
    if args.service_discovery == "static":
        if args.static_backends is None:
            raise ValueError("Static backends must be provided when using static service discovery.")
        if args.static_models is None:
            raise ValueError("Static models must be provided when using static service discovery.")
        if args.static_backend_health_checks:
            if args.static_model_types is None:
                raise ValueError(
                    "Static model types must be provided when using the backend healthcheck.")
            allowed = ModelType.get_all_fields()
            for t in parse_comma_separated_args(args.static_model_types):
                if t not in allowed:
                    raise ValueError(f"The model type '{t}' is not supported. "
                                     f"Supported model types are '{','.join(allowed)}'")
    if args.service_discovery == "k8s" and args.k8s_port is None:
        raise ValueError("K8s port must be provided when using K8s service discovery.")
    if args.routing_logic == "session" and args.session_key is None:
        raise ValueError("Session key must be provided when using session routing logic.")
    if getattr(args, "router_workers", 1) < 1:
        raise ValueError("--router-workers must be at least 1.")
    if args.log_stats and args.log_stats_interval <= 0:
        raise ValueError("Log stats interval must be greater than 0.")
    if args.engine_stats_interval <= 0:
        raise ValueError("Engine stats interval must be greater than 0.")
    if args.request_stats_window <= 0:
        raise ValueError("Request stats window must be greater than 0.")
    if not 0.0 <= args.sentry_traces_sample_rate <= 1.0:
        raise ValueError("Sentry traces sample rate must be between 0.0 and 1.0.")
    if not 0.0 <= args.sentry_profile_session_sample_rate <= 1.0:
        raise ValueError("Sentry profile session sample rate must be between 0.0 and 1.0.")


def parse_args(argv: list[str] | None = None) -> argparse.Namespace:
    parser = build_parser()
    args = parser.parse_args(argv)
    path = args.dynamic_config_yaml or args.dynamic_config_json
    if path:
        # file values become defaults; explicit flags still win
        parser.set_defaults(**read_config_file(path))
        args = parser.parse_args(argv)
    validate_args(args)
    return args


# ------------------------------------------------------------------ hot-reloadable subset
@dataclass
class DynamicRouterConfig:
    service_discovery: str
    routing_logic: str
    static_backends: str | None = None
    static_models: str | None = None
    static_aliases: str | None = None
    static_model_labels: str | None = None
    static_model_types: str | None = None
    static_backend_health_checks: bool | None = False
    prefill_model_labels: str | None = None
    decode_model_labels: str | None = None
    k8s_port: int | None = None
    k8s_namespace: str | None = None
    k8s_label_selector: str | None = None
    k8s_service_discovery_type: str | None = None
    session_key: str | None = None
    kv_aware_threshold: int | None = None
    callbacks: str | None = None

    @classmethod
    def _known(cls) -> set[str]:
        return set(cls.__dataclass_fields__)

    @classmethod
    def from_args(cls, args: argparse.Namespace) -> "DynamicRouterConfig":
        return cls(**{k: getattr(args, k, None) for k in cls._known()})

    @classmethod
    def from_file(cls, path: str) -> "DynamicRouterConfig":
        cfg = read_config_file(path)
        return cls(**{k: v for k, v in cfg.items() if k in cls._known()})

    def to_dict(self) -> dict:
        return asdict(self)

    def to_json_str(self) -> str:
        return json.dumps(self.to_dict(), sort_keys=True, indent=4)
