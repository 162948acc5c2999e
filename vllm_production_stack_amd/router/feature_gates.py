"""Kubernetes-style feature gates: ``--feature-gates Name=true,Other=false``.
Reference: /root/reference/src/vllm_router/experimental/feature_gates.py:12-109."""

from __future__ import annotations

import enum
import os

from .log import init_logger

logger = init_logger(__name__)


class FeatureStage(enum.Enum):
    ALPHA = "Alpha"
    BETA = "Beta"
    GA = "GA"


KNOWN_FEATURES = {
    "SemanticCache": ("Semantic cache for chat completions", FeatureStage.ALPHA, False),
    "PIIDetection": ("Block requests that contain personal data", FeatureStage.ALPHA, False),
}


class FeatureGates:
    def __init__(self, spec: str = ""):
        self.enabled: set[str] = {n for n, (_, _, d) in KNOWN_FEATURES.items() if d}
        self.configure(spec)

    def configure(self, spec: str) -> None:
        for item in (spec or "").split(","):
            item = item.strip()
            if not item:
                continue
            name, sep, val = item.partition("=")
            if not sep:
                raise ValueError(f"Invalid feature gate format: {item}. Expected format: feature=true|false")
            name, val = name.strip(), val.strip().lower()
            if name not in KNOWN_FEATURES:
                raise ValueError(f"Unknown feature gate: {name}")
            if val not in ("true", "false"):
                raise ValueError(f"Invalid feature gate value for {name}: {val}")
            (self.enabled.add if val == "true" else self.enabled.discard)(name)
            logger.info("Feature gate %s (%s) = %s", name, KNOWN_FEATURES[name][1].value, val)

    def is_enabled(self, name: str) -> bool:
        return name in self.enabled


_gates: FeatureGates | None = None


def initialize_feature_gates(spec: str = "") -> FeatureGates:
    global _gates
    env = os.environ.get("VLLM_FEATURE_GATES", "")
    _gates = FeatureGates(",".join(x for x in (env, spec) if x))
    return _gates


def get_feature_gates() -> FeatureGates:
    global _gates
    if _gates is None:
        _gates = FeatureGates()
    return _gates
