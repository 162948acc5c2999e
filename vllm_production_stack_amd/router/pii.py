"""PII detection (alpha, behind the ``PIIDetection`` feature gate): request text is scanned
and a request containing personal data is rejected with HTTP 400.

Analyzers are plug-ins behind one interface (``PIIAnalyzer``) and one registry (``ANALYZERS``,
``register_analyzer``, ``create_analyzer``): ``regex`` (e-mail, phone, US SSN, credit card with
Luhn check, IPv4, provider-shaped API keys), ``presidio`` (names and places too; needs the
package and a spaCy model), or any class named as ``package.module:Class``.  Reference:
/root/reference/src/vllm_router/experimental/pii/ (middleware.py:43-154,
analyzers/regex.py:13-92, analyzers/presidio.py:32-160, types.py:6-52).  Unlike the
reference, where the middleware is never attached to a route, the check here runs in the
request path whenever the gate is on."""

from __future__ import annotations

import abc
import enum
import re
from dataclasses import dataclass, field

from prometheus_client import Counter, Histogram

from .log import init_logger
from .metrics import REGISTRY

logger = init_logger(__name__)

pii_requests = Counter("vllm_pii_requests", "Requests scanned for PII", registry=REGISTRY)
pii_detected = Counter("vllm_pii_detected", "Requests in which PII was found", ["type"], registry=REGISTRY)
pii_blocked = Counter("vllm_pii_blocked", "Requests blocked because of PII", registry=REGISTRY)
pii_errors = Counter("vllm_pii_errors", "PII analyzer failures", registry=REGISTRY)
pii_latency = Histogram("vllm_pii_analysis_seconds", "PII analysis latency", registry=REGISTRY)


class PIIType(str, enum.Enum):
    EMAIL = "email"
    PHONE = "phone"
    SSN = "ssn"
    CREDIT_CARD = "credit_card"
    IP_ADDRESS = "ip_address"
    API_KEY = "api_key"
    PERSON = "person"
    LOCATION = "location"


class PIIAction(str, enum.Enum):
    BLOCK = "block"
    ALLOW = "allow"


@dataclass
class PIIMatch:
    type: PIIType
    start: int
    end: int
    score: float = 1.0


@dataclass
class PIIResult:
    has_pii: bool
    matches: list[PIIMatch] = field(default_factory=list)

    @property
    def types(self) -> set[PIIType]:
        return {m.type for m in self.matches}


# ------------------------------------------------------------------ analyzer plug-in surface
class PIIAnalyzer(abc.ABC):
    """What an analyzer has to be.  Life cycle: constructed with its config dictionary,
    ``initialize()`` once before the first request (load models here; raise to refuse the
    start), ``analyze()`` per request on the event loop (keep it fast or offload), ``shutdown()``
    when the router stops.  Reference interface:
    /root/reference/src/vllm_router/experimental/pii/analyzers/base.py:31-68 (async there; the
    regex pass takes microseconds and Presidio is CPU-bound either way, so it is synchronous here)."""

    def __init__(self, config: dict | None = None):
        self.config = dict(config or {})

    def initialize(self) -> None:
        """Called once before the first ``analyze``."""

    def shutdown(self) -> None:
        """Called when the router stops."""

    @abc.abstractmethod
    def analyze(self, text: str, types: set[PIIType] | None = None,
                score_threshold: float = 0.5) -> PIIResult:
        """Find personal data in ``text``.  ``types``: what to look for (None = everything the
        analyzer knows); matches scoring below ``score_threshold`` are dropped."""


def _luhn(digits: str) -> bool:
    s, alt = 0, False
    for ch in reversed(digits):
        d = int(ch)
        if alt:
            d *= 2
            if d > 9:
                d -= 9
        s += d
        alt = not alt
    return s % 10 == 0


class RegexAnalyzer(PIIAnalyzer):
    """Pattern matching, no dependencies.  Config: ``extra_patterns`` = {type name: regex} adds to
    (or replaces) the built-in patterns."""

    PATTERNS = {
        PIIType.EMAIL: re.compile(r"\b[A-Za-z0-9._%+-]+@[A-Za-z0-9.-]+\.[A-Za-z]{2,}\b"),
        PIIType.PHONE: re.compile(r"(?<!\d)(?:\+?\d{1,3}[\s.-]?)?(?:\(\d{3}\)|\d{3})[\s.-]\d{3}[\s.-]\d{4}(?!\d)"),
        PIIType.SSN: re.compile(r"(?<!\d)\d{3}-\d{2}-\d{4}(?!\d)"),
        PIIType.CREDIT_CARD: re.compile(r"(?<!\d)(?:\d[ -]?){13,19}(?!\d)"),
        PIIType.IP_ADDRESS: re.compile(r"(?<!\d)(?:(?:25[0-5]|2[0-4]\d|1?\d?\d)\.){3}(?:25[0-5]|2[0-4]\d|1?\d?\d)(?!\d)"),
        # provider-shaped secrets: OpenAI-style sk-..., AWS access key ids, GitHub and Slack tokens,
        # Hugging Face tokens, and "Bearer <long token>"
        PIIType.API_KEY: re.compile(r"\b(?:sk-[A-Za-z0-9_-]{20,}|AKIA[0-9A-Z]{16}|gh[pousr]_[A-Za-z0-9]{36,}"
                                    r"|xox[baprs]-[A-Za-z0-9-]{10,}|hf_[A-Za-z0-9]{30,})\b"
                                    r"|(?<=Bearer )[A-Za-z0-9._~+/-]{24,}=*"),
    }

    def initialize(self) -> None:
        self.patterns = dict(self.PATTERNS)
        for name, pat in (self.config.get("extra_patterns") or {}).items():
            self.patterns[PIIType(name)] = re.compile(pat)

    def analyze(self, text, types=None, score_threshold=0.5):
        out = []
        for t, pat in getattr(self, "patterns", self.PATTERNS).items():
            if types and t not in types:
                continue
            for m in pat.finditer(text):
                if t is PIIType.CREDIT_CARD:
                    digits = re.sub(r"\D", "", m.group())
                    if not (13 <= len(digits) <= 19 and _luhn(digits)):
                        continue
                out.append(PIIMatch(t, m.start(), m.end()))
        return PIIResult(bool(out), out)


class PresidioAnalyzer(PIIAnalyzer):
    """Microsoft Presidio (named-entity recognition on top of patterns: finds names and places
    too).  Needs ``presidio-analyzer`` and a spaCy model; ``initialize`` says exactly what is
    missing.  Config: ``language`` (en), ``spacy_model`` (en_core_web_sm), ``entities`` = extra
    {Presidio entity: type name} mappings.  Reference:
    /root/reference/src/vllm_router/experimental/pii/analyzers/presidio.py:32-160."""

    TO_PRESIDIO = {PIIType.EMAIL: "EMAIL_ADDRESS", PIIType.PHONE: "PHONE_NUMBER", PIIType.SSN: "US_SSN",
                   PIIType.CREDIT_CARD: "CREDIT_CARD", PIIType.IP_ADDRESS: "IP_ADDRESS",
                   PIIType.PERSON: "PERSON", PIIType.LOCATION: "LOCATION"}

    def initialize(self) -> None:
        missing = []
        try:
            from presidio_analyzer import AnalyzerEngine  # type: ignore
            from presidio_analyzer.nlp_engine import NlpEngineProvider  # type: ignore
        except ImportError:
            missing.append("pip install presidio-analyzer")
        model = self.config.get("spacy_model", "en_core_web_sm")
        try:
            import spacy  # type: ignore

            if not spacy.util.is_package(model):
                missing.append(f"python -m spacy download {model}")
        except ImportError:
            missing.append(f"pip install spacy && python -m spacy download {model}")
        if missing:
            raise RuntimeError("the presidio analyzer needs: " + "; ".join(missing))
        lang = self.config.get("language", "en")
        nlp = NlpEngineProvider(nlp_configuration={"nlp_engine_name": "spacy",
                                                   "models": [{"lang_code": lang, "model_name": model}]}).create_engine()
        self.language = lang
        self.engine = AnalyzerEngine(nlp_engine=nlp, supported_languages=[lang])
        self.from_presidio = {v: k for k, v in self.TO_PRESIDIO.items()}
        for entity, name in (self.config.get("entities") or {}).items():
            self.from_presidio[entity] = PIIType(name)

    def shutdown(self) -> None:
        self.engine = None

    def analyze(self, text, types=None, score_threshold=0.5):
        if getattr(self, "engine", None) is None:
            raise RuntimeError("analyzer not initialised")
        entities = [e for e, t in self.from_presidio.items() if not types or t in types]
        out = []
        for r in self.engine.analyze(text=text, language=self.language, entities=entities,
                                     score_threshold=score_threshold):
            t = self.from_presidio.get(r.entity_type)
            if t is not None:
                out.append(PIIMatch(t, r.start, r.end, r.score))
        return PIIResult(bool(out), out)


# name -> class.  ``register_analyzer`` adds to it; ``create_analyzer`` also accepts a dotted path
# ``package.module:ClassName`` so that a deployment can bring its own analyzer without touching
# this file.  Reference registry: /root/reference/src/vllm_router/experimental/pii/analyzers/factory.py:13-55.
ANALYZERS: dict[str, type[PIIAnalyzer]] = {"regex": RegexAnalyzer, "presidio": PresidioAnalyzer}


def register_analyzer(name: str, cls: type[PIIAnalyzer]) -> None:
    if not (isinstance(cls, type) and issubclass(cls, PIIAnalyzer)):
        raise TypeError("an analyzer must subclass PIIAnalyzer")
    ANALYZERS[name] = cls


def create_analyzer(kind: str = "regex", config: dict | None = None) -> PIIAnalyzer:
    """Construct and initialise an analyzer.  ``ValueError`` for a name nobody registered,
    ``RuntimeError`` when the analyzer cannot start (missing packages or models) unless
    ``config["fallback"]`` names another analyzer to use instead."""
    cls = ANALYZERS.get(kind)
    if cls is None and ":" in kind:
        import importlib

        mod, _, attr = kind.partition(":")
        try:
            cls = getattr(importlib.import_module(mod), attr)
        except (ImportError, AttributeError) as e:
            raise ValueError(f"cannot load PII analyzer {kind}: {e}") from e
        register_analyzer(kind, cls)
    if cls is None:
        raise ValueError(f"unknown PII analyzer {kind!r}; known: {sorted(ANALYZERS)}")
    analyzer = cls(config)
    try:
        analyzer.initialize()
    except Exception as e:
        fallback = (config or {}).get("fallback")
        if fallback and fallback != kind:
            logger.warning("PII analyzer %s could not start (%s); using %s", kind, e, fallback)
            return create_analyzer(fallback, {k: v for k, v in (config or {}).items() if k != "fallback"})
        raise RuntimeError(f"PII analyzer {kind} could not start: {e}") from e
    logger.info("PII analyzer %s ready", kind)
    return analyzer


@dataclass
class PIIConfig:
    enabled: bool = False
    analyzer: str = "regex"
    action: PIIAction = PIIAction.BLOCK
    types: set[PIIType] | None = None
    score_threshold: float = 0.5
    analyzer_config: dict = field(default_factory=dict)


def extract_text(request_json: dict) -> str:
    parts = []
    if isinstance(request_json.get("prompt"), str):
        parts.append(request_json["prompt"])
    for m in request_json.get("messages") or []:
        c = m.get("content")
        if isinstance(c, str):
            parts.append(c)
        elif isinstance(c, list):
            parts += [p.get("text", "") for p in c if isinstance(p, dict)]
    if isinstance(request_json.get("input"), str):
        parts.append(request_json["input"])
    return "\n".join(parts)


class PIIGuard:
    def __init__(self, config: PIIConfig):
        self.config = config
        self.analyzer = create_analyzer(config.analyzer, config.analyzer_config)

    def close(self) -> None:
        self.analyzer.shutdown()

    def check(self, request_json: dict) -> PIIResult | None:
        """-> result when the request must be blocked, else None."""
        if not self.config.enabled:
            return None
        pii_requests.inc()
        try:
            with pii_latency.time():
                res = self.analyzer.analyze(extract_text(request_json), self.config.types,
                                            self.config.score_threshold)
        except Exception as e:
            pii_errors.inc()
            logger.error("PII analysis failed: %s", e)
            return None
        if res.has_pii:
            for t in res.types:
                pii_detected.labels(type=t.value).inc()
            if self.config.action is PIIAction.BLOCK:
                pii_blocked.inc()
                return res
        return None
