"""PII detection (alpha, behind the ``PIIDetection`` feature gate): request text is scanned
and a request containing personal data is rejected with HTTP 400.

Analyzers: ``regex`` (e-mail, phone, US SSN, credit card with Luhn check, IPv4) and
``presidio`` when that package is importable.  Reference:
/root/reference/src/vllm_router/experimental/pii/ (middleware.py:43-154,
analyzers/regex.py:13-92, analyzers/presidio.py:32-160, types.py:6-52).  Unlike the
reference, where the middleware is never attached to a route, the check here runs in the
request path whenever the gate is on."""

from __future__ import annotations

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


class RegexAnalyzer:
    PATTERNS = {
        PIIType.EMAIL: re.compile(r"\b[A-Za-z0-9._%+-]+@[A-Za-z0-9.-]+\.[A-Za-z]{2,}\b"),
        PIIType.PHONE: re.compile(r"(?<!\d)(?:\+?\d{1,3}[\s.-]?)?(?:\(\d{3}\)|\d{3})[\s.-]\d{3}[\s.-]\d{4}(?!\d)"),
        PIIType.SSN: re.compile(r"(?<!\d)\d{3}-\d{2}-\d{4}(?!\d)"),
        PIIType.CREDIT_CARD: re.compile(r"(?<!\d)(?:\d[ -]?){13,19}(?!\d)"),
        PIIType.IP_ADDRESS: re.compile(r"(?<!\d)(?:(?:25[0-5]|2[0-4]\d|1?\d?\d)\.){3}(?:25[0-5]|2[0-4]\d|1?\d?\d)(?!\d)"),
    }

    def analyze(self, text: str, types: set[PIIType] | None = None) -> PIIResult:
        out = []
        for t, pat in self.PATTERNS.items():
            if types and t not in types:
                continue
            for m in pat.finditer(text):
                if t is PIIType.CREDIT_CARD:
                    digits = re.sub(r"\D", "", m.group())
                    if not (13 <= len(digits) <= 19 and _luhn(digits)):
                        continue
                out.append(PIIMatch(t, m.start(), m.end()))
        return PIIResult(bool(out), out)


class PresidioAnalyzer:
    _MAP = {"EMAIL_ADDRESS": PIIType.EMAIL, "PHONE_NUMBER": PIIType.PHONE, "US_SSN": PIIType.SSN,
            "CREDIT_CARD": PIIType.CREDIT_CARD, "IP_ADDRESS": PIIType.IP_ADDRESS,
            "PERSON": PIIType.PERSON, "LOCATION": PIIType.LOCATION}

    def __init__(self, score_threshold: float = 0.5):
        from presidio_analyzer import AnalyzerEngine  # type: ignore

        self.engine = AnalyzerEngine()
        self.threshold = score_threshold

    def analyze(self, text: str, types: set[PIIType] | None = None) -> PIIResult:
        out = []
        for r in self.engine.analyze(text=text, language="en"):
            t = self._MAP.get(r.entity_type)
            if t and r.score >= self.threshold and (not types or t in types):
                out.append(PIIMatch(t, r.start, r.end, r.score))
        return PIIResult(bool(out), out)


def create_analyzer(kind: str = "regex", **kw):
    if kind == "presidio":
        try:
            return PresidioAnalyzer(**kw)
        except ImportError:
            logger.warning("presidio is not installed; falling back to the regex analyzer")
    return RegexAnalyzer()


@dataclass
class PIIConfig:
    enabled: bool = False
    analyzer: str = "regex"
    action: PIIAction = PIIAction.BLOCK
    types: set[PIIType] | None = None


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
        self.analyzer = create_analyzer(config.analyzer)

    def check(self, request_json: dict) -> PIIResult | None:
        """-> result when the request must be blocked, else None."""
        if not self.config.enabled:
            return None
        pii_requests.inc()
        try:
            with pii_latency.time():
                res = self.analyzer.analyze(extract_text(request_json), self.config.types)
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
