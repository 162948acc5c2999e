"""Request tracing: W3C trace-context propagation and an OTLP span exporter with no SDK.

The reference turns tracing on for its engines purely through the standard OpenTelemetry
environment (``OTEL_SERVICE_NAME``, ``OTEL_EXPORTER_OTLP_ENDPOINT=http://otel-collector:4317``,
``OTEL_RESOURCE_ATTRIBUTES`` -- /root/reference/tutorials/assets/values-12-otel-vllm.yaml:15-22,
/root/reference/docs/source/use_cases/distributed-tracing.rst:49-116) and leaves the span
production to the engine image.  Here the engine and the router produce the spans themselves:

* ``traceparent`` (W3C trace context) is read from the incoming request, a child span is
  opened, and the header is rewritten on the way to the next hop, so client -> router ->
  engine (-> prefill engine for disaggregated prefill) show up as one trace;
* the engine's request span carries the ``gen_ai.*`` attributes collectors and Jaeger
  dashboards for LLM servers key on (prompt / completion tokens, time in queue, time to first
  token, end-to-end latency, sampling settings);
* finished spans are batched on a background thread and sent as OTLP
  ``ExportTraceServiceRequest`` messages.  The protobuf is encoded by hand (the message is a
  few nested length-delimited fields), so neither the OpenTelemetry SDK nor generated stubs are
  needed.  Transports: gRPC (``OTEL_EXPORTER_OTLP_PROTOCOL=grpc``, the default, port 4317) and
  OTLP/HTTP (``http/protobuf``, port 4318, ``POST /v1/traces``).

Tracing is off (every call is a cheap no-op) unless an OTLP endpoint is configured.
"""

from __future__ import annotations

import os
import queue
import random
import struct
import threading
import time
import urllib.request
from dataclasses import dataclass, field

from .router.log import init_logger

logger = init_logger(__name__)

KIND_INTERNAL, KIND_SERVER, KIND_CLIENT = 1, 2, 3
STATUS_UNSET, STATUS_OK, STATUS_ERROR = 0, 1, 2
_EXPORT_PATH = "/opentelemetry.proto.collector.trace.v1.TraceService/Export"


# ------------------------------------------------------------------ W3C trace context
def parse_traceparent(value: str | None) -> tuple[bytes, bytes, int] | None:
    """``00-<32 hex trace id>-<16 hex parent id>-<2 hex flags>`` -> (trace_id, span_id, flags)."""
    if not value:
        return None
    parts = value.strip().split("-")
    if len(parts) < 4 or len(parts[0]) != 2 or len(parts[1]) != 32 or len(parts[2]) != 16 or len(parts[3]) != 2:
        return None
    try:
        version = int(parts[0], 16)
        trace_id, span_id, flags = bytes.fromhex(parts[1]), bytes.fromhex(parts[2]), int(parts[3], 16)
    except ValueError:
        return None
    if version == 0xFF or trace_id == bytes(16) or span_id == bytes(8):
        return None
    return trace_id, span_id, flags


def format_traceparent(trace_id: bytes, span_id: bytes, flags: int = 1) -> str:
    return f"00-{trace_id.hex()}-{span_id.hex()}-{flags & 0xFF:02x}"


# ------------------------------------------------------------------ protobuf wire format
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field_no: int, payload: bytes) -> bytes:  # length-delimited
    return _varint((field_no << 3) | 2) + _varint(len(payload)) + payload


def _vi(field_no: int, n: int) -> bytes:
    return _varint(field_no << 3) + _varint(n)


def _f64(field_no: int, n: int) -> bytes:  # fixed64
    return _varint((field_no << 3) | 1) + struct.pack("<Q", n)


def _any_value(v) -> bytes:
    if isinstance(v, bool):
        return _vi(2, 1 if v else 0)
    if isinstance(v, int):
        return _vi(3, v)
    if isinstance(v, float):
        return _varint((4 << 3) | 1) + struct.pack("<d", v)
    if isinstance(v, (list, tuple)):
        return _ld(5, b"".join(_ld(1, _any_value(x)) for x in v))  # ArrayValue.values
    return _ld(1, str(v).encode())


def _key_value(k: str, v) -> bytes:
    return _ld(1, k.encode()) + _ld(2, _any_value(v))


@dataclass
class Span:
    name: str
    trace_id: bytes
    span_id: bytes
    parent_span_id: bytes = b""
    kind: int = KIND_SERVER
    start_ns: int = 0
    end_ns: int = 0
    flags: int = 1
    attributes: dict = field(default_factory=dict)
    events: list = field(default_factory=list)  # (time_ns, name, attributes)
    status: int = STATUS_UNSET
    status_message: str = ""
    _tracer: "Tracer | None" = None

    # -- recording
    def set(self, key: str, value) -> "Span":
        if value is not None:
            self.attributes[key] = value
        return self

    def event(self, name: str, **attrs) -> None:
        self.events.append((time.time_ns(), name, attrs))

    def error(self, message: str) -> None:
        self.status, self.status_message = STATUS_ERROR, message[:512]

    def end(self, end_ns: int | None = None) -> None:
        if self.end_ns:
            return
        self.end_ns = end_ns or time.time_ns()
        if self._tracer is not None:
            self._tracer._finish(self)

    @property
    def traceparent(self) -> str:
        return format_traceparent(self.trace_id, self.span_id, self.flags)

    def encode(self) -> bytes:
        out = [_ld(1, self.trace_id), _ld(2, self.span_id)]
        if self.parent_span_id:
            out.append(_ld(4, self.parent_span_id))
        out += [_ld(5, self.name.encode()), _vi(6, self.kind), _f64(7, self.start_ns), _f64(8, self.end_ns)]
        out += [_ld(9, _key_value(k, v)) for k, v in self.attributes.items()]
        for t, name, attrs in self.events:
            ev = _f64(1, t) + _ld(2, name.encode()) + b"".join(_ld(3, _key_value(k, v)) for k, v in attrs.items())
            out.append(_ld(11, ev))
        if self.status != STATUS_UNSET:
            st = (_ld(2, self.status_message.encode()) if self.status_message else b"") + _vi(3, self.status)
            out.append(_ld(15, st))
        return b"".join(out)


class _NoopSpan:
    """Stands in when tracing is off: same surface, records nothing."""

    traceparent = ""
    trace_id = span_id = b""

    def set(self, key, value):
        return self

    def event(self, name, **attrs):
        pass

    def error(self, message):
        pass

    def end(self, end_ns=None):
        pass


NOOP_SPAN = _NoopSpan()


def encode_export_request(spans: list[Span], resource: dict, scope: str, scope_version: str = "") -> bytes:
    res = b"".join(_ld(1, _key_value(k, v)) for k, v in resource.items())
    sc = _ld(1, scope.encode()) + (_ld(2, scope_version.encode()) if scope_version else b"")
    scope_spans = _ld(1, sc) + b"".join(_ld(2, s.encode()) for s in spans)
    return _ld(1, _ld(1, res) + _ld(2, scope_spans))


def _parse_kv_list(text: str) -> dict:
    out = {}
    for item in (text or "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip()] = urllib.request.unquote(v.strip())
    return out


class Tracer:
    """Span factory + batching exporter.  ``Tracer.from_env`` is the normal constructor."""

    def __init__(self, endpoint: str | None, protocol: str = "grpc", service_name: str = "vpsa",
                 resource: dict | None = None, headers: dict | None = None, insecure: bool = True,
                 flush_interval_s: float = 2.0, max_batch: int = 512, max_queue: int = 8192,
                 timeout_s: float = 10.0, sample_ratio: float = 1.0):
        self.endpoint = endpoint
        self.protocol = protocol
        self.resource = {"service.name": service_name, "telemetry.sdk.name": "vpsa-tracing",
                         "telemetry.sdk.language": "python", **(resource or {})}
        self.headers = headers or {}
        self.insecure = insecure
        self.flush_interval_s = flush_interval_s
        self.max_batch = max_batch
        self.timeout_s = timeout_s
        self.sample_ratio = sample_ratio
        self.exported = 0
        self.dropped = 0
        self.export_errors = 0
        self._q: queue.Queue[Span] = queue.Queue(maxsize=max_queue)
        self._stop = threading.Event()
        self._thread: threading.Thread | None = None
        self._channel = None
        self._rpc = None
        self._rng = random.Random(os.urandom(8))

    # -- construction
    @classmethod
    def from_env(cls, default_service: str, env=None) -> "Tracer":
        env = os.environ if env is None else env
        endpoint = env.get("OTEL_EXPORTER_OTLP_TRACES_ENDPOINT") or env.get("OTEL_EXPORTER_OTLP_ENDPOINT")
        if env.get("OTEL_SDK_DISABLED", "").lower() == "true" or env.get("OTEL_TRACES_EXPORTER", "otlp") == "none":
            endpoint = None
        protocol = (env.get("OTEL_EXPORTER_OTLP_TRACES_PROTOCOL") or env.get("OTEL_EXPORTER_OTLP_PROTOCOL")
                    or "grpc").lower()
        resource = _parse_kv_list(env.get("OTEL_RESOURCE_ATTRIBUTES", ""))
        service = env.get("OTEL_SERVICE_NAME") or resource.pop("service.name", None) or default_service
        resource.pop("service.name", None)
        headers = _parse_kv_list(env.get("OTEL_EXPORTER_OTLP_TRACES_HEADERS") or env.get("OTEL_EXPORTER_OTLP_HEADERS", ""))
        insecure_env = env.get("OTEL_EXPORTER_OTLP_TRACES_INSECURE") or env.get("OTEL_EXPORTER_OTLP_INSECURE")
        insecure = (insecure_env.lower() == "true") if insecure_env else not (endpoint or "").startswith("https://")
        ratio = 1.0
        sampler = env.get("OTEL_TRACES_SAMPLER", "parentbased_always_on")
        if sampler in ("always_off", "parentbased_always_off"):
            ratio = 0.0
        elif sampler in ("traceidratio", "parentbased_traceidratio"):
            try:
                ratio = min(1.0, max(0.0, float(env.get("OTEL_TRACES_SAMPLER_ARG", "1.0"))))
            except ValueError:
                ratio = 1.0
        try:
            interval = float(env.get("OTEL_BSP_SCHEDULE_DELAY", "2000")) / 1e3
        except ValueError:
            interval = 2.0
        if endpoint and protocol.startswith("http") and env.get("OTEL_EXPORTER_OTLP_TRACES_ENDPOINT") is None:
            endpoint = endpoint.rstrip("/") + "/v1/traces"  # the generic endpoint is a base URL
        t = cls(endpoint, protocol, service, resource, headers, insecure, flush_interval_s=interval,
                sample_ratio=ratio)
        if t.enabled:
            logger.info("tracing: exporting spans of service %s to %s (%s)", service, endpoint, protocol)
        return t

    @property
    def enabled(self) -> bool:
        return bool(self.endpoint)

    # -- spans
    def start_span(self, name: str, traceparent: str | None = None, kind: int = KIND_SERVER,
                   parent: Span | None = None, start_ns: int | None = None):
        if not self.enabled:
            return NOOP_SPAN
        if parent is not None and isinstance(parent, Span):
            trace_id, parent_id, flags = parent.trace_id, parent.span_id, parent.flags
        else:
            ctx = parse_traceparent(traceparent)
            if ctx is not None:
                trace_id, parent_id, flags = ctx
            else:
                trace_id, parent_id = self._rng.randbytes(16), b""
                flags = 1 if self._rng.random() < self.sample_ratio else 0
        return Span(name=name, trace_id=trace_id, span_id=self._rng.randbytes(8), parent_span_id=parent_id,
                    kind=kind, start_ns=start_ns or time.time_ns(), flags=flags, _tracer=self)

    def _finish(self, span: Span) -> None:
        if not (span.flags & 1):
            return  # not sampled: context still propagated, nothing exported
        try:
            self._q.put_nowait(span)
        except queue.Full:
            self.dropped += 1
            return
        if self._thread is None:
            self._start_worker()

    # -- export
    def _start_worker(self) -> None:
        self._thread = threading.Thread(target=self._run, name="otlp-export", daemon=True)
        self._thread.start()

    def _run(self) -> None:
        while not self._stop.is_set():
            self._stop.wait(self.flush_interval_s)
            self.flush()

    def flush(self) -> int:
        """Send everything queued; returns the number of spans handed to the collector."""
        sent = 0
        while True:
            batch: list[Span] = []
            try:
                while len(batch) < self.max_batch:
                    batch.append(self._q.get_nowait())
            except queue.Empty:
                pass
            if not batch:
                return sent
            try:
                self._send(encode_export_request(batch, self.resource, "vllm_production_stack_amd"))
                self.exported += len(batch)
                sent += len(batch)
            except Exception as e:  # a collector outage must never reach the serving path
                self.export_errors += 1
                self.dropped += len(batch)
                if self.export_errors in (1, 10, 100) or self.export_errors % 1000 == 0:
                    logger.warning("tracing: export to %s failed (%d so far): %s", self.endpoint,
                                   self.export_errors, e)
                return sent

    def _send(self, payload: bytes) -> None:
        if self.protocol.startswith("http"):
            if self.protocol == "http/json":
                raise ValueError("OTLP http/json is not supported; use grpc or http/protobuf")
            req = urllib.request.Request(self.endpoint, data=payload, method="POST",
                                         headers={"Content-Type": "application/x-protobuf", **self.headers})
            with urllib.request.urlopen(req, timeout=self.timeout_s) as r:
                r.read()
            return
        if self._rpc is None:
            import grpc

            target = self.endpoint
            for p in ("http://", "https://", "grpc://"):
                if target.startswith(p):
                    target = target[len(p):]
            target = target.rstrip("/")
            self._channel = (grpc.insecure_channel(target) if self.insecure
                             else grpc.secure_channel(target, grpc.ssl_channel_credentials()))
            self._rpc = self._channel.unary_unary(_EXPORT_PATH, request_serializer=lambda b: b,
                                                  response_deserializer=lambda b: b)
        self._rpc(payload, timeout=self.timeout_s, metadata=[(k.lower(), v) for k, v in self.headers.items()])

    def shutdown(self) -> None:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=self.timeout_s)
        if self.enabled:
            self.flush()
        if self._channel is not None:
            self._channel.close()
            self._channel = self._rpc = None


DISABLED = Tracer(None)
