"""A collector that fits in a terminal: receives OTLP/HTTP trace exports (protobuf) and prints
one line per span, children indented under their parents.

    python tools/otlp_dump.py --port 4318 &
    OTEL_EXPORTER_OTLP_ENDPOINT=http://127.0.0.1:4318 OTEL_EXPORTER_OTLP_PROTOCOL=http/protobuf vpsa-router ...

For checking that router and engines export what they should before a real collector (Jaeger,
Tempo) is in place.  Decodes the few fields it prints straight from the wire format
(opentelemetry-proto field numbers: ResourceSpans 1 -> ScopeSpans 2 -> Span 2; Span: trace_id 1,
span_id 2, parent 4, name 5, start 7, end 8, attributes 9, events 11), no protobuf package needed.
"""

from __future__ import annotations

import argparse
import struct
import sys
from http.server import BaseHTTPRequestHandler, HTTPServer


def _varint(buf: bytes, i: int) -> tuple[int, int]:
    shift = n = 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, i


def _fields(buf: bytes):
    i = 0
    while i < len(buf):
        key, i = _varint(buf, i)
        no, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        yield no, v


def _all(buf: bytes, no: int) -> list:
    return [v for n, v in _fields(buf) if n == no]


def _value(buf: bytes):
    for no, v in _fields(buf):
        if no == 1:
            return v.decode(errors="replace")
        if no == 2:
            return bool(v)
        if no == 3:
            return v if v < (1 << 63) else v - (1 << 64)
        if no == 4:
            return round(struct.unpack("<d", v)[0], 6)
        if no == 5:
            return [_value(x) for x in _all(v, 1)]
    return None


def _attrs(kvs: list[bytes]) -> dict:
    return {_all(kv, 1)[0].decode(): _value(_all(kv, 2)[0]) for kv in kvs if _all(kv, 1) and _all(kv, 2)}


def decode(payload: bytes) -> list[dict]:
    spans = []
    for rs in _all(payload, 1):
        res = _all(rs, 1)
        service = _attrs(_all(res[0], 1)).get("service.name", "?") if res else "?"
        for ss in _all(rs, 2):
            for sp in _all(ss, 2):
                start = struct.unpack("<Q", _all(sp, 7)[0])[0]
                end = struct.unpack("<Q", _all(sp, 8)[0])[0]
                spans.append({"service": service, "trace": _all(sp, 1)[0].hex(), "id": _all(sp, 2)[0].hex(),
                              "parent": (_all(sp, 4) or [b""])[0].hex(), "name": _all(sp, 5)[0].decode(),
                              "start": start, "ms": (end - start) / 1e6, "attrs": _attrs(_all(sp, 9)),
                              "events": [_all(e, 2)[0].decode() for e in _all(sp, 11) if _all(e, 2)]})
    return spans


def show(spans: list[dict], out=sys.stdout) -> None:
    ids = {s["id"] for s in spans}
    kids: dict[str, list[dict]] = {}
    for s in spans:
        kids.setdefault(s["parent"] if s["parent"] in ids else "", []).append(s)

    def walk(parent: str, depth: int) -> None:
        for s in sorted(kids.get(parent, []), key=lambda x: x["start"]):
            attrs = " ".join(f"{k}={v}" for k, v in s["attrs"].items())
            ev = f" events={s['events']}" if s["events"] else ""
            print(f"{'  ' * depth}[{s['service']}] {s['name']}  {s['ms']:.1f} ms  trace={s['trace'][:8]}… "
                  f"{attrs}{ev}", file=out, flush=True)
            walk(s["id"], depth + 1)

    walk("", 0)


class _Handler(BaseHTTPRequestHandler):
    def do_POST(self):  # noqa: N802
        body = self.rfile.read(int(self.headers.get("Content-Length", "0")))
        try:
            show(decode(body))
        except Exception as e:  # noqa: BLE001
            print(f"undecodable export of {len(body)} bytes: {e}", flush=True)
        self.send_response(200)
        self.send_header("Content-Type", "application/x-protobuf")
        self.send_header("Content-Length", "0")
        self.end_headers()

    def log_message(self, *a):
        pass


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=4318)
    a = ap.parse_args()
    print(f"listening for OTLP/HTTP on http://{a.host}:{a.port}/v1/traces", flush=True)
    HTTPServer((a.host, a.port), _Handler).serve_forever()


if __name__ == "__main__":
    main()
