"""Play the gateway's side of the ext_proc protocol against a running endpoint-picker shim.

    vpsa-extproc --port 9003 --spawn-picker prefix --pods llama3-0=10.0.0.1,llama3-1=10.0.0.2 &
    python examples/gateway_extproc_probe.py --port 9003 --prompt "You are a helpful assistant. ..."

Sends what Envoy's ext_proc filter sends for one HTTP request (request headers, then the buffered
JSON body, then the response-headers phase) and prints what the shim answers: the
``x-gateway-destination-endpoint`` header mutation, or the immediate response when the pool is empty.
Useful for checking a picker Deployment before a Gateway is wired to it."""

from __future__ import annotations

import argparse
import asyncio
import json

import grpc

from vllm_production_stack_amd.gateway import extproc_pb as pb


async def probe(port: int, doc: dict, path: str, user: str) -> None:
    channel = grpc.aio.insecure_channel(f"127.0.0.1:{port}")
    health = channel.unary_unary(f"/{pb.HEALTH_SERVICE}/Check", request_serializer=lambda m: m.SerializeToString(),
                                 response_deserializer=pb.HealthCheckResponse.FromString)
    print("health:", "SERVING" if (await health(pb.HealthCheckRequest(service=""))).status == 1 else "NOT_SERVING")
    process = channel.stream_stream(pb.METHOD, request_serializer=lambda m: m.SerializeToString(),
                                    response_deserializer=pb.ProcessingResponse.FromString)
    call = process()
    m = pb.ProcessingRequest()
    for k, v in ((":method", "POST"), (":path", path), ("content-type", "application/json"), ("x-user-id", user)):
        if v:
            h = m.request_headers.headers.headers.add()
            h.key, h.raw_value = k, v.encode()
    await call.write(m)
    r = await call.read()
    print("request_headers  ->", r.WhichOneof("response"), "(no decision yet: the prompt is in the body)")
    m = pb.ProcessingRequest()
    m.request_body.body, m.request_body.end_of_stream = json.dumps(doc).encode(), True
    await call.write(m)
    r = await call.read()
    kind = r.WhichOneof("response")
    if kind == "immediate_response":
        print("request_body     -> immediate_response", r.immediate_response.status.code, r.immediate_response.body.decode())
    else:
        for s in r.request_body.response.header_mutation.set_headers:
            print(f"request_body     -> set header {s.header.key}: {s.header.raw_value.decode()}"
                  f"  clear_route_cache={r.request_body.response.clear_route_cache}")
    await call.done_writing()
    await channel.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--port", type=int, default=9003)
    ap.add_argument("--path", default="/v1/completions")
    ap.add_argument("--model", default="llama3")
    ap.add_argument("--prompt", default="Hello")
    ap.add_argument("--user", default="")
    a = ap.parse_args()
    asyncio.run(probe(a.port, {"model": a.model, "prompt": a.prompt}, a.path, a.user))


if __name__ == "__main__":
    main()
