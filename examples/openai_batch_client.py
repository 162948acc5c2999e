#!/usr/bin/env python3
"""OpenAI Files + Batches against the router (start it with --enable-batch-api).

Uploads a JSONL file of chat-completion requests, creates a batch over it, polls until the batch
finishes and prints the output file.  Uses only the standard library.

    python examples/openai_batch_client.py --base-url http://localhost:30080 --input examples/batch.jsonl
"""
import argparse
import json
import time
import urllib.request
import uuid


def call(method, url, body=None, headers=None):
    data = body if isinstance(body, (bytes, type(None))) else json.dumps(body).encode()
    hdr = dict(headers or {})
    if body is not None and "Content-Type" not in hdr:
        hdr["Content-Type"] = "application/json"
    req = urllib.request.Request(url, data=data, method=method, headers=hdr)
    with urllib.request.urlopen(req, timeout=300) as r:  # noqa: S310
        return r.read()


def upload(base, path):
    boundary = uuid.uuid4().hex
    with open(path, "rb") as f:
        content = f.read()
    parts = [f'--{boundary}\r\nContent-Disposition: form-data; name="purpose"\r\n\r\nbatch\r\n'.encode(),
             f'--{boundary}\r\nContent-Disposition: form-data; name="file"; filename="{path}"\r\n'
             f"Content-Type: application/jsonl\r\n\r\n".encode() + content + b"\r\n", f"--{boundary}--\r\n".encode()]
    out = call("POST", base + "/v1/files", b"".join(parts),
               {"Content-Type": f"multipart/form-data; boundary={boundary}"})
    return json.loads(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-url", default="http://localhost:30080")
    ap.add_argument("--input", default="examples/batch.jsonl")
    a = ap.parse_args()
    f = upload(a.base_url, a.input)
    print("uploaded", f["id"], f["bytes"], "bytes")
    batch = json.loads(call("POST", a.base_url + "/v1/batches",
                            {"input_file_id": f["id"], "endpoint": "/v1/chat/completions", "completion_window": "1h"}))
    print("batch", batch["id"], batch["status"])
    while batch["status"] not in ("completed", "failed", "cancelled", "expired"):
        time.sleep(1)
        batch = json.loads(call("GET", f"{a.base_url}/v1/batches/{batch['id']}"))
    print("finished:", batch["status"])
    if batch.get("output_file_id"):
        print(call("GET", f"{a.base_url}/v1/files/{batch['output_file_id']}/content").decode())


if __name__ == "__main__":
    main()
