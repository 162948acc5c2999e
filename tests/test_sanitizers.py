"""Host-side native code under sanitizers: the routing core's concurrency self-test is built with
AddressSanitizer+UBSan and with ThreadSanitizer and must finish clean.  (GPU code cannot run under
a sanitizer on the test machines; its checks are the numerics tests in test_kernels_gpu.py.)"""

import os
import random
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = [str(ROOT / "csrc/routercore/selftest.cpp"), str(ROOT / "csrc/routercore/routercore.cpp")]


@pytest.mark.parametrize("name,flags", [
    ("asan-ubsan", ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]),
    ("tsan", ["-fsanitize=thread"]),
])
def test_routercore_selftest_under_sanitizer(tmp_path, name, flags):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / f"selftest-{name}"
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", *flags, "-I", str(ROOT / "csrc"), *SRC,
                            "-o", str(exe), "-lpthread"], capture_output=True, text=True)
    if build.returncode != 0 and "cannot find" in build.stderr:
        pytest.skip(f"sanitizer runtime not installed: {build.stderr.splitlines()[-1]}")
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", TSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)  # sanitizer runtimes insist on being the first loaded library
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "selftest ok" in run.stdout
    assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr


SAN_FLAGS = ["-fsanitize=address,undefined", "-fsanitize=float-cast-overflow", "-fno-sanitize-recover=undefined",
             "-fno-omit-frame-pointer"]
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
           "TSAN_OPTIONS": "halt_on_error=1"}


def _build(tmp_path, name, sources, flags, include=("csrc",), libs=()):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / name
    inc = [x for d in include for x in ("-I", str(ROOT / d))]
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", *flags, *inc, *[str(ROOT / s) for s in sources],
                            "-o", str(exe), *libs, "-lpthread"], capture_output=True, text=True)
    if build.returncode != 0 and "cannot find" in build.stderr:
        pytest.skip(f"sanitizer runtime not installed: {build.stderr.splitlines()[-1]}")
    assert build.returncode == 0, build.stderr
    return exe


def _san_env():
    env = dict(os.environ, **SAN_ENV)
    env.pop("LD_PRELOAD", None)
    return env


def _clean(stderr: str) -> None:
    for mark in ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "stack-overflow",
                 "terminate called"):
        assert mark not in stderr, stderr[-4000:]


def test_range_coder_survives_malformed_streams_under_asan_ubsan(tmp_path):
    """The KV codec's entropy coder decodes bytes that come back from disk and from remote cache
    servers: truncated, empty, random and bit-flipped streams, and valid streams decoded with the
    wrong geometry, under AddressSanitizer + UBSan."""
    exe = _build(tmp_path, "rc-selftest", ["csrc/kvio/rc_selftest.cpp"], SAN_FLAGS, include=("csrc/kvio",))
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=_san_env())
    assert run.returncode == 0, run.stdout + run.stderr
    assert "rangecoder selftest ok" in run.stdout
    _clean(run.stderr)


@pytest.mark.parametrize("name,flags", [("asan-ubsan", SAN_FLAGS), ("tsan", ["-fsanitize=thread"])])
def test_json_and_http_parsers_survive_hostile_input(tmp_path, name, flags):
    """common/json.hpp and common/http.hpp (every native control-plane binary is built on them):
    malformed and mutated JSON, 100000-deep nesting, lone surrogates, non-UTF-8 bytes, numbers out
    of range; a live server fed truncated / oversized / mis-framed / binary requests on many
    connections at once; the client and the watch-stream reader against an upstream that answers
    with broken status lines, lying chunk sizes and endless headers."""
    exe = _build(tmp_path, f"parse-selftest-{name}", ["csrc/common/parse_selftest.cpp"], flags, include=("csrc/common",))
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=_san_env())
    assert run.returncode == 0, run.stdout + run.stderr
    assert "parse selftest ok" in run.stdout
    _clean(run.stderr)


def _raw(port: int, payload: bytes, timeout=5.0) -> bytes:
    import socket

    with socket.create_connection(("127.0.0.1", port), timeout=timeout) as s:
        try:
            s.sendall(payload)
            s.shutdown(socket.SHUT_WR)
        except OSError:
            pass  # the server hung up first: allowed
        out = b""
        try:
            while chunk := s.recv(65536):
                out += chunk
        except OSError:
            pass
        return out


def _post(port: int, path: str, body: bytes) -> bytes:
    return _raw(port, b"POST " + path.encode() + b" HTTP/1.1\r\nContent-Length: " + str(len(body)).encode()
                + b"\r\n\r\n" + body)


def test_picker_service_under_asan_ubsan_with_hostile_requests(tmp_path):
    """The real picker binary (the gateway's endpoint-picker shim), sanitizer-built, is sent
    requests no client should send; it must answer or hang up, stay alive, and keep serving."""
    import json
    import random
    import re

    exe = _build(tmp_path, "picker-asan", ["csrc/routercore/picker_service.cpp", "csrc/routercore/routercore.cpp"],
                 SAN_FLAGS)
    proc = subprocess.Popen([str(exe), "--picker", "prefix", "--host", "127.0.0.1", "--port", "0"],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_san_env())
    try:
        port = int(re.search(r":(\d+)\s*$", proc.stdout.readline()).group(1))
        pods = [{"name": f"p{i}", "address": f"10.0.0.{i}"} for i in range(4)]
        good = json.dumps({"pods": pods, "prompt": "hello world " * 50, "model": "m"}).encode()
        assert b" 200 " in _post(port, "/pick", good).split(b"\r\n")[0]
        hostile = [
            b"", b"{", b"null", b"[]", b"42", b'"str"', b'{"pods":null}', b'{"pods":{}}', b'{"pods":[]}',
            b'{"pods":[1,2,"x",null,[],{}]}', b'{"pods":[{"name":5,"address":[]}],"prompt":{"a":1}}',
            b'{"pods":[{"name":"a","address":"b"}],"prompt":"\\ud800\\udfff\\ud800"}',
            b'{"pods":[{"name":"a","address":"b"}],"prompt":"' + bytes(range(128, 256)) * 40 + b'"}',
            b'{"pods":[{"name":"a","address":"b"}],"prompt":"' + b"x" * (8 << 20) + b'"}',
            b'{"pods":[{"name":"","address":""}],"prompt":""}',
            b'{"pods":[' + b",".join(b'{"name":"n%d","address":"a"}' % i for i in range(20000)) + b'],"prompt":"p"}',
            b"[" * 200000, b'{"a":' * 100000, b'{"pods":[{"name":"a","address":"b"}],"prompt":1e999}',
            b'{"pods":[{"name":"a\x00b","address":"b"}],"prompt":"\x00"}',
        ]
        for body in hostile:
            first = _post(port, "/pick", body).split(b"\r\n")[0]
            assert first == b"" or b" 400 " in first or b" 200 " in first or b" 500 " in first, first
            assert proc.poll() is None, proc.stderr.read()[-3000:]
        rng = random.Random(7)
        for i in range(300):
            b = bytearray(good)
            for _ in range(rng.randint(1, 6)):
                k = rng.randrange(len(b))
                op = rng.randrange(4)
                if op == 0:
                    b[k] = rng.randrange(256)
                elif op == 1:
                    del b[k:k + rng.randint(1, 30)]
                elif op == 2:
                    b[k:k] = bytes([rng.randrange(256)]) * rng.randint(1, 200)
                else:
                    del b[k:]
                if not b:
                    b = bytearray(b"{")
            _post(port, "/pick", bytes(b))
        for raw in [b"GARBAGE\r\n\r\n", b"POST /pick HTTP/1.1\r\nContent-Length: 99999999999\r\n\r\n", b"\x00" * 5000,
                    b"POST /pick HTTP/1.1\r\nContent-Length: 50\r\n\r\nshort", b"GET /" + b"a" * (2 << 20)]:
            _raw(port, raw)
        assert proc.poll() is None
        assert b" 200 " in _post(port, "/pick", good).split(b"\r\n")[0]
        assert b" 200 " in _raw(port, b"GET /health HTTP/1.1\r\n\r\n").split(b"\r\n")[0]
    finally:
        proc.terminate()
        try:
            proc.wait(10)
        except subprocess.TimeoutExpired:
            proc.kill()
    _clean(proc.stderr.read())


def test_operator_under_asan_ubsan_with_a_hostile_api_server(tmp_path):
    """The real operator binary, sanitizer-built, against an API server whose objects have the
    wrong types everywhere, whose lists are cut off mid-document and whose watch streams carry
    junk lines: it must keep running, and still reconcile the one well-formed object."""
    import sys
    import time

    sys.path.insert(0, str(ROOT / "tests"))
    from aiohttp import web
    from fake_k8s import FakeKube
    from test_native_control_plane import RUNTIME_CR

    exe = _build(tmp_path, "operator-asan", ["csrc/operator/operator_main.cpp"], SAN_FLAGS)
    crs = "/apis/production-stack.vllm.ai/v1alpha1/namespaces/default"

    class HostileKube(FakeKube):
        async def handle(self, request):
            if request.method == "GET" and request.path.endswith("/cacheservers"):
                if request.query.get("watch") == "true":
                    resp = web.StreamResponse()
                    resp.enable_chunked_encoding()
                    await resp.prepare(request)
                    for junk in (b"not json\n", b'{"type":"ADDED","object":', b"\xff\xfe\x00\n", b'{"type":5,"object":[]}\n',
                                 b'{"type":"ERROR","object":{"code":"410"}}\n', b"[" * 100000 + b"\n", b"\n\n\r\n",
                                 b'{"type":"MODIFIED","object":{"metadata":{"resourceVersion":{"a":1}}}}\n'):
                        await resp.write(junk)
                    return resp
                return web.Response(body=b'{"kind":"List","metadata":{"resourceVersion":"1"},"items":[{"metadata":{"na',
                                    content_type="application/json")
            return await super().handle(request)

    kube = HostileKube().start()
    proc = None
    try:
        weird = [
            {"metadata": {"name": "w1"}, "spec": "a string"},
            {"metadata": {"name": "w2"}, "spec": None},
            {"metadata": {"name": "w3"}, "spec": {"replicas": 1e300, "vllmConfig": {"port": -5, "extraArgs": "x", "env": [1, None, {"name": 3}]},
                                                   "model": [], "deploymentConfig": {"replicas": "many", "resources": 7}}},
            {"metadata": {"name": "w4", "finalizers": "nope", "labels": [1, 2]}, "spec": {"model": {"modelURL": {"a": 1}, "maxModelLen": "x"}}},
            {"metadata": {"name": "w5é\U0001F600 \"quoted\"\n"}, "spec": {"vllmConfig": {"port": 1e9, "extraArgs": [[], {}, None, 5]}}},
            {"metadata": "not an object", "spec": {}},
            {"spec": {"replicas": -1e300}},
        ]
        for kind in ("vllmruntimes", "vllmrouters", "loraadapters"):
            for i, w in enumerate(weird):
                kube.put(f"{crs}/{kind}/weird{i}", dict(w, apiVersion="production-stack.vllm.ai/v1alpha1",
                                                        kind={"vllmruntimes": "VLLMRuntime", "vllmrouters": "VLLMRouter",
                                                              "loraadapters": "LoraAdapter"}[kind]))
        kube.put("/api/v1/namespaces/default/pods/badpod", {"metadata": {"name": "badpod", "labels": {"model": "llama"}},
                                                          "status": {"podIP": 12345, "containerStatuses": {"ready": "yes"}}})
        kube.put(f"{crs}/vllmruntimes/vllmruntime-sample", RUNTIME_CR)
        proc = subprocess.Popen([str(exe), "run", "--api", kube.url, "--namespace", "default", "--interval", "1",
                                 "--health-port", "0"], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True,
                                env=_san_env())
        key = "/apis/apps/v1/namespaces/default/deployments/vllmruntime-sample"
        t0 = time.time()
        while time.time() - t0 < 20 and key not in kube.objects:
            assert proc.poll() is None, proc.stderr.read()[-4000:]
            time.sleep(0.1)
        assert key in kube.objects, "the well-formed VLLMRuntime was not reconciled"
        # keep it under fire for a few more passes and a burst of events on the hostile objects
        for n in range(30):
            kube.mutate(f"{crs}/vllmrouters/weird2", lambda o, n=n: o.update(spec={"replicas": n * 1e18, "port": [n]}))
        time.sleep(3.0)
        assert proc.poll() is None, proc.stderr.read()[-4000:]
    finally:
        if proc is not None:
            proc.terminate()
            try:
                proc.wait(10)
            except subprocess.TimeoutExpired:
                proc.kill()
        kube.stop()
    _clean(proc.stderr.read())


def test_operator_tls_paths_under_asan_ubsan(tmp_path):
    """The TLS side of the operator, sanitizer-built: the API server over https with the service
    account's CA bundle and token, and the secured metrics port fed what a port on the pod network
    gets: plain-text requests, a connection that stops in the middle of the handshake, random
    bytes, a token the API server does not know, and proper scrapes in between."""
    import socket
    import ssl
    import sys
    import time
    import urllib.error
    import urllib.request

    if shutil.which("openssl") is None:
        pytest.skip("needs the openssl tool to make certificates")
    sys.path.insert(0, str(ROOT / "tests"))
    from fake_k8s import FakeKube

    exe = _build(tmp_path, "operator-tls-asan", ["csrc/operator/operator_main.cpp"], SAN_FLAGS + ["-DVHTTP_TLS"],
                 libs=("-lssl", "-lcrypto"))
    crt, key = tmp_path / "tls.crt", tmp_path / "tls.key"
    subprocess.run(["openssl", "req", "-x509", "-newkey", "rsa:2048", "-nodes", "-keyout", str(key), "-out", str(crt), "-days", "2",
                    "-subj", "/CN=127.0.0.1", "-addext", "subjectAltName=IP:127.0.0.1"], check=True, capture_output=True)
    kube = FakeKube(tls=(str(crt), str(key)), token="sa").start()
    kube.users = {"prom": "prometheus"}
    kube.metrics_readers = {"prometheus"}
    sa = tmp_path / "sa"
    sa.mkdir()
    (sa / "token").write_text("sa")
    shutil.copy(crt, sa / "ca.crt")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        mport = s.getsockname()[1]
    env = dict(_san_env(), VPSA_SA_DIR=str(sa), KUBERNETES_SERVICE_HOST="127.0.0.1", KUBERNETES_SERVICE_PORT=str(kube.port))
    proc = subprocess.Popen([str(exe), "run", "--namespace", "default", "--interval", "1", "--health-port", "0",
                             "--metrics-bind-address", f"127.0.0.1:{mport}", "--metrics-cert-path", str(tmp_path)],
                            stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, env=env)

    def scrape(token):
        req = urllib.request.Request(f"https://127.0.0.1:{mport}/metrics", headers={"Authorization": f"Bearer {token}"})
        try:
            with urllib.request.urlopen(req, timeout=10, context=ssl.create_default_context(cafile=str(crt))) as r:
                return r.status
        except urllib.error.HTTPError as e:
            return e.code

    try:
        t0 = time.time()
        while time.time() - t0 < 20:
            assert proc.poll() is None, proc.stderr.read()[-4000:]
            try:
                if scrape("prom") == 200:
                    break
            except OSError:
                time.sleep(0.1)
        else:
            raise AssertionError("metrics port never answered")
        rng = random.Random(7)
        for i in range(40):
            with socket.create_connection(("127.0.0.1", mport), timeout=5) as c:
                kind = i % 4
                if kind == 0:
                    c.sendall(b"GET /metrics HTTP/1.1\r\nHost: x\r\n\r\n")  # plain text to a TLS port
                elif kind == 1:
                    c.sendall(bytes([0x16, 0x03, 0x01, 0x02, 0x00, 0x01, 0x00]))  # a ClientHello that never continues
                elif kind == 2:
                    c.sendall(bytes(rng.randrange(256) for _ in range(rng.randrange(1, 3000))))
                # kind 3: connect and say nothing
        assert scrape("unknown") == 401
        assert scrape("prom") == 200
        assert any(p.endswith("/vllmruntimes") for _m, p in kube.requests), "the control loop reached the API server over TLS"
    finally:
        proc.terminate()
        try:
            proc.wait(10)
        except subprocess.TimeoutExpired:
            proc.kill()
        err = proc.stderr.read()
        kube.stop()
    _clean(err)
