"""The operator without a proxy sidecar: the API server over TLS, verified against a CA bundle,
authenticated with the service account's token, and a metrics port of its own over TLS that only
answers callers the API server authenticates and authorizes.

What the reference gets from controller-runtime
(/root/reference/operator/cmd/main.go:107-179 certificate watchers, :140-152 secure metrics with
the authentication-and-authorization filter, :181 in-cluster config).  The API server is
tests/fake_k8s.py behind a certificate made with the ``openssl`` tool; engine pods are the fake
engines of the end-to-end operator test."""

import os
import shutil
import ssl
import subprocess
import time
import urllib.error
import urllib.request

import pytest

from fake_k8s import FakeKube
from test_operator_e2e import CRS, OPERATOR, PODS, Backends, _free_port, lora_cr, pod, wait_for

pytestmark = pytest.mark.skipif(shutil.which("openssl") is None, reason="needs the openssl tool to make certificates")

SA_TOKEN = "sa-token-of-the-operator"


def _cert(dir_, name, cn="127.0.0.1", days=2):
    """Self-signed certificate for 127.0.0.1 (an IP subjectAltName, as an API server's has)."""
    crt, key = dir_ / f"{name}.crt", dir_ / f"{name}.key"
    subprocess.run(["openssl", "req", "-x509", "-newkey", "rsa:2048", "-nodes", "-keyout", str(key), "-out", str(crt),
                    "-days", str(days), "-subj", f"/CN={cn}", "-addext", "subjectAltName=IP:127.0.0.1,DNS:localhost"],
                   check=True, capture_output=True)
    return str(crt), str(key)


@pytest.fixture
def cluster(tmp_path):
    """A TLS API server that wants the service account's token, and the volume a pod would see."""
    crt, key = _cert(tmp_path, "apiserver")
    kube = FakeKube(tls=(crt, key), token=SA_TOKEN).start()
    sa = tmp_path / "serviceaccount"
    sa.mkdir()
    (sa / "token").write_text(SA_TOKEN + "\n")
    shutil.copy(crt, sa / "ca.crt")
    (sa / "namespace").write_text("default")
    yield kube, sa, tmp_path
    kube.stop()


def _spawn(kube, sa, *extra, api=False, health_port=None):
    env = dict(os.environ, VPSA_SA_DIR=str(sa), KUBERNETES_SERVICE_HOST="127.0.0.1", KUBERNETES_SERVICE_PORT=str(kube.port))
    hp = health_port or _free_port()
    args = [str(OPERATOR), "run", "--namespace", "default", "--interval", "30", "--health-port", str(hp), *extra]
    if api:
        args += ["--api", kube.url]
    return subprocess.Popen(args, env=env, stderr=subprocess.PIPE, text=True), hp


def _stop(proc):
    proc.terminate()
    try:
        proc.wait(5)
    except subprocess.TimeoutExpired:
        proc.kill()
    return proc.stderr.read()


def test_in_cluster_config_reconciles_over_verified_tls(cluster):
    kube, sa, _ = cluster
    backends = Backends()
    backends.add("127.0.0.2")
    proc, _hp = _spawn(kube, sa, "--sidecar-port", str(backends.sidecar_port))
    try:
        kube.put(f"{PODS}/pod-a", pod("pod-a", "127.0.0.2"))
        cr = lora_cr()
        cr["spec"]["vllmPort"] = backends.engine_port
        kube.put(f"{CRS}/loraadapters/ad-sample", cr)
        wait_for(lambda: kube.objects[f"{CRS}/loraadapters/ad-sample"].get("status", {}).get("phase") == "Ready",
                 what="the adapter to be loaded through the TLS API connection")
        assert "sql-lora" in backends.engines["127.0.0.2"].lora
    finally:
        err = _stop(proc)
        backends.stop()
    assert "TLS, CA bundle" in err and "service-account token" in err, err


def test_a_server_certificate_from_another_ca_is_refused(cluster):
    kube, sa, tmp = cluster
    other_crt, _ = _cert(tmp, "other")
    shutil.copy(other_crt, sa / "ca.crt")  # the pod trusts a CA that did not sign the server
    proc, _hp = _spawn(kube, sa, "--once")
    proc.wait(20)
    proc.stderr.read()
    assert not [r for r in kube.requests], "no request may get through an unverified connection"
    # the same server with verification switched off (what a developer does against a throw-away cluster)
    proc, _hp = _spawn(kube, sa, "--once", "--insecure-skip-tls-verify")
    proc.wait(20)
    assert "NOT verified" in proc.stderr.read()
    assert kube.requests


def test_a_rotated_token_is_picked_up(cluster, monkeypatch):
    kube, sa, _ = cluster
    (sa / "token").write_text("stale")
    proc, _hp = _spawn(kube, sa, "--once")
    proc.wait(20)
    proc.stderr.read()
    # every call was turned away: the server saw requests, none created or listed anything
    assert kube.requests and not kube.objects
    (sa / "token").write_text(SA_TOKEN)
    kube.put(f"{CRS}/loraadapters/ad-sample", lora_cr())
    proc, _hp = _spawn(kube, sa, "--once")
    proc.wait(20)
    proc.stderr.read()
    assert kube.objects[f"{CRS}/loraadapters/ad-sample"].get("status"), "the fresh token authenticates"


def _scrape(port, token=None, cafile=None, scheme="https"):
    ctx = None
    if scheme == "https":
        ctx = ssl.create_default_context(cafile=cafile) if cafile else ssl._create_unverified_context()
    req = urllib.request.Request(f"{scheme}://127.0.0.1:{port}/metrics")
    if token:
        req.add_header("Authorization", f"Bearer {token}")
    try:
        with urllib.request.urlopen(req, timeout=5, context=ctx) as r:
            return r.status, r.read().decode()
    except urllib.error.HTTPError as e:
        return e.code, e.read().decode()


def test_secure_metrics_ask_the_api_server_who_may_read(cluster):
    kube, sa, tmp = cluster
    kube.users = {"prom-token": "system:serviceaccount:monitoring:prometheus", "guest-token": "guest"}
    kube.metrics_readers = {"system:serviceaccount:monitoring:prometheus"}
    crt, key = _cert(tmp, "tls")  # the names a cert-manager secret mounts
    mport = _free_port()
    proc, hp = _spawn(kube, sa, "--metrics-bind-address", f"127.0.0.1:{mport}", f"--metrics-cert-path={tmp}")
    try:
        wait_for(lambda: _try(lambda: _scrape(mport, "prom-token", cafile=crt)[0]) == 200, what="the metrics port")
        status, text = _scrape(mport, "prom-token", cafile=crt)
        assert status == 200 and "vpsa_operator_leader 1" in text
        assert _scrape(mport, cafile=crt)[0] == 401  # no token
        assert _scrape(mport, "nobody-knows-this", cafile=crt)[0] == 401  # not authenticated
        assert _scrape(mport, "guest-token", cafile=crt)[0] == 403  # authenticated, not allowed
        # verdicts are remembered: three more scrapes, no more reviews
        reviews = sum(1 for m, p in kube.requests if p.endswith("tokenreviews"))
        for _ in range(3):
            assert _scrape(mport, "prom-token", cafile=crt)[0] == 200
        assert sum(1 for m, p in kube.requests if p.endswith("tokenreviews")) == reviews
        # the probe port no longer carries the series once they have a port of their own
        with pytest.raises(urllib.error.HTTPError):
            urllib.request.urlopen(f"http://127.0.0.1:{hp}/metrics", timeout=5)
        assert urllib.request.urlopen(f"http://127.0.0.1:{hp}/healthz", timeout=5).status == 200
        # plain HTTP to the TLS port gets nothing
        with pytest.raises(Exception):
            _scrape(mport, "prom-token", scheme="http")

        # certificate rotation: new files under the same names are served without a restart
        time.sleep(1.1)  # a distinct modification time
        new_crt, _ = _cert(tmp, "tls", cn="rotated")
        assert _scrape(mport, "prom-token", cafile=new_crt)[0] == 200
        der = ssl.PEM_cert_to_DER_cert(open(new_crt).read())
        assert ssl.PEM_cert_to_DER_cert(ssl.get_server_certificate(("127.0.0.1", mport))) == der
    finally:
        err = _stop(proc)
    assert "TLS, token review" in err, err


def _try(fn):
    try:
        return fn()
    except Exception:
        return None


def test_metrics_without_a_mounted_certificate_use_a_self_signed_one(cluster):
    kube, sa, _ = cluster
    kube.users = {"prom-token": "prom"}
    kube.metrics_readers = {"prom"}
    mport = _free_port()
    proc, _hp = _spawn(kube, sa, "--metrics-bind-address", f"127.0.0.1:{mport}")
    try:
        wait_for(lambda: _try(lambda: _scrape(mport, "prom-token")[0]) == 200, what="the metrics port")
        pem = ssl.get_server_certificate(("127.0.0.1", mport))
        assert "BEGIN CERTIFICATE" in pem
    finally:
        _stop(proc)


def test_plain_metrics_port_when_security_is_switched_off(cluster):
    kube, sa, _ = cluster
    mport = _free_port()
    proc, _hp = _spawn(kube, sa, "--metrics-bind-address", f"127.0.0.1:{mport}", "--metrics-secure=false")
    try:
        wait_for(lambda: _try(lambda: _scrape(mport, scheme="http")[0]) == 200, what="the metrics port")
        assert "vpsa_operator_leader" in _scrape(mport, scheme="http")[1]
    finally:
        _stop(proc)
