"""Release bookkeeping (tools/release.py) and the workflows that call it."""

import re
import sys
from pathlib import Path

import pytest
import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import release  # noqa: E402


def test_semver_bumps():
    assert release.bump("0.1.0", "patch") == "0.1.1"
    assert release.bump("0.1.9", "minor") == "0.2.0"
    assert release.bump("1.4.7", "major") == "2.0.0"
    for bad in ("1.2", "v1.2.3", "1.2.3-rc1", ""):
        with pytest.raises(ValueError):
            release.bump(bad, "patch")
    with pytest.raises(ValueError):
        release.bump("1.2.3", "huge")


def test_rewrite_touches_chart_and_package_and_nothing_else(tmp_path):
    chart = tmp_path / "Chart.yaml"
    chart.write_text('apiVersion: v2\nname: vllm-stack\nversion: 0.3.9\nappVersion: "0.3.9"\ndependencies:\n- name: x\n  version: 1.2.3\n')
    py = tmp_path / "pyproject.toml"
    py.write_text('[project]\nname = "p"\nversion = "0.3.9"\n[tool.x]\nversion = "9.9.9"\n')
    assert release.current(chart) == "0.3.9"
    changed = release.rewrite("0.4.0", chart, py)
    assert set(changed) == {chart, py}
    doc = yaml.safe_load(chart.read_text())
    assert doc["version"] == "0.4.0" and doc["appVersion"] == "0.4.0"
    assert doc["dependencies"][0]["version"] == "1.2.3"  # a dependency's version is not ours
    assert 'version = "0.4.0"' in py.read_text() and 'version = "9.9.9"' in py.read_text()
    assert release.rewrite("0.4.0", chart, py) == []  # idempotent


def test_chart_and_package_agree_on_the_version():
    chart = release.current()
    m = re.search(r'^version\s*=\s*"([0-9.]+)"', (ROOT / "pyproject.toml").read_text(), re.M)
    assert m and m[1] == chart


def test_cli_outputs_are_what_the_workflow_parses(capsys):
    assert release.main(["bump", "minor", "--dry-run"]) == 0
    out = capsys.readouterr().out.splitlines()
    cur = release.current()
    assert out == [f"current_version={cur}", f"new_version={release.bump(cur, 'minor')}"]


def test_workflows_are_valid_and_cover_every_image():
    wf = ROOT / ".github" / "workflows"
    docs = {p.name: yaml.safe_load(p.read_text()) for p in wf.glob("*.yml")}
    for name in ("scheduled-release.yml", "router-docker-release.yml", "helm-release.yml"):
        assert name in docs and "jobs" in docs[name], name
    rel = docs["scheduled-release.yml"]
    # PyYAML reads the key `on` as boolean True
    triggers = rel.get("on") or rel.get(True)
    assert triggers["schedule"][0]["cron"] == "0 16 * * 1"
    assert set(triggers["workflow_dispatch"]["inputs"]) == {"release_type", "dry_run"}
    images = rel["jobs"]["images"]["strategy"]["matrix"]["image"]
    dockerfiles = {p.name.split(".", 1)[1] for p in (ROOT / "docker").glob("Dockerfile.*")}
    assert set(images) == dockerfiles
    assert rel["jobs"]["github-release"]["needs"] == ["prepare", "chart", "images"]
    text = (wf / "scheduled-release.yml").read_text()
    assert "tools/release.py bump" in text and "tools/release.py notes" in text
