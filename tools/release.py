"""Version bookkeeping of a release: one semantic version shared by the Helm chart
(deploy/helm/Chart.yaml: version and appVersion), the Python package (pyproject.toml) and the
image tags.

    python tools/release.py current                 # 0.1.0
    python tools/release.py next patch|minor|major  # what the next version would be
    python tools/release.py bump patch [--dry-run]  # rewrite the files, print old and new
    python tools/release.py notes <since-tag>       # release notes from the commit subjects

The release workflow (.github/workflows/scheduled-release.yml) is a thin wrapper around these
commands, so what it does can be run and tested locally.  Counterpart of the version-bump steps
in /root/reference/.github/workflows/scheduled-release.yml:60-100 (there: inline shell + semver).
"""

from __future__ import annotations

import argparse
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CHART = ROOT / "deploy" / "helm" / "Chart.yaml"
PYPROJECT = ROOT / "pyproject.toml"
SEMVER = re.compile(r"^(\d+)\.(\d+)\.(\d+)$")


def parse(v: str) -> tuple[int, int, int]:
    m = SEMVER.match(v.strip())
    if not m:
        raise ValueError(f"not a MAJOR.MINOR.PATCH version: {v!r}")
    return int(m[1]), int(m[2]), int(m[3])


def bump(v: str, kind: str) -> str:
    major, minor, patch = parse(v)
    if kind == "major":
        return f"{major + 1}.0.0"
    if kind == "minor":
        return f"{major}.{minor + 1}.0"
    if kind == "patch":
        return f"{major}.{minor}.{patch + 1}"
    raise ValueError(f"bump kind must be patch, minor or major, not {kind!r}")


def current(chart: Path = CHART) -> str:
    m = re.search(r"^version:\s*([0-9.]+)\s*$", chart.read_text(), re.M)
    if not m:
        raise ValueError(f"no version line in {chart}")
    parse(m[1])
    return m[1]


def rewrite(new: str, chart: Path = CHART, pyproject: Path = PYPROJECT) -> list[Path]:
    """Set the version everywhere it is written down; returns the files that changed."""
    parse(new)
    changed = []
    text = chart.read_text()
    out = re.sub(r"^version:\s*[0-9.]+\s*$", f"version: {new}", text, count=1, flags=re.M)
    out = re.sub(r'^appVersion:\s*"?[0-9.]+"?\s*$', f'appVersion: "{new}"', out, count=1, flags=re.M)
    if out != text:
        chart.write_text(out)
        changed.append(chart)
    if pyproject.exists():
        text = pyproject.read_text()
        out = re.sub(r'^version\s*=\s*"[0-9.]+"\s*$', f'version = "{new}"', text, count=1, flags=re.M)
        if out != text:
            pyproject.write_text(out)
            changed.append(pyproject)
    return changed


def notes(since: str | None) -> str:
    rng = f"{since}..HEAD" if since else "HEAD"
    log = subprocess.run(["git", "log", "--no-merges", "--pretty=%s", rng], cwd=ROOT, capture_output=True, text=True)
    subjects = [s for s in log.stdout.splitlines() if s and not s.startswith("Bump version")]
    if not subjects:
        return "No changes since the previous release.\n"
    return "".join(f"* {s}\n" for s in subjects)


def main(argv: list[str] | None = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    sub.add_parser("current")
    p = sub.add_parser("next")
    p.add_argument("kind", choices=["patch", "minor", "major"])
    p = sub.add_parser("bump")
    p.add_argument("kind", choices=["patch", "minor", "major"])
    p.add_argument("--dry-run", action="store_true")
    p = sub.add_parser("notes")
    p.add_argument("since", nargs="?")
    a = ap.parse_args(argv)
    if a.cmd == "current":
        print(current())
    elif a.cmd == "next":
        print(bump(current(), a.kind))
    elif a.cmd == "bump":
        old = current()
        new = bump(old, a.kind)
        files = [] if a.dry_run else rewrite(new)
        print(f"current_version={old}")
        print(f"new_version={new}")
        for f in files:
            print(f"changed={f.relative_to(ROOT)}", file=sys.stderr)
    elif a.cmd == "notes":
        sys.stdout.write(notes(a.since))
    return 0


if __name__ == "__main__":
    sys.exit(main())
