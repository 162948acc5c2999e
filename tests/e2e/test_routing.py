#!/usr/bin/env python3
"""Routing e2e check against a live router: sends traffic, then verifies the routing decisions by
parsing the router's log lines

    Routing request <id> with session id <sid|None> to <url> at <time>, process time = <s>

(the format is a contract: dashboards and these checks depend on it).  Works against fake engines
(CPU) and real engines alike.

    test_routing.py --router-url http://localhost:30080 --log-file router.log \\
        --routing-logic session|roundrobin|prefixaware|disaggregated_prefill|kvaware [--model M]
"""

import argparse
import collections
import json
import re
import sys
import time
import urllib.request
import uuid

LINE = re.compile(r"Routing request (\S+) with session id (\S+) to (\S+) at ([\d.]+), process time = ([\d.]+)")


def post(url, body, headers=None, timeout=120):
    req = urllib.request.Request(url, data=json.dumps(body).encode(),
                                 headers={"Content-Type": "application/json", **(headers or {})})
    with urllib.request.urlopen(req, timeout=timeout) as r:  # noqa: S310
        return r.status, r.read(), dict(r.headers)


def routed(log_file, ids):
    """request id -> [urls in log order]"""
    out = collections.defaultdict(list)
    with open(log_file, errors="replace") as f:
        for line in f:
            m = LINE.search(line)
            if m and m.group(1) in ids:
                out[m.group(1)].append(m.group(3))
    return out


def chat(args, content, rid, headers=None, max_tokens=8):
    body = {"model": args.model, "messages": [{"role": "user", "content": content}], "max_tokens": max_tokens,
            "temperature": 0}
    st, _, _ = post(args.router_url + "/v1/chat/completions", body, {"X-Request-Id": rid, **(headers or {})})
    assert st == 200, st


def check_session(args):
    ids = {}
    for user in range(6):
        for turn in range(4):
            rid = uuid.uuid4().hex
            ids[rid] = user
            chat(args, f"user {user} turn {turn}", rid, {args.session_key: f"user-{user}"})
    time.sleep(1.0)
    r = routed(args.log_file, set(ids))
    per_user = collections.defaultdict(set)
    for rid, urls in r.items():
        assert len(urls) == 1, (rid, urls)
        per_user[ids[rid]].add(urls[0])
    assert len(r) == len(ids), f"{len(ids) - len(r)} requests have no routing log line"
    assert all(len(v) == 1 for v in per_user.values()), dict(per_user)
    print(f"session: {len(per_user)} sessions each pinned to one endpoint; endpoints used: "
          f"{len(set().union(*per_user.values()))}")


def check_roundrobin(args):
    ids = [uuid.uuid4().hex for _ in range(20)]
    for rid in ids:
        chat(args, "hello", rid)
    time.sleep(1.0)
    r = routed(args.log_file, set(ids))
    seq = [r[i][0] for i in ids]
    counts = collections.Counter(seq)
    assert max(counts.values()) - min(counts.values()) <= 1, counts
    assert all(a != b for a, b in zip(seq, seq[1:])) or len(counts) == 1, "two consecutive requests hit one endpoint"
    print(f"roundrobin: {dict(counts)}")


def check_prefixaware(args):
    # conversations that grow: every turn extends the previous prompt, so it must stay on one endpoint
    convs = {c: (f"conversation {c} " + "lorem ipsum dolor sit amet " * 40) for c in range(5)}
    ids = {}
    for turn in range(4):
        for c in convs:
            convs[c] += f" follow-up {turn} " + "consectetur " * 20
            rid = uuid.uuid4().hex
            ids[rid] = c
            chat(args, convs[c], rid)
    time.sleep(1.0)
    r = routed(args.log_file, set(ids))
    per_conv = collections.defaultdict(set)
    for rid, urls in r.items():
        per_conv[ids[rid]].add(urls[0])
    assert all(len(v) == 1 for v in per_conv.values()), dict(per_conv)
    print(f"prefixaware: {len(per_conv)} growing conversations each stayed on one endpoint; "
          f"{len(set().union(*per_conv.values()))} endpoints used")


def check_disagg(args):
    ids = [uuid.uuid4().hex for _ in range(8)]
    for rid in ids:
        body = {"model": args.model, "prompt": "Explain KV cache transfer. " * 10, "max_tokens": 8}
        st, _, _ = post(args.router_url + "/v1/completions", body, {"X-Request-Id": rid})
        assert st == 200
    time.sleep(1.0)
    r = routed(args.log_file, set(ids))
    for rid in ids:
        assert len(r[rid]) == 2 and r[rid][0] != r[rid][1], (rid, r[rid])
    print("disaggregated_prefill: every request was routed twice, to two different endpoints (prefill, decode)")


def check_kvaware(args):
    prompt = "shared document " * 300
    first, second = uuid.uuid4().hex, uuid.uuid4().hex
    for rid in (first, second):
        body = {"model": args.model, "prompt": prompt, "max_tokens": 4}
        st, _, _ = post(args.router_url + "/v1/completions", body, {"X-Request-Id": rid})
        assert st == 200
        time.sleep(2.0)  # engines report admitted chunks to the controller asynchronously
    r = routed(args.log_file, {first, second})
    assert r[first] and r[second] and r[first][0] == r[second][0], r
    print("kvaware: the repeat of a long prompt went to the replica that holds its KV")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--router-url", default="http://localhost:30080")
    ap.add_argument("--log-file", required=True)
    ap.add_argument("--routing-logic", required=True)
    ap.add_argument("--model", default="fake_model_name")
    ap.add_argument("--session-key", default="x-user-id")
    a = ap.parse_args()
    {"session": check_session, "roundrobin": check_roundrobin, "prefixaware": check_prefixaware,
     "disaggregated_prefill": check_disagg, "kvaware": check_kvaware}[a.routing_logic](a)
    print("PASS")


if __name__ == "__main__":
    try:
        main()
    except AssertionError as e:
        print("FAIL:", e, file=sys.stderr)
        sys.exit(1)
