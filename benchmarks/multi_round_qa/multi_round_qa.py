"""Multi-round QA load generator (the headline benchmark of the stack).

Same workload definition, flags and CSV columns as
/root/reference/benchmarks/multi-round-qa/multi-round-qa.py (:130-149 request settings,
:214-250 prompt construction, :285-326 session pacing, :328-338 CSV columns, :436-514
summary): every user owns a conversation whose first message carries a shared system
prompt plus a long user-specific history; each round appends a question and the previous
answer; requests are streamed chat completions with ``temperature=0``,
``stream_options.include_usage`` and header ``x-user-id``.  TTFT = time to the first
non-empty content delta.

Two drivers:
  * open loop (reference behaviour): ``--qps`` paces session turns; runs for ``--time`` s.
  * closed loop (``run_rounds``): all users issue round r together, round r+1 starts when
    every answer of round r is complete - used by ``bench.py`` to time exactly K rounds.

Implemented on aiohttp (no openai package); the summary adds p50/p90/p99 TTFT to the
reference's mean.
"""

from __future__ import annotations

import argparse
import asyncio
import csv
import json
import statistics
import sys
import time
from dataclasses import dataclass, field

import aiohttp


@dataclass
class Response:
    body: str
    ttft: float
    generation_time: float
    prompt_tokens: int
    generation_tokens: int
    launch_time: float
    finish_time: float
    status: int = 200


@dataclass
class WorkloadConfig:
    num_users: int
    system_prompt_len: int
    user_info_len: int
    answer_len: int
    num_rounds: int
    qps: float
    model: str
    enable_user_id: bool = True


def dummy_text(n: int) -> str:
    return " ".join(["hi"] * n)


def system_prompt(sys_len: int, user_id: int, user_len: int) -> str:
    return (f"Hi, here's some system prompt: {dummy_text(sys_len)}."
            f"For user {user_id}, here are some other context: {dummy_text(user_len)}.")


class RequestExecutor:
    def __init__(self, base_url: str, api_key: str, model: str):
        self.base_url = base_url.rstrip("/")
        if not self.base_url.endswith("/v1"):
            self.base_url += "/v1"
        self.api_key = api_key
        self.model = model
        self.session: aiohttp.ClientSession | None = None

    async def start(self) -> None:
        self.session = aiohttp.ClientSession(
            connector=aiohttp.TCPConnector(limit=0), timeout=aiohttp.ClientTimeout(total=None),
            read_bufsize=1 << 20)

    async def close(self) -> None:
        if self.session:
            await self.session.close()

    async def chat(self, messages: list[dict], max_tokens: int, extra_headers: dict | None = None) -> Response:
        headers = {"Authorization": f"Bearer {self.api_key}", **(extra_headers or {})}
        body = {"model": self.model, "messages": messages, "temperature": 0, "stream": True,
                "max_tokens": max_tokens, "stream_options": {"include_usage": True}}
        start = time.time()
        first = None
        parts: list[str] = []
        usage = None
        async with self.session.post(self.base_url + "/chat/completions", json=body,
                                     headers=headers) as r:
            status = r.status
            if status != 200:
                text = await r.text()
                return Response(text, -1, -1, 0, 0, start, time.time(), status)
            buf = b""
            async for chunk in r.content.iter_any():
                buf += chunk
                while b"\n\n" in buf:
                    event, buf = buf.split(b"\n\n", 1)
                    if not event.startswith(b"data: "):
                        continue
                    payload = event[6:]
                    if payload == b"[DONE]":
                        continue
                    d = json.loads(payload)
                    if d.get("usage"):
                        usage = d["usage"]
                    for c in d.get("choices") or []:
                        t = (c.get("delta") or {}).get("content")
                        if t:
                            if first is None:
                                first = time.time()
                            parts.append(t)
        end = time.time()
        if first is None:
            first = end
        usage = usage or {}
        return Response("".join(parts), first - start, end - first, usage.get("prompt_tokens", 0),
                        usage.get("completion_tokens", 0), start, end, status)


@dataclass
class UserSession:
    user_id: int
    cfg: WorkloadConfig
    history: list[dict] = field(default_factory=list)
    question_id: int = 0
    results: list[Response] = field(default_factory=list)
    next_launch: float = 0.0
    busy: bool = False
    finished: bool = False

    def next_messages(self) -> list[dict]:
        self.question_id += 1
        q = (f"Here's question #{self.question_id}: can you tell me a new long story with a "
             "happy ending?")
        if not self.history:
            q = system_prompt(self.cfg.system_prompt_len, self.user_id, self.cfg.user_info_len) + q
        self.history.append({"role": "user", "content": q})
        return list(self.history)

    def on_answer(self, resp: Response) -> None:
        self.history.append({"role": "assistant", "content": resp.body})
        self.results.append(resp)
        if self.question_id >= self.cfg.num_rounds:
            self.finished = True

    def headers(self) -> dict:
        return {"x-user-id": str(self.user_id)} if self.cfg.enable_user_id else {}


CSV_COLUMNS = ["prompt_tokens", "generation_tokens", "ttft", "generation_time", "user_id",
               "question_id", "launch_time", "finish_time"]


def rows_of(sessions: list[UserSession]) -> list[dict]:
    rows = []
    for s in sessions:
        for qi, r in enumerate(s.results):
            rows.append({"prompt_tokens": r.prompt_tokens, "generation_tokens": r.generation_tokens,
                         "ttft": r.ttft, "generation_time": r.generation_time, "user_id": s.user_id,
                         "question_id": qi + 1, "launch_time": r.launch_time,
                         "finish_time": r.finish_time})
    return rows


def write_csv(path: str, rows: list[dict]) -> None:
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=CSV_COLUMNS)
        w.writeheader()
        w.writerows(rows)


def pct(xs: list[float], p: float) -> float:
    if not xs:
        return float("nan")
    xs = sorted(xs)
    k = (len(xs) - 1) * p
    lo, hi = int(k), min(int(k) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


def summarize(rows: list[dict], start: float | None = None, end: float | None = None,
              pending: int = 0, qps: float | None = None, out=sys.stdout) -> dict:
    ok = [r for r in rows if r["ttft"] >= 0]
    if not ok:
        print("no finished requests", file=out)
        return {}
    start = start if start is not None else min(r["launch_time"] for r in ok)
    end = end if end is not None else max(r["finish_time"] for r in ok)
    dur = max(end - start, 1e-9)
    ttfts = [r["ttft"] for r in ok]
    gen_speed = [r["generation_tokens"] / r["generation_time"] for r in ok if r["generation_time"] > 0]
    s = {
        "duration_s": dur, "requests": len(ok), "pending": pending,
        "finished_qps": len(ok) / dur,
        "input_tokens_per_s": sum(r["prompt_tokens"] for r in ok) / dur,
        "output_tokens_per_s": sum(r["generation_tokens"] for r in ok) / dur,
        "avg_generation_speed_per_request": statistics.fmean(gen_speed) if gen_speed else 0.0,
        "mean_ttft_s": statistics.fmean(ttfts), "p50_ttft_s": pct(ttfts, 0.5),
        "p90_ttft_s": pct(ttfts, 0.9), "p99_ttft_s": pct(ttfts, 0.99),
    }
    print("\n==================== Performance summary ======================", file=out)
    if qps is not None:
        print(f"  QPS: {qps:.4f} reqs/s", file=out)
    print(f"  Processing speed: {s['finished_qps']:.4f} reqs/s", file=out)
    print(f"  Requests on-the-fly: {pending}", file=out)
    print(f"  Input tokens per second: {s['input_tokens_per_s']:.4f} tokens/s", file=out)
    print(f"  Output tokens per second: {s['output_tokens_per_s']:.4f} tokens/s", file=out)
    print(f"  Average generation throughput (per request): "
          f"{s['avg_generation_speed_per_request']:.4f} tokens/req/s", file=out)
    print(f"  Average TTFT: {s['mean_ttft_s']:.4f}s", file=out)
    print(f"  P50 / P90 / P99 TTFT: {s['p50_ttft_s']:.4f}s / {s['p90_ttft_s']:.4f}s / "
          f"{s['p99_ttft_s']:.4f}s", file=out)
    print(f"Time range: {start} - {end} ({dur:.2f}s)", file=out)
    print("===============================================================\n", file=out)
    return s


# ------------------------------------------------------------------ closed loop (bench.py)
async def run_rounds(base_url: str, model: str, user_ids: list[int], rounds: int, sys_len: int,
                     hist_len: int, answer_len: int, api_key: str = "EMPTY",
                     on_round=None, sessions: list[UserSession] | None = None,
                     stagger_s: float = 0.0) -> list[UserSession]:
    """Every user issues one request per round; a round ends when all answers are complete."""
    cfg = WorkloadConfig(len(user_ids), sys_len, hist_len, answer_len, 10 ** 9, 0.0, model)
    ex = RequestExecutor(base_url, api_key, model)
    await ex.start()
    sessions = sessions or [UserSession(u, cfg) for u in user_ids]
    try:
        for rnd in range(rounds):
            t0 = time.time()

            async def one(i: int, s: UserSession):
                if stagger_s:
                    await asyncio.sleep(stagger_s * i)
                r = await ex.chat(s.next_messages(), answer_len, s.headers())
                s.on_answer(r)

            await asyncio.gather(*[one(i, s) for i, s in enumerate(sessions)])
            if on_round:
                on_round(rnd, time.time() - t0, sessions)
    finally:
        await ex.close()
    return sessions


# ------------------------------------------------------------------ open loop (reference CLI)
async def run_open_loop(args) -> list[UserSession]:
    cfg = WorkloadConfig(args.num_users, args.shared_system_prompt, args.user_history_prompt,
                         args.answer_len, args.num_rounds, args.qps, args.model,
                         args.request_with_user_id)
    ex = RequestExecutor(args.base_url, args.api_key, args.model)
    await ex.start()
    # engine warm-up: ten short prompts
    await asyncio.gather(*[ex.chat([{"role": "user", "content": dummy_text(100)}], args.answer_len)
                           for _ in range(10)])
    gap_between_requests_per_user = cfg.num_users / cfg.qps
    session_len = gap_between_requests_per_user * cfg.num_rounds
    gap_between_users = session_len / cfg.num_users
    start = time.time()
    sessions: list[UserSession] = []
    all_sessions: list[UserSession] = []
    next_uid = args.init_user_id
    tasks: set[asyncio.Task] = set()

    def launch(s: UserSession):
        s.busy = True

        async def go():
            try:
                r = await ex.chat(s.next_messages(), cfg.answer_len, s.headers())
                if r.status != 200:
                    s.history.pop()
                    s.question_id -= 1
                else:
                    s.on_answer(r)
            finally:
                s.busy = False

        t = asyncio.create_task(go())
        tasks.add(t)
        t.add_done_callback(tasks.discard)

    # ramp-up: pre-age sessions so the system starts in steady state
    for i in range(cfg.num_users):
        s = UserSession(next_uid, cfg)
        next_uid += 1
        offset = i * gap_between_users
        s.question_id = min(int(offset / gap_between_requests_per_user), cfg.num_rounds - 1)
        if s.question_id:
            # an aged session already holds (synthetic) earlier turns
            s.history.append({"role": "user", "content": system_prompt(
                cfg.system_prompt_len, s.user_id, cfg.user_info_len) + "Earlier questions."})
            s.history.append({"role": "assistant", "content": dummy_text(cfg.answer_len * s.question_id)})
        s.next_launch = start + (offset % gap_between_requests_per_user)
        sessions.append(s)
        all_sessions.append(s)
    last_join = start
    last_log = start
    try:
        while True:
            now = time.time()
            if args.time and now - start > args.time:
                break
            if now - last_join >= gap_between_users:
                last_join = now
                s = UserSession(next_uid, cfg)
                next_uid += 1
                s.next_launch = now
                sessions.append(s)
                all_sessions.append(s)
            for s in sessions:
                if not s.busy and not s.finished and now >= s.next_launch:
                    s.next_launch = now + gap_between_requests_per_user
                    launch(s)
            sessions = [s for s in sessions if not (s.finished and not s.busy)]
            if not args.time and not sessions:
                break
            if now - last_log >= args.log_interval:
                last_log = now
                summarize(rows_of(all_sessions), start, now, len(tasks), cfg.qps)
            await asyncio.sleep(0.05)
    finally:
        for t in list(tasks):
            t.cancel()
        await ex.close()
    end = time.time()
    rows = rows_of(all_sessions)
    summarize(rows, start, end, 0, cfg.qps)
    write_csv(args.output, rows)
    return all_sessions


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Multi-round QA benchmark")
    p.add_argument("--num-users", type=int, required=True)
    p.add_argument("--shared-system-prompt", type=int, required=True)
    p.add_argument("--user-history-prompt", type=int, required=True)
    p.add_argument("--answer-len", type=int, required=True)
    p.add_argument("--num-rounds", type=int, required=True)
    p.add_argument("--qps", type=float, required=True)
    p.add_argument("--model", type=str, required=True)
    p.add_argument("--api-key", type=str, default="EMPTY")
    p.add_argument("--base-url", type=str, required=True)
    p.add_argument("--time", type=int, default=None)
    p.add_argument("--output", type=str, default="summary.csv")
    p.add_argument("--init-user-id", type=int, default=0)
    p.add_argument("--request-with-user-id", action="store_true")
    p.add_argument("--log-interval", type=int, default=30)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--sharegpt", action="store_true")
    return p


def main() -> None:
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--process-summary", default=None)
    a, _ = pre.parse_known_args()
    if a.process_summary:
        with open(a.process_summary) as f:
            rows = [{k: float(v) for k, v in r.items()} for r in csv.DictReader(f)]
        summarize(rows)
        return
    args = build_parser().parse_args()
    if args.sharegpt:
        sys.exit("--sharegpt needs the ShareGPT dataset file, which is not bundled; "
                 "use the synthetic workload")
    asyncio.run(run_open_loop(args))


if __name__ == "__main__":
    main()
