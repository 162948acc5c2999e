"""Speech-to-text replica: a Whisper-class model behind the engine HTTP contract.

``vpsa-engine serve openai/whisper-small --task transcription`` starts this server instead
of the text engine.  It answers the routes the router and the operator expect from any
replica (``/health``, ``/metrics``, ``/v1/models``, ``/version``, sleep mode) plus
``POST /v1/audio/transcriptions`` (multipart: file, model, language, prompt,
response_format, temperature), the route the reference's router proxies at
/root/reference/src/vllm_router/routers/main_router.py:239-246.

Concurrent uploads are cut into 30 s windows and batched: one worker thread drains the
window queue, runs encoder + greedy decoder for up to ``max_num_seqs`` windows at a time
and resolves the per-request futures.
"""

from __future__ import annotations

import asyncio
import logging
import queue
import threading
import time
import uuid
from dataclasses import dataclass, field

import numpy as np
import torch
from aiohttp import web

from .. import __version__
from ..models.whisper import (N_SAMPLES, SAMPLE_RATE, WhisperModel, pad_or_trim, resolve_whisper)
from .audio import AudioError, load_audio
from .config import EngineConfig
from .metrics import EngineMetrics

log = logging.getLogger("vpsa.speech")

# language codes in the order of the model's language tokens
LANGUAGES = (
    "en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur "
    "hr bg lt la mi ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si "
    "km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl mg as tt haw ln ha "
    "ba jw su yue").split()
_OPEN_PATHS = {"/health", "/ping", "/metrics", "/version"}


class SpeechTokenizer:
    """Token ids -> text.  A checkpoint directory brings the real vocabulary; without one
    (random weights) ids are rendered as placeholder words so the response shape holds."""

    def __init__(self, ckpt_dir, eos_token_id: int):
        self.hf = None
        self.eos = eos_token_id
        if ckpt_dir is not None:
            try:
                from transformers import AutoTokenizer

                self.hf = AutoTokenizer.from_pretrained(str(ckpt_dir), local_files_only=True)
            except Exception as e:  # noqa: BLE001 - any tokenizer problem degrades to placeholders
                log.warning("no usable tokenizer in %s (%s); rendering token ids", ckpt_dir, e)

    def decode(self, ids: list[int]) -> str:
        if self.hf is not None:
            return self.hf.decode(ids, skip_special_tokens=True).strip()
        return " ".join(f"w{t}" for t in ids)

    def encode_prompt(self, text: str) -> list[int]:
        if self.hf is not None and text:
            return self.hf.encode(" " + text.strip(), add_special_tokens=False)
        return []


@dataclass
class _Window:
    samples: np.ndarray
    prompt: list[int]
    temperature: float
    done: threading.Event = field(default_factory=threading.Event)
    tokens: list[int] | None = None
    error: Exception | None = None


class SpeechEngine:
    def __init__(self, cfg: EngineConfig, device: torch.device | None = None):
        self.cfg = cfg
        self.model_cfg, self.ckpt_dir = resolve_whisper(cfg.model)
        self.device = device or cfg.resolve_device()
        dtype = cfg.torch_dtype() if self.device.type == "cuda" else torch.float32
        self.model = WhisperModel(self.model_cfg, dtype, self.device)
        t0 = time.time()
        if self.ckpt_dir is not None and cfg.load_format != "dummy":
            self.model.load_hf(self.ckpt_dir)
        else:
            self.model.init_random(cfg.seed)
        log.info("speech model %s ready on %s: %.2f GiB weights in %.1f s", cfg.model, self.device,
                 self.model.weight_bytes() / 2**30, time.time() - t0)
        self.tokenizer = SpeechTokenizer(self.ckpt_dir, self.model_cfg.eos_token_id)
        self.metrics = EngineMetrics(cfg.model_name)
        self.max_model_len = self.model_cfg.max_target_positions
        self.max_batch = max(1, min(cfg.max_num_seqs, 32))
        self.model.graph_max_batch = self.max_batch
        if cfg.enforce_eager:
            self.model.use_graphs = False
        self._q: "queue.Queue[_Window | None]" = queue.Queue()
        self._pending = 0
        self._lock = threading.Lock()
        self._sleeping = False
        self._host_weights: dict[str, torch.Tensor] | None = None
        self.healthy = True
        self._thread = threading.Thread(target=self._loop, name="speech-worker", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ worker
    def _loop(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)  # per-thread setting; a new thread starts on GPU 0
            try:
                t0 = time.time()
                n = self.model.capture_graphs()
                if n:
                    log.info("decode step captured for %d batch sizes in %.1f s", n, time.time() - t0)
            except Exception:  # noqa: BLE001 - the eager loop serves
                log.exception("graph capture failed; decoding eagerly")
                self.model.use_graphs = False
                self.model.drop_graphs()
        while True:
            first = self._q.get()
            if first is None:
                return
            batch = [first]
            deadline = time.time() + 0.005
            while len(batch) < self.max_batch:
                try:
                    nxt = self._q.get(timeout=max(0.0, deadline - time.time()))
                except queue.Empty:
                    break
                if nxt is None:
                    self._q.put(None)
                    break
                batch.append(nxt)
            self.metrics.running.set(len(batch))
            self.metrics.waiting.set(self._q.qsize())
            try:
                # one temperature per launch: group by it (greedy requests share a launch)
                groups: dict[float, list[_Window]] = {}
                for wnd in batch:
                    groups.setdefault(wnd.temperature, []).append(wnd)
                for temp, group in groups.items():
                    # prompts of one launch must have one length
                    by_len: dict[int, list[_Window]] = {}
                    for wnd in group:
                        by_len.setdefault(len(wnd.prompt), []).append(wnd)
                    for same in by_len.values():
                        x = torch.from_numpy(np.stack([w.samples for w in same]))
                        toks = self.model.transcribe_windows(x, [w.prompt for w in same], temperature=temp)
                        for w, t in zip(same, toks):
                            w.tokens = t
            except Exception as e:  # noqa: BLE001 - reported to every waiting request
                log.exception("transcription batch failed")
                if isinstance(e, torch.OutOfMemoryError):
                    self.healthy = False
                for w in batch:
                    if w.tokens is None:
                        w.error = e
            finally:
                for w in batch:
                    w.done.set()
                self.metrics.running.set(0)
                self.metrics.waiting.set(self._q.qsize())

    # ------------------------------------------------------------------ API
    def language_token(self, language: str | None) -> int | None:
        if not language:
            return None
        code = language.strip().lower()
        if code not in LANGUAGES:
            raise ValueError(f"unsupported language {language!r}")
        idx = LANGUAGES.index(code)
        c = self.model_cfg
        n_lang = (c.suppress_hi or c.vocab_size) - c.first_language_token_id
        if c.suppress_hi and idx >= n_lang:
            idx = 0
        return c.first_language_token_id + idx

    def transcribe(self, samples: np.ndarray, language: str | None = None, task: str = "transcribe",
                   temperature: float = 0.0) -> list[list[int]]:
        """Blocking: token ids per 30 s window of the waveform (16 kHz mono float32)."""
        if self._sleeping:
            raise RuntimeError("the replica is sleeping")
        prompt = self.model.prompt_ids(self.language_token(language), task)
        windows = [_Window(w, prompt, float(temperature)) for w in pad_or_trim(samples)]
        for w in windows:
            self._q.put(w)
        self.metrics.waiting.set(self._q.qsize())
        out = []
        for w in windows:
            w.done.wait()
            if w.error is not None:
                raise w.error
            out.append(w.tokens or [])
        return out

    # ------------------------------------------------------------------ sleep mode
    def sleep(self, level: int = 1) -> int:
        if self._sleeping:
            return 0
        freed = self.model.weight_bytes() if self.device.type == "cuda" else 0
        if self.device.type == "cuda":
            self.model.drop_graphs()  # captured steps hold the weights' addresses
            self._host_weights = {k: v.to("cpu") for k, v in self.model.w.items()} if level == 1 else None
            self.model.w = {}
            torch.cuda.empty_cache()
        self._sleeping = True
        return freed

    def wake_up(self) -> None:
        if not self._sleeping:
            return
        if self.device.type == "cuda":
            if self._host_weights is not None:
                self.model.w = {k: v.to(self.device) for k, v in self._host_weights.items()}
                self._host_weights = None
            elif self.ckpt_dir is not None:
                self.model.load_hf(self.ckpt_dir)
            else:
                self.model.init_random(self.cfg.seed)
        self._sleeping = False

    @property
    def is_sleeping(self) -> bool:
        return self._sleeping

    def shutdown(self) -> None:
        self._q.put(None)
        self._thread.join(timeout=5)


def _err(status: int, message: str, etype: str = "BadRequestError") -> web.Response:
    return web.json_response({"object": "error", "message": message, "type": etype, "param": None,
                              "code": status}, status=status)


def _stamp(seconds: float, sep: str) -> str:
    ms = int(round(seconds * 1000))
    h, ms = divmod(ms, 3600_000)
    m, ms = divmod(ms, 60_000)
    s, ms = divmod(ms, 1000)
    return f"{h:02d}:{m:02d}:{s:02d}{sep}{ms:03d}"


class SpeechServer:
    def __init__(self, engine: SpeechEngine):
        self.engine = engine
        self.cfg = engine.cfg
        self.created = int(time.time())
        self.app = web.Application(client_max_size=512 * 1024 * 1024, middlewares=[self._auth])
        r = self.app.router
        r.add_get("/health", self.health)
        r.add_get("/ping", self.health)
        r.add_get("/version", self.version)
        r.add_get("/metrics", self.metrics)
        r.add_get("/v1/models", self.models)
        r.add_post("/v1/audio/transcriptions", self.transcriptions)
        r.add_post("/v1/audio/translations", self.translations)
        r.add_post("/sleep", self.sleep)
        r.add_post("/wake_up", self.wake_up)
        r.add_get("/is_sleeping", self.is_sleeping)
        for path in ("/v1/chat/completions", "/v1/completions", "/v1/embeddings"):
            r.add_post(path, self.not_served)

    @web.middleware
    async def _auth(self, request: web.Request, handler):
        key = self.cfg.api_key
        if key and request.path not in _OPEN_PATHS:
            if request.headers.get("Authorization", "") != f"Bearer {key}":
                return web.json_response({"error": "Unauthorized"}, status=401)
        return await handler(request)

    async def health(self, request):
        return web.Response(status=200 if self.engine.healthy else 503)

    async def version(self, request):
        return web.json_response({"version": __version__})

    async def metrics(self, request):
        return web.Response(body=self.engine.metrics.render(),
                            headers={"Content-Type": "text/plain; version=0.0.4; charset=utf-8"})

    async def models(self, request):
        card = {"id": self.cfg.model_name, "object": "model", "created": self.created,
                "owned_by": "vllm", "root": self.cfg.model, "parent": None,
                "max_model_len": self.engine.max_model_len}
        return web.json_response({"object": "list", "data": [card]})

    async def not_served(self, request):
        return _err(400, f"{self.cfg.model_name} is a speech-to-text model; use /v1/audio/transcriptions")

    async def sleep(self, request):
        if not self.cfg.enable_sleep_mode:
            return _err(400, "sleep mode is not enabled (--enable-sleep-mode)")
        level = int(request.query.get("level", "1"))
        freed = await asyncio.get_running_loop().run_in_executor(None, self.engine.sleep, level)
        return web.json_response({"status": "sleeping", "freed_bytes": freed})

    async def wake_up(self, request):
        if not self.cfg.enable_sleep_mode:
            return _err(400, "sleep mode is not enabled (--enable-sleep-mode)")
        await asyncio.get_running_loop().run_in_executor(None, self.engine.wake_up)
        return web.json_response({"status": "awake"})

    async def is_sleeping(self, request):
        return web.json_response({"is_sleeping": self.engine.is_sleeping})

    async def transcriptions(self, request):
        return await self._speech(request, "transcribe")

    async def translations(self, request):
        return await self._speech(request, "translate")

    async def _speech(self, request: web.Request, task: str):
        if not (request.content_type or "").startswith("multipart/"):
            return _err(400, "expected multipart/form-data with a 'file' part")
        fields: dict[str, str] = {}
        audio = b""
        reader = await request.multipart()
        async for part in reader:
            if part.name == "file":
                audio = bytes(await part.read(decode=False))
            else:
                fields[part.name] = (await part.read(decode=False)).decode(errors="replace")
        if not audio:
            return _err(400, "missing audio file")
        model = fields.get("model") or self.cfg.model_name
        if model not in (self.cfg.model_name, self.cfg.model):
            return _err(404, f"The model `{model}` does not exist.", "NotFoundError")
        if self.engine.is_sleeping:
            return _err(503, "the replica is sleeping", "ServiceUnavailable")
        try:
            temperature = float(fields.get("temperature") or 0.0)
            samples = load_audio(audio)
            self.engine.language_token(fields.get("language"))
        except (AudioError, ValueError) as e:
            return _err(400, str(e))
        t0 = time.time()
        loop = asyncio.get_running_loop()
        try:
            per_window = await loop.run_in_executor(
                None, lambda: self.engine.transcribe(samples, fields.get("language"), task, temperature))
        except RuntimeError as e:
            return _err(500, str(e), "InternalServerError")
        m = self.engine.metrics
        n_tokens = sum(len(t) for t in per_window)
        m.generation_tokens.inc(n_tokens)
        m.success.inc()
        m.e2e.observe(time.time() - t0)
        texts = [self.engine.tokenizer.decode(t) for t in per_window]
        text = " ".join(t for t in texts if t).strip()
        duration = samples.size / SAMPLE_RATE
        fmt = (fields.get("response_format") or "json").lower()
        if fmt == "text":
            return web.Response(text=text + "\n")
        spans = []
        for i, (t, ids) in enumerate(zip(texts, per_window)):
            a = i * N_SAMPLES / SAMPLE_RATE
            b = min(duration, (i + 1) * N_SAMPLES / SAMPLE_RATE)
            spans.append({"id": i, "seek": i * 3000, "start": round(a, 3), "end": round(b, 3),
                          "text": t, "tokens": ids, "temperature": temperature})
        if fmt in ("srt", "vtt"):
            sep = "," if fmt == "srt" else "."
            lines = ["WEBVTT", ""] if fmt == "vtt" else []
            for s in spans:
                if fmt == "srt":
                    lines.append(str(s["id"] + 1))
                lines += [f"{_stamp(s['start'], sep)} --> {_stamp(s['end'], sep)}", s["text"], ""]
            return web.Response(text="\n".join(lines))
        if fmt == "verbose_json":
            return web.json_response({"task": task, "language": fields.get("language") or "en",
                                      "duration": round(duration, 3), "text": text, "segments": spans,
                                      "usage": {"type": "duration", "seconds": int(round(duration))},
                                      "id": "transcribe-" + uuid.uuid4().hex})
        if fmt != "json":
            return _err(400, f"unsupported response_format {fmt!r}")
        return web.json_response({"text": text,
                                  "usage": {"type": "duration", "seconds": int(round(duration))}})


async def serve_speech(engine: SpeechEngine, host: str, port: int) -> web.AppRunner:
    server = SpeechServer(engine)
    runner = web.AppRunner(server.app, access_log=None)
    await runner.setup()
    site = web.TCPSite(runner, host, port, reuse_address=True, backlog=1024)
    await site.start()
    log.info("speech replica serving on http://%s:%d", host, port)
    return runner
