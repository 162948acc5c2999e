"""Minimal audio front end for the transcription endpoint: WAV validation + energy-based
voice activity.  A replica started with ``--task transcription`` but without speech-model
weights can only report that (no) speech was detected; it keeps the endpoint contract
(multipart upload in, JSON/text out) so routing and health probes behave as in production."""

from __future__ import annotations

import io
import struct
import wave


def transcribe_pcm(data: bytes, language: str | None = None) -> str:
    try:
        with wave.open(io.BytesIO(data)) as w:
            n = w.getnframes()
            width = w.getsampwidth()
            frames = w.readframes(n)
    except (wave.Error, EOFError) as e:
        raise ValueError(f"unsupported audio payload: {e}") from e
    if width != 2 or not frames:
        return ""
    samples = struct.unpack(f"<{len(frames) // 2}h", frames[: len(frames) // 2 * 2])
    energy = sum(abs(s) for s in samples) / max(1, len(samples))
    return "" if energy < 200 else "[speech]"
