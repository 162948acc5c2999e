"""Audio payload decoding for the transcription endpoint: RIFF/WAV (8/16/24/32-bit PCM and
32-bit float, any channel count, any sample rate) -> mono float32 at 16 kHz.

Compressed containers (mp3, flac, ogg, m4a) need a codec library that this image does not
ship; they are rejected with a message that says so instead of being mis-read.
The router's health probe for speech backends uploads a 0.1 s silent WAV
(/root/reference/src/vllm_router/utils.py:18-33), which this module accepts.
"""

from __future__ import annotations

import struct
from math import gcd

import numpy as np

TARGET_RATE = 16000


class AudioError(ValueError):
    pass


def _riff_chunks(data: bytes):
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        kind = "an unknown container"
        if data[:3] == b"ID3" or data[:2] in (b"\xff\xfb", b"\xff\xf3"):
            kind = "mp3"
        elif data[:4] == b"fLaC":
            kind = "flac"
        elif data[:4] == b"OggS":
            kind = "ogg"
        elif data[4:8] == b"ftyp":
            kind = "mp4/m4a"
        raise AudioError(f"unsupported audio format ({kind}); send RIFF/WAV PCM")
    pos = 12
    while pos + 8 <= len(data):
        cid = data[pos: pos + 4]
        size = struct.unpack("<I", data[pos + 4: pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        yield cid, body
        pos += 8 + size + (size & 1)


def decode_wav(data: bytes) -> tuple[np.ndarray, int]:
    """WAV bytes -> (mono float32 samples in [-1, 1], sample rate)."""
    fmt = None
    pcm = None
    for cid, body in _riff_chunks(data):
        if cid == b"fmt " and len(body) >= 16:
            tag, channels, rate, _br, _align, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, channels, rate, bits)
        elif cid == b"data":
            pcm = body
    if fmt is None or pcm is None:
        raise AudioError("WAV file lacks a fmt or data chunk")
    tag, channels, rate, bits = fmt
    if channels < 1 or rate < 1:
        raise AudioError("WAV header declares no channels or a zero sample rate")
    if tag == 1:  # integer PCM
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            raw = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4").astype(np.float32) / float(1 << 31)
        else:
            raise AudioError(f"unsupported PCM width: {bits} bits")
    elif tag == 3 and bits == 32:  # IEEE float
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(pcm[: len(pcm) // 8 * 8], dtype="<f8").astype(np.float32)
    else:
        raise AudioError(f"unsupported WAV encoding (format tag {tag}, {bits} bits)")
    if channels > 1:
        x = x[: x.size // channels * channels].reshape(-1, channels).mean(axis=1)
    return np.ascontiguousarray(x, dtype=np.float32), rate


def resample(x: np.ndarray, rate: int, target: int = TARGET_RATE) -> np.ndarray:
    if rate == target or x.size == 0:
        return x.astype(np.float32, copy=False)
    from scipy.signal import resample_poly

    g = gcd(rate, target)
    return resample_poly(x, target // g, rate // g).astype(np.float32)


def load_audio(data: bytes) -> np.ndarray:
    """Any supported payload -> mono float32 at 16 kHz."""
    if not data:
        raise AudioError("empty audio payload")
    x, rate = decode_wav(data)
    return resample(x, rate)


def silent_wav(seconds: float = 0.1, rate: int = TARGET_RATE) -> bytes:
    """A valid 16-bit mono WAV of silence (health probes, tests)."""
    n = int(seconds * rate)
    body = b"\x00\x00" * n
    return (b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" +
            struct.pack("<I", len(body)) + body)


def pcm16_wav(samples: np.ndarray, rate: int = TARGET_RATE) -> bytes:
    body = (np.clip(samples, -1.0, 1.0) * 32767.0).astype("<i2").tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" +
            struct.pack("<I", len(body)) + body)
