"""Whisper-class speech-to-text model (audio encoder + text decoder with cross-attention).

Serves ``POST /v1/audio/transcriptions`` for replicas started with a speech model
(``--task transcription``; the reference proxies that route to such a replica:
/root/reference/src/vllm_router/services/request_service/request.py:513-689, and its
tutorial deploys ``openai/whisper-small``: /root/reference/tutorials/23-whisper-api-transcription.md:20-30).

Shape of the work: the encoder sees a fixed 30 s window (3000 mel frames -> 1500 positions),
so a batch of windows is one dense [B, 1500, d] problem; the decoder emits at most 448 tokens
against a cross-attention K/V that is computed once per window.  Everything is dense
(no paged pool): the weights are <= 3 GiB, a window's cross K/V is <= 1500 x d per layer, and
the matmuls go to the library GEMM.  Attention is written out as two batched GEMMs and a
softmax so that no hidden JIT kernel is involved.

On a GPU the greedy decode loop is a captured HIP graph: one replay per token over static
self-/cross-attention K/V buffers, the position and the running token kept on the device, and
the host looking at the end-of-text flags only every few steps.  An eager decode step of a
12-layer model is about 400 launches of a few microseconds of work each; the replay removes the
launch gaps and the per-token device-to-host round trip.

Weights follow the Hugging Face layout (``model.encoder.*`` / ``model.decoder.*``) when a
checkpoint directory is given; otherwise they are drawn from a seeded generator.
"""

from __future__ import annotations

import json
import math
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
CHUNK_SECONDS = 30
N_SAMPLES = SAMPLE_RATE * CHUNK_SECONDS
N_FRAMES = N_SAMPLES // HOP  # 3000


@dataclass
class WhisperConfig:
    name: str = "tiny-whisper"
    vocab_size: int = 51865
    num_mel_bins: int = 80
    d_model: int = 384
    encoder_layers: int = 4
    decoder_layers: int = 4
    encoder_heads: int = 6
    decoder_heads: int = 6
    encoder_ffn_dim: int = 1536
    decoder_ffn_dim: int = 1536
    max_source_positions: int = 1500
    max_target_positions: int = 448
    eos_token_id: int = 50257
    sot_token_id: int = 50258  # <|startoftranscript|>
    transcribe_token_id: int = 50359
    translate_token_id: int = 50358
    no_timestamps_token_id: int = 50363
    first_language_token_id: int = 50259  # <|en|>
    # ids in [suppress_lo, suppress_hi) are control tokens (start, languages, tasks, timestamps)
    # and are never emitted in no-timestamps transcription; 0 = (eos+1, vocab_size)
    suppress_lo: int = 0
    suppress_hi: int = 0
    architecture: str = "whisper"

    def num_params(self) -> int:
        d = self.d_model
        enc = self.encoder_layers * (4 * d * d + 2 * d * self.encoder_ffn_dim)
        dec = self.decoder_layers * (8 * d * d + 2 * d * self.decoder_ffn_dim)
        return enc + dec + self.vocab_size * d


def _multilingual(name, d, layers, heads, vocab=51865, mels=80, **kw) -> WhisperConfig:
    return WhisperConfig(name=name, vocab_size=vocab, num_mel_bins=mels, d_model=d,
                         encoder_layers=layers, decoder_layers=layers, encoder_heads=heads,
                         decoder_heads=heads, encoder_ffn_dim=4 * d, decoder_ffn_dim=4 * d, **kw)


WHISPER_PRESETS: dict[str, WhisperConfig] = {
    "whisper-tiny": _multilingual("whisper-tiny", 384, 4, 6),
    "whisper-base": _multilingual("whisper-base", 512, 6, 8),
    "whisper-small": _multilingual("whisper-small", 768, 12, 12),
    "whisper-medium": _multilingual("whisper-medium", 1024, 24, 16),
    "whisper-large-v2": _multilingual("whisper-large-v2", 1280, 32, 20),
    # large-v3: 128 mel bins and one more language token, so the task tokens move up by one
    "whisper-large-v3": _multilingual("whisper-large-v3", 1280, 32, 20, vocab=51866, mels=128,
                                      transcribe_token_id=50360, translate_token_id=50359,
                                      no_timestamps_token_id=50364),
    "tiny-whisper": WhisperConfig(name="tiny-whisper", vocab_size=600, d_model=64, encoder_layers=2,
                                  decoder_layers=2, encoder_heads=4, decoder_heads=4,
                                  encoder_ffn_dim=128, decoder_ffn_dim=128, max_target_positions=64,
                                  eos_token_id=2, sot_token_id=3, transcribe_token_id=5,
                                  translate_token_id=4, no_timestamps_token_id=6,
                                  first_language_token_id=7, suppress_lo=3, suppress_hi=16),
}


def is_whisper_name(model: str) -> bool:
    base = model.lower().rstrip("/").split("/")[-1]
    return "whisper" in base


def resolve_whisper(model: str) -> tuple[WhisperConfig, Path | None]:
    """Model id or checkpoint directory -> (config, directory or None for random weights)."""
    p = Path(model)
    if p.is_dir() and (p / "config.json").exists():
        with open(p / "config.json") as f:
            c = json.load(f)
        cfg = WhisperConfig(
            name=model, vocab_size=c["vocab_size"], num_mel_bins=c.get("num_mel_bins", 80),
            d_model=c["d_model"], encoder_layers=c["encoder_layers"],
            decoder_layers=c["decoder_layers"], encoder_heads=c["encoder_attention_heads"],
            decoder_heads=c["decoder_attention_heads"], encoder_ffn_dim=c["encoder_ffn_dim"],
            decoder_ffn_dim=c["decoder_ffn_dim"],
            max_source_positions=c.get("max_source_positions", 1500),
            max_target_positions=c.get("max_target_positions", 448),
            eos_token_id=c.get("eos_token_id", 50257),
            sot_token_id=c.get("decoder_start_token_id", 50258))
        gen = p / "generation_config.json"
        if gen.exists():
            with open(gen) as f:
                g = json.load(f)
            t2i = g.get("task_to_id") or {}
            cfg.transcribe_token_id = t2i.get("transcribe", cfg.transcribe_token_id)
            cfg.translate_token_id = t2i.get("translate", cfg.translate_token_id)
            cfg.no_timestamps_token_id = g.get("no_timestamps_token_id", cfg.no_timestamps_token_id)
            l2i = g.get("lang_to_id") or {}
            if l2i:
                cfg.first_language_token_id = min(l2i.values())
        return cfg, p
    base = model.lower().rstrip("/").split("/")[-1]
    for key in sorted(WHISPER_PRESETS, key=len, reverse=True):
        if key == base or key in base:
            c = WHISPER_PRESETS[key]
            return WhisperConfig(**{**c.__dict__, "name": model}), None
    raise ValueError(f"unknown speech model {model!r}")


# ---------------------------------------------------------------------------- audio front end
def _hz_to_mel(f: np.ndarray) -> np.ndarray:
    # Slaney scale: linear below 1 kHz, logarithmic above
    f = np.asarray(f, dtype=np.float64)
    mel = f / (200.0 / 3)
    log_region = f >= 1000.0
    mel[log_region] = 15.0 + np.log(f[log_region] / 1000.0) / (math.log(6.4) / 27.0)
    return mel


def _mel_to_hz(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    f = m * (200.0 / 3)
    log_region = m >= 15.0
    f[log_region] = 1000.0 * np.exp((math.log(6.4) / 27.0) * (m[log_region] - 15.0))
    return f


def mel_filterbank(n_mels: int, n_fft: int = N_FFT, sr: int = SAMPLE_RATE) -> np.ndarray:
    """[n_mels, n_fft/2+1] triangular filters on the Slaney mel scale, area-normalised."""
    fft_freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    mel_pts = _mel_to_hz(np.linspace(_hz_to_mel(np.array([0.0]))[0],
                                     _hz_to_mel(np.array([sr / 2.0]))[0], n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    fb = np.maximum(0.0, np.minimum(lower, upper))
    fb *= (2.0 / (mel_pts[2:] - mel_pts[:-2]))[:, None]
    return fb.astype(np.float32)


def log_mel(samples: torch.Tensor, n_mels: int, filters: torch.Tensor | None = None) -> torch.Tensor:
    """[B, N_SAMPLES] float waveform at 16 kHz -> [B, n_mels, 3000] normalised log-mel."""
    window = torch.hann_window(N_FFT, periodic=True, device=samples.device, dtype=torch.float32)
    spec = torch.stft(samples.float(), N_FFT, HOP, window=window, center=True, pad_mode="reflect",
                      return_complex=True)
    power = spec[..., :-1].abs().pow(2)
    if filters is None:
        filters = torch.from_numpy(mel_filterbank(n_mels)).to(samples.device)
    mel = filters @ power
    log = torch.clamp(mel, min=1e-10).log10()
    log = torch.maximum(log, log.amax(dim=(-2, -1), keepdim=True) - 8.0)
    return (log + 4.0) / 4.0


def pad_or_trim(samples: np.ndarray) -> list[np.ndarray]:
    """Split a waveform into 30 s windows, the last one zero-padded."""
    if samples.size == 0:
        return [np.zeros(N_SAMPLES, dtype=np.float32)]
    out = []
    for a in range(0, samples.size, N_SAMPLES):
        w = samples[a: a + N_SAMPLES].astype(np.float32)
        if w.size < N_SAMPLES:
            w = np.concatenate([w, np.zeros(N_SAMPLES - w.size, dtype=np.float32)])
        out.append(w)
    return out


def _sinusoids(length: int, channels: int) -> torch.Tensor:
    inc = math.log(10000.0) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


# ---------------------------------------------------------------------------- model
def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int,
            mask: torch.Tensor | None = None) -> torch.Tensor:
    """q [B, Tq, d], k/v [B, Tk, d] -> [B, Tq, d].  Unmasked attention of 64-channel heads in bf16
    on the GPU (the whole encoder, and the decoder prompt's cross-attention) runs the fused kernel of
    ``csrc/kernels/dense_attention.hip``.  Otherwise: the score matrix makes three trips through
    memory in the model's dtype (GEMM out, softmax in and out, GEMM in); the softmax accumulates
    in fp32 internally.  The 1/sqrt(head) factor (a power of two for 64-channel heads, so exact)
    goes onto q.  An fp32 copy of the scores - 3.4 GB per layer for 32 windows of Whisper-small -
    would add nothing: the GEMM has already rounded them to the model's dtype."""
    b, tq, d = q.shape
    tk = k.shape[1]
    hd = d // heads
    if mask is None and hd == 64 and ops.dense_attention_supported(q, k, v, heads):
        return ops.dense_attention(q, k, v, heads)  # bf16 on the GPU: MFMA flash kernel
    qh = (q * (hd ** -0.5)).view(b, tq, heads, hd).transpose(1, 2)
    kh = k.view(b, tk, heads, hd).transpose(1, 2)
    vh = v.view(b, tk, heads, hd).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2))
    if mask is not None:
        s = s + mask.to(s.dtype)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vh).transpose(1, 2).reshape(b, tq, d)


class WhisperModel:
    def __init__(self, cfg: WhisperConfig, dtype: torch.dtype, device: torch.device):
        self.cfg = cfg
        self.dtype = dtype
        self.device = device
        self.w: dict[str, torch.Tensor] = {}
        self._conv_w: dict[str, tuple] = {}  # convolution kernels in GEMM layout, by layer
        self.filters = torch.from_numpy(mel_filterbank(cfg.num_mel_bins)).to(device)
        # graph decode (GPU only): static buffers for ``graph_max_batch`` windows, one captured
        # step per power-of-two batch size
        self.use_graphs = device.type == "cuda"
        self.graph_max_batch = 32
        self.graph_check_every = 8
        self._gstate: dict | None = None

    # ------------------------------------------------------------------ weights
    def _names(self) -> dict[str, tuple[int, ...]]:
        c = self.cfg
        d = c.d_model
        shapes: dict[str, tuple[int, ...]] = {
            "encoder.conv1.weight": (d, c.num_mel_bins, 3), "encoder.conv1.bias": (d,),
            "encoder.conv2.weight": (d, d, 3), "encoder.conv2.bias": (d,),
            "encoder.embed_positions.weight": (c.max_source_positions, d),
            "encoder.layer_norm.weight": (d,), "encoder.layer_norm.bias": (d,),
            "decoder.embed_tokens.weight": (c.vocab_size, d),
            "decoder.embed_positions.weight": (c.max_target_positions, d),
            "decoder.layer_norm.weight": (d,), "decoder.layer_norm.bias": (d,),
        }

        def attn(prefix):
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                shapes[f"{prefix}.{proj}.weight"] = (d, d)
                if proj != "k_proj":
                    shapes[f"{prefix}.{proj}.bias"] = (d,)

        def block(prefix, ffn, cross):
            attn(f"{prefix}.self_attn")
            shapes[f"{prefix}.self_attn_layer_norm.weight"] = (d,)
            shapes[f"{prefix}.self_attn_layer_norm.bias"] = (d,)
            if cross:
                attn(f"{prefix}.encoder_attn")
                shapes[f"{prefix}.encoder_attn_layer_norm.weight"] = (d,)
                shapes[f"{prefix}.encoder_attn_layer_norm.bias"] = (d,)
            shapes[f"{prefix}.fc1.weight"] = (ffn, d)
            shapes[f"{prefix}.fc1.bias"] = (ffn,)
            shapes[f"{prefix}.fc2.weight"] = (d, ffn)
            shapes[f"{prefix}.fc2.bias"] = (d,)
            shapes[f"{prefix}.final_layer_norm.weight"] = (d,)
            shapes[f"{prefix}.final_layer_norm.bias"] = (d,)

        for i in range(c.encoder_layers):
            block(f"encoder.layers.{i}", c.encoder_ffn_dim, False)
        for i in range(c.decoder_layers):
            block(f"decoder.layers.{i}", c.decoder_ffn_dim, True)
        return shapes

    def init_random(self, seed: int = 0) -> None:
        g = torch.Generator().manual_seed(seed)
        for name, shape in self._names().items():
            if name == "encoder.embed_positions.weight":
                t = _sinusoids(*shape)
            elif "layer_norm.weight" in name:
                t = torch.ones(shape)
            elif name.endswith(".bias"):
                t = torch.zeros(shape)
            else:
                t = torch.empty(shape).normal_(0.0, 0.02, generator=g)
            self.w[name] = t.to(device=self.device, dtype=self.dtype)

    def load_hf(self, ckpt_dir: Path) -> None:
        """Read ``model.*`` tensors from the safetensors shards of a Hugging Face checkpoint."""
        from safetensors import safe_open

        want = self._names()
        files = sorted(Path(ckpt_dir).glob("*.safetensors"))
        if not files:
            raise FileNotFoundError(f"no .safetensors files in {ckpt_dir}")
        for fp in files:
            with safe_open(str(fp), framework="pt") as f:
                for key in f.keys():
                    name = key[6:] if key.startswith("model.") else key
                    if name not in want:
                        continue
                    t = f.get_tensor(key)
                    if tuple(t.shape) != want[name]:
                        raise ValueError(f"{key}: shape {tuple(t.shape)}, expected {want[name]}")
                    self.w[name] = t.to(device=self.device, dtype=self.dtype)
        if "encoder.embed_positions.weight" not in self.w:
            self.w["encoder.embed_positions.weight"] = _sinusoids(
                *want["encoder.embed_positions.weight"]).to(device=self.device, dtype=self.dtype)
        missing = [n for n in want if n not in self.w]
        if missing:
            raise ValueError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.w.values())

    # ------------------------------------------------------------------ blocks
    def _ln(self, x, prefix):
        return F.layer_norm(x, (x.shape[-1],), self.w[prefix + ".weight"], self.w[prefix + ".bias"], 1e-5)

    def _lin(self, x, prefix):
        return F.linear(x, self.w[prefix + ".weight"], self.w.get(prefix + ".bias"))

    def _mlp(self, x, prefix):
        return self._lin(F.gelu(self._lin(x, prefix + ".fc1")), prefix + ".fc2")

    def _conv3(self, x: torch.Tensor, prefix: str, stride: int) -> torch.Tensor:
        """Width-3 convolution over time with padding 1 as ONE GEMM: x [B, T, C_in] (channels
        last) -> [B, T_out, C_out].  In a zero-padded channels-last buffer the three taps of output
        position t are the 3 * C_in contiguous values starting at row t * stride, so the unfolded
        operand is an overlapping strided view of the input, and the weight is the [C_out, C_in, 3]
        kernel with its last two axes swapped.  (The convolution library has no tuned bf16 kernel
        for the 80-channel first layer and falls back to a scalar loop: 10 ms per 32 windows, a
        third of the encoder.)"""
        wt = self.w[prefix + ".weight"]
        key = (prefix, wt.data_ptr())
        w2 = self._conv_w.get(prefix)
        if w2 is None or w2[0] != key:
            w2 = (key, wt.permute(0, 2, 1).reshape(wt.shape[0], -1).contiguous())
            self._conv_w[prefix] = w2
        b, t, ci = x.shape
        t_out = (t - 1) // stride + 1
        xp = F.pad(x, (0, 0, 1, 1))
        cols = xp.as_strided((b, t_out, 3 * ci), ((t + 2) * ci, stride * ci, 1))
        return F.linear(cols, w2[1], self.w[prefix + ".bias"])

    # ------------------------------------------------------------------ encoder
    @torch.inference_mode()
    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        """[B, n_mels, 3000] -> [B, 1500, d]."""
        c, w = self.cfg, self.w
        x = F.gelu(self._conv3(mel.to(self.dtype).transpose(1, 2), "encoder.conv1", 1))
        x = F.gelu(self._conv3(x, "encoder.conv2", 2))
        x = x + w["encoder.embed_positions.weight"][: x.shape[1]]
        for i in range(c.encoder_layers):
            p = f"encoder.layers.{i}"
            h = self._ln(x, p + ".self_attn_layer_norm")
            a = _attend(self._lin(h, p + ".self_attn.q_proj"), self._lin(h, p + ".self_attn.k_proj"),
                        self._lin(h, p + ".self_attn.v_proj"), c.encoder_heads)
            x = x + self._lin(a, p + ".self_attn.out_proj")
            x = x + self._mlp(self._ln(x, p + ".final_layer_norm"), p)
        return self._ln(x, "encoder.layer_norm")

    @torch.inference_mode()
    def cross_kv(self, enc: torch.Tensor) -> list[tuple[torch.Tensor, torch.Tensor]]:
        """Per decoder layer K/V of the encoder output: computed once per window."""
        return [(self._lin(enc, f"decoder.layers.{i}.encoder_attn.k_proj"),
                 self._lin(enc, f"decoder.layers.{i}.encoder_attn.v_proj"))
                for i in range(self.cfg.decoder_layers)]

    # ------------------------------------------------------------------ decoder
    @torch.inference_mode()
    def decode(self, tokens: torch.Tensor, offset: int, cross: list, cache: list) -> torch.Tensor:
        """Run ``tokens`` [B, T] at positions offset.. through the decoder; ``cache`` holds the
        self-attention K/V of earlier positions per layer and is extended in place.
        Returns logits of the last position [B, vocab] (fp32)."""
        c, w = self.cfg, self.w
        b, t = tokens.shape
        x = F.embedding(tokens, w["decoder.embed_tokens.weight"]) + \
            w["decoder.embed_positions.weight"][offset: offset + t]
        mask = None
        if t > 1:
            mask = torch.full((t, offset + t), float("-inf"), device=x.device)
            mask = torch.triu(mask, diagonal=offset + 1)
        for i in range(c.decoder_layers):
            p = f"decoder.layers.{i}"
            h = self._ln(x, p + ".self_attn_layer_norm")
            k_new = self._lin(h, p + ".self_attn.k_proj")
            v_new = self._lin(h, p + ".self_attn.v_proj")
            if cache[i] is None:
                k_all, v_all = k_new, v_new
            else:
                k_all = torch.cat([cache[i][0], k_new], dim=1)
                v_all = torch.cat([cache[i][1], v_new], dim=1)
            cache[i] = (k_all, v_all)
            a = _attend(self._lin(h, p + ".self_attn.q_proj"), k_all, v_all, c.decoder_heads, mask)
            x = x + self._lin(a, p + ".self_attn.out_proj")
            h = self._ln(x, p + ".encoder_attn_layer_norm")
            a = _attend(self._lin(h, p + ".encoder_attn.q_proj"), cross[i][0], cross[i][1], c.decoder_heads)
            x = x + self._lin(a, p + ".encoder_attn.out_proj")
            x = x + self._mlp(self._ln(x, p + ".final_layer_norm"), p)
        x = self._ln(x[:, -1], "decoder.layer_norm")
        return F.linear(x, w["decoder.embed_tokens.weight"]).float()

    # ------------------------------------------------------------------ graph decode
    def drop_graphs(self) -> None:
        """Forget the captured steps, their buffers and derived weights (before the weights move
        or are freed)."""
        self._gstate = None
        self._conv_w = {}

    def _suppress_mask(self) -> torch.Tensor:
        c = self.cfg
        ids = torch.arange(c.vocab_size, device=self.device)
        lo = c.suppress_lo or c.eos_token_id + 1
        hi = c.suppress_hi or c.vocab_size
        return (ids >= lo) & (ids < hi) & (ids != c.eos_token_id)

    def _graph_state(self) -> dict:
        """Static device state of the graph decode path, allocated on first use: the running
        token / position / finished flags, the emitted-token history, self-attention K/V for
        every target position and cross-attention K/V, all for ``graph_max_batch`` windows
        (whisper-large at 32 windows: 2.3 GB + 7.9 GB), and q/k/v weights fused per layer."""
        if self._gstate is not None:
            return self._gstate
        c, w, dev = self.cfg, self.w, self.device
        nb, t, s, d = self.graph_max_batch, c.max_target_positions, c.max_source_positions, c.d_model

        def z(*shape):
            return torch.zeros(shape, dtype=self.dtype, device=dev)

        wqkv, bqkv = [], []
        for i in range(c.decoder_layers):
            p = f"decoder.layers.{i}.self_attn"
            wqkv.append(torch.cat([w[p + ".q_proj.weight"], w[p + ".k_proj.weight"], w[p + ".v_proj.weight"]]))
            bqkv.append(torch.cat([w[p + ".q_proj.bias"], torch.zeros_like(w[p + ".q_proj.bias"]),
                                   w[p + ".v_proj.bias"]]))
        layers = range(c.decoder_layers)
        self._gstate = {
            "tok": torch.zeros(nb, dtype=torch.long, device=dev),
            "pos": torch.zeros(1, dtype=torch.long, device=dev),
            "done": torch.ones(nb, dtype=torch.bool, device=dev),
            "hist": torch.full((nb, t), c.eos_token_id, dtype=torch.long, device=dev),
            "kc": [z(nb, t, d) for _ in layers], "vc": [z(nb, t, d) for _ in layers],
            "ck": [z(nb, s, d) for _ in layers], "cv": [z(nb, s, d) for _ in layers],
            "steps": torch.arange(t, device=dev), "suppress": self._suppress_mask(),
            "wqkv": wqkv, "bqkv": bqkv, "graphs": {}, "pool": None,
        }
        return self._gstate

    def _attend1(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pos: torch.Tensor | None,
                 mask: torch.Tensor | None) -> torch.Tensor:
        """One query per row: q [B, d] against k / v [B, T, d] -> [B, d].  bf16 on the GPU with
        64-channel heads (every published Whisper size) runs the fused dense single-query kernel
        (``csrc/kernels/dense_attention.hip``); anything else the batched-GEMM definition."""
        heads = self.cfg.decoder_heads
        if ops.dense_decode_attention_supported(q, k, v, heads):
            return ops.dense_decode_attention(q, k, v, heads, pos)
        return _attend(q[:, None], k, v, heads, mask)[:, 0]

    def _graph_step(self, st: dict, b: int) -> None:
        """One greedy decoder step for rows [0, b) of the static state: consumes ``tok`` at
        ``pos``, appends its K/V, writes the next token to ``tok`` and ``hist[:, pos]`` and
        advances ``pos``.  Finished rows keep emitting end-of-text.  Projections of up to 16
        rows take the weight-streaming kernels of ``ops.linear``."""
        c, w = self.cfg, self.w
        d, eos = c.d_model, c.eos_token_id
        tok, pos, done = st["tok"][:b], st["pos"], st["done"][:b]

        def lin(x, prefix):
            return ops.linear(x, w[prefix + ".weight"], w.get(prefix + ".bias"))

        x = F.embedding(tok, w["decoder.embed_tokens.weight"]) + \
            w["decoder.embed_positions.weight"].index_select(0, pos)
        mask = torch.where(st["steps"] <= pos, 0.0, float("-inf"))  # keys beyond pos are not written yet
        for i in range(c.decoder_layers):
            p = f"decoder.layers.{i}"
            h = self._ln(x, p + ".self_attn_layer_norm")
            qkv = ops.linear(h, st["wqkv"][i], st["bqkv"][i])
            kc, vc = st["kc"][i][:b], st["vc"][i][:b]
            kc.index_copy_(1, pos, qkv[:, None, d:2 * d])
            vc.index_copy_(1, pos, qkv[:, None, 2 * d:])
            x = x + lin(self._attend1(qkv[:, :d], kc, vc, pos, mask), p + ".self_attn.out_proj")
            h = self._ln(x, p + ".encoder_attn_layer_norm")
            a = self._attend1(lin(h, p + ".encoder_attn.q_proj"), st["ck"][i][:b], st["cv"][i][:b], None, None)
            x = x + lin(a, p + ".encoder_attn.out_proj")
            h = self._ln(x, p + ".final_layer_norm")
            x = x + lin(F.gelu(lin(h, p + ".fc1")), p + ".fc2")
        logits = ops.linear(self._ln(x, "decoder.layer_norm"), w["decoder.embed_tokens.weight"]).float()
        logits.masked_fill_(st["suppress"], float("-inf"))
        nxt = torch.where(done, eos, logits.argmax(dim=-1))
        done.logical_or_(nxt == eos)
        st["hist"][:b].index_copy_(1, pos, nxt[:, None])
        tok.copy_(nxt)
        pos.add_(1)

    def _graph_for(self, st: dict, b: int) -> "torch.cuda.CUDAGraph":
        """The captured step for batch size ``b``; captured (after one eager run that lets the
        libraries load their kernels) on first use.  Leaves the row state dirty."""
        g = st["graphs"].get(b)
        if g is None:
            st["pos"].zero_()
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._graph_step(st, b)
            cur.wait_stream(side)
            st["pos"].zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=st["pool"]):
                self._graph_step(st, b)
            if st["pool"] is None:
                st["pool"] = g.pool()
            st["graphs"][b] = g
        return g

    @torch.inference_mode()
    def capture_graphs(self) -> int:
        """Capture the decode step for every batch bucket now (replica start) instead of on the
        first request that needs it; returns the number of graphs held."""
        if not self.use_graphs:
            return 0
        st = self._graph_state()
        b = 1
        while True:
            self._graph_for(st, min(b, self.graph_max_batch))
            if b >= self.graph_max_batch:
                break
            b *= 2
        return len(st["graphs"])

    def _generate_graph(self, cross: list, tokens: torch.Tensor, steps: int) -> list[list[int]]:
        """Greedy decoding with one graph replay per token.  The prompt runs eagerly (several
        positions at once); its K/V and the cross K/V are copied into the static buffers."""
        c = self.cfg
        b, n0 = tokens.shape
        eos = c.eos_token_id
        nb = min(1 << (b - 1).bit_length(), self.graph_max_batch)
        st = self._graph_state()
        g = self._graph_for(st, nb)
        cache: list = [None] * c.decoder_layers
        logits = self.decode(tokens, 0, cross, cache)
        logits[:, st["suppress"]] = float("-inf")
        first = logits.argmax(dim=-1)
        st["tok"][:b] = first
        st["pos"].fill_(n0)
        st["done"][:nb] = True
        st["done"][:b] = first == eos
        st["hist"][:nb].fill_(eos)
        for i in range(c.decoder_layers):
            st["kc"][i][:b, :n0] = cache[i][0]
            st["vc"][i][:b, :n0] = cache[i][1]
            st["ck"][i][:b] = cross[i][0]
            st["cv"][i][:b] = cross[i][1]
        ran = 0
        for _ in range(steps - 1):
            g.replay()
            ran += 1
            if ran % self.graph_check_every == 0 and bool(st["done"][:b].all()):
                break
        rows = torch.cat([first[:, None], st["hist"][:b, n0:n0 + ran]], dim=1).tolist()
        return [row[:row.index(eos)] if eos in row else row for row in rows]

    # ------------------------------------------------------------------ generation
    def prompt_ids(self, language_token: int | None, task: str = "transcribe") -> list[int]:
        c = self.cfg
        lang = c.first_language_token_id if language_token is None else language_token
        task_id = c.translate_token_id if task == "translate" else c.transcribe_token_id
        return [c.sot_token_id, lang, task_id, c.no_timestamps_token_id]

    @torch.inference_mode()
    def transcribe_windows(self, windows: torch.Tensor, prompts: list[list[int]],
                           max_new_tokens: int | None = None, temperature: float = 0.0,
                           generator: torch.Generator | None = None) -> list[list[int]]:
        """Greedy (or sampled) decoding of a batch of 30 s windows [B, N_SAMPLES].  Prompts must
        have equal length (they do: sot, language, task, no-timestamps).  Special tokens at or
        above the end-of-text id other than end-of-text itself are never emitted."""
        c = self.cfg
        b = windows.shape[0]
        mel = log_mel(windows.to(self.device), c.num_mel_bins, self.filters)
        cross = self.cross_kv(self.encode(mel))
        tokens = torch.tensor(prompts, dtype=torch.long, device=self.device)
        limit = c.max_target_positions - tokens.shape[1]
        steps = limit if max_new_tokens is None else min(limit, max_new_tokens)
        if self.use_graphs and temperature <= 0 and b <= self.graph_max_batch and steps > 0:
            return self._generate_graph(cross, tokens, steps)
        cache: list = [None] * c.decoder_layers
        done = torch.zeros(b, dtype=torch.bool, device=self.device)
        out: list[list[int]] = [[] for _ in range(b)]
        offset, feed = 0, tokens
        suppress = self._suppress_mask()
        for step in range(steps):
            logits = self.decode(feed, offset, cross, cache)
            offset += feed.shape[1]
            logits[:, suppress] = float("-inf")
            if temperature > 0:
                probs = torch.softmax(logits / temperature, dim=-1)
                nxt = torch.multinomial(probs, 1, generator=generator).squeeze(1)
            else:
                nxt = logits.argmax(dim=-1)
            nxt = torch.where(done, torch.full_like(nxt, c.eos_token_id), nxt)
            done = done | (nxt == c.eos_token_id)
            host = nxt.tolist()
            for i, tkn in enumerate(host):
                if tkn != c.eos_token_id:
                    out[i].append(tkn)
            if bool(done.all()):
                break
            feed = nxt[:, None]
        return out
