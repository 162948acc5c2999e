"""Router logging: coloured level tags, records up to INFO on stdout and WARNING and above on
stderr, no propagation to the root logger (behaviour of
/root/reference/src/vllm_router/log.py:6-60)."""

from __future__ import annotations

import logging
import sys

_COLORS = {"DEBUG": "\033[37m", "INFO": "\033[32m", "WARNING": "\033[33m", "ERROR": "\033[31m",
           "CRITICAL": "\033[1;31m"}
_RESET = "\033[0m"
_LEVEL = logging.INFO


class _Formatter(logging.Formatter):
    def __init__(self, color: bool):
        super().__init__("[%(asctime)s] %(levelname)s: %(message)s (%(filename)s:%(lineno)d:%(name)s)",
                         "%Y-%m-%d %H:%M:%S")
        self.color = color

    def format(self, record):
        s = super().format(record)
        if self.color:
            c = _COLORS.get(record.levelname)
            if c:
                s = s.replace(record.levelname, f"{c}{record.levelname}{_RESET}", 1)
        return s


class _MaxLevel(logging.Filter):
    def __init__(self, level):
        super().__init__()
        self.level = level

    def filter(self, record):
        return record.levelno <= self.level


def set_level(level: str) -> None:
    global _LEVEL
    lv = {"trace": logging.DEBUG, "critical": logging.CRITICAL}.get(level.lower())
    _LEVEL = lv if lv is not None else getattr(logging, level.upper(), logging.INFO)
    for name, lg in logging.root.manager.loggerDict.items():
        if isinstance(lg, logging.Logger) and getattr(lg, "_vpsa_router", False):
            lg.setLevel(_LEVEL)


def init_logger(name: str) -> logging.Logger:
    lg = logging.getLogger(name)
    if getattr(lg, "_vpsa_router", False):
        return lg
    lg.setLevel(_LEVEL)
    out = logging.StreamHandler(sys.stdout)
    out.addFilter(_MaxLevel(logging.INFO))
    out.setFormatter(_Formatter(sys.stdout.isatty()))
    err = logging.StreamHandler(sys.stderr)
    err.setLevel(logging.WARNING)
    err.setFormatter(_Formatter(sys.stderr.isatty()))
    lg.addHandler(out)
    lg.addHandler(err)
    lg.propagate = False
    lg._vpsa_router = True
    return lg
