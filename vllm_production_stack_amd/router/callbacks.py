"""User callbacks: ``pre_request`` may short-circuit with a response, ``post_request`` runs
in the background with the full response body.  The handler is named on the command line as
``<module path without .py>.<instance attribute>``.
Reference: /root/reference/src/vllm_router/services/callbacks_service/callbacks.py:23-32,
custom_callbacks.py:19-55."""

from __future__ import annotations

import importlib
import importlib.util
import os
import sys


class CustomCallbackHandler:
    def pre_request(self, request, request_body: bytes, request_json: dict):
        """Return None to continue, or a response object to answer immediately."""
        return None

    def post_request(self, request, response_content: bytes):
        return None


def load_callbacks(spec: str) -> CustomCallbackHandler:
    module_name, _, attr = spec.rpartition(".")
    if not module_name:
        raise ValueError(f"callbacks spec '{spec}' must look like <module>.<instance>")
    path = module_name if module_name.endswith(".py") else module_name + ".py"
    if os.path.sep in module_name or os.path.exists(path):
        name = os.path.basename(module_name).replace("-", "_")
        sp = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(sp)
        sys.modules[name] = mod
        sp.loader.exec_module(mod)
    else:
        mod = importlib.import_module(module_name)
    return getattr(mod, attr)
