"""MI355X-native multi-instance LLM serving stack (router + engine + KV tiering)."""

__version__ = "0.1.0"
