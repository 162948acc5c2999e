"""Chat prompts against Hugging Face: a tokenizer directory written by ``save_pretrained`` (with
a chat template) must render and tokenise conversations exactly like
``PreTrainedTokenizerFast.apply_chat_template`` - whitespace control, special-token strings,
tools and generation prompt included - and the engine must stop on every end-of-sequence id
the checkpoint declares."""

import json

import pytest

from vllm_production_stack_amd.engine.tokenizer import load_tokenizer

transformers = pytest.importorskip("transformers")

LLAMA3_STYLE = """{{- bos_token }}
{%- if tools %}
    {{- "<|start_header_id|>system<|end_header_id|>\\n\\nTools: " }}
    {%- for t in tools %}
        {{- t | tojson(indent=2) }}
        {{- "\\n" }}
    {%- endfor %}
    {{- "<|eot_id|>" }}
{%- endif %}
{%- for message in messages %}
    {%- if message.role not in ["system", "user", "assistant"] %}
        {{- raise_exception("unknown role " + message.role) }}
    {%- endif %}
    {{- "<|start_header_id|>" + message.role + "<|end_header_id|>\\n\\n" + message.content | trim + "<|eot_id|>" }}
{%- endfor %}
{%- if add_generation_prompt %}
    {{- "<|start_header_id|>assistant<|end_header_id|>\\n\\n" }}
{%- endif %}
"""

# relies on trim_blocks / lstrip_blocks (no explicit whitespace control) and on loop controls
CHATML_STYLE = """{% for message in messages %}
    {% if message.role == "ignored" %}{% continue %}{% endif %}
{{ "<|im_start|>" + message.role }}
{{ message.content + "<|im_end|>" }}
{% endfor %}
{% if add_generation_prompt %}
{{ "<|im_start|>assistant" }}
{% endif %}
"""

MESSAGES = [
    {"role": "system", "content": "You are terse.  "},
    {"role": "user", "content": "What is 2 + 2?\nAnswer with a digit: café."},
    {"role": "assistant", "content": "4"},
    {"role": "user", "content": "And 3 + 3?"},
]
TOOLS = [{"type": "function", "function": {"name": "add", "description": "adds é",
                                            "parameters": {"type": "object", "properties": {}}}}]


def _make_tokenizer(tmp_path, template):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    words = sorted({w for m in MESSAGES for w in m["content"].split()} |
                   {"system", "user", "assistant", "Tools:", "ignored"})
    specials = ["<unk>", "<s>", "</s>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>",
                "<|im_start|>", "<|im_end|>"]
    vocab = {t: i for i, t in enumerate(specials + words)}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="<|eot_id|>",
                                   unk_token="<unk>", additional_special_tokens=specials[3:])
    fast.chat_template = template
    fast.save_pretrained(tmp_path)
    return fast


@pytest.mark.parametrize("template", [LLAMA3_STYLE, CHATML_STYLE], ids=["llama3-style", "chatml-style"])
@pytest.mark.parametrize("gen", [True, False])
def test_rendering_and_ids_match_hugging_face(tmp_path, template, gen):
    hf = _make_tokenizer(tmp_path, template)
    ours = load_tokenizer(tmp_path, vocab_size=0)
    assert ours.chat_template is not None and ours.bos_token == "<s>" and ours.eos_token == "<|eot_id|>"
    tools = TOOLS if template is LLAMA3_STYLE else None
    want_text = hf.apply_chat_template(MESSAGES, tokenize=False, add_generation_prompt=gen, tools=tools)
    assert ours.render_chat(MESSAGES, gen, tools) == want_text
    want = hf.apply_chat_template(MESSAGES, tokenize=True, add_generation_prompt=gen, tools=tools)
    want_ids = want["input_ids"] if isinstance(want, dict) or hasattr(want, "keys") else want
    assert ours.encode_chat(MESSAGES, gen, tools) == list(want_ids)


def test_template_errors_surface(tmp_path):
    _make_tokenizer(tmp_path, LLAMA3_STYLE)
    ours = load_tokenizer(tmp_path, vocab_size=0)
    with pytest.raises(Exception, match="unknown role"):
        ours.render_chat([{"role": "robot", "content": "x"}])


def test_every_declared_eos_id_stops_generation(tmp_path):
    _make_tokenizer(tmp_path, LLAMA3_STYLE)
    (tmp_path / "generation_config.json").write_text(json.dumps({"eos_token_id": [2, 7, 5]}))
    ours = load_tokenizer(tmp_path, vocab_size=0)
    assert ours.eos_token_id == 5  # the tokenizer's own end-of-turn token
    assert ours.eos_token_ids == {2, 5, 7}
