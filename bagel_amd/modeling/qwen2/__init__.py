"""``modeling.qwen2`` of the reference as far as the inference entry scripts use it: ``Qwen2Tokenizer``
(app.py:19,68; eval/gen/gen_images_mp.py:15,131).

The reference vendors HuggingFace's Qwen2 tokenizer (modeling/qwen2/tokenization_qwen2.py is a copy of
transformers.models.qwen2.tokenization_qwen2); tokenisation is host text processing outside the hot path (SURVEY.md section 8,
DESIGN.md section 7), so this module hands out the upstream class instead of carrying a third copy.  Vocabulary files are the
caller's (``Qwen2Tokenizer.from_pretrained(model_path)``); none exist offline, which is why tests use a stub tokenizer.
"""


def __getattr__(name):
    if name in ("Qwen2Tokenizer", "Qwen2TokenizerFast"):
        import transformers
        return getattr(transformers, name)
    if name == "Qwen2Config":
        from ..bagel.qwen2_navit import Qwen2Config
        return Qwen2Config
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
