"""``from modeling.bagel import ...`` surface of the reference (modeling/bagel/__init__.py:5-18)."""
from .bagel import Bagel, BagelConfig
from .qwen2_navit import NaiveCache, Qwen2Config, Qwen2ForCausalLM, Qwen2Model
from .siglip_navit import SiglipVisionConfig, SiglipVisionModel

__all__ = ["BagelConfig", "Bagel", "Qwen2Config", "Qwen2Model", "Qwen2ForCausalLM", "SiglipVisionConfig",
           "SiglipVisionModel", "NaiveCache"]
