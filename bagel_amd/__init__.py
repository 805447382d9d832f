"""bagel_amd -- MI355X-native (gfx950) implementation of BAGEL's unified multimodal forward path.

Drop-in for the reference's ``modeling.bagel`` / ``inferencer`` surface:

    from bagel_amd.modeling.bagel import BagelConfig, Bagel, Qwen2Config, Qwen2ForCausalLM, SiglipVisionConfig, SiglipVisionModel
    from bagel_amd.modeling.bagel.qwen2_navit import NaiveCache
    from bagel_amd.inferencer import InterleaveInferencer

``bagel_amd.install_as_reference()`` aliases the package so unmodified reference scripts that do
``from modeling.bagel import ...`` / ``from inferencer import InterleaveInferencer`` pick this implementation up.
All arithmetic runs in libbagel_hip.so (hand-written HIP); there is no CPU or eager fallback.
"""
import importlib
import sys

__version__ = "0.1.0"


def install_as_reference():
    """Make ``modeling``, ``modeling.bagel``, ``modeling.bagel.qwen2_navit``, ``modeling.autoencoder``, ``inferencer``
    and ``data.data_utils`` import THIS implementation (for unmodified reference entry scripts)."""
    names = {
        "modeling": "bagel_amd.modeling",
        "modeling.bagel": "bagel_amd.modeling.bagel",
        "modeling.bagel.bagel": "bagel_amd.modeling.bagel.bagel",
        "modeling.bagel.qwen2_navit": "bagel_amd.modeling.bagel.qwen2_navit",
        "modeling.bagel.siglip_navit": "bagel_amd.modeling.bagel.siglip_navit",
        "modeling.bagel.modeling_utils": "bagel_amd.modeling.bagel.modeling_utils",
        "modeling.autoencoder": "bagel_amd.modeling.autoencoder",
        "inferencer": "bagel_amd.inferencer",
        "data": "bagel_amd.data",
        "data.data_utils": "bagel_amd.data.data_utils",
    }
    for alias, real in names.items():
        try:
            sys.modules[alias] = importlib.import_module(real)
        except ModuleNotFoundError:
            pass
