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
    """Make ``modeling`` (``.bagel[.*]``, ``.autoencoder``, ``.cache_utils.taylorseer``, ``.qwen2``), ``inferencer`` and ``data``
    (``.data_utils``, ``.transforms``) import THIS implementation: every import of the reference's inference entry scripts
    (app.py:10-19, eval/gen/gen_images_mp.py:11-19, inferencer.py:10-11) then resolves here."""
    names = {
        "modeling": "bagel_amd.modeling",
        "modeling.bagel": "bagel_amd.modeling.bagel",
        "modeling.bagel.bagel": "bagel_amd.modeling.bagel.bagel",
        "modeling.bagel.qwen2_navit": "bagel_amd.modeling.bagel.qwen2_navit",
        "modeling.bagel.siglip_navit": "bagel_amd.modeling.bagel.siglip_navit",
        "modeling.bagel.modeling_utils": "bagel_amd.modeling.bagel.modeling_utils",
        "modeling.autoencoder": "bagel_amd.modeling.autoencoder",
        "modeling.cache_utils": "bagel_amd.modeling.cache_utils",
        "modeling.cache_utils.taylorseer": "bagel_amd.modeling.cache_utils.taylorseer",
        "modeling.qwen2": "bagel_amd.modeling.qwen2",
        "inferencer": "bagel_amd.inferencer",
        "data": "bagel_amd.data",
        "data.data_utils": "bagel_amd.data.data_utils",
        "data.transforms": "bagel_amd.data.transforms",
    }
    for alias, real in names.items():
        try:
            sys.modules[alias] = importlib.import_module(real)
        except ModuleNotFoundError:
            pass
