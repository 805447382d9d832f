"""Lifecycle of the packed weight copies the engines keep beside the nn.Parameters (fused Wqkv, interleaved gate/up, padded heads,
NHWC conv filters).  A packed copy is valid for exactly one state of the parameters; it has to go when the parameters change by
ANY of the routes a reference user has (app.py:105-113 accelerate load, gen_images_mp.py:165-176 load_state_dict on the parent
Bagel, the EMA swap of train/pretrain_unified_navit.py, ``param.data.copy_``):

* ``module.to() / .cuda() / .half()``            -> ``_apply`` override
* ``load_state_dict`` on the module OR ANY PARENT -> ``register_load_state_dict_post_hook`` (fires inside the recursive load)
* ``param.copy_() / optimizer steps / param.data = new`` -> a (data_ptr, _version) signature of every parameter and buffer,
                                                     compared at the public entry points (once per prefill / generate_* call)
* writes through a detached alias (``param.data.copy_(x)``) leave pointer AND version counter untouched -- nothing cheap can see
  them: call ``module.invalidate_packed()`` afterwards (``init_moe`` does).
"""
from torch import nn


class PackedWeights(nn.Module):
    """Mixin base: subclasses implement ``_drop_packed()`` (forget the copies) and call ``_packed_fresh()`` right after (re)packing
    and ``_check_packed()`` at their public entry points."""

    def __init__(self):
        super().__init__()
        self._packed_sig = None
        self.register_load_state_dict_post_hook(lambda module, _incompatible: module.invalidate_packed())

    def _drop_packed(self):
        raise NotImplementedError

    def invalidate_packed(self):
        self._packed_sig = None
        self._drop_packed()

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def _signature(self):
        # inference tensors (model built / loaded / moved under torch.inference_mode()) keep no version counter and cannot be
        # rewritten in place outside inference mode: their storage pointer is the whole signature
        def one(t):
            return (t.data_ptr(), 0 if t.is_inference() else t._version)
        sig = [one(t) for t in self.parameters()]
        sig += [one(t) for t in self.buffers()]
        return hash(tuple(sig))

    def _packed_fresh(self):
        self._packed_sig = self._signature()

    def _check_packed(self):
        """Drop the packed copies if any parameter was rewritten or re-seated since they were built."""
        if self._packed_sig is not None and self._packed_sig != self._signature():
            self.invalidate_packed()
