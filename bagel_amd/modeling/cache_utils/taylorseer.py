"""TaylorSeer step skipping for the rectified-flow sampler -- the reference's ``modeling/cache_utils/taylorseer.py``
(cache_init :120-153, cal_type :83-117, force_scheduler :64-77, derivative_approximation :11-30, taylor_formula :32-46)
re-planned for MI355X.

What the reference does: on a 'full' step every decoder layer stores its output and its finite differences; on a
'Taylor' step every layer REPLACES the sequence by the Taylor extrapolation of its own cache (qwen2_navit.py:824-829),
so the layers overwrite one another and only the LAST layer's extrapolation reaches the final norm.  Here exactly that
surviving computation is kept: one feature cache (the last layer's output) per forward stream, updated by
``bagel_taylor_update_bf16`` after a full forward and evaluated by ``bagel_taylor_eval_bf16`` instead of the 28-layer
forward on a Taylor step -- bit-identical to the reference's result (tests/test_oracle_golden.py pins the equivalence
on the CPU oracle) at 1/28 of its cache traffic and memory.

The schedule is host integer logic and follows the reference's constants: 5 full warm-up steps, then every third step
full (fresh_threshold 3), orders up to 6.  49 Euler steps -> 19 full forwards per stream.
"""
import torch

from ... import ops

BF16 = torch.bfloat16


class TaylorSeerState:
    """(cache_dic, current) of ONE forward stream (cond / cfg-text / cfg-img each own one, bagel.py:680-684)."""

    FRESH_THRESHOLD = 3      # taylorseer.py:136
    MAX_ORDER = 6            # taylorseer.py:139
    FIRST_ENHANCE = 5        # taylorseer.py:140

    def __init__(self, num_steps):
        self.num_steps = num_steps
        self.cache_counter = 0
        self.cal_threshold = None
        self.activated_steps = [0]
        self.step = 0
        self.type = None
        self.n_factors = 0           # orders currently held (0 = nothing cached yet)
        self._bufs = []              # [rows, cols] bf16 buffers, order i at index i
        self.full_steps = 0
        self.taylor_steps = 0

    # ---- cal_type + force_scheduler (taylor_cache=True, fresh_ratio=0 => step_factor = 1) -----------------------
    def next_type(self):
        first = self.step < self.FIRST_ENHANCE
        fresh_interval = self.FRESH_THRESHOLD if first else self.cal_threshold
        if first or self.cache_counter == fresh_interval - 1:
            self.type = "full"
            self.cache_counter = 0
            self.activated_steps.append(self.step)
            self.cal_threshold = int(round(self.FRESH_THRESHOLD / 1.0))
        else:
            self.cache_counter += 1
            self.type = "Taylor"
        return self.type

    def _buffers(self, n, like):
        while len(self._bufs) < n:
            self._bufs.append(torch.empty((like.shape[0], like.shape[1]), dtype=BF16, device=like.device))
        if self._bufs and self._bufs[0].shape != like.shape:
            raise ValueError("TaylorSeer cache shape changed between steps")
        return self._bufs

    # ---- derivative_approximation on the last layer's output --------------------------------------------------
    def update(self, feature):
        if self.step == 0:
            self.n_factors = 0                                       # taylor_cache_init, taylorseer.py:48-56
        n_diff = min(self.n_factors, self.MAX_ORDER) if self.step > self.FIRST_ENHANCE - 2 else 0
        dist = self.activated_steps[-1] - self.activated_steps[-2]
        bufs = self._buffers(n_diff + 1, feature)
        ops.taylor_update(feature, bufs, n_diff, dist)
        self.n_factors = n_diff + 1
        self.full_steps += 1

    # ---- taylor_formula -------------------------------------------------------------------------------------------
    def eval_into(self, out):
        if self.n_factors == 0:
            raise RuntimeError("TaylorSeer: a Taylor step before any full step")
        x = self.step - self.activated_steps[-1]
        ops.taylor_eval(self._bufs, self.n_factors, x, out)
        self.taylor_steps += 1

    def advance(self):
        self.step += 1


def cache_init(self, num_steps):
    """Reference-shaped helper (taylorseer.py:120): one state object plays both roles of (cache_dic, current)."""
    st = TaylorSeerState(num_steps)
    return st, st
