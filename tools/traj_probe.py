"""First Euler steps of the 49-step 7B-width trajectory (tests/golden/wide7b_traj49.pt) under whatever BAGEL_* switches the environment sets:
rel-L2 of the displacement x_n - x_0 against the reference and against its fp32-accumulating restatement, n = 1, 2, 3.  Also the same first
step computed by ONE direct forward (generate_image's step function) for comparison with the loop.
    python tools/traj_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.configs import NEW_TOKEN_IDS_TINY, WIDE7B, StubTokenizer  # noqa: E402
from tests.test_model_gpu import cfg_kwargs, new_cache, rel_l2  # noqa: E402
from tests.test_wide_gpu import _wide_model  # noqa: E402


def main():
    cfg = WIDE7B
    g = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "wide7b_traj49.pt"), weights_only=False)
    model = _wide_model()
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    gi, _, _ = model.prepare_prompts([0], [0], [g["prompt"]], tok, NEW_TOKEN_IDS_TINY)
    x0 = g["latent_inputs"]["packed_init_noises"].float()
    kw = dict(g["gen_kwargs"])
    for rep in range(2):
        cache = model.forward_cache_update_text(new_cache(cfg), **gi)
        snaps = {}
        model.step_hook = lambda n, x: snaps.__setitem__(n, x.detach().float().cpu().clone()) if n <= 3 else None
        kw3 = dict(kw)
        try:
            model.generate_image(past_key_values=cache, **cfg_kwargs("cfg_text", new_cache(cfg), g["cfg_inputs"]), **kw3, **g["latent_inputs"])
        finally:
            model.step_hook = None
        out = []
        for n, d, d32 in zip(g["snap_steps"], g["snap_disp"], g["snap_disp_f32acc"]):
            if n > 3:
                break
            out.append(f"n={n}: vs reference {rel_l2(snaps[n] - x0, d.float()):.3e}, vs fp32-acc {rel_l2(snaps[n] - x0, d32.float()):.3e}, "
                       f"reference vs fp32-acc {rel_l2(d.float(), d32.float()):.3e}")
        print(f"rep {rep} [{' '.join(k + '=' + v for k, v in os.environ.items() if k.startswith('BAGEL_'))}] displacement rel-L2: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
