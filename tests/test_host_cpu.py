"""CPU-only checks of the product's host logic: C-ABI exports, state-dict compatibility, bit-exact packers."""
import ctypes
import os

import pytest
import torch

from bagel_amd import _lib
from bagel_amd.factory import build_bagel
from oracle import packers as P
from oracle.configs import TINY, TINY_D128, NEW_TOKEN_IDS_TINY, StubTokenizer
from oracle.shapes import bagel_shapes, vae_shapes


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 15
    assert os.path.exists(_lib.LIB_PATH), "run `python -m bagel_amd.build` (done by __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(L, name), f"{name} declared in include/bagel_hip.h but not exported"
    L.bagel_hip_version.restype = ctypes.c_int
    assert L.bagel_hip_version() >= 100
    L.bagel_hip_arch.restype = ctypes.c_char_p
    assert L.bagel_hip_arch() == b"gfx950"


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (no GPU needed)."""
    L = _lib.lib()
    rc = L.bagel_gemm_bf16(None, 0, None, None, None, None, 0, None, None, None, None, 0, 0, None, 0, None, 0, 8, 64, 0, 0, None)
    assert rc < 0 and b"null" in L.bagel_hip_last_error()
    rc = L.bagel_rmsnorm_bf16(1, 8, 1, None, None, 1, 8, 4, 7, 1e-6, None)
    assert rc < 0 and b"multiples of 8" in L.bagel_hip_last_error()


def test_no_cpu_fallback():
    from bagel_amd import ops
    with pytest.raises(_lib.BagelHipError):
        ops.rmsnorm(torch.zeros(4, 8, dtype=torch.bfloat16), torch.ones(8, dtype=torch.bfloat16),
                    torch.zeros(4, 8, dtype=torch.bfloat16), 1e-6)


@pytest.mark.parametrize("cfg", [TINY, TINY_D128], ids=lambda c: c["name"])
def test_state_dict_keys_and_shapes(cfg):
    model, vae = build_bagel(cfg, device="cpu")
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == bagel_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in vae.state_dict().items()} == vae_shapes(cfg["vae"])


def _eq(a, b):
    assert set(a) == set(b), set(a) ^ set(b)
    for k in a:
        if torch.is_tensor(a[k]):
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


def test_packers_bit_exact(golden):
    cfg = TINY
    model, _ = build_bagel(cfg, device="cpu")
    tok = StubTokenizer(cfg["llm"]["vocab_size"])
    ids = NEW_TOKEN_IDS_TINY
    ds, pdim = 16, 64
    g = golden("tiny_t2i")
    for kv, rope, prompts in (([0, 0], [0, 0], g["prompts"]), ([5, 0, 9], [3, 0, 2], ["x", "hello world", ""])):
        a = model.prepare_prompts(kv, rope, prompts, tok, ids)
        b = P.prepare_prompts(kv, rope, prompts, tok, ids)
        _eq(a[0], b[0]); assert a[1:] == b[1:]
    _eq(model.prepare_prompts([0, 0], [0, 0], g["prompts"], tok, ids)[0], g["prompt_inputs"])   # vs the reference itself
    for kv, rope, sizes in ((g["newlens"], g["newrope"], g["image_sizes"]), ([0], [0], [(1024, 1024)]), ([7, 0], [2, 5], [(48, 80), (16, 16)])):
        torch.manual_seed(42); a = model.prepare_vae_latent(kv, rope, sizes, ids)
        torch.manual_seed(42); b = P.prepare_vae_latent(kv, rope, sizes, ids, ds, 64, pdim)
        _eq(a, b)
        _eq(model.prepare_vae_latent_cfg(kv, rope, sizes), P.prepare_vae_latent_cfg(kv, rope, sizes, ds))
        _eq(model.prepare_start_tokens(kv, rope, ids), P.prepare_start_tokens(kv, rope, ids))
    torch.manual_seed(42)
    _eq(model.prepare_vae_latent(g["newlens"], g["newrope"], g["image_sizes"], ids), g["latent_inputs"])
    _eq(model.prepare_vae_latent_cfg([0, 0], [0, 0], g["image_sizes"]), g["cfg_inputs"])
    e = golden("tiny_editund")
    ident = lambda t: t  # noqa: E731
    a = model.prepare_vae_images([0], [0], [e["img_vae"]], ident, ids)
    _eq(a[0], e["vae_inputs"])
    b = model.prepare_vit_images(a[1], a[2], [e["img_vit"]], ident, ids)
    _eq(b[0], e["vit_inputs"])
    assert [a[1], b[1]] == e["lens"][:2] and [a[2], b[2]] == e["ropes"][:2]
    imgs = [torch.randn(3, 32, 48), torch.randn(3, 64, 16)]
    _eq(model.prepare_vae_images([3, 1], [1, 4], imgs, ident, ids, timestep=0)[0],
        P.prepare_vae_images([3, 1], [1, 4], imgs, ident, ids, ds, 64)[0])
    imgs = [torch.randn(3, 28, 42), torch.randn(3, 14, 14)]
    _eq(model.prepare_vit_images([3, 1], [1, 4], imgs, ident, ids)[0], P.prepare_vit_images([3, 1], [1, 4], imgs, ident, ids, 14, 10)[0])
    _eq(model.prepare_start_tokens(e["lens"][2], e["ropes"][2], ids), e["start_inputs"])


def test_flow_schedule_matches_reference_schedule():
    from bagel_amd.modeling.bagel import Bagel
    from oracle.bagel_oracle import flow_schedule
    for T, s in ((5, 3.0), (50, 3.0), (24, 1.0)):
        a, b = Bagel.flow_schedule(T, s), flow_schedule(T, s)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_split_structure_is_recovered_from_the_reference_masks():
    """qwen2_navit.splits_from_mask: the additive masks of prepare_attention_mask_per_sample (data_utils.py:72-103) decode
    back to an equivalent (split_lens, attn_modes); anything else is refused."""
    import pytest
    import torch
    from bagel_amd.modeling.bagel.qwen2_navit import splits_from_mask
    from oracle import bagel_oracle as O
    cases = [([6, 14, 9], ["causal", "full", "causal"]), ([5, 14, 14, 4, 10], ["causal", "full", "noise", "causal", "noise"]),
             ([3], ["causal"]), ([4, 1, 2], ["full", "noise", "causal"]), ([2, 3, 3], ["causal", "causal", "full"]),
             ([7, 7], ["noise", "noise"]), ([1, 1, 1], ["causal", "noise", "full"])]
    for lens, modes in cases:
        m = O.attention_mask_per_sample(lens, modes)
        l2, m2 = splits_from_mask(m)
        assert sum(l2) == sum(lens)
        assert torch.equal(O.attention_mask_per_sample(l2, m2), m), (lens, modes, l2, m2)
    bad = O.attention_mask_per_sample([4, 4], ["causal", "causal"])
    bad[1, 3] = 0.0                                   # a key from the future inside a causal split
    with pytest.raises(NotImplementedError):
        splits_from_mask(bad)


def test_install_as_reference_resolves_the_entry_script_imports():
    """The import lines of app.py:10-19 / eval/gen/gen_images_mp.py:11-19 / inferencer.py:10-11, verbatim, after
    bagel_amd.install_as_reference() -- in a fresh interpreter so the aliases cannot leak into this one."""
    import subprocess
    import sys
    code = (
        "import bagel_amd; bagel_amd.install_as_reference()\n"
        "from data.data_utils import add_special_tokens, pil_img2rgb\n"
        "from data.transforms import ImageTransform\n"
        "from inferencer import InterleaveInferencer\n"
        "from modeling.autoencoder import load_ae\n"
        "from modeling.bagel.qwen2_navit import NaiveCache\n"
        "from modeling.bagel import (BagelConfig, Bagel, Qwen2Config, Qwen2ForCausalLM, SiglipVisionConfig, SiglipVisionModel)\n"
        "from modeling.qwen2 import Qwen2Tokenizer\n"
        "from modeling.cache_utils.taylorseer import cache_init\n"
        "import bagel_amd.inferencer as I\n"
        "assert InterleaveInferencer is I.InterleaveInferencer and NaiveCache(3).num_layers == 3\n"
        "assert Qwen2Tokenizer.__name__ == 'Qwen2Tokenizer'\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_position_id_variants_product_equals_oracle():
    """get_flattened_position_ids_extrapolate / _interpolate (data/data_utils.py:53-69; BagelConfig.interpolate_pos): integers, bit-exact."""
    from bagel_amd.data import data_utils as D
    for (h, w, p, side) in ((64, 64, 16, 64), (1024, 1024, 16, 64), (980, 980, 14, 70), (42, 56, 14, 10), (48, 80, 16, 32), (16, 16, 16, 4),
                            (224, 448, 14, 70)):
        assert torch.equal(D.get_flattened_position_ids_extrapolate(h, w, p, side), P.position_ids_extrapolate(h, w, p, side))
        assert torch.equal(D.get_flattened_position_ids_interpolate(h, w, p, side), P.position_ids_interpolate(h, w, p, side))
    assert D.get_flattened_position_ids_extrapolate(32, 48, 16, 8).tolist() == [0, 1, 2, 8, 9, 10]


def test_bench_roofline_traffic_sources_resolve():
    """bench.py's roofline.traffic fields come from the committed PMC summary (a profiler cannot run inside the timed region): the
    keys the bench looks up must exist, or the field silently degrades to null."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = bench.pmc_traffic("gemm_pq_kernel<*>")
    assert t is not None and 0.5e9 < t < 50e9            # HBM-side bytes per launch of the dominant GEMM (0.61 GB algorithmic)
    d = bench.pmc_decode_traffic()
    assert d is not None and 14.0e9 < d < 16.0e9         # bytes of one decode step (14.43 GB algorithmic)
    assert bench.pmc_traffic("no_such_kernel") is None


def test_vae_launch_wrappers_match_the_header(monkeypatch):
    """The fp32 VAE wrappers of bagel_amd.ops hand the C ABI exactly the argument list include/bagel_hip.h declares (count and
    kinds: pointer / integer / float) -- checked without a GPU by swapping the library for a recorder that validates every call
    against the parsed prototypes (the other CPU tests replace these wrappers wholesale, so nothing else executes their bodies)."""
    import ctypes
    from bagel_amd import _lib, ops
    protos = _lib.parse_header()
    calls = []

    class Recorder:
        def __getattr__(self, name):
            restype, argtypes = protos[name]

            def fn(*args):
                assert len(args) == len(argtypes), (name, len(args), len(argtypes))
                for i, (a, t) in enumerate(zip(args, argtypes)):
                    if t is ctypes.c_void_p:
                        assert a is None or isinstance(a, int), (name, i, a)
                    elif t in (ctypes.c_int32, ctypes.c_int64):
                        assert isinstance(a, int) and not isinstance(a, bool), (name, i, a)
                    else:
                        assert isinstance(a, float), (name, i, a)
                calls.append(name)
                return 0
            return fn
    monkeypatch.setattr(ops, "lib", lambda: Recorder())
    monkeypatch.setattr(ops, "_ptr", lambda t: None if t is None else t.data_ptr())
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    monkeypatch.setattr(ops, "_req", lambda t, dtype, name: None)
    f = lambda *s: torch.zeros(*s)  # noqa: E731
    ops.conv_gemm_f32(f(2, 4, 4, 32), 32, f(8, 288), 288, f(8), f(2, 4, 4, 8), f(2, 4, 4, 8), 8, 2, 4, 4, 32, 4, 4, 8, 1)
    ops.conv_gemm_f32(f(6, 5), 5, f(7, 5), 5, None, None, f(6, 7), 7, 1, 1, 6, 5, 1, 6, 7, 0)
    ops.groupnorm_f32(f(1, 4, 4, 32), f(1, 4, 4, 32), f(4160), f(32), f(32), 1, 16, 32, 32, 1e-6, True)
    ops.softmax_rows_f32(f(3, 5), 5, 3, 5, 0.25)
    ops.vae_reparam_f32(f(16, 8), f(16, 4), f(16, 4), 16, 4, 0.3611, 0.1159)
    ops.vae_unscale_f32(f(64), f(64), 64, 0.3611, 0.1159)
    assert calls == ["bagel_conv_gemm_f32"] * 2 + ["bagel_groupnorm_f32", "bagel_softmax_rows_f32", "bagel_vae_reparam_f32", "bagel_vae_unscale_f32"]
    # the bf16-autocast VAE's wrappers and the batched-decode projection
    del calls[:]
    monkeypatch.setattr(ops, "_req_any_stride", lambda t, dtype, name: None)
    b = lambda *s: torch.zeros(*s, dtype=torch.bfloat16)  # noqa: E731
    ops.conv_gemm_bf16(b(2, 4, 4, 64), 64, b(8, 576), 576, b(8), b(2, 4, 4, 8), b(2, 4, 4, 8), 8, 2, 4, 4, 64, 4, 4, 8, 1)
    ops.conv_gemm_bf16(b(6, 8), 8, b(7, 8), 8, None, None, f(6, 8), 8, 1, 1, 6, 8, 1, 6, 7, 0)
    ops.groupnorm_bf16(b(1, 4, 4, 32), b(1, 4, 4, 32), f(65664), f(32), f(32), 1, 16, 32, 32, 1e-6, True)
    ops.softmax_rows_bf16(f(3, 8), b(3, 8), 3, 5, 0.25)
    ops.vae_reparam_bf16(b(16, 8), b(16, 4), b(16, 4), 16, 4, 0.3611, 0.1159)
    ops.chw_bf16_to_u8(b(3, 4, 6))
    ops.gemv_mb(b(4, 128), b(64, 128), b(4, 64), M=4)
    assert calls == ["bagel_conv_gemm_bf16"] * 2 + ["bagel_groupnorm_bf16", "bagel_softmax_rows_bf16", "bagel_vae_reparam_bf16", "bagel_chw_bf16_to_u8",
                     "bagel_gemv_mb_bf16"]


def test_ops_gemm_routing_of_the_vit_epilogues_and_small_row_counts(monkeypatch):
    """Which kernel ``ops.gemm`` hands a launch to, pinned without a GPU (the library is swapped for a recorder): the ViT's epilogues on the persistent kernel's
    SGPR-base form (variant 5) since round 6 -- except bias + residual on fewer than half a round of tiles with a short K, which stays on the 128 x 128 kernel
    (measured: profiles/r06_vit_epilogues.log) --, 2..32 dense rows on the register-resident weight stream, 33..64 on the skinny MFMA kernel."""
    from bagel_amd import ops
    seen = []

    class Recorder:
        def __getattr__(self, name):
            def fn(*args):
                seen.append((name, args))
                return 0
            return fn
    monkeypatch.setattr(ops, "lib", lambda: Recorder())
    monkeypatch.setattr(ops, "_ptr", lambda t: None if t is None else 4096)
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    monkeypatch.setattr(ops, "_req", lambda t, dtype, name: None)
    monkeypatch.setattr(ops, "_gemm_workspace", lambda device: torch.zeros(16))
    monkeypatch.setattr(ops, "_ws_key", lambda device: ("cpu", 0))
    monkeypatch.setattr(torch.Tensor, "data_ptr", lambda self: 4096, raising=False)
    b = lambda *s: torch.empty(*s, dtype=torch.bfloat16)  # noqa: E731

    def variant_of(M, N, K, **kw):
        del seen[:]
        ops.gemm(b(M, K), b(N, K), b(M, N), **kw)
        name, args = seen[-1]
        return name, (args[20] if name == "bagel_gemm_bf16_ws" else None)
    bias = lambda N: dict(bias0=b(N))  # noqa: E731
    # SigLIP at a 980^2 image (M = 4900): fc1 (bias + GELU) and fc2 (bias + residual, K = 4352) -> variant 5; the out projection (K = 2048 / 1152) -> 128 x 128
    assert variant_of(4900, 4352, 1152, epilogue=ops.EPI_GELU_TANH, **bias(4352)) == ("bagel_gemm_bf16_ws", 5)
    assert variant_of(4900, 1152, 4352, residual=b(4900, 1152), **bias(1152)) == ("bagel_gemm_bf16_ws", 5)
    assert variant_of(4900, 1152, 2048, residual=b(4900, 1152), **bias(1152)) == ("bagel_gemm_bf16_ws", 0)
    assert variant_of(4900, 1152, 1152, residual=b(4900, 1152), **bias(1152)) == ("bagel_gemm_bf16_ws", 0)
    # the earlier routing behind the A/B knob: few-tile bias + residual -> 128 x 128, GELU -> the one-tile ping-pong kernel
    monkeypatch.setattr(ops, "GEMM_VIT_EPI", False)
    assert variant_of(4900, 1152, 4352, residual=b(4900, 1152), **bias(1152)) == ("bagel_gemm_bf16_ws", 0)
    assert variant_of(4900, 4352, 1152, epilogue=ops.EPI_GELU_TANH, **bias(4352)) == ("bagel_gemm_bf16_ws", 3)
    monkeypatch.setattr(ops, "GEMM_VIT_EPI", True)
    # the LLM's projections keep the persistent kernel at every M >= 2048
    assert variant_of(4902, 3584, 18944, residual=b(4902, 3584)) == ("bagel_gemm_bf16_ws", 5)
    assert variant_of(32784, 4608, 3584, **bias(4608)) == ("bagel_gemm_bf16_ws", 5)
    # row counts: 2..32 -> gemv_mb (one or two blocks of request rows), 33..64 -> skinny, 1 -> the lane-FMA gemv
    assert variant_of(16, 4608, 3584, **bias(4608))[0] == "bagel_gemv_mb_bf16"
    assert variant_of(32, 4608, 3584, **bias(4608))[0] == "bagel_gemv_mb_bf16"
    assert variant_of(34, 4608, 3584, **bias(4608))[0] == "bagel_gemm_skinny_bf16"
    assert variant_of(1, 4608, 3584, **bias(4608))[0] == "bagel_gemv_bf16"
    monkeypatch.setattr(ops, "MB_MAX_ROWS", 16)
    assert variant_of(32, 4608, 3584, **bias(4608))[0] == "bagel_gemm_skinny_bf16"


def test_gemv_mb_workspace_mirror_matches_the_library():
    """``ops.mb_workspace_floats`` / ``ops._mb_slices`` mirror csrc/gemv_mb.hip's geometry (the decode session sizes its K-slice workspace with the Python side,
    the launcher checks it against its own arithmetic): the library's ``bagel_gemv_mb_workspace_bytes`` is a host-only function -- no GPU needed -- and must
    agree for every row length the models use, incl. the round-6 bound of the two-block (17..32 rows) form: slices of at most 8 x 14 steps, 32-row slabs."""
    import ctypes
    from bagel_amd import _lib, ops
    L = _lib.lib()
    for N, K in [(3584, 18944), (3584, 3584), (4608, 3584), (37888, 3584), (152064, 3584), (64, 128), (528, 4864), (160, 4896), (1024, 8192), (256, 7168)]:
        out = ctypes.c_int64(-1)
        assert L.bagel_gemv_mb_workspace_bytes(N, K, ctypes.byref(out)) == 0
        assert out.value == ops.mb_workspace_floats(N, K) * 4, (N, K, out.value)
    # one slice up to 8 x 19 steps at <= 16 rows, up to 8 x 14 steps (K = 3584) at 17..32 rows; beyond that the minimum slice count
    assert ops._mb_slices(3584, 16) == (14, 1) and ops._mb_slices(3584, 32) == (14, 1)
    assert ops._mb_slices(4864, 16) == (19, 1) and ops._mb_slices(4864, 17) == (None, 2)
    assert ops._mb_slices(18944, 16) == (None, 4) and ops._mb_slices(18944, 32) == (None, 6)
    assert ops.MB_MAX_ROWS == 32


def test_decode_engine_isa_invariants():
    """csrc/engine.hip relies on three things hipcc does not promise: (1) nothing spills to scratch (a scratch reload is a vector-memory load: its wait would drain the
    loaders' counted LDS-DMA queue), (2) the compiler has no use of M0 of its own in this kernel (the DMA statements set M0 and do NOT restore it), (3) the loaders' only
    vector-memory waits are the counted ones written in the source.  Checked on the ISA hipcc emits for gfx950 (cross-compiles without a GPU)."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from bagel_amd.build import FLAGS
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "engine.s")
        flags = [f for f in FLAGS if f != "-fPIC"]
        r = subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", os.path.join(root, "bagel_amd", "csrc", "engine.hip"), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(out).read()
    meta = asm[asm.index("amdhsa.kernels"):]
    assert re.search(r"\.private_segment_fixed_size:\s*0\b", meta), "decode_engine_kernel uses scratch"
    assert re.search(r"\.vgpr_spill_count:\s*0\b", meta)
    inside = False
    for line in asm.splitlines():
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside and re.search(r"\bm0\b", line) and not line.lstrip().startswith((";", ".")):
            raise AssertionError(f"hipcc uses M0 outside the DMA statements: {line.strip()}")
    assert asm.count("global_load_lds_dwordx4") >= 8
    assert "buffer_wbl2" not in asm and "buffer_inv" not in asm, "an agent-scope fence crept into the engine (its hand-offs are write-through stores + sc1 loads)"


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md's table ("entry point | replaces") is the reader's map of the C ABI: every prototype of include/bagel_hip.h must appear in it (the header
    and the library are compared symbol for symbol by the build check; this keeps the document in step with both)."""
    from bagel_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [n for n in sorted(_lib.parse_header()) if n not in doc]
    assert not missing, f"INTEGRATION.md does not mention {missing}"
