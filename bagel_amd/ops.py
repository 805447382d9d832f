"""Tensor-level wrappers over the C ABI (include/bagel_hip.h).  PyTorch is used for device memory and the
current HIP stream only; every op below is one or two launches of a hand-written gfx950 kernel.

All tensors must live on the GPU; there is no CPU path (see _lib.py)."""
import ctypes
import os

import torch

from ._lib import BagelHipError, check, lib

BF16 = torch.bfloat16

EPI_NONE, EPI_GELU_TANH, EPI_SILU, EPI_SWIGLU16 = 0, 1, 2, 3
RENORM_MODES = {"global": 0, "channel": 1, "text_channel": 2}


def _ptr(t):
    """Device address of ``t`` (None -> NULL).  Every pointer handed to the C ABI goes through here, so a host tensor
    can never reach a kernel."""
    if t is None:
        return None
    if not t.is_cuda:
        raise BagelHipError(f"expected a GPU tensor, got {t.device} {tuple(t.shape)} {t.dtype} (bagel_amd has no CPU path)")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu_bf16(t, what):
    """The engines run bf16 weights resident on an MI355X; anything else is refused before a kernel could see it."""
    if not t.is_cuda or t.dtype != BF16:
        raise BagelHipError(f"{what}: bagel_amd runs bf16 weights on an MI355X (got {t.device}, {t.dtype}): "
                            "call model.to('cuda', torch.bfloat16) first")


def _req(t, dtype, name):
    if not t.is_cuda:
        raise BagelHipError(f"{name}: expected a GPU tensor (bagel_amd has no CPU path)")
    if t.dtype != dtype:
        raise BagelHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise BagelHipError(f"{name}: innermost dimension must be contiguous")


def _req_any_stride(t, dtype, name):
    if not t.is_cuda:
        raise BagelHipError(f"{name}: expected a GPU tensor (bagel_amd has no CPU path)")
    if t.dtype != dtype:
        raise BagelHipError(f"{name}: expected {dtype}, got {t.dtype}")


def _ld(t):
    return t.stride(0) if t.dim() == 2 else t.shape[-1]


def default_gemm_variant(M, N, K):
    v = os.environ.get("BAGEL_GEMM_VARIANT")
    if v is not None:
        return int(v)
    # measured on MI355X at the denoise shapes (M = 16392; profiles/r01_kernel_probe*.json, r01_gemm_persistent.log): the
    # 256x256 two-group ping-pong kernel wins on every projection, and its persistent form (variant 4: one workgroup per CU
    # walks the tile list, next tile's first DMA under the current epilogue) is bit-identical to the one-tile-per-workgroup
    # form (variant 3) and 3.5 % faster over the four GEMMs of a layer; the library falls back to variant 3 for the epilogue
    # combinations variant 4 does not instantiate and to the plain 256x256 kernel (variant 1) when K % 64 != 0.  Small-M
    # launches (prefill of a short prompt, decode, time embedder) keep the 128x128 tile so the grid still covers the chip.
    if M >= 2048 and N >= 512:
        # fewer than half a round of 256x256 tiles on a long K that the ping-pong kernel cannot take (K % 64 != 0 -> the plain 256x256 kernel):
        # SigLIP fc2 at a 980^2 image, 100 tiles, K = 4304 -- the 128x128 kernel puts 351 workgroups on the chip instead, 92 vs 136 us
        # (tools/gemm_prefill_shapes.py).  Only that measured case: the LLM's prefill / edit projections (K % 64 == 0) stay on variant 4 at
        # every M, so neighbouring prompt lengths share one kernel and one accumulation order.
        if -(-M // 256) * -(-N // 256) <= 128 and K >= 2048 and K % 64 != 0:
            return 0
        # ... and a SHORT K on fewer than half a round of tiles: SigLIP's out projection at a 980^2 image (100 tiles, K = 1152) -- 29.3 us on the 128x128 kernel
        # against 37.7 us on the persistent one, whose pipeline fill and K-split reduce pass weigh on an 18-k-tile loop (round 6, tools/gemm_prefill_shapes.py).
        # (The MLP width of that tower is padded to the k-tile in its packed weights, siglip_navit.py: fc2 at K = 4352 stays on variant 4, 66 vs 91 us.)
        if -(-M // 256) * -(-N // 256) <= 128 and K <= 1536:
            return 0
        return 4
    return 0


def gemm(A, W0, C, *, bias0=None, a_rows0=None, c_rows0=None, M0=None, W1=None, bias1=None, a_rows1=None, c_rows1=None,
         M1=0, residual=None, epilogue=EPI_NONE, variant=None, splitk=True):
    """C = A @ W^T (+bias)(act)(+residual); see bagel_gemm_bf16.  A:[*,K] W:[N,K] C:[*,N or N/2]."""
    _req(A, BF16, "gemm.A"); _req(W0, BF16, "gemm.W0"); _req(C, BF16, "gemm.C")
    N, K = W0.shape
    if A.shape[-1] != K:
        raise BagelHipError(f"gemm: A has K={A.shape[-1]}, W has K={K}")
    if M0 is None:
        M0 = a_rows0.numel() if a_rows0 is not None else A.shape[0]
    if W1 is not None:
        _req(W1, BF16, "gemm.W1")
        assert W1.shape == W0.shape and W1.stride(0) == W0.stride(0)
    for r in (a_rows0, c_rows0, a_rows1, c_rows1):
        if r is not None:
            _req(r, torch.int32, "gemm.rows")
    if residual is not None:
        _req(residual, BF16, "gemm.residual")
    dense1 = variant is None and W1 is None and M1 == 0 and a_rows0 is None and c_rows0 is None
    if dense1 and 2 <= M0 <= MB_MAX_ROWS and gemv_mb_supported(A, W0, C, bias0, residual, epilogue, False, M=M0):
        # 2..32 rows (batched decode steps, the marker rows of a denoise forward, a short text prefill): weight stream with the activations in registers
        return gemv_mb(A, W0, C, bias=bias0, residual=residual, epilogue=epilogue, M=M0)
    if (dense1 and 2 <= M0 <= SKINNY_MAX_ROWS and K % 32 == 0 and N % (32 if epilogue == EPI_SWIGLU16 else 16) == 0
            and A.data_ptr() % 16 == 0 and W0.data_ptr() % 16 == 0 and C.data_ptr() % 8 == 0 and _ld(C) % 4 == 0
            and (bias0 is None or bias0.data_ptr() % 8 == 0)
            and (residual is None or (residual.data_ptr() % 8 == 0 and _ld(residual) % 4 == 0))):
        # 2..64 rows: still a weight stream, but through the MFMA with no LDS tile (skinny.hip)
        return gemm_skinny(A, W0, C, bias=bias0, residual=residual, epilogue=epilogue, M=M0)
    if (dense1 and 0 < M0 <= GEMV_MAX_ROWS
            and N % 2 == 0 and K * 2 <= GEMV_MAX_K_BYTES and A.data_ptr() % 16 == 0 and W0.data_ptr() % 16 == 0
            and C.data_ptr() % 4 == 0 and _ld(C) % 2 == 0
            and (residual is None or (residual.data_ptr() % 4 == 0 and _ld(residual) % 2 == 0))):
        # one row (or a few that the MFMA path cannot take): lane-FMA weight streaming (decode.hip)
        return gemv(A, W0, C, bias=bias0, residual=residual, epilogue=epilogue, M=M0)
    if variant is None:
        variant = default_gemm_variant(M0 + M1, N, K)
        # The ViT's epilogues -- bias + residual (out / fc2), bias + GELU-tanh (fc1) -- exist in the persistent kernel's SGPR-base form since round 6 (variant 5 below,
        # with the K-split of leftover tiles): fc1 at a 980^2 image 101 -> 87 us, fc2 (100 tiles, K = 4352) 72 (128 x 128 kernel) -> 66 us.  On fewer than half a
        # round of tiles with a SHORT K the 128 x 128 kernel still covers the chip better than two half-length parts per tile + a reduce pass: the out projection
        # (K = 2048 with the padded heads) 44.6 vs 47.2 us (tools/gemm_prefill_shapes.py, profiles/r06_vit_epilogues.log).
        # BAGEL_GEMM_VIT_EPI=0 restores the earlier routing for a same-box A/B (bias + residual on few tiles -> 128 x 128, the rest -> the library's fallback, variant 3).
        vit_few = variant == 4 and bias0 is not None and residual is not None and W1 is None and -(-(M0 + M1) // 256) * -(-N // 256) <= 128
        if vit_few and (K <= 3072 or not GEMM_VIT_EPI):
            variant = 0
        elif variant == 4 and not GEMM_VIT_EPI and bias0 is not None and W1 is None and (residual is not None or epilogue == EPI_GELU_TANH):
            variant = 3
        # variant 5 = variant 4 with SGPR-base DMA addresses (one address register per LDS-DMA instruction: the persistent kernel is
        # DMA-issue bound, gate+up 1 304 -> 1 411 TFLOP/s at M = 32 768): legal when every operand row lies within 4 GiB of the operand's
        # base pointer, which the tensors' extents tell here (gathered rows index into A, so A's extent bounds them).  Same arithmetic,
        # bit-identical results.  BAGEL_GEMM_SADDR=0 keeps variant 4 (same-box A/B).
        if variant == 4 and GEMM_SADDR and A.shape[0] * _ld(A) * 2 < 2 ** 32 and N * W0.stride(0) * 2 < 2 ** 32:
            variant = 5
    ws = _gemm_workspace(A.device) if (variant in (4, 5) and splitk and GEMM_SPLITK) else None
    check(lib().bagel_gemm_bf16_ws(_ptr(A), _ld(A), _ptr(W0), _ptr(bias0), _ptr(a_rows0), _ptr(c_rows0), M0,
                                   _ptr(W1), _ptr(bias1), _ptr(a_rows1), _ptr(c_rows1), M1, W0.stride(0),
                                   _ptr(residual), _ld(residual) if residual is not None else 0, _ptr(C), _ld(C),
                                   N, K, epilogue, variant, _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream()), "bagel_gemm_bf16_ws")
    return C


# K-split of a nearly empty last round of the persistent GEMM (bagel_gemm_bf16_ws): one fp32 workspace per device, 64 MB = 256 partial
# tiles of 256 x 256 (every leftover tile cut into at most 256 / leftover parts).  BAGEL_GEMM_SPLITK=0 switches it off (same-box A/B).
GEMM_SPLITK = os.environ.get("BAGEL_GEMM_SPLITK", "1") != "0"
GEMM_SADDR = os.environ.get("BAGEL_GEMM_SADDR", "1") != "0"
GEMM_VIT_EPI = os.environ.get("BAGEL_GEMM_VIT_EPI", "1") != "0"
_GEMM_WS = {}


def _ws_key(device):
    """Workspaces are keyed by (device, stream): two streams (or threads on their own streams) launching at the same time must not share
    partial tiles, and a buffer is only ever replaced by a larger one while no launch on ITS stream can still be reading it (launches on
    one stream are ordered)."""
    return (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)


def _gemm_workspace(device):
    key = _ws_key(device)
    ws = _GEMM_WS.get(key)
    if ws is None:
        # 256 partial tiles of 256 x 256 fp32 = 64 MB: the shipped rule cuts `rest` leftover tiles into at most wgs / rest parts.  The round-5 A/B knobs
        # (multi-pass splits: the 3-way split of 160 leftover tiles needs 480 partial tiles) get twice that.
        tiles = 512 if (os.environ.get("BAGEL_GEMM_SPLIT_POLICY", "0") != "0" or os.environ.get("BAGEL_GEMM_SPLIT_FORCE")) else 256
        ws = _GEMM_WS[key] = torch.empty(tiles * 256 * 256, dtype=torch.float32, device=device)
    return ws


GEMV_MAX_ROWS = 8
GEMV_MAX_K_BYTES = 144 * 1024
SKINNY_MAX_ROWS = 64
MB_MAX_ROWS = 32
# BAGEL_GEMV_MB=0: 2..32 rows go back to rmsnorm + skinny.hip (same-box A/B of the batched decode step); BAGEL_GEMV_MB_ROWS=16: only 17..32 rows do
# (the round-5 routing: A/B of the two-block form)
GEMV_MB = os.environ.get("BAGEL_GEMV_MB", "1") != "0"
MB_MAX_ROWS = min(MB_MAX_ROWS, int(os.environ.get("BAGEL_GEMV_MB_ROWS", MB_MAX_ROWS)))
_MB_WS = {}


def _mb_slices(K, M=1):
    """(steps per wave if the row runs as ONE K slice else None, minimum K slices) of bagel_gemv_mb_bf16 -- mirrors mb_geometry in
    csrc/gemv_mb.hip (rows of more than 8 x 19 steps of 32 are cut over workgroups; the launcher picks the slice count between the
    minimum and 4 x the minimum so that the column blocks divide evenly over the CUs).  M > 16 (two blocks of 16 request rows per wave): the
    longest instantiation is 14 steps, so one slice holds 8 x 14 steps (K = 3584)."""
    nsteps = K // 32
    nsm = 14 if M > 16 else 19
    if nsteps <= 8 * nsm:
        return -(-nsteps // 8), 1
    return None, -(-nsteps // (8 * nsm))


def mb_workspace_floats(N, K):
    """fp32 elements of the K-slice workspace bagel_gemv_mb_bf16 may need for an [N, K] weight at ANY row count (bagel_gemv_mb_workspace_bytes: the bound
    of the 32-row form -- shorter slices, 32-row slabs); 0: the row runs in one slice whatever M is."""
    per, ks = _mb_slices(K, 32)
    return 0 if per is not None else 4 * ks * 32 * N


def gemv_mb_supported(A, W, C, bias, residual, epilogue, has_norm, M=None):
    if not GEMV_MB:
        return False
    N, K = W.shape
    if K % 32 or N % (32 if epilogue == EPI_SWIGLU16 else 16):
        return False
    M = A.shape[0] if M is None else M
    if not 1 <= M <= MB_MAX_ROWS:
        return False
    per, ks = _mb_slices(K, M)
    if per is not None:
        ns = next(n for n in (4, 10, 14, 19) if per <= n)
        if K // 32 < ns:
            return False
    elif has_norm or epilogue == EPI_SWIGLU16:
        return False
    if epilogue == EPI_SWIGLU16 and (bias is not None or residual is not None):
        return False
    ok = A.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0 and _ld(A) % 8 == 0 and W.stride(0) % 8 == 0 and C.data_ptr() % 8 == 0 and _ld(C) % 4 == 0
    ok = ok and (bias is None or bias.data_ptr() % 8 == 0)
    return ok and (residual is None or (residual.data_ptr() % 8 == 0 and _ld(residual) % 4 == 0))


def gemv_mb(A, W, C, *, bias=None, residual=None, epilogue=EPI_NONE, M=None, norm_w=None, eps=0.0, workspace=None):
    """C[M <= 32, N] = norm(A) W^T with the epilogues of ``gemm``: the batched-decode weight stream (bagel_gemv_mb_bf16): the waves of a
    workgroup partition K and hold their activation fragments in registers (one or two blocks of 16 rows), optional fused RMSNorm.  Long rows
    (K > 3584 at 7B: the down projection) run as K slices through an fp32 workspace (``workspace`` or a per-(device, stream) one) and a second, tiny launch."""
    _req(A, BF16, "gemv_mb.A"); _req(W, BF16, "gemv_mb.W"); _req(C, BF16, "gemv_mb.C")
    N, K = W.shape
    if A.shape[-1] != K:
        raise BagelHipError(f"gemv_mb: A has K={A.shape[-1]}, W has K={K}")
    if M is None:
        M = A.shape[0]
    if residual is not None:
        _req(residual, BF16, "gemv_mb.residual")
    if norm_w is not None:
        _req(norm_w, BF16, "gemv_mb.norm_w")
    need = mb_workspace_floats(N, K) * 4                # bagel_gemv_mb_workspace_bytes
    ws = workspace
    if need and ws is None:
        key = _ws_key(A.device)
        ws = _MB_WS.get(key)
        if ws is None or ws.numel() * 4 < need:
            ws = _MB_WS[key] = torch.empty(max(need // 4, 1 << 20), dtype=torch.float32, device=A.device)
    check(lib().bagel_gemv_mb_bf16(_ptr(A), _ld(A), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual),
                                   _ld(residual) if residual is not None else 0, _ptr(C), _ld(C), _ptr(norm_w), float(eps), M, N, K,
                                   epilogue, _ptr(ws) if need else None, ws.numel() * 4 if need else 0, _stream()), "bagel_gemv_mb_bf16")
    return C


def gemm_skinny(A, W, C, *, bias=None, residual=None, epilogue=EPI_NONE, M=None):
    """C[M <= 64, N] = A W^T with the epilogues of ``gemm``; every wave streams 16 weight rows from HBM straight into the
    MFMA (bagel_gemm_skinny_bf16)."""
    _req(A, BF16, "gemm_skinny.A"); _req(W, BF16, "gemm_skinny.W"); _req(C, BF16, "gemm_skinny.C")
    N, K = W.shape
    if A.shape[-1] != K:
        raise BagelHipError(f"gemm_skinny: A has K={A.shape[-1]}, W has K={K}")
    if M is None:
        M = A.shape[0]
    if residual is not None:
        _req(residual, BF16, "gemm_skinny.residual")
    check(lib().bagel_gemm_skinny_bf16(_ptr(A), _ld(A), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual),
                                       _ld(residual) if residual is not None else 0, _ptr(C), _ld(C), M, N, K, epilogue, _stream()),
          "bagel_gemm_skinny_bf16")
    return C


def gemv(A, W, C, *, bias=None, residual=None, epilogue=EPI_NONE, M=None, norm_w=None, eps=0.0):
    """Skinny GEMM (M <= a few rows) with optional fused RMSNorm of the A rows; see bagel_gemv_bf16."""
    _req(A, BF16, "gemv.A"); _req(W, BF16, "gemv.W"); _req(C, BF16, "gemv.C")
    N, K = W.shape
    if A.shape[-1] != K:
        raise BagelHipError(f"gemv: A has K={A.shape[-1]}, W has K={K}")
    if M is None:
        M = A.shape[0]
    if residual is not None:
        _req(residual, BF16, "gemv.residual")
    if norm_w is not None:
        _req(norm_w, BF16, "gemv.norm_w")
    check(lib().bagel_gemv_bf16(_ptr(A), _ld(A), _ptr(W), W.stride(0), _ptr(bias), _ptr(residual),
                                _ld(residual) if residual is not None else 0, _ptr(C), _ld(C), _ptr(norm_w), float(eps),
                                M, N, K, epilogue, _stream()), "bagel_gemv_bf16")
    return C


def decode_engine_workgroups():
    n = lib().bagel_decode_engine_workgroups()
    if n <= 0:
        check(n, "bagel_decode_engine_workgroups")
    return n


def decode_engine_sync_words(n_phases):
    """uint32 words of hand-off flags one bagel_decode_engine_bf16 launch of ``n_phases`` phases needs (zero at launch)."""
    n = lib().bagel_decode_engine_sync_bytes(int(n_phases))
    if n < 0:
        check(n, "bagel_decode_engine_sync_bytes")
    return n // 4


def decode_engine_supported(phases):
    """The shapes the persistent decode engine serves (csrc/engine.hip): what bagel_decode_engine_bf16 would refuse is refused here
    without a launch, so callers can fall back to the launch form."""
    if not 1 <= len(phases) <= 4:
        return False
    n_wg = lib().bagel_decode_engine_workgroups()
    if n_wg <= 0:
        return False
    slot = max_k = 0
    for ph in phases:
        N, K = ph["W"].shape
        if K % 8 or N % 2 or ph["W"].stride(0) % 8:
            return False
        if ph.get("norm_w") is not None and K > 4096:
            return False
        split = ph.get("norm_w") is None and K >= 8192 and N // 2 <= 8192
        groups = -(-(K // 8) // 64)
        unit_groups = -(-groups // 4) if split else groups
        if 2 * unit_groups > 60:
            return False
        if split and -(-(N // 2) // n_wg) > 32:                                       # ENG_MAX_DPAIRS split-K pairs per workgroup
            return False
        if ph.get("epilogue", EPI_NONE) == EPI_SWIGLU16 and (N % 32 or ph.get("bias") is not None or ph.get("residual") is not None):
            return False
        for key, align in (("A", 16), ("W", 16), ("norm_w", 16), ("C", 4), ("residual", 4)):
            t = ph.get(key)
            if t is not None and t.data_ptr() % align:
                return False
        slot, max_k = max(slot, 2 * unit_groups * 1024), max(max_k, K)
    # the LDS ring beside the activation vector and the sync block (ENG_LDS_MAX, ENG_SYNC_BYTES, ENG_MAX_SLOTS; two loaders own alternate slots)
    nslot = min((160 * 1024 - (max_k * 2 + 1023) // 1024 * 1024 - 2560) // slot, 8)
    nslot -= nslot % 2
    return nslot >= 4


def decode_engine(phases, eps, sync_ws, status, trace=None):
    """ONE persistent launch for a chain of batch-1 projections; see bagel_decode_engine_bf16.  ``phases``: list of dicts with the keys of
    ``gemv`` -- A, W, C and optionally bias, residual, norm_w, epilogue -- where A of phase i > 0 is C of phase i - 1.  ``sync_ws``: int32 /
    uint32 words, at least decode_engine_sync_words(len(phases)), ZERO at launch; ``status``: 4 words, checked by the caller."""
    n = len(phases)
    ptrs = (ctypes.c_void_p * (6 * n))()
    dims = (ctypes.c_int64 * (4 * n))()
    for i, ph in enumerate(phases):
        A, W, C = ph["A"], ph["W"], ph["C"]
        _req(A, BF16, "decode_engine.A"); _req(W, BF16, "decode_engine.W"); _req(C, BF16, "decode_engine.C")
        N, K = W.shape
        if A.numel() != K:
            raise BagelHipError(f"decode_engine: phase {i}: A has {A.numel()} elements, W has K={K} (one activation row)")
        for key in ("bias", "residual", "norm_w"):
            if ph.get(key) is not None:
                _req(ph[key], BF16, f"decode_engine.{key}")
        ptrs[6 * i + 0] = _ptr(A)
        ptrs[6 * i + 1] = _ptr(W)
        ptrs[6 * i + 2] = _ptr(ph.get("bias"))
        ptrs[6 * i + 3] = _ptr(ph.get("norm_w"))
        ptrs[6 * i + 4] = _ptr(ph.get("residual"))
        ptrs[6 * i + 5] = _ptr(C)
        dims[4 * i + 0], dims[4 * i + 1], dims[4 * i + 2], dims[4 * i + 3] = N, K, W.stride(0), int(ph.get("epilogue", EPI_NONE))
    if sync_ws.numel() * sync_ws.element_size() < 4 * decode_engine_sync_words(n) or sync_ws.element_size() != 4:
        raise BagelHipError("decode_engine: sync_ws is smaller than decode_engine_sync_words(len(phases)) 4-byte words")
    if status.numel() < 4 or status.element_size() != 4:
        raise BagelHipError("decode_engine: status needs 4 4-byte words")
    if trace is not None:          # diagnostic: int64 [workgroups, 4, 16] event times, see bagel_decode_engine_traced_bf16
        if trace.numel() < decode_engine_workgroups() * 64 or trace.element_size() != 8:
            raise BagelHipError("decode_engine: trace needs workgroups x 4 x 16 8-byte words")
        check(lib().bagel_decode_engine_traced_bf16(ctypes.addressof(ptrs), ctypes.addressof(dims), n, float(eps), _ptr(sync_ws), _ptr(status),
                                                    _ptr(trace), _stream()), "bagel_decode_engine_traced_bf16")
        return
    check(lib().bagel_decode_engine_bf16(ctypes.addressof(ptrs), ctypes.addressof(dims), n, float(eps), _ptr(sync_ws), _ptr(status),
                                         _stream()), "bagel_decode_engine_bf16")


def quantize_rows_i8(W):
    """bf16 [N, K] -> (u8 [N, K] = round(W / s) + 128, fp32 s [N] = rowwise absmax / 127); see bagel_quantize_rows_i8."""
    _req(W, BF16, "quantize_rows_i8.W")
    N, K = W.shape
    q = torch.empty((N, K), dtype=torch.uint8, device=W.device)
    s = torch.empty((N,), dtype=torch.float32, device=W.device)
    check(lib().bagel_quantize_rows_i8(_ptr(W), W.stride(0), _ptr(q), q.stride(0), _ptr(s), N, K, _stream()), "bagel_quantize_rows_i8")
    return q, s


def quantize_rows_fp8(X, q=None, scale=None):
    """bf16 [rows, K] -> (u8 [rows, K] holding OCP e4m3 bytes, fp32 scale [rows] = rowwise absmax / 448); see bagel_quantize_rows_fp8.
    ``q`` / ``scale``: optional preallocated outputs (the activation buffers of the engine)."""
    _req(X, BF16, "quantize_rows_fp8.X")
    rows, K = X.shape
    if q is None:
        q = torch.empty((rows, K), dtype=torch.uint8, device=X.device)
    if scale is None:
        scale = torch.empty((rows,), dtype=torch.float32, device=X.device)
    _req(q, torch.uint8, "quantize_rows_fp8.q"); _req(scale, torch.float32, "quantize_rows_fp8.scale")
    check(lib().bagel_quantize_rows_fp8(_ptr(X), X.stride(0), _ptr(q), q.stride(0), _ptr(scale), rows, K, _stream()), "bagel_quantize_rows_fp8")
    return q, scale


def rmsnorm_fp8(x, w, q, scale, eps):
    """q, scale = quantize_rows_fp8(rmsnorm(x, w)) in one pass; see bagel_rmsnorm_fp8."""
    _req(x, BF16, "rmsnorm_fp8.x"); _req(w, BF16, "rmsnorm_fp8.w"); _req(q, torch.uint8, "rmsnorm_fp8.q"); _req(scale, torch.float32, "rmsnorm_fp8.scale")
    rows, cols = x.shape
    check(lib().bagel_rmsnorm_fp8(_ptr(x), _ld(x), _ptr(w), _ptr(q), q.stride(0), _ptr(scale), rows, cols, float(eps), _stream()), "bagel_rmsnorm_fp8")
    return q, scale


def gemm_fp8(Aq, sa, Wq, sw, C, *, bias=None, rows=None, M=None, residual=None, epilogue=EPI_NONE):
    """C[rows] = epilogue(sa[rows] * sw * (Aq[rows] @ Wq^T)) on the fp8 MFMA (fp32 accumulate); ``rows`` = the row list of the one
    row group (gather of A / scatter of C and the residual); see bagel_gemm_fp8_bf16."""
    _req(Aq, torch.uint8, "gemm_fp8.Aq"); _req(Wq, torch.uint8, "gemm_fp8.Wq"); _req(C, BF16, "gemm_fp8.C")
    _req(sa, torch.float32, "gemm_fp8.sa"); _req(sw, torch.float32, "gemm_fp8.sw")
    N, K = Wq.shape
    if Aq.shape[-1] != K or sw.numel() != N or sa.numel() < Aq.shape[0]:
        raise BagelHipError("gemm_fp8: shape mismatch")
    if rows is not None:
        _req(rows, torch.int32, "gemm_fp8.rows")
    if M is None:
        M = rows.numel() if rows is not None else Aq.shape[0]
    if residual is not None:
        _req(residual, BF16, "gemm_fp8.residual")
    check(lib().bagel_gemm_fp8_bf16(_ptr(Aq), Aq.stride(0), _ptr(sa), _ptr(Wq), Wq.stride(0), _ptr(sw), _ptr(bias), _ptr(rows), _ptr(rows), M,
                                    _ptr(residual), _ld(residual) if residual is not None else 0, _ptr(C), _ld(C), N, K, epilogue, _stream()),
          "bagel_gemm_fp8_bf16")
    return C


def gemm_fp8_swiglu_q8(Aq, sa, Wq, sw, Cq, cs, cmax, *, rows=None, M=None):
    """The FP8 gen expert's gate/up projection with the SwiGLU result written as e4m3 bytes under a DELAYED row scale: Cq[rows] =
    e4m3(clamp(swiglu(sa[rows] * sw * (Aq[rows] @ Wq^T)) / cs[rows], +-448)), cmax[rows] = max(cmax[rows], rowmax |swiglu(..)|) (fp32 bit patterns);
    see bagel_gemm_fp8_swiglu_q8 / fp8_delayed_scales."""
    _req(Aq, torch.uint8, "gemm_fp8_swiglu_q8.Aq"); _req(Wq, torch.uint8, "gemm_fp8_swiglu_q8.Wq"); _req(Cq, torch.uint8, "gemm_fp8_swiglu_q8.Cq")
    for t, n in ((sa, "sa"), (sw, "sw"), (cs, "cs"), (cmax, "cmax")):
        _req(t, torch.float32, "gemm_fp8_swiglu_q8." + n)
    N, K = Wq.shape
    if Aq.shape[-1] != K or sw.numel() != N or Cq.shape[-1] < N // 2 or min(sa.numel(), cs.numel(), cmax.numel()) < Aq.shape[0] or Cq.shape[0] < Aq.shape[0]:
        raise BagelHipError("gemm_fp8_swiglu_q8: shape mismatch")
    if rows is not None:
        _req(rows, torch.int32, "gemm_fp8_swiglu_q8.rows")
    if M is None:
        M = rows.numel() if rows is not None else Aq.shape[0]
    check(lib().bagel_gemm_fp8_swiglu_q8(_ptr(Aq), Aq.stride(0), _ptr(sa), _ptr(Wq), Wq.stride(0), _ptr(sw), _ptr(rows), _ptr(rows), M,
                                         _ptr(Cq), Cq.stride(0), _ptr(cs), _ptr(cmax), N, K, _stream()), "bagel_gemm_fp8_swiglu_q8")
    return Cq


def fp8_delayed_scales(amax, scale, *, rows=None, n=None, margin=2.0):
    """scale[r] = margin * amax[r] / 448 (1.0 where amax is 0) and amax[r] = 0 for the rows in use: the step between two denoise forwards of the delayed
    scaling (bagel_fp8_delayed_scales); ``amax`` holds non-negative fp32 values collected by atomicMax on their bit patterns."""
    _req(amax, torch.float32, "fp8_delayed_scales.amax"); _req(scale, torch.float32, "fp8_delayed_scales.scale")
    if rows is not None:
        _req(rows, torch.int32, "fp8_delayed_scales.rows")
    if n is None:
        n = rows.numel() if rows is not None else amax.numel()
    check(lib().bagel_fp8_delayed_scales(_ptr(amax), _ptr(scale), _ptr(rows), n, float(margin), _stream()), "bagel_fp8_delayed_scales")
    return scale


def gemv_w8(A, Wq, scale, C, *, bias=None, residual=None, epilogue=EPI_NONE, M=None, norm_w=None, eps=0.0):
    """``gemv`` on row-wise INT8 weights (u8 + fp32 scales), activations bf16; see bagel_gemv_w8_bf16."""
    _req(A, BF16, "gemv_w8.A"); _req(Wq, torch.uint8, "gemv_w8.Wq"); _req(scale, torch.float32, "gemv_w8.scale"); _req(C, BF16, "gemv_w8.C")
    N, K = Wq.shape
    if A.shape[-1] != K or scale.numel() != N:
        raise BagelHipError("gemv_w8: shape mismatch")
    if M is None:
        M = A.shape[0]
    if residual is not None:
        _req(residual, BF16, "gemv_w8.residual")
    check(lib().bagel_gemv_w8_bf16(_ptr(A), _ld(A), _ptr(Wq), Wq.stride(0), _ptr(scale), _ptr(bias), _ptr(residual),
                                   _ld(residual) if residual is not None else 0, _ptr(C), _ld(C), _ptr(norm_w), float(eps), M, N, K,
                                   epilogue, _stream()), "bagel_gemv_w8_bf16")
    return C


def quantize_nf4(W):
    """bf16 [N, K] (K % 64 == 0) -> (u8 [N, K/2] NF4 codes, even element in the high nibble; fp32 absmax [N, K/64]); bagel_quantize_nf4 /
    oracle/nf4.py: the reference's own 4-bit load mode (app.py:114-125)."""
    _req(W, BF16, "quantize_nf4.W")
    N, K = W.shape
    if K % 64:
        raise BagelHipError(f"quantize_nf4: K={K} is not a multiple of the 64-weight block")
    q = torch.empty((N, K // 2), dtype=torch.uint8, device=W.device)
    a = torch.empty((N, K // 64), dtype=torch.float32, device=W.device)
    check(lib().bagel_quantize_nf4(_ptr(W), W.stride(0), _ptr(q), q.stride(0), _ptr(a), N, K, _stream()), "bagel_quantize_nf4")
    return q, a


def dequantize_nf4(q, absmax, out=None):
    """(u8 [N, K/2] NF4 codes, fp32 absmax [N, K/64]) -> bf16 [N, K] = bf16(code_book[code] * absmax): bagel_dequantize_nf4_bf16, what bitsandbytes'
    matmul_4bit does in front of F.linear for more than one activation row.  ``out``: a bf16 tensor with >= N rows of >= K columns (a view is fine)."""
    _req(q, torch.uint8, "dequantize_nf4.q"); _req(absmax, torch.float32, "dequantize_nf4.absmax")
    N, K = q.shape[0], 2 * q.shape[1]
    if out is None:
        out = torch.empty((N, K), dtype=BF16, device=q.device)
    _req(out, BF16, "dequantize_nf4.out")
    if out.shape[0] < N or out.shape[1] < K or out.stride(1) != 1:
        raise BagelHipError(f"dequantize_nf4: out {tuple(out.shape)} cannot hold [{N}, {K}]")
    check(lib().bagel_dequantize_nf4_bf16(_ptr(q), q.stride(0), _ptr(absmax), _ptr(out), out.stride(0), N, K, _stream()), "bagel_dequantize_nf4_bf16")
    return out[:N, :K]


def dequantize_rows_i8(q, scale, out=None):
    """(u8 [N, K], fp32 row scales [N]) of quantize_rows_i8 -> bf16 [N, K] = bf16((q - 128) * scale[row]); bagel_dequantize_rows_i8_bf16."""
    _req(q, torch.uint8, "dequantize_rows_i8.q"); _req(scale, torch.float32, "dequantize_rows_i8.scale")
    N, K = q.shape
    if out is None:
        out = torch.empty((N, K), dtype=BF16, device=q.device)
    _req(out, BF16, "dequantize_rows_i8.out")
    if out.shape[0] < N or out.shape[1] < K or out.stride(1) != 1:
        raise BagelHipError(f"dequantize_rows_i8: out {tuple(out.shape)} cannot hold [{N}, {K}]")
    check(lib().bagel_dequantize_rows_i8_bf16(_ptr(q), q.stride(0), _ptr(scale), _ptr(out), out.stride(0), N, K, _stream()), "bagel_dequantize_rows_i8_bf16")
    return out[:N, :K]


def gemv_nf4(A, Wq, absmax, C, *, bias=None, residual=None, epilogue=EPI_NONE, M=None, norm_w=None, eps=0.0):
    """``gemv`` on NF4 weights (packed codes + fp32 block absmax), activations bf16; see bagel_gemv_nf4_bf16."""
    _req(A, BF16, "gemv_nf4.A"); _req(Wq, torch.uint8, "gemv_nf4.Wq"); _req(absmax, torch.float32, "gemv_nf4.absmax"); _req(C, BF16, "gemv_nf4.C")
    N, K = Wq.shape[0], 2 * Wq.shape[1]
    if A.shape[-1] != K or absmax.shape != (N, K // 64) or not absmax.is_contiguous():
        raise BagelHipError("gemv_nf4: shape mismatch")
    if M is None:
        M = A.shape[0]
    if residual is not None:
        _req(residual, BF16, "gemv_nf4.residual")
    check(lib().bagel_gemv_nf4_bf16(_ptr(A), _ld(A), _ptr(Wq), Wq.stride(0), _ptr(absmax), _ptr(bias), _ptr(residual),
                                    _ld(residual) if residual is not None else 0, _ptr(C), _ld(C), _ptr(norm_w), float(eps), M, N, K,
                                    epilogue, _stream()), "bagel_gemv_nf4_bf16")
    return C


def quantize_rows_mxfp4(W):
    """bf16 [N, K] (K % 128 == 0) -> (u8 [N, K/2] E2M1 codes, two per byte; u8 [N, 16 * ceil(K/512)] E8M0 block scales in device order);
    see bagel_quantize_rows_mxfp4 / oracle/mxfp4.py."""
    _req(W, BF16, "quantize_rows_mxfp4.W")
    N, K = W.shape
    if K % 128:
        raise BagelHipError(f"quantize_rows_mxfp4: K={K} is not a multiple of 128")
    q = torch.empty((N, K // 2), dtype=torch.uint8, device=W.device)
    s = torch.empty((N, ((K // 128 + 3) // 4) * 16), dtype=torch.uint8, device=W.device)
    check(lib().bagel_quantize_rows_mxfp4(_ptr(W), W.stride(0), _ptr(q), q.stride(0), _ptr(s), s.stride(0), N, K, _stream()),
          "bagel_quantize_rows_mxfp4")
    return q, s


def gemv_w4(A, Wq, Ws, C, *, bias=None, residual=None, epilogue=EPI_NONE, M=None, norm_w=None, eps=0.0):
    """``gemv`` on MXFP4 weights (codes + block scales of ``quantize_rows_mxfp4``): activations quantised to FP8 inside the kernel, the
    product on the block-scaled MFMA; M <= 4 rows; see bagel_gemv_w4_bf16."""
    _req(A, BF16, "gemv_w4.A"); _req(Wq, torch.uint8, "gemv_w4.Wq"); _req(Ws, torch.uint8, "gemv_w4.Ws"); _req(C, BF16, "gemv_w4.C")
    N, K = Wq.shape[0], Wq.shape[1] * 2
    if A.shape[-1] != K or Ws.shape[0] != N:
        raise BagelHipError("gemv_w4: shape mismatch")
    if M is None:
        M = A.shape[0]
    if residual is not None:
        _req(residual, BF16, "gemv_w4.residual")
    if norm_w is not None:
        _req(norm_w, BF16, "gemv_w4.norm_w")
    check(lib().bagel_gemv_w4_bf16(_ptr(A), _ld(A), _ptr(Wq), Wq.stride(0), _ptr(Ws), Ws.stride(0), _ptr(bias), _ptr(residual),
                                   _ld(residual) if residual is not None else 0, _ptr(C), _ld(C), _ptr(norm_w), float(eps), M, N, K,
                                   epilogue, _stream()), "bagel_gemv_w4_bf16")
    return C


KV_PAGE = 64          # tokens per KV page (BAGEL_KV_PAGE in decode.hip)
DECODE_CHUNK = 64     # smallest keys-per-split the library may use (DEC_CH 128, or 64 with BAGEL_DEC_CH=64): sizes the workspace


def kv_append_paged(k_new, v_new, kpool, vpool, block_table, kv_len, batch, width):
    _req(k_new, BF16, "kv_append.k_new"); _req(v_new, BF16, "kv_append.v_new")
    _req(kpool, BF16, "kv_append.kpool"); _req(vpool, BF16, "kv_append.vpool")
    _req(block_table, torch.int32, "kv_append.block_table"); _req(kv_len, torch.int32, "kv_append.kv_len")
    if k_new.stride(0) != v_new.stride(0) or kpool.stride(0) != vpool.stride(0):
        raise BagelHipError("kv_append_paged: K and V must share row strides")
    check(lib().bagel_kv_append_paged_bf16(_ptr(k_new), _ptr(v_new), k_new.stride(0), _ptr(kpool), _ptr(vpool), kpool.stride(0),
                                           _ptr(block_table), block_table.stride(0), _ptr(kv_len), batch, width, _stream()),
          "bagel_kv_append_paged_bf16")


def decode_qkv_post(qkv, cos, sin, q_w, k_w, kpool, vpool, block_table, kv_len, batch, nq, nkv, head_dim, head_dim_padded, eps,
                    use_norm):
    """q/k norm + RoPE in place on q, finished K row and V row into page slot kv_len[b]; see bagel_decode_qkv_post_bf16."""
    _req(qkv, BF16, "decode_qkv_post.qkv"); _req(kpool, BF16, "decode_qkv_post.kpool"); _req(vpool, BF16, "decode_qkv_post.vpool")
    _req(block_table, torch.int32, "decode_qkv_post.block_table"); _req(kv_len, torch.int32, "decode_qkv_post.kv_len")
    if kpool.stride(0) != vpool.stride(0):
        raise BagelHipError("decode_qkv_post: K and V pools must share the row stride")
    check(lib().bagel_decode_qkv_post_bf16(_ptr(qkv), qkv.stride(0), _ptr(cos), _ptr(sin), _ptr(q_w), _ptr(k_w), _ptr(kpool),
                                           _ptr(vpool), kpool.stride(0), _ptr(block_table), block_table.stride(0), _ptr(kv_len),
                                           batch, nq, nkv, head_dim, head_dim_padded, float(eps), int(use_norm), _stream()),
          "bagel_decode_qkv_post_bf16")
    return qkv


def attn_decode_workspace(batch, nq, head_dim, max_len, device):
    ns = (max_len + DECODE_CHUNK - 1) // DECODE_CHUNK
    return (torch.empty((batch * nq * ns * head_dim,), dtype=torch.float32, device=device),
            torch.empty((batch * nq * ns * 2,), dtype=torch.float32, device=device))


def attn_decode_paged(q, kpool, vpool, block_table, kv_len, len_add, max_len, part_o, part_ml, out, batch, nq, nkv, head_dim,
                      softmax_scale):
    """Lq = 1 attention over keys [0, kv_len[b] + len_add) of the paged cache; q:[B, >= nq*D] rows, out:[B, nq*D]."""
    _req(q, BF16, "attn_decode.q"); _req(out, BF16, "attn_decode.out")
    _req(kpool, BF16, "attn_decode.kpool"); _req(vpool, BF16, "attn_decode.vpool")
    _req(block_table, torch.int32, "attn_decode.block_table"); _req(kv_len, torch.int32, "attn_decode.kv_len")
    _req(part_o, torch.float32, "attn_decode.part_o"); _req(part_ml, torch.float32, "attn_decode.part_ml")
    ns = (max_len + DECODE_CHUNK - 1) // DECODE_CHUNK
    if part_o.numel() < batch * nq * ns * head_dim or part_ml.numel() < batch * nq * ns * 2:
        raise BagelHipError("attn_decode_paged: workspace too small for max_len")
    check(lib().bagel_attn_decode_paged_bf16(_ptr(q), q.stride(0), _ptr(kpool), _ptr(vpool), kpool.stride(0), _ptr(block_table),
                                             block_table.stride(0), _ptr(kv_len), len_add, max_len, _ptr(part_o), _ptr(part_ml),
                                             _ptr(out), out.stride(0), batch, nq, nkv, head_dim, float(softmax_scale), _stream()),
          "bagel_attn_decode_paged_bf16")
    return out


def attn_decode_fused(qkv, cos, sin, q_w, k_w, kpool, vpool, block_table, kv_len, max_len, part_o, part_ml, out, batch, nq, nkv,
                      head_dim, head_dim_padded, eps, use_norm, softmax_scale):
    """decode_qkv_post + attn_decode_paged in one launch (+ combine); qkv = raw fused projection rows."""
    _req(qkv, BF16, "attn_decode_fused.qkv"); _req(out, BF16, "attn_decode_fused.out")
    _req(kpool, BF16, "attn_decode_fused.kpool"); _req(vpool, BF16, "attn_decode_fused.vpool")
    _req(block_table, torch.int32, "attn_decode_fused.block_table"); _req(kv_len, torch.int32, "attn_decode_fused.kv_len")
    ns = (max_len + DECODE_CHUNK - 1) // DECODE_CHUNK
    if part_o.numel() < batch * nq * ns * head_dim_padded or part_ml.numel() < batch * nq * ns * 2:
        raise BagelHipError("attn_decode_fused: workspace too small for max_len")
    check(lib().bagel_attn_decode_fused_bf16(_ptr(qkv), qkv.stride(0), _ptr(cos), _ptr(sin), _ptr(q_w), _ptr(k_w), _ptr(kpool), _ptr(vpool),
                                             kpool.stride(0), _ptr(block_table), block_table.stride(0), _ptr(kv_len), max_len,
                                             _ptr(part_o), _ptr(part_ml), _ptr(out), out.stride(0), batch, nq, nkv, head_dim,
                                             head_dim_padded, float(eps), int(use_norm), float(softmax_scale), _stream()),
          "bagel_attn_decode_fused_bf16")
    return out


def decode_advance(next_tok, cur_tok32, tokens_out, pos, kv_len, step, batch, max_steps):
    _req(next_tok, torch.int64, "decode_advance.next_tok"); _req(cur_tok32, torch.int32, "decode_advance.cur_tok32")
    _req(tokens_out, torch.int64, "decode_advance.tokens_out"); _req(pos, torch.int64, "decode_advance.pos")
    _req(kv_len, torch.int32, "decode_advance.kv_len"); _req(step, torch.int32, "decode_advance.step")
    check(lib().bagel_decode_advance(_ptr(next_tok), _ptr(cur_tok32), _ptr(tokens_out), _ptr(pos), _ptr(kv_len), _ptr(step),
                                     batch, max_steps, _stream()), "bagel_decode_advance")


def resample_u8(src, dst, bounds, kk, vertical):
    """One pass of the 8-bit bicubic resample.  src/dst: (H, W, C) uint8 on the GPU (rows may be strided);
    bounds int32 [out, 2], kk int32 [out, ksize] on the GPU (host-computed fixed-point taps)."""
    for t, n in ((src, "src"), (dst, "dst")):
        _req(t, torch.uint8, "resample_u8." + n)
        if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
            raise BagelHipError("resample_u8: expected (H, W, C) uint8 with packed pixels")
    _req(bounds, torch.int32, "resample_u8.bounds"); _req(kk, torch.int32, "resample_u8.kk")
    C = src.shape[2]
    if vertical:
        if dst.shape[1] != src.shape[1] or bounds.shape[0] != dst.shape[0]:
            raise BagelHipError("resample_u8: vertical pass shape mismatch")
        n_lines, out_len = src.shape[1] * C, dst.shape[0]
    else:
        if dst.shape[0] != src.shape[0] or bounds.shape[0] != dst.shape[1]:
            raise BagelHipError("resample_u8: horizontal pass shape mismatch")
        n_lines, out_len = src.shape[0], dst.shape[1]
    check(lib().bagel_resample_u8(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), n_lines, out_len, C, _ptr(bounds), _ptr(kk),
                                  kk.shape[1], int(vertical), _stream()), "bagel_resample_u8")
    return dst


def u8_to_chw_f32(src, mean, std):
    """(H, W, C) uint8 -> (C, H, W) fp32 = ((x / 255) - mean) / std  (ToTensor + Normalize)."""
    _req(src, torch.uint8, "u8_to_chw_f32.src")
    H, W, C = src.shape
    if src.stride(2) != 1 or src.stride(1) != C:
        raise BagelHipError("u8_to_chw_f32: expected packed (H, W, C) pixels")
    out = torch.empty((C, H, W), dtype=torch.float32, device=src.device)
    m = (ctypes.c_float * C)(*[float(x) for x in mean])
    s = (ctypes.c_float * C)(*[float(x) for x in std])
    check(lib().bagel_u8_to_chw_f32(_ptr(src), src.stride(0), _ptr(out), H, W, C, ctypes.addressof(m), ctypes.addressof(s), _stream()),
          "bagel_u8_to_chw_f32")
    return out


def chw_f32_to_u8(src):
    """(C, H, W) fp32 in [-1, 1] -> (H, W, C) uint8 = trunc(clamp(x * 0.5 + 0.5, 0, 1) * 255)  (decode_image)."""
    _req(src, torch.float32, "chw_f32_to_u8.src")
    C, H, W = src.shape
    out = torch.empty((H, W, C), dtype=torch.uint8, device=src.device)
    check(lib().bagel_chw_f32_to_u8(_ptr(src), src.stride(0), src.stride(1), _ptr(out), out.stride(0), H, W, C, _stream()),
          "bagel_chw_f32_to_u8")
    return out


def attn_varlen_ranges(q, k_new, vt_new, out, q_start, q_end, vt_new_col, batch, max_lq, nq, nkv, head_dim, causal, softmax_scale,
                       k_ctx=None, vt_ctx=None, ctx_start=None, ctx_end=None, vt_ctx_col=None, lse=None):
    """attn_varlen with explicit [start, end) row ranges per sequence (ranges may skip rows; context ranges may overlap).
    ``lse`` (fp32 [nq, rows]) additionally receives log2 of every row's softmax denominator -- the statistics the training backward
    reads instead of recomputing them (every query row must then belong to exactly one range)."""
    for t, n in ((q, "q"), (k_new, "k_new"), (vt_new, "vt_new"), (out, "out")):
        _req(t, BF16, "attn_ranges." + n)
    for t in (q_start, q_end, vt_new_col, ctx_start, ctx_end, vt_ctx_col):
        if t is not None:
            _req(t, torch.int32, "attn_ranges.index")
    if lse is not None:
        _req(lse, torch.float32, "attn_ranges.lse")
        if lse.dim() != 2 or lse.shape[0] != nq or lse.shape[1] < q.shape[0]:
            raise BagelHipError("attn_ranges: lse must be fp32 [nq, rows]")
        check(lib().bagel_attn_varlen_ranges_lse_bf16(
            _ptr(q), q.stride(0), _ptr(k_new), k_new.stride(0), _ptr(vt_new), vt_new.stride(0),
            _ptr(k_ctx), k_ctx.stride(0) if k_ctx is not None else 0, _ptr(vt_ctx), vt_ctx.stride(0) if vt_ctx is not None else 0,
            _ptr(out), out.stride(0), _ptr(q_start), _ptr(q_end), _ptr(ctx_start), _ptr(ctx_end), _ptr(vt_new_col), _ptr(vt_ctx_col),
            batch, max_lq, nq, nkv, head_dim, int(causal), float(softmax_scale), _ptr(lse), lse.stride(0), _stream()), "bagel_attn_varlen_ranges_lse_bf16")
        return out
    check(lib().bagel_attn_varlen_ranges_bf16(
        _ptr(q), q.stride(0), _ptr(k_new), k_new.stride(0), _ptr(vt_new), vt_new.stride(0),
        _ptr(k_ctx), k_ctx.stride(0) if k_ctx is not None else 0, _ptr(vt_ctx), vt_ctx.stride(0) if vt_ctx is not None else 0,
        _ptr(out), out.stride(0), _ptr(q_start), _ptr(q_end), _ptr(ctx_start), _ptr(ctx_end), _ptr(vt_new_col), _ptr(vt_ctx_col),
        batch, max_lq, nq, nkv, head_dim, int(causal), float(softmax_scale), _stream()), "bagel_attn_varlen_ranges_bf16")
    return out


def flow_mix(clean, noise, t):
    """bf16((1 - t[row]) * clean + t[row] * noise); clean/noise fp32 [n, cols], t fp32 [n]."""
    _req(clean, torch.float32, "flow_mix.clean"); _req(noise, torch.float32, "flow_mix.noise"); _req(t, torch.float32, "flow_mix.t")
    if not (clean.is_contiguous() and noise.is_contiguous() and t.is_contiguous()) or clean.shape != noise.shape or t.numel() != clean.shape[0]:
        raise BagelHipError("flow_mix: contiguous [n, cols] inputs and t[n] expected")
    out = torch.empty(clean.shape, dtype=BF16, device=clean.device)
    check(lib().bagel_flow_mix_bf16(_ptr(clean), _ptr(noise), _ptr(t), _ptr(out), clean.shape[0], clean.shape[1], _stream()),
          "bagel_flow_mix_bf16")
    return out


def flow_add_rows(seq, rows, temb, temb_ids, pos_table, pos_ids):
    _req(seq, BF16, "flow_add_rows.seq"); _req(rows, torch.int32, "flow_add_rows.rows"); _req(pos_ids, torch.int64, "flow_add_rows.pos_ids")
    _req(temb, BF16, "flow_add_rows.temb"); _req(temb_ids, torch.int32, "flow_add_rows.temb_ids"); _req(pos_table, BF16, "flow_add_rows.pos_table")
    check(lib().bagel_flow_add_rows_bf16(_ptr(seq), seq.stride(0), _ptr(rows), _ptr(temb), temb.stride(0), _ptr(temb_ids), _ptr(pos_table),
                                         pos_table.stride(0), _ptr(pos_ids), rows.numel(), seq.shape[1], _stream()),
          "bagel_flow_add_rows_bf16")
    return seq


def mse_rows(pred, noise, clean, src_rows):
    """(pred - (noise - clean)[src_rows])**2 -> fp32 [n, cols]."""
    _req(pred, BF16, "mse_rows.pred"); _req(noise, torch.float32, "mse_rows.noise"); _req(clean, torch.float32, "mse_rows.clean")
    _req(src_rows, torch.int32, "mse_rows.src_rows")
    n, cols = src_rows.numel(), noise.shape[1]
    out = torch.empty((n, cols), dtype=torch.float32, device=pred.device)
    check(lib().bagel_mse_rows_f32(_ptr(pred), pred.stride(0), _ptr(noise), _ptr(clean), _ptr(src_rows), _ptr(out), n, cols, _stream()),
          "bagel_mse_rows_f32")
    return out


def cross_entropy(logits, labels):
    """Per-row -log softmax(logits)[label], fp32 math on bf16 logits."""
    _req(logits, BF16, "cross_entropy.logits"); _req(labels, torch.int64, "cross_entropy.labels")
    out = torch.empty((logits.shape[0],), dtype=torch.float32, device=logits.device)
    check(lib().bagel_cross_entropy_bf16(_ptr(logits), logits.stride(0), _ptr(labels), _ptr(out), logits.shape[0], logits.shape[1],
                                         _stream()), "bagel_cross_entropy_bf16")
    return out


# ---- training backward (the reverse kernels of backward.hip / attention_bwd.hip; chained by modeling/bagel/train_step.py) ----
def _colsum_ws(rows, cols, device, extra=0):
    return torch.empty((-(-max(int(rows), 1) // 64) * int(cols) + int(extra),), dtype=torch.float32, device=device)


def transpose(src, dst=None, rows=None, n=None):
    """dst[c, j] = src[rows[j] if rows is not None else j, c] for j < n; dst[:, n:] = 0.  src [*, C] bf16 -> dst [C, ceil64(n)]."""
    _req(src, BF16, "transpose.src")
    if rows is not None:
        _req(rows, torch.int32, "transpose.rows")
    if n is None:
        n = rows.numel() if rows is not None else src.shape[0]
    C = src.shape[1]
    npad = -(-max(n, 1) // 64) * 64
    if dst is None:
        dst = torch.empty((C, npad), dtype=BF16, device=src.device)
    _req(dst, BF16, "transpose.dst")
    if dst.shape[0] < C or dst.shape[1] < npad:
        raise BagelHipError(f"transpose: dst {tuple(dst.shape)} too small for [{C}, {npad}]")
    check(lib().bagel_transpose_bf16(_ptr(src), _ld(src), _ptr(rows), n, C, _ptr(dst), _ld(dst), npad, _stream()), "bagel_transpose_bf16")
    return dst[:C, :npad]


def rmsnorm_bwd(x, dy, w0, g, eps, w1=None, expert=None, accumulate=True):
    """g = bf16((g if accumulate else 0) + d rmsnorm / dx); -> (dw0, dw1) bf16 [cols] (dw1 None without a second expert)."""
    _req(x, BF16, "rmsnorm_bwd.x"); _req(dy, BF16, "rmsnorm_bwd.dy"); _req(g, BF16, "rmsnorm_bwd.g"); _req(w0, BF16, "rmsnorm_bwd.w0")
    rows, cols = x.shape
    dw0 = torch.empty((cols,), dtype=BF16, device=x.device)
    dw1 = torch.empty_like(dw0) if w1 is not None else None
    ws = _colsum_ws(rows, 2 * cols, x.device, extra=rows)
    check(lib().bagel_rmsnorm_bwd_bf16(_ptr(x), _ld(x), _ptr(dy), _ld(dy), _ptr(w0), _ptr(w1), _ptr(expert if w1 is not None else None),
                                       _ptr(g), _ld(g), int(bool(accumulate)), _ptr(dw0), _ptr(dw1), _ptr(ws), rows, cols, float(eps),
                                       _stream()), "bagel_rmsnorm_bwd_bf16")
    return dw0, dw1


def layernorm_bwd(x, dy, w, g, eps, accumulate=True):
    """g = bf16((g if accumulate else 0) + d layernorm / dx); -> (dw, db) bf16 [cols]."""
    _req(x, BF16, "layernorm_bwd.x"); _req(dy, BF16, "layernorm_bwd.dy"); _req(g, BF16, "layernorm_bwd.g"); _req(w, BF16, "layernorm_bwd.w")
    rows, cols = x.shape
    dw = torch.empty((cols,), dtype=BF16, device=x.device)
    db = torch.empty_like(dw)
    ws = _colsum_ws(rows, 2 * cols, x.device, extra=2 * rows)
    check(lib().bagel_layernorm_bwd_bf16(_ptr(x), _ld(x), _ptr(dy), _ld(dy), _ptr(w), _ptr(g), _ld(g), int(bool(accumulate)), _ptr(dw), _ptr(db),
                                         _ptr(ws), rows, cols, float(eps), _stream()), "bagel_layernorm_bwd_bf16")
    return dw, db


def qknorm_rope_bwd(dqkv, qkv_raw, cos, sin, q_w0, k_w0, q_w1, k_w1, expert, nq, nkv, head_dim, head_dim_padded, eps, use_norm):
    """In place: gradient of the rotated [q | k | v] rows -> gradient of the raw projection; -> (dqw0, dkw0, dqw1, dkw1)."""
    _req(dqkv, BF16, "qknorm_rope_bwd.dqkv"); _req(qkv_raw, BF16, "qknorm_rope_bwd.qkv_raw")
    dev = dqkv.device
    two = use_norm and q_w1 is not None
    mk = lambda on: torch.empty((head_dim,), dtype=BF16, device=dev) if on else None  # noqa: E731
    dq0, dk0, dq1, dk1 = mk(use_norm), mk(use_norm), mk(two), mk(two)
    ws = _colsum_ws(dqkv.shape[0], 4 * head_dim, dev)
    check(lib().bagel_qknorm_rope_bwd_bf16(_ptr(dqkv), _ld(dqkv), _ptr(qkv_raw), _ld(qkv_raw), _ptr(cos), _ptr(sin), _ptr(q_w0), _ptr(k_w0),
                                           _ptr(q_w1 if two else None), _ptr(k_w1 if two else None), _ptr(expert if two else None),
                                           _ptr(dq0), _ptr(dk0), _ptr(dq1), _ptr(dk1), _ptr(ws), dqkv.shape[0], nq, nkv, head_dim,
                                           head_dim_padded, float(eps), int(use_norm), _stream()), "bagel_qknorm_rope_bwd_bf16")
    return dq0, dk0, dq1, dk1


def swiglu_bwd(gu, d_act):
    """gu [rows, 2 I] (interleaved [16 gate | 16 up]) <- its gradient given d_act [rows, I]."""
    _req(gu, BF16, "swiglu_bwd.gu"); _req(d_act, BF16, "swiglu_bwd.d_act")
    if gu.shape[1] != 2 * d_act.shape[1] or gu.shape[0] != d_act.shape[0]:
        raise BagelHipError("swiglu_bwd: gu must be [rows, 2 * I] for d_act [rows, I]")
    check(lib().bagel_swiglu_bwd_bf16(_ptr(gu), _ld(gu), _ptr(d_act), _ld(d_act), gu.shape[0], d_act.shape[1], _stream()), "bagel_swiglu_bwd_bf16")
    return gu


def swiglu_fwd(gu, act):
    """act [rows, I] = SwiGLU of the stored un-activated projection gu [rows, 2 I] (bit-identical to the EPI_SWIGLU16 epilogue)."""
    _req(gu, BF16, "swiglu_fwd.gu"); _req(act, BF16, "swiglu_fwd.act")
    if gu.shape[1] != 2 * act.shape[1] or gu.shape[0] != act.shape[0]:
        raise BagelHipError("swiglu_fwd: gu must be [rows, 2 * I] for act [rows, I]")
    check(lib().bagel_swiglu_fwd_bf16(_ptr(gu), _ld(gu), _ptr(act), _ld(act), gu.shape[0], act.shape[1], _stream()), "bagel_swiglu_fwd_bf16")
    return act


def act_bwd(pre, d_out, kind):
    """pre <- d_out * act'(pre) for the GELU-tanh / SiLU epilogues (kind = EPI_GELU_TANH / EPI_SILU)."""
    _req(pre, BF16, "act_bwd.pre"); _req(d_out, BF16, "act_bwd.d_out")
    if pre.shape != d_out.shape or kind not in (EPI_GELU_TANH, EPI_SILU):
        raise BagelHipError("act_bwd: same-shape operands and kind in (EPI_GELU_TANH, EPI_SILU) expected")
    check(lib().bagel_act_bwd_bf16(_ptr(pre), _ld(pre), _ptr(d_out), _ld(d_out), pre.shape[0], pre.shape[1], kind, _stream()), "bagel_act_bwd_bf16")
    return pre


def cross_entropy_bwd(logits, labels, d_loss):
    """logits <- (softmax(logits) - onehot(labels)) * d_loss[:, None], in place."""
    _req(logits, BF16, "cross_entropy_bwd.logits"); _req(labels, torch.int64, "cross_entropy_bwd.labels")
    _req(d_loss, torch.float32, "cross_entropy_bwd.d_loss")
    if d_loss.numel() != logits.shape[0] or not d_loss.is_contiguous():
        raise BagelHipError("cross_entropy_bwd: d_loss must be a contiguous [rows] vector")
    check(lib().bagel_cross_entropy_bwd_bf16(_ptr(logits), logits.stride(0), _ptr(labels), _ptr(d_loss), logits.shape[0], logits.shape[1],
                                             _stream()), "bagel_cross_entropy_bwd_bf16")
    return logits


def mse_rows_bwd(pred, noise, clean, src_rows, d_loss):
    """-> bf16 [n, cols]: 2 (pred - (noise - clean)[src_rows]) d_loss."""
    _req(pred, BF16, "mse_rows_bwd.pred"); _req(noise, torch.float32, "mse_rows_bwd.noise"); _req(clean, torch.float32, "mse_rows_bwd.clean")
    _req(src_rows, torch.int32, "mse_rows_bwd.src_rows"); _req(d_loss, torch.float32, "mse_rows_bwd.d_loss")
    n, cols = src_rows.numel(), noise.shape[1]
    if tuple(d_loss.shape) != (n, cols) or not d_loss.is_contiguous():
        raise BagelHipError("mse_rows_bwd: d_loss must be contiguous [n, cols]")
    out = torch.empty((n, cols), dtype=BF16, device=pred.device)
    check(lib().bagel_mse_rows_bwd_bf16(_ptr(pred), pred.stride(0), _ptr(noise), _ptr(clean), _ptr(src_rows), _ptr(d_loss), _ptr(out),
                                        out.stride(0), n, cols, _stream()), "bagel_mse_rows_bwd_bf16")
    return out


def rows_segment_sum(src, order, seg_off, dst_rows, dst):
    """dst[dst_rows[s]] = sum(src[order[seg_off[s]:seg_off[s + 1]]]) (fp32 sums, bf16 result)."""
    _req(src, BF16, "rows_segment_sum.src"); _req(dst, BF16, "rows_segment_sum.dst")
    for t, nm in ((order, "order"), (seg_off, "seg_off"), (dst_rows, "dst_rows")):
        _req(t, torch.int32, "rows_segment_sum." + nm)
    check(lib().bagel_rows_segment_sum_bf16(_ptr(src), _ld(src), _ptr(order), _ptr(seg_off), _ptr(dst_rows), _ptr(dst), _ld(dst),
                                            dst_rows.numel(), src.shape[1], _stream()), "bagel_rows_segment_sum_bf16")
    return dst


def colsum(src, rows=None, n=None):
    """-> bf16 [cols]: sum over the listed rows (all rows when rows is None)."""
    _req(src, BF16, "colsum.src")
    if rows is not None:
        _req(rows, torch.int32, "colsum.rows")
    if n is None:
        n = rows.numel() if rows is not None else src.shape[0]
    out = torch.empty((src.shape[1],), dtype=BF16, device=src.device)
    ws = _colsum_ws(n, src.shape[1], src.device)
    check(lib().bagel_colsum_bf16(_ptr(src), _ld(src), _ptr(rows), n, src.shape[1], _ptr(ws), _ptr(out), _stream()), "bagel_colsum_bf16")
    return out


def attn_bwd_blockmask(q, k, v, o, d_o, dq, dk, dv, q_items, k_items, noise_bits, nq, nkv, head_dim, softmax_scale, lse=None):
    """Reverse of the block-masked packed attention (bagel_attn_bwd_blockmask_bf16); q / k / v / o / d_o / dq / dk / dv are
    [rows, heads * head_dim] views (any row stride); head_dim = the padded width (64 or 128).  ``lse`` = the forward's row statistics
    (attn_varlen_ranges(..., lse=)): contiguous fp32 [nq, rows]; without it the first kernel recomputes them."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (d_o, "d_o"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _req(t, BF16, "attn_bwd." + nm)
    _req(q_items, torch.int32, "attn_bwd.q_items"); _req(k_items, torch.int32, "attn_bwd.k_items"); _req(noise_bits, torch.int64, "attn_bwd.noise_bits")
    rows = q.shape[0]
    qt, dot, kt = transpose(q), transpose(d_o), transpose(k)
    if noise_bits.numel() * 64 < qt.shape[1]:
        raise BagelHipError("attn_bwd: noise_bits must cover ceil64(rows) keys")
    ws = torch.empty((2, nq, rows), dtype=torch.float32, device=q.device)
    if lse is not None:
        _req(lse, torch.float32, "attn_bwd.lse")
        if tuple(lse.shape) != (nq, rows):
            raise BagelHipError("attn_bwd: lse must be fp32 [nq, rows]")
        ws[0].copy_(lse)
    check(lib().bagel_attn_bwd_blockmask_bf16(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o), o.stride(0),
                                              _ptr(d_o), d_o.stride(0), _ptr(qt), _ptr(dot), _ptr(kt), qt.stride(0), _ptr(dq), dq.stride(0),
                                              _ptr(dk), dk.stride(0), _ptr(dv), dv.stride(0), _ptr(q_items), q_items.shape[0], _ptr(k_items),
                                              k_items.shape[0], _ptr(noise_bits), _ptr(ws), int(lse is not None), rows, nq, nkv, head_dim, float(softmax_scale),
                                              _stream()), "bagel_attn_bwd_blockmask_bf16")
    return dq, dk, dv


def _ptr_array(tensors, n):
    arr = (ctypes.c_void_p * max(n, 1))()
    for i in range(n):
        _req(tensors[i], BF16, "taylor.factor")
        if not tensors[i].is_contiguous():
            raise BagelHipError("taylor: factor buffers must be contiguous")
        arr[i] = _ptr(tensors[i])
    return arr


def taylor_update(feature, factors, n_diff, distance):
    """TaylorSeer full step: factors[0] <- feature, factors[i+1] <- bf16(bf16(new_i - old_i) / distance), i < n_diff."""
    _req(feature, BF16, "taylor_update.feature")
    rows, cols = feature.shape
    arr = _ptr_array(factors, n_diff + 1)
    check(lib().bagel_taylor_update_bf16(_ptr(feature), feature.stride(0), ctypes.addressof(arr), n_diff, int(distance), rows, cols,
                                         _stream()), "bagel_taylor_update_bf16")


def taylor_eval(factors, n, x, out):
    """TaylorSeer skipped step: out = sum_i bf16(bf16(factors[i] / i!) * x**i) with a bf16 running sum."""
    _req(out, BF16, "taylor_eval.out")
    rows, cols = out.shape
    arr = _ptr_array(factors, n)
    check(lib().bagel_taylor_eval_bf16(ctypes.addressof(arr), n, int(x), _ptr(out), out.stride(0), rows, cols, _stream()),
          "bagel_taylor_eval_bf16")
    return out


class HipGraph:
    """A captured launch sequence (hipGraph).  ``with HipGraph.capture(stream) as g: <launch ops>`` then ``g.launch()``."""

    def __init__(self, stream):
        self.stream = stream
        self._exec = None

    def __enter__(self):
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        try:
            check(lib().bagel_graph_begin(self.stream.cuda_stream), "bagel_graph_begin")
        except Exception:
            self._ctx.__exit__(None, None, None)
            raise
        return self

    def __exit__(self, et, ev, tb):
        handle = ctypes.c_void_p(0)
        try:
            rc = lib().bagel_graph_end(self.stream.cuda_stream, ctypes.addressof(handle))
        finally:
            self._ctx.__exit__(None, None, None)
        if et is None:
            check(rc, "bagel_graph_end")
            self._exec = handle.value
        elif rc == 0 and handle.value:
            lib().bagel_graph_destroy(handle.value)
        return False

    @classmethod
    def capture(cls, stream):
        return cls(stream)

    def launch(self):
        check(lib().bagel_graph_launch(self._exec, self.stream.cuda_stream), "bagel_graph_launch")

    def __del__(self):
        if getattr(self, "_exec", None):
            try:
                lib().bagel_graph_destroy(self._exec)
            except Exception:
                pass
            self._exec = None


def rmsnorm(x, w0, out, eps, w1=None, expert=None):
    _req(x, BF16, "rmsnorm.x"); _req(out, BF16, "rmsnorm.out"); _req(w0, BF16, "rmsnorm.w0")
    rows, cols = x.shape
    check(lib().bagel_rmsnorm_bf16(_ptr(x), _ld(x), _ptr(w0), _ptr(w1), _ptr(expert), _ptr(out), _ld(out), rows, cols,
                                   float(eps), _stream()), "bagel_rmsnorm_bf16")
    return out


def layernorm(x, w, b, out, eps):
    _req(x, BF16, "layernorm.x"); _req(out, BF16, "layernorm.out")
    rows, cols = x.shape
    check(lib().bagel_layernorm_bf16(_ptr(x), _ld(x), _ptr(w), _ptr(b), _ptr(out), _ld(out), rows, cols, float(eps),
                                     _stream()), "bagel_layernorm_bf16")
    return out


def rope_table(position_ids, inv_freq):
    _req(position_ids, torch.int64, "rope_table.position_ids"); _req(inv_freq, torch.float32, "rope_table.inv_freq")
    rows, half = position_ids.numel(), inv_freq.numel()
    cos = torch.empty((rows, half), dtype=BF16, device=position_ids.device)
    sin = torch.empty_like(cos)
    check(lib().bagel_rope_table(_ptr(position_ids), _ptr(inv_freq), _ptr(cos), _ptr(sin), rows, half, _stream()),
          "bagel_rope_table")
    return cos, sin


def rope_table_into(position_ids, inv_freq, cos, sin):
    """rope_table writing into caller-owned [rows, half] bf16 buffers (no allocation: graph-capturable)."""
    _req(position_ids, torch.int64, "rope_table.position_ids"); _req(inv_freq, torch.float32, "rope_table.inv_freq")
    _req(cos, BF16, "rope_table.cos"); _req(sin, BF16, "rope_table.sin")
    rows, half = position_ids.numel(), inv_freq.numel()
    if cos.numel() != rows * half or sin.numel() != rows * half or not cos.is_contiguous() or not sin.is_contiguous():
        raise BagelHipError("rope_table_into: cos/sin must be contiguous [rows, half]")
    check(lib().bagel_rope_table(_ptr(position_ids), _ptr(inv_freq), _ptr(cos), _ptr(sin), rows, half, _stream()),
          "bagel_rope_table")
    return cos, sin


def qknorm_rope(qkv, cos, sin, q_w0, k_w0, q_w1, k_w1, expert, nq, nkv, head_dim, head_dim_padded, eps, gen_mode, use_norm):
    _req(qkv, BF16, "qknorm_rope.qkv")
    check(lib().bagel_qknorm_rope_bf16(_ptr(qkv), _ld(qkv), _ptr(cos), _ptr(sin), _ptr(q_w0), _ptr(k_w0), _ptr(q_w1),
                                       _ptr(k_w1), _ptr(expert), qkv.shape[0], nq, nkv, head_dim, head_dim_padded,
                                       float(eps), int(gen_mode), int(use_norm), _stream()), "bagel_qknorm_rope_bf16")
    return qkv


def rope2d(qkv, tables, pos_ids, nheads, head_dim, head_dim_padded):
    """SigLIP 2-D RoPE in place on the first ``nheads`` heads (q then k) of every row; tables = (cos_h, sin_h, cos_w, sin_w)."""
    _req(qkv, BF16, "rope2d.qkv"); _req(pos_ids, torch.int64, "rope2d.pos_ids")
    for t in tables:
        _req(t, BF16, "rope2d.table")
        if not t.is_contiguous() or t.shape[1] != head_dim // 2:
            raise BagelHipError("rope2d: tables must be contiguous [positions, head_dim/2]")
    check(lib().bagel_rope2d_bf16(_ptr(qkv), qkv.stride(0), _ptr(tables[0]), _ptr(tables[1]), _ptr(tables[2]), _ptr(tables[3]),
                                  _ptr(pos_ids), qkv.shape[0], nheads, head_dim, head_dim_padded, _stream()), "bagel_rope2d_bf16")
    return qkv


def attn_varlen(q, k_new, vt_new, out, cu_q, vt_new_col, batch, max_lq, nq, nkv, head_dim, causal, softmax_scale,
                k_ctx=None, vt_ctx=None, cu_ctx=None, vt_ctx_col=None):
    """q:[M, >=nq*D] rows with stride; k_new likewise; vt_new:[nkv*D, cols]; ctx triple optional."""
    for t, n in ((q, "q"), (k_new, "k_new"), (vt_new, "vt_new"), (out, "out")):
        _req(t, BF16, "attn." + n)
    check(lib().bagel_attn_varlen_bf16(
        _ptr(q), q.stride(0), _ptr(k_new), k_new.stride(0), _ptr(vt_new), vt_new.stride(0),
        _ptr(k_ctx), k_ctx.stride(0) if k_ctx is not None else 0, _ptr(vt_ctx), vt_ctx.stride(0) if vt_ctx is not None else 0,
        _ptr(out), out.stride(0), _ptr(cu_q), _ptr(cu_ctx), _ptr(vt_new_col), _ptr(vt_ctx_col), batch, max_lq, nq, nkv,
        head_dim, int(causal), float(softmax_scale), _stream()), "bagel_attn_varlen_bf16")
    return out


class AttnPlan:
    """Work list of the persistent attention kernel (csrc/attention2.hip) for ONE forward shape: built on the host by the library's
    planner (``bagel_attn_plan``: pure host code, runs without a GPU), copied to the device once and reused by every layer.

    ``q_start / q_len`` = first row and length of every sample's query (= new key) rows, ``ctx_start / ctx_len`` the same for its
    context rows (None = no context), ``vt_new_col / vt_ctx_col`` the V^T column of each sample's first key -- all HOST int lists in
    the layout ``attn_varlen`` uses."""

    def __init__(self, q_start, q_len, vt_new_col, nq, nkv, head_dim, causal, device, ctx_start=None, ctx_len=None, vt_ctx_col=None,
                 n_workers=None, split_min_tiles=0):
        import numpy as np
        B = len(q_len)
        self.q_start, self.q_len, self.vt_new_col = [int(x) for x in q_start], [int(x) for x in q_len], [int(x) for x in vt_new_col]
        self.has_ctx = ctx_len is not None and any(int(x) > 0 for x in ctx_len)
        self.ctx_start = [int(x) for x in ctx_start] if self.has_ctx else [0] * B
        self.ctx_len = [int(x) for x in ctx_len] if self.has_ctx else [0] * B
        self.vt_ctx_col = [int(x) for x in vt_ctx_col] if self.has_ctx else [0] * B
        self.nq, self.nkv, self.head_dim, self.causal = int(nq), int(nkv), int(head_dim), bool(causal)
        if n_workers is None:      # one persistent workgroup per CU; 256 (an MI355X) when planned on the host alone
            n_workers = torch.cuda.get_device_properties(device).multi_processor_count // 8 * 8 if torch.device(device).type == "cuda" else 256
        self.n_workers = int(n_workers)
        i32 = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.int32))  # noqa: E731
        arrs = [i32(self.q_start), i32(self.q_len), i32(self.ctx_start), i32(self.ctx_len), i32(self.vt_new_col), i32(self.vt_ctx_col)]
        items_max = 2 * sum(-(-l // 256) for l in self.q_len) * self.nq + 4 * self.n_workers      # key-split sub-items included (the planner checks)
        cap = 8 + self.n_workers + 1 + 16 + 16 * items_max + 8 * self.n_workers
        buf = np.zeros(cap, dtype=np.int32)
        ptr = lambda a: a.ctypes.data  # noqa: E731
        check(lib().bagel_attn_plan(ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), ptr(arrs[3]), ptr(arrs[4]), ptr(arrs[5]), B, self.nq,
                                    self.nkv, int(self.causal), self.n_workers, int(split_min_tiles), ptr(buf), cap), "bagel_attn_plan")
        self.n_items, self.n_comb, self.n_slots = int(buf[1]), int(buf[2]), int(buf[3])
        self.off_items, self.off_comb = int(buf[4]), int(buf[5])
        self.makespan, self.total = int(buf[6]), int(buf[7])
        used = self.off_comb + 8 * self.n_comb
        self.host = buf[:used].copy()
        self.dev = torch.from_numpy(self.host).to(device) if torch.device(device).type == "cuda" else None

    def items(self):
        """[n_items, 16] host view of the item table (tests)."""
        return self.host[self.off_items:self.off_items + 16 * self.n_items].reshape(self.n_items, 16)

    def worker_off(self):
        return self.host[8:8 + self.n_workers + 1]

    def combines(self):
        return self.host[self.off_comb:self.off_comb + 8 * self.n_comb].reshape(self.n_comb, 8)


_ATTN_PARTIALS = {}


def _attn_partials(device, n_slots, head_dim):
    """fp32 workspace of the key-split items, one per (device, stream): launches on one stream are ordered, so a buffer is never replaced
    while a launch that reads it can still be running."""
    need = n_slots * 256 * (head_dim + 2)
    key = _ws_key(device)
    ws = _ATTN_PARTIALS.get(key)
    if ws is None or ws.numel() < need:
        ws = _ATTN_PARTIALS[key] = torch.empty(need, dtype=torch.float32, device=device)
    return ws


def attn_planned(q, k_new, vt_new, out, aplan, softmax_scale, k_ctx=None, vt_ctx=None):
    """attn_varlen through a prebuilt AttnPlan (persistent kernel, head-per-wave tail tiles, key-split last round)."""
    for t, n in ((q, "q"), (k_new, "k_new"), (vt_new, "vt_new"), (out, "out")):
        _req(t, BF16, "attn_planned." + n)
    if aplan.has_ctx and (k_ctx is None or vt_ctx is None):
        raise BagelHipError("attn_planned: the plan has context keys but no context tensors were given")
    if aplan.dev is None or aplan.dev.device != q.device:
        raise BagelHipError("attn_planned: the plan was not built for this device")
    part = _attn_partials(q.device, aplan.n_slots, aplan.head_dim) if aplan.n_comb else None
    check(lib().bagel_attn_planned_bf16(
        _ptr(q), q.stride(0), _ptr(k_new), k_new.stride(0), _ptr(vt_new), vt_new.stride(0),
        _ptr(k_ctx) if aplan.has_ctx else None, k_ctx.stride(0) if aplan.has_ctx else 0,
        _ptr(vt_ctx) if aplan.has_ctx else None, vt_ctx.stride(0) if aplan.has_ctx else 0,
        _ptr(out), out.stride(0), _ptr(aplan.dev), aplan.n_workers, aplan.n_comb, aplan.off_items, aplan.off_comb, _ptr(part),
        aplan.head_dim, float(softmax_scale), _stream()), "bagel_attn_planned_bf16")
    return out


def v_transpose(v, vt, cu_rows, col_start, batch, max_len, nkv, head_dim):
    _req(v, BF16, "v_transpose.v"); _req(vt, BF16, "v_transpose.vt")
    check(lib().bagel_v_transpose_bf16(_ptr(v), v.stride(0), _ptr(vt), vt.stride(0), _ptr(cu_rows), _ptr(col_start), batch,
                                       max_len, nkv, head_dim, _stream()), "bagel_v_transpose_bf16")
    return vt


def copy_rows(src, dst, n, cols, src_rows=None, dst_rows=None):
    _req(src, BF16, "copy_rows.src"); _req(dst, BF16, "copy_rows.dst")
    check(lib().bagel_copy_rows_bf16(_ptr(src), src.stride(0), _ptr(src_rows), _ptr(dst), dst.stride(0), _ptr(dst_rows), n,
                                     cols, _stream()), "bagel_copy_rows_bf16")
    return dst


def f32_to_bf16(src, dst=None, cols_padded=None):
    """2-D fp32 -> bf16; optional zero padding of the row width (K padding for the GEMM)."""
    _req(src, torch.float32, "f32_to_bf16.src")
    if src.dim() != 2:
        raise BagelHipError("f32_to_bf16 expects a 2-D tensor")
    rows, cols = src.shape
    cp = cols if cols_padded is None else cols_padded
    if dst is None:
        dst = torch.empty((rows, cp), dtype=BF16, device=src.device)
    check(lib().bagel_f32_to_bf16(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), rows, cols, cp, _stream()),
          "bagel_f32_to_bf16")
    return dst


def timestep_sinusoid(t, freqs, out):
    check(lib().bagel_timestep_sinusoid(float(t), _ptr(freqs), _ptr(out), freqs.numel(), _stream()), "bagel_timestep_sinusoid")
    return out


def flow_add(seq, rows, temb, pos_table, pos_ids):
    _req(seq, BF16, "flow_add.seq"); _req(rows, torch.int32, "flow_add.rows"); _req(pos_ids, torch.int64, "flow_add.pos_ids")
    _req(temb, BF16, "flow_add.temb"); _req(pos_table, BF16, "flow_add.pos_table")
    check(lib().bagel_flow_add_bf16(_ptr(seq), seq.stride(0), _ptr(rows), _ptr(temb), _ptr(pos_table), pos_table.stride(0),
                                    _ptr(pos_ids), rows.numel(), seq.shape[1], _stream()), "bagel_flow_add_bf16")
    return seq


def add_table_rows(x, table, ids):
    _req(x, BF16, "add_table_rows.x"); _req(ids, torch.int64, "add_table_rows.ids"); _req(table, BF16, "add_table_rows.table")
    check(lib().bagel_add_table_rows_bf16(_ptr(x), x.stride(0), _ptr(table), table.stride(0), _ptr(ids), x.shape[0],
                                          x.shape[1], _stream()), "bagel_add_table_rows_bf16")
    return x


_MAX_PARTIALS = 256


def cfg_stage1(v, v_ct, v_ci, tmp, partials, text_scale, img_scale, renorm_min, mode):
    n_rows, cols = v.shape
    nparts = ctypes.c_int32(0)
    check(lib().bagel_cfg_stage1(_ptr(v), _ptr(v_ct), _ptr(v_ci), _ptr(tmp), _ptr(partials), _MAX_PARTIALS, n_rows, cols,
                                 float(text_scale), float(img_scale), float(renorm_min), mode,
                                 ctypes.addressof(nparts), _stream()), "bagel_cfg_stage1")
    return nparts.value


def cfg_stage2_euler(x_t, v_or_tmp, partials, nparts, renorm_min, dt, use_global_scale):
    _req(x_t, torch.float32, "cfg_stage2.x_t")
    check(lib().bagel_cfg_stage2_euler(_ptr(x_t), _ptr(v_or_tmp), _ptr(partials), nparts, float(renorm_min), float(dt),
                                       x_t.numel(), int(use_global_scale), _stream()), "bagel_cfg_stage2_euler")
    return x_t


# ---- VAE (fp32, NHWC): thin launch wrappers, tensors in / raw pointers out (bagel_amd/modeling/vae_engine.py) ---------------
def conv_gemm_f32(x, ld_in, w, ld_w, bias, residual, out, ld_out, B, Hin, Win, Cin, Hout, Wout, Cout, mode):
    """bagel_conv_gemm_f32: mode 0 plain GEMM rows x w^T, 1 = 3x3 s1 p1, 2 = 3x3 s2 pad (0,1,0,1), 3 = nearest-2x + 3x3."""
    _req(x, torch.float32, "conv_gemm_f32.x"); _req(w, torch.float32, "conv_gemm_f32.w"); _req(out, torch.float32, "conv_gemm_f32.out")
    check(lib().bagel_conv_gemm_f32(_ptr(x), ld_in, _ptr(w), ld_w, _ptr(bias), _ptr(residual), _ptr(out), ld_out, B, Hin, Win, Cin,
                                    Hout, Wout, Cout, mode, _stream()), "bagel_conv_gemm_f32")
    return out


def groupnorm_f32(x, y, workspace, gamma, beta, B, HW, C, groups, eps, swish):
    _req(x, torch.float32, "groupnorm_f32.x"); _req(y, torch.float32, "groupnorm_f32.y")
    check(lib().bagel_groupnorm_f32(_ptr(x), _ptr(y), _ptr(workspace), _ptr(gamma), _ptr(beta), B, HW, C, groups, float(eps),
                                    int(swish), _stream()), "bagel_groupnorm_f32")
    return y


def softmax_rows_f32(x, ld, rows, cols, scale):
    _req(x, torch.float32, "softmax_rows_f32.x")
    check(lib().bagel_softmax_rows_f32(_ptr(x), ld, rows, cols, float(scale), _stream()), "bagel_softmax_rows_f32")
    return x


def conv_gemm_bf16(x, ld_in, w, ld_w, bias, residual, out, ld_out, B, Hin, Win, Cin, Hout, Wout, Cout, mode):
    """bagel_conv_gemm_bf16 (the VAE under the inferencer's bf16 autocast): modes as conv_gemm_f32; ``out`` bf16, or fp32 for raw scores."""
    _req(x, BF16, "conv_gemm_bf16.x"); _req(w, BF16, "conv_gemm_bf16.w")
    if out.dtype not in (BF16, torch.float32):
        raise BagelHipError("conv_gemm_bf16: out must be bf16 or fp32")
    check(lib().bagel_conv_gemm_bf16(_ptr(x), ld_in, _ptr(w), ld_w, _ptr(bias), _ptr(residual), _ptr(out), ld_out, int(out.dtype == torch.float32),
                                     B, Hin, Win, Cin, Hout, Wout, Cout, mode, _stream()), "bagel_conv_gemm_bf16")
    return out


def groupnorm_bf16_workspace_floats(B, C, groups):
    """[B * groups][1024 slices][2] partial sums + (mean, rstd) per group + (scale, shift) per channel (include/bagel_hip.h)."""
    return B * groups * 2050 + B * C * 2


def groupnorm_bf16(x, y, workspace, gamma, beta, B, HW, C, groups, eps, swish):
    _req(x, BF16, "groupnorm_bf16.x"); _req(y, BF16, "groupnorm_bf16.y"); _req(gamma, torch.float32, "groupnorm_bf16.gamma")
    _req(workspace, torch.float32, "groupnorm_bf16.workspace")
    if workspace.numel() < groupnorm_bf16_workspace_floats(B, C, groups):
        raise BagelHipError(f"groupnorm_bf16: workspace of {workspace.numel()} floats < {groupnorm_bf16_workspace_floats(B, C, groups)}")
    check(lib().bagel_groupnorm_bf16(_ptr(x), _ptr(y), _ptr(workspace), _ptr(gamma), _ptr(beta), B, HW, C, groups, float(eps),
                                     int(swish), _stream()), "bagel_groupnorm_bf16")
    return y


def softmax_rows_bf16(x, y, rows, cols, scale):
    _req(x, torch.float32, "softmax_rows_bf16.x"); _req(y, BF16, "softmax_rows_bf16.y")
    check(lib().bagel_softmax_rows_bf16(_ptr(x), x.stride(0), _ptr(y), y.stride(0), rows, cols, float(scale), _stream()), "bagel_softmax_rows_bf16")
    return y


def vae_reparam_bf16(moments, noise, z, n_pix, z_channels, scale, shift):
    _req(moments, BF16, "vae_reparam_bf16.moments"); _req(z, BF16, "vae_reparam_bf16.z")
    check(lib().bagel_vae_reparam_bf16(_ptr(moments), moments.stride(-2), _ptr(noise), _ptr(z), n_pix, z_channels, float(scale), float(shift), _stream()),
          "bagel_vae_reparam_bf16")
    return z


def chw_bf16_to_u8(src):
    """(C, H, W) bf16 (any strides) -> (H, W, C) uint8 with the eager-bf16 rounding points of decode_image under autocast."""
    _req_any_stride(src, BF16, "chw_bf16_to_u8.src")
    C, H, W = src.shape
    out = torch.empty((H, W, C), dtype=torch.uint8, device=src.device)
    check(lib().bagel_chw_bf16_to_u8(_ptr(src), src.stride(0), src.stride(1), src.stride(2), _ptr(out), out.stride(0), H, W, C, _stream()),
          "bagel_chw_bf16_to_u8")
    return out


def vae_reparam_f32(moments, noise, z, n_pix, z_channels, scale, shift):
    _req(moments, torch.float32, "vae_reparam_f32.moments"); _req(noise, torch.float32, "vae_reparam_f32.noise")
    check(lib().bagel_vae_reparam_f32(_ptr(moments), _ptr(noise), _ptr(z), n_pix, z_channels, float(scale), float(shift), _stream()),
          "bagel_vae_reparam_f32")
    return z


def vae_unscale_f32(z, out, n, scale, shift):
    _req(z, torch.float32, "vae_unscale_f32.z")
    check(lib().bagel_vae_unscale_f32(_ptr(z), _ptr(out), n, float(scale), float(shift), _stream()), "bagel_vae_unscale_f32")
    return out


def require_gpu_f32(t, what):
    if not t.is_cuda or t.dtype != torch.float32:
        raise BagelHipError(f"{what}: the VAE runs in fp32 on an MI355X (as the reference; got {t.device}, {t.dtype}): "
                            "call vae.to('cuda') and keep it fp32")


def argmax_into(logits, out):
    _req(logits, BF16, "argmax.logits"); _req(out, torch.int64, "argmax.out")
    check(lib().bagel_argmax_bf16(_ptr(logits), logits.stride(0), _ptr(out), logits.shape[0], logits.shape[1], _stream()),
          "bagel_argmax_bf16")
    return out


def sample_gumbel_into(logits, out, temperature, seed, step_ctr=None):
    """Next-token sampling on the device: out[b] ~ Categorical(softmax(logits[b] / temperature)) by Gumbel-max over Philox uniforms (bagel_sample_gumbel_bf16)."""
    _req(logits, BF16, "sample_gumbel.logits"); _req(out, torch.int64, "sample_gumbel.out")
    if step_ctr is not None:
        _req(step_ctr, torch.int32, "sample_gumbel.step_ctr")
    check(lib().bagel_sample_gumbel_bf16(_ptr(logits), logits.stride(0), _ptr(out), logits.shape[0], logits.shape[1], float(temperature), int(seed),
                                         _ptr(step_ctr), _stream()), "bagel_sample_gumbel_bf16")
    return out


def debug_gumbel_of_u32(x):
    """Test hook: the sampler's uniform -> Gumbel map on caller-chosen uint32 draws (int32 tensor, bit pattern) -> fp32."""
    _req(x, torch.int32, "gumbel_of.x")
    g = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().bagel_debug_gumbel_of_u32(_ptr(x), _ptr(g), x.numel(), _stream()), "bagel_debug_gumbel_of_u32")
    return g


def argmax(logits):
    _req(logits, BF16, "argmax.logits")
    out = torch.empty((logits.shape[0],), dtype=torch.int64, device=logits.device)
    check(lib().bagel_argmax_bf16(_ptr(logits), logits.stride(0), _ptr(out), logits.shape[0], logits.shape[1], _stream()),
          "bagel_argmax_bf16")
    return out
