#!/usr/bin/env python
"""Fold the per-kernel PMC averages written by tools/gpu_pmc.sh (pmc_denoise_*.txt, pmc_decode_*.txt) into one JSON:

    python tools/pmc_make_summary.py <dir with the .txt files> <out.json>

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KB and counts HALF the bytes of wide (16 B/lane) streaming
reads on gfx950 -> doubled; WRITE_SIZE in KB taken as is (calibrated in round 1 on an in-place kernel).  The summary is stamped with
the git commit and the sha1 of the kernel sources its figures belong to; bench.py quotes `traffic` only while those digests match."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    """-> {kernel: {counter: (avg, n)}}"""
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        if not line.startswith(" "):
            cur = re.sub(r"^void\s+", "", line.strip())
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+([0-9.eE+-]+)\s+\(n=(\d+)\)", line)
            if m and cur is not None:
                out[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


def digest(names):
    h = hashlib.sha1()
    for n in names:
        with open(os.path.join(ROOT, "bagel_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def find(d, prefix):
    for f in sorted(os.listdir(d)):
        if f.startswith(prefix) and f.endswith(".txt"):
            yield os.path.join(d, f)


def git_commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        return ""


def stamp(path):
    """`--stamp <summary.json>`: run in the build container after copying a summary out of gpurun_out/ (the GPU box has no .git, so `commit` is
    empty there): fills in the commit whose kernel sources match the summary's digests, and refuses when they do not."""
    d = json.load(open(path))
    now = {"gemm": digest(["gemm.hip", "common.h"]), "decode": digest(["decode.hip", "skinny.hip", "common.h"]), "attention": digest(["attention2.hip", "common.h"])}
    stale = [k for k in now if d.get("source_digest", {}).get(k) != now[k]]
    if stale:
        sys.exit(f"{path}: kernel sources changed since the PMC run ({stale}): not stamping")
    d["commit"] = git_commit() + (" (+ uncommitted changes)" if subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "bagel_amd/csrc"], capture_output=True, text=True).stdout.strip() else "")
    json.dump(d, open(path, "w"), indent=1)
    print(f"{path}: commit = {d['commit']}")


def main():
    if sys.argv[1] == "--stamp":
        return stamp(sys.argv[2])
    src, dst = sys.argv[1], sys.argv[2]
    m = re.match(r"(r\d+)_", os.path.basename(dst))
    tag = m.group(1) if m else "rNN"                      # the round the raw per-kernel files are committed under (profiles/<tag>_pmc_*.txt)
    den = {}
    for f in find(src, "pmc_denoise_"):
        for k, c in parse(f).items():
            den.setdefault(k, {}).update(c)
    kernels = {}
    M, H, I = 32768, 3584, 18944
    algo = {"gemm_pq_kernel<0, false": 2.0 * (M * H + 2 * I * H) + 2.0 * M * I,                       # gate+up: A + W read, act written
            "gemm_pq_kernel<2, false": 2.0 * (M * H + 4608 * H) + 2.0 * M * 4608,                     # qkv   (keys = name prefixes: the SGPR-base-DMA
                                                                                                      #        instantiations carry a third argument)
            # the stream-batched denoise launch: Q read + O written (8 x 4098 rows x 28 heads x 128) + K and V^T of every sample once
            # (prefix "attn2_kernel<128": round 6 gave the kernel a second template argument -- "<128, false>" is the unified-ring form the product launches)
            "attn2_kernel<128": 2.0 * 2 * (8 * 4098) * 3584 + 2.0 * 2 * (8 * 4098 + 4 * 32) * 512}
    for k, c in den.items():
        if not (k.startswith("gemm_p") or k.startswith("attn") or k.startswith("rmsnorm") or k.startswith("qknorm")):
            continue
        e = {"launches_sampled": max((n for _, n in c.values()), default=0)}
        if "FETCH_SIZE" in c:
            e["fetch_size_kb_raw"] = c["FETCH_SIZE"][0]
        if "WRITE_SIZE" in c:
            e["write_size_kb"] = c["WRITE_SIZE"][0]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["traffic_bytes_per_launch_corrected"] = int((2 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024)
            e["traffic_bytes_per_launch_raw"] = int((c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024)
        for pre, b in algo.items():
            if k.startswith(pre):
                e["algorithmic_bytes_per_launch"] = int(b)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            e["tcc_hit"], e["tcc_miss"] = c["TCC_HIT_sum"][0], c["TCC_MISS_sum"][0]
            e["l2_hit_rate"] = round(e["tcc_hit"] / max(e["tcc_hit"] + e["tcc_miss"], 1.0), 4)
            e["tcc_miss_x_128B"] = int(e["tcc_miss"] * 128)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            e["sq_valu_mfma_busy_cycles"], e["grbm_gui_active"] = c["SQ_VALU_MFMA_BUSY_CYCLES"][0], c["GRBM_GUI_ACTIVE"][0]
            e["mfma_busy_frac"] = round(e["sq_valu_mfma_busy_cycles"] / 1024.0 / (e["grbm_gui_active"] / 8.0), 4)
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"][0]
            e["wave_cycles_split"] = {n: round(c[m][0] / wc, 4) for n, m in (("parked_waitcnt_barrier", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"),
                                                                              ("issuing", "SQ_ACTIVE_INST_ANY")) if m in c}
        kernels[k] = e
    # launch-weighted aggregate of the persistent GEMM instantiations (what bench.py's roofline object reports)
    g = {k: v for k, v in kernels.items() if k.startswith("gemm_pq_kernel") and "traffic_bytes_per_launch_corrected" in v}
    if g:
        n = sum(v["launches_sampled"] for v in g.values())
        agg = {"launches_sampled": n,
               "traffic_bytes_per_launch_corrected": int(sum(v["traffic_bytes_per_launch_corrected"] * v["launches_sampled"] for v in g.values()) / n),
               "traffic_bytes_per_launch_raw": int(sum(v["traffic_bytes_per_launch_raw"] * v["launches_sampled"] for v in g.values()) / n),
               "note": "launch-weighted average over the instantiations the denoise forward launches (gate+up : o + down : qkv = 1 : 2 : 1 per layer, M = 32 768 "
                       "latent rows of a stream-batched forward)"}
        if all("l2_hit_rate" in v for v in g.values()):
            agg["l2_hit_rate"] = round(sum(v["tcc_hit"] for v in g.values()) / sum(v["tcc_hit"] + v["tcc_miss"] for v in g.values()), 4)
        if all("mfma_busy_frac" in v for v in g.values()):
            agg["mfma_busy_frac"] = round(sum(v["sq_valu_mfma_busy_cycles"] * v["launches_sampled"] for v in g.values())
                                          / sum(v["grbm_gui_active"] / 8.0 * 1024.0 * v["launches_sampled"] for v in g.values()), 4)
        kernels["gemm_pq_kernel<*>"] = agg
    dec = {}
    for f in find(src, "pmc_decode_"):
        for k, c in parse(f).items():
            dec.setdefault(k, {}).update(c)
    decode = None
    step_f = step_w = 0.0
    per = {}
    for k, c in dec.items():
        if not (k.startswith("gemv_kernel") or k.startswith("attn_decode") or k.startswith("decode_") or k.startswith("argmax") or k.startswith("sample_gumbel")):
            continue
        f, w = c.get("FETCH_SIZE", (0, 0)), c.get("WRITE_SIZE", (0, 0))
        n = max(f[1], w[1])
        per[k] = {"fetch_kb_raw": f[0], "write_kb": w[0], "launches_sampled": n}
    if per:
        # launches per decode step: the sampled run has 8 warm-up + 24 timed tokens in two prefill+decode rounds; per-step counts from the launch sequence
        # (one token-selection launch per decode step: argmax on the greedy legs, the Gumbel sampler on the sampled one)
        sel = [v["launches_sampled"] for k, v in per.items() if k.startswith("argmax") or k.startswith("sample_gumbel")]
        steps = max(1, sum(sel)) if sel else 32
        for k, v in per.items():
            v["launches_per_step"] = round(v["launches_sampled"] / steps, 2)
            step_f += v["fetch_kb_raw"] * v["launches_sampled"] / steps
            step_w += v["write_kb"] * v["launches_sampled"] / steps
        L, Hd, Id, V, nkv, hd, ctx = 28, 3584, 18944, 152064, 4, 128, 4936 + 16
        algo_step = 2.0 * (L * (2 * Hd * Hd + 2 * Hd * nkv * hd + 3 * Hd * Id) + V * Hd) + 2.0 * nkv * hd * 2 * L * ctx
        decode = {"per_kernel": per, "traffic_bytes_per_step_corrected": int((2 * step_f + step_w) * 1024), "algorithmic_bytes_per_step": int(algo_step),
                  "steps_sampled": steps}
    commit = git_commit()                                 # empty on the GPU box (no .git there): `--stamp` fills it in after the copy into profiles/
    out = {"source": f"tools/gpu_pmc.sh (rocprofv3 --kernel-trace --pmc <one group per pass>); raw per-kernel averages in profiles/{tag}_pmc_denoise_*.txt / "
                     f"profiles/{tag}_pmc_decode_*.txt",
           "correction": "FETCH_SIZE (KB) doubled for the 16-B/lane streaming patterns (MI355X_MICROARCH.md, HBM section; cross-checked in round 1 against "
                         "TCC_MISS x 128 B and on an in-place kernel); WRITE_SIZE (KB) as reported",
           "commit": commit,
           "source_digest": {"gemm": digest(["gemm.hip", "common.h"]), "decode": digest(["decode.hip", "skinny.hip", "common.h"]),
                             "attention": digest(["attention2.hip", "common.h"])},
           "kernels": kernels, "decode_step": decode}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
