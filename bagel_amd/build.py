"""Build libbagel_hip.so (gfx950) in-tree with hipcc.  No torch headers, no cmake: one .o per .hip, one link.

    python -m bagel_amd.build [--force] [--keep-temps]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_build")
LIB = os.path.join(HERE, "libbagel_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wno-unused-value",
         "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libbagel_hip.so)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    """Content hash of the inputs of one object (source + shared headers + flags): robust against the mtime churn of a
    repository snapshot copied to another box, unlike a timestamp comparison."""
    import hashlib
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()


def _manifest_path():
    return os.path.join(OBJ, "manifest.json")


def _load_manifest():
    import json
    try:
        with open(_manifest_path()) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def build(force=False, keep_temps=False, verbose=True):
    import json
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(os.path.dirname(HERE), "include", "bagel_hip.h")]
    manifest = _load_manifest()
    digests = {s: _digest([os.path.join(CSRC, s)] + headers) for s in sources()}       # (common.h includes the public header: it is an input of every object)
    todo = [s for s in sources() if force or manifest.get(s) != digests[s] or not os.path.exists(os.path.join(OBJ, s[:-4] + ".o"))]
    if not todo and os.path.exists(LIB) and manifest.get("__lib__") == sorted(digests.values()):
        return LIB                                     # everything up to date (by content)
    try:
        hipcc = _hipcc()
    except RuntimeError:
        if os.path.exists(LIB) and not todo:
            return LIB
        raise
    jobs = []
    for s in todo:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if keep_temps:
            cmd.append("-save-temps=obj")
        jobs.append((s, cmd))
    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        return name, r
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name, r in ex.map(run, jobs):
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError(f"hipcc failed on {name}")
                if verbose:
                    print(f"[bagel_amd.build] compiled {name}")
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link of libbagel_hip.so failed")
    if verbose:
        print(f"[bagel_amd.build] linked {LIB}")
    manifest = dict(digests)
    manifest["__lib__"] = sorted(digests.values())
    with open(_manifest_path(), "w") as f:
        json.dump(manifest, f, indent=1)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--keep-temps", action="store_true")
    a = ap.parse_args()
    build(a.force, a.keep_temps)
