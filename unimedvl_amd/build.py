"""Build libunimedvl_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m unimedvl_amd.build [--force]

The library is built IN-TREE (unimedvl_amd/lib/) so that it travels with the repo
snapshot to the GPU box.  No torch headers are involved: the ABI is plain C
(include/unimedvl_hip.h).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libunimedvl_hip.so")
SOURCES = ["elementwise.hip", "gemm.hip", "gemm_decode.hip", "decode_engine.hip", "gemm_fp8mfma.hip", "attention.hip", "attention_prefill.hip", "attention_decode.hip", "vision.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-fgpu-rdc" if False else "-fno-gpu-rdc"]
# per-file additions.  attention_prefill: MFMA destinations stay in VGPRs (the compiler's default parks the 64 O accumulators in
# AGPRs and moves them out and back around every rescale)
FILE_FLAGS = {"attention_prefill.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stamp():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/unimedvl_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = os.path.join(LIBDIR, "build.stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp_file, "w").write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
