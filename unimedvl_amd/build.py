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
# the product library (include/unimedvl_hip.h): everything the entry points call.  The kernels that were measured and not adopted
# live in experimental/ with a build target of their own (python -m experimental.build).
SOURCES = ["host_error.hip", "elementwise.hip", "pack.hip", "gemm.hip", "gemm_w4.hip", "gemm_fp8mfma.hip", "attention.hip", "attention_prefill.hip", "vision.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-fgpu-rdc" if False else "-fno-gpu-rdc"]
# per-file additions.  attention_prefill: MFMA destinations stay in VGPRs (the compiler's default parks the 64 O accumulators in
# AGPRs and moves them out and back around every rescale)
FILE_FLAGS = {"attention_prefill.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# UMV_GEMM_ABLATIONS=1: also instantiate the tiled GEMM's timing-only ablations and its 32x32x16 variant (UMV_GEMM_TILE=966x / 566...,
# git history: tools/r04_gemm_abl.sh); never in the default build
if os.environ.get("UMV_ATTN_TRACE", "0") not in ("0", ""):      # timing study of the prefill attention (tools/attn_trace.py); never in the default build
    FILE_FLAGS["attention_prefill.hip"] = FILE_FLAGS["attention_prefill.hip"] + ["-DUMV_ATTN_TRACE"]
if os.environ.get("UMV_ATTN_PAIR_DEBUG", "0") not in ("0", ""):  # bisecting forms of the paired lazy-softmax kernel (tools/attn_pair_debug.py); never in the default build
    FILE_FLAGS["attention_prefill.hip"] = FILE_FLAGS["attention_prefill.hip"] + ["-DUMV_ATTN_PAIR_DEBUG=1"]
if os.environ.get("UMV_GEMM_ABLATIONS", "0") not in ("0", ""):
    FILE_FLAGS["gemm.hip"] = ["-DUMV_GEMM_ABLATIONS"]
    FILE_FLAGS["gemm_w4.hip"] = ["-DUMV_GEMM_ABLATIONS"]


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
    procs = []
    groups = []
    # per-object stamps: a source is recompiled only when it, a header (any csrc/*.h or the public header) or its flags changed
    hh = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/unimedvl_hip.h"]:
        if f.endswith(".h"):
            hh.update(open(os.path.join(CSRC, f), "rb").read())
    headers = hh.hexdigest()
    obj_stamps = {}
    for lib, sources, extra, suffix in ((LIB, SOURCES, [], ""),):
        objs = []
        for src in sources:
            sp = os.path.join(CSRC, src)
            obj = os.path.join(LIBDIR, src.replace(".hip", suffix + ".o"))
            cmd = [hipcc] + FLAGS + extra + FILE_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
            ostamp = hashlib.sha256((headers + " ".join(cmd)).encode() + open(sp, "rb").read()).hexdigest()
            objs.append(obj)
            if not force and os.path.exists(obj) and os.path.exists(obj + ".stamp") and open(obj + ".stamp").read() == ostamp:
                continue
            if os.path.exists(obj + ".stamp"):
                os.remove(obj + ".stamp")
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            obj_stamps[src] = (obj + ".stamp", ostamp)
        groups.append((lib, objs))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out.strip():
            print(out.decode())
        open(obj_stamps[src][0], "w").write(obj_stamps[src][1])
    for lib, objs in groups:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    open(stamp_file, "w").write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
