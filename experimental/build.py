"""Build experimental/lib/libunimedvl_hip_experimental.so for gfx950 (cross-compiles without a GPU).

    python -m experimental.build [--force]

Shares common.h / gemm_epilogue.h / attention_combine.h with the product kernels (-I unimedvl_amd/csrc); exports exactly what
experimental/include/unimedvl_hip_experimental.h declares.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libunimedvl_hip_experimental.so")
SOURCES = ["host_error_exp.hip", "gemm_decode.hip", "attention_decode.hip", "attention_prefill32.hip", "decode_engine.hip", "prefetch.hip"]
INCLUDES = [os.path.join(ROOT, "unimedvl_amd", "csrc"), os.path.join(ROOT, "include"), os.path.join(HERE, "include")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-fno-gpu-rdc"]


def _stamp():
    h = hashlib.sha256()
    for d in [CSRC] + INCLUDES:
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if os.path.isfile(p) and f.endswith((".h", ".hip")):
                h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = os.path.join(LIBDIR, "build.stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    inc = [f"-I{d}" for d in INCLUDES]
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + inc + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    open(stamp_file, "w").write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
