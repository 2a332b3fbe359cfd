#!/usr/bin/env python
"""Debug aid (round 6): rebuild csrc/attention_prefill.hip (UMV_ATTN_PAIR_DEBUG forms included) from PATCHED device assembly and link it with
the product's other objects into tools/bin/libunimedvl_hip_<name>.so - hipcc's own steps by hand (device asm -> patch -> assemble -> lld ->
offload bundle -> host compile with that bundle).  Patches (applied to the paired-call kernel attn_prefill_kernel<128, 2, 2> only, unless said):
  movb64      every register-to-register v_mov_b64 of the file -> two v_mov_b32
  always_rare the branch that skips the rare path of the lazy softmax -> s_nop (the rare path runs in every block; rows that did not trigger
              select d = 0, alpha = 1: same results)
  wait_merge  s_waitcnt vmcnt(0) lgkmcnt(0) + s_nop 7 at every label inside the key loop
  nop_branch  s_nop 7 in front of every conditional branch inside the key loop
  mfma_nop    s_nop 15 x 2 behind every v_mfma inside the key loop (an MFMA has finished before anything else issues)
  ds_wait     s_waitcnt lgkmcnt(0) behind every ds_read inside the key loop (an LDS read has landed before anything else issues)
  mfma_pre    s_nop 15 x 2 IN FRONT of every v_mfma inside the key loop
  occ1        the kernel descriptor claims 512 registers per lane (accum_offset 256): ONE wave per SIMD, same code
usage: python tools/asm_patch.py <name> <patch> [<patch> ...]   then
       UMV_LIB_PATH=tools/bin/libunimedvl_hip_<name>.so UMV_ATTN_PAIR_DEBUG=1 python tools/attn_pair_debug.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LL = "/opt/rocm/lib/llvm/bin"
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form -DUMV_ATTN_PAIR_DEBUG=1".split()
PAIR = "_Z19attn_prefill_kernelILi128ELi2ELi2ELb0ELb0EEv13umv_attn_argsfi"


def sh(*cmd, **kw):
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, **kw)


def patch_pair(s, patches):
    a = s.index(PAIR + ":")
    b = s.index(".Lfunc_end", a)
    lines = s[a:b].split("\n")
    out, in_loop, n = [], False, {p: 0 for p in patches}
    for i, ln in enumerate(lines):
        if ln.startswith(".LBB") or ln.startswith("; %bb."):
            in_loop = "Loop" in ln
            out.append(ln)
            if in_loop and ln.startswith(".LBB") and "wait_merge" in patches:
                out += ["\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_nop 7"]
                n["wait_merge"] += 1
            continue
        st = ln.strip()
        if in_loop and st.startswith("s_cbranch_vccz") and "always_rare" in patches and any("v_permlane16_swap" in x for x in lines[i + 1:i + 8]):
            out.append("\ts_nop 0")
            n["always_rare"] += 1
            continue
        if in_loop and st.startswith("s_cbranch_") and "nop_branch" in patches:
            out.append("\ts_nop 7")
            n["nop_branch"] += 1
        if in_loop and st.startswith("v_mfma") and "mfma_pre" in patches:
            out += ["\ts_nop 15", "\ts_nop 15"]
            n["mfma_pre"] += 1
        out.append(ln)
        if in_loop and st.startswith("v_mfma") and "mfma_nop" in patches:
            out += ["\ts_nop 15", "\ts_nop 15"]
            n["mfma_nop"] += 1
        if in_loop and st.startswith("ds_read") and "ds_wait" in patches:
            out.append("\ts_waitcnt lgkmcnt(0)")
            n["ds_wait"] += 1
    print("patched the paired kernel:", n)
    return s[:a] + "\n".join(out) + s[b:]


def main():
    name, patches = sys.argv[1], set(sys.argv[2:])
    w = tempfile.mkdtemp()
    for f in os.listdir(os.path.join(ROOT, "unimedvl_amd", "csrc")):
        if f.endswith(".h"):
            open(os.path.join(w, f), "w").write(open(os.path.join(ROOT, "unimedvl_amd", "csrc", f)).read())
    src = open(os.path.join(ROOT, "unimedvl_amd", "csrc", "attention_prefill.hip")).read()
    open(os.path.join(w, "ap.hip"), "w").write(src.replace('"../../include/unimedvl_hip.h"', f'"{ROOT}/include/unimedvl_hip.h"'))
    sh("/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", "ap.s", "ap.hip", cwd=w)
    s = open(os.path.join(w, "ap.s")).read()
    if "movb64" in patches:
        s, k = re.subn(r"v_mov_b64_e32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]",
                       lambda m: f"v_mov_b32_e32 v{m.group(1)}, v{m.group(3)}\n\tv_mov_b32_e32 v{m.group(2)}, v{m.group(4)}", s)
        print(f"split {k} v_mov_b64")
    if "occ1" in patches:
        a = s.index(".amdhsa_kernel " + PAIR)
        b = s.index(".end_amdhsa_kernel", a)
        blk = re.sub(r"\.amdhsa_next_free_vgpr \d+", ".amdhsa_next_free_vgpr 512", s[a:b])
        blk = re.sub(r"\.amdhsa_accum_offset \d+", ".amdhsa_accum_offset 256", blk)
        s = s[:a] + blk + s[b:]
        print("occupancy of the paired kernel forced to one wave per SIMD")
    if patches - {"movb64", "occ1", "plain"}:
        s = patch_pair(s, patches - {"movb64", "occ1", "plain"})
    open(os.path.join(w, "ap_patched.s"), "w").write(s)
    sh(f"{LL}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", "ap_patched.s", "-o", "dev.o", cwd=w)
    sh(f"{LL}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", "dev.out", "dev.o", cwd=w)
    sh(f"{LL}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
       "-input=/dev/null", "-input=dev.out", "-output=dev.hipfb", cwd=w)
    sh("/opt/rocm/bin/hipcc", *FLAGS, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", "dev.hipfb", "-c", "ap.hip", "-o", "ap_patched.o", cwd=w)
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    objs = [os.path.join(ROOT, "unimedvl_amd", "lib", f + ".o") for f in ("host_error", "elementwise", "pack", "gemm", "gemm_w4", "gemm_fp8mfma", "attention", "vision")]
    lib = os.path.join(ROOT, "tools", "bin", f"libunimedvl_hip_{name}.so")
    sh("/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, os.path.join(w, "ap_patched.o"))
    print("built", lib)


if __name__ == "__main__":
    main()
