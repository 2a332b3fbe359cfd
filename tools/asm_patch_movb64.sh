#!/bin/bash
# Debug aid (round 6, profiles/r06_attn_pair_nondeterminism.txt): rebuild csrc/attention_prefill.hip (UMV_ATTN_PAIR_DEBUG forms included) with
# every register-to-register v_mov_b64 of the DEVICE assembly split into two v_mov_b32, and link it with the product's other objects into
# tools/bin/libunimedvl_hip_nomovb64.so.  hipcc's own steps, done by hand: device asm -> (patch) -> assemble -> lld -> offload bundle ->
# host compile with that bundle.   UMV_LIB_PATH=tools/bin/libunimedvl_hip_nomovb64.so UMV_ATTN_PAIR_DEBUG=1 python tools/attn_pair_debug.py
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD; W=$(mktemp -d); LL=/opt/rocm/lib/llvm/bin
cp unimedvl_amd/csrc/*.h $W/
sed "s#\"../../include/unimedvl_hip.h\"#\"$ROOT/include/unimedvl_hip.h\"#" unimedvl_amd/csrc/attention_prefill.hip > $W/ap.hip
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form -DUMV_ATTN_PAIR_DEBUG=1"
cd $W
/opt/rocm/bin/hipcc $F -S --cuda-device-only -o ap.s ap.hip 2>/dev/null
python3 - <<'PY'
import re
s = open("ap.s").read()
s2, n = re.subn(r"v_mov_b64_e32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", lambda m: f"v_mov_b32_e32 v{m.group(1)}, v{m.group(3)}\n\tv_mov_b32_e32 v{m.group(2)}, v{m.group(4)}", s)
open("ap_patched.s", "w").write(s2)
print(f"split {n} v_mov_b64 (left: {s2.count('v_mov_b64')} with literal sources)")
PY
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c ap_patched.s -o dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o dev.out dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev.out -output=dev.hipfb
/opt/rocm/bin/hipcc $F --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang dev.hipfb -c ap.hip -o ap_patched.o 2>/dev/null
cd $ROOT; mkdir -p tools/bin
objs=""; for f in host_error elementwise pack gemm gemm_w4 gemm_fp8mfma attention vision; do objs="$objs unimedvl_amd/lib/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libunimedvl_hip_nomovb64.so $objs $W/ap_patched.o
echo built tools/bin/libunimedvl_hip_nomovb64.so
