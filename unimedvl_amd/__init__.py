"""unimedvl_amd - UniMedVL's forward path (ViT encode, Qwen2-MoT prefill / decode, rectified-flow image head, FLUX VAE)
on AMD Instinct MI355X: a Python host mirror of the reference API over hand-written gfx950 kernels behind a C ABI
(include/unimedvl_hip.h -> unimedvl_amd/lib/libunimedvl_hip.so).  There is no CPU fallback: importing is cheap and has no
side effects, every op raises without the library or a GPU.  See DESIGN.md / INTEGRATION.md."""

__version__ = "0.1.0"
