"""Kernels that were built, checked against the product kernels and MEASURED, but are not on the product path (DESIGN.md
section 5b says why each of them lost): the persistent decode GEMM, the fused decode attention, the L2 prefetch, the decode
layer engine and the prefill attention on the 32x32x16 matrix instruction.  A package of its own with its own build target

    python -m experimental.build          ->  experimental/lib/libunimedvl_hip_experimental.so

Nothing in unimedvl_amd imports it; __graft_entry__.build() does not build it (UMV_BUILD_EXPERIMENTAL=1 does); its tests skip when
the library has not been built."""
