"""Measurement helpers (micro-benchmarks, rocprofv3 wrappers, probes); nothing here is imported by the product."""
