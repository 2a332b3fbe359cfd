#!/usr/bin/env python
"""Sustained timing of umv_gemm_bf16 under several forced tile configurations, one process per (shape, arm):
    ARMS=266,466,4664 SHAPES="8192,8192,8192;2064,37888,3584,swiglu" SECONDS=2 python tools/tile_arms.py
An arm is a UMV_GEMM_TILE value, optionally with env assignments appended: "266:UMV_GEMM_XLINE=0"."""
import os
import subprocess as sp
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
arms = os.environ.get("ARMS", "266,466").split(",")
shapes = os.environ.get("SHAPES", "8192,8192,8192;2064,37888,3584,swiglu;8208,3584,18944").split(";")
secs = os.environ.get("SECONDS", "2")
for shape in shapes:
    for arm in arms:
        parts = arm.split(":")
        env = dict(os.environ, UMV_GEMM_TILE=parts[0], SHAPE=shape, SECONDS=secs)
        for kv in parts[1:]:
            k, v = kv.split("=")
            env[k] = v
        r = sp.run([sys.executable, os.path.join(ROOT, "tools", "gemm_power.py")], capture_output=True, text=True, timeout=600, env=env)
        line = (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]
        print(f"arm {arm:>12s} {line}", flush=True)
