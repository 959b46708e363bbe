"""Builds tests/fake_rccl.so, the test-only loopback stand-in for the nine nccl* entry points libpco_gfx.so dlopens (tests/fake_rccl.cpp).
Kept apart from the test modules so that __graft_entry__.build() can call it without importing pytest or the oracle binding."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    so, src = os.path.join(HERE, "fake_rccl.so"), os.path.join(HERE, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")   # (the same discovery as pcodec_amd.build)
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so])
    return so
