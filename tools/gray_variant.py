"""A/B builds of ONE kernel file: recompiles csrc/<file>.hip with extra flags and links it with the shipped objects into
competitive_rl_amd/libcrl_hip_<tag>.so (loaded with CRL_LIB_VARIANT=<tag>).  Usage: python tools/gray_variant.py <tag> <file> <flags...>"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from competitive_rl_amd import build as B

tag, fname, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
objs = []
for s in B.SOURCES:
    o = os.path.join(B.CSRC, s.replace(".hip", ".o"))
    if s == fname:
        o = os.path.join(B.CSRC, tag + "_" + s.replace(".hip", ".o"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *flags, "-I", os.path.join(B.ROOT, "include"), "-c", os.path.join(B.CSRC, s), "-o", o])
    objs.append(o)
out = os.path.join(B.PKG, "libcrl_hip_%s.so" % tag)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print(out)
