"""Build a variant of libcanonswap_hip.so for same-box A/B runs: python tools/build_variant.py NAME [extra hipcc flags...]
-> ab/NAME.so (objects under canonswap_amd/build_NAME/).  Use on the GPU box through CANONSWAP_LIB=ab/NAME.so."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from canonswap_amd import _lib  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
out = os.path.join(ROOT, "ab", name + ".so")
print(_lib.build(lib_path=out, extra_flags=flags, obj_dir=os.path.join(ROOT, "canonswap_amd", "build_" + name)))
