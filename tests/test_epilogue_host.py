"""The shared conv epilogue (canonswap_amd/csrc/conv_epilogue.h) executed on the HOST: tests/epilogue_host/harness.cpp emulates
the lanes of a workgroup one by one over whole tensors (2-D / volume / tiny-spatial tiles, ragged channels, odd batches, fp32 and
fp16 residuals, second output, per-position scale, SPADE with an up-sampled operand, T blend, pixel shuffle, statistics) and
prints a checksum per case.  tests/epilogue_host/expected.txt holds the checksums of the version whose results the GPU parity
tests validated, so any later edit of the macro (addressing, prefetch order, activation forms) that changes a single stored bit
shows up here without a GPU.
Round 3: the out1 affine and the SPADE modulation say fmaf explicitly (every copy of the epilogue must fuse alike, conv_epilogue.h): the
checksums of cases 1 (out1), 3 and 7 (out1) changed with that edit, every other one stayed; cases 9-12 run the tensor combinations that have
a branch-free copy (EP_FAST), and the harness is built with and without those copies - the same checksums."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "epilogue_host")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang (ext_vector_type, _Float16)")
def test_epilogue_macro_on_host(tmp_path):
    csrc = os.path.join(ROOT, "canonswap_amd", "csrc")
    common = open(os.path.join(csrc, "common.h")).read().replace("#include <hip/hip_runtime.h>", "")
    (tmp_path / "common_stub.h").write_text(common[: common.index("#define CS_CHECK_HIP")])
    (tmp_path / "ep.h").write_text(open(os.path.join(csrc, "conv_epilogue.h")).read().replace('#include "common.h"', ""))
    shutil.copy(os.path.join(HERE, "harness.cpp"), tmp_path / "h.cpp")
    # The harness fills the accumulators as a function of (packed weight row, position), so the kernels' channel pairing (EP_PAIR:
    # which lane / fragment a weight row goes to, joint 16-byte stores of a lane's 8 channels) must give the same bytes as the
    # plain mapping: both variants are compared with the same expected checksums.
    # ... and with / without the branch-free copies of the epilogue (EP_FAST): the same bits (tests/test_gpu_epilogue_fast.py on the GPU)
    for k, flags in enumerate((["-DHARNESS_NO_PAIRING"], [], ["-DHARNESS_NO_FAST"], ["-DHARNESS_NO_PAIRING", "-DHARNESS_NO_FAST"])):
        exe = str(tmp_path / ("h" + str(k)))
        subprocess.run([CLANG, "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-everything", "-DEP_HOST_EMULATION", '-DEPH="ep.h"', *flags, "h.cpp", "-o", exe],
                       cwd=tmp_path, check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
        assert out == open(os.path.join(HERE, "expected.txt")).read(), flags
