"""ctypes binding of libcanonswap_hip.so (C ABI: include/canonswap_hip.h) and its in-tree build.

The library is the product: if it cannot be loaded, every entry point raises -- there is no CPU or
PyTorch fallback (the CPU restatement under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("conv_halo.hip", "conv_wide.hip", "conv_lat.hip", "vol32.hip", "vol32_fused.hip", "kernels.hip", "motion.hip", "imgops.hip", "engine.hip")]
# test-only cross-check kernel (the first-generation implicit-GEMM conv): its own library, never linked into the product
TEST_SRC = os.path.join(os.path.dirname(HERE), "tests", "csrc", "test_igemm.hip")
TEST_LIB_PATH = os.path.join(os.path.dirname(HERE), "tests", "libcanonswap_test.so")
HEADERS = [os.path.join(CSRC, f) for f in ("common.h", "conv_epilogue.h", "conv_halo_kernel.h")] + \
          [os.path.join(os.path.dirname(HERE), "include", "canonswap_hip.h")]
HALO_NGROUPS = 8          # conv_halo.hip is compiled once per -DHALO_GROUP=k (slices of its instantiation table)
OBJ_DIR = os.path.join(HERE, "build")
LIB_PATH = os.environ.get("CANONSWAP_LIB") or os.path.join(HERE, "libcanonswap_hip.so")      # CANONSWAP_LIB: A/B builds (tools/)
ABI_SYMBOLS = [
    "cs_create", "cs_destroy", "cs_last_error", "cs_abi_version", "cs_upload", "cs_finalize_weights", "cs_set_identity", "cs_set_latency_mode",
    "cs_extract_feature_3d", "cs_warp", "cs_warp_out", "cs_swap", "cs_swap_ids", "cs_swap_frames_ids", "cs_refine", "cs_warp_forward", "cs_spade_decode",
    "cs_pack_u8", "cs_unpack_u8", "cs_soft_erosion", "cs_prepare_crops", "cs_warp_affine_u8", "cs_warp_affine_f32", "cs_paste_back",
    "cs_motion_extract", "cs_motion_keypoints", "cs_soft_erosion_frames", "cs_paste_back_batch", "cs_swap_frames", "cs_animate_frames", "cs_profile_begin", "cs_profile_end", "cs_profile_exec_flops", "cs_op_conv", "cs_op_grid_sample3d",
    "cs_op_chan_stats", "cs_op_chan_stats_partial_floats", "cs_op_pair_ragged", "cs_op_resblock3d", "cs_op_t_mask",
]
ABI_VERSION = 4          # CS_ABI_VERSION of include/canonswap_hip.h
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _units():
    """(source, object name, extra flags) of every translation unit."""
    units = []
    for src in SOURCES:
        base = os.path.splitext(os.path.basename(src))[0]
        if base == "conv_halo":
            units += [(src, f"conv_halo_g{g}.o", [f"-DHALO_GROUP={g}"]) for g in range(HALO_NGROUPS)]
        else:
            units.append((src, base + ".o", []))
    return units


def build(force: bool = False, lib_path: str | None = None, extra_flags=(), obj_dir: str | None = None, jobs: int | None = None) -> str:
    """Compile the HIP sources for gfx950 into canonswap_amd/libcanonswap_hip.so (hipcc cross-compiles without a GPU).

    Objects are cached under canonswap_amd/build/ and rebuilt when their source, any header or the flags changed; the
    translation units compile in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    lib_path = lib_path or LIB_PATH
    obj_dir = obj_dir or OBJ_DIR
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = HIPCC_FLAGS + list(extra_flags)
    stamp = " ".join([hipcc] + flags)
    hdr_time = max(os.path.getmtime(p) for p in HEADERS)
    todo, objs = [], []
    for src, oname, extra in _units():
        obj = os.path.join(obj_dir, oname)
        objs.append(obj)
        tag = obj + ".flags"
        fresh = (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_time, os.path.getmtime(src))
                 and os.path.exists(tag) and open(tag).read() == stamp + " " + " ".join(extra))
        if not fresh:
            todo.append((src, obj, extra, tag))

    def compile_one(job):
        src, obj, extra, tag = job
        subprocess.run([hipcc, *flags, *extra, "-c", src, "-o", obj], check=True)
        with open(tag, "w") as f:
            f.write(stamp + " " + " ".join(extra))

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(lib_path) or os.path.getmtime(lib_path) < max(os.path.getmtime(o) for o in objs):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib_path], check=True)
    return lib_path


def build_test_lib(force: bool = False) -> str:
    """tests/libcanonswap_test.so = tests/csrc/test_igemm.hip (+ conv_igemm.hip): used by tests/hip_ops.py only."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [TEST_SRC, os.path.join(os.path.dirname(TEST_SRC), "conv_igemm.hip")] + HEADERS
    if force or not os.path.exists(TEST_LIB_PATH) or os.path.getmtime(TEST_LIB_PATH) < max(os.path.getmtime(d) for d in deps):
        subprocess.run([hipcc, *HIPCC_FLAGS, "-shared", "-Wl,-Bsymbolic", TEST_SRC, "-o", TEST_LIB_PATH], check=True)
    return TEST_LIB_PATH


_test_lib = None


def load_test_lib():
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise RuntimeError(f"{TEST_LIB_PATH} is missing: run __graft_entry__.build()")
        lib = C.CDLL(TEST_LIB_PATH)
        lib.cs_test_last_error.restype = C.c_char_p
        lib.cs_test_conv_igemm.argtypes = [C.POINTER(ConvDesc), C.c_void_p]
        _test_lib = lib
    return _test_lib


def toolchain() -> str:
    """hipcc version string (recorded by build(): the dynamic-shape conv kernels rely on hand-counted waits, see conv_halo_kernel.h)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    return " | ".join(l.strip() for l in out.splitlines() if "version" in l.lower())[:200]


def isa_check(obj_dir: str | None = None) -> dict:
    """Compile-time guard for the hand-counted weight ring of the dynamic-shape conv kernels (conv_halo_kernel.h, ST = 0).

    Those kernels stream weight fragments with inline-asm `global_load_dwordx4` the compiler does not track and wait for them with
    an explicit `s_waitcnt vmcnt(WCH * (PFD - 1))`.  A compiler change that moves, duplicates or drops one of these would read stale
    registers without any diagnostic.  This disassembles every conv_halo object and checks, per
    ST = 0 kernel: exactly PFD = 4 counted waits `s_waitcnt vmcnt(3 * WCH)`; after each of them an MFMA that reads the ring slot's
    registers and then that slot's reload (WCH `global_load_dwordx4` into the same registers the prologue primed it with); and the
    final `s_waitcnt vmcnt(0)` drain before the epilogue.  Raises RuntimeError on any mismatch; returns {kernel: waits found}."""
    import re
    obj_dir = obj_dir or OBJ_DIR
    objdump = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not os.path.exists(objdump):
        raise FileNotFoundError(f"isa_check: {objdump} not found")      # build() reports the check as skipped; a failed check is a RuntimeError
    seen = dict(_isa_check_wide(obj_dir, objdump))
    seen.update(_isa_check_lat(obj_dir, objdump))
    for g in range(HALO_NGROUPS):
        obj = os.path.join(obj_dir, f"conv_halo_g{g}.o")
        subprocess.run([objdump, "--offloading", obj], check=True, capture_output=True)
        co = obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
        try:
            txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        finally:
            for f in os.listdir(obj_dir):
                if f.startswith(f"conv_halo_g{g}.o.0."):
                    os.remove(os.path.join(obj_dir, f))
        for m in re.finditer(r"^[0-9a-f]+ <(_Z16conv_halo_kernelILi\d+ELi\d+ELi(\d+)ELi\d+ELi\d+ELi\d+ELb[01]ELb[01]ELi0EEv10ConvParams)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)",
                             txt, re.S | re.M):
            name, wch, body = m.group(1), int(m.group(2)), m.group(3).split("\n")
            want = f"s_waitcnt vmcnt({3 * wch})"
            kend = max(i for i, l in enumerate(body) if "v_mfma" in l)        # the epilogue's own (compiler-counted) waits are not the ring's
            waits = [i for i, l in enumerate(body) if i < kend and l.strip().startswith(want) and "lgkmcnt" not in l]
            if len(waits) < 4:
                raise RuntimeError(f"isa_check: {name}: {len(waits)} x `{want}` in the K loop, expected 4 (one per ring slot)")
            waits = waits[:4]     # (a later wait with the same count is the compiler's own, for the spill stores of the loop's drain)
            loads = [(i, re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off", l)) for i, l in enumerate(body)]
            loads = [(i, int(q.group(1))) for i, q in loads if q]
            prime = [r for i, r in loads if i < waits[0]][-4 * wch:]          # the ring as the prologue primed it: 4 slots x WCH fragments
            if len(prime) != 4 * wch:
                raise RuntimeError(f"isa_check: {name}: the prologue primes {len(prime)} ring registers, expected {4 * wch}")
            for k, w in enumerate(waits):
                slot = prime[k * wch:(k + 1) * wch]
                end = waits[k + 1] if k + 1 < 4 else len(body)
                seg = body[w:end]
                rel = [i for i, r in loads if w < i < end and r in slot]
                if len(rel) < wch:
                    raise RuntimeError(f"isa_check: {name}: ring slot {k} is not reloaded into v{slot} after its wait")
                first_reload = min(rel) - w
                uses = [j for j, l in enumerate(seg[:first_reload]) if "v_mfma" in l and any(f"v[{r}:{r + 3}]" in l for r in slot)]
                if not uses:
                    raise RuntimeError(f"isa_check: {name}: no MFMA reads ring slot {k} (v{slot}) between its wait and its reload")
            tail = "\n".join(body[waits[-1]:])
            if "s_waitcnt vmcnt(0)" not in tail:
                raise RuntimeError(f"isa_check: {name}: the ring is not drained (s_waitcnt vmcnt(0)) before the epilogue")
            seen[name] = len(waits)
    if not seen:
        raise RuntimeError("isa_check: no dynamic-shape conv_halo kernel found in the objects (name mangling changed?)")
    return seen


def _isa_check_lat(obj_dir: str, objdump: str) -> dict:
    """conv_lat.hip stages its halo with LDS-DMA instructions hidden from the compiler (inline asm) next to compiler-counted weight loads,
    and waits for a chunk's DMA at the chunk's head with a hand-counted `s_waitcnt vmcnt(N)` + `s_barrier`.  vmcnt retires in order, so the
    wait covers a DMA piece iff at least N vector-memory instructions were issued after it.  Check that for every DMA piece of every such
    kernel at the first barrier-wait behind it, and that 8 chunks x 6 K-steps of MFMAs sit between 10 barriers (straight-line code)."""
    import re
    oname = "conv_lat.o"
    obj = os.path.join(obj_dir, oname)
    subprocess.run([objdump, "--offloading", obj], check=True, capture_output=True)
    co = obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
    try:
        txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    finally:
        for f in os.listdir(obj_dir):
            if f.startswith(oname + ".0."):
                os.remove(os.path.join(obj_dir, f))
    seen = {}
    for m in re.finditer(r"^[0-9a-f]+ <(\S*conv_lat_kernel\S*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S | re.M):
        name, body = m.group(1), [l.split("//")[0].strip() for l in m.group(2).split("\n")]
        body = [l for l in body if l]
        mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
        if not mf:
            raise RuntimeError(f"isa_check: {name}: no MFMA")
        if any(l.startswith(("s_cbranch", "s_branch")) for l in body[mf[0]:mf[-1]]):
            raise RuntimeError(f"isa_check: {name}: a branch inside the K loop (the walk assumes straight-line code)")
        # m0 belongs to the hidden DMA: `s_mov_b32 m0, sN` is legal only as the head of `s_mov_b32 m0 | s_nop | buffer_load ... lds`, and nothing else
        # in the kernel may read or write it (a compiler that started to keep a value in m0 would be caught here)
        for i, l in enumerate(body):
            if re.search(r"\bm0\b", l):
                ok = l.startswith("s_mov_b32 m0,") and any(b.startswith("buffer_load") and b.endswith("lds") for b in body[i + 1:i + 3])
                if not ok:
                    raise RuntimeError(f"isa_check: {name}: `{l}` touches m0 outside the hidden LDS-DMA sequence")
        nvm, open_dma, nbar, ndma = 0, [], 0, 0
        for i, l in enumerate(body[:mf[-1] + 1]):
            op = l.split()[0]
            if op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "scratch_load", "scratch_store")):
                if op.startswith("buffer_load") and l.endswith("lds"):
                    open_dma.append(nvm); ndma += 1
                nvm += 1
            if op == "s_barrier":
                nbar += 1
                # the wait that covers this barrier: the nearest vmcnt wait above it
                w = next((body[j] for j in range(i - 1, max(i - 6, -1), -1) if body[j].startswith("s_waitcnt") and "vmcnt" in body[j]), None)
                if w is None and not open_dma:
                    continue          # (the reduction's barriers behind the last chunk: nothing is in flight)
                if w is None:
                    raise RuntimeError(f"isa_check: {name}: barrier {nbar} of the K loop has no vmcnt wait in front of it")
                n = int(re.search(r"vmcnt\((\d+)\)", w).group(1))
                for idx in open_dma:      # every piece issued so far is this chunk's (the next chunk's go out behind the barrier)
                    if nvm - idx - 1 < n:
                        raise RuntimeError(f"isa_check: {name}: `{w}` before barrier {nbar} does not cover a DMA piece issued {nvm - idx - 1} vector-memory instructions earlier")
                open_dma = []
        if open_dma:
            raise RuntimeError(f"isa_check: {name}: {len(open_dma)} DMA pieces are never waited for at a chunk head")
        if len(mf) not in (8 * 6 * 8, 8 * 6 * 16) or ndma != 24:
            raise RuntimeError(f"isa_check: {name}: {len(mf)} MFMAs / {ndma} DMA instructions, expected 384 or 768 / 24")
        seen[name] = len(mf)
    if not seen:
        raise RuntimeError("isa_check: no conv_lat kernel found in conv_lat.o (name mangling changed?)")
    return seen


def _isa_check_wide(obj_dir: str, objdump: str) -> dict:
    """conv_wide.hip and the 256 x 160 tiles of conv_halo (ASMR) stream their weight ring with untracked inline-asm loads and wait with
    counted `s_waitcnt vmcnt(N)` that name the slot's registers.  Check in the disassembly of every such kernel: each MFMA's weight operand
    (src0) is a VGPR quadruple whose last writer is a `global_load_dwordx4` (never a copy of one: a copy made before the data landed would
    be stale) with a vmcnt wait between that load and the MFMA; conv_wide: 18 K-steps x 64 MFMAs per kernel."""
    seen = _isa_check_ring(obj_dir, objdump, "conv_wide.o", r"\S*conv_wide_kernel\S*", 18 * 64)
    seen.update(_isa_check_ring(obj_dir, objdump, "conv_halo_g1.o", r"_Z16conv_halo_kernelILi32ELi8ELi5ELi2ELi2ELi0ELb1ELb0ELi[789]E\S*", None))
    return seen


def _isa_check_ring(obj_dir: str, objdump: str, oname: str, name_re: str, expect_mfma) -> dict:
    import re
    obj = os.path.join(obj_dir, oname)
    subprocess.run([objdump, "--offloading", obj], check=True, capture_output=True)
    co = obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
    try:
        txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    finally:
        for f in os.listdir(obj_dir):
            if f.startswith(oname + ".0."):
                os.remove(os.path.join(obj_dir, f))
    seen = {}
    for m in re.finditer(r"^[0-9a-f]+ <(" + name_re + r")>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        writer, waited, issued, nmfma = {}, {}, {}, 0      # issued[r]: VMEM ops issued before the load that last wrote v<r>
        pending, nvm = [], 0          # ring loads in flight: (destination registers, VMEM ops issued before it); VMEM ops issued so far
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        lo, hi = (mf[0], mf[-1]) if mf else (0, 0)      # the in-flight check follows the K loop and what lies behind it up to the drain (straight-line there)
        for li, l in enumerate(body):
            t = l.split("//")[0].strip()
            if not t:
                continue
            op = t.split()[0]
            if op.startswith("s_waitcnt") and "vmcnt" in t:
                n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                for r in waited:          # vmcnt(n) leaves at most the n youngest VMEM ops outstanding: a load is covered only if it is older
                    if nvm - issued.get(r, -1) - 1 >= n:
                        waited[r] = True
                pending = [q for q in pending if nvm - q[1] - 1 < n]          # at most n VMEM ops are outstanding: the older ones have landed
            # a load the compiler does not track writes its registers when the data lands, not where the instruction stands: nothing else may
            # write them before a counted wait covers the load (the bug this guards against: a fetch whose result nobody reads looks dead to the
            # register allocator, which hands its destination to another value)
            if op.startswith(("s_cbranch", "s_branch", "s_endpgm")) or li < lo:
                pending = []          # (the walk is textual: across a branch the issue order is not the text order)
            if pending and not op.startswith(("s_", "global_store", "buffer_store", "ds_write", "scratch_store")):
                mm0 = re.match(r"\S+\s+(v\[(\d+):(\d+)\]|v(\d+))", t)
                if mm0 and not (op.startswith("buffer_load") and t.endswith("lds")):
                    dst = set(range(int(mm0.group(2)), int(mm0.group(3)) + 1)) if mm0.group(2) else {int(mm0.group(4))}
                    for regs, _ in pending:
                        if regs & dst:
                            raise RuntimeError(f"isa_check: {name}: `{t}` writes v{sorted(regs & dst)} while an untracked ring load into them is in flight")
            if op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "scratch_load", "scratch_store", "global_atomic")):
                if op.startswith("global_load_dwordx4") and li >= lo:
                    q = re.match(r"\S+\s+v\[(\d+):(\d+)\]", t)
                    if q:
                        pending.append((set(range(int(q.group(1)), int(q.group(2)) + 1)), nvm))
                nvm += 1
            if op.startswith("v_mfma"):
                ops = re.findall(r"([av])\[(\d+):(\d+)\]", t)
                if len(ops) < 4 or ops[1][0] != "v":
                    raise RuntimeError(f"isa_check: {name}: MFMA with a weight operand outside the VGPRs: {t}")
                nmfma += 1
                for r in range(int(ops[1][1]), int(ops[1][2]) + 1):
                    if not writer.get(r, "").startswith("global_load_dwordx4"):
                        raise RuntimeError(f"isa_check: {name}: v{r}, a weight operand of `{t}`, was last written by `{writer.get(r)}` (a copy of a ring register?)")
                    if not waited.get(r, False):
                        raise RuntimeError(f"isa_check: {name}: no vmcnt wait between the ring load of v{r} and `{t}`")
                continue
            if op.startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "s_", "buffer_load_dwordx4")) and "lds" in t + " lds" and not op.startswith("global_load"):
                if not op.startswith(("v_", "ds_read", "global_load", "scratch_load")):
                    continue
            mm = re.match(r"\S+\s+(v\[(\d+):(\d+)\]|v(\d+))", t)
            if mm and not op.startswith(("global_store", "buffer_store", "ds_write", "scratch_store")) and not (op.startswith("buffer_load") and t.endswith("lds")):
                rng = range(int(mm.group(2)), int(mm.group(3)) + 1) if mm.group(2) else [int(mm.group(4))]
                is_vm = op.startswith(("global_load", "buffer_load", "scratch_load"))
                for r in rng:
                    writer[r] = op
                    waited[r] = False
                    issued[r] = (nvm - 1) if is_vm else -1      # (nvm was advanced for this load above)
        if expect_mfma is not None and nmfma != expect_mfma:
            raise RuntimeError(f"isa_check: {name}: {nmfma} MFMAs in the kernel, expected {expect_mfma}")
        seen[name] = nmfma
    if not seen:
        raise RuntimeError(f"isa_check: no kernel matching {name_re} found in {oname} (name mangling changed?)")
    return seen


class ConvDesc(C.Structure):
    """Mirror of cs_conv_desc (include/canonswap_hip.h)."""
    _fields_ = [
        ("in_", C.c_void_p), ("in_sN", C.c_long), ("in_sD", C.c_long), ("in_sH", C.c_long), ("in_sW", C.c_long),
        ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("up_shift", C.c_int),
        ("KD", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
        ("wgt", C.c_void_p), ("Cout_pad", C.c_int), ("Cout", C.c_int),
        ("bias", C.c_void_p), ("bias2", C.c_void_p),
        ("act0", C.c_int), ("slope0", C.c_float),
        ("res", C.c_void_p), ("res_f32", C.c_int), ("res_shift", C.c_int),
        ("res_sN", C.c_long), ("res_sD", C.c_long), ("res_sH", C.c_long), ("res_sW", C.c_long),
        ("pixscale", C.c_void_p), ("ps_stride", C.c_int),
        ("out0", C.c_void_p), ("out0_f32", C.c_int),
        ("out0_sN", C.c_long), ("out0_sD", C.c_long), ("out0_sH", C.c_long), ("out0_sW", C.c_long),
        ("s2", C.c_void_p), ("t2", C.c_void_p), ("act1", C.c_int), ("slope1", C.c_float),
        ("out1", C.c_void_p), ("out1_sN", C.c_long), ("out1_sD", C.c_long), ("out1_sH", C.c_long), ("out1_sW", C.c_long),
        ("stats", C.c_void_p),
        ("mode", C.c_int), ("cfg", C.c_int), ("tile_w", C.c_int), ("tile_h", C.c_int), ("ck", C.c_int), ("xcd_map", C.c_int), ("ragged", C.c_int),
        ("hilo", C.c_int), ("stat_out", C.c_void_p),
        ("xf_kind", C.c_int), ("xf_y", C.c_void_p), ("xf_res", C.c_void_p), ("xf_out", C.c_void_p),
        ("xf_stats", C.c_void_p), ("xf_gamma", C.c_void_p), ("xf_beta", C.c_void_p), ("xf_slope", C.c_float),
        ("ep_general", C.c_int), ("pool_hw", C.c_int),
    ]


MAX_IDENTITY_SLOTS = 8      # CS_MAX_IDENTITY_SLOTS (include/canonswap_hip.h)
_lib = None


def load():
    """Load the shared library; raises RuntimeError if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc); canonswap_amd has no CPU fallback.")
    # The process must hold ONE HIP runtime.  This package's callers pass torch tensors and torch's stream, and torch carries its own
    # libamdhip64 / libhsa-runtime64: imported first, its copies satisfy this library's NEEDED entries too.  Loaded the other way round (this
    # library first - `python __graft_entry__.py smoke`, whose build() checks the ABI before anything imports torch) the system runtime and
    # torch's both initialise and cs_create sees 0 devices.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    if lib.cs_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} speaks C-ABI version {lib.cs_abi_version()}, this package binds version {ABI_VERSION} "
                           "(include/canonswap_hip.h: CS_ABI_VERSION): rebuild with __graft_entry__.build()")
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.cs_last_error.restype = C.c_char_p
    lib.cs_create.argtypes = [ci, ci, C.POINTER(vp)]
    lib.cs_destroy.argtypes = [vp]
    lib.cs_destroy.restype = None
    lib.cs_upload.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.cs_finalize_weights.argtypes = [vp]
    lib.cs_set_identity.argtypes = [vp, ci, vp, vp]
    lib.cs_set_latency_mode.argtypes = [vp, ci]
    lib.cs_extract_feature_3d.argtypes = [vp, ci, vp, vp, vp]
    lib.cs_warp.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    lib.cs_warp_out.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.cs_swap.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.cs_swap_ids.argtypes = [vp, C.POINTER(ci), ci, vp, vp, vp]
    lib.cs_swap_frames_ids.argtypes = [vp, C.POINTER(ci), ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cs_refine.argtypes = [vp, ci, vp, vp, vp]
    lib.cs_warp_forward.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.cs_spade_decode.argtypes = [vp, ci, vp, vp, vp]
    lib.cs_pack_u8.argtypes = [vp, ci, vp, vp, ci, ci, vp]
    lib.cs_unpack_u8.argtypes = [vp, ci, vp, vp, ci, ci, vp]
    lib.cs_motion_extract.argtypes = [vp, ci, vp, vp, vp]
    d6 = C.POINTER(C.c_double)
    lib.cs_soft_erosion.argtypes = [vp, ci, ci, ci, vp, vp, ci, cf, ci, vp, vp, vp]
    lib.cs_prepare_crops.argtypes = [vp, ci, vp, ci, ci, vp, vp]
    lib.cs_warp_affine_u8.argtypes = [vp, vp, ci, ci, d6, vp, ci, ci, vp]
    lib.cs_warp_affine_f32.argtypes = [vp, vp, ci, ci, d6, vp, ci, ci, vp]
    lib.cs_paste_back.argtypes = [vp, vp, vp, vp, ci, ci, d6, vp, vp, ci, ci, vp]
    lib.cs_soft_erosion_frames.argtypes = [vp, ci, ci, ci, vp, ci, vp, ci, cf, ci, vp, vp, vp]
    lib.cs_paste_back_batch.argtypes = [vp, ci, vp, vp, ci, ci, C.POINTER(C.c_double), vp, vp, ci, ci, vp]
    lib.cs_motion_keypoints.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    lib.cs_animate_frames.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, vp, vp]
    lib.cs_swap_frames.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cs_profile_begin.argtypes = [vp]
    lib.cs_profile_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]
    lib.cs_profile_exec_flops.argtypes = [vp, C.POINTER(C.c_double)]
    lib.cs_op_conv.argtypes = [C.POINTER(ConvDesc), vp]
    lib.cs_op_grid_sample3d.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.cs_op_chan_stats.argtypes = [vp, ci, ci, C.c_long, ci, cf, vp, vp, vp]
    lib.cs_op_pair_ragged.argtypes = [vp, ci, ci, ci, ci, ci, vp]
    lib.cs_op_t_mask.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp]
    lib.cs_op_resblock3d.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, cf, vp]
    lib.cs_op_chan_stats_partial_floats.argtypes = [ci, C.c_long, ci]
    lib.cs_op_chan_stats_partial_floats.restype = C.c_long
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().cs_last_error().decode(errors='replace')}")
