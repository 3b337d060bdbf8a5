"""Per-wave phase timeline of conv_halo launches (GPU box; needs the instrumented build ab/timeline.so, made on the build host by
`python tools/build_variant.py timeline -DCS_TIMELINE`).

Every wave stamps s_memtime at: 0 kernel entry, 1 prologue done (offsets computed), 2 first halo chunk landed (after the barrier),
3 main loop done, 4 epilogue issued, 5 stores acknowledged; plus HW_ID / XCC_ID.  For the representative layer shapes of the
engine (B = 32 unless --batch) this prints where a workgroup's lifetime goes and how many workgroups shared a CU.

    CANONSWAP_LIB=ab/timeline.so python tools/timeline.py [--batch 32] [--only name]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as ops  # noqa: E402
from canonswap_amd import _lib  # noqa: E402

DEV = "cuda:0"


def vol(N, C=32, dtype=torch.float16, rnd=None):
    """feature volume [N][H][W][D][C] seen as [N, D, H, W, C]"""
    t = torch.empty(N, 64, 64, 16, C, dtype=dtype, device=DEV)
    if rnd is not None:
        t.copy_(torch.from_numpy(rnd.standard_normal(t.shape).astype(np.float32)).to(DEV))
    return t.permute(0, 3, 1, 2, 4)


def cases(B, r):
    def rn(*shape, scale=1.0, dtype=np.float16):
        return torch.from_numpy((r.standard_normal(shape) * scale).astype(dtype)).to(DEV)

    def wgt(cin, cout_pad, k):
        return rn(((cin + 31) // 32) * int(np.prod(k)), cout_pad, 32, scale=0.02)

    out = []
    # 32->32 volume convs (ResBlock3d conv1 / conv2 with fp32 residual + second output)
    x = torch.relu(vol(B, rnd=r))
    out.append(("v32.c1", dict(x=x, w=wgt(32, 32, (3, 3, 3)), cout_pad=32, cout=32, k=(3, 3, 3), act0="relu", out0=vol(B), cfg=12, tile=(4, 4)),
                27, 4 * 2, 3))
    out.append(("v32.c2", dict(x=x, w=wgt(32, 32, (3, 3, 3)), cout_pad=32, cout=32, k=(3, 3, 3), res=vol(B, dtype=torch.float32, rnd=r),
                               out0=vol(B, dtype=torch.float32), out1=vol(B), s2=rn(32, dtype=np.float32), t2=rn(32, dtype=np.float32),
                               act1="relu", cfg=12, tile=(4, 4)), 27, 4 * 2, 3))
    # T fused [W; w_mod] conv with the blend epilogue
    x2 = torch.relu(rn(B, 1, 64, 64, 512))        # the engine's conv inputs are post-activation (about half zeros: matters for DVFS)
    out.append(("T.blend", dict(x=x2, w=wgt(512, 1024, (1, 3, 3)), cout_pad=1024, cout=512, k=(1, 3, 3), mode=1, bias=rn(512, dtype=np.float32),
                                pixscale=torch.rand(B * 4096 * 4, device=DEV), ps_stride=4, res=rn(B, 1, 64, 64, 512, dtype=np.float32),
                                out0=torch.empty(B, 1, 64, 64, 512, dtype=torch.float32, device=DEV),
                                out1=torch.empty(B, 1, 64, 64, 512, dtype=torch.float16, device=DEV), cfg=17), 144, 8 * 4, 2))
    out.append(("T.mask", dict(x=x2, w=wgt(512, 16, (1, 3, 3)), cout_pad=16, cout=4, k=(1, 3, 3), bias=rn(4, dtype=np.float32), act0="sigmoid",
                               out0=torch.empty(B, 1, 64, 64, 4, dtype=torch.float32, device=DEV), cfg=14), 144, 2 * 1, 3))
    # G 3x3 512->512 with residual
    out.append(("G.c512", dict(x=x2, w=wgt(512, 512, (1, 3, 3)), cout_pad=512, cout=512, k=(1, 3, 3), bias=rn(512, dtype=np.float32),
                               res=rn(B, 1, 64, 64, 512), out0=torch.empty(B, 1, 64, 64, 512, dtype=torch.float16, device=DEV), cfg=10), 144, 8 * 2, 3))
    # SPADE gamma/beta conv 128 -> 2x512
    xa = torch.relu(rn(B, 1, 64, 64, 128))
    st = torch.stack([rn(B, 512, dtype=np.float32), torch.rand(B, 512, device=DEV) + 0.5], dim=2).contiguous()
    out.append(("G.gb512", dict(x=xa, w=wgt(128, 1024, (1, 3, 3)), cout_pad=1024, cout=512, k=(1, 3, 3), mode=2, bias=rn(512, dtype=np.float32),
                                bias2=rn(512, dtype=np.float32), res=rn(B, 1, 64, 64, 512), stats=st, act0="lrelu", slope0=0.2,
                                out0=torch.empty(B, 1, 64, 64, 512, dtype=torch.float16, device=DEV), cfg=10), 36, 8 * 2, 3))
    out.append(("G.gb512w", dict(x=xa, w=wgt(128, 1024, (1, 3, 3)), cout_pad=1024, cout=512, k=(1, 3, 3), mode=2, bias=rn(512, dtype=np.float32),
                                 bias2=rn(512, dtype=np.float32), res=rn(B, 1, 64, 64, 512), stats=st, act0="lrelu", slope0=0.2,
                                 out0=torch.empty(B, 1, 64, 64, 512, dtype=torch.float16, device=DEV), cfg=17), 36, 8 * 4, 2))
    # dense-motion hourglass: tail, mask, first encoder block
    Bd = min(B, 16)
    xd = torch.relu(rn(Bd, 16, 64, 64, 144))
    out.append(("W.tail", dict(x=xd, w=wgt(144, 160, (3, 3, 3)), cout_pad=160, cout=144, k=(3, 3, 3), act0="relu",
                               out0=torch.empty(Bd, 16, 64, 64, 144, dtype=torch.float16, device=DEV), cfg=18, tile=(8, 8)), 135, 4 * 5, 2))
    out.append(("W.mask", dict(x=xd, w=wgt(144, 160, (7, 7, 1)), cout_pad=160, cout=160, k=(7, 7, 1),
                               out0=torch.empty(Bd, 16, 64, 64, 160, dtype=torch.float32, device=DEV), cfg=18, tile=(2, 8)), 245, 4 * 5, 2))
    out.append(("W.tail256", dict(x=xd, w=wgt(144, 160, (3, 3, 3)), cout_pad=160, cout=144, k=(3, 3, 3), act0="relu",
                                  out0=torch.empty(Bd, 16, 64, 64, 144, dtype=torch.float16, device=DEV), cfg=19, tile=(8, 8)), 135, 8 * 5, 1))
    out.append(("W.mask256", dict(x=xd, w=wgt(144, 160, (7, 7, 1)), cout_pad=160, cout=160, k=(7, 7, 1),
                                  out0=torch.empty(Bd, 16, 64, 64, 160, dtype=torch.float32, device=DEV), cfg=19, tile=(2, 8)), 245, 8 * 5, 1))
    out.append(("W.enc0", dict(x=xd[..., :112], w=wgt(112, 64, (3, 3, 3)), cout_pad=64, cout=64, k=(3, 3, 3), act0="relu", cin=112,
                               out0=torch.empty(Bd, 16, 64, 64, 64, dtype=torch.float16, device=DEV), cfg=11, tile=(8, 8)), 108, 4 * 2, 2))
    out.append(("W.enc0_256", dict(x=xd[..., :112], w=wgt(112, 64, (3, 3, 3)), cout_pad=64, cout=64, k=(3, 3, 3), act0="relu", cin=112,
                                   out0=torch.empty(Bd, 16, 64, 64, 64, dtype=torch.float16, device=DEV), cfg=20, tile=(8, 8)), 108, 8 * 2, 2))
    # G's last up block at 256 x 256 (64 output channels: c0 256 -> 64 with statistics, n1 SPADE 128 -> 2 x 64, c1 64 -> 64 + shortcut + lrelu copy) and
    # the SPADE convs with 256 modulated channels at 128 / 256 (up_0.n1, up_1.n0); cfg / chunk size as pick_halo_cfg / go() choose them
    Bu = min(B, 16)

    def d0(t):      # the size-1 depth axis at stride 0, as the engine's nhwc() descriptors have it (axis strides are 24-bit quantities)
        return t.as_strided(t.shape, (t.stride(0), 0, t.stride(2), t.stride(3), 1), t.storage_offset())
    xu = d0(torch.relu(rn(Bu, 1, 256, 256, 256)))
    sto = torch.empty(Bu * 2048 * 64 * 2 * 4, dtype=torch.float32, device=DEV)
    out.append(("up1.c0", dict(x=xu, w=wgt(256, 64, (1, 3, 3)), cout_pad=64, cout=64, k=(1, 3, 3), bias=rn(64, dtype=np.float32), stat_out=sto,
                               out0=d0(torch.empty(Bu, 1, 256, 256, 64, dtype=torch.float16, device=DEV)), cfg=20, tile=(16, 16)), 72, 8 * 2, 2))
    a384 = d0(torch.relu(rn(Bu, 1, 256, 256, 384)))
    st64 = torch.stack([rn(Bu, 64, dtype=np.float32), torch.rand(Bu, 64, device=DEV) + 0.5], dim=2).contiguous()
    x64 = d0(rn(Bu, 1, 256, 256, 64))
    out.append(("up1.n1", dict(x=a384[..., 128:256], w=wgt(128, 128, (1, 3, 3)), cout_pad=128, cout=64, k=(1, 3, 3), mode=2, bias=rn(64, dtype=np.float32),
                               bias2=rn(64, dtype=np.float32), res=x64, stats=st64, act0="lrelu", slope0=0.2, ck=32,
                               out0=d0(torch.empty(Bu, 1, 256, 256, 64, dtype=torch.float16, device=DEV)), cfg=10), 36, 8 * 2, 3))
    for nm, cf, tl, mps, sl in (("up1.c1", 20, (16, 16), 8 * 2, 2), ("up1.c1s", 11, (0, 0), 4 * 2, 3)):
        out.append((nm, dict(x=d0(torch.relu(x64)), w=wgt(64, 64, (1, 3, 3)), cout_pad=64, cout=64, k=(1, 3, 3), bias=rn(64, dtype=np.float32), res=x64,
                             out1=d0(torch.empty(Bu, 1, 256, 256, 64, dtype=torch.float16, device=DEV)), act1="lrelu", slope1=0.2, cfg=cf, tile=tl), 18, mps, sl))
    st256 = torch.stack([rn(Bu, 256, dtype=np.float32), torch.rand(Bu, 256, device=DEV) + 0.5], dim=2).contiguous()
    out.append(("up1.n0", dict(x=a384[..., 0:128], w=wgt(128, 512, (1, 3, 3)), cout_pad=512, cout=256, k=(1, 3, 3), mode=2, bias=rn(256, dtype=np.float32),
                               bias2=rn(256, dtype=np.float32), res=d0(rn(Bu, 1, 128, 128, 256)), res_shift=1, stats=st256, act0="lrelu", slope0=0.2,
                               out0=d0(torch.empty(Bu, 1, 256, 256, 256, dtype=torch.float16, device=DEV)), cfg=17), 36, 8 * 4, 2))
    return out


def run_case(lib, name, kw, nsteps, mfma_per_step, slots, nrep=3):
    kw = dict(kw)
    x, w = kw.pop("x"), kw.pop("w")
    cout_pad, cout, k = kw.pop("cout_pad"), kw.pop("cout"), kw.pop("k")
    cap = 1 << 17
    buf = torch.zeros(cap, 12, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.conv(x, w, cout_pad, cout, k, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep):
        ops.conv(x, w, cout_pad, cout, k, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / nrep
    lib.cs_debug_set_timeline(C.c_void_p(buf.data_ptr()), C.c_long(cap))
    ops.conv(x, w, cout_pad, cout, k, **kw)
    torch.cuda.synchronize()
    lib.cs_debug_set_timeline(C.c_void_p(0), C.c_long(0))
    t = buf.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] != 0]
    T = t[:, :8].astype(np.float64)
    life = T[:, 5] - T[:, 0]
    life_ns = (t[:, 9] - t[:, 8]).astype(np.float64) * 10.0          # s_memrealtime: 100 MHz
    ok = life_ns > 0
    ghz = float(np.median(life[ok] / life_ns[ok]))                    # shader clock seen by the waves
    # phases: prologue | first halo wait | main loop | epilogue: consts+fetch issue | first block (fetch wait, store) |
    #         remaining blocks | store drain
    ph = np.stack([T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 6] - T[:, 3], T[:, 7] - T[:, 6], T[:, 4] - T[:, 7],
                   T[:, 5] - T[:, 4]], axis=1)
    if os.environ.get("TL_SLOTS"):
        print("   wave slots (HW_ID[3:0]) histogram:", np.bincount((t[:, 10] & 0xF).astype(np.int64), minlength=16).tolist(),
              " simd (HW_ID[5:4]):", np.bincount(((t[:, 10] >> 4) & 3).astype(np.int64), minlength=4).tolist())
    ncu = len(np.unique((t[:, 11] << 16) | ((t[:, 10] >> 8) & 0xFF)))
    kernel_ns = ms * 1e6
    # one workgroup per CU: time between the end of a workgroup and the start of the next one on the same CU (s_memrealtime, 10 ns ticks)
    gap_ns = -1.0
    if slots == 1:
        cu = (t[:, 11] << 16) | ((t[:, 10] >> 8) & 0xFFFF)
        gaps = []
        for c in np.unique(cu):
            m = t[cu == c]
            st = np.sort(np.unique(m[:, 8])); en = np.sort(np.unique(m[:, 9]))
            # workgroup = 4 waves: take per workgroup the earliest start / latest end by clustering on start order
            o = np.argsort(m[:, 8]); m = m[o]
            wg_s = m[0::4, 8]; wg_e = np.maximum.reduceat(m[:, 9], np.arange(0, len(m), 4))
            if len(wg_s) > 1:
                gaps.extend(((wg_s[1:] - wg_e[:-1]) * 10.0).tolist())
        gap_ns = float(np.median(gaps)) if gaps else -1.0
    slot_occ = life_ns.sum() / (ncu * slots * 4 * kernel_ns)          # share of the kernel a wave slot is occupied
    ideal_loop = nsteps * mfma_per_step * 16.0                        # cycles of back-to-back MFMA issue for one wave alone
    rec = {
        "name": name, "ms": round(ms, 4), "waves": int(len(t)), "cus_seen": int(ncu), "shader_ghz": round(ghz, 3),
        "slot_occupancy": round(float(slot_occ), 3), "slots_per_cu": slots,
        "phase_cycles_mean": [round(float(v), 0) for v in ph.mean(axis=0)],
        "phase_cycles_p50": [round(float(v), 0) for v in np.percentile(ph, 50, axis=0)],
        "phase_share": [round(float(v), 3) for v in ph.mean(axis=0) / life.mean()],
        "life_us_mean": round(float(life_ns.mean()) / 1e3, 2),
        "loop_over_ideal": round(float(ph[:, 2].mean()) / ideal_loop, 2),
        "wg_gap_ns_p50": round(gap_ns, 0),
    }
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "timeline.json"))
    a = ap.parse_args()
    lib = _lib.load()
    if not hasattr(lib, "cs_debug_set_timeline"):
        raise SystemExit("this library is not an instrumented build (CANONSWAP_LIB=ab/timeline.so)")
    lib.cs_debug_set_timeline.argtypes = [C.c_void_p, C.c_long]
    lib.cs_debug_set_timeline.restype = None
    r = np.random.Generator(np.random.PCG64(3))
    recs = []
    print("phase cycles: prologue | first halo wait | main loop | epi consts+fetch issue | epi first block | epi other blocks | store drain")
    for name, kw, nsteps, mps, slots in cases(a.batch, r):
        if a.only and a.only not in name:
            continue
        rec = run_case(lib, name, kw, nsteps, mps, slots)
        recs.append(rec)
        print("%-8s %7.3f ms  %.2f GHz  slot occ %.2f (%d WG/CU)  life %7.2f us  cycles %s  share %s  loop/ideal %.2f  wg gap %.0f ns" % (
            name, rec["ms"], rec["shader_ghz"], rec["slot_occupancy"], slots, rec["life_us_mean"],
            [int(v) for v in rec["phase_cycles_mean"]], rec["phase_share"], rec["loop_over_ideal"], rec["wg_gap_ns_p50"]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(recs, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
