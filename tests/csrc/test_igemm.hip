// Test-only library (libcanonswap_test.so, built by canonswap_amd/_lib.py build_test_lib): the first-generation implicit-GEMM
// conv kernel, kept as an independently written cross-check of conv_halo in the operator tests (tests/test_gpu_ops.py).  It is NOT
// linked into libcanonswap_hip.so and no product path reaches it.
#include "conv_igemm.hip"
#include "../../include/canonswap_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_terr[1024] = "";
void cs_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_terr, sizeof(g_terr), fmt, ap);
    va_end(ap);
}
extern "C" const char* cs_test_last_error(void) { return g_terr; }

const half_t* cs_zero_page()
{
    static void* zp = nullptr;
    if (!zp) {
        if (hipMalloc(&zp, 256) != hipSuccess) return nullptr;
        (void)hipMemset(zp, 0, 256);
    }
    return (const half_t*)zp;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// cs_op_conv's descriptor -> ConvParams for the conv_igemm tile configurations (cfg -1: by Cout_pad, 0..3: CFG_128x128 .. CFG_256x16)
extern "C" int cs_test_conv_igemm(const cs_conv_desc* d, void* stream)
{
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = (const half_t*)d->in; p.zero = cs_zero_page(); p.in_sN = d->in_sN; p.in_sD = d->in_sD; p.in_sH = d->in_sH; p.in_sW = d->in_sW;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.inD = d->D; p.Cin = d->Cin; p.nchunks = (d->Cin + 31) / 32; p.up_shift = d->up_shift;
    p.KD = d->KD; p.KH = d->KH; p.KW = d->KW; p.PD = d->KD / 2; p.PH = d->KH / 2; p.PW = d->KW / 2;
    p.wgt = (const half_t*)d->wgt; p.Cout_pad = d->Cout_pad; p.Cout = d->Cout;
    p.bias = d->bias; p.bias2 = d->bias2; p.act0 = d->act0; p.slope0 = d->slope0;
    p.res = TDesc{(void*)d->res, d->res_sN, d->res_sD, d->res_sH, d->res_sW}; p.res_f32 = d->res_f32; p.res_shift = d->res_shift;
    p.pixscale = d->pixscale; p.ps_stride = d->ps_stride ? d->ps_stride : 1;
    p.out0 = TDesc{d->out0, d->out0_sN, d->out0_sD, d->out0_sH, d->out0_sW}; p.out0_f32 = d->out0_f32;
    p.s2 = d->s2; p.t2 = d->t2; p.act1 = d->act1; p.slope1 = d->slope1;
    p.out1 = TDesc{d->out1, d->out1_sN, d->out1_sD, d->out1_sH, d->out1_sW};
    p.stats = d->stats;
    int cfg = d->cfg;
    if (cfg < 0) cfg = p.Cout_pad % 128 == 0 ? CFG_128x128 : p.Cout_pad % 64 == 0 ? CFG_128x64 : p.Cout_pad % 32 == 0 ? CFG_256x32 : CFG_256x16;
    if (cfg > CFG_256x16) { cs_set_error("cs_test_conv_igemm: cfg %d is not a conv_igemm configuration", cfg); return -1; }
    const int BM = (cfg == CFG_128x128 || cfg == CFG_128x64) ? 128 : 256;
    int tw = d->tile_w ? d->tile_w : 16, th = d->tile_h ? d->tile_h : BM / 16;
    if (tw > p.W) tw = p.W;
    if (th > p.H) th = p.H;
    while (tw * th > BM) th >>= 1;
    int tdd = BM / (tw * th);
    if (tdd > p.D) tdd = p.D;
    const int tn = BM / (tw * th * tdd);
    p.lgTW = ilog2(tw); p.lgTH = ilog2(th); p.lgTD = ilog2(tdd);
    p.nTW = p.W / tw; p.nTH = p.H / th; p.nTD = p.D / tdd; p.nTN = (p.N + tn - 1) / tn;
    return launch_conv(p, cfg, d->mode, (hipStream_t)stream);
}
