// Dispatch of the LDS-staged convolution kernel (conv_halo_kernel.h) for gfx950.
//
// The kernel template has a few hundred instantiations; to keep the build parallel this one source is compiled once per
// HALO_GROUP (0..HALO_NGROUPS-1, see canonswap_amd/_lib.py), each translation unit holding a slice of the (cfg, mode) -> template
// table as launch_conv_halo_g<k>().  Without -DHALO_GROUP everything lands in one translation unit.
#include "conv_halo_kernel.h"

#ifndef HALO_GROUP
#define HALO_GROUP (-1)
#endif
#define IN_GROUP(g) (HALO_GROUP == -1 || HALO_GROUP == (g))
#define NOT_MINE (-2)

// picks the static-shape instantiation STV when the launch matches it, else the dynamic kernel
template <int CK, int WPX, int WCH, int WVP, int WVC, int MODE, bool SK, int STV>
static int launch_halo_cfg(const ConvParams& p, hipStream_t st)
{
    if constexpr (STV != 0) {
        using SS = StaticShape<STV>;
        if (p.KD == SS::KD && p.KH == SS::KH && p.KW == SS::KW && p.lgTW == SS::LW && p.lgTH == SS::LH && p.lgTD == SS::LD &&
            p.nchunks % (CK / 32) == 0)
            return launch_halo_st<CK, WPX, WCH, WVP, WVC, MODE, SK, STV>(p, st);
    }
    // The 128x256 statistics tile has no dynamic-shape variant: with the 32 statistics accumulators on top of 128 accumulator registers
    // hipcc spills the hand-counted weight ring of the dynamic K loop (the loads it does not track: _lib.isa_check), and no layer needs it
    // (every conv that emits statistics is a 3x3 or 3x3x3 on one of the static tiles).
    if constexpr (WCH == 4 && MODE == MODE_STDSTAT) {
        cs_set_error("conv_halo: the 128x256 statistics tile exists for the static shapes only (%dx%dx%d taps, tile 2^%d x 2^%d x 2^%d)",
                     p.KD, p.KH, p.KW, p.lgTW, p.lgTH, p.lgTD);
        return -1;
    } else {
        return launch_halo_st<CK, WPX, WCH, WVP, WVC, MODE, SK, 0>(p, st);
    }
}

#define HALO_CASE(CFG, WPX, WCH, WVP, WVC, MODE, SK, ST2D, ST3D)                                         \
    if (cfg == CFG && mode == MODE) {                                                                    \
        if (ck == 64) return launch_halo_cfg<64, WPX, WCH, WVP, WVC, MODE, SK, ST2D>(p, st);            \
        if (stv == 1) return launch_halo_cfg<32, WPX, WCH, WVP, WVC, MODE, SK, ST2D>(p, st);            \
        if (stv == 5 && ST3D == 2) return launch_halo_cfg<32, WPX, WCH, WVP, WVC, MODE, SK, (ST3D == 2 ? 5 : 0)>(p, st); \
        return launch_halo_cfg<32, WPX, WCH, WVP, WVC, MODE, SK, ST3D>(p, st);                           \
    }

#define GROUP_FN(g) int launch_conv_halo_g##g(const ConvParams& p, int cfg, int ck, int mode, int stv, hipStream_t st)
GROUP_FN(0); GROUP_FN(1); GROUP_FN(2); GROUP_FN(3); GROUP_FN(4); GROUP_FN(5); GROUP_FN(6); GROUP_FN(7);

#if IN_GROUP(0)
GROUP_FN(0)     // special shapes that take precedence over the generic table: per-phase up-sampling convs, 1x1 convs
{
    (void)stv;
    if (p.KD == 3 && p.KH == 2 && p.KW == 2 && ck == 32 && mode == MODE_STD) {
        if (cfg == CFG_H_256x32) return launch_halo_cfg<32, 4, 2, 4, 1, MODE_STD, false, 12>(p, st);
        if (cfg == CFG_H_128x64) return launch_halo_cfg<32, 4, 2, 2, 2, MODE_STD, false, 13>(p, st);
        if (cfg == CFG_H_128x128 && p.lgTW == 2) return launch_halo_cfg<32, 8, 2, 1, 4, MODE_STD, false, 17>(p, st);      // up-block 0 (512 channels, 4x4 source grid)
        if (cfg == CFG_H_128x128) return launch_halo_cfg<32, 8, 2, 1, 4, MODE_STD, false, 13>(p, st);      // up-blocks 1 / 2 (256 / 128 channels)
    }
    if (mode == MODE_STD && p.KD == 1 && p.KH == 1 && p.KW == 1) {      // 1x1 convs (shortcuts, the motion extractor's linear layers)
        if (cfg == CFG_H_128x128 && ck == 32) return launch_halo_cfg<32, 8, 2, 1, 4, MODE_STD, false, 15>(p, st);
        if (cfg == CFG_H_128x64 && ck == 64) return launch_halo_cfg<64, 4, 2, 2, 2, MODE_STD, false, 15>(p, st);
        if (cfg == CFG_H_128x64 && ck == 32) return launch_halo_cfg<32, 4, 2, 2, 2, MODE_STD, false, 15>(p, st);
        if (cfg == CFG_H_128x32 && ck == 64) return launch_halo_cfg<64, 2, 2, 4, 1, MODE_STD, false, 15>(p, st);
        if (cfg == CFG_H_128x32 && ck == 32) return launch_halo_cfg<32, 2, 2, 4, 1, MODE_STD, false, 15>(p, st);
    }
    return NOT_MINE;
}
#endif

#if IN_GROUP(1)
GROUP_FN(1)     // mlp_shared phase convs (1x2x2 family, ahead of the generic 128x128 STD entry); the 160-wide tiles
{
    if (cfg == CFG_H_128x128 && mode == MODE_STD && ck == 64 && p.KD == 1 && p.KH <= 2 && p.KW <= 2) {
        if (p.KH == 2 && p.KW == 2) return launch_halo_cfg<64, 8, 2, 1, 4, MODE_STD, false, 10>(p, st);
        if (p.KH == 2) return launch_halo_cfg<64, 8, 2, 1, 4, MODE_STD, false, 11>(p, st);
        if (p.KW == 2) return launch_halo_cfg<64, 8, 2, 1, 4, MODE_STD, false, 14>(p, st);
        return launch_halo_cfg<64, 8, 2, 1, 4, MODE_STD, false, 15>(p, st);
    }
    if (cfg == CFG_H_256x64 && mode == MODE_STD && ck == 32) return launch_halo_cfg<32, 8, 2, 2, 2, MODE_STD, false, 7>(p, st);
    if (cfg == CFG_H_256x64 && ck == 64) {        // 2-D 3x3 on 16x16 tiles, static shape only
        if (p.KD != 1 || p.KH != 3 || p.KW != 3 || p.lgTW != 4 || p.lgTH != 4 || p.lgTD != 0) {
            cs_set_error("conv_halo: the 2-D 256x64 tile runs 3x3 convs on 16x16 tiles only");
            return -1;
        }
        if (mode == MODE_STD) return launch_halo_st<64, 8, 2, 2, 2, MODE_STD, false, 16>(p, st);
        if (mode == MODE_STDSTAT) return launch_halo_st<64, 8, 2, 2, 2, MODE_STDSTAT, false, 16>(p, st);
    }
    if (cfg == CFG_H_256x160 && mode == MODE_STD && ck == 32) {
        if (p.KD == 7 && p.lgTW == 2) return launch_halo_cfg<32, 8, 5, 2, 2, MODE_STD, false, 9>(p, st);
        if (p.KD == 7) return launch_halo_cfg<32, 8, 5, 2, 2, MODE_STD, false, 8>(p, st);
        return launch_halo_cfg<32, 8, 5, 2, 2, MODE_STD, false, 7>(p, st);
    }
    if (cfg == CFG_H_128x160 && mode == MODE_STD && ck == 32) {
        if (stv == 4) return launch_halo_cfg<32, 4, 5, 2, 2, MODE_STD, false, 4>(p, st);
        if (stv == 6) return launch_halo_cfg<32, 4, 5, 2, 2, MODE_STD, false, 6>(p, st);
        return launch_halo_cfg<32, 4, 5, 2, 2, MODE_STD, false, 2>(p, st);
    }
    return NOT_MINE;
}
#endif

#if IN_GROUP(2)
GROUP_FN(2)
{
    HALO_CASE(CFG_H_128x256, 8, 4, 1, 4, MODE_STD, false, 1, 0)
    HALO_CASE(CFG_H_128x256, 8, 4, 1, 4, MODE_TBLEND, false, 1, 0)
    return NOT_MINE;
}
#endif

#if IN_GROUP(3)
GROUP_FN(3)
{
    HALO_CASE(CFG_H_128x256, 8, 4, 1, 4, MODE_SPADE, false, 1, 0)
    HALO_CASE(CFG_H_128x256, 8, 4, 1, 4, MODE_STDSTAT, false, 1, 0)
    return NOT_MINE;
}
#endif

#if IN_GROUP(4)
GROUP_FN(4)
{
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_STD, false, 1, 2)
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_STDSTAT, false, 1, 0)
    return NOT_MINE;
}
#endif

#if IN_GROUP(5)
GROUP_FN(5)
{
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_TBLEND, false, 1, 0)
    HALO_CASE(CFG_H_128x128, 8, 2, 1, 4, MODE_SPADE, false, 1, 0)
    HALO_CASE(CFG_H_128x64, 4, 2, 2, 2, MODE_STDSTAT, false, 1, 0)
    return NOT_MINE;
}
#endif

#if IN_GROUP(6)
GROUP_FN(6)
{
    HALO_CASE(CFG_H_128x64, 4, 2, 2, 2, MODE_STD, false, 1, 2)
    HALO_CASE(CFG_H_256x32, 4, 2, 4, 1, MODE_STD, false, 0, 3)
    HALO_CASE(CFG_H_256x32, 4, 2, 4, 1, MODE_STDSTAT, false, 0, 3)
    return NOT_MINE;
}
#endif

#if IN_GROUP(7)
GROUP_FN(7)
{
    HALO_CASE(CFG_H_128x32, 2, 2, 4, 1, MODE_STD, false, 0, 2)
    HALO_CASE(CFG_H_128x16, 2, 1, 4, 1, MODE_STD, false, 1, 0)
    HALO_CASE(CFG_H_256x16, 4, 1, 4, 1, MODE_PIXSHUF, false, 0, 0)
    HALO_CASE(CFG_H_SK128x32, 8, 2, 4, 1, MODE_STD, true, 0, 0)
    return NOT_MINE;
}
#endif

#if IN_GROUP(0) && defined(CS_TIMELINE)
unsigned long long* g_cs_tl = nullptr;
long g_cs_tl_cap = 0;
// instrumented builds only: device buffer of cap x 8 u64 that every following conv_halo launch stamps (nullptr: off)
extern "C" void cs_debug_set_timeline(void* buf, long cap) { g_cs_tl = (unsigned long long*)buf; g_cs_tl_cap = cap; }
#endif

#if IN_GROUP(0)
// cfg: CFG_H_* (common.h); ck: 32 or 64
int launch_conv_halo(const ConvParams& p, int cfg, int ck, int mode, hipStream_t st)
{
    if (mode == MODE_STD && p.stat_out) mode = MODE_STDSTAT;
    // candidate static shape for this launch
    const int stv = p.KD == 1 ? 1 : (p.KD == 7 ? (p.lgTW == 1 ? 6 : 4) : (p.lgTD == 4 ? 3 : (p.lgTD == 3 ? 5 : 2)));
    int r;
#define TRY_GROUP(g) if ((r = launch_conv_halo_g##g(p, cfg, ck, mode, stv, st)) != NOT_MINE) return r;
    TRY_GROUP(0) TRY_GROUP(1) TRY_GROUP(2) TRY_GROUP(3) TRY_GROUP(4) TRY_GROUP(5) TRY_GROUP(6) TRY_GROUP(7)
#undef TRY_GROUP
    cs_set_error("conv_halo: unsupported cfg/mode %d/%d", cfg, mode);
    return -1;
}
#endif
