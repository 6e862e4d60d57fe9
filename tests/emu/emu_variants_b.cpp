// TEST-ONLY: tuning variants of the kernel on the CPU shim, part b (see emu_common.h).
#include "emu_common.h"

int emu_variants_b(int n, const std::string &v, int in_kind, int mt, const fsea::FftArgs &a, unsigned grid) {
        EMU_VARIANT(8192, "nd", FSEA_CFG_8192_ND)
        EMU_VARIANT(8192, "v2", FSEA_CFG_8192_V2)
        EMU_VARIANT(8192, "v2s", FSEA_CFG_8192_V2S)
        EMU_VARIANT(8192, "x0", FSEA_CFG_8192_X0)
        EMU_VARIANT(8192, "A", FSEA_CFG_8192_A)
        EMU_VARIANT(8192, "B", FSEA_CFG_8192_B)
        EMU_VARIANT(8192, "D", FSEA_CFG_8192_D)
        EMU_VARIANT(8192, "notwl", FSEA_CFG_8192_NOTWL)
        EMU_VARIANT(8192, "notwr", FSEA_CFG_8192_NOTWR)
        EMU_VARIANT(4096, "x0", FSEA_CFG_4096_X0)
        EMU_VARIANT(4096, "df", FSEA_CFG_4096_DF)
        EMU_VARIANT(4096, "B", FSEA_CFG_4096_B)
        EMU_VARIANT(256, "p16", FSEA_CFG_256_P16)
        EMU_VARIANT(128, "p16", FSEA_CFG_128_P16)
    return -2;
}
