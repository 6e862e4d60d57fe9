// TEST-ONLY: tuning variants of the kernel on the CPU shim, part c (see emu_common.h).
#include "emu_common.h"

int emu_variants_c(int n, const std::string &v, int in_kind, int mt, const fsea::FftArgs &a, unsigned grid) {
        EMU_VARIANT(4096, "f1", FSEA_CFG_4096_F1)
        EMU_VARIANT(4096, "t256", FSEA_CFG_4096_T256)
        EMU_VARIANT(4096, "B3", FSEA_CFG_4096_B3)
        EMU_VARIANT(4096, "C", FSEA_CFG_4096_C)
        EMU_VARIANT(4096, "D", FSEA_CFG_4096_D)
        EMU_VARIANT(2048, "x0", FSEA_CFG_2048_X0)
        EMU_VARIANT(2048, "df", FSEA_CFG_2048_DF)
        EMU_VARIANT(2048, "B", FSEA_CFG_2048_B)
        EMU_VARIANT(2048, "C", FSEA_CFG_2048_C)
        EMU_VARIANT(1024, "x0", FSEA_CFG_1024_X0)
        EMU_VARIANT(1024, "B", FSEA_CFG_1024_B)
        EMU_VARIANT(1024, "C", FSEA_CFG_1024_C)
        EMU_VARIANT(1024, "D", FSEA_CFG_1024_D)
        EMU_VARIANT(16384, "nd", FSEA_CFG_16384_ND)
        EMU_VARIANT(16384, "B", FSEA_CFG_16384_B)
        EMU_VARIANT(512, "f8", FSEA_CFG_512_F8)
        EMU_VARIANT(256, "f8", FSEA_CFG_256_F8)
    return -2;
}
