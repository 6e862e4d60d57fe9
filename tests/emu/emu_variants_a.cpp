// TEST-ONLY: tuning variants of the kernel on the CPU shim, part a (see emu_common.h).
#include "emu_common.h"

int emu_variants_a(int n, const std::string &v, int in_kind, int mt, const fsea::FftArgs &a, unsigned grid) {
        EMU_VARIANT(4096, "w64", FSEA_CFG_4096_W64)
        EMU_VARIANT(4096, "s2", FSEA_CFG_4096_S2)
        EMU_VARIANT(4096, "w64b", FSEA_CFG_4096_W64B)
        EMU_VARIANT(4096, "pk", FSEA_CFG_4096_PK)
        EMU_VARIANT(4096, "px0", FSEA_CFG_4096_PX0)
        EMU_VARIANT(8192, "pk", FSEA_CFG_8192_PK)
        EMU_VARIANT(8192, "px0", FSEA_CFG_8192_PX0)
        EMU_VARIANT(256, "pk", FSEA_CFG_256_PK)
        EMU_VARIANT(256, "px0", FSEA_CFG_256_PX0)
        EMU_VARIANT(1024, "px0", FSEA_CFG_1024_PX0)
        EMU_VARIANT(256, "p64", FSEA_CFG_256_P64)
        EMU_VARIANT(8192, "B2", FSEA_CFG_8192_B2)
        EMU_VARIANT(8192, "D2", FSEA_CFG_8192_D2)
        EMU_VARIANT(8192, "W", FSEA_CFG_8192_W)
        EMU_VARIANT(4096, "nr", FSEA_CFG_4096_LR)
        EMU_VARIANT(2048, "nr", FSEA_CFG_2048_LR)
    EMU_VARIANT(1024, "r2", FSEA_CFG_1024_R2)
    EMU_VARIANT(1024, "e", FSEA_CFG_1024_E)
    EMU_VARIANT(1024, "h", FSEA_CFG_1024_H)
    return -2;
}
