// Tuning variants of the 8192-point kernel: pass orders, twiddle sources, round-1 configuration.
#include "fsea_configs_tune.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192r1, "r1", FSEA_CFG_8192_R1)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192nd, "nd", FSEA_CFG_8192_ND)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192A, "A", FSEA_CFG_8192_A)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192B, "B", FSEA_CFG_8192_B)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192D, "D", FSEA_CFG_8192_D)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192notwl, "notwl", FSEA_CFG_8192_NOTWL)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192notwr, "notwr", FSEA_CFG_8192_NOTWR)
FSEA_REGISTER_BEGIN(tune_8192a)
FSEA_REGISTER(fsea_fft8192r1)
FSEA_REGISTER(fsea_fft8192nd)
FSEA_REGISTER(fsea_fft8192A)
FSEA_REGISTER(fsea_fft8192B)
FSEA_REGISTER(fsea_fft8192D)
FSEA_REGISTER(fsea_fft8192notwl)
FSEA_REGISTER(fsea_fft8192notwr)
FSEA_REGISTER_END
