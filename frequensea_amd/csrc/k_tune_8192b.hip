// Tuning variants of the 8192-point kernel: schedule options and the V2 schedule.
#include "fsea_configs_tune.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192x0, "x0", FSEA_CFG_8192_X0)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192x7, "x7", FSEA_CFG_8192_X7)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192tk, "tk", FSEA_CFG_8192_TK)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192pr, "pr", FSEA_CFG_8192_PR)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192v2, "v2", FSEA_CFG_8192_V2)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192v2s, "v2s", FSEA_CFG_8192_V2S)
FSEA_REGISTER_BEGIN(tune_8192b)
FSEA_REGISTER(fsea_fft8192x0)
FSEA_REGISTER(fsea_fft8192x7)
FSEA_REGISTER(fsea_fft8192tk)
FSEA_REGISTER(fsea_fft8192pr)
FSEA_REGISTER(fsea_fft8192v2)
FSEA_REGISTER(fsea_fft8192v2s)
FSEA_REGISTER_END
