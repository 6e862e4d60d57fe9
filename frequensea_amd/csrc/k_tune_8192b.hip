// Tuning variants of the 8192-point kernel: schedule options and the V2 schedule.
#include "fsea_configs_tune.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192x0, "x0", FSEA_CFG_8192_X0)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192v2, "v2", FSEA_CFG_8192_V2)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192v2s, "v2s", FSEA_CFG_8192_V2S)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192stnt, "st_nt", FSEA_CFG_8192_STNT)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192ldnt, "ld_nt", FSEA_CFG_8192_LDNT)
FSEA_DEFINE_KERNEL_LITE(fsea_fft8192cp0, "cp0", FSEA_CFG_8192_CP0)
FSEA_REGISTER_BEGIN(tune_8192b)
FSEA_REGISTER(fsea_fft8192x0)
FSEA_REGISTER(fsea_fft8192v2)
FSEA_REGISTER(fsea_fft8192v2s)
FSEA_REGISTER(fsea_fft8192stnt)
FSEA_REGISTER(fsea_fft8192ldnt)
FSEA_REGISTER(fsea_fft8192cp0)
FSEA_REGISTER_END
