// N = 8192: the product configuration (fsea_configs.h).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft8192, "", FSEA_CFG_8192)
FSEA_DEFINE_HALF_OVERLAP(fsea_fft8192)
FSEA_DEFINE_WINDOWED(fsea_fft8192, FSEA_WIN)
FSEA_DEFINE_HALF_OVERLAP_WIN(fsea_fft8192, FSEA_WIN)
FSEA_REGISTER_BEGIN(8192)
FSEA_REGISTER_HALF_WIN(fsea_fft8192)
FSEA_REGISTER_END
