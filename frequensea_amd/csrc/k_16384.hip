// N = 16384: the product configuration (fsea_configs.h).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft16384, "", FSEA_CFG_16384)
FSEA_DEFINE_HALF_OVERLAP(fsea_fft16384)
FSEA_DEFINE_WINDOWED(fsea_fft16384, FSEA_WIN)
FSEA_DEFINE_HALF_OVERLAP_WIN(fsea_fft16384, FSEA_WIN)
FSEA_REGISTER_BEGIN(16384)
FSEA_REGISTER_HALF_WIN(fsea_fft16384)
FSEA_REGISTER_END
