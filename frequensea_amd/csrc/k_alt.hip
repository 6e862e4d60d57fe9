// Per-(size, mode) product configurations (fsea_configs.h): 256 points for f32 rows, 512 points for u8 pixels, 1024 points
// for COMPLEX_F32 rows.  Full kernel sets, so that a plan that takes one of them serves every entry point with it.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft256rows, "rows", FSEA_CFG_256_ROWS)
FSEA_DEFINE_KERNEL(fsea_fft512px, "px", FSEA_CFG_512_PX)
FSEA_DEFINE_KERNEL(fsea_fft1024rt, "rt", FSEA_CFG_1024_RT)
FSEA_DEFINE_WINDOWED(fsea_fft256rows, FSEA_WIN)
FSEA_DEFINE_WINDOWED(fsea_fft512px, FSEA_WIN)
FSEA_DEFINE_WINDOWED(fsea_fft1024rt, FSEA_WIN)
FSEA_REGISTER_BEGIN(alt)
FSEA_REGISTER_WIN(fsea_fft256rows)
FSEA_REGISTER_WIN(fsea_fft512px)
FSEA_REGISTER_WIN(fsea_fft1024rt)
FSEA_REGISTER_END
