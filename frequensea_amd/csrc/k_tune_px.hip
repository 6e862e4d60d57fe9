// Measurement-only ablations of the pixel (DB5 / DB10) kernels at 4096 and 8192 points (wrong results by design).
#include "fsea_configs_tune.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_abl4096_px_nolog, "abl_px_nolog", FSEA_CFG_4096_PXNOLOG)
FSEA_DEFINE_KERNEL(fsea_abl4096_px_nost, "abl_px_nost", FSEA_CFG_4096_PXNOST)
FSEA_DEFINE_KERNEL(fsea_abl4096_px_io, "abl_px_io", FSEA_CFG_4096_PXIO)
FSEA_DEFINE_KERNEL(fsea_abl8192_px_nolog, "abl_px_nolog", FSEA_CFG_8192_PXNOLOG)
FSEA_DEFINE_KERNEL(fsea_fft128st0, "st0", FSEA_CFG_128_ST0)
FSEA_DEFINE_KERNEL(fsea_fft256st0, "st0", FSEA_CFG_256_ST0)
FSEA_DEFINE_KERNEL(fsea_fft512st0, "st0", FSEA_CFG_512_ST0)
FSEA_DEFINE_KERNEL(fsea_fft1024st0, "st0", FSEA_CFG_1024_ST0)
FSEA_REGISTER_BEGIN(tune_px)
FSEA_REGISTER(fsea_fft128st0)
FSEA_REGISTER(fsea_fft256st0)
FSEA_REGISTER(fsea_fft512st0)
FSEA_REGISTER(fsea_fft1024st0)
FSEA_REGISTER(fsea_abl4096_px_nolog)
FSEA_REGISTER(fsea_abl4096_px_nost)
FSEA_REGISTER(fsea_abl4096_px_io)
FSEA_REGISTER(fsea_abl8192_px_nolog)
FSEA_REGISTER_END
