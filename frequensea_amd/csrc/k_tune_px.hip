// Measurement-only ablations of the pixel (DB5 / DB10) kernels at 4096 and 8192 points (wrong results by design).
#include "fsea_configs_tune.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_abl4096_px_nolog, "abl_px_nolog", FSEA_CFG_4096_PXNOLOG)
FSEA_DEFINE_KERNEL(fsea_abl4096_px_nost, "abl_px_nost", FSEA_CFG_4096_PXNOST)
FSEA_DEFINE_KERNEL(fsea_abl4096_px_wide, "abl_px_wide", FSEA_CFG_4096_PXWIDE)
FSEA_DEFINE_KERNEL(fsea_abl4096_px_io, "abl_px_io", FSEA_CFG_4096_PXIO)
FSEA_DEFINE_KERNEL(fsea_abl4096_px_io_wide, "abl_px_io_wide", FSEA_CFG_4096_PXIOWIDE)
FSEA_DEFINE_KERNEL(fsea_abl8192_px_nolog, "abl_px_nolog", FSEA_CFG_8192_PXNOLOG)
FSEA_DEFINE_KERNEL(fsea_abl8192_px_wide, "abl_px_wide", FSEA_CFG_8192_PXWIDE)
FSEA_REGISTER_BEGIN(tune_px)
FSEA_REGISTER(fsea_abl4096_px_nolog)
FSEA_REGISTER(fsea_abl4096_px_nost)
FSEA_REGISTER(fsea_abl4096_px_wide)
FSEA_REGISTER(fsea_abl4096_px_io)
FSEA_REGISTER(fsea_abl4096_px_io_wide)
FSEA_REGISTER(fsea_abl8192_px_nolog)
FSEA_REGISTER(fsea_abl8192_px_wide)
FSEA_REGISTER_END
