// Tuning library only: the windowed kernels with the taper weights fetched per frame ("w1") or register-resident ("w2"),
// on the product configurations, for the A/B behind fsea_configs.h's FSEA_WIN (profiles/r04_window_cost.txt).
#include "fsea_configs.h"
#include "fsea_registry.h"
#define FSEA_WIN_AB(N, CFG)                                                 \
    FSEA_DEFINE_KERNEL_LITE(fsea_fft##N##_w1, "w1", CFG)                   \
    FSEA_DEFINE_WINDOWED(fsea_fft##N##_w1, 1)                              \
    FSEA_DEFINE_KERNEL_LITE(fsea_fft##N##_w2, "w2", CFG)                   \
    FSEA_DEFINE_WINDOWED(fsea_fft##N##_w2, 2)
FSEA_WIN_AB(256, FSEA_CFG_256)
FSEA_WIN_AB(1024, FSEA_CFG_1024)
FSEA_WIN_AB(2048, FSEA_CFG_2048)
FSEA_WIN_AB(4096, FSEA_CFG_4096)
FSEA_WIN_AB(8192, FSEA_CFG_8192)
FSEA_WIN_AB(16384, FSEA_CFG_16384)
FSEA_DEFINE_HALF_OVERLAP(fsea_fft8192_w1)
FSEA_DEFINE_HALF_OVERLAP_WIN(fsea_fft8192_w1, 1)
FSEA_DEFINE_HALF_OVERLAP(fsea_fft8192_w2)
FSEA_DEFINE_HALF_OVERLAP_WIN(fsea_fft8192_w2, 2)
FSEA_DEFINE_HALF_OVERLAP(fsea_fft16384_w1)
FSEA_DEFINE_HALF_OVERLAP_WIN(fsea_fft16384_w1, 1)
FSEA_DEFINE_HALF_OVERLAP(fsea_fft16384_w2)
FSEA_DEFINE_HALF_OVERLAP_WIN(fsea_fft16384_w2, 2)
FSEA_REGISTER_BEGIN(tune_win)
FSEA_REGISTER_WIN(fsea_fft256_w1)
FSEA_REGISTER_WIN(fsea_fft256_w2)
FSEA_REGISTER_WIN(fsea_fft1024_w1)
FSEA_REGISTER_WIN(fsea_fft1024_w2)
FSEA_REGISTER_WIN(fsea_fft2048_w1)
FSEA_REGISTER_WIN(fsea_fft2048_w2)
FSEA_REGISTER_WIN(fsea_fft4096_w1)
FSEA_REGISTER_WIN(fsea_fft4096_w2)
FSEA_REGISTER_HALF_WIN(fsea_fft8192_w1)
FSEA_REGISTER_HALF_WIN(fsea_fft8192_w2)
FSEA_REGISTER_HALF_WIN(fsea_fft16384_w1)
FSEA_REGISTER_HALF_WIN(fsea_fft16384_w2)
FSEA_REGISTER_END
