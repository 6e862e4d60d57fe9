// Schedule-option variants of the single-wave sizes (same results, different instruction order).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft2048x0, "x0", FSEA_CFG_2048_X0)
FSEA_DEFINE_KERNEL(fsea_fft1024x0, "x0", FSEA_CFG_1024_X0)
extern "C" int fsea_kernels_exp2(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft2048x0_entry();
    if (n < cap) out[n++] = fsea_fft1024x0_entry();
    return n;
}
