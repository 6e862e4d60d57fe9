// Tuning variants of the 1024- and 4096-point kernels.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft1024B, "B", FSEA_CFG_1024_B)
FSEA_DEFINE_KERNEL(fsea_fft1024C, "C", FSEA_CFG_1024_C)
FSEA_DEFINE_KERNEL(fsea_fft1024D, "D", FSEA_CFG_1024_D)
FSEA_DEFINE_KERNEL(fsea_fft4096B, "B", FSEA_CFG_4096_B)
extern "C" int fsea_kernels_varsmall(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft1024B_entry();
    if (n < cap) out[n++] = fsea_fft1024C_entry();
    if (n < cap) out[n++] = fsea_fft1024D_entry();
    if (n < cap) out[n++] = fsea_fft4096B_entry();
    return n;
}
