// N = 2048: one wavefront per frame, 16 x 16 x 8.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft2048, "", FSEA_CFG_2048)
extern "C" int fsea_kernels_2048(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft2048_entry();
    return n;
}
