// N = 8192: 256 lanes x 32 points, 16 x 16 x 32, two LDS exchanges.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft8192, "", FSEA_CFG_8192)
extern "C" int fsea_kernels_8192(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft8192_entry();
    return n;
}
