// N = 16384: 512 lanes x 32 points, 16 x 32 x 32, one workgroup per CU.
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_fft16384, "", FSEA_CFG_16384)
extern "C" int fsea_kernels_16384(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_fft16384_entry();
    return n;
}
