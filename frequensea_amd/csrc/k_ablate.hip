// Measurement-only ablations of the 8192-point kernel (wrong results by design).
#include "fsea_configs.h"
#include "fsea_registry.h"
FSEA_DEFINE_KERNEL(fsea_abl8192_nostore, "abl_nostore", FSEA_CFG_8192_B_NOST)
FSEA_DEFINE_KERNEL(fsea_abl8192_nolds, "abl_nolds", FSEA_CFG_8192_B_NOLDS)
FSEA_DEFINE_KERNEL(fsea_abl8192_noflop, "abl_noflop", FSEA_CFG_8192_B_NOFLOP)
FSEA_DEFINE_KERNEL(fsea_abl8192_io, "abl_io", FSEA_CFG_8192_B_IO)
FSEA_DEFINE_KERNEL(fsea_abl8192_valu, "abl_valu", FSEA_CFG_8192_B_VALU)
extern "C" int fsea_kernels_ablate(fsea::KernelEntry *out, int cap) {
    int n = 0;
    if (n < cap) out[n++] = fsea_abl8192_nostore_entry();
    if (n < cap) out[n++] = fsea_abl8192_nolds_entry();
    if (n < cap) out[n++] = fsea_abl8192_noflop_entry();
    if (n < cap) out[n++] = fsea_abl8192_io_entry();
    if (n < cap) out[n++] = fsea_abl8192_valu_entry();
    return n;
}
