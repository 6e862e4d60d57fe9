"""frequensea_amd -- MI355X-native IQ-FFT spectrum path behind frequensea's nrf_fft API.

The product is two shared libraries built from frequensea_amd/csrc and
frequensea_amd/host:

  libfsea_hip.so   hand-written gfx950 kernels + the C ABI of include/fsea.h
  libfsea_nrf.so   C99 host layer with the reference's nut_buffer / nrf_fft /
                   nrf_device (dummy source) API on top of that C ABI

This Python package is only a ctypes view of those libraries for tests,
bench.py and scripting; it contains no compute of its own and no CPU fallback.
"""
from .fsea import (  # noqa: F401
    MODE_MAG_F32, MODE_DB10_U8, MODE_DB5_U8_DCFIX, MODE_COMPLEX_F32, MODE_MAG_NODC_F32, MODE_DB_F32,
    FseaError, Plan, build, device_count, hip_lib, composite_max_device, stitch_tiles_device, lib_path,
)
