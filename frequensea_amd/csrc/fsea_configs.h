// fsea_configs.h -- the kernel configurations compiled into libfsea_hip.so (and, for index-math
// tests on the CPU, into tests/emu): one per transform size.
// FftCfg arguments: N, T, FPW, WPE, NP, R0, R1, R2, R3, TWL, TWR, ABL, OPT (schedule options, fsea_opt.h: fo::...).
// Every configuration streams its output rows (fo::ST_NT: nt stores -- they are not read again by the kernel, and kept out of
// the caches they would only evict other lines) and converts, clamps and packs its u8 pixels with v_cvt_pk_u8_f32, biased so
// that its round-to-nearest gives the reference's truncation (fo::PX_PACK | fo::PX_BIAS: two VALU ops per pixel fewer than cast
// + clamp + shift/or; profiles/r03_w64_and_pixel_epilogue.txt: +2 % at 8192 points, +4 % at 4096, +5.5 % at 256): together
// fo::STREAMING_PIXELS, which since round 5 also carries fo::BALANCE_PX (the pixel kernels' frame loop does not wait for the
// previous frame's row stores: FftKernel::balance_vmcnt, +2...6 % at every size; fo::BALANCE_MAG: the same in the MAG kernels at
// 1024 ... 8192 points, +2...4 % on long launches; no effect at 16384, none or negative at 512 and below).  fo::LD_NT: the input bytes streamed as well -- sizes whose pass-0 loads are at least a dword per lane
// (profiles/r02_tune_nt_*).
#pragma once

#include "fsea_opt.h"

// single-wave frames, no s_barrier; 32 points per lane from 128 points up (dword pass-0 loads)
#define FSEA_CFG_32 32, 4, 64, 2, 2, 8, 4, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS
#define FSEA_CFG_64 64, 4, 64, 2, 2, 16, 4, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS
#define FSEA_CFG_128 128, 4, 64, 2, 2, 16, 8, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS
#define FSEA_CFG_256 256, 8, 32, 2, 2, 16, 16, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS
#define FSEA_CFG_512 512, 16, 16, 2, 2, 32, 16, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS
// 1024 points: round 3 moved from 32 x 32 (2-byte pass-0 loads, one pixel / one f32 bin per lane and store) to 8 x 16 x 8:
// dwordx2 loads, four adjacent bins per lane in the last pass (dword pixel stores, 16-byte f32 stores), deferred middle-pass
// twiddles; a second exchange, still no barrier (profiles/r03_1024_three_pass.txt: DB5 / DB10 pixels +6...8 %, f32 rows +1 %)
#define FSEA_CFG_1024 1024, 32, 8, 2, 3, 8, 16, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::BALANCE_MAG | fo::WIN_DC_REGS_MAG | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_2048 2048, 64, 4, 2, 3, 16, 16, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::BALANCE_MAG | fo::WIN_DC_REGS_MAG | fo::LD_NT | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
// multi-wave frames: 32 points per lane (4096: two waves per frame, two frames per workgroup), the
// middle pass's twiddles deferred and register-resident (fo::DEFER)
#define FSEA_CFG_4096 4096, 128, 2, 2, 3, 16, 16, 16, 1, true, true, 0, fo::STREAMING_PIXELS | fo::BALANCE_MAG | fo::LD_NT | fo::DEFER | fo::LANE_ROT_LAST | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_8192 8192, 256, 1, 2, 3, 16, 16, 32, 1, true, true, 0, fo::STREAMING_PIXELS | fo::BALANCE_MAG | fo::WIN_DC_REGS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_16384 16384, 512, 1, 2, 3, 16, 32, 32, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS_MAG | fo::LD_NT | fo::DEFER | fo::TW_FUSE

// Per-(size, mode) configurations: where the modes of one size prefer different radix orders, a plan takes the one its
// mode prefers (fsea_api.hip: preferred_variant; registry variants "rows" / "px" / "rt", full kernel sets).  Measured in one
// process against the size's first configuration (profiles/r03_layouts_other_sizes.txt, r03_1024_three_pass.txt,
// r04_mode_rates.txt):
//   256 points, f32 rows (MAG, MAG_NODC, DB_F32): 4 x 8 x 8 -- dwordx4 pass-0 loads, four adjacent bins per lane: +4 %
//       (the 16 x 16 of FSEA_CFG_256 stays the pixel layout: c/fft-batch-broad.c's own size, -13 % as 4 x 8 x 8);
//   512 points, u8 pixels (DB10, DB5): 16 x 32 -- one pixel per lane and store, dword loads: +11 % (f32 rows: -3 %);
//   1024 points, the modes of the run-time-mode kernel (COMPLEX_F32, MAG_NODC_F32, DB_F32): the 32 x 32 layout of round 2 --
//       one bin per lane = one 8-byte complex store per row, where the 8 x 16 x 8 of FSEA_CFG_1024 holds four bins per lane,
//       two 16-byte stores each writing every other 16 bytes of the row: 0.297 against 0.208 ms per 2^27 samples; the f32
//       rows of that kernel 0.164 against 0.153 ms (the compile-time MAG and pixel kernels prefer 8 x 16 x 8).
#define FSEA_CFG_256_ROWS 256, 8, 32, 2, 3, 4, 8, 8, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS | fo::LD_NT | fo::DEFER | fo::LANE_ROT | fo::TW_FUSE | fo::TW_HOIST | fo::BATCH_READS
#define FSEA_CFG_512_PX 512, 16, 16, 2, 2, 16, 32, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS | fo::LD_NT | fo::TW_FUSE | fo::BATCH_READS
#define FSEA_CFG_1024_RT 1024, 32, 8, 2, 2, 32, 32, 1, 1, true, true, 0, fo::STREAMING_PIXELS | fo::WIN_DC_REGS | fo::TW_FUSE | fo::BATCH_READS

// Windowed kernels (FftKernel<..., WIN>; fsea_plan_set_window): the lane's P taper weights stay in registers for the
// workgroup's lifetime at every size (WIN = 2).  Fetching them again for every frame (WIN = 1, the tuning library's "w1"
// variants) frees 32 registers between pass 0 and the last pass and loses 2-5 % to the extra loads
// (profiles/r04_window_cost.txt).  fo::WIN_DC_REGS in a configuration above: the windowed kernels of that size also keep the
// lane's share of the DC table in registers (round 5: the per-frame LDS read of it cost 2 % of the headline launch,
// profiles/r05_window_prologue.txt); fo::WIN_DC_REGS_MAG: only in the compile-time MAG kernels (1024 as 8 x 16 x 8, 2048, 16384: the
// other kinds spill with the 2 CL extra register pairs); 4096 keeps the LDS form throughout.
#define FSEA_WIN 2
