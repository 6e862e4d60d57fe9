// fsea_opt.h -- names of the FftCfg schedule options (template argument OPT) and measurement ablations (ABL).
// Every option gives identical results; what each one changes is described where FftKernel reads it (fsea_fft_core.h).
#pragma once

namespace fsea {
namespace opt {
enum : int {
    // ---- used by the product configurations (fsea_configs.h) ----
    BATCH_READS = 2,        // the LDS reads of an exchange stay one batch (scheduling fence behind them)
    TW_HOIST = 4,           // a middle pass fetches all its twiddles from LDS together with the data
    TW_FUSE = 8,            // twiddle multiply fused into the first butterfly level (dft_regs_tw)
    LANE_ROT = 16,          // middle passes: lanes renumbered so that the padded ds_read_b128 groups are conflict-free
    LANE_ROT_LAST = 32,     // the same for the last pass
    DEFER = 128,            // middle-pass twiddles deferred into the butterflies, register-resident (dft_regs_def)
    ST_NT = 4096,           // row stores streaming (nt) where one instruction writes a whole 128-byte line per frame
    LD_NT = 32768,          // input loads streaming (nt)
    PX_PACK = 2097152,      // pixel epilogue: v_cvt_pk_u8_f32 converts, clamps and packs
    PX_BIAS = 4194304,      // ... its round-to-nearest biased into the reference's truncation (no v_trunc)
    WIN_DC_REGS = 16777216, // windowed kernels only: the lane's share of the DC table stays in registers (no LDS read per frame)
    BALANCE_PX = 67108864,  // compile-time pixel kernels: the prologue issues an iteration's worth of stores into a zero-sized window, so
                            // that the frame loop's vmcnt waits leave the previous frame's row stores out (FftKernel::balance_vmcnt)
    BALANCE_MAG = 134217728, // the same in the compile-time MAG kernel
    WIN_DC_REGS_MAG = 33554432, // ... in the compile-time MAG kernels only (the nrf_fft_process / STFT path), where the other kinds would spill
    // ---- tuning library only (fsea_fft_tune.h; measured and not adopted) ----
    V2 = 64,                // the two-barrier schedule (FftKernel::run_v2)
    W64 = 1048576,          // 4096 points as 64 x 64 in one wavefront (FftKernel::run_w64)
    PW = 8388608,           // last butterfly level in power form (dft_regs_tw_pw)
    TUNE_ONLY = V2 | W64 | PW,
    // the options every product configuration shares
    STREAMING_PIXELS = ST_NT | PX_PACK | PX_BIAS | BALANCE_PX,
};
}  // namespace opt
namespace abl {  // measurement-only ablations (wrong results by design); always 0 in a product configuration
enum : int {
    NO_STORES = 1,          // no output stores
    NO_LDS = 2,             // no LDS exchange, no barriers
    NO_FLOPS = 4,           // no butterflies, no twiddles
    V2_STATIC = 8, V2_V1_LOADS = 16, V2_NO_EXCHANGE_A = 32,   // V2-schedule ablations
    NO_LOADS = 64,          // the first unit's bytes are reused for every frame
    NO_EPILOGUE_MATH = 128, // no magnitude arithmetic / no logarithm
    DROP_16_OPS = 256,      // the middle pass leaves out 16 packed ops per lane-frame (the `minus` FMA of its first level), DROP_32_OPS = 512:
    DROP_32_OPS = 512,      // of its first two levels -- what a radix-4 regrouping could save at most, as a rate (profiles/r04_radix4_rejected.txt)
};
}  // namespace abl
}  // namespace fsea
namespace fo = fsea::opt;  // short form for the configuration lists (fsea_configs.h)
