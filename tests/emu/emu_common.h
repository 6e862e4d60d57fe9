// TEST-ONLY: what the translation units of tests/emu/libfsea_emu.so share (see hip/hip_runtime.h): the emulated launch of
// one kernel configuration.  emu_main.cpp holds the product configurations and the C entry points, emu_variants_*.cpp the
// tuning variants, so that the four compile in parallel (tests/emu_util.py).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <barrier>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "fsea_configs_tune.h"
#include "fsea_fft_core.h"
#include "fsea_tables.h"

extern thread_local std::barrier<> *g_barrier;
extern thread_local std::barrier<> *g_wave_barrier;
extern thread_local unsigned *g_wave_slot;   // one word per wave for readfirstlane
extern thread_local unsigned *g_wave_lanes;  // 64 words per wave for the cross-lane reads

// the taper window of the next calls (emu_set_window): weights, and which windowed kernel form runs (FftKernel's WIN: 1 =
// weights fetched per frame, 2 = register-resident); forced form 0 = as build_window_tables decides, 1 / 2 = that one
extern std::vector<float> g_window;
extern int g_window_mode, g_window_form;

template <class Cfg, int IN, int MODE_T, bool ROT = false, bool RUNS = false, int WIN = 0>
static inline void run_grid(fsea::FftArgs a, unsigned grid) {
    std::vector<float> win_perm;
    std::vector<fsea::TwPair> win_dc;
    if constexpr (WIN != 0) {
        int form = fsea::build_window_tables(Cfg::N, Cfg::T, Cfg::R(0), Cfg::R(Cfg::NP - 1), g_window.data(), win_perm, win_dc);
        if (g_window_form == 2 && form == 1) {  // the offset-binary form forced on a window that qualifies for the centred one
            for (auto &e : win_dc) e = fsea::TwPair{0.f, 0.f};
            form = 2;
        }
        a.win = win_perm.data();
        a.win_dc = reinterpret_cast<const fsea::cf *>(win_dc.data());
        a.win_offset = form == 2 ? 1u : 0u;
    }
    if (ROT) {
        fsea::TwPair rows[32];
        fsea::build_rotation_rows(Cfg::N, Cfg::R(0), a.rot_delta, rows);
        for (int r = 0; r < 32; ++r) a.rot_row[r] = fsea::cf{rows[r].re, rows[r].im};
    }
    std::vector<fsea::TwPair> tw;
    size_t off[5];
    const int radix[4] = {Cfg::R(0), Cfg::R(1), Cfg::R(2), Cfg::R(3)};
    fsea::build_twiddles(Cfg::NP, radix, tw, off);
    for (int i = 0; i < 4; ++i) a.tw[i] = reinterpret_cast<const fsea::cf *>(tw.data()) + off[i];
    a.tw_small = reinterpret_cast<const fsea::cf *>(tw.data());
    std::vector<fsea::TwPair> twd;
    fsea::build_deferred_table(Cfg::R(0), Cfg::R(1), twd);
    twd.resize(twd.size() + 2);  // 16-byte alignment slack
    const fsea::TwPair *twd_p = twd.data();
    if (reinterpret_cast<uintptr_t>(twd_p) & 15) {
        std::memmove(twd.data() + 1, twd.data(), (twd.size() - 2) * sizeof(fsea::TwPair));
        twd_p = twd.data() + 1;
    }
    a.tw_def = reinterpret_cast<const fsea::cf *>(twd_p);
    std::vector<unsigned> ctr(9 * 32 + 2048, 0u);
    a.ctr = ctr.data();
    for (unsigned b = 0; b < grid; ++b) {
        std::vector<fsea::cf> lds_store(fsea::FftKernel<Cfg, IN, MODE_T, ROT, RUNS, WIN>::LDS_CF + 2);
        fsea::cf *lds = lds_store.data();
        if (reinterpret_cast<uintptr_t>(lds) & 15) lds += 1;  // 16-byte alignment as on the device
        std::barrier<> bar(Cfg::WG);
        constexpr int WAVES = (Cfg::WG + 63) / 64;
        std::vector<std::unique_ptr<std::barrier<>>> wbar;
        std::vector<unsigned> wslot(WAVES, 0);
        std::vector<unsigned> wlanes(WAVES * 64, 0);
        for (int w = 0; w < WAVES; ++w) {
            wbar.emplace_back(new std::barrier<>(std::min(64, Cfg::WG - 64 * w)));
        }
        std::vector<std::thread> th;
        for (int t = 0; t < Cfg::WG; ++t) {
            th.emplace_back([&, t] {
                threadIdx.x = (unsigned)t;
                blockIdx.x = b;
                blockDim.x = Cfg::WG;
                gridDim.x = grid;
                g_barrier = &bar;
                g_wave_barrier = wbar[t / 64].get();
                g_wave_slot = &wslot[t / 64];
                g_wave_lanes = &wlanes[(t / 64) * 64];
                fsea::FftKernel<Cfg, IN, MODE_T, ROT, RUNS, WIN>::run(a, lds);
            });
        }
        for (auto &x : th) x.join();
    }
    // the last worker of the launch must have reset the ticket counter for the next launch
    for (unsigned c : ctr) {
        if (c != 0) std::abort();
    }
}

template <class Cfg>
static inline int dispatch(int in_kind, int mode_t, const fsea::FftArgs &a, unsigned grid) {
    constexpr bool U8_ONLY = (Cfg::OPT & 1048576) != 0;  // the W64 schedule has u8 kernels only
    if constexpr (U8_ONLY) {
        if (in_kind != fsea::IN_U8) return -3;
    }
    if constexpr (U8_ONLY) {
        if (mode_t == fsea::MODE_MAG) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG>(a, grid);
        else if (mode_t == fsea::MODE_DB5_U8_DCFIX) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX>(a, grid);
        else if (mode_t == fsea::MODE_DB10_U8) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB10_U8>(a, grid);
        else run_grid<Cfg, fsea::IN_U8, -1>(a, grid);
        return 0;
    } else
    if (!g_window.empty()) {  // windowed kernels: K_U8_MAG_WIN / K_U8_DB5_WIN / K_U8_DB10_WIN (compile-time mode), K_U8_WIN, K_U8_MAG_HALF_WIN
        if constexpr ((Cfg::OPT & (64 | 1048576 | 8388608)) == 0 && Cfg::TWR) {
            if ((int)g_window.size() != Cfg::N) return -5;
            if (in_kind == fsea::IN_U8_ROT) {         // K_U8_ROT_WIN: the fused frequency shift with the taper
                run_grid<Cfg, fsea::IN_U8, -1, true, false, 2>(a, grid);
            } else if (in_kind == fsea::IN_F32) {     // K_F32_WIN: f32-complex input with the taper
                run_grid<Cfg, fsea::IN_F32, -1, false, false, 2>(a, grid);
            } else if (in_kind == 3) {
                if constexpr (Cfg::FPW == 1 && (Cfg::OPT & 512) == 0) {
                    if (g_window_mode == 1) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG, false, true, 1>(a, grid);
                    else run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG, false, true, 2>(a, grid);
                } else return -4;
            } else if (mode_t == fsea::MODE_MAG) {
                if (g_window_mode == 1) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG, false, false, 1>(a, grid);
                else run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG, false, false, 2>(a, grid);
            } else if (mode_t == fsea::MODE_DB5_U8_DCFIX) {
                if (g_window_mode == 1) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX, false, false, 1>(a, grid);
                else run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX, false, false, 2>(a, grid);
            } else if (mode_t == fsea::MODE_DB10_U8) {
                if (g_window_mode == 1) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB10_U8, false, false, 1>(a, grid);
                else run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB10_U8, false, false, 2>(a, grid);
            } else {
                if (g_window_mode == 1) run_grid<Cfg, fsea::IN_U8, -1, false, false, 1>(a, grid);
                else run_grid<Cfg, fsea::IN_U8, -1, false, false, 2>(a, grid);
            }
        } else return -5;
    } else
    if (in_kind == 3) {  // the half-overlap MAG kernel (K_U8_MAG_HALF): hop == N/2, runs of g_run_len frames
        if constexpr (Cfg::FPW == 1 && (Cfg::OPT & (64 | 512 | 1048576)) == 0) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG, false, true>(a, grid);
        else return -4;
    } else
    if (in_kind == fsea::IN_U8_ROT) run_grid<Cfg, fsea::IN_U8, -1, true>(a, grid);
    else if (in_kind == fsea::IN_U8 && mode_t == fsea::MODE_MAG) run_grid<Cfg, fsea::IN_U8, fsea::MODE_MAG>(a, grid);
    else if (in_kind == fsea::IN_U8 && mode_t == fsea::MODE_DB5_U8_DCFIX) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB5_U8_DCFIX>(a, grid);
    else if (in_kind == fsea::IN_U8 && mode_t == fsea::MODE_DB10_U8) run_grid<Cfg, fsea::IN_U8, fsea::MODE_DB10_U8>(a, grid);
    else if (in_kind == fsea::IN_U8) run_grid<Cfg, fsea::IN_U8, -1>(a, grid);
    else run_grid<Cfg, fsea::IN_F32, -1>(a, grid);
    return 0;
}


// the tuning variants, three translation units: 0 = ran, -2 = not one of mine
int emu_variants_a(int n, const std::string &v, int in_kind, int mt, const fsea::FftArgs &a, unsigned grid);
int emu_variants_b(int n, const std::string &v, int in_kind, int mt, const fsea::FftArgs &a, unsigned grid);
int emu_variants_c(int n, const std::string &v, int in_kind, int mt, const fsea::FftArgs &a, unsigned grid);

#define EMU_VARIANT(NN, NAME, CFG) \
    if (n == NN && v == NAME) return dispatch<fsea::FftCfg<CFG>>(in_kind, mt, a, grid);
