#!/usr/bin/env python3
"""Generate tests/golden/lua_scene_traces.json: the call traces of the reference's five FFT scenes.

Run in the build container only.  oracle/_ref/lua_trace (oracle/Makefile: the reference's vendored Lua 5.3 interpreter,
/root/reference/externals/lua/src, under this repository's tracing host oracle/lua_trace.c) runs
  lua/fft.lua, lua/fft-shifted.lua, lua/fft-sea.lua, lua/fft-sea-auto.lua, lua/fft-sea-sick.lua
UNMODIFIED, with lua/_keys.lua loaded first as src/main.cpp:1201 does, for a fixed number of draw()s with scripted key
events, and writes down every call the scripts make into the nrf_* / ngl_texture_update surface: function, arguments as
the C function receives them (nrf_fft_shift's d narrowed to float, src/main.cpp:788), ids of the objects involved, type /
length / channels and a checksum of every buffer that crosses the boundary, and the garbage collections that release
buffers (src/main.cpp:1316).  The numbers behind the calls come from the oracle (oracle/fsea_oracle.c).

The replay device reads a file of four 262144-byte blocks made from the committed fixtures (the first 32768 bytes of
rf-100.900-1, rf-202.500-1/2/3 from tests/golden/rfdata_golden.npz, zero beyond): the scenes name
"../rfdata/rf-200.500-big.raw", which the reference does not ship (its own replay would then use one zero block,
src/nrf.c:271-276); the requested name is recorded, the substitute is what tests/test_gpu_parity.py rebuilds.

The traces are DATA (events and numbers), not the scripts: neither the interpreter nor any .lua file travels to the GPU
box.  tests/test_gpu_parity.py::test_lua_scene_trace_replay replays them call by call against libfsea_nrf.so.

Key codes (lua/_keys.lua): KEY_A 65, KEY_LEFT_BRACKET 91, KEY_RIGHT_BRACKET 93, KEY_RIGHT 262, KEY_LEFT 263; mods: 1 Shift,
4 Alt.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HERE_OUT = HERE            # where the trace file is written (a test redirects it to compare with the committed one)
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_LUA = "/root/reference/lua"
TRACER = os.path.join(ROOT, "oracle", "_ref", "lua_trace")
BLOCK_KEYS = ["rf_100p900_1", "rf_202p500_1", "rf_202p500_2", "rf_202p500_3"]

# scene: (frames, key events "frame:key:mods")
RUNS = {
    # _keys.lua's frequency handler: right (d = 0.1), left, Shift+right (d = 10: the history is cleared), Alt+left (d = 0.001)
    "fft.lua": (9, "3:262:0,5:263:0,6:262:1,8:263:4"),
    # its own handler on top: [ and ] make a NEW shifter at shift -/+ 10 kHz (phase starts over), then a retune
    "fft-shifted.lua": (9, "2:91:0,4:93:0,5:93:0,7:262:0"),
    # right / Alt+left retunes, then KEY_A: switch_freq() fades out over 100 frames and jumps to the next station with live
    # rows in the history (97.6 -> 169.8 MHz: d = 72.2), then fades in
    "fft-sea.lua": (112, "2:262:0,4:263:4,6:65:0"),
    # retunes by itself: set_freq(freq + 0.01) at the end of EVERY draw (nrf_fft_shift(fft, ~500): round(128 / 500) = 0)
    "fft-sea-auto.lua": (8, "3:262:0"),
    # _keys.lua's handler again, on a 128 x 128 history
    "fft-sea-sick.lua": (7, "2:262:0,4:263:1"),
}

# THIS REPOSITORY'S scenes (tests/golden/scenes/, not the reference's): a scene that chooses its taper through
# nrf_fft_set_window, the addition of include/nrf.h, traced under the same interpreter and host.  Kept apart in the file
# ("own_scenes") so that "scenes" stays what the reference's five scripts do.
# KEY_B 66 (Blackman-Harris), a retune (right), KEY_F 70 (flat top), KEY_R 82 (rectangular), KEY_H 72 (Hann again)
OWN_SCENES_DIR = os.path.join(HERE, "scenes")
OWN_RUNS = {
    "fft-windowed.lua": (10, "3:66:0,5:262:0,6:70:0,8:82:0,9:72:0"),
}


def replay_blocks():
    g = np.load(os.path.join(HERE, "rfdata_golden.npz"))
    blocks = []
    for key in BLOCK_KEYS:
        blk = np.zeros(262144, np.uint8)
        raw = g[key + "__raw"]
        blk[: raw.size] = raw
        blocks.append(blk)
    return np.concatenate(blocks)


def main():
    if not (os.path.exists(TRACER) and os.path.isdir(REF_LUA)):
        sys.exit("needs oracle/_ref/lua_trace (make -C oracle) and %s: the build container only" % REF_LUA)
    out = {"generator": "tests/golden/make_lua_traces.py + oracle/lua_trace.c under the reference's vendored Lua 5.3",
           "replay_blocks": BLOCK_KEYS, "scenes": {}}
    with tempfile.TemporaryDirectory() as tmp:
        replay = os.path.join(tmp, "replay.raw")
        replay_blocks().tofile(replay)
        out["own_scenes"] = {}
        runs = [("scenes", scene, None, fk) for scene, fk in RUNS.items()]
        runs += [("own_scenes", scene, os.path.join(OWN_SCENES_DIR, scene), fk) for scene, fk in OWN_RUNS.items()]
        for group, scene, own_path, (frames, keys) in runs:
            trace = os.path.join(tmp, scene + ".jsonl")
            cmd = [TRACER, "--lua-dir", REF_LUA, "--scene", scene, "--replay", replay, "--frames", str(frames), "--keys", keys,
                   "--out", trace]
            if own_path:
                cmd += ["--scene-path", own_path]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit("%s: lua_trace failed (%d)\n%s" % (scene, r.returncode, r.stderr))
            events, stubs = [], {}
            for line in open(trace):
                ev = json.loads(line)
                if ev["ev"] == "stub":                       # the GL / window / audio side: counted per frame, not replayed
                    stubs[ev["fn"]] = stubs.get(ev["fn"], 0) + 1
                    continue
                if ev["ev"] in ("frame", "setup_done"):
                    ev["stubs"] = stubs
                    stubs = {}
                events.append(ev)
            out[group][scene] = {"events": events, "script_output": r.stdout.splitlines()}
            calls = [e["fn"] for e in events if e["ev"] == "call"]
            print("%-18s %4d frames, %5d events: %s" % (scene, frames, len(events),
                  ", ".join("%s x%d" % (f, calls.count(f)) for f in sorted(set(calls)))))
    path = os.path.join(HERE_OUT, "lua_scene_traces.json")
    with open(path, "w") as fp:
        json.dump(out, fp, separators=(",", ":"))
        fp.write("\n")
    print("wrote %s (%d bytes)" % (path, os.path.getsize(path)))


if __name__ == "__main__":
    main()
