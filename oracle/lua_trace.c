/*
 * lua_trace.c -- runs the reference's own FFT scenes (lua/fft.lua, fft-shifted.lua, fft-sea.lua, fft-sea-auto.lua,
 * fft-sea-sick.lua) UNMODIFIED under the reference's vendored Lua 5.3 interpreter and writes down every call they make
 * into the nrf_* / nut_* / ngl_* surface, with the arguments the C functions receive and checksums of the buffers that
 * cross the boundary.  TEST INFRASTRUCTURE ONLY (a checker, like oracle/_ref/fft-stitch-broad): it is built into
 * oracle/_ref/ in the build container, where /root/reference exists, by oracle/Makefile; the traces it writes are committed
 * under tests/golden/ (tests/golden/make_lua_traces.py) and replayed against libfsea_nrf.so by the GPU tier.  Neither this
 * binary nor the interpreter nor any .lua file is needed on the GPU box.
 *
 * What it restates (paths under /root/reference) -- this file is this repository's own code, not a copy of main.cpp:
 *   the Lua environment of src/main.cpp: l_init (1091-1208: the globals and constants a scene may name, then
 *     ../lua/_keys.lua), the scene file (1248), setup() (1269), then per rendered frame draw() (978), key events
 *     (on_key(key, mods), 1074-1078) and a full garbage collection (1316);
 *   the object tables of src/main.cpp:36-47 ({__type__, __ptr__}; buffers also carry length / channels / size_bytes,
 *     121-137; devices carry sample_rate, 632-640) and the fatal type check of 49-63;
 *   l_nrf_fft_shift's narrowing of its argument to float (786-790): nrf_fft_shift receives (double)(float)d;
 *   ngl_texture_update's size check (src/ngl.c:224-227: width * height > buffer->length is fatal) and its f64 -> f32
 *     narrowing (228-239), of which the trace keeps a checksum;
 *   the file-replay device (src/nrf.c:255-283: whole 262144-byte blocks of the file, one zero block when it cannot be
 *     opened, sample rate 5e6; 95-110: the byte flip; 352-357: the samples buffer).  The receive thread's 60 Hz pacing
 *     (153-170) is replaced by one block per rendered frame, so that a trace is deterministic.
 * The numbers behind the calls (spectra, history, shifter) come from the oracle (fsea_oracle.h), so that the scripts'
 * control flow sees real values; every other ngl_* / nwm_* / nosc_* / nrf_player_* global is a recording stub.
 *
 * Usage: lua_trace --lua-dir /root/reference/lua --scene fft.lua --replay blocks.raw --frames 8
 *                  [--keys FRAME:KEY:MODS,...] --out trace.jsonl
 * One JSON object per line: {"ev": "call", "fn": ..., ...} in call order, {"ev": "frame", "n": k} after each draw().
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lauxlib.h"
#include "lua.h"
#include "lualib.h"

#include "fsea_oracle.h"

#define BLOCK_BYTES 262144   /* NRF_BUFFER_SIZE_BYTES, src/nrf.h:19 */
#define BLOCK_SAMPLES 131072 /* NRF_SAMPLES_LENGTH, src/nrf.h:20 */
#define BUF_U8 1             /* NUT_BUFFER_U8, src/nut.h:14-18 */
#define BUF_F64 2

typedef struct {
    int id, type, length, channels;
    uint8_t *u8;
    double *f64;
} tbuffer;

typedef struct {
    int id;
    uint8_t *blocks; /* as read from the file: raw bytes */
    int n_blocks, index;
    uint8_t samples[BLOCK_BYTES]; /* after the byte flip: what get_samples_buffer hands out */
} tdevice;

typedef struct {
    int id, n, h;
    double *history;
    double *window; /* nrf_fft_set_window (this repository's addition, include/nrf.h): NULL = the reference's rectangular frames */
} tfft;

typedef struct {
    int id, freq_offset, sample_rate;
    double cosine, sine;
    double *out; /* BLOCK_SAMPLES complex values */
    int have;
} tshifter;

static FILE *g_out;
static const char *g_replay;
static int g_next_id = 1;
static int g_frame = 0;
static tdevice *g_devices[16];
static int g_n_devices = 0;

/* ---- JSON helpers ---- */
static void jstr(const char *s) {
    fputc('"', g_out);
    for (; s && *s; s++) {
        if (*s == '"' || *s == '\\') fputc('\\', g_out);
        if ((unsigned char)*s >= 0x20) fputc(*s, g_out);
    }
    fputc('"', g_out);
}
static void jnum(double v) {
    if (isinf(v)) fprintf(g_out, v > 0 ? "\"inf\"" : "\"-inf\"");
    else if (isnan(v)) fprintf(g_out, "\"nan\"");
    else fprintf(g_out, "%.17g", v);
}
/* FNV-1a over bytes: the identity of a u8 buffer */
static uint64_t fnv(const uint8_t *p, size_t n) {
    uint64_t hsh = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) {
        hsh ^= p[i];
        hsh *= 1099511628211ull;
    }
    return hsh;
}
static void jbuffer(const char *key, const tbuffer *b) {
    fprintf(g_out, ", \"%s\": {\"id\": %d, \"type\": %d, \"length\": %d, \"channels\": %d", key, b->id, b->type, b->length, b->channels);
    if (b->type == BUF_U8) {
        fprintf(g_out, ", \"fnv\": \"%016llx\"", (unsigned long long)fnv(b->u8, (size_t)b->length * (size_t)b->channels));
    } else {
        double sum = 0, asum = 0;
        for (size_t i = 0; i < (size_t)b->length * (size_t)b->channels; i++) {
            sum += b->f64[i];
            asum += fabs(b->f64[i]);
        }
        fprintf(g_out, ", \"sum\": ");
        jnum(sum);
        fprintf(g_out, ", \"abs_sum\": ");
        jnum(asum);
    }
    fputc('}', g_out);
}

/* ---- object tables (src/main.cpp:36-63) ---- */
static void push_object(lua_State *L, const char *type, void *ptr) {
    lua_newtable(L);
    luaL_getmetatable(L, type);
    lua_setmetatable(L, -2);
    lua_pushstring(L, type);
    lua_setfield(L, -2, "__type__");
    lua_pushlightuserdata(L, ptr);
    lua_setfield(L, -2, "__ptr__");
}
static void *from_object(lua_State *L, const char *type, int index) {
    luaL_checktype(L, index, LUA_TTABLE);
    lua_getfield(L, index, "__type__");
    const char *have = lua_tostring(L, -1);
    if (have == NULL || strcmp(type, have) != 0) {
        fprintf(stderr, "Lua: invalid type for param %d: expected %s, was %s\n", index, type, have ? have : "(none)");
        exit(EXIT_FAILURE); /* main.cpp:60-61 */
    }
    lua_getfield(L, index, "__ptr__");
    void *p = lua_touserdata(L, -1);
    lua_pop(L, 2);
    return p;
}
static void push_buffer(lua_State *L, tbuffer *b) {
    push_object(L, "nut_buffer", b);
    lua_pushinteger(L, b->length);
    lua_setfield(L, -2, "length");
    lua_pushinteger(L, b->channels);
    lua_setfield(L, -2, "channels");
    lua_pushinteger(L, (lua_Integer)b->length * b->channels * (b->type == BUF_U8 ? 1 : 8));
    lua_setfield(L, -2, "size_bytes");
}
static tbuffer *new_buffer(int type, int length, int channels) {
    tbuffer *b = (tbuffer *)calloc(1, sizeof(tbuffer));
    b->id = g_next_id++;
    b->type = type;
    b->length = length;
    b->channels = channels;
    if (type == BUF_U8) b->u8 = (uint8_t *)calloc((size_t)length * (size_t)channels, 1);
    else b->f64 = (double *)calloc((size_t)length * (size_t)channels, sizeof(double));
    return b;
}
static int gc_buffer(lua_State *L) { /* the __gc of main.cpp:1095: nut_buffer_free */
    lua_getfield(L, 1, "__ptr__");
    tbuffer *b = (tbuffer *)lua_touserdata(L, -1);
    if (b) {
        fprintf(g_out, "{\"ev\": \"gc\", \"buffer\": %d}\n", b->id);
        free(b->u8);
        free(b->f64);
        free(b);
    }
    return 0;
}

/* ---- nrf_device (file replay) ---- */
static void device_ingest(tdevice *d) { /* src/nrf.c:95-110: the block the receive thread would have processed */
    orc_flip_u8(d->blocks + (size_t)d->index * BLOCK_BYTES, d->samples, BLOCK_BYTES);
}
static int l_device_new(lua_State *L) {
    const double freq = luaL_checknumber(L, 1);
    const char *file = lua_tostring(L, 2);
    tdevice *d = (tdevice *)calloc(1, sizeof(tdevice));
    d->id = g_next_id++;
    const char *path = g_replay ? g_replay : file;
    FILE *fp = path ? fopen(path, "rb") : NULL;
    if (fp) { /* src/nrf.c:262-270 */
        fseek(fp, 0L, SEEK_END);
        long size = ftell(fp);
        rewind(fp);
        d->n_blocks = (int)(size / BLOCK_BYTES);
        d->blocks = (uint8_t *)calloc((size_t)(d->n_blocks > 0 ? d->n_blocks : 1), BLOCK_BYTES);
        if (d->n_blocks > 0 && fread(d->blocks, (size_t)d->n_blocks * BLOCK_BYTES, 1, fp) != 1) {
            fprintf(stderr, "lua_trace: short read of %s\n", path);
            exit(EXIT_FAILURE);
        }
        if (d->n_blocks == 0) d->n_blocks = 1;
        fclose(fp);
    } else { /* src/nrf.c:271-276: "Using empty buffer" */
        d->blocks = (uint8_t *)calloc(1, BLOCK_BYTES);
        d->n_blocks = 1;
    }
    device_ingest(d);
    if (g_n_devices < 16) g_devices[g_n_devices++] = d;
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_device_new\", \"freq_mhz\": ");
    jnum(freq);
    fprintf(g_out, ", \"data_file\": ");
    jstr(file ? file : "");
    fprintf(g_out, ", \"replayed_blocks\": %d, \"ret\": %d, \"sample_rate\": 5000000}\n", d->n_blocks, d->id);
    push_object(L, "nrf_device", d);
    lua_pushinteger(L, 5000000); /* DUMMY_DEFAULT_SAMPLE_RATE, src/nrf.c:254 */
    lua_setfield(L, -2, "sample_rate");
    return 1;
}
static int l_device_set_frequency(lua_State *L) {
    tdevice *d = (tdevice *)from_object(L, "nrf_device", 1);
    const double freq = luaL_checknumber(L, 2);
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_device_set_frequency\", \"device\": %d, \"freq_mhz\": ", d->id);
    jnum(freq);
    fprintf(g_out, ", \"ret\": ");
    jnum(freq); /* the replay device does not clamp: src/nrf.c:85-93 */
    fprintf(g_out, "}\n");
    lua_pushnumber(L, freq);
    return 1;
}
static int l_device_get_samples_buffer(lua_State *L) {
    tdevice *d = (tdevice *)from_object(L, "nrf_device", 1);
    tbuffer *b = new_buffer(BUF_U8, BLOCK_SAMPLES, 2); /* src/nrf.c:352-357 */
    memcpy(b->u8, d->samples, BLOCK_BYTES);
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_device_get_samples_buffer\", \"device\": %d, \"block\": %d", d->id, d->index);
    jbuffer("ret", b);
    fprintf(g_out, "}\n");
    push_buffer(L, b);
    return 1;
}

/* ---- nrf_fft ---- */
static int l_fft_new(lua_State *L) {
    tfft *f = (tfft *)calloc(1, sizeof(tfft));
    f->id = g_next_id++;
    f->n = (int)luaL_checkinteger(L, 1);
    f->h = (int)luaL_checkinteger(L, 2);
    f->history = (double *)calloc((size_t)f->n * (size_t)f->h, sizeof(double));
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_fft_new\", \"fft_size\": %d, \"fft_history_size\": %d, \"ret\": %d}\n", f->n, f->h, f->id);
    push_object(L, "nrf_fft", f);
    return 1;
}
static int l_fft_shift(lua_State *L) {
    tfft *f = (tfft *)from_object(L, "nrf_fft", 1);
    const double d_lua = luaL_checknumber(L, 2);
    const float d = (float)d_lua; /* src/main.cpp:788: `float d = luaL_checknumber(L, 2);` */
    orc_fft_shift(f->history, f->n, f->h, (double)d);
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_fft_shift\", \"fft\": %d, \"d_lua\": ", f->id);
    jnum(d_lua);
    fprintf(g_out, ", \"d\": ");
    jnum((double)d);
    fprintf(g_out, "}\n");
    return 0;
}
static int l_fft_process(lua_State *L) {
    tfft *f = (tfft *)from_object(L, "nrf_fft", 1);
    tbuffer *b = (tbuffer *)from_object(L, "nut_buffer", 2);
    const int n = f->n;
    double *x = (double *)calloc((size_t)n * 2, sizeof(double)), *spec = (double *)calloc((size_t)n * 2, sizeof(double));
    const int have = b->length * b->channels / 2;
    /* src/nrf.c:599-614 on the first fft_size samples (the rest of the unpack loop is never read, 615) */
    if (b->type == BUF_U8) {
        uint8_t *pad = (uint8_t *)calloc((size_t)n * 2, 1);
        memcpy(pad, b->u8, (size_t)(have < n ? have : n) * 2);
        orc_unpack_center_u8(pad, (size_t)n, x);
        free(pad);
    } else {
        double *pad = (double *)calloc((size_t)n * 2, sizeof(double));
        memcpy(pad, b->f64, sizeof(double) * (size_t)(have < n ? have : n) * 2);
        orc_unpack_center_f64(pad, (size_t)n, x);
        free(pad);
    }
    if (f->window) { /* the taper beside the (-1)^ii of the unpack loop: x[ii] = (-1)^ii * w[ii] * value (orc_rows_windowed) */
        for (int j = 0; j < n; j++) {
            x[2 * j] *= f->window[j];
            x[2 * j + 1] *= f->window[j];
        }
    }
    orc_fft_forward(x, spec, n);              /* 615 */
    orc_history_scroll(f->history, n, f->h);  /* 616-617 */
    orc_mag_row(spec, n, f->history);         /* 619-630 */
    free(x);
    free(spec);
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_fft_process\", \"fft\": %d, \"buffer\": %d}\n", f->id, b->id);
    return 0;
}
/* nrf_fft_set_window(fft, name): NOT a function of the reference (its scenes never call it); the binding INTEGRATION.md shows
 * beside l_nrf_fft_shift, so that a scene written for this library (tests/golden/scenes/) can be traced like the reference's.
 * The weights are the oracle's, rounded to float as the kernel holds them. */
static int l_fft_set_window(lua_State *L) {
    tfft *f = (tfft *)from_object(L, "nrf_fft", 1);
    const char *name = luaL_checkstring(L, 2);
    static const char *const names[] = {"rect", "hann", "hamming", "blackman", "blackmanharris", "flattop"};
    int kind = -1;
    for (int k = 0; k < 6; k++) {
        if (!strcmp(name, names[k])) kind = k;
    }
    if (!strcmp(name, "none") || name[0] == '\0') kind = 0;
    if (kind < 0) return luaL_error(L, "nrf_fft_set_window: unknown taper '%s'", name);
    free(f->window);
    f->window = NULL;
    if (kind > 0) {
        f->window = (double *)calloc((size_t)f->n, sizeof(double));
        if (orc_window_fill(kind, f->n, f->window) != 0) return luaL_error(L, "nrf_fft_set_window: orc_window_fill failed");
        for (int j = 0; j < f->n; j++) f->window[j] = (double)(float)f->window[j];
    }
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_fft_set_window\", \"fft\": %d, \"name\": ", f->id);
    jstr(name);
    fprintf(g_out, "}\n");
    return 0;
}
static int l_fft_get_buffer(lua_State *L) {
    tfft *f = (tfft *)from_object(L, "nrf_fft", 1);
    tbuffer *b = new_buffer(BUF_F64, f->n * f->h, 1); /* src/nrf.c:633-635 */
    memcpy(b->f64, f->history, sizeof(double) * (size_t)f->n * (size_t)f->h);
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_fft_get_buffer\", \"fft\": %d", f->id);
    jbuffer("ret", b);
    fprintf(g_out, "}\n");
    push_buffer(L, b);
    return 1;
}

/* ---- nrf_freq_shifter (src/nrf.c:826-880) ---- */
static int l_shifter_new(lua_State *L) {
    tshifter *s = (tshifter *)calloc(1, sizeof(tshifter));
    s->id = g_next_id++;
    s->freq_offset = (int)luaL_checkinteger(L, 1); /* src/main.cpp:854-855 (a float with an integral value converts) */
    s->sample_rate = (int)luaL_checkinteger(L, 2);
    s->cosine = 1.0;
    s->sine = 0.0;
    s->out = (double *)calloc((size_t)BLOCK_SAMPLES * 2, sizeof(double));
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_freq_shifter_new\", \"freq_offset\": %d, \"sample_rate\": %d, \"ret\": %d}\n",
            s->freq_offset, s->sample_rate, s->id);
    push_object(L, "nrf_freq_shifter", s);
    return 1;
}
static int l_shifter_process(lua_State *L) {
    tshifter *s = (tshifter *)from_object(L, "nrf_freq_shifter", 1);
    tbuffer *b = (tbuffer *)from_object(L, "nut_buffer", 2);
    const size_t n = (size_t)b->length * (size_t)b->channels / 2;
    orc_freq_shift(b->type == BUF_U8 ? b->u8 : NULL, b->type == BUF_U8 ? NULL : b->f64, n < BLOCK_SAMPLES ? n : BLOCK_SAMPLES,
                   s->freq_offset, s->sample_rate, &s->cosine, &s->sine, s->out);
    s->have = (int)(n < BLOCK_SAMPLES ? n : BLOCK_SAMPLES);
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_freq_shifter_process\", \"shifter\": %d, \"buffer\": %d}\n", s->id, b->id);
    return 0;
}
static int l_shifter_get_buffer(lua_State *L) {
    tshifter *s = (tshifter *)from_object(L, "nrf_freq_shifter", 1);
    /* src/nrf.c:853-856: the shifter's buffer is nut_buffer_new_f64(length * channels, 2): `length` counts VALUES, so the
     * buffer holds twice as many doubles as the block has, the second half zero */
    tbuffer *b = new_buffer(BUF_F64, s->have * 2, 2);
    memcpy(b->f64, s->out, sizeof(double) * (size_t)s->have * 2);
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"nrf_freq_shifter_get_buffer\", \"shifter\": %d", s->id);
    jbuffer("ret", b);
    fprintf(g_out, "}\n");
    push_buffer(L, b);
    return 1;
}

/* ---- ngl_texture_update: the consumer of the fft_buffer (src/ngl.c:222-245) ---- */
static int l_texture_update(lua_State *L) {
    tbuffer *b = (tbuffer *)from_object(L, "nut_buffer", 2);
    const int width = (int)luaL_checkinteger(L, 3), height = (int)luaL_checkinteger(L, 4);
    if (width * height > b->length) { /* src/ngl.c:224-227 */
        fprintf(stderr, "ERROR ngl_texture_update: Invalid width / height %d %d (buffer length %d)\n", width, height, b->length);
        exit(EXIT_FAILURE);
    }
    const size_t size = (size_t)width * (size_t)height * (size_t)b->channels;
    double sum = 0;
    for (size_t i = 0; i < size; i++) sum += b->type == BUF_F64 ? (double)(float)b->f64[i] : (double)b->u8[i];
    fprintf(g_out, "{\"ev\": \"call\", \"fn\": \"ngl_texture_update\", \"buffer\": %d, \"width\": %d, \"height\": %d, \"f32_sum\": ", b->id, width, height);
    jnum(sum);
    fprintf(g_out, "}\n");
    return 0;
}

/* ---- everything else a scene may name: recording stubs ---- */
static int l_stub(lua_State *L) {
    const char *name = lua_tostring(L, lua_upvalueindex(1));
    fprintf(g_out, "{\"ev\": \"stub\", \"fn\": ");
    jstr(name);
    fprintf(g_out, ", \"nargs\": %d}\n", lua_gettop(L));
    if (strcmp(name, "nwm_get_time") == 0) {
        lua_pushnumber(L, g_frame / 60.0);
        return 1;
    }
    lua_newtable(L); /* a camera, shader, model, texture, font, player, server ...: an opaque object */
    lua_pushstring(L, name);
    lua_setfield(L, -2, "__stub__");
    return 1;
}
static int l_global_index(lua_State *L) { /* _G's __index: a global nobody defined */
    const char *key = lua_tostring(L, 2);
    if (key && (strncmp(key, "ngl_", 4) == 0 || strncmp(key, "nwm_", 4) == 0 || strncmp(key, "nosc_", 5) == 0 ||
                strncmp(key, "nrf_player_", 11) == 0 || strncmp(key, "nvr_", 4) == 0)) {
        lua_pushstring(L, key);
        lua_pushcclosure(L, l_stub, 1);
        return 1;
    }
    lua_pushnil(L);
    return 1;
}

static void reg(lua_State *L, const char *name, lua_CFunction fn) {
    lua_pushcfunction(L, fn);
    lua_setglobal(L, name);
}
static void regtype(lua_State *L, const char *type, lua_CFunction gc) {
    luaL_newmetatable(L, type);
    if (gc) {
        lua_pushcfunction(L, gc);
        lua_setfield(L, -2, "__gc");
    }
    lua_pop(L, 1);
}
static void konst(lua_State *L, const char *name, int v) {
    lua_pushinteger(L, v);
    lua_setglobal(L, name);
}

static int call_global(lua_State *L, const char *name, int nargs) { /* src/main.cpp:94-113 */
    if (lua_pcall(L, nargs, 0, 0)) {
        fprintf(stderr, "Error calling %s(): %s\n", name, lua_tostring(L, -1));
        lua_pop(L, 1);
        return -1;
    }
    return 0;
}

int main(int argc, char **argv) {
    const char *lua_dir = NULL, *scene = NULL, *scene_path = NULL, *keys = "";
    int frames = 8;
    g_out = stdout;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--lua-dir") && i + 1 < argc) lua_dir = argv[++i];
        else if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene = argv[++i];
        else if (!strcmp(argv[i], "--scene-path") && i + 1 < argc) scene_path = argv[++i]; /* a scene outside --lua-dir (this repository's own) */
        else if (!strcmp(argv[i], "--replay") && i + 1 < argc) g_replay = argv[++i];
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) { /* the trace; stdout stays the scripts' own print() */
            g_out = fopen(argv[++i], "w");
            if (!g_out) {
                perror("lua_trace: --out");
                return 2;
            }
        }
    }
    if (!lua_dir || !scene) {
        fprintf(stderr, "usage: lua_trace --lua-dir DIR --scene FILE.lua [--replay BLOCKS.raw] [--frames N] [--keys F:KEY:MODS,...] [--out TRACE.jsonl]\n");
        return 2;
    }
    lua_State *L = luaL_newstate();
    /* The scenes are the reference's files: upstream, untrusted content run in the build container.  They need
     * arithmetic, strings and tables; they get those and nothing that reaches the file system or a process: no io, os,
     * package (require / loadlib) or debug library, and the base library's file loaders are removed.  main() below loads
     * the two files itself through the C API. */
    static const luaL_Reg safe_libs[] = {{"_G", luaopen_base},         {LUA_TABLIBNAME, luaopen_table}, {LUA_STRLIBNAME, luaopen_string},
                                         {LUA_MATHLIBNAME, luaopen_math}, {LUA_UTF8LIBNAME, luaopen_utf8}, {NULL, NULL}};
    for (const luaL_Reg *lib = safe_libs; lib->func; lib++) {
        luaL_requiref(L, lib->name, lib->func, 1);
        lua_pop(L, 1);
    }
    static const char *const removed[] = {"dofile", "loadfile", "load", "loadstring", "require", NULL};
    for (int k = 0; removed[k]; k++) {
        lua_pushnil(L);
        lua_setglobal(L, removed[k]);
    }
    regtype(L, "nut_buffer", gc_buffer);
    regtype(L, "nrf_device", NULL);
    regtype(L, "nrf_fft", NULL);
    regtype(L, "nrf_freq_shifter", NULL);
    reg(L, "nrf_device_new", l_device_new);
    reg(L, "nrf_device_set_frequency", l_device_set_frequency);
    reg(L, "nrf_device_get_samples_buffer", l_device_get_samples_buffer);
    reg(L, "nrf_fft_new", l_fft_new);
    reg(L, "nrf_fft_shift", l_fft_shift);
    reg(L, "nrf_fft_process", l_fft_process);
    reg(L, "nrf_fft_get_buffer", l_fft_get_buffer);
    reg(L, "nrf_fft_set_window", l_fft_set_window);
    reg(L, "nrf_freq_shifter_new", l_shifter_new);
    reg(L, "nrf_freq_shifter_process", l_shifter_process);
    reg(L, "nrf_freq_shifter_get_buffer", l_shifter_get_buffer);
    reg(L, "ngl_texture_update", l_texture_update);
    konst(L, "NUT_BUFFER_U8", BUF_U8);
    konst(L, "NUT_BUFFER_F64", BUF_F64);
    konst(L, "NRF_SAMPLES_LENGTH", BLOCK_SAMPLES);
    konst(L, "NRF_DEMODULATE_RAW", 0);
    konst(L, "NRF_DEMODULATE_WBFM", 1);
    konst(L, "GL_POINTS", 0);
    konst(L, "GL_LINES", 1);
    konst(L, "GL_LINE_LOOP", 2);
    konst(L, "GL_LINE_STRIP", 3);
    konst(L, "GL_TRIANGLES", 4);
    konst(L, "GL_TRIANGLE_STRIP", 5);
    konst(L, "GL_TRIANGLE_FAN", 6);
    /* any other ngl_ / nwm_ / nosc_ / nrf_player_ global resolves to a recording stub */
    lua_pushglobaltable(L);
    lua_newtable(L);
    lua_pushcfunction(L, l_global_index);
    lua_setfield(L, -2, "__index");
    lua_setmetatable(L, -2);
    lua_pop(L, 1);

    char path[4096];
    snprintf(path, sizeof(path), "%s/_keys.lua", lua_dir); /* src/main.cpp:1201 */
    if (luaL_loadfile(L, path) || lua_pcall(L, 0, 0, 0)) {
        fprintf(stderr, "%s\n", lua_tostring(L, -1));
        return 1;
    }
    if (scene_path) snprintf(path, sizeof(path), "%s", scene_path);
    else snprintf(path, sizeof(path), "%s/%s", lua_dir, scene); /* 1248 */
    if (luaL_loadfile(L, path) || lua_pcall(L, 0, 0, 0)) {
        fprintf(stderr, "%s\n", lua_tostring(L, -1));
        return 1;
    }
    fprintf(g_out, "{\"ev\": \"scene\", \"file\": ");
    jstr(scene);
    fprintf(g_out, ", \"frames\": %d, \"keys\": ", frames);
    jstr(keys);
    fprintf(g_out, "}\n");
    lua_getglobal(L, "setup"); /* 1269 */
    if (!lua_isfunction(L, -1) || call_global(L, "setup", 0)) return 1;
    fprintf(g_out, "{\"ev\": \"setup_done\"}\n");
    for (g_frame = 1; g_frame <= frames; g_frame++) {
        lua_getglobal(L, "draw"); /* 978 */
        if (!lua_isfunction(L, -1) || call_global(L, "draw", 0)) return 1;
        /* nwm_poll_events (1315): the key events scripted for this frame */
        const char *p = keys;
        while (*p) {
            int f = 0, key = 0, mods = 0;
            if (sscanf(p, "%d:%d:%d", &f, &key, &mods) == 3 && f == g_frame) {
                fprintf(g_out, "{\"ev\": \"key\", \"key\": %d, \"mods\": %d}\n", key, mods);
                lua_getglobal(L, "on_key"); /* 1074-1078 */
                if (lua_isfunction(L, -1)) {
                    lua_pushinteger(L, key);
                    lua_pushinteger(L, mods);
                    if (call_global(L, "on_key", 2)) return 1;
                } else {
                    lua_pop(L, 1);
                }
            }
            p = strchr(p, ',');
            if (!p) break;
            p++;
        }
        lua_gc(L, LUA_GCCOLLECT, 0); /* 1316 */
        /* the replay device moves on by one block per rendered frame (src/nrf.c:153-170 at the frame rate) */
        for (int k = 0; k < g_n_devices; k++) {
            tdevice *d = g_devices[k];
            d->index = (d->index + 1) % d->n_blocks;
            device_ingest(d);
        }
        fprintf(g_out, "{\"ev\": \"frame\", \"n\": %d}\n", g_frame);
    }
    lua_close(L);
    if (g_out != stdout) fclose(g_out);
    return 0;
}
