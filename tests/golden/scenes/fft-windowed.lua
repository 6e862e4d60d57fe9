-- fft-windowed.lua -- THIS REPOSITORY'S scene, not one of the reference's: what a scene looks like that chooses its own
-- taper through nrf_fft_set_window (include/nrf.h; the binding of INTEGRATION.md section 1).  Two spectrum histories of one
-- device: `fft` starts with a Hann taper, `plain` stays rectangular (the reference's frames).  Keys: H = Hann,
-- B = Blackman-Harris, F = flat top, R = rectangular for `fft`; the frequency keys of _keys.lua work as in the
-- reference's scenes.  Traced by tests/golden/make_lua_traces.py, replayed on libfsea_nrf.so by tests/test_gpu_parity.py.

local KEY_B, KEY_F, KEY_H, KEY_R = 66, 70, 72, 82

function setup()
    freq = 97
    device = nrf_device_new(freq, "../rfdata/rf-200.500-big.raw")
    fft = nrf_fft_new(1024, 8)
    plain = nrf_fft_new(1024, 8)
    nrf_fft_set_window(fft, "hann")
    shader = ngl_shader_new(GL_TRIANGLES, "", "")
    texture = ngl_texture_new(shader, "uTexture")
end

function draw()
    samples_buffer = nrf_device_get_samples_buffer(device)
    nrf_fft_process(fft, samples_buffer)
    nrf_fft_process(plain, samples_buffer)
    fft_buffer = nrf_fft_get_buffer(fft)
    plain_buffer = nrf_fft_get_buffer(plain)
    ngl_texture_update(texture, fft_buffer, 1024, 8)
end

function on_key(key, mods)
    if key == KEY_H then nrf_fft_set_window(fft, "hann")
    elseif key == KEY_B then nrf_fft_set_window(fft, "blackmanharris")
    elseif key == KEY_F then nrf_fft_set_window(fft, "flattop")
    elseif key == KEY_R then nrf_fft_set_window(fft, "rect")
    else keys_frequency_handler(key, mods) end
end
